"""nbody_amd -- Python binding of libnbody_mi355.so (ctypes over the C-ABI in include/nbody_c.h).

`Simulation` mirrors the reference's `Simulation<float>` class (reference:
include/simulation.hpp:11-47): a `Params` record with the same fields, `spawn_uniform()`,
`spawn_sphere()`, `spawn_disk()`, `update()`, `get_num_octree_nodes()`,
`write_validation_history()`.  Everything is computed by the HIP library; there is no CPU
fallback: importing works without a GPU, any compute call without the library raises.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NBODY_AMD_LIB") or os.path.join(_HERE, "libnbody_mi355.so")

OK = 0
ERR_CAPACITY = 6
FLAG_TIMING = 1 << 0
FLAG_NO_FUSED_VEL = 1 << 1
FLAG_TIMING_STAGES = 1 << 2
FLAG_FP64 = 1 << 3
FLAG_FIT_KEYS = 1 << 4
FLAG_NO_UNIFORM_MASS = 1 << 5
FLAG_NO_COHERENT_SORT = 1 << 6
FLAG_ALWAYS_COHERENT_SORT = 1 << 11
FLAG_DEBUG_WALK_ONLY = 1 << 8
FLAG_DEBUG_SMALL_STACK = 1 << 9
FLAG_DEBUG_LATE_TILES = 1 << 10

IC_UNIFORM, IC_SPHERE, IC_DISK, IC_PLUMMER = 0, 1, 2, 3


class CParams(C.Structure):
    """struct nbody_params (include/nbody_c.h); first 11 fields == Simulation<T>::Params."""
    _fields_ = [
        ("num_points", C.c_int32),
        ("max_num_codes_per_leaf", C.c_int32),
        ("traversal_group_size", C.c_int32),
        ("mem_traversal_queues", C.c_uint64),
        ("theta", C.c_float),
        ("dt", C.c_float),
        ("gravity", C.c_float),
        ("softening_factor", C.c_float),
        ("domain_size", C.c_float),
        ("velocity_dampening", C.c_float),
        ("num_steps_validator", C.c_int32),
        ("morton_bits", C.c_int32),
        ("device_id", C.c_int32),
        ("flags", C.c_uint32),
        ("seed", C.c_uint64),
    ]


class CDevView(C.Structure):
    _fields_ = [("n", C.c_int32)] + [(k, C.c_void_p) for k in
                                     ("x", "y", "z", "vx", "vy", "vz", "ax", "ay", "az", "m",
                                      "posm", "id", "stream")]


_OCT_FIELDS = [
    ("codes_sorted", np.uint64), ("perm", np.int32), ("unique", np.uint64),
    ("first_pt", np.int32), ("leaf_first_code_idx", np.int32), ("first_child", np.int32),
    ("num_children", np.int32), ("leaves_begin", np.int32), ("leaves_end", np.int32),
    ("depth", np.int32), ("level", np.int32), ("size", np.float32), ("points_begin", np.int32),
    ("points_end", np.int32), ("parent", np.int32), ("comx", np.float32), ("comy", np.float32),
    ("comz", np.float32), ("mass", np.float32), ("list_len", np.uint32),
]


class COctreeView(C.Structure):
    _fields_ = [(k, C.c_void_p) for k, _ in _OCT_FIELDS]


class CComm(C.Structure):
    """struct nbody_comm: the collectives nbody_dd_* calls back into."""
    ALL_GATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
    ALL_TO_ALL = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t),
                             C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_size_t),
                             C.POINTER(C.c_size_t), C.c_void_p)
    _fields_ = [("rank", C.c_int32), ("size", C.c_int32), ("ctx", C.c_void_p),
                ("all_gather", ALL_GATHER), ("all_to_all", ALL_TO_ALL)]


class CDDPhases(C.Structure):
    _fields_ = [(k, C.c_float) for k in ("ms_build", "ms_frontier", "ms_export", "ms_early_traversal",
                                         "ms_sizing", "ms_exchange", "ms_graft", "ms_traversal", "ms_total",
                                         "ms_host_wait", "ms_host_total")] + \
               [("pad", C.c_int32), ("bytes_frontier", C.c_int64), ("bytes_sent", C.c_int64),
                ("bytes_received", C.c_int64)]


class CDDStats(C.Structure):
    _fields_ = [("n_local", C.c_int64), ("records_sent", C.c_int64), ("records_received", C.c_int64),
                ("migrated_out", C.c_int64), ("frontier_records", C.c_int32),
                ("export_capacity", C.c_int32), ("nodes_sent", C.c_int64), ("nodes_received", C.c_int64)]


LET_REC = 12          # floats per frontier record: lo xyz, hi xyz, com xyz, mass, size, count
LET_MAX_RECORDS = 32  # frontier records per domain


class CTimings(C.Structure):
    _fields_ = [("steps", C.c_int32), ("ms_integrate", C.c_float), ("ms_sort", C.c_float),
                ("ms_build", C.c_float), ("ms_com", C.c_float), ("ms_traverse", C.c_float),
                ("ms_total", C.c_float), ("interactions", C.c_uint64)]


EXPORTS = [
    "nbody_default_params", "nbody_create", "nbody_create_external", "nbody_destroy", "nbody_last_error",
    "nbody_spawn_uniform", "nbody_spawn_sphere", "nbody_spawn_disk", "nbody_spawn_plummer",
    "nbody_sample_ic", "nbody_set_state", "nbody_update", "nbody_step", "nbody_sync",
    "nbody_get_num_octree_nodes", "nbody_write_validation_history", "nbody_get_state",
    "nbody_get_device_ptrs", "nbody_get_tree_counts", "nbody_get_octree", "nbody_get_timings",
    "nbody_direct_sum_dev", "nbody_direct_sum", "nbody_set_traversal_range", "nbody_finish_step",
    "nbody_get_validation_history", "nbody_get_bbox", "nbody_let_export", "nbody_set_sources_dev",
    "nbody_traverse_from", "nbody_save_snapshot", "nbody_load_snapshot", "nbody_set_particles",
    "nbody_tree_energy", "nbody_set_state_f64", "nbody_get_state_f64", "nbody_sample_ic_f64",
    "nbody_get_octree_com_f64", "nbody_direct_sum_f64", "nbody_get_device_ptrs_f64",
    "nbody_direct_energy_dev", "nbody_direct_energy_dev_noalloc",
    "nbody_let_frontier", "nbody_let_frontier_dev", "nbody_let_export_dev", "nbody_sample_ic_range",
    "nbody_dd_create", "nbody_dd_destroy", "nbody_dd_last_error", "nbody_dd_spawn",
    "nbody_dd_rebalance", "nbody_dd_step", "nbody_dd_get_stats", "nbody_dd_choose_edges",
    "nbody_dd_refine_edge", "nbody_get_key_domain",
    "nbody_let_export_tree_dev", "nbody_import_trees_dev", "nbody_traverse_forest",
    "nbody_dd_set_phase_timing", "nbody_dd_get_phases", "nbody_get_sort_stats",
]

_lib = None


def load_library() -> C.CDLL:
    """Loads libnbody_mi355.so (built in-tree by `make -C nbody_amd/csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `make -C nbody_amd/csrc` "
            "(or python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
    # PyTorch-ROCm wheels bundle their own ROCm runtime.  If this library (linked against the
    # system runtime) is loaded first and torch afterwards, the process ends up with two
    # runtimes and torch sees no GPU; loaded in the other order both share torch's copy.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    fp = C.POINTER(C.c_float)
    lib.nbody_default_params.argtypes = [C.POINTER(CParams)]
    lib.nbody_default_params.restype = None
    lib.nbody_create.argtypes = [C.POINTER(CParams), C.POINTER(C.c_void_p)]
    lib.nbody_create_external.argtypes = [C.POINTER(CParams), C.POINTER(C.c_void_p),
                                          C.POINTER(C.c_void_p)]
    lib.nbody_destroy.argtypes = [C.c_void_p]
    lib.nbody_destroy.restype = None
    lib.nbody_last_error.argtypes = [C.c_void_p]
    lib.nbody_last_error.restype = C.c_char_p
    for name in ("nbody_spawn_uniform", "nbody_spawn_sphere", "nbody_spawn_disk", "nbody_update",
                 "nbody_sync", "nbody_finish_step"):
        getattr(lib, name).argtypes = [C.c_void_p]
    lib.nbody_spawn_plummer.argtypes = [C.c_void_p, C.c_float, C.c_float]
    lib.nbody_sample_ic.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_float, C.c_float, C.c_float,
                                    C.c_float] + [fp] * 7
    lib.nbody_set_state.argtypes = [C.c_void_p] + [fp] * 7
    lib.nbody_step.argtypes = [C.c_void_p, C.c_int]
    lib.nbody_get_num_octree_nodes.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    lib.nbody_write_validation_history.argtypes = [C.c_void_p, C.c_char_p]
    lib.nbody_get_state.argtypes = [C.c_void_p] + [fp] * 10 + [C.POINTER(C.c_uint32)]
    lib.nbody_get_device_ptrs.argtypes = [C.c_void_p, C.POINTER(CDevView)]
    lib.nbody_get_tree_counts.argtypes = [C.c_void_p] + [C.POINTER(C.c_int32)] * 3
    lib.nbody_get_octree.argtypes = [C.c_void_p, C.POINTER(COctreeView)]
    lib.nbody_get_timings.argtypes = [C.c_void_p, C.POINTER(CTimings)]
    lib.nbody_direct_sum_dev.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_float,
                                         C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                         C.c_void_p]
    lib.nbody_direct_sum.argtypes = [C.c_void_p] + [fp] * 3
    lib.nbody_set_traversal_range.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    dp = C.POINTER(C.c_double)
    lib.nbody_get_validation_history.argtypes = [C.c_void_p, dp, dp, dp, C.c_int32,
                                                 C.POINTER(C.c_int32)]
    lib.nbody_tree_energy.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.nbody_save_snapshot.argtypes = [C.c_void_p, C.c_char_p]
    lib.nbody_load_snapshot.argtypes = [C.c_void_p, C.c_char_p]
    lib.nbody_set_particles.argtypes = [C.c_void_p, C.c_int32] + [fp] * 10 + [C.POINTER(C.c_uint32)]
    lib.nbody_get_bbox.argtypes = [C.c_void_p, fp]
    lib.nbody_let_export.argtypes = [C.c_void_p, fp, C.c_int32, C.c_void_p, C.c_int32,
                                     C.POINTER(C.c_int32)]
    lib.nbody_set_sources_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    lib.nbody_traverse_from.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    lib.nbody_set_state_f64.argtypes = [C.c_void_p] + [dp] * 7
    lib.nbody_get_state_f64.argtypes = [C.c_void_p] + [dp] * 10 + [C.POINTER(C.c_uint32)]
    lib.nbody_sample_ic_f64.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_float, C.c_float, C.c_float,
                                        C.c_float] + [dp] * 7
    lib.nbody_get_octree_com_f64.argtypes = [C.c_void_p] + [dp] * 4
    lib.nbody_direct_sum_f64.argtypes = [C.c_void_p] + [dp] * 3
    lib.nbody_get_device_ptrs_f64.argtypes = [C.c_void_p, C.POINTER(CDevView)]
    lib.nbody_direct_energy_dev.argtypes = [C.c_void_p] * 4 + [C.c_int32, C.c_float, C.c_float,
                                                              C.c_void_p, C.c_void_p]
    lib.nbody_direct_energy_dev_noalloc.argtypes = lib.nbody_direct_energy_dev.argtypes
    lib.nbody_let_frontier.argtypes = [C.c_void_p, fp, C.c_int32]
    lib.nbody_let_frontier_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    lib.nbody_let_export_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_void_p, C.c_int32, C.c_void_p]
    lib.nbody_let_export_tree_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                              C.c_void_p, C.c_size_t, C.c_void_p]
    lib.nbody_import_trees_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_size_t),
                                           C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.nbody_traverse_forest.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    lib.nbody_sample_ic_range.argtypes = [C.c_int, C.c_int64, C.c_int32, C.c_int64, C.c_uint64,
                                          C.c_float, C.c_float, C.c_float, C.c_float] + [fp] * 7
    lib.nbody_dd_create.argtypes = [C.c_void_p, C.POINTER(CComm), C.POINTER(C.c_void_p)]
    lib.nbody_dd_destroy.argtypes = [C.c_void_p]
    lib.nbody_dd_destroy.restype = None
    lib.nbody_dd_last_error.argtypes = [C.c_void_p]
    lib.nbody_dd_last_error.restype = C.c_char_p
    lib.nbody_dd_spawn.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_float]
    lib.nbody_dd_rebalance.argtypes = [C.c_void_p]
    lib.nbody_dd_step.argtypes = [C.c_void_p]
    lib.nbody_dd_get_stats.argtypes = [C.c_void_p, C.POINTER(CDDStats), C.POINTER(C.c_uint64)]
    lib.nbody_dd_set_phase_timing.argtypes = [C.c_void_p, C.c_int32]
    lib.nbody_dd_get_phases.argtypes = [C.c_void_p, C.POINTER(CDDPhases)]
    lib.nbody_dd_choose_edges.argtypes = [C.POINTER(C.c_uint64), C.c_int32, C.c_int32, C.c_uint64,
                                          C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.nbody_dd_choose_edges.restype = None
    lib.nbody_dd_refine_edge.argtypes = [C.POINTER(C.c_uint64), C.c_int32, C.c_uint64, C.c_uint64,
                                         C.c_uint64, C.c_uint64, C.c_int32]
    lib.nbody_dd_refine_edge.restype = C.c_int32
    lib.nbody_get_key_domain.argtypes = [C.c_void_p, fp]
    lib.nbody_get_sort_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                         C.POINTER(C.c_int32)]
    _lib = lib
    return lib


def _fptr(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def _dptr(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def sample_ic(kind: int, n: int, seed: int = 42, domain_size: float = 5.0, gravity: float = 1.0,
              p0: float = 0.0, p1: float = 0.0, dtype=np.float32):
    """Host-side initial-condition sampler of the library: dict of float32 (or float64) arrays."""
    lib = load_library()
    f64 = np.dtype(dtype) == np.float64
    out = {k: np.empty(n, np.float64 if f64 else np.float32) for k in ("x", "y", "z", "vx", "vy", "vz", "m")}
    fn, ptr = (lib.nbody_sample_ic_f64, _dptr) if f64 else (lib.nbody_sample_ic, _fptr)
    rc = fn(kind, n, seed, domain_size, gravity, p0, p1,
            *[ptr(out[k]) for k in ("x", "y", "z", "vx", "vy", "vz", "m")])
    if rc != OK:
        raise RuntimeError(f"nbody_sample_ic failed with status {rc}")
    return out


def sample_ic_range(kind: int, first: int, count: int, n_total: int, seed: int = 42,
                    domain_size: float = 5.0, gravity: float = 1.0, p0: float = 0.0, p1: float = 0.0):
    """Particles [first, first + count) of the n_total-particle set sample_ic(kind, n_total, seed)
    returns (float32): a particle's draws depend on its index in the whole set only."""
    lib = load_library()
    out = {k: np.empty(count, np.float32) for k in ("x", "y", "z", "vx", "vy", "vz", "m")}
    rc = lib.nbody_sample_ic_range(kind, first, count, n_total, seed, domain_size, gravity, p0, p1,
                                   *[_fptr(out[k]) for k in ("x", "y", "z", "vx", "vy", "vz", "m")])
    if rc != OK:
        raise RuntimeError(f"nbody_sample_ic_range failed with status {rc}")
    return out


def choose_edges(hist, parts: int, slack: int):
    """nbody_dd_choose_edges: (edge, bin) arrays of parts - 1 entries for a histogram of
    particle counts per key-ordered bin (host code of the library, no GPU involved)."""
    lib = load_library()
    h = np.ascontiguousarray(hist, np.uint64)
    edge = np.empty(max(parts - 1, 0), np.int32)
    bin_ = np.empty(max(parts - 1, 0), np.int32)
    lib.nbody_dd_choose_edges(h.ctypes.data_as(C.POINTER(C.c_uint64)), len(h), parts, int(slack),
                              edge.ctypes.data_as(C.POINTER(C.c_int32)),
                              bin_.ctypes.data_as(C.POINTER(C.c_int32)))
    return edge, bin_


def refine_edge(hist, base_count: int, target: int, slack: int, base_edge: int, edge_bits: int) -> int:
    """nbody_dd_refine_edge (host code of the library)."""
    h = np.ascontiguousarray(hist, np.uint64)
    return int(load_library().nbody_dd_refine_edge(h.ctypes.data_as(C.POINTER(C.c_uint64)), len(h),
                                                   int(base_count), int(target), int(slack),
                                                   int(base_edge), int(edge_bits)))


@dataclass
class Params:
    """Simulation<T>::Params (reference: include/simulation.hpp:15-28) + extensions.
    Defaults are the reference renderer's (src/renderer.cpp:297-321, disk dt/softening)."""
    num_points: int = 1 << 18
    max_num_codes_per_leaf: int = 32
    traversal_group_size: int = 64
    mem_traversal_queues: int = (1 << 20) * 50
    theta: float = 0.75
    dt: float = 1e-4
    gravity: float = 1.0
    softening_factor: float = 0.001
    domain_size: float = 5.0
    velocity_dampening: float = 0.0
    num_steps_validator: int = 0
    morton_bits: int = 10
    device_id: int = -1
    flags: int = 0
    seed: int = 42

    def to_c(self) -> CParams:
        return CParams(**{k: getattr(self, k) for k, _ in CParams._fields_})


class Simulation:
    """Headless drop-in for the reference's Simulation<float> (include/simulation.hpp); with
    FLAG_FP64 in params.flags, for its Simulation<double> (src/cuda/simulation.cu:256)."""

    def __init__(self, params: Params, external=None):
        """external: optional 7 device pointers (ints, 0 = skip) [pos.x, pos.y, pos.z, com.x,
        com.y, com.z, node_size] -- the reference's Simulation(Params&, GLuint[7])."""
        self._lib = load_library()
        self.params = params
        self._h = C.c_void_p()
        cp = params.to_c()
        if external is None:
            rc = self._lib.nbody_create(C.byref(cp), C.byref(self._h))
        else:
            assert len(external) == 7
            arr = (C.c_void_p * 7)(*[int(p) or None for p in external])
            rc = self._lib.nbody_create_external(C.byref(cp), arr, C.byref(self._h))
        if rc != OK:
            self._h = C.c_void_p()
            raise RuntimeError(f"nbody_create failed with status {rc}")

    # -- lifetime -----------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.nbody_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc: int):
        if rc != OK:
            msg = self._lib.nbody_last_error(self._h)
            raise RuntimeError(f"nbody status {rc}: {msg.decode() if msg else ''}")

    # -- the reference's surface ---------------------------------------------
    def spawn_uniform(self):
        self._check(self._lib.nbody_spawn_uniform(self._h))

    def spawn_sphere(self):
        self._check(self._lib.nbody_spawn_sphere(self._h))

    def spawn_disk(self):
        self._check(self._lib.nbody_spawn_disk(self._h))

    def spawn_plummer(self, a: float = 0.1, r_max: float = 1.0):
        self._check(self._lib.nbody_spawn_plummer(self._h, a, r_max))

    def update(self):
        self._check(self._lib.nbody_update(self._h))

    def step(self, n: int):
        self._check(self._lib.nbody_step(self._h, n))

    def get_num_octree_nodes(self) -> int:
        out = C.c_int32()
        self._check(self._lib.nbody_get_num_octree_nodes(self._h, C.byref(out)))
        return out.value

    def write_validation_history(self, csv_file_path: str):
        self._check(self._lib.nbody_write_validation_history(self._h, csv_file_path.encode()))

    def validation_history(self) -> dict:
        """Arrays sys_energy, sys_energy_ap, avg_acc_err (one entry per validated step)."""
        cap = max(self.params.num_steps_validator, 1)
        e, ea, er = (np.zeros(cap, np.float64) for _ in range(3))
        cnt = C.c_int32()
        dp = C.POINTER(C.c_double)
        self._check(self._lib.nbody_get_validation_history(
            self._h, e.ctypes.data_as(dp), ea.ctypes.data_as(dp), er.ctypes.data_as(dp), cap,
            C.byref(cnt)))
        k = cnt.value
        return {"sys_energy": e[:k], "sys_energy_ap": ea[:k], "avg_acc_err": er[:k]}

    # -- additions -----------------------------------------------------------
    def sync(self):
        self._check(self._lib.nbody_sync(self._h))

    @property
    def fp64(self) -> bool:
        return bool(self.params.flags & FLAG_FP64)

    def set_state(self, x, y, z, m, vx=None, vy=None, vz=None, dtype=None):
        """dtype: precision the arrays are handed over in (default: the handle's own)."""
        n = self.params.num_points
        f64 = self.fp64 if dtype is None else np.dtype(dtype) == np.float64
        arrs = []
        for a in (x, y, z, vx, vy, vz, m):
            if a is None:
                arrs.append(None)
            else:
                a = np.ascontiguousarray(a, np.float64 if f64 else np.float32)
                assert a.shape == (n,)
                arrs.append(a)
        if f64:
            self._check(self._lib.nbody_set_state_f64(self._h, *[_dptr(a) for a in arrs]))
        else:
            self._check(self._lib.nbody_set_state(self._h, *[_fptr(a) for a in arrs]))

    def count(self) -> int:
        """Particles currently held (num_points unless set_particles/set_sources changed it)."""
        return int(self.device_view().n)

    def set_particles(self, x, y, z, m, vx=None, vy=None, vz=None, ax=None, ay=None, az=None, id=None):
        """Variable-count set_state that keeps accelerations and ids (domain re-balancing)."""
        n = len(x)
        arrs = [None if a is None else np.ascontiguousarray(a, np.float32)
                for a in (x, y, z, vx, vy, vz, m, ax, ay, az)]
        assert all(a is None or a.shape == (n,) for a in arrs)
        ids = None if id is None else np.ascontiguousarray(id, np.uint32)
        self._check(self._lib.nbody_set_particles(
            self._h, n, *[_fptr(a) for a in arrs],
            None if ids is None else ids.ctypes.data_as(C.POINTER(C.c_uint32))))

    def get_state(self, dtype=None) -> dict:
        """dtype: precision of the returned arrays (default: the handle's own)."""
        n = self.count()
        f64 = self.fp64 if dtype is None else np.dtype(dtype) == np.float64
        names = ("x", "y", "z", "vx", "vy", "vz", "m", "ax", "ay", "az")
        out = {k: np.empty(n, np.float64 if f64 else np.float32) for k in names}
        out["id"] = np.empty(n, np.uint32)
        fn, ptr = (self._lib.nbody_get_state_f64, _dptr) if f64 else (self._lib.nbody_get_state, _fptr)
        self._check(fn(self._h, *[ptr(out[k]) for k in names],
                       out["id"].ctypes.data_as(C.POINTER(C.c_uint32))))
        return out

    def key_domain(self) -> np.ndarray:
        """(lo x, lo y, lo z, side) of the cube the Morton keys are taken relative to."""
        out = np.empty(4, np.float32)
        self._check(self._lib.nbody_get_key_domain(self._h, _fptr(out)))
        return out

    def sort_stats(self) -> dict:
        """Sorts that took the nearly-sorted path, sorts by digit passes over everything, and the
        elements found out of place by the last sort that counted them."""
        a, b, m = C.c_uint64(), C.c_uint64(), C.c_int32()
        self._check(self._lib.nbody_get_sort_stats(self._h, C.byref(a), C.byref(b), C.byref(m)))
        return {"coherent_sorts": a.value, "full_sorts": b.value, "last_moved": m.value}

    def tree_counts(self):
        u, m, no = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self._lib.nbody_get_tree_counts(self._h, C.byref(u), C.byref(m), C.byref(no)))
        return u.value, m.value, no.value

    def get_octree(self) -> dict:
        """All build products of the last update/spawn as numpy arrays."""
        n = self.count()
        U, M, no = self.tree_counts()
        ngroups = (n + self.params.traversal_group_size - 1) // self.params.traversal_group_size
        sizes = {"codes_sorted": n, "perm": n, "unique": U, "first_pt": U + 1,
                 "leaf_first_code_idx": M + 1, "list_len": ngroups}
        out, view = {}, COctreeView()
        for k, dt in _OCT_FIELDS:
            out[k] = np.empty(sizes.get(k, no), dt)
            setattr(view, k, out[k].ctypes.data)
        self._check(self._lib.nbody_get_octree(self._h, C.byref(view)))
        if self.fp64:   # the exact centres of mass instead of the narrowed copies
            for k in ("comx", "comy", "comz", "mass"):
                out[k] = np.empty(no, np.float64)
            self._check(self._lib.nbody_get_octree_com_f64(
                self._h, *[_dptr(out[k]) for k in ("comx", "comy", "comz", "mass")]))
        out["U"], out["M"], out["n_oct"] = U, M, no
        return out

    def get_list_len(self) -> np.ndarray:
        """Per group: number of sources of the last traversal (nbody_octree_view.list_len only)."""
        n = self.count()
        g = self.params.traversal_group_size
        out = np.empty((n + g - 1) // g, np.uint32)
        view = COctreeView()
        view.list_len = out.ctypes.data
        self._check(self._lib.nbody_get_octree(self._h, C.byref(view)))
        return out

    def get_timings(self) -> dict:
        t = CTimings()
        self._check(self._lib.nbody_get_timings(self._h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in CTimings._fields_}

    def device_view(self) -> CDevView:
        v = CDevView()
        self._check(self._lib.nbody_get_device_ptrs(self._h, C.byref(v)))
        return v

    def device_view_f64(self) -> CDevView:
        """The double arrays of a handle with FLAG_FP64 (same field names; posm = double4)."""
        v = CDevView()
        self._check(self._lib.nbody_get_device_ptrs_f64(self._h, C.byref(v)))
        return v

    def direct_sum(self, dtype=None):
        """dtype float64: the sum evaluated in double (on a float handle over its float state)."""
        n = self.count()
        f64 = self.fp64 if dtype is None else np.dtype(dtype) == np.float64
        out = [np.empty(n, np.float64 if f64 else np.float32) for _ in range(3)]
        if f64:
            self._check(self._lib.nbody_direct_sum_f64(self._h, *[_dptr(a) for a in out]))
        else:
            self._check(self._lib.nbody_direct_sum(self._h, *[_fptr(a) for a in out]))
        return out

    def set_traversal_range(self, g0: int, g1: int):
        self._check(self._lib.nbody_set_traversal_range(self._h, g0, g1))

    def finish_step(self):
        self._check(self._lib.nbody_finish_step(self._h))

    def tree_energy(self):
        """(kinetic, potential) of the current state, potential through the tree."""
        ke, pe = C.c_double(), C.c_double()
        self._check(self._lib.nbody_tree_energy(self._h, C.byref(ke), C.byref(pe)))
        return ke.value, pe.value

    def save_snapshot(self, path: str):
        self._check(self._lib.nbody_save_snapshot(self._h, os.fsencode(path)))

    def load_snapshot(self, path: str):
        self._check(self._lib.nbody_load_snapshot(self._h, os.fsencode(path)))

    # -- multi-GPU locally-essential-tree exchange ------------------------------
    def get_bbox(self) -> np.ndarray:
        """(lo_x, lo_y, lo_z, hi_x, hi_y, hi_z) of the current positions."""
        out = np.empty(6, np.float32)
        self._check(self._lib.nbody_get_bbox(self._h, _fptr(out)))
        return out

    def let_frontier(self, max_records: int = LET_MAX_RECORDS) -> np.ndarray:
        """(max_records, 12) float32: the top of the tree -- at most max_records nodes that hold
        every particle once: lo xyz, hi xyz of the node's particles, centre of mass xyz, mass,
        cell size, particle count; unused records are empty boxes (lo = 3e38)."""
        out = np.empty((max_records, LET_REC), np.float32)
        self._check(self._lib.nbody_let_frontier(self._h, _fptr(out), max_records))
        return out

    def let_frontier_dev(self, records_dev_ptr: int, max_records: int = LET_MAX_RECORDS):
        self._check(self._lib.nbody_let_frontier_dev(self._h, records_dev_ptr, max_records))

    def let_export_dev(self, frontiers_dev_ptr: int, ndomains: int, records_per_domain: int,
                       skip_domain: int, out_dev_ptr: int, capacity: int, counts_dev_ptr: int):
        self._check(self._lib.nbody_let_export_dev(self._h, frontiers_dev_ptr, ndomains,
                                                   records_per_domain, skip_domain, out_dev_ptr,
                                                   capacity, counts_dev_ptr))

    def let_export(self, boxes, out_dev_ptr: int, capacity: int) -> np.ndarray:
        """Writes the records every box needs to out_dev_ptr + 16*capacity*b; returns the
        per-box record counts.  Raises CapacityError (with .counts) if one exceeds capacity."""
        boxes = np.ascontiguousarray(boxes, np.float32).reshape(-1, 6)
        counts = np.zeros(len(boxes), np.int32)
        rc = self._lib.nbody_let_export(self._h, _fptr(boxes), len(boxes), out_dev_ptr, capacity,
                                        counts.ctypes.data_as(C.POINTER(C.c_int32)))
        if rc == ERR_CAPACITY:
            raise CapacityError(counts)
        self._check(rc)
        return counts

    def let_export_tree_dev(self, frontiers_dev_ptr: int, ndomains: int, records_per_domain: int,
                            skip_domain: int, out_dev_ptr: int, capacity_bytes: int, counts_dev_ptr: int):
        """Pruned copies of this handle's tree, one per remote domain (nbody_let_export_tree_dev):
        counts_dev = 2 int32 per domain (node records of 32 bytes, then particle records of 16)."""
        self._check(self._lib.nbody_let_export_tree_dev(self._h, frontiers_dev_ptr, ndomains,
                                                        records_per_domain, skip_domain, out_dev_ptr,
                                                        capacity_bytes, counts_dev_ptr))

    def import_trees_dev(self, recv_dev_ptr: int, byte_offsets, node_counts, record_counts):
        """Hangs the given tree copies (device memory) behind the own tree until the next update()."""
        k = len(node_counts)
        off = (C.c_size_t * max(k, 1))(*[int(v) for v in byte_offsets])
        nn = (C.c_int32 * max(k, 1))(*[int(v) for v in node_counts])
        nr = (C.c_int32 * max(k, 1))(*[int(v) for v in record_counts])
        self._check(self._lib.nbody_import_trees_dev(self._h, recv_dev_ptr, k, off, nn, nr))

    def traverse_forest(self, what: int = 1, group_begin: int = 0, group_end: int = -1):
        """Forces on groups [group_begin, group_end) (-1: all) from the own tree (what=0), the own
        tree and the imported copies in one walk (1), or the copies alone, added (2); no kick."""
        self._check(self._lib.nbody_traverse_forest(self._h, what, group_begin, group_end))

    def set_sources_dev(self, posm_dev_ptr: int, count: int):
        self._check(self._lib.nbody_set_sources_dev(self._h, posm_dev_ptr, count))

    def traverse_from(self, sources: "Simulation", accumulate: bool = True):
        self._check(self._lib.nbody_traverse_from(self._h, sources._h, 1 if accumulate else 0))


class CapacityError(RuntimeError):
    def __init__(self, counts):
        super().__init__(f"record buffer too small: need {int(max(counts))} per box")
        self.counts = counts
