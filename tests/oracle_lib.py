"""ctypes binding of oracle/liboracle.so -- the host restatement of the reference algorithm.

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and the cpu_baseline
leg of bench.py, never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")

_lib = None


def build():
    subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    L = C.CDLL(LIB_PATH)
    L.orc_unique.restype = C.c_int
    L.orc_coarsen.restype = C.c_int
    L.orc_build_octree.restype = C.c_int
    L.orc_energy_f64.restype = C.c_double
    L.orc_sim_create.restype = C.c_void_p
    L.orc_num_threads.restype = C.c_int
    L.orc_let_export.restype = C.c_int
    L.orc_sim_create_dp.restype = C.c_void_p
    L.orc_sim_create_fit.restype = C.c_void_p
    _lib = L
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


def morton(x, y, z, domain, bits):
    x, y, z = _f32(x), _f32(y), _f32(z)
    codes = np.empty(x.size, np.uint64)
    lib().orc_morton(_p(x), _p(y), _p(z), C.c_int(x.size), C.c_float(domain), C.c_int(bits), _p(codes))
    return codes


def sort(codes):
    codes = np.ascontiguousarray(codes, np.uint64)
    out = np.empty_like(codes)
    perm = np.empty(codes.size, np.int32)
    lib().orc_sort(_p(codes), C.c_int(codes.size), _p(out), _p(perm))
    return out, perm


def unique(sorted_codes):
    s = np.ascontiguousarray(sorted_codes, np.uint64)
    uq = np.empty(s.size, np.uint64)
    fp = np.empty(s.size + 1, np.int32)
    U = lib().orc_unique(_p(s), C.c_int(s.size), _p(uq), _p(fp))
    return uq[:U].copy(), fp[:U + 1].copy()


def coarsen(uq, bits, max_codes):
    uq = np.ascontiguousarray(uq, np.uint64)
    lfc = np.empty(uq.size + 1, np.int32)
    M = lib().orc_coarsen(_p(uq), C.c_int(uq.size), C.c_int(bits), C.c_int(max_codes), _p(lfc))
    return lfc[:M + 1].copy()


def build_octree(uq, lfc, first_pt, bits, domain):
    uq = np.ascontiguousarray(uq, np.uint64)
    lfc = np.ascontiguousarray(lfc, np.int32)
    first_pt = np.ascontiguousarray(first_pt, np.int32)
    M = lfc.size - 1
    cap = max(2 * M - 1, 1)
    names_i = ["first_child", "num_children", "leaves_begin", "leaves_end", "depth", "level"]
    out = {k: np.empty(cap, np.int32) for k in names_i}
    out["size"] = np.empty(cap, np.float32)
    out["points_begin"] = np.empty(cap, np.int32)
    out["points_end"] = np.empty(cap, np.int32)
    n_oct = lib().orc_build_octree(
        _p(uq), C.c_int(uq.size), _p(lfc), C.c_int(M), _p(first_pt), C.c_int(bits),
        C.c_float(domain), _p(out["first_child"]), _p(out["num_children"]),
        _p(out["leaves_begin"]), _p(out["leaves_end"]), _p(out["depth"]), _p(out["level"]),
        _p(out["size"]), _p(out["points_begin"]), _p(out["points_end"]))
    out = {k: v[:n_oct].copy() for k, v in out.items()}
    out["n_oct"] = n_oct
    return out


def com(tree, x, y, z, m):
    n_oct = tree["n_oct"]
    x, y, z, m = _f32(x), _f32(y), _f32(z), _f32(m)
    out = {k: np.empty(n_oct, np.float32) for k in ("comx", "comy", "comz", "mass")}
    lib().orc_com(C.c_int(n_oct), _p(tree["first_child"]), _p(tree["num_children"]),
                  _p(tree["points_begin"]), _p(tree["points_end"]), _p(x), _p(y), _p(z), _p(m),
                  _p(out["comx"]), _p(out["comy"]), _p(out["comz"]), _p(out["mass"]))
    return out


def traverse(tree, comd, x, y, z, m, group, theta, eps, gravity, domain):
    x, y, z, m = _f32(x), _f32(y), _f32(z), _f32(m)
    n = x.size
    ax, ay, az = (np.empty(n, np.float32) for _ in range(3))
    ll = np.empty((n + group - 1) // group, np.uint32)
    a = {k: np.ascontiguousarray(comd[k], np.float32) for k in ("comx", "comy", "comz", "mass")}
    size = np.ascontiguousarray(tree["size"], np.float32)
    lib().orc_traverse(C.c_int(n), C.c_int(group), _p(x), _p(y), _p(z), _p(m),
                       C.c_int(tree["n_oct"]), _p(np.ascontiguousarray(tree["first_child"], np.int32)),
                       _p(np.ascontiguousarray(tree["num_children"], np.int32)),
                       _p(np.ascontiguousarray(tree["points_begin"], np.int32)),
                       _p(np.ascontiguousarray(tree["points_end"], np.int32)), _p(size),
                       _p(a["comx"]), _p(a["comy"]), _p(a["comz"]), _p(a["mass"]),
                       C.c_float(theta), C.c_float(np.float32(eps) * np.float32(eps)),
                       C.c_float(gravity), C.c_float(domain), _p(ax), _p(ay), _p(az), _p(ll))
    return ax, ay, az, ll


def _tree_args(tree, comd):
    i32 = lambda k: _p(np.ascontiguousarray(tree[k], np.int32))
    a = [np.ascontiguousarray(comd[k], np.float32) for k in ("comx", "comy", "comz", "mass")]
    size = np.ascontiguousarray(tree["size"], np.float32)
    keep = a + [size]
    return [i32("first_child"), i32("num_children"), i32("points_begin"), i32("points_end"),
            _p(size)] + [_p(v) for v in a], keep


def traverse_from(bodies, tree, comd, x, y, z, m, group, theta, eps, gravity, domain, acc=None):
    """Accelerations of `bodies` = (bx, by, bz) due to the tree over the points (x, y, z, m);
    acc = (ax, ay, az) float32 arrays to add to (None: fresh result)."""
    bx, by, bz = (_f32(b) for b in bodies)
    x, y, z, m = _f32(x), _f32(y), _f32(z), _f32(m)
    nb = bx.size
    out = [np.zeros(nb, np.float32) for _ in range(3)] if acc is None else [_f32(a).copy() for a in acc]
    ll = np.empty((nb + group - 1) // group, np.uint32)
    targs, keep = _tree_args(tree, comd)
    lib().orc_traverse_from(C.c_int(nb), C.c_int(group), _p(bx), _p(by), _p(bz), _p(x), _p(y), _p(z),
                            _p(m), *targs, C.c_float(theta),
                            C.c_float(np.float32(eps) * np.float32(eps)), C.c_float(gravity),
                            C.c_float(domain), *[_p(o) for o in out], _p(ll),
                            C.c_int(0 if acc is None else 1))
    return out[0], out[1], out[2], ll


def let_export(tree, comd, x, y, z, m, box, theta):
    """(k, 4) float32 records a domain needs from this tree; `box` = the (6,) bounding box of
    the domain's particles or (b, 6) boxes whose union holds them (lo xyz, hi xyz each)."""
    x, y, z, m = _f32(x), _f32(y), _f32(z), _f32(m)
    box = _f32(box).reshape(-1, 6)
    targs, keep = _tree_args(tree, comd)
    cap = x.size + tree["n_oct"]
    out = np.empty((cap, 4), np.float32)
    k = lib().orc_let_export(*targs, _p(x), _p(y), _p(z), _p(m), _p(box), C.c_int(len(box)),
                             C.c_float(theta), _p(out), C.c_int(cap))
    assert k <= cap
    return out[:k].copy()


def frontier(tree, comd, xs, ys, zs, T):
    """Top-of-tree records of nbody_let_frontier, restated in numpy: expand the breadth-first
    prefix [0, K) of the octree, K the largest prefix that leaves at most T nodes; the frontier
    is every child of an expanded node that is not itself expanded plus the leaves among the
    expanded ones, ordered by first particle.  (T, 12) float32: lo xyz, hi xyz of the node's
    particles, centre of mass xyz, mass, cell size, particle count; unused records lo = 3e38."""
    nc, fc = tree["num_children"], tree["first_child"]
    n_oct = tree["n_oct"]
    L = min(n_oct, T)
    inc = np.where(nc[:L] > 0, nc[:L] - 1, 0)
    size = 1 + np.concatenate([[0], np.cumsum(inc)])        # frontier size after expanding [0, K)
    K = int(np.nonzero(size <= T)[0].max())
    expanded = [o for o in range(K) if nc[o] > 0]
    C = max((int(fc[o] + nc[o]) for o in expanded), default=1)
    nodes = [o for o in range(C) if o >= K or nc[o] == 0]
    nodes.sort(key=lambda o: int(tree["points_begin"][o]))
    assert len(nodes) <= T
    rec = np.zeros((T, 12), np.float32)
    rec[:, :3] = np.float32(3.0e38)
    for j, o in enumerate(nodes):
        pb, pe = int(tree["points_begin"][o]), int(tree["points_end"][o])
        for k, a in enumerate((xs, ys, zs)):
            rec[j, k] = a[pb:pe + 1].min()
            rec[j, 3 + k] = a[pb:pe + 1].max()
        rec[j, 6:10] = [comd["comx"][o], comd["comy"][o], comd["comz"][o], comd["mass"][o]]
        rec[j, 10] = tree["size"][o]
        rec[j, 11] = pe - pb + 1
    return rec, nodes


def direct_f64(x, y, z, m, eps, gravity, t0=0, t1=None):
    x, y, z, m = _f32(x), _f32(y), _f32(z), _f32(m)
    n = x.size
    t1 = n if t1 is None else t1
    out = [np.empty(t1 - t0, np.float64) for _ in range(3)]
    lib().orc_direct_f64(C.c_int(n), C.c_int(t0), C.c_int(t1), _p(x), _p(y), _p(z), _p(m),
                         C.c_double(float(eps) ** 2), C.c_double(gravity), *[_p(o) for o in out])
    return out


def direct_f32(x, y, z, m, eps, gravity, t0=0, t1=None):
    x, y, z, m = _f32(x), _f32(y), _f32(z), _f32(m)
    n = x.size
    t1 = n if t1 is None else t1
    out = [np.empty(t1 - t0, np.float32) for _ in range(3)]
    lib().orc_direct_f32(C.c_int(n), C.c_int(t0), C.c_int(t1), _p(x), _p(y), _p(z), _p(m),
                         C.c_float(np.float32(eps) * np.float32(eps)), C.c_float(gravity),
                         *[_p(o) for o in out])
    return out


def energy_f64(x, y, z, vx, vy, vz, m, eps, gravity):
    a = [_f32(v) for v in (x, y, z, vx, vy, vz, m)]
    return lib().orc_energy_f64(C.c_int(a[0].size), *[_p(v) for v in a],
                                C.c_double(float(eps) ** 2), C.c_double(gravity))


def acc_error(ax, ay, az, rx, ry, rz):
    ax, ay, az = _f32(ax), _f32(ay), _f32(az)
    rx, ry, rz = (np.ascontiguousarray(v, np.float64) for v in (rx, ry, rz))
    mean, mx = C.c_double(), C.c_double()
    lib().orc_acc_error(C.c_int(ax.size), _p(ax), _p(ay), _p(az), _p(rx), _p(ry), _p(rz),
                        C.byref(mean), C.byref(mx))
    return mean.value, mx.value


def leapfrog_pos(dt, domain, x, y, z, vx, vy, vz, ax, ay, az):
    x, y, z = _f32(x).copy(), _f32(y).copy(), _f32(z).copy()
    n = x.size
    vh = [np.empty(n, np.float32) for _ in range(3)]
    lib().orc_leapfrog_pos(C.c_int(n), C.c_float(dt), C.c_float(domain), _p(x), _p(y), _p(z),
                           _p(_f32(vx)), _p(_f32(vy)), _p(_f32(vz)), _p(_f32(ax)), _p(_f32(ay)),
                           _p(_f32(az)), *[_p(v) for v in vh])
    return x, y, z, vh


def full_build(x, y, z, m, domain, bits, max_codes):
    """Morton -> sort -> unique -> coarsen -> octree -> COM on host arrays."""
    codes = morton(x, y, z, domain, bits)
    sorted_codes, perm = sort(codes)
    xs, ys, zs, ms = (_f32(a)[perm] for a in (x, y, z, m))
    uq, first_pt = unique(sorted_codes)
    lfc = coarsen(uq, bits, max_codes)
    tree = build_octree(uq, lfc, first_pt, bits, domain)
    comd = com(tree, xs, ys, zs, ms)
    return dict(codes=codes, codes_sorted=sorted_codes, perm=perm, unique=uq, first_pt=first_pt,
                leaf_first_code_idx=lfc, tree=tree, com=comd, xs=xs, ys=ys, zs=zs, ms=ms)


class OracleSim:
    """Whole-step host simulation (Simulation<T>::update sequencing over the oracle stages)."""

    def __init__(self, x, y, z, m, vx=None, vy=None, vz=None, *, bits=10, max_codes=32, group=64,
                 theta=0.75, dt=1e-4, gravity=1.0, eps=1e-3, domain=5.0, damp=0.0, fit=False):
        """fit: keys relative to a cube fitted to the particles (the library's FLAG_FIT_KEYS)."""
        self.n = len(x)
        arrs = [_f32(a) if a is not None else None for a in (x, y, z, vx, vy, vz, m)]
        create = lib().orc_sim_create_fit if fit else lib().orc_sim_create
        self._group = group
        self._h = C.c_void_p(create(
            C.c_int(self.n), C.c_int(bits), C.c_int(max_codes), C.c_int(group), C.c_float(theta),
            C.c_float(dt), C.c_float(gravity), C.c_float(eps), C.c_float(domain), C.c_float(damp),
            *[_p(a) for a in arrs]))

    def step(self):
        lib().orc_sim_step(self._h)

    def state(self):
        names = ("x", "y", "z", "vx", "vy", "vz", "m", "ax", "ay", "az")
        out = {k: np.empty(self.n, np.float32) for k in names}
        out["id"] = np.empty(self.n, np.uint32)
        lib().orc_sim_get_state(self._h, *[_p(out[k]) for k in names], _p(out["id"]))
        return out

    def counts(self):
        U, M, no, it = C.c_int32(), C.c_int32(), C.c_int32(), C.c_uint64()
        lib().orc_sim_get_counts(self._h, C.byref(U), C.byref(M), C.byref(no), C.byref(it))
        return U.value, M.value, no.value, it.value

    def timings(self):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        lib().orc_sim_get_timings(self._h, C.byref(a), C.byref(b), C.byref(c))
        return dict(ms_integrate=a.value, ms_build=b.value, ms_traverse=c.value)

    def key_domain(self):
        out = np.zeros(4, np.float32)
        lib().orc_sim_get_key_domain(self._h, _p(out))
        return out

    def tree(self):
        """Sorted codes, octree arrays, centres of mass and list lengths of the last build/step."""
        U, M, no, _ = self.counts()
        out = dict(codes_sorted=np.empty(self.n, np.uint64))
        for k in ("first_child", "num_children", "points_begin", "points_end"):
            out[k] = np.empty(no, np.int32)
        for k in ("size", "comx", "comy", "comz", "mass"):
            out[k] = np.empty(no, np.float32)
        out["list_len"] = np.empty((self.n + self._group - 1) // self._group, np.uint32)
        lib().orc_sim_get_tree(self._h, _p(out["codes_sorted"]), _p(out["first_child"]),
                               _p(out["num_children"]), _p(out["points_begin"]), _p(out["points_end"]),
                               _p(out["size"]), _p(out["comx"]), _p(out["comy"]), _p(out["comz"]),
                               _p(out["mass"]), _p(out["list_len"]))
        out["n_oct"] = no
        return out

    def close(self):
        if self._h:
            lib().orc_sim_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- double-precision instantiation (Simulation<double>, src/cuda/simulation.cu:256) ----

def _f64(a):
    return np.ascontiguousarray(a, np.float64)


def morton_dp(x, y, z, domain, bits):
    x, y, z = _f64(x), _f64(y), _f64(z)
    codes = np.empty(x.size, np.uint64)
    lib().orc_morton_dp(_p(x), _p(y), _p(z), C.c_int(x.size), C.c_float(domain), C.c_int(bits), _p(codes))
    return codes


def com_dp(tree, x, y, z, m):
    n_oct = tree["n_oct"]
    x, y, z, m = _f64(x), _f64(y), _f64(z), _f64(m)
    out = {k: np.empty(n_oct, np.float64) for k in ("comx", "comy", "comz", "mass")}
    lib().orc_com_dp(C.c_int(n_oct), _p(tree["first_child"]), _p(tree["num_children"]),
                     _p(tree["points_begin"]), _p(tree["points_end"]), _p(x), _p(y), _p(z), _p(m),
                     _p(out["comx"]), _p(out["comy"]), _p(out["comz"]), _p(out["mass"]))
    return out


def traverse_dp(tree, comd, x, y, z, m, group, theta, eps, gravity, domain):
    """theta, eps, gravity, domain are the float32 parameters, widened like the library does."""
    x, y, z, m = _f64(x), _f64(y), _f64(z), _f64(m)
    n = x.size
    ax, ay, az = (np.empty(n, np.float64) for _ in range(3))
    ll = np.empty((n + group - 1) // group, np.uint32)
    a = {k: _f64(comd[k]) for k in ("comx", "comy", "comz", "mass")}
    size = _f64(tree["size"])
    e = float(np.float32(eps))
    lib().orc_traverse_dp(C.c_int(n), C.c_int(group), _p(x), _p(y), _p(z), _p(m),
                          C.c_int(tree["n_oct"]), _p(np.ascontiguousarray(tree["first_child"], np.int32)),
                          _p(np.ascontiguousarray(tree["num_children"], np.int32)),
                          _p(np.ascontiguousarray(tree["points_begin"], np.int32)),
                          _p(np.ascontiguousarray(tree["points_end"], np.int32)), _p(size),
                          _p(a["comx"]), _p(a["comy"]), _p(a["comz"]), _p(a["mass"]),
                          C.c_double(float(np.float32(theta))), C.c_double(e * e),
                          C.c_double(float(np.float32(gravity))), C.c_double(float(np.float32(domain))),
                          _p(ax), _p(ay), _p(az), _p(ll))
    return ax, ay, az, ll


def direct_dp(x, y, z, m, eps, gravity, t0=0, t1=None):
    x, y, z, m = _f64(x), _f64(y), _f64(z), _f64(m)
    n = x.size
    t1 = n if t1 is None else t1
    out = [np.empty(t1 - t0, np.float64) for _ in range(3)]
    e = float(np.float32(eps))
    lib().orc_direct_dp(C.c_int(n), C.c_int(t0), C.c_int(t1), _p(x), _p(y), _p(z), _p(m),
                        C.c_double(e * e), C.c_double(float(np.float32(gravity))), *[_p(o) for o in out])
    return out


def full_build_dp(x, y, z, m, domain, bits, max_codes):
    """full_build for double-precision particles: same integer stages, COM in double."""
    codes = morton_dp(x, y, z, domain, bits)
    sorted_codes, perm = sort(codes)
    xs, ys, zs, ms = (_f64(a)[perm] for a in (x, y, z, m))
    uq, first_pt = unique(sorted_codes)
    lfc = coarsen(uq, bits, max_codes)
    tree = build_octree(uq, lfc, first_pt, bits, domain)
    comd = com_dp(tree, xs, ys, zs, ms)
    return dict(codes=codes, codes_sorted=sorted_codes, perm=perm, unique=uq, first_pt=first_pt,
                leaf_first_code_idx=lfc, tree=tree, com=comd, xs=xs, ys=ys, zs=zs, ms=ms)


class OracleSimDP:
    """OracleSim with every floating-point array and operation in double."""

    def __init__(self, x, y, z, m, vx=None, vy=None, vz=None, *, bits=10, max_codes=32, group=64,
                 theta=0.75, dt=1e-4, gravity=1.0, eps=1e-3, domain=5.0, damp=0.0):
        self.n = len(x)
        arrs = [_f64(a) if a is not None else None for a in (x, y, z, vx, vy, vz, m)]
        self._h = C.c_void_p(lib().orc_sim_create_dp(
            C.c_int(self.n), C.c_int(bits), C.c_int(max_codes), C.c_int(group), C.c_float(theta),
            C.c_float(dt), C.c_float(gravity), C.c_float(eps), C.c_float(domain), C.c_float(damp),
            *[_p(a) for a in arrs]))

    def step(self):
        lib().orc_sim_step_dp(self._h)

    def state(self):
        names = ("x", "y", "z", "vx", "vy", "vz", "m", "ax", "ay", "az")
        out = {k: np.empty(self.n, np.float64) for k in names}
        out["id"] = np.empty(self.n, np.uint32)
        lib().orc_sim_get_state_dp(self._h, *[_p(out[k]) for k in names], _p(out["id"]))
        return out

    def counts(self):
        U, M, no, it = C.c_int32(), C.c_int32(), C.c_int32(), C.c_uint64()
        lib().orc_sim_get_counts_dp(self._h, C.byref(U), C.byref(M), C.byref(no), C.byref(it))
        return U.value, M.value, no.value, it.value

    def tree(self):
        U, M, no, _ = self.counts()
        out = dict(codes_sorted=np.empty(self.n, np.uint64))
        for k in ("first_child", "num_children", "points_begin", "points_end"):
            out[k] = np.empty(no, np.int32)
        for k in ("comx", "comy", "comz", "mass"):
            out[k] = np.empty(no, np.float64)
        lib().orc_sim_get_tree_dp(self._h, _p(out["codes_sorted"]), _p(out["first_child"]),
                                  _p(out["num_children"]), _p(out["points_begin"]),
                                  _p(out["points_end"]), _p(out["comx"]), _p(out["comy"]),
                                  _p(out["comz"]), _p(out["mass"]), None)
        return out

    def close(self):
        if self._h:
            lib().orc_sim_destroy_dp(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
