"""GPU tests of the multi-GPU code paths that a one-GPU box can exercise: the C++ mirror's
example binary, and the torch/RCCL plumbing of nbody_amd.parallel with a single rank (zero
copy views of library memory, in-place all-gather, stream hand-over)."""
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import nbody_amd as nb
import oracle_lib as O
from nbody_amd import parallel

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_mirror_example_runs():
    exe = os.path.join(ROOT, "examples", "headless_main")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    out = subprocess.run([exe, "disk", "16384", "3", "21"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    m = re.search(r"octree_nodes=(\d+) ms_per_step=([0-9.]+) mean_x=([0-9.]+)", out.stdout)
    assert m, out.stdout
    # same spawn + 3 updates through the Python binding: identical tree size
    with nb.Simulation(nb.Params(num_points=16384, morton_bits=21, theta=0.75, dt=1e-4,
                                 softening_factor=0.001)) as sim:
        sim.spawn_disk()
        sim.step(3)
        assert sim.get_num_octree_nodes() == int(m.group(1))
        assert abs(sim.get_state()["x"].mean() - float(m.group(3))) < 1e-4
    # Simulation<double>
    out = subprocess.run([exe, "disk", "16384", "3", "21", "double"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    m = re.search(r"precision=double octree_nodes=(\d+) ms_per_step=([0-9.]+) mean_x=([0-9.]+)", out.stdout)
    assert m, out.stdout
    with nb.Simulation(nb.Params(num_points=16384, morton_bits=21, theta=0.75, dt=1e-4,
                                 softening_factor=0.001, flags=nb.FLAG_FP64)) as sim:
        sim.spawn_disk()
        sim.step(3)
        assert sim.get_num_octree_nodes() == int(m.group(1))
        assert abs(sim.get_state()["x"].mean() - float(m.group(3))) < 1e-8


def test_cpp_rccl_example_single_rank():
    """nbody_dd_* driven from C++ over RCCL (examples/dd_rccl_main.cpp) with one rank: the RCCL
    vtable is bound and called; the tree equals the Python binding's for the same particles."""
    exe = os.path.join(ROOT, "examples", "dd_rccl_main")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    n = 200000
    out = subprocess.run([exe, "0", "1", "-", "sphere", str(n), "3"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.search(r"rank=0/1 particles=(\d+) octree_nodes=(\d+) records_sent=0 records_received=0", out.stdout)
    assert m and int(m.group(1)) == n, out.stdout
    with nb.Simulation(nb.Params(num_points=n, morton_bits=21, theta=0.6, dt=1e-6, softening_factor=0.1)) as sim:
        sim.spawn_sphere()
        sim.step(3)
        assert sim.get_num_octree_nodes() == int(m.group(2))


@pytest.fixture(scope="module")
def single_rank_group():
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("flags", [0, nb.FLAG_FP64])
def test_sharded_stepper_single_rank_matches_plain_update(single_rank_group, flags):
    n = 8192
    ic = nb.sample_ic(nb.IC_DISK, n, 3)
    kw = dict(num_points=n, morton_bits=21, theta=0.5, dt=1e-3, softening_factor=0.01, device_id=0,
              flags=flags)
    with nb.Simulation(nb.Params(**kw)) as a, nb.Simulation(nb.Params(**kw)) as b:
        for s in (a, b):
            s.set_state(ic["x"], ic["y"], ic["z"], ic["m"], ic["vx"], ic["vy"], ic["vz"])
        st = parallel.ShardedTraversal(parallel.GpuBackend(b, "cuda:0"))
        assert st.even and (st.g0, st.g1) == (0, n // 64)
        for _ in range(3):
            a.update()
            st.step()
        sa, sb = a.get_state(), b.get_state()
    for k in ("x", "vx", "vy", "vz", "ax", "ay", "az"):
        assert np.array_equal(sa[k], sb[k]), k


def test_ring_direct_sum_single_rank_gpu(single_rank_group):
    n = 3000
    ic = nb.sample_ic(nb.IC_SPHERE, n, 3)
    posm = torch.from_numpy(np.stack([ic["x"], ic["y"], ic["z"], ic["m"] / n], 1).astype(np.float32)).cuda()
    out = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(3)]
    kern = parallel.gpu_tile_kernel(nb.load_library(), *out, 0.05, 1.0)
    parallel.ring_direct_sum(posm, kern)
    torch.cuda.synchronize()
    rx, ry, rz = O.direct_f64(ic["x"], ic["y"], ic["z"], ic["m"] / n, 0.05, 1.0)
    mean, mx = O.acc_error(out[0].cpu().numpy(), out[1].cpu().numpy(), out[2].cpu().numpy(), rx, ry, rz)
    assert mx < 2e-5
    # accumulate=1 adds a second pass on top
    kern(posm, posm, True)
    torch.cuda.synchronize()
    assert np.allclose(out[0].cpu().numpy(), 2 * rx, rtol=1e-4, atol=1e-6)


def test_domain_decomposition_single_rank_is_a_plain_step(single_rank_group):
    """world = 1: the native stepper has nothing to exchange and reproduces update()."""
    n = 8192
    ic = nb.sample_ic(nb.IC_DISK, n, 3)
    kw = dict(num_points=n, morton_bits=21, theta=0.5, dt=1e-3, softening_factor=0.01, device_id=0)
    with nb.Simulation(nb.Params(**kw)) as a, nb.Simulation(nb.Params(**kw)) as b:
        for s in (a, b):
            s.set_state(ic["x"], ic["y"], ic["z"], ic["m"], ic["vx"], ic["vy"], ic["vz"])
        with parallel.DomainDecomposition(b, "cuda:0") as dd:
            dd.rebalance()
            for _ in range(3):
                a.update()
                dd.step()
            st = dd.stats()
        sa, sb = a.get_state(), b.get_state()
    assert st["n_local"] == n and st["records_sent"] == 0 and st["splitters"] == [0, 1 << 63]
    for k in ("x", "vx", "vy", "vz", "ax", "ay", "az"):
        assert np.array_equal(sa[k], sb[k]), k


def test_domain_decomposition_accepts_double_and_fitted_handles(single_rank_group):
    """A double handle steps like a plain double update(); a handle with a fitted key domain is
    accepted too (the ranks then draw one cube together)."""
    kw = dict(num_points=3000, morton_bits=21, theta=0.5, dt=1e-3, softening_factor=0.01, device_id=0, seed=9,
              flags=nb.FLAG_FP64)
    with nb.Simulation(nb.Params(**kw)) as a, nb.Simulation(nb.Params(**kw)) as b:
        a.spawn_sphere()
        with parallel.DomainDecomposition(b, "cuda:0") as dd:
            dd.spawn(nb.IC_SPHERE, 3000)
            for _ in range(2):
                a.update()
                dd.step()
        sa, sb = a.get_state(), b.get_state()
    assert sb["x"].dtype == np.float64
    for k in ("x", "vx", "ax", "az"):
        assert np.array_equal(sa[k][np.argsort(sa["id"])], sb[k][np.argsort(sb["id"])]), k
    with nb.Simulation(nb.Params(num_points=1000, morton_bits=10, device_id=0, flags=nb.FLAG_FIT_KEYS)) as sim:
        with parallel.DomainDecomposition(sim, "cuda:0") as dd:
            dd.spawn(nb.IC_SPHERE, 1000)
            dd.step()
            assert dd.stats()["n_local"] == 1000 and np.isfinite(sim.get_state()["ax"]).all()


def test_torch_comm_callbacks_single_rank_nccl(single_rank_group):
    """The two collectives the library calls back into, over RCCL with one rank: raw device
    pointers, the library's own stream, scattered and packed layouts."""
    import ctypes as C
    comm = parallel.TorchComm("cuda:0")
    n = 1000
    with nb.Simulation(nb.Params(num_points=64, device_id=0)) as sim:
        stream = sim.device_view().stream
        src = torch.arange(n, dtype=torch.float32, device="cuda")
        dst = torch.zeros(n, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        assert comm._all_gather(None, src.data_ptr(), dst.data_ptr(), 4 * n, stream) == 0
        torch.cuda.synchronize()
        assert torch.equal(src, dst)
        dst.zero_()
        torch.cuda.synchronize()
        arr = lambda v: (C.c_size_t * 1)(v)
        # 100 floats from offset 40 floats to offset 8 floats
        assert comm._all_to_all(None, src.data_ptr(), arr(400), arr(160), dst.data_ptr(), arr(400), arr(32), stream) == 0
        torch.cuda.synchronize()
        assert torch.equal(dst[8:108], src[40:140]) and dst[:8].sum() == 0 and dst[108:].sum() == 0
        assert comm._all_to_all(None, None, arr(0), arr(0), None, arr(0), arr(0), stream) == 0


_DD_WORKER = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import nbody_amd as nb
from nbody_amd import parallel
rank, world, n = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(sys.argv[2])
bits = int(sys.argv[4])
mode = sys.argv[5] if len(sys.argv) > 5 else ""
fit = {"fit": nb.FLAG_FIT_KEYS, "fp64": nb.FLAG_FP64}.get(mode, 0)
dist.init_process_group("gloo", rank=rank, world_size=world)
cap = int(n / world * 1.5) + 64                      # head-room for the re-balance
sim = nb.Simulation(nb.Params(num_points=cap, morton_bits=bits, theta=0.6, dt=1e-3,
                              softening_factor=0.01, device_id=0, seed=5, flags=nb.FLAG_TIMING | fit))
dd = parallel.DomainDecomposition(sim, "cuda:0", via_host=True)
dd.spawn(nb.IC_SPHERE, n)
if mode == "two_masses":     # every rank's particles have one mass, but not the same one everywhere
    st = sim.get_state()
    sim.set_particles(st["x"], st["y"], st["z"], st["m"] * (2.0 if rank == 0 else 1.0), st["vx"], st["vy"], st["vz"],
                      st["ax"], st["ay"], st["az"], id=st["id"])
    dd.rebalance()
st0 = dd.stats()
keys0 = sim.get_octree()["codes_sorted"]
dd.step()
s1 = sim.get_state()
st1 = dd.stats()
# a second step, then a re-balance and a third: the state must stay consistent
dd.step()
dd.rebalance()
dd.step()
s = sim.get_state()
t = sim.get_timings()
np.savez(sys.argv[3] + f"/rank{rank}.npz", gid=s["id"], gid1=s1["id"], keys0=keys0,
         split=np.array(st0["splitters"], np.uint64), sent=st1["records_sent"], recv=st1["records_received"],
         n0=st0["n_local"], steps=t["steps"], inter=t["interactions"], cube=np.asarray(sim.key_domain() if mode != "fp64" else [0, 0, 0, 5], np.float32),
         **{k: s[k] for k in ("x", "y", "z", "m", "ax", "ay", "az")},
         **{k + "1": s1[k] for k in ("x", "y", "z", "m", "ax", "ay", "az")})
dist.barrier()
dd.close()
sim.close()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world,n,bits,mode", [(2, 12000, 21, ""), (3, 9000, 21, ""), (4, 8000, 10, ""),
                                               (3, 9000, 10, "fit"), (2, 8000, 10, "fit"),
                                               (2, 9000, 21, "fp64"), (3, 8000, 21, "fp64")])
def test_native_domain_decomposition_processes_share_the_gpu(tmp_path, world, n, bits, mode):
    """nbody_dd_spawn / step / rebalance with `world` ranks (processes that share this GPU; the
    collectives are staged through host memory and gloo): ownership is by Morton-key range,
    every particle is owned once, forces checked against a float64 direct sum over all
    particles and against the single-handle tree code.  Modes: 30-bit keys fitted to one cube
    across the ranks; double-precision handles (copies, migration records and sums in double)."""
    fit, f64 = mode == "fit", mode == "fp64"
    real = np.float64 if f64 else np.float32
    flags = nb.FLAG_FIT_KEYS if fit else nb.FLAG_FP64 if f64 else 0
    script = tmp_path / "worker.py"
    script.write_text(_DD_WORKER)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(n), str(tmp_path), str(bits), mode],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                      text=True))
    for p in procs:
        out, _ = p.communicate(timeout=170)
        assert p.returncode == 0, out
    ic = nb.sample_ic(nb.IC_SPHERE, n, 5, dtype=real)
    res = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    split = res[0]["split"]
    assert int(split[0]) == 0 and int(split[-1]) == 1 << (3 * bits)
    errs = {}
    for tag in ("1", ""):
        full = {k: np.full(n, np.nan, real) for k in ("x", "y", "z", "m", "ax", "ay", "az")}
        for r, d in enumerate(res):
            gid = d["gid" + tag]
            for k in full:
                full[k][gid] = d[k + tag]
        assert not np.isnan(full["x"]).any()            # every particle is owned by exactly one rank
        assert sum(len(d["gid" + tag]) for d in res) == n
        ref = O.direct_f64(full["x"], full["y"], full["z"], full["m"], 0.01, 1.0)
        mean, mx = O.acc_error(full["ax"], full["ay"], full["az"], *ref)
        # (inside a uniform sphere the forces largely cancel: the single-tree code is at 3e-3 here)
        errs[tag] = (mean, mx)
        assert mean < 1e-2 and mx < 0.3, (tag, mean, mx)
    if fit:   # 30-bit keys relative to ONE cube, fitted to all ranks' particles together
        cube = res[0]["cube"]
        assert all(np.array_equal(d["cube"], cube) for d in res)
        assert 1.0 < cube[3] < 1.6 and np.all(cube[:3] > 1.6) and np.all(cube[:3] + cube[3] < 3.4), cube
    for r, d in enumerate(res):
        assert np.array_equal(d["split"], split)
        # after the spawn every rank holds exactly its key range, about n / world particles
        assert np.all(d["keys0"] >= split[r]) and np.all(d["keys0"] < split[r + 1])
        assert abs(int(d["n0"]) - n / world) <= 2 * 0.005 * n / world + 8
        assert 0 < int(d["sent"]) < n and 0 < int(d["recv"]) < n
        assert int(d["steps"]) == 1 and int(d["inter"]) > 0   # the steps since the last re-balance
    # first step against the single-handle tree code on the same particles (same positions:
    # v = a = 0 at the first kick-drift; forces differ by the trees only)
    with nb.Simulation(nb.Params(num_points=n, morton_bits=bits, theta=0.6, dt=1e-3,
                                 softening_factor=0.01, flags=flags)) as one:
        one.set_state(ic["x"], ic["y"], ic["z"], ic["m"])
        one.update()
        s1 = one.get_state()
    full1 = {k: np.empty(n, real) for k in ("x", "y", "z", "m", "ax", "ay", "az")}
    for d in res:
        for k in full1:
            full1[k][d["gid1"]] = d[k + "1"]
    assert np.array_equal(full1["x"][s1["id"]], s1["x"])
    # every component, per body, against the single tree: a lost or doubled copy of a
    # neighbour's nodes would show here at once
    scale = np.sqrt(s1["ax"].astype(np.float64) ** 2 + s1["ay"] ** 2 + s1["az"] ** 2).max()
    for k in ("ax", "ay", "az"):
        assert np.max(np.abs(full1[k][s1["id"]] - s1[k])) < 1e-2 * scale, k
    # and the error against the direct sum is the single tree's, not worse
    ref = O.direct_f64(s1["x"], s1["y"], s1["z"], s1["m"], 0.01, 1.0)
    mean_one, max_one = O.acc_error(s1["ax"], s1["ay"], s1["az"], *ref)
    assert errs["1"][0] < 1.5 * mean_one, (errs, mean_one)
    assert errs["1"][1] < 3 * max_one + 5e-2, (errs, max_one)


def test_native_domain_decomposition_with_one_mass_per_rank_but_not_overall(tmp_path):
    """Rank 0's particles all have mass 2, the others' mass 1: each handle alone would leave the
    mass out of its pair arithmetic, the walk over own tree and imported copies must not (the
    re-balancing compares the ranks' masses).  Forces against a float64 direct sum with the real
    masses, after the first step and after a second re-balance has mixed the two kinds."""
    world, n, bits = 3, 9000, 21
    script = tmp_path / "worker.py"
    script.write_text(_DD_WORKER)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(n), str(tmp_path), str(bits), "two_masses"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for p in procs:
        out, _ = p.communicate(timeout=170)
        assert p.returncode == 0, out
    res = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for tag in ("1", ""):
        full = {k: np.full(n, np.nan, np.float32) for k in ("x", "y", "z", "m", "ax", "ay", "az")}
        for d in res:
            for k in full:
                full[k][d["gid" + tag]] = d[k + tag]
        assert not np.isnan(full["x"]).any()
        heavy = int((full["m"] == 2.0).sum())
        assert 0.3 * n < heavy < 0.37 * n and int((full["m"] == 1.0).sum()) == n - heavy
        ref = O.direct_f64(full["x"], full["y"], full["z"], full["m"], 0.01, 1.0)
        mean, mx = O.acc_error(full["ax"], full["ay"], full["az"], *ref)
        assert mean < 1e-2 and mx < 0.3, (tag, mean, mx)
