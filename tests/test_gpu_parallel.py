"""GPU tests of the multi-GPU code paths that a one-GPU box can exercise: the C++ mirror's
example binary, and the torch/RCCL plumbing of nbody_amd.parallel with a single rank (zero
copy views of library memory, in-place all-gather, stream hand-over)."""
import os
import re
import socket
import subprocess

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


@pytest.fixture(scope="module")
def single_rank_group():
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def test_sharded_stepper_single_rank_matches_plain_update(single_rank_group):
    n = 8192
    ic = nb.sample_ic(nb.IC_DISK, n, 3)
    kw = dict(num_points=n, morton_bits=21, theta=0.5, dt=1e-3, softening_factor=0.01, device_id=0)
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
