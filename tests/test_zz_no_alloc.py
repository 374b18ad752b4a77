"""nbody_direct_energy_dev_noalloc -- the entry point that allocates nothing (its scratch is a
fixed array in device memory) -- against nbody_direct_energy_dev, which allocates its scratch
per call: same kernel, same order of summation, so the two doubles must be the same bits.  Two
calls in a row, the second on a smaller input (the scratch keeps the first call's partial
sums); then the size limit."""
import ctypes as C

import numpy as np
import pytest

import nbody_amd as nb

pytestmark = pytest.mark.gpu


def _energy(fn, ic, n, eps):
    import torch
    posm = torch.from_numpy(np.stack([ic[k][:n] for k in ("x", "y", "z", "m")], 1).astype(np.float32)).cuda()
    v = [torch.from_numpy(np.ascontiguousarray(ic[k][:n])).cuda() for k in ("vx", "vy", "vz")]
    out = torch.full((2,), np.nan, dtype=torch.float64, device="cuda")
    rc = fn(posm.data_ptr(), v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(), n, eps, 1.0,
            out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_no_alloc_energy_equals_allocating_entry_point():
    lib = nb.load_library()
    fn = lib.nbody_direct_energy_dev_noalloc
    ic = nb.sample_ic(nb.IC_DISK, 70000, 4)
    for n in (70000, 12345, 12345, 1, 300):       # larger first, then smaller inputs, repeated
        want = _energy(lib.nbody_direct_energy_dev, ic, n, 0.01)
        got = _energy(fn, ic, n, 0.01)
        assert np.isfinite(want).all() and want[1] <= 0.0
        assert np.array_equal(got, want), (n, got, want)
    # beyond the fixed scratch: rejected, nothing launched
    assert fn(1, 1, 1, 1, 4194304 + 1, 0.01, 1.0, 1, None) == nb.ERR_CAPACITY
    assert fn(None, None, None, None, 16, 0.01, 1.0, None, None) == 1


def test_no_alloc_entry_point_under_graph_capture():
    """Two calls captured into one HIP graph and replayed twice: the entry point does nothing a
    capture forbids (the scratch address was resolved when a handle was created on the device)
    and a replay gives the bits of a plain call."""
    import torch
    lib = nb.load_library()
    with nb.Simulation(nb.Params(num_points=64, device_id=0)):
        pass                                        # nbody_create resolves the scratch address
    ic = nb.sample_ic(nb.IC_DISK, 30000, 8)
    want = [_energy(lib.nbody_direct_energy_dev, ic, n, 0.01) for n in (30000, 777)]
    dev = []
    for n in (30000, 777):
        posm = torch.from_numpy(np.stack([ic[k][:n] for k in ("x", "y", "z", "m")], 1).astype(np.float32)).cuda()
        v = [torch.from_numpy(np.ascontiguousarray(ic[k][:n])).cuda() for k in ("vx", "vy", "vz")]
        dev.append((posm, v, torch.full((2,), np.nan, dtype=torch.float64, device="cuda"), n))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.graph(g, stream=side):
        for posm, v, out, n in dev:
            rc = lib.nbody_direct_energy_dev_noalloc(posm.data_ptr(), v[0].data_ptr(), v[1].data_ptr(),
                                                     v[2].data_ptr(), n, 0.01, 1.0, out.data_ptr(),
                                                     torch.cuda.current_stream().cuda_stream)
            assert rc == 0
    for _ in range(2):
        for _, _, out, _ in dev:
            out.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        for (_, _, out, n), w in zip(dev, want):
            assert np.array_equal(out.cpu().numpy(), w), n
