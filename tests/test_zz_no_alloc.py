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
