"""CPU checks of the C-ABI library: it loads, exports every symbol include/nbody_c.h
declares, its host-only entry points work, and compute entry points fail loudly (never
fall back to a CPU path) when no GPU is present."""
import ctypes as C
import os
import re

import numpy as np

import nbody_amd as nb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "nbody_c.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nbody_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = nb.load_library()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in nbody_c.h but not exported"
    assert sorted(nb.EXPORTS) == syms


def test_params_layout_matches_reference_struct():
    # Simulation<T>::Params on LP64: 3 x i32, pad, u64, 6 x f32, i32 (include/simulation.hpp:15-28)
    f = dict((n, getattr(nb.CParams, n).offset) for n, _ in nb.CParams._fields_)
    assert (f["num_points"], f["max_num_codes_per_leaf"], f["traversal_group_size"]) == (0, 4, 8)
    assert f["mem_traversal_queues"] == 16 and f["theta"] == 24 and f["num_steps_validator"] == 48
    p = nb.CParams()
    nb.load_library().nbody_default_params(C.byref(p))
    assert (p.max_num_codes_per_leaf, p.theta, p.gravity, p.domain_size) == (32, 0.75, 1.0, 5.0)
    assert p.mem_traversal_queues == 50 << 20 and p.seed == 42


def test_sampler_distributions():
    d = nb.sample_ic(nb.IC_DISK, 20000, 42)
    assert d["m"][0] == 10.0 and np.allclose(d["m"][1:], 1.0 / 20000)
    r = np.hypot(d["x"][1:] - 2.5, d["y"][1:] - 2.5)
    assert r.min() >= 0.03 - 1e-5 and r.max() <= 0.30 + 1e-5
    # Keplerian: v^2 r = G * enclosed mass in [10, 11]
    v2r = (d["vx"][1:] ** 2 + d["vy"][1:] ** 2) * r
    assert v2r.min() > 9.9 and v2r.max() < 11.1
    # velocity is tangential
    assert np.abs((d["x"][1:] - 2.5) * d["vx"][1:] + (d["y"][1:] - 2.5) * d["vy"][1:]).max() < 1e-3
    s = nb.sample_ic(nb.IC_SPHERE, 20000, 7)
    rr = np.sqrt((s["x"] - 2.5) ** 2 + (s["y"] - 2.5) ** 2 + (s["z"] - 2.5) ** 2)
    assert rr.max() <= 0.5 + 1e-5 and abs((rr < 0.5 * 0.5 ** (1 / 3)).mean() - 0.5) < 0.02
    u = nb.sample_ic(nb.IC_UNIFORM, 20000, 7)
    assert u["x"].min() >= 2.0 and u["x"].max() <= 3.0 and abs(u["x"].mean() - 2.5) < 0.01
    pl = nb.sample_ic(nb.IC_PLUMMER, 20000, 7, p0=0.1, p1=1.0)
    rp = np.sqrt((pl["x"] - 2.5) ** 2 + (pl["y"] - 2.5) ** 2 + (pl["z"] - 2.5) ** 2)
    assert rp.max() <= 1.0 + 1e-5 and abs(np.median(rp) - 0.1 * 1.3048) < 0.01
    assert np.isclose(pl["m"].sum(), 1.0, rtol=1e-4)
    # reproducible and seed dependent
    # counter-based: particle i's draws depend on (seed, i) only, not on how many are sampled
    assert np.array_equal(nb.sample_ic(nb.IC_DISK, 100, 42)["x"], d["x"][:100])
    assert np.array_equal(nb.sample_ic(nb.IC_DISK, 20000, 42)["x"], d["x"])
    assert not np.array_equal(nb.sample_ic(nb.IC_DISK, 20000, 43)["x"], d["x"])


def test_invalid_params_rejected():
    lib = nb.load_library()
    for field, val in (("morton_bits", 16), ("traversal_group_size", 48), ("num_points", 0),
                       ("softening_factor", 0.0), ("domain_size", -1.0)):
        p = nb.CParams()
        lib.nbody_default_params(C.byref(p))
        setattr(p, field, val)
        h = C.c_void_p()
        assert lib.nbody_create(C.byref(p), C.byref(h)) == 1 and not h.value


def test_no_cpu_fallback_without_gpu():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the failure path of a GPU-less host cannot be exercised")
    try:
        nb.Simulation(nb.Params(num_points=1024))
    except RuntimeError as e:
        assert "status 2" in str(e)
    else:
        raise AssertionError("Simulation() must fail without a GPU")


def test_bench_refuses_to_run_without_a_gpu():
    """No CPU fallback anywhere: without a GPU the benchmark says so and prints no result line."""
    import subprocess
    import sys
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 2
    assert r.stdout.strip() == ""
    assert "needs a GPU" in r.stderr
