"""The sort of nearly sorted keys (63-bit keys: elements whose key stayed between their neighbours'
previous keys are in order already, the others are sorted by the digit passes and merged in;
csrc/k_sort.hip, k_coh_*) must give the stable sort of everything -- the permutation of the digit
passes over all elements, which tests/test_gpu_parity.py pins to the oracle's
(reference: cub::DeviceMergeSort::SortPairs + thrust::sequence, src/cuda/points.cu:73-84).

NBODY_FLAG_ALWAYS_COHERENT_SORT takes the path on every sort after the first whatever share of the
elements moved, NBODY_FLAG_NO_COHERENT_SORT never: sorted codes and permutation of every step
and the state at the end are compared bit for bit, and the first steps against the oracle."""
import numpy as np
import pytest

import nbody_amd as nb
import oracle_lib as O

pytestmark = pytest.mark.gpu


def run(ic, n, steps, flags, **kw):
    p = nb.Params(num_points=n, morton_bits=21, flags=flags, **kw)
    per_step = []
    with nb.Simulation(p) as sim:
        sim.set_state(ic["x"], ic["y"], ic["z"], ic["m"], ic.get("vx"), ic.get("vy"), ic.get("vz"))
        for _ in range(steps):
            sim.update()
            o = sim.get_octree()
            per_step.append((o["codes_sorted"].copy(), o["perm"].copy(), o["list_len"].copy()))
        st = sim.get_state()
        stats = sim.sort_stats()
    return per_step, st, stats


def assert_same_run(a, b):
    (pa, sa, _), (pb, sb, _) = a, b
    for k, (x, y) in enumerate(zip(pa, pb)):
        assert np.array_equal(x[0], y[0]), f"sorted codes of step {k}"
        assert np.array_equal(x[1], y[1]), f"permutation of step {k}"
        assert np.array_equal(x[2], y[2]), f"interaction lists of step {k}"
    for key in ("id", "x", "y", "z", "vx", "vy", "vz", "ax", "ay", "az"):
        assert np.array_equal(sa[key], sb[key]), key


def slow_sphere(n, seed):
    """Sphere at rest with total mass 1: nothing moves further than a few cells a step."""
    ic = nb.sample_ic(nb.IC_SPHERE, n, seed)
    ic["m"] = np.full(n, 1.0 / n, np.float32)
    return ic


@pytest.mark.parametrize("case", ["many_moved", "few_moved", "none_moved", "fp64"])
def test_nearly_sorted_path_gives_the_permutation_of_the_digit_passes(case):
    if case == "many_moved":      # the reference's disk: about a fifth of the elements leave their place
        n, steps = 150000, 6
        ic = nb.sample_ic(nb.IC_DISK, n, 11)
        kw = dict(theta=0.7, dt=1e-4, softening_factor=1e-3)
    elif case == "few_moved":
        n, steps = 300000, 8
        ic = slow_sphere(n, 5)
        rng = np.random.default_rng(1)
        for k in ("vx", "vy", "vz"):
            ic[k] = rng.normal(0, 1.0, n).astype(np.float32)
        kw = dict(theta=0.7, dt=2e-6, softening_factor=0.05)
    elif case == "none_moved":    # cold start: keys do not change at all for the first steps
        n, steps = 200000, 4
        ic = slow_sphere(n, 7)
        ic["vx"] = ic["vy"] = ic["vz"] = np.zeros(n, np.float32)
        kw = dict(theta=0.7, dt=1e-7, softening_factor=0.05)
    else:
        n, steps = 100000, 5
        ic = nb.sample_ic(nb.IC_DISK, n, 3)
        kw = dict(theta=0.7, dt=2e-5, softening_factor=1e-3)
    extra = nb.FLAG_FP64 if case == "fp64" else 0
    full = run(ic, n, steps, nb.FLAG_NO_COHERENT_SORT | extra, **kw)
    forced = run(ic, n, steps, nb.FLAG_ALWAYS_COHERENT_SORT | extra, **kw)
    assert full[2]["coherent_sorts"] == 0 and full[2]["full_sorts"] == steps + 1
    assert forced[2]["coherent_sorts"] == steps and forced[2]["full_sorts"] == 1
    moved = forced[2]["last_moved"]
    if case == "many_moved":
        assert moved > n // 50
    elif case == "few_moved":
        assert 0 < moved < n // 50
    elif case == "none_moved":
        assert moved == 0
    assert_same_run(full, forced)
    auto = run(ic, n, steps, extra, **kw)     # whichever way each step's sort went
    assert_same_run(full, auto)


def test_nearly_sorted_path_with_equal_keys_and_long_runs_against_the_oracle():
    """Thousands of particles in one cell of domain / 2^13, thousands on one point (equal keys: index
    order decides), and a background, all drifting slowly: every step's codes and permutation
    must be the oracle's stable sort of that step's positions."""
    rng = np.random.default_rng(9)
    n = 50000
    p = rng.uniform(0.5, 4.5, (n, 3))
    cell = 5.0 / 8192
    corner = np.floor(np.array([1.7, 2.9, 3.3]) / cell) * cell
    p[:6000] = corner + rng.uniform(0.02, 0.98, (6000, 3)) * cell
    p[6000:9000] = np.array([3.125, 0.625, 2.0])
    p = p[rng.permutation(n)].astype(np.float32)
    v = rng.normal(0, 0.3, (n, 3)).astype(np.float32)
    v[np.all(p == np.array([3.125, 0.625, 2.0], np.float32), axis=1)] = 0     # they stay on one point
    m = np.full(n, 1e-9, np.float32)                                          # (no forces to speak of)
    par = nb.Params(num_points=n, morton_bits=21, theta=0.5, dt=2e-5, softening_factor=0.05,
                    flags=nb.FLAG_ALWAYS_COHERENT_SORT)
    with nb.Simulation(par) as sim:
        sim.set_state(p[:, 0].copy(), p[:, 1].copy(), p[:, 2].copy(), m, v[:, 0].copy(), v[:, 1].copy(), v[:, 2].copy())
        for step in range(4):
            before = sim.get_state()
            sim.update()
            o = sim.get_octree()
            after = sim.get_state()
            # the oracle's sort of the positions this update arrived at, taken in the order the
            # update found the particles in: ids connect the two orders
            inv = np.empty(n, np.int64)
            inv[after["id"]] = np.arange(n)
            at = inv[before["id"]]          # where particle i of the old order sits now
            ref = O.full_build(after["x"][at], after["y"][at], after["z"][at], after["m"][at], 5.0, 21, 32)
            assert np.array_equal(o["codes_sorted"], ref["codes_sorted"]), step
            assert np.array_equal(o["perm"], ref["perm"]), step
        stats = sim.sort_stats()
    assert stats["coherent_sorts"] == 4 and stats["last_moved"] > 0


def test_nearly_sorted_path_with_late_workgroups():
    """The digit passes over the moved elements run with fewer workgroups than tiles (a workgroup
    goes on to its next tile) and with the helping look-back: late workgroups change nothing."""
    n = 1600000
    ic = nb.sample_ic(nb.IC_DISK, n, 2)
    kw = dict(theta=0.9, dt=2e-4, softening_factor=1e-3)
    a = run(ic, n, 3, nb.FLAG_NO_COHERENT_SORT, **kw)
    b = run(ic, n, 3, nb.FLAG_ALWAYS_COHERENT_SORT | nb.FLAG_DEBUG_LATE_TILES, **kw)
    assert b[2]["coherent_sorts"] == 3
    assert b[2]["last_moved"] > 256 * 4096      # more tiles of moved elements than workgroups
    assert_same_run(a, b)
