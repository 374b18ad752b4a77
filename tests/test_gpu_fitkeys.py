"""GPU parity of the fitted key domain (NBODY_FLAG_FIT_KEYS): Morton keys relative to a cube
around the particles instead of the reference's [0, domain_size]^3 (src/cuda/points.cu:28-30).
The host oracle applies the same rule (oracle.cpp, key_fit); cube, codes, tree, centres of mass
and list lengths are compared bit for bit -- at the spawn, after a first update whose drift
leaves the cube (re-draw), after one that contracts the system below 0.4 of the cube
(re-draw), and after one that does neither (cube kept)."""
import numpy as np
import pytest

import nbody_amd as nb
import oracle_lib as O

pytestmark = pytest.mark.gpu


def compare(sim, osim, group, check_lists):
    g = sim.get_octree()
    t = osim.tree()
    assert np.array_equal(sim.key_domain(), osim.key_domain())
    assert np.array_equal(g["codes_sorted"], t["codes_sorted"])
    assert g["n_oct"] == t["n_oct"]
    inner = t["num_children"] > 0
    assert np.array_equal(g["num_children"], t["num_children"])
    assert np.array_equal(g["first_child"][inner], t["first_child"][inner])
    for a, b in (("points_begin", "points_begin"), ("points_end", "points_end"), ("size", "size"),
                 ("comx", "comx"), ("comy", "comy"), ("comz", "comz"), ("mass", "mass")):
        assert np.array_equal(g[a], t[b]), a
    assert g["size"][0] == sim.key_domain()[3]          # the root cell is the cube
    if check_lists:
        assert np.array_equal(g["list_len"], t["list_len"])


@pytest.mark.parametrize("bits,group", [(10, 64), (21, 32)])
@pytest.mark.parametrize("motion", ["still", "expand", "contract"])
def test_fitted_keys_match_oracle(bits, group, motion):
    n, theta, eps, dt = 30000, 0.6, 0.01, 1e-2
    ic = nb.sample_ic(nb.IC_DISK, n, 21)
    c = np.float32(2.5)
    scale = {"still": 0.0, "expand": 0.5 / dt, "contract": -0.7 / dt}[motion]
    v = {k: ((ic[q] - c) * np.float32(scale)).astype(np.float32) for k, q in (("vx", "x"), ("vy", "y"), ("vz", "z"))}
    kw = dict(theta=theta, dt=dt, softening_factor=eps)
    osim = O.OracleSim(ic["x"], ic["y"], ic["z"], ic["m"], v["vx"], v["vy"], v["vz"], bits=bits, max_codes=32,
                       group=group, theta=theta, dt=dt, eps=eps, fit=True)
    p = nb.Params(num_points=n, morton_bits=bits, traversal_group_size=group, flags=nb.FLAG_FIT_KEYS, **kw)
    with nb.Simulation(p) as sim:
        sim.set_state(ic["x"], ic["y"], ic["z"], ic["m"], v["vx"], v["vy"], v["vz"])
        cube0 = sim.key_domain()
        compare(sim, osim, group, False)
        # the cube hugs the disk: 1.25 x its diameter of ~0.6, not the domain of 5
        assert 0.7 < cube0[3] < 0.8 and np.all(cube0[:3] > 2.0)
        sim.update()
        osim.step()
        cube1 = sim.key_domain()
        compare(sim, osim, group, True)
        st, so = sim.get_state(), osim.state()
        assert np.array_equal(st["x"], so["x"]) and np.array_equal(st["id"], so["id"])
    if motion == "still":
        assert np.array_equal(cube0, cube1)
    elif motion == "expand":
        assert cube1[3] > 1.4 * cube0[3]
    else:
        assert cube1[3] < 0.5 * cube0[3]
    # every particle inside the cube
    for k, a in enumerate("xyz"):
        assert st[a].min() >= cube1[k] and st[a].max() <= cube1[k] + cube1[3]


def test_fitted_keys_resolve_what_the_domain_cube_cannot():
    """30-bit keys on the 524,288-particle disk: relative to the domain of 5.0 the disk is ~120
    cells across and an eighth of the particles have a code of their own; relative to the
    fitted cube nearly all do."""
    n = 524288
    ic = nb.sample_ic(nb.IC_DISK, n, 42)
    U = {}
    for flags in (0, nb.FLAG_FIT_KEYS):
        with nb.Simulation(nb.Params(num_points=n, morton_bits=10, theta=0.75, dt=1e-4, softening_factor=1e-3,
                                     flags=flags)) as sim:
            sim.set_state(ic["x"], ic["y"], ic["z"], ic["m"], ic["vx"], ic["vy"], ic["vz"])
            sim.step(2)
            U[flags] = sim.tree_counts()[0]
            st = sim.get_state()
            assert np.isfinite(st["ax"]).all()
    assert U[0] < 0.2 * n and U[nb.FLAG_FIT_KEYS] > 0.7 * n, U


def test_fitted_keys_rejected_where_unsupported():
    import ctypes as C
    lib = nb.load_library()
    p = nb.Params(num_points=64, flags=nb.FLAG_FIT_KEYS | nb.FLAG_FP64).to_c()
    h = C.c_void_p()
    assert lib.nbody_create(C.byref(p), C.byref(h)) == 1
