"""GPU parity of the reflecting walls and of velocity damping (reference:
impose_boundary_conditions and leapfrog_integrate_pos/vel, src/cuda/physics_common.cu:69-147):
a coordinate that leaves [0, domain] flips its velocity component and is clamped to the wall;
the kick that ends a step multiplies the velocity by (1 - velocity_dampening).  Particles are
driven through each of the six walls in one update(), with domain_size and
velocity_dampening away from the defaults, on float and double handles."""
import numpy as np
import pytest

import nbody_amd as nb
import oracle_lib as O

pytestmark = pytest.mark.gpu

DOMAIN, DAMP, DT, EPS, THETA = 3.0, 0.2, 1e-2, 0.05, 0.5


def wall_ic(n, dtype, seed=3):
    """A cloud in the middle of the box plus, for every wall, a batch of particles closer to it
    than their velocity carries them in one step (some exactly on the wall, moving outwards),
    and a few that come to rest exactly on a wall or stop just short of it."""
    rng = np.random.default_rng(seed)
    p = rng.uniform(0.8, 2.2, (n, 3))
    v = rng.normal(0.0, 0.5, (n, 3))
    k = 24
    i = 0
    crossing = {}
    for axis in range(3):
        for side in (0, 1):
            idx = np.arange(i, i + k)
            i += k
            speed = rng.uniform(2.0, 9.0, k)                  # moves 0.02 .. 0.09 in one step
            gap = rng.uniform(0.0, 0.9, k) * speed * DT      # < speed * dt: crosses
            gap[:3] = 0.0                                     # on the wall, moving out
            p[idx, axis] = gap if side == 0 else DOMAIN - gap
            v[idx, axis] = -speed if side == 0 else speed
            crossing[(axis, side)] = idx
    # not crossing: heading for the wall but stopping short of it
    idx = np.arange(i, i + k)
    p[idx, 0] = 0.5
    v[idx, 0] = -rng.uniform(1.0, 9.0, k)
    m = rng.uniform(0.5, 1.5, n) * 1e-3            # |a| dt stays small beside the wall speeds
    ic = dict(x=p[:, 0], y=p[:, 1], z=p[:, 2], vx=v[:, 0], vy=v[:, 1], vz=v[:, 2], m=m)
    return {q: np.ascontiguousarray(a, dtype) for q, a in ic.items()}, crossing


@pytest.mark.parametrize("bits,group", [(21, 64), (10, 32)])
def test_walls_and_damping_float(bits, group):
    n = 6000
    ic, crossing = wall_ic(n, np.float32)
    osim = O.OracleSim(ic["x"], ic["y"], ic["z"], ic["m"], ic["vx"], ic["vy"], ic["vz"], bits=bits,
                       max_codes=32, group=group, theta=THETA, dt=DT, gravity=1.0, eps=EPS,
                       domain=DOMAIN, damp=DAMP)
    p = nb.Params(num_points=n, morton_bits=bits, traversal_group_size=group, theta=THETA, dt=DT,
                  softening_factor=EPS, domain_size=DOMAIN, velocity_dampening=DAMP)
    with nb.Simulation(p) as sim:
        sim.set_state(ic["x"], ic["y"], ic["z"], ic["m"], ic["vx"], ic["vy"], ic["vz"])
        sim.update()
        osim.step()
        g1, o1 = sim.get_state(), osim.state()
        t1 = sim.get_octree()
        sim.update()
        osim.step()
        g2, o2 = sim.get_state(), osim.state()
    # the first kick-drift has a = 0 on both sides: positions, hence codes, order and ids, are
    # the same bits
    assert np.array_equal(g1["id"], o1["id"])
    for q in ("x", "y", "z", "m"):
        assert np.array_equal(g1[q], o1[q]), q
    # every batch really went through its wall: clamped onto it, velocity component reversed
    where = np.empty(n, np.int64)
    where[g1["id"]] = np.arange(n)
    keep = np.float32(1.0) - np.float32(DAMP)
    for (axis, side), idx in crossing.items():
        c = g1["xyz"[axis]][where[idx]]
        wall = np.float32(0.0 if side == 0 else DOMAIN)
        assert np.all(c == wall), (axis, side)
        a = g1["a" + "xyz"[axis]][where[idx]]
        vh = -ic["v" + "xyz"[axis]][idx]                       # reflected half-step velocity
        want = (vh.astype(np.float64) + 0.5 * a.astype(np.float64) * np.float64(np.float32(DT))) * keep
        got = g1["v" + "xyz"[axis]][where[idx]]
        assert np.all(np.abs(got - want) <= 3e-7 * (np.abs(vh) + np.abs(a) * DT) + 1e-30), (axis, side)
    assert np.all(g1["x"] >= 0) and np.all(g1["x"] <= np.float32(DOMAIN))
    assert t1["size"][0] == np.float32(DOMAIN)
    # velocities and accelerations: same sources, float32 sums in a different order
    for q in ("ax", "ay", "az"):
        scale = np.sqrt(o1["ax"].astype(np.float64) ** 2 + o1["ay"] ** 2 + o1["az"] ** 2)
        assert np.max(np.abs(g1[q] - o1[q]) / scale) < 1e-4, q
    amag = np.sqrt(o1["ax"].astype(np.float64) ** 2 + o1["ay"] ** 2 + o1["az"] ** 2)
    for q in ("vx", "vy", "vz"):
        # the kick adds a dt / 2 with accelerations that agree to 1e-4 |a|
        assert np.all(np.abs(g1[q] - o1[q]) <= 1e-4 * 0.5 * DT * amag + 1e-6 * np.abs(o1[q]) + 1e-7), q
    # second step: the kick-drift now uses the forces (equal to 1e-4), more wall contacts
    # (particles left on a wall with an outward kick); ids still line up
    assert np.array_equal(np.sort(g2["id"]), np.arange(n, dtype=np.uint32))
    gi, oi = np.argsort(g2["id"]), np.argsort(o2["id"])
    for q in ("x", "y", "z"):
        assert np.allclose(g2[q][gi], o2[q][oi], rtol=0, atol=2e-5), q
    for q in ("vx", "vy", "vz"):
        assert np.allclose(g2[q][gi], o2[q][oi], rtol=1e-3, atol=1e-3), q


def test_walls_and_damping_double():
    n, bits, group = 6000, 21, 64
    ic, crossing = wall_ic(n, np.float64, seed=4)
    osim = O.OracleSimDP(ic["x"], ic["y"], ic["z"], ic["m"], ic["vx"], ic["vy"], ic["vz"], bits=bits,
                         max_codes=32, group=group, theta=THETA, dt=DT, gravity=1.0, eps=EPS,
                         domain=DOMAIN, damp=DAMP)
    p = nb.Params(num_points=n, morton_bits=bits, traversal_group_size=group, theta=THETA, dt=DT,
                  softening_factor=EPS, domain_size=DOMAIN, velocity_dampening=DAMP,
                  flags=nb.FLAG_FP64)
    with nb.Simulation(p) as sim:
        sim.set_state(ic["x"], ic["y"], ic["z"], ic["m"], ic["vx"], ic["vy"], ic["vz"])
        sim.update()
        osim.step()
        g1, o1 = sim.get_state(), osim.state()
        sim.update()
        osim.step()
        g2, o2 = sim.get_state(), osim.state()
    assert np.array_equal(g1["id"], o1["id"])
    for q in ("x", "y", "z", "m"):
        assert np.array_equal(g1[q], o1[q]), q
    where = np.empty(n, np.int64)
    where[g1["id"]] = np.arange(n)
    for (axis, side), idx in crossing.items():
        c = g1["xyz"[axis]][where[idx]]
        assert np.all(c == (0.0 if side == 0 else float(np.float32(DOMAIN)))), (axis, side)
        # reversed and damped: sign opposite to the initial velocity
        got = g1["v" + "xyz"[axis]][where[idx]]
        assert np.all(np.sign(got) == -np.sign(ic["v" + "xyz"[axis]][idx])), (axis, side)
    scale = np.sqrt(o1["ax"] ** 2 + o1["ay"] ** 2 + o1["az"] ** 2)
    for q in ("ax", "ay", "az"):
        assert np.max(np.abs(g1[q] - o1[q]) / scale) < 1e-12, q
    for q in ("vx", "vy", "vz"):
        assert np.max(np.abs(g1[q] - o1[q])) < 1e-12 * max(1.0, np.abs(o1[q]).max()), q
    assert np.array_equal(g2["id"], o2["id"])
    for q in ("x", "y", "z", "vx", "vy", "vz"):
        assert np.max(np.abs(g2[q] - o2[q])) < 1e-10, q
