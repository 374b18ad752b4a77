"""GPU parity tests of the double-precision handle (NBODY_FLAG_FP64, the reference's
Simulation<double>, src/cuda/simulation.cu:256) against the double instantiation of the host
oracle.  Integer stages, centres of mass and interaction lists bit for bit; accelerations
with the tolerance written at each assert."""
import numpy as np
import pytest

import nbody_amd as nb
import oracle_lib as O
from test_gpu_parity import assert_build_equal

pytestmark = pytest.mark.gpu


def make_ic64(kind, n, seed=42):
    """Double-precision particles that are NOT representable in float."""
    if kind == "clusters":
        rng = np.random.default_rng(seed)
        c = rng.uniform(1, 4, (7, 3))
        p = (c[rng.integers(0, 7, n)] + rng.normal(0, 0.03, (n, 3))).clip(0, 5)
        v = rng.normal(0, 0.1, (n, 3))
        return dict(x=p[:, 0].copy(), y=p[:, 1].copy(), z=p[:, 2].copy(), vx=v[:, 0].copy(),
                    vy=v[:, 1].copy(), vz=v[:, 2].copy(), m=rng.uniform(0.5, 1.5, n))
    code = {"disk": nb.IC_DISK, "sphere": nb.IC_SPHERE, "uniform": nb.IC_UNIFORM, "plummer": nb.IC_PLUMMER}[kind]
    return nb.sample_ic(code, n, seed, p0=0.1, p1=1.0, dtype=np.float64)


def new_sim64(n, **kw):
    kw.setdefault("softening_factor", 0.01)
    kw["flags"] = kw.get("flags", 0) | nb.FLAG_FP64
    return nb.Simulation(nb.Params(num_points=n, **kw))


def rel_err(a, r):
    """max_i |a_i - r_i| / |r_i| over bodies, a and r = (x, y, z) component arrays."""
    d = np.sqrt(sum((np.asarray(p, np.float64) - q) ** 2 for p, q in zip(a, r)))
    return float(np.max(d / np.sqrt(sum(q ** 2 for q in r))))


@pytest.mark.parametrize("bits", [10, 21])
@pytest.mark.parametrize("kind,n,max_codes", [
    ("disk", 32768, 32), ("sphere", 20000, 32), ("clusters", 1000, 1), ("clusters", 131072, 16),
    ("plummer", 65536, 32),
])
def test_fp64_build_bit_exact(bits, kind, n, max_codes):
    ic = make_ic64(kind, n)
    assert ic["x"].dtype == np.float64 and np.any(ic["x"] != ic["x"].astype(np.float32))
    with new_sim64(n, morton_bits=bits, max_num_codes_per_leaf=max_codes) as sim:
        sim.set_state(ic["x"], ic["y"], ic["z"], ic["m"], ic["vx"], ic["vy"], ic["vz"])
        gpu = sim.get_octree()
        st = sim.get_state()
    ref = O.full_build_dp(ic["x"], ic["y"], ic["z"], ic["m"], 5.0, bits, max_codes)
    assert gpu["comx"].dtype == np.float64
    assert_build_equal(gpu, ref, exact_com=True)
    # the state is the input, permuted, bit for bit
    assert st["x"].dtype == np.float64
    for k in ("x", "y", "z", "m", "vx", "vy", "vz"):
        assert np.array_equal(st[k], ic[k][ref["perm"]]), k
    assert np.array_equal(st["id"], ref["perm"].astype(np.uint32))


@pytest.mark.parametrize("group", [64, 32])
@pytest.mark.parametrize("kind,n,theta,bits", [
    ("disk", 32768, 0.5, 21), ("clusters", 50000, 0.75, 21), ("sphere", 30001, 0.6, 10),
    ("clusters", 777, 0.3, 21), ("uniform", 40, 0.5, 10),
])
def test_fp64_traversal_lists_and_forces(group, kind, n, theta, bits):
    """One update() with dt = 0 leaves the particles where they are and evaluates the forces:
    per-group source counts equal the oracle's, accelerations agree to 1e-12 (the oracle sums
    a group's sources depth-first, the kernel in stack order; 1/sqrt differs by an ulp)."""
    ic = make_ic64(kind, n)
    eps = 0.01
    with new_sim64(n, morton_bits=bits, traversal_group_size=group, theta=theta, dt=0.0,
                   softening_factor=eps) as sim:
        sim.set_state(ic["x"], ic["y"], ic["z"], ic["m"])
        sim.update()
        gpu = sim.get_octree()
        st = sim.get_state()
    ref = O.full_build_dp(ic["x"], ic["y"], ic["z"], ic["m"], 5.0, bits, 32)
    # the step re-sorted particles that were already in order: its permutation is the identity
    assert np.array_equal(gpu["perm"], np.arange(n))
    gpu["perm"] = ref["perm"]
    assert_build_equal(gpu, ref, exact_com=True)
    ax, ay, az, ll = O.traverse_dp(ref["tree"], ref["com"], ref["xs"], ref["ys"], ref["zs"], ref["ms"],
                                   group, theta, eps, 1.0, 5.0)
    assert np.array_equal(gpu["list_len"], ll)
    assert rel_err((st["ax"], st["ay"], st["az"]), (ax, ay, az)) < 1e-12
    # dt = 0: v = (v_half + 0) * 1 = 0 and positions untouched
    assert np.array_equal(st["x"], ref["xs"]) and not st["vx"].any()


def test_fp64_trajectory_matches_oracle_sim():
    """Five steps against the oracle's Simulation<double>: tree sizes equal, positions and
    velocities to 1e-11 of the domain (forces differ in the last bits, see above)."""
    n = 20000
    ic = make_ic64("clusters", n, seed=3)
    kw = dict(theta=0.6, dt=1e-3, gravity=1.0, domain=5.0)
    osim = O.OracleSimDP(ic["x"], ic["y"], ic["z"], ic["m"], ic["vx"], ic["vy"], ic["vz"], bits=21,
                         max_codes=32, group=64, eps=0.01, **kw)
    with new_sim64(n, morton_bits=21, theta=0.6, dt=1e-3, softening_factor=0.01) as sim:
        sim.set_state(ic["x"], ic["y"], ic["z"], ic["m"], ic["vx"], ic["vy"], ic["vz"])
        for _ in range(5):
            sim.update()
            osim.step()
        st = sim.get_state()
        U, M, no = sim.tree_counts()
    ref = osim.state()
    assert (U, M, no) == osim.counts()[:3]
    assert np.array_equal(st["id"], ref["id"])
    for k in ("x", "y", "z", "vx", "vy", "vz"):
        assert np.max(np.abs(st[k] - ref[k])) < 5e-11, k
    assert rel_err((st["ax"], st["ay"], st["az"]), (ref["ax"], ref["ay"], ref["az"])) < 1e-10


def test_fp64_direct_sum_and_tree_accuracy():
    n = 6000
    ic = make_ic64("plummer", n, seed=5)
    eps = 0.01
    with new_sim64(n, morton_bits=21, theta=0.5, dt=0.0, softening_factor=eps) as sim:
        sim.set_state(ic["x"], ic["y"], ic["z"], ic["m"])
        sim.update()
        st = sim.get_state()
        d = sim.direct_sum()
        d32 = sim.direct_sum(dtype=np.float32)
    assert d[0].dtype == np.float64 and d32[0].dtype == np.float32
    ref = O.direct_dp(st["x"], st["y"], st["z"], st["m"], eps, 1.0)
    assert rel_err(d, ref) < 1e-13       # same source order; 1/sqrt within an ulp or two
    assert rel_err(d32, ref) < 1e-6      # the narrowed copy
    # theta = 0.5 monopole tree against the direct sum: the method's error, not rounding
    e = rel_err((st["ax"], st["ay"], st["az"]), ref)
    assert 1e-8 < e < 2e-2
    # a float handle's direct sum evaluated in double over its float state
    x32, y32, z32, m32 = (ic[k].astype(np.float32) for k in ("x", "y", "z", "m"))
    with nb.Simulation(nb.Params(num_points=n, morton_bits=21, softening_factor=eps)) as sim:
        sim.set_state(x32, y32, z32, m32)
        s32 = sim.get_state()
        dd = sim.direct_sum(dtype=np.float64)
    # the library squares the float parameter widened to double
    ref32 = O.direct_f64(s32["x"], s32["y"], s32["z"], s32["m"], float(np.float32(eps)), 1.0)
    assert rel_err(dd, ref32) < 1e-13


@pytest.mark.parametrize("group,n", [(64, 3000), (32, 3000), (32, 2967)])
def test_fp64_theta_zero_is_the_direct_sum(group, n):
    ic = make_ic64("sphere", n)
    with new_sim64(n, morton_bits=21, theta=0.0, dt=0.0, softening_factor=0.05,
                   traversal_group_size=group) as sim:
        sim.set_state(ic["x"], ic["y"], ic["z"], ic["m"])
        sim.update()
        st = sim.get_state()
        d = sim.direct_sum()
        ll = sim.get_octree()["list_len"]
        ke, pe = sim.tree_energy()
    assert np.all(ll == n)
    assert rel_err((st["ax"], st["ay"], st["az"]), d) < 1e-12
    # energy through the tree at theta = 0 is the all-pairs energy
    x, y, z, m = (st[k] for k in ("x", "y", "z", "m"))
    r2 = (x[:, None] - x) ** 2 + (y[:, None] - y) ** 2 + (z[:, None] - z) ** 2 + np.float64(np.float32(0.05)) ** 2
    inv = 1.0 / np.sqrt(r2)
    np.fill_diagonal(inv, 0.0)
    pe_ref = -0.5 * np.sum(m[:, None] * m * inv)
    assert ke == 0.0
    assert abs(pe - pe_ref) < 1e-12 * abs(pe_ref)


def test_fp64_validator_records():
    """Validator<double>: energies of the tree run and of the all-pairs twin, step-local
    acceleration error; at theta = 0 the two runs are the same computation."""
    n = 2048
    ic = make_ic64("disk", n)
    with new_sim64(n, morton_bits=21, theta=0.0, dt=1e-4, softening_factor=0.01, num_steps_validator=8) as sim:
        sim.set_state(ic["x"], ic["y"], ic["z"], ic["m"], ic["vx"], ic["vy"], ic["vz"])
        sim.step(4)
        h = sim.validation_history()
    assert len(h["sys_energy"]) == 4
    assert np.all(h["avg_acc_err"] < 1e-12)
    assert np.allclose(h["sys_energy"], h["sys_energy_ap"], rtol=1e-11, atol=0)
    # four leapfrog steps at dt = 1e-4 (1 % of the innermost orbit) hold the energy
    assert abs(h["sys_energy"][-1] / h["sys_energy"][0] - 1.0) < 1e-4
    with new_sim64(n, morton_bits=21, theta=0.6, dt=1e-4, softening_factor=0.01, num_steps_validator=8) as sim:
        sim.set_state(ic["x"], ic["y"], ic["z"], ic["m"], ic["vx"], ic["vy"], ic["vz"])
        sim.step(3)
        h = sim.validation_history()
    assert np.all((h["avg_acc_err"] > 1e-7) & (h["avg_acc_err"] < 1e-2))


def test_fp64_float_arguments_are_converted_and_snapshot_restarts(tmp_path):
    n = 5000
    ic = make_ic64("clusters", n, seed=9)
    path = str(tmp_path / "snap64.bin")
    with new_sim64(n, morton_bits=21, dt=1e-3) as sim:
        # float arrays in: widened exactly
        f = {k: v.astype(np.float32) for k, v in ic.items()}
        sim.set_state(f["x"], f["y"], f["z"], f["m"], f["vx"], f["vy"], f["vz"], dtype=np.float32)
        st = sim.get_state()
        assert np.array_equal(np.sort(st["x"]), np.sort(f["x"].astype(np.float64)))
        # float arrays out: narrowed
        assert np.array_equal(sim.get_state(dtype=np.float32)["x"], st["x"].astype(np.float32))
        sim.set_state(ic["x"], ic["y"], ic["z"], ic["m"], ic["vx"], ic["vy"], ic["vz"])
        sim.step(3)
        sim.save_snapshot(path)
        sim.step(3)
        want = sim.get_state()
    with new_sim64(n, morton_bits=21, dt=1e-3) as sim:
        sim.load_snapshot(path)
        sim.step(3)
        got = sim.get_state()
    for k in want:
        assert np.array_equal(want[k], got[k]), k
    # a float handle reads the double file with conversion
    with nb.Simulation(nb.Params(num_points=n, morton_bits=21, dt=1e-3, softening_factor=0.01)) as sim:
        sim.load_snapshot(path)
        assert sim.get_state()["x"].dtype == np.float32


def test_fp64_sharded_traversal_equals_whole():
    """nbody_set_traversal_range / nbody_finish_step on a double handle: two group ranges computed
    one after the other give the accelerations and velocities of the single-GPU step, bit for bit."""
    n = 10000
    ic = make_ic64("clusters", n, seed=5)
    args = (ic["x"], ic["y"], ic["z"], ic["m"], ic["vx"], ic["vy"], ic["vz"])
    with new_sim64(n, morton_bits=21, theta=0.5, dt=1e-3) as sim:
        sim.set_state(*args)
        sim.update()
        whole = sim.get_state()
    ngroups = (n + 63) // 64
    for g0, g1 in ((0, ngroups // 2), (ngroups // 2, ngroups)):
        with new_sim64(n, morton_bits=21, theta=0.5, dt=1e-3) as sim:
            sim.set_state(*args)
            sim.set_traversal_range(g0, g1)
            sim.update()
            sim.finish_step()
            st = sim.get_state()
        b0, b1 = g0 * 64, min(g1 * 64, n)
        for k in ("ax", "ay", "az", "vx"):
            assert np.array_equal(st[k][b0:b1], whole[k][b0:b1]), k


def test_fp64_domain_decomposition_calls_are_refused_and_views():
    n = 1000
    ic = make_ic64("uniform", n)
    with new_sim64(n) as sim:
        sim.set_state(ic["x"], ic["y"], ic["z"], ic["m"])
        with pytest.raises(RuntimeError, match="double-precision"):
            sim.get_bbox()
        with pytest.raises(RuntimeError, match="double-precision"):
            sim.traverse_from(sim, accumulate=False)
        v32, v64 = sim.device_view(), sim.device_view_f64()
        assert v32.n == n and not v32.x and not v32.posm
        assert v64.n == n and v64.x and v64.posm and v64.stream


def test_fp64_external_double_buffers():
    """Simulation<double>(Params&, buffers[7]): the viewer buffers hold doubles."""
    import torch
    n = 4096
    ic = make_ic64("sphere", n)
    bufs = [torch.full((n if i < 3 else 2 * n,), -1.0, dtype=torch.float64, device="cuda") for i in range(7)]
    p = nb.Params(num_points=n, morton_bits=21, flags=nb.FLAG_FP64, softening_factor=0.01)
    with nb.Simulation(p, external=[b.data_ptr() for b in bufs]) as sim:
        sim.set_state(ic["x"], ic["y"], ic["z"], ic["m"])
        sim.update()
        st = sim.get_state()
        oc = sim.get_octree()
    torch.cuda.synchronize()
    no = oc["n_oct"]
    assert np.array_equal(bufs[0].cpu().numpy(), st["x"])
    assert np.array_equal(bufs[2].cpu().numpy(), st["z"])
    assert np.array_equal(bufs[3].cpu().numpy()[:no], oc["comx"])
    assert np.array_equal(bufs[6].cpu().numpy()[:no], oc["size"].astype(np.float64))
    assert np.all(bufs[6].cpu().numpy()[no:] == -1.0)


def test_fp64_spawn_and_small_cases():
    """spawn_* draw doubles; N = 1 and a single leaf work."""
    with new_sim64(3000, morton_bits=21) as sim:
        sim.spawn_disk()
        st = sim.get_state()
        assert np.any(st["x"] != st["x"].astype(np.float32))
        sim.step(2)
        assert np.all(np.isfinite(sim.get_state()["ax"]))
    for n in (1, 2, 33):
        x = np.linspace(1.0, 1.001, n)
        with new_sim64(n, morton_bits=10, dt=0.0, softening_factor=0.1) as sim:
            sim.set_state(x, x, x, np.ones(n))
            sim.update()
            st = sim.get_state()
            d = sim.direct_sum()
            assert sim.get_num_octree_nodes() == 1
        if n == 1:
            assert st["ax"][0] == 0.0
        else:
            assert rel_err((st["ax"], st["ay"], st["az"]), d) < 1e-12


@pytest.mark.parametrize("group", [64, 32])
def test_fp64_tiny_stack_gives_the_same_lists(group):
    """NBODY_FLAG_DEBUG_SMALL_STACK: 32 usable stack entries force the depth-first single-node
    rounds; source lists are unchanged, sums differ only by their order."""
    n = 60000
    ic = make_ic64("clusters", n, seed=17)
    res = []
    for flags in (0, nb.FLAG_DEBUG_SMALL_STACK):
        with new_sim64(n, morton_bits=21, theta=0.5, dt=0.0, traversal_group_size=group, flags=flags) as sim:
            sim.set_state(ic["x"], ic["y"], ic["z"], ic["m"])
            sim.update()
            sim.sync()
            res.append((sim.get_octree()["list_len"], sim.get_state()))
    assert np.array_equal(res[0][0], res[1][0])
    a, b = res[0][1], res[1][1]
    assert rel_err((a["ax"], a["ay"], a["az"]), (b["ax"], b["ay"], b["az"])) < 1e-12


def test_fp64_headline_workload_lists_match_oracle():
    """The headline workload (2,097,152-particle sphere, theta = 0.6, 63-bit keys) on a double
    handle: every tree array, every group's list length, accelerations to 1e-12."""
    n, theta, eps = 1 << 21, 0.6, 0.1
    ic = make_ic64("sphere", n)
    with new_sim64(n, morton_bits=21, theta=theta, dt=0.0, softening_factor=eps) as sim:
        sim.set_state(ic["x"], ic["y"], ic["z"], ic["m"])
        sim.update()
        gpu = sim.get_octree()
        st = sim.get_state()
    ref = O.full_build_dp(ic["x"], ic["y"], ic["z"], ic["m"], 5.0, 21, 32)
    assert np.array_equal(gpu["perm"], np.arange(n))
    gpu["perm"] = ref["perm"]
    assert_build_equal(gpu, ref, exact_com=True)
    ax, ay, az, ll = O.traverse_dp(ref["tree"], ref["com"], ref["xs"], ref["ys"], ref["zs"], ref["ms"],
                                   64, theta, eps, 1.0, 5.0)
    assert np.array_equal(gpu["list_len"], ll)
    assert rel_err((st["ax"], st["ay"], st["az"]), (ax, ay, az)) < 1e-12


def test_fp64_library_against_committed_fixture():
    """tests/golden/double (written by make_golden.py from the oracle's double instantiation)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "double", "clusters_dp_b21_g64.npz"))
    bits, max_codes, group = (int(v) for v in g["params"])
    theta, eps = float(g["fparams"][0]), float(g["fparams"][1])
    p, m = g["pos"], g["mass"]
    with new_sim64(len(m), morton_bits=bits, max_num_codes_per_leaf=max_codes, traversal_group_size=group,
                   theta=theta, softening_factor=eps, dt=0.0) as sim:
        sim.set_state(p[:, 0], p[:, 1], p[:, 2], m)
        t = sim.get_octree()
        sim.update()
        ll = sim.get_list_len()
        st = sim.get_state()
    for k in ("codes_sorted", "perm", "leaf_first_code_idx"):
        assert np.array_equal(t[k], g[k]), k
    for k in ("num_children", "points_begin", "points_end", "level"):
        assert np.array_equal(t[k], g["tree_" + k]), k
    for k in ("comx", "comy", "comz", "mass"):
        assert np.array_equal(t[k], g["com_" + k]), k
    assert np.array_equal(ll, g["list_len"])
    assert rel_err((st["ax"], st["ay"], st["az"]), (g["ax"], g["ay"], g["az"])) < 1e-12
