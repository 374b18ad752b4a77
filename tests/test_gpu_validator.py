"""GPU tests of the validation path (reference: Validator<T>, src/cuda/validator.cu): energies
of the tree run and of the all-pairs twin, step-local relative acceleration error, CSV."""
import os

import numpy as np
import pytest

import nbody_amd as nb
import oracle_lib as O

pytestmark = pytest.mark.gpu


def test_validator_records_match_oracle(tmp_path):
    n, eps, dt, theta = 4096, 0.01, 1e-3, 0.5
    ic = nb.sample_ic(nb.IC_DISK, n, 42)
    with nb.Simulation(nb.Params(num_points=n, morton_bits=21, theta=theta, dt=dt,
                                 softening_factor=eps, num_steps_validator=16)) as sim:
        sim.set_state(ic["x"], ic["y"], ic["z"], ic["m"], ic["vx"], ic["vy"], ic["vz"])
        sim.step(6)
        h = sim.validation_history()
        st = sim.get_state()
        csv = os.path.join(tmp_path, "hist.csv")
        sim.write_validation_history(csv)
    assert len(h["sys_energy"]) == 6
    # energy of the tree run == float64 energy of the state the library reports; 1e-4 because
    # the GPU sums each particle's potential in float32 (one term, the star, dwarfs the rest)
    # and E = KE + PE cancels to half of either
    e_ref = O.energy_f64(st["x"], st["y"], st["z"], st["vx"], st["vy"], st["vz"], st["m"], eps, 1.0)
    assert np.isclose(h["sys_energy"][-1], e_ref, rtol=1e-4)
    # step-local error in the published band for theta = 0.5 and far above float32 noise
    assert np.all(h["avg_acc_err"] > 1e-5) and np.all(h["avg_acc_err"] < 2e-3)
    # the twin starts from the same state: energies agree to the theta-approximation level
    assert np.allclose(h["sys_energy"], h["sys_energy_ap"], rtol=2e-3)
    lines = open(csv).read().strip().split("\n")
    assert lines[0] == "sys_energy,sys_energy_ap,avg_acc_err" and len(lines) == 7
    row = [float(v) for v in lines[-1].split(",")]
    assert np.isclose(row[0], h["sys_energy"][-1], rtol=1e-7)


def test_validator_twin_equals_oracle_direct_sum_integration():
    """The all-pairs twin is a plain leapfrog over direct sums: reproduce 3 steps on the host."""
    n, eps, dt = 1024, 0.05, 1e-3
    ic = nb.sample_ic(nb.IC_SPHERE, n, 5)
    m = (ic["m"] / n).astype(np.float32)
    with nb.Simulation(nb.Params(num_points=n, morton_bits=21, theta=0.5, dt=dt,
                                 softening_factor=eps, num_steps_validator=8)) as sim:
        sim.set_state(ic["x"], ic["y"], ic["z"], m)
        sim.step(3)
        h = sim.validation_history()
    x, y, z = ic["x"].copy(), ic["y"].copy(), ic["z"].copy()
    v = [np.zeros(n, np.float32) for _ in range(3)]
    a = [np.zeros(n, np.float32) for _ in range(3)]
    for _ in range(3):
        x, y, z, vh = O.leapfrog_pos(dt, 5.0, x, y, z, v[0], v[1], v[2], a[0], a[1], a[2])
        a = O.direct_f32(x, y, z, m, eps, 1.0)
        v = [(np.float32(0.5) * a[k] * np.float32(dt) + vh[k]).astype(np.float32) for k in range(3)]
    e_ref = O.energy_f64(x, y, z, v[0], v[1], v[2], m, eps, 1.0)
    assert np.isclose(h["sys_energy_ap"][-1], e_ref, rtol=1e-4)


def test_energy_drift_1000_steps_32k_disk():
    """BASELINE config 1 on the GPU: 32,768-particle disk, theta = 0.5, eps = 0.01, dt = 1e-3,
    1000 steps.  Bounds read off the reference's plots/disk-energy-32k-eps0.01-dt0.001.png
    (all-pairs ~ -82.39, theta=0.5 within ~0.2 % of it, start-up transient ~0.75 %)."""
    n, steps = 32768, 1000
    ic = nb.sample_ic(nb.IC_DISK, n, 42)
    with nb.Simulation(nb.Params(num_points=n, morton_bits=21, traversal_group_size=32, theta=0.5,
                                 dt=1e-3, softening_factor=0.01,
                                 num_steps_validator=steps)) as sim:
        sim.set_state(ic["x"], ic["y"], ic["z"], ic["m"], ic["vx"], ic["vy"], ic["vz"])
        sim.step(steps)
        h = sim.validation_history()
    e, ea, err = h["sys_energy"], h["sys_energy_ap"], h["avg_acc_err"]
    assert len(e) == steps and np.isfinite(e).all() and np.isfinite(ea).all()
    assert np.abs((e - ea) / ea).max() < 3e-3          # tree run vs all-pairs run (two chaotic runs)
    # record 0 is taken after the first step, i.e. after the start-up transient caused by the
    # spawn doing no force evaluation (first kick with a = 0, as in the reference)
    assert np.abs((e - e[0]) / e[0]).max() < 1e-2       # drift of the tree run
    assert np.abs((ea - ea[0]) / ea[0]).max() < 1e-2    # drift of the all-pairs run
    assert 3e-4 < err.mean() < 8e-4 and err.max() < 2e-3   # published theta = 0.5 band (two-sided)
    # Energy LEVEL.  The published all-pairs curve sits at -82.39; the shipped sampler gives
    # -77.9 at this eps = 0.01 before the first step (see test_disk_initial_energy_analytic_and_
    # published in test_oracle.py: -82.39 belongs to eps <= 1e-3), and the first step, whose
    # kick-drift runs with a = 0 (exec_post_spawn does no force evaluation), lifts it once by
    # ~3 % to -75.5; from there both runs are flat.
    assert abs(ea[0] + 75.47) < 0.3 and abs(ea[100:].mean() + 75.5) < 0.5, (ea[0], ea[100:].mean())


def test_disk_energy_level_at_renderer_softening_matches_published_plot():
    """The level of the reference's energy plot (-82.39, plots/disk-energy-32k-eps0.01-dt0.001.png)
    against the energy of the spawned 32,768-particle disk evaluated on the GPU (all pairs:
    theta = 0 sends every particle to every group) at the renderer's disk softening 1e-3
    (src/renderer.cpp:311): within 1 %."""
    n = 32768
    p = nb.Params(num_points=n, morton_bits=21, theta=0.0, dt=1e-3, softening_factor=1e-3)
    with nb.Simulation(p) as sim:
        sim.spawn_disk()
        ke, pe = sim.tree_energy()
        st = sim.get_state()
    e_ref = O.energy_f64(st["x"], st["y"], st["z"], st["vx"], st["vy"], st["vz"], st["m"], 1e-3, 1.0)
    assert abs((ke + pe) - e_ref) < 2e-4 * abs(e_ref), (ke + pe, e_ref)
    assert abs((ke + pe) + 82.39) / 82.39 < 0.01, ke + pe


@pytest.mark.parametrize("theta,rtol", [(0.0, 5e-5), (0.5, 2e-3)])   # float32 per-body sums
def test_tree_energy_matches_all_pairs_energy(theta, rtol):
    """nbody_tree_energy: kinetic exact, potential through the tree; theta = 0 makes the tree
    sum an all-pairs sum."""
    n = 8192
    ic = nb.sample_ic(nb.IC_DISK, n, 2)
    p = nb.Params(num_points=n, theta=theta, dt=1e-3, softening_factor=0.01, morton_bits=21)
    with nb.Simulation(p) as sim:
        sim.set_state(ic["x"], ic["y"], ic["z"], ic["m"], ic["vx"], ic["vy"], ic["vz"])
        sim.step(2)
        st = sim.get_state()
        ke, pe = sim.tree_energy()
        again = sim.get_state()
    for k in ("x", "vx", "ax"):
        assert np.array_equal(st[k], again[k])             # the state is untouched
    ke_ref = float(np.sum(0.5 * st["m"].astype(np.float64) * (st["vx"].astype(np.float64) ** 2 +
                                                              st["vy"].astype(np.float64) ** 2 +
                                                              st["vz"].astype(np.float64) ** 2)))
    e_ref = O.energy_f64(st["x"], st["y"], st["z"], st["vx"], st["vy"], st["vz"], st["m"], 0.01, 1.0)
    assert abs(ke - ke_ref) <= 1e-12 * abs(ke_ref)
    assert abs((ke + pe) - e_ref) <= rtol * abs(e_ref), (ke + pe, e_ref)


def test_tree_energy_counts_coincident_bodies_like_the_all_pairs_energy():
    """Bodies at the SAME position (clamped into a wall corner, or equal by construction) are
    sources of each other at zero distance: m_j / eps each.  Only a body's own term is left
    out, as the reference leaves out idx == i (src/cuda/validator.cu:101), not everything at
    zero distance."""
    import torch
    n = 4096
    ic = nb.sample_ic(nb.IC_SPHERE, n, 12)
    for k in "xyz":                                   # 1200 bodies in 400 triplets of one position
        ic[k][400:800] = ic[k][:400]
        ic[k][800:1200] = ic[k][:400]
    ic["m"] = (ic["m"] * np.linspace(0.5, 1.5, n)).astype(np.float32)   # unequal masses
    eps = 0.05
    p = nb.Params(num_points=n, theta=0.0, dt=1e-3, softening_factor=eps, morton_bits=21)
    with nb.Simulation(p) as sim:
        sim.set_state(ic["x"], ic["y"], ic["z"], ic["m"])
        ke, pe = sim.tree_energy()
        st = sim.get_state()
    e_ref = O.energy_f64(st["x"], st["y"], st["z"], st["vx"], st["vy"], st["vz"], st["m"], eps, 1.0)
    assert abs((ke + pe) - e_ref) <= 5e-5 * abs(e_ref), (ke + pe, e_ref)
    # and the all-pairs energy on the device agrees as well
    posm = torch.from_numpy(np.stack([st["x"], st["y"], st["z"], st["m"]], 1).astype(np.float32)).cuda()
    v = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(3)]
    out = torch.full((2,), np.nan, dtype=torch.float64, device="cuda")
    rc = nb.load_library().nbody_direct_energy_dev(posm.data_ptr(), v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(),
                                                   n, eps, 1.0, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    assert abs(out.cpu().numpy()[1] - pe) <= 5e-5 * abs(pe)
    # the coincident bodies matter at this tolerance: without their mutual terms the energy is off
    m = st["m"].astype(np.float64)
    same = 0.0
    key = np.stack([st["x"], st["y"], st["z"]], 1)
    _, inv, cnt = np.unique(key, axis=0, return_inverse=True, return_counts=True)
    for g in np.flatnonzero(cnt > 1):
        mg = m[inv.ravel() == g]
        same += (mg.sum() ** 2 - (mg ** 2).sum()) / 2 / eps
    assert same > 2e-4 * abs(e_ref)


@pytest.mark.parametrize("kind,n,eps", [(nb.IC_DISK, 20000, 0.01), (nb.IC_PLUMMER, 4097, 0.02),
                                        (nb.IC_SPHERE, 1, 0.1), (nb.IC_DISK, 255, 1e-3)])
def test_direct_energy_on_device_matches_oracle(kind, n, eps):
    """nbody_direct_energy_dev: kinetic and all-pairs potential energy reduced on the GPU (the
    reference's compute_conserved_quantities + cub::DeviceReduce::Sum) against the oracle's
    float64 energy.  Kinetic is a double sum of float products (1e-12); the per-body potential
    sums are float32 (1e-4 of the total, the star's term dominating the disk's)."""
    import torch
    ic = nb.sample_ic(kind, n, 9, p0=0.1, p1=1.0)
    posm = torch.from_numpy(np.stack([ic["x"], ic["y"], ic["z"], ic["m"]], 1).astype(np.float32)).cuda()
    v = [torch.from_numpy(ic[k]).cuda() for k in ("vx", "vy", "vz")]
    out = torch.full((2,), np.nan, dtype=torch.float64, device="cuda")
    lib = nb.load_library()
    for _ in range(2):     # twice: the scratch counter is left ready for the next call
        rc = lib.nbody_direct_energy_dev(posm.data_ptr(), v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(),
                                         n, eps, 1.0, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        ke, pe = out.cpu().numpy()
        m = ic["m"].astype(np.float64)
        ke_ref = float(np.sum(0.5 * m * (ic["vx"].astype(np.float64) ** 2 + ic["vy"].astype(np.float64) ** 2 +
                                         ic["vz"].astype(np.float64) ** 2)))
        e_ref = O.energy_f64(ic["x"], ic["y"], ic["z"], ic["vx"], ic["vy"], ic["vz"], ic["m"], eps, 1.0)
        assert abs(ke - ke_ref) <= 1e-12 * abs(ke_ref) + 1e-300
        assert abs((ke + pe) - e_ref) <= 1e-4 * max(abs(e_ref), abs(pe)) + 1e-300, (ke + pe, e_ref)
    assert lib.nbody_direct_energy_dev(posm.data_ptr(), v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(),
                                       0, eps, 1.0, out.data_ptr(), None) == 1
