import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbody_amd as nb
n = 1 << 21
ic = nb.sample_ic(nb.IC_SPHERE, n, 42)
for g in (64, 32):
    sim = nb.Simulation(nb.Params(num_points=n, theta=0.6, dt=1e-6, softening_factor=0.1, morton_bits=21,
                                  traversal_group_size=g, flags=nb.FLAG_TIMING))
    sim.set_state(ic["x"], ic["y"], ic["z"], ic["m"])
    sim.step(3); sim.get_timings()
    sim.sync(); t0 = time.perf_counter(); sim.step(20); sim.sync(); t1 = time.perf_counter()
    t = sim.get_timings()
    print(g, round((t1 - t0) / 20 * 1e3, 3), "ms/step trav", round(t["ms_traverse"] / 20, 3), "inter/step %.3e" % (t["interactions"] / 20))
    sim.close()
