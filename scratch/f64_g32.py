import sys, time, numpy as np
sys.path.insert(0, '.')
import nbody_amd as nb
n = 1 << 21
ic = nb.sample_ic(nb.IC_SPHERE, n, 42)
for g in (64, 32):
    with nb.Simulation(nb.Params(num_points=n, morton_bits=21, theta=0.6, dt=1e-6, softening_factor=0.1, traversal_group_size=g, flags=nb.FLAG_FP64 | nb.FLAG_TIMING)) as sim:
        sim.set_state(ic["x"], ic["y"], ic["z"], ic["m"])
        sim.step(2); sim.get_timings()
        sim.step(10); t = sim.get_timings()
        print("G", g, "traverse ms", t["ms_traverse"] / 10, "total", t["ms_total"] / 10, "inter/step %.3e" % (t["interactions"] / 10))
