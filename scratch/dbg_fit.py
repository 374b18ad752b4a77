import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nbody_amd as nb
for n in (1000, 1025, 4000, 30000):
    ic = nb.sample_ic(nb.IC_DISK, n, 21)
    with nb.Simulation(nb.Params(num_points=n, morton_bits=10, flags=nb.FLAG_FIT_KEYS)) as sim:
        sim.set_state(ic["x"], ic["y"], ic["z"], ic["m"])
        print(n, "after", sim.key_domain(), sim.tree_counts())
        sim.update()
        print(n, "step", sim.key_domain(), sim.tree_counts())
