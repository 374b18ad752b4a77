import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nbody_amd as nb
from nbody_amd import parallel
which = sys.argv[1]
if which == "full":
    ic = nb.sample_ic(nb.IC_SPHERE, 1 << 21, 42); idx = np.arange(1 << 21)
else:
    ic = nb.sample_ic(nb.IC_SPHERE, 8 << 21, 42)
    idx = parallel.domain_slices(ic["x"], ic["y"], ic["z"], 8)[int(which)]
s = nb.Simulation(nb.Params(num_points=len(idx), theta=0.6, dt=1e-6, softening_factor=0.1, morton_bits=21, flags=nb.FLAG_TIMING))
s.set_state(ic["x"][idx], ic["y"][idx], ic["z"][idx], ic["m"][idx])
s.step(4); s.sync()
t = s.get_timings(); print(which, "trav ms", t["ms_traverse"] / 4)
