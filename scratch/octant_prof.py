"""One octant of the headline sphere as a single domain: own-tree traversal time, walk profile
(with the NB_WALK_PROFILE library) and tree statistics, for the octants 0 (low x,y,z) and 7."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nbody_amd as nb
from nbody_amd import parallel
n = 8 << 21
ic = nb.sample_ic(nb.IC_SPHERE, n, 42)
parts = parallel.domain_slices(ic["x"], ic["y"], ic["z"], 8)
which = [int(a) for a in sys.argv[1:]] or [0, 7]
for r in which:
    idx = parts[r]
    s = nb.Simulation(nb.Params(num_points=len(idx), theta=0.6, dt=1e-6, softening_factor=0.1, morton_bits=21, flags=nb.FLAG_TIMING))
    s.set_state(ic["x"][idx], ic["y"][idx], ic["z"][idx], ic["m"][idx])
    s.step(3); s.get_timings()
    s.step(5); t = s.get_timings()
    oc = s.get_octree()
    leaf = oc["num_children"] == 0
    print("octant", r, "trav ms %.3f" % (t["ms_traverse"] / 5), "G/ms %.3f" % (t["interactions"] / t["ms_traverse"] / 1e9),
          "U M n_oct", oc["U"], oc["M"], oc["n_oct"], "leaf mean pts %.2f" % np.mean((oc["points_end"] - oc["points_begin"] + 1)[leaf]),
          "levels of leaves", np.bincount(oc["level"][leaf])[5:12].tolist(), "list_len mean %.0f" % oc["list_len"].mean(), flush=True)
    s.close()
