import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nbody_amd as nb
n = 1 << 21
with nb.Simulation(nb.Params(num_points=n, morton_bits=21, flags=nb.FLAG_NO_COHERENT_SORT, seed=42, theta=0.6, dt=1e-6, softening_factor=0.1)) as s:
    ic = nb.sample_ic(nb.IC_SPHERE, n, 42)
    s.set_state(ic["x"], ic["y"], ic["z"], ic["m"])
    prev = s.get_octree()["codes_sorted"].copy()
    for k in range(6):
        s.step(1)
        o = s.get_octree()
        c, p = o["codes_sorted"], o["perm"]
        st = s.get_state()
        print(k, "keys changed", int((c != prev).sum()), "perm != identity", int((p != np.arange(n)).sum()),
              "max |v|", float(np.abs(st["vx"]).max()), "distinct keys", len(np.unique(c)))
        # stay criterion on the host
        if k > 0:
            pass
        prev = c.copy()
