import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nbody_amd as nb
n = 524288
ic = nb.sample_ic(nb.IC_DISK, n, 42)
p = nb.Params(num_points=n, morton_bits=21, theta=0.75, dt=1e-4, softening_factor=1e-3)
with nb.Simulation(p) as s:
    s.set_state(ic["x"], ic["y"], ic["z"], ic["m"], ic["vx"], ic["vy"], ic["vz"])
    s.step(3)
    ll = s.get_list_len().astype(np.float64)
m = ll.mean()
print("groups", ll.size, "mean", m, "max", ll.max(), "frac>2x", (ll > 2 * m).mean(), "frac>1.5x", (ll > 1.5 * m).mean(), "frac>3x", (ll > 3 * m).mean())
h = np.nonzero(ll > 2 * m)[0]
print("heavy idx quantiles", np.quantile(h, [0, .1, .5, .9, 1]) if h.size else None)
print("per-XCD-chunk heavy counts", [int(((h >= k * ll.size // 8) & (h < (k + 1) * ll.size // 8)).sum()) for k in range(8)])
print("sum heavy / sum all", ll[h].sum() / ll.sum())
