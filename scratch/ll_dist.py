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
s = nb.Simulation(nb.Params(num_points=len(idx), theta=0.6, dt=1e-6, softening_factor=0.1, morton_bits=21))
s.set_state(ic["x"][idx], ic["y"][idx], ic["z"][idx], ic["m"][idx])
s.step(2)
ll = s.get_list_len().astype(np.int64)
st = s.get_state()
print(which, "mean", ll.mean(), "median", np.median(ll), "p99", np.quantile(ll, 0.99), "p99.9", np.quantile(ll, 0.999), "max", ll.max())
top = np.argsort(ll)[::-1][:12]
for g in top:
    sl = slice(g * 64, g * 64 + 64)
    ext = [float(st[k][sl].max() - st[k][sl].min()) for k in "xyz"]
    print("  group", int(g), "of", len(ll), "len", int(ll[g]), "extent", np.round(ext, 3))
print("sum of top 32 / total: %.4f" % (np.sort(ll)[::-1][:32].sum() / ll.sum()))
g = int(top[0])
sl = slice(g * 64, g * 64 + 64)
print("z values", np.unique(st["z"][sl])[:5], "x", np.sort(st["x"][sl])[:4], np.sort(st["x"][sl])[-4:], "y", np.sort(st["y"][sl])[:3], np.sort(st["y"][sl])[-3:])
oc = s.get_octree()
print("codes", [hex(int(c)) for c in oc["codes_sorted"][sl][[0, 1, 32, 63]]])
zc = st["z"]
print("particles with z >= 2.4999:", int((zc >= 2.4999).sum()), "max z", zc.max(), "count z==max", int((zc == zc.max()).sum()))
print("low quantiles / mean:", [round(float(np.quantile(ll, q) / ll.mean()), 3) for q in (0.01, 0.05, 0.1, 0.15, 0.2, 0.3)])
