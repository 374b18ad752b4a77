"""Work per SIMD of one traversal launch: list length (sources per group) summed over the groups
each SIMD ran, from the profiling build's per-workgroup trace (NB_WALK_TRACE) and the launch's
list lengths.  python scratch/simd_balance.py [disk|sphere] [n]   (NBODY_AMD_LIB = profiling build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
os.environ.setdefault("NB_WALK_TRACE", "/tmp/simd_balance_trace.bin")
import nbody_amd as nb
kind = sys.argv[1] if len(sys.argv) > 1 else "disk"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 524288
kw = dict(theta=0.75, dt=1e-4, softening_factor=1e-3) if kind == "disk" else dict(theta=0.6, dt=1e-6, softening_factor=0.1)
with nb.Simulation(nb.Params(num_points=n, morton_bits=21, **kw)) as s:
    if kind == "disk":
        s.spawn_disk()
    else:
        ic = nb.sample_ic(nb.IC_SPHERE, n, 42); s.set_state(ic["x"], ic["y"], ic["z"], ic["m"])
    s.step(3); s.sync()
    ll = s.get_octree()["list_len"].astype(np.float64)
tr = np.fromfile(os.environ["NB_WALK_TRACE"], np.uint64).reshape(-1, 3)
ng = len(ll)
per_xcd = (ng + 7) // 8
assert len(tr) >= ng
b, e, hw = tr[:ng, 0].astype(np.int64), tr[:ng, 1].astype(np.int64), tr[:ng, 2]
x = np.arange(ng) // per_xcd
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = (((x * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
u, inv = np.unique(key, return_inverse=True)
work = np.bincount(inv, weights=ll)
cnt = np.bincount(inv)
print(kind, n, "groups", ng, "SIMDs seen", len(u), "groups per SIMD min/mean/max", cnt.min(), round(cnt.mean(), 2), cnt.max())
print("sources per SIMD: mean %.0f  min %.0f  max %.0f  max/mean %.3f  std/mean %.3f" % (work.mean(), work.min(), work.max(), work.max() / work.mean(), work.std() / work.mean()))
cukey = key // 4
u2, inv2 = np.unique(cukey, return_inverse=True)
w2 = np.bincount(inv2, weights=ll)
print("sources per CU:   CUs %d  max/mean %.3f  std/mean %.3f" % (len(u2), w2.max() / w2.mean(), w2.std() / w2.mean()))
T = (e.max() - b.min()) / 1e5
last_end = np.zeros(len(u)); np.maximum.at(last_end, inv, (e - b.min()) / 1e5)
print("kernel span %.3f ms; a SIMD's last wave ends at min %.3f  median %.3f  max %.3f ms" % (T, last_end.min(), np.median(last_end), last_end.max()))
# ideal: the span if every SIMD had the mean work and ran at the busiest SIMD's rate
rate = work / last_end
print("sources per ms and SIMD: median %.0f; span if work were even at that rate: %.3f ms" % (np.median(rate), work.mean() / np.median(rate)))
