"""Timeline of the traversal kernel from the NB_WALK_TRACE dump: concurrency over time."""
import sys, numpy as np
tr = np.fromfile(sys.argv[1], np.uint64).reshape(-1, 3)
b, e, hw = tr[:, 0].astype(np.int64), tr[:, 1].astype(np.int64), tr[:, 2]
t0 = b.min(); b -= t0; e -= t0
T = e.max()
print("groups", len(b), "kernel span %.3f ms" % (T / 1e5), "mean group %.3f ms" % ((e - b).mean() / 1e5),
      "min %.3f max %.3f" % ((e - b).min() / 1e5, (e - b).max() / 1e5))
# concurrency in 20 time bins
edges = np.linspace(0, T, 21)
conc = [(np.minimum(e, hi) - np.maximum(b, lo)).clip(0).sum() / (hi - lo) for lo, hi in zip(edges[:-1], edges[1:])]
print("concurrent waves per 5% of the span:", [int(c) for c in conc])
print("slot-time used: %.1f%% of 5120 x span" % (100 * (e - b).sum() / (5120 * T)))
# hw id: bits wave[3:0] simd[5:4] ... cu[11:8] sh[12] se[15:13]  (gfx9 HW_ID layout)
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
xcc = (hw >> 16) & 0xffff
key = (se.astype(np.int64) * 2 + sh) * 16 + cu
print("distinct (se,sh,cu):", len(np.unique(key)), "se values", np.unique(se).tolist(), "upper bits", np.unique(xcc)[:8].tolist())
# per-group duration vs group index (8 chunks)
n = len(b)
for k in range(8):
    sl = slice(n * k // 8, n * (k + 1) // 8)
    print("chunk %d: start %.2f-%.2f ms  end max %.2f ms  mean dur %.3f ms" % (k, b[sl].min() / 1e5, b[sl].max() / 1e5, e[sl].max() / 1e5, (e[sl] - b[sl]).mean() / 1e5))
d = e - b
top = np.argsort(d)[::-1][:10]
print("longest groups: (group, start ms, dur ms, hw)")
for g in top:
    print("  ", int(g), round(b[g] / 1e5, 3), round(d[g] / 1e5, 3), hex(int(hw[g])))
# duration histogram
print("duration quantiles ms:", [round(float(np.quantile(d, q)) / 1e5, 3) for q in (0.01, 0.1, 0.5, 0.9, 0.99, 0.999, 1.0)])
# per-slot (hw id incl. wave slot) total busy time and count
slot = hw & 0xffffffff
u, inv = np.unique(slot, return_inverse=True)
busy = np.bincount(inv, weights=d.astype(np.float64))
cnt = np.bincount(inv)
print("wave slots seen", len(u), "groups per slot min/mean/max", cnt.min(), cnt.mean(), cnt.max(), "busy ms min/mean/max", busy.min() / 1e5, busy.mean() / 1e5, busy.max() / 1e5)
