"""Emulates k_traverse's pop policy on the CPU for sample groups: MAC rounds and lane fill."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import oracle_lib as O
import nbody_amd as nb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2097152
theta = 0.6
ic = nb.sample_ic(nb.IC_SPHERE, n, 42)
b = O.full_build(ic["x"], ic["y"], ic["z"], ic["m"], 5.0, 21, 32)
T, Cm = b["tree"], b["com"]
fc, nc, size = T["first_child"], T["num_children"], T["size"]
pb, pe = T["points_begin"], T["points_end"]
C = np.stack([Cm["comx"], Cm["comy"], Cm["comz"]], 1)
X_all = np.stack([b["xs"], b["ys"], b["zs"]], 1)
th2 = np.float32(theta * theta)
rng = np.random.default_rng(1)
G = 64
res = []
for g in rng.integers(0, n // G, 40):
    X = X_all[g*G:(g+1)*G]
    stack, leaves = [0], []
    rounds, tests, leaf_rounds, runs, hist = 0, 0, 0, 0, []
    def do_leaves():
        global leaf_rounds, runs, leaves
        cnt = min(64, len(leaves))
        ids = leaves[-cnt:]; del leaves[-cnt:]
        leaf_rounds += 1
        prev_end = None
        for o in ids:
            if prev_end is None or pb[o] != prev_end: runs += 1
            prev_end = pe[o] + 1
    while True:
        while len(leaves) >= 64 or (not stack and leaves): do_leaves()
        if not stack: break
        k = min(len(stack), 64, max((512 - 168 - len(stack)) // 7, 1), (640 - len(leaves)) >> 3)
        nodes = np.array(stack[-k:]); del stack[-k:]
        rounds += 1; tests += k; hist.append(k)
        d2 = ((X[None] - C[nodes][:, None]) ** 2).sum(2).min(1)
        acc = size[nodes] ** 2 / d2 < th2
        for o in nodes[~acc]:
            for c in range(fc[o], fc[o] + nc[o]):
                (leaves if nc[c] == 0 else stack).append(c)
    res.append((rounds, tests, leaf_rounds, runs))
    if len(res) == 1: print("k per round:", hist)
r = np.array(res, float).mean(0)
print(dict(rounds=r[0], tests=r[1], leaf_rounds=r[2], runs=r[3]))
