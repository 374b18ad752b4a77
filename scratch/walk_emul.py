"""CPU emulation of k_traverse's node-stack discipline (pop / defer / push-back / depth-first)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import oracle_lib as O
import nbody_amd as nb
kind, n, bits, theta, limit = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5])
RESERVE, LEAF, PEND, G = 168, 544, 96, 64
ic = nb.sample_ic(kind, n, 17, p0=0.1, p1=1.0)
b = O.full_build(ic["x"], ic["y"], ic["z"], ic["m"], 5.0, bits, 32)
T, Cm = b["tree"], b["com"]
fc, nc, size = T["first_child"], T["num_children"], T["size"]
C = np.stack([Cm["comx"], Cm["comy"], Cm["comz"]], 1)
X_all = np.stack([b["xs"], b["ys"], b["zs"]], 1)
th2 = np.float32(theta * theta)
maxsp = 0; maxrounds = 0
ng = (n + G - 1) // G
for g in range(0, ng, int(sys.argv[6]) if len(sys.argv) > 6 else 1):
    X = X_all[g*G:(g+1)*G]
    lo, hi = X.min(0), X.max(0)
    samp = X[[0, 1, 20, 21, 42, 43, len(X)-2 if len(X) > 1 else 0, len(X)-1]] if len(X) == 64 else X
    stack, pend, leaf_n, rounds, acc_n, leaf_tot = ([] if nc[0] == 0 else [0]), [], 0, 0, 0, 0
    def classify(nodes):
        c = C[nodes]
        d2 = ((X[None] - c[:, None]) ** 2).sum(2).min(1)
        sz2 = size[nodes] ** 2
        with np.errstate(divide="ignore", invalid="ignore"):
            ex = sz2 / d2 < th2
            dbox = (np.maximum(0, np.maximum(lo[None] - c, c - hi[None])) ** 2).sum(1)
            qa = sz2 / dbox < th2
            ds = ((samp[None] - c[:, None]) ** 2).sum(2).min(1)
            so = ~(sz2 / ds < th2)
        return ex, qa, so
    while True:
        while leaf_n >= 64 or (not stack and not pend and leaf_n > 0):
            leaf_n -= min(64, leaf_n)
        if not stack and not pend: break
        rounds += 1
        if rounds > 200000: print("LIVELOCK group", g, len(stack), len(pend), leaf_n); sys.exit(1)
        sp = len(stack)
        room = limit - RESERVE - sp
        leaf_room = LEAF - leaf_n
        kx = min(min(len(pend), 32), min(max(room, 0) >> 3, leaf_room >> 3)) if (len(pend) >= 32 or sp == 0) else 0
        if kx > 0:
            nodes = np.array(pend[-kx:]); del pend[-kx:]
            ex, _, _ = classify(nodes)
            approx = ex; opened = ~ex; may_defer = False
        else:
            may_defer = room >= 7
            k = min(sp, 64) if may_defer else 1
            assert k >= 1, (sp, len(pend), room)
            nodes = np.array(stack[-k:]); del stack[-k:]
            ex, qa, so = classify(nodes)
            if len(pend) + k <= PEND:
                assert not (qa & ~ex).any() and not (so & ex).any()
                und = ~qa & ~so
                pend.extend(nodes[und].tolist())
                approx = qa; opened = so & ~qa
            else:
                approx = ex; opened = ~ex
        acc_n += approx.sum()
        inner = np.array([sum(1 for c in range(fc[o], fc[o] + nc[o]) if nc[c] > 0) if op else 0 for o, op in zip(nodes, opened)])
        leafc = np.array([sum(1 for c in range(fc[o], fc[o] + nc[o]) if nc[c] == 0) if op else 0 for o, op in zip(nodes, opened)])
        inc_i, inc_l = np.cumsum(inner), np.cumsum(leafc)
        fits = opened & ((not may_defer) | ((inc_i <= room + 1) & (inc_l <= leaf_room)))
        back = opened & ~fits
        stack.extend(nodes[back].tolist())
        for o in nodes[fits]:
            for c in range(fc[o], fc[o] + nc[o]):
                if nc[c] == 0: leaf_n += 1; leaf_tot += 1
                else: stack.append(c)
        assert len(stack) <= 512 and leaf_n <= LEAF, (len(stack), leaf_n)
        if may_defer or kx > 0: assert len(stack) <= limit - RESERVE, (len(stack), may_defer, kx)
        maxsp = max(maxsp, len(stack))
    maxrounds = max(maxrounds, rounds)
print("ok: max stack", maxsp, "max rounds/group", maxrounds)
