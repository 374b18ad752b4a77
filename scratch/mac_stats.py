"""CPU statistics of the group walk at bench scale: how many MAC tests, how many could be
decided by a bounding-box / sample-body prefilter."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import oracle_lib as O
import nbody_amd as nb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2097152
theta = 0.6
ic = nb.sample_ic(nb.IC_SPHERE, n, 42)
t0 = time.time()
b = O.full_build(ic["x"], ic["y"], ic["z"], ic["m"], 5.0, 21, 32)
print("build", time.time() - t0)
T, Cm = b["tree"], b["com"]
fc, nc, size = T["first_child"], T["num_children"], T["size"]
pb, pe = T["points_begin"], T["points_end"]
cx, cy, cz = Cm["comx"], Cm["comy"], Cm["comz"]
xs, ys, zs = b["xs"], b["ys"], b["zs"]
th2 = np.float32(theta * theta)
rng = np.random.default_rng(1)
G = 64
tot = dict(tests=0, acc=0, opened=0, leaves=0, leafpts=0, qacc=0, qopen=0, amb=0, qopen4=0)
groups = rng.integers(0, n // G, 40)
for g in groups:
    X = np.stack([xs[g*G:(g+1)*G], ys[g*G:(g+1)*G], zs[g*G:(g+1)*G]], 1).astype(np.float32)
    lo, hi = X.min(0), X.max(0)
    ctr = (lo + hi) / 2
    near = np.argmin(((X - ctr) ** 2).sum(1))
    samp = X[[0, 21, 42, 63]]
    stack = [0]
    while stack:
        nodes = np.array(stack); stack = []
        leafm = nc[nodes] == 0
        for o in nodes[leafm]:
            tot["leaves"] += 1; tot["leafpts"] += pe[o] - pb[o] + 1
        inner = nodes[~leafm]
        if len(inner) == 0: continue
        C = np.stack([cx[inner], cy[inner], cz[inner]], 1)
        d2 = ((X[None, :, :] - C[:, None, :]) ** 2).sum(2)
        dmin = d2.min(1)
        sz2 = size[inner] ** 2
        acc = sz2 / dmin < th2
        # prefilter
        dbox = np.maximum(0, np.maximum(lo[None] - C, C - hi[None]))
        dbox2 = (dbox ** 2).sum(1)
        with np.errstate(divide="ignore"):
            qacc = sz2 / dbox2 < th2
        dnear = ((X[near][None] - C) ** 2).sum(1)
        qopen = ~(sz2 / dnear < th2)
        d4 = ((samp[None] - C[:, None]) ** 2).sum(2).min(1)
        qopen4 = ~(sz2 / d4 < th2)
        assert not (qacc & ~acc).any() and not (qopen & acc).any()
        tot["tests"] += len(inner); tot["acc"] += acc.sum(); tot["opened"] += (~acc).sum()
        tot["qacc"] += qacc.sum(); tot["qopen"] += qopen.sum(); tot["qopen4"] += qopen4.sum()
        tot["amb"] += (~qacc & ~qopen).sum()
        for o in inner[~acc]:
            stack.extend(range(fc[o], fc[o] + nc[o]))
k = len(groups)
print({a: round(v / k, 1) for a, v in tot.items()})
