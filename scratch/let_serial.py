"""8 emulated LET domains on one GPU, one rank's work at a time (every call synchronised), so
that the per-rank kernel times are not disturbed by the other ranks' kernels."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nbody_amd as nb
from nbody_amd import parallel
world = int(sys.argv[1]); n_per = int(sys.argv[2]); kind = sys.argv[3] if len(sys.argv) > 3 else "sphere"
n = world * n_per
ic = nb.sample_ic(nb.IC_SPHERE, n, 42) if kind == "sphere" else nb.sample_ic(nb.IC_PLUMMER, n, 42, p0=0.1, p1=1.0)
EPS = 0.1 if kind == "sphere" else 1e-3
parts = parallel.domain_slices(ic["x"], ic["y"], ic["z"], world, grid=(5.0, 21) if os.environ.get("NO_SNAP") != "1" else None)
sims, bes = [None] * world, [None] * world
alloc_order = range(world) if os.environ.get("ALLOC_REV") != "1" else range(world - 1, -1, -1)
for r in alloc_order:
    idx = parts[r]
    s = nb.Simulation(nb.Params(num_points=len(idx), theta=0.6, dt=1e-6, softening_factor=EPS, morton_bits=21))
    s.set_state(ic["x"][idx], ic["y"][idx], ic["z"][idx], ic["m"][idx])
    sims[r] = s; bes[r] = parallel.GpuLetBackend(s, "cuda:0")
for r in range(world):
    p = np.stack([ic[k][parts[r]] for k in "xyz"], 1)
    print("rank", r, "box", np.round(p.min(0), 3), np.round(p.max(0), 3))
def timed(f, s):
    s.sync(); torch.cuda.synchronize(); t = time.perf_counter(); f(); s.sync()
    if bes[0].remote is not None:
        pass
    return (time.perf_counter() - t) * 1e3
for it in range(3):
    rows = []
    for b in bes: b.update()
    for s in sims: s.sync()
    boxes = np.stack([b.bbox() for b in bes])
    recs = []
    for r, b in enumerate(bes):
        others = [q for q in range(world) if q != r]
        recs.append({q: t.clone() for q, t in zip(others, b.let_export(boxes[others]))})
    order = list(range(world)) if len(sys.argv) <= 4 else list(range(world))[::-1]
    rows = [None] * world
    for r in order:
        b = bes[r]
        t_own = timed(b.begin_local_forces, b.sim)
        ll_own = b.sim.get_list_len().astype(np.int64).sum() * 64
        inc = torch.cat([recs[q][r] for q in range(world) if q != r])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if it == 0:
            b.add_remote(inc); b.sim.sync(); t_bld = 0.0
        else:
            b.remote.set_sources_dev(inc.data_ptr(), int(inc.shape[0])); b.remote.sync()
            t_bld = (time.perf_counter() - t0) * 1e3
            b.sim.traverse_from(b.remote, accumulate=True); b.sim.sync()
        t_imp = (time.perf_counter() - t0) * 1e3
        ll_imp = b.sim.get_list_len().astype(np.int64).sum() * 64
        b.finish_step(); b.sim.sync()
        rows[r] = (t_own, ll_own / 1e9, t_imp, ll_imp / 1e9, inc.shape[0], t_bld)
    if it == 2:
        for r, row in enumerate(rows):
            print("rank %d own %.2f ms %.2f G (%.2f G/ms) | import build %.2f + trav %.2f ms %.2f G records %d" % (r, row[0], row[1], row[1] / row[0], row[5], row[2] - row[5], row[3], row[4]))
        a = np.array(rows)
        tot = a[:, 0] + a[:, 2]
        print("own+import ms: max %.2f mean %.2f" % (tot.max(), tot.mean()), flush=True)
