"""Single-GPU emulation of `world` LET domains: per-rank interaction counts with equal-count
cuts and after one work-weighted re-draw of the cuts (LetStepper.rebalance's rule)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nbody_amd as nb
from nbody_amd import parallel

world = int(sys.argv[1]); n_per = int(sys.argv[2]); kind = sys.argv[3] if len(sys.argv) > 3 else "plummer"
n = world * n_per
ic = nb.sample_ic(nb.IC_SPHERE, n, 42) if kind == "sphere" else nb.sample_ic(nb.IC_PLUMMER, n, 42, p0=0.1, p1=1.0)
EPS = 0.1 if kind == "sphere" else 1e-3
pos = np.stack([ic["x"], ic["y"], ic["z"]], 1)

def run(parts, label):
    sims, bes = [], []
    for idx in parts:
        s = nb.Simulation(nb.Params(num_points=len(idx), theta=0.6, dt=1e-6, softening_factor=EPS,
                                    morton_bits=21, flags=nb.FLAG_TIMING))
        s.set_particles(ic["x"][idx], ic["y"][idx], ic["z"][idx], ic["m"][idx], id=idx.astype(np.uint32))
        sims.append(s); bes.append(parallel.GpuLetBackend(s, "cuda:0"))
    for it in range(2):
        for b in bes: b.update()
        boxes = np.stack([b.bbox() for b in bes])
        recs = []
        for r, b in enumerate(bes):
            others = [q for q in range(world) if q != r]
            recs.append({q: t.clone() for q, t in zip(others, b.let_export(boxes[others]))})
            b.begin_local_forces()
        for r, b in enumerate(bes):
            b.add_remote(torch.cat([recs[q][r] for q in range(world) if q != r]))
        for b in bes: b.finish_step()
        tm = [s.get_timings() for s in sims]
    inter = np.array([t["interactions"] for t in tm], np.float64)
    ms = np.array([t["ms_traverse"] for t in tm])
    print(label, "particles", [len(p) for p in parts])
    print(label, "interactions 1e9", np.round(inter / 1e9, 2).tolist(), "max/mean %.3f" % (inter.max() / inter.mean()))
    print(label, "traversal ms", np.round(ms, 2).tolist(), "max/mean %.3f" % (ms.max() / ms.mean()), flush=True)
    # what LetStepper.rebalance gathers: a weighted sample from every rank
    samp = []
    for r, b in enumerate(bes):
        w = b.work_weights()
        st = b.get_particles()
        k = len(st["x"])
        pick = np.random.default_rng(r).choice(k, 4096, replace=False)
        samp.append(np.column_stack([st["x"][pick], st["y"][pick], st["z"][pick], w[pick] * (k / 4096)]))
    for s in sims: s.close()
    return np.concatenate(samp)

parts = parallel.domain_slices(ic["x"], ic["y"], ic["z"], world)
samp = run(parts, "equal-count")
cuts = parallel.orb_cuts(samp[:, :3], world, samp[:, 3])
dom = parallel.orb_assign(cuts, pos)
parts2 = [np.nonzero(dom == r)[0] for r in range(world)]
samp = run(parts2, "work-weighted")
cuts = parallel.orb_cuts(samp[:, :3], world, samp[:, 3])
dom = parallel.orb_assign(cuts, pos)
run([np.nonzero(dom == r)[0] for r in range(world)], "work-weighted-2")
