"""8 emulated LET domains, one rank at a time: wall time of the force phase of a step
(begin_local_forces .. finish_step done) with the import traversal on the side stream and on
the main stream (NBODY_FLAG_NO_SIDE_STREAM)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nbody_amd as nb
from nbody_amd import parallel
world, n_per = 8, 2097152
kind = sys.argv[1] if len(sys.argv) > 1 else "sphere"
ic = nb.sample_ic(nb.IC_SPHERE, world * n_per, 42) if kind == "sphere" else nb.sample_ic(nb.IC_PLUMMER, world * n_per, 42, p0=0.1, p1=1.0)
EPS = 0.1 if kind == "sphere" else 1e-3
parts = parallel.domain_slices(ic["x"], ic["y"], ic["z"], world, grid=(5.0, 21) if os.environ.get("NO_SNAP") != "1" else None)
for flags, label in ((0, "side stream"), (nb.FLAG_NO_SIDE_STREAM, "one stream")):
    sims, bes = [], []
    for idx in parts:
        s = nb.Simulation(nb.Params(num_points=len(idx), theta=0.6, dt=1e-6, softening_factor=EPS, morton_bits=21, flags=flags))
        s.set_state(ic["x"][idx], ic["y"][idx], ic["z"][idx], ic["m"][idx])
        sims.append(s); bes.append(parallel.GpuLetBackend(s, "cuda:0"))
    for it in range(4):
        for b in bes: b.update()
        for s in sims: s.sync()
        boxes = np.stack([b.bbox() for b in bes])
        recs = []
        for r, b in enumerate(bes):
            others = [q for q in range(world) if q != r]
            recs.append({q: t.clone() for q, t in zip(others, b.let_export(boxes[others]))})
        times = []
        for r, b in enumerate(bes):
            inc = torch.cat([recs[q][r] for q in range(world) if q != r])
            b.sim.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
            b.begin_local_forces()
            b.add_remote(inc)
            b.finish_step()
            b.sim.sync()
            times.append((time.perf_counter() - t0) * 1e3)
    print(label, "force phase ms per rank", np.round(times, 2).tolist(), "max %.2f mean %.2f" % (max(times), np.mean(times)), flush=True)
    for s in sims: s.close()
