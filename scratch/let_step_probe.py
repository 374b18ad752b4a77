import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nbody_amd as nb
from nbody_amd import parallel
world, n_per = 8, 2097152
ic = nb.sample_ic(nb.IC_SPHERE, world * n_per, 42)
parts = parallel.domain_slices(ic["x"], ic["y"], ic["z"], world)
r = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for flags, label in ((0, "side stream"), (nb.FLAG_NO_SIDE_STREAM, "one stream")):
    sims, bes = [], []
    for idx in parts:
        s = nb.Simulation(nb.Params(num_points=len(idx), theta=0.6, dt=1e-6, softening_factor=0.1, morton_bits=21, flags=flags))
        s.set_state(ic["x"][idx], ic["y"][idx], ic["z"][idx], ic["m"][idx])
        sims.append(s); bes.append(parallel.GpuLetBackend(s, "cuda:0"))
    for it in range(4):
        for b in bes: b.update()
        for s in sims: s.sync()
        boxes = np.stack([b.bbox() for b in bes])
        others = [q for q in range(world) if q != r]
        inc = torch.cat([bes[q].let_export(boxes[[r]])[0].clone() for q in others])
        for q in others: bes[q].finish_step()
        b = bes[r]
        if it == 0:
            b.begin_local_forces(); b.add_remote(inc); b.finish_step(); b.sim.sync(); continue
        b.sim.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
        b.begin_local_forces(); t1 = time.perf_counter()
        b.remote.set_sources_dev(inc.data_ptr(), int(inc.shape[0]))
        if os.environ.get("SYNC_BUILD") == "1": b.remote.sync()
        t2 = time.perf_counter()
        b.sim.traverse_from(b.remote, accumulate=True); t3 = time.perf_counter()
        b.finish_step(); t4 = time.perf_counter()
        b.sim.sync(); t5 = time.perf_counter()
        print(label, "it", it, "launch own %.2f | remote build (host waits) %.2f | launch import %.2f | finish call %.2f | drain %.2f | total %.2f" %
              tuple(1e3 * x for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5 - t0)), flush=True)
    for s in sims: s.close()
