"""One LET rank as a real process would hold it (own handle, import handle, side stream): wall
time of the force phase with the imported records of the 8-domain sphere."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nbody_amd as nb
from nbody_amd import parallel
world, n_per = 8, 2097152
ic = nb.sample_ic(nb.IC_SPHERE, world * n_per, 42)
parts = parallel.domain_slices(ic["x"], ic["y"], ic["z"], world)
r = int(sys.argv[1]) if len(sys.argv) > 1 else 0
def mk(idx, flags=0):
    s = nb.Simulation(nb.Params(num_points=len(idx), theta=0.6, dt=1e-6, softening_factor=0.1, morton_bits=21, flags=flags))
    s.set_state(ic["x"][idx], ic["y"][idx], ic["z"][idx], ic["m"][idx])
    return s
me = mk(parts[r]); box = me.get_bbox(); me.close()
incs = []
for q in range(world):
    if q == r: continue
    s = mk(parts[q]); b = parallel.GpuLetBackend(s, "cuda:0"); b.update()
    incs.append(b.let_export(box[None])[0].clone()); s.close()
inc = torch.cat(incs)
torch.cuda.synchronize()
for flags, label in ((0, "two launches"), (0, "one launch")):
    s = mk(parts[r], flags); b = parallel.GpuLetBackend(s, "cuda:0")
    for it in range(5):
        b.update(); s.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
        if label == "one launch" and it > 0:
            b.remote.set_sources_dev(inc.data_ptr(), int(inc.shape[0]))
            s.traverse_pair(b.remote)
            b.finish_step(); s.sync()
        else:
            b.begin_local_forces(); b.add_remote(inc); b.finish_step(); s.sync()
        t = (time.perf_counter() - t0) * 1e3
        if it >= 2: print(label, "rank", r, "force phase %.2f ms" % t, flush=True)
    s.close(); b.remote.close()
