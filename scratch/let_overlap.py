"""Would the import-tree traversal hide behind the own-tree traversal?  One emulated rank:
own traversal on the rank's handle, import traversal of the same bodies on a clone handle (its
own stream) against the import tree, one after the other and at the same time."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nbody_amd as nb
from nbody_amd import parallel
world, n_per = 8, 2097152
ic = nb.sample_ic(nb.IC_SPHERE, world * n_per, 42)
parts = parallel.domain_slices(ic["x"], ic["y"], ic["z"], world)
def mk(idx):
    s = nb.Simulation(nb.Params(num_points=len(idx), theta=0.6, dt=1e-6, softening_factor=0.1, morton_bits=21))
    s.set_state(ic["x"][idx], ic["y"][idx], ic["z"][idx], ic["m"][idx])
    return s
sims = [mk(idx) for idx in parts]
bes = [parallel.GpuLetBackend(s, "cuda:0") for s in sims]
for b in bes: b.update()
boxes = np.stack([b.bbox() for b in bes])
for r in [int(a) for a in sys.argv[1:]] or [0, 4]:
    others = [q for q in range(world) if q != r]
    inc = torch.cat([bes[q].let_export(boxes[[r]])[0].clone() for q in others])
    own = sims[r]
    clone = mk(parts[r]); clone.set_traversal_range(0, 0); clone.update(); clone.sync()
    remote = nb.Simulation(nb.Params(num_points=int(inc.shape[0]) + 1024, theta=0.6, dt=1e-6, softening_factor=0.1,
                                     morton_bits=21, max_num_codes_per_leaf=8))
    remote.set_sources_dev(inc.data_ptr(), int(inc.shape[0])); remote.sync()
    def sync():
        own.sync(); clone.sync(); torch.cuda.synchronize()
    res = {}
    for rep in range(3):
        sync(); t = time.perf_counter(); own.traverse_from(own, accumulate=False); own.sync(); res["own"] = (time.perf_counter() - t) * 1e3
        sync(); t = time.perf_counter(); clone.traverse_from(remote, accumulate=False); clone.sync(); res["import"] = (time.perf_counter() - t) * 1e3
        sync(); t = time.perf_counter()
        own.traverse_from(own, accumulate=False); clone.traverse_from(remote, accumulate=False)
        sync(); res["both at once"] = (time.perf_counter() - t) * 1e3
    print("rank", r, {k: round(v, 2) for k, v in res.items()}, "serial sum %.2f" % (res["own"] + res["import"]), flush=True)
