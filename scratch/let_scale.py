"""Single-GPU emulation of the LET scheme at bench scale: per-rank stage times."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nbody_amd as nb
from nbody_amd import parallel

world = int(sys.argv[1]); n_per = int(sys.argv[2]); bits = int(sys.argv[3]) if len(sys.argv) > 3 else 21
leafc = int(sys.argv[4]) if len(sys.argv) > 4 else 8
kind = sys.argv[5] if len(sys.argv) > 5 else "sphere"
n = world * n_per
t0 = time.time()
ic = nb.sample_ic(nb.IC_SPHERE, n, 42) if kind == "sphere" else nb.sample_ic(nb.IC_PLUMMER, n, 42, p0=0.1, p1=1.0)
EPS = 0.1 if kind == "sphere" else 1e-3
parts = parallel.domain_slices(ic["x"], ic["y"], ic["z"], world)
print(f"ic+orb {time.time()-t0:.1f}s", flush=True)
sims, bes = [], []
for idx in parts:
    s = nb.Simulation(nb.Params(num_points=len(idx), theta=0.6, dt=1e-6, softening_factor=EPS,
                                morton_bits=bits, flags=nb.FLAG_TIMING))
    s.set_state(ic["x"][idx], ic["y"][idx], ic["z"][idx], ic["m"][idx])
    sims.append(s); bes.append(parallel.GpuLetBackend(s, "cuda:0", remote_codes_per_leaf=leafc))
print("created", flush=True)
sync = torch.cuda.synchronize
def tick():
    for s in sims: s.sync()
    sync(); return time.perf_counter()
for it in range(4):
    T = {}
    t = tick()
    for b in bes: b.update()
    T["update"] = (tick() - t) / world
    t = tick()
    boxes = np.stack([b.bbox() for b in bes])
    T["bbox"] = (tick() - t) / world
    recs = []
    t = tick()
    for r, b in enumerate(bes):
        others = [q for q in range(world) if q != r]
        out = b.let_export(boxes[others])
        recs.append({q: t_.clone() for q, t_ in zip(others, out)})
        b.begin_local_forces()
    T["export"] = (tick() - t) / world
    inc = [torch.cat([recs[q][r] for q in range(world) if q != r]) for r in range(world)]
    t = tick()
    for r, b in enumerate(bes):
        b.add_remote(inc[r])
    T["import"] = (tick() - t) / world
    t = tick()
    for b in bes: b.finish_step()
    T["kick"] = (tick() - t) / world
    sent = [sum(int(v.shape[0]) for v in recs[r].values()) for r in range(world)]
    tm = [s.get_timings() for s in sims]
    if it == 3:
        print("per-rank interactions (1e9):", [round(t_["interactions"] / 1e9, 2) for t_ in tm], flush=True)
    print(it, {k: round(v * 1e3, 3) for k, v in T.items()}, "sum_ms", round(sum(T.values()) * 1e3, 3),
          "sent/rank", int(np.mean(sent)), "max", max(sent),
          "own_trav_ms", round(np.mean([t_["ms_traverse"] for t_ in tm]), 3),
          "own_inter", int(np.mean([t_["interactions"] for t_ in tm])), flush=True)
