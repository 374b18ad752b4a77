"""One GPU plays the ranks of a W-GPU domain-decomposed step one after the other, with the
calls nbody_dd_step makes, and times every phase of every rank (host clock around a
synchronised call; phases last 0.1 ms or more).
    python scratch/dd_emulate.py [W=8] [particles_per_rank=2097152] [sphere|plummer]
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import nbody_amd as nb
import oracle_lib as O

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ppr = int(sys.argv[2]) if len(sys.argv) > 2 else 2097152
kind = sys.argv[3] if len(sys.argv) > 3 else "sphere"
n = W * ppr
if kind == "plummer":
    ic = nb.sample_ic(nb.IC_PLUMMER, n, 42, p0=0.1, p1=1.0); eps = 1e-3
else:
    ic = nb.sample_ic(nb.IC_SPHERE, n, 42); eps = 0.1
keys = O.morton(ic["x"], ic["y"], ic["z"], 5.0, 21)
order = np.argsort(keys, kind="stable")
del keys
T = int(os.environ.get("DD_T", nb.LET_MAX_RECORDS))


def params(m, **kw):
    return nb.Params(num_points=m, theta=0.6, dt=1e-6, softening_factor=eps, morton_bits=21, domain_size=5.0, **kw)


sims = []
for r in range(W):
    idx = np.sort(order[n * r // W:n * (r + 1) // W])
    s = nb.Simulation(params(len(idx)))
    s.set_state(ic["x"][idx], ic["y"][idx], ic["z"][idx], ic["m"][idx])
    s.set_traversal_range(0, 0)
    sims.append(s)
imp = nb.Simulation(params(ppr, max_num_codes_per_leaf=int(os.environ.get("DD_LEAF", "4"))))


def timed(f, sync):
    torch.cuda.synchronize()
    t = time.perf_counter()
    f()
    sync()
    return (time.perf_counter() - t) * 1e3


rows = []
for step in range(3):
    ph = [dict() for _ in range(W)]
    fr = torch.zeros(W, T, nb.LET_REC, dtype=torch.float32, device="cuda")
    for r, s in enumerate(sims):
        ph[r]["build"] = timed(s.update, s.sync)
        ph[r]["frontier"] = timed(lambda: s.let_frontier_dev(fr[r].data_ptr(), T), s.sync)
    cap = ppr // 2
    recs = [None] * W
    for r, s in enumerate(sims):
        out = torch.zeros(W, cap, 4, dtype=torch.float32, device="cuda")
        cnt = torch.zeros(W, dtype=torch.int32, device="cuda")
        ph[r]["export"] = timed(lambda: s.let_export_dev(fr.data_ptr(), W, T, r, out.data_ptr(), cap, cnt.data_ptr()), s.sync)
        c = cnt.cpu().numpy()
        assert c.max() <= cap
        ph[r]["sent"] = int(c.sum())
        recs[r] = [out[q, :c[q]].clone() for q in range(W)]
        del out
    for r, s in enumerate(sims):
        ph[r]["own_traversal"] = timed(lambda: s.traverse_from(s, accumulate=False), s.sync)
        ll0 = s.get_list_len().astype(np.float64)
        rec = torch.cat([recs[q][r] for q in range(W) if q != r]).contiguous()
        ph[r]["received"] = int(rec.shape[0])
        ph[r]["import_build"] = timed(lambda: imp.set_sources_dev(rec.data_ptr(), rec.shape[0]), imp.sync)
        ph[r]["import_traversal"] = timed(lambda: s.traverse_from(imp, accumulate=True), s.sync)
        ll1 = s.get_list_len().astype(np.float64)
        ph[r]["own_list_mean"] = float(ll0.mean()); ph[r]["list_after_import_mean"] = float(ll1.mean()); ph[r]["list_after_import_max"] = float(ll1.max())
        ph[r]["import_nodes"] = int(imp.tree_counts()[2])
        ph[r]["finish"] = timed(s.finish_step, s.sync)
        ph[r]["sum"] = sum(v for k, v in ph[r].items() if k in ("build", "frontier", "export", "own_traversal", "import_build", "import_traversal", "finish"))
    if step:
        rows.append(ph)
last = rows[-1]
for r in range(W):
    print(json.dumps({"rank": r, **{k: (round(v, 3) if isinstance(v, float) else v) for k, v in last[r].items()}}))
print(json.dumps({"world": W, "particles_per_rank": ppr, "ic": kind,
                  "max_over_ranks_sum_ms": round(max(p["sum"] for p in last), 3),
                  "mean_over_ranks_sum_ms": round(sum(p["sum"] for p in last) / W, 3),
                  "records_sent_mean": round(sum(p["sent"] for p in last) / W)}))
for s in sims + [imp]:
    s.close()
