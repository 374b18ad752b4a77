"""One GPU plays the ranks of a W-GPU domain-decomposed step one after the other, with the
calls nbody_dd_step makes, and times every phase of every rank (host clock around a
synchronised call; phases last 0.1 ms or more).
    python scratch/dd_emulate.py [W=8] [particles_per_rank=2097152] [sphere|plummer] [early_permille=0]
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
early_permille = int(sys.argv[4]) if len(sys.argv) > 4 else 0
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
    if os.environ.get("DD_NO_UNIFORM"):   # the general kernel: every pair multiplied by its mass
        kw["flags"] = kw.get("flags", 0) | nb.FLAG_NO_UNIFORM_MASS
    return nb.Params(num_points=m, theta=0.6, dt=1e-6, softening_factor=eps, morton_bits=21, domain_size=5.0, **kw)


sims = []
for r in range(W):
    idx = np.sort(order[n * r // W:n * (r + 1) // W])
    s = nb.Simulation(params(len(idx)))
    s.set_state(ic["x"][idx], ic["y"][idx], ic["z"][idx], ic["m"][idx])
    s.set_traversal_range(0, 0)
    sims.append(s)


def timed(f, sync):
    torch.cuda.synchronize()
    t = time.perf_counter()
    f()
    sync()
    return (time.perf_counter() - t) * 1e3


rows = []
cap = ppr * 6          # bytes per destination
for step in range(3):
    ph = [dict() for _ in range(W)]
    fr = torch.zeros(W, T, nb.LET_REC, dtype=torch.float32, device="cuda")
    for r, s in enumerate(sims):
        ph[r]["build"] = timed(s.update, s.sync)
        ph[r]["frontier"] = timed(lambda: s.let_frontier_dev(fr[r].data_ptr(), T), s.sync)
    copies = [None] * W
    for r, s in enumerate(sims):
        out = torch.zeros(W, cap, dtype=torch.uint8, device="cuda")
        cnt = torch.zeros(2 * W, dtype=torch.int32, device="cuda")
        ph[r]["export"] = timed(lambda: s.let_export_tree_dev(fr.data_ptr(), W, T, r, out.data_ptr(), cap, cnt.data_ptr()), s.sync)
        c = cnt.cpu().numpy().reshape(W, 2)
        assert (32 * c[:, 0] + 16 * c[:, 1]).max() <= cap
        ph[r]["nodes_sent"] = int(c[:, 0].sum()); ph[r]["records_sent"] = int(c[:, 1].sum())
        ph[r]["bytes_sent"] = int((32 * c[:, 0] + 16 * c[:, 1]).sum())
        copies[r] = [(out[q, :32 * c[q, 0] + 16 * c[q, 1]].clone(), int(c[q, 0]), int(c[q, 1])) for q in range(W)]
        del out
    for r, s in enumerate(sims):
        ng = (s.count() + 63) // 64
        early = ng * early_permille // 1000
        ph[r]["early_own_traversal"] = timed(lambda: s.traverse_forest(0, 0, early), s.sync) if early else 0.0
        inc = [copies[q][r] for q in range(W) if q != r]
        off, nn, nr, at = [], [], [], 0
        for raw, a, b in inc:
            off.append(at); nn.append(a); nr.append(b); at += raw.numel()
        recv = torch.cat([raw for raw, _, _ in inc]).contiguous()
        ph[r]["nodes_received"] = sum(nn); ph[r]["records_received"] = sum(nr)
        ph[r]["graft"] = timed(lambda: s.import_trees_dev(recv.data_ptr(), off, nn, nr), s.sync)
        ph[r]["forest_traversal"] = timed(lambda: s.traverse_forest(1, early, ng), s.sync)
        ll = s.get_list_len().astype(np.float64)[early:]
        ph[r]["late_copies_traversal"] = timed(lambda: s.traverse_forest(2, 0, early), s.sync) if early else 0.0
        ph[r]["list_mean"] = float(ll.mean()); ph[r]["list_max"] = float(ll.max())
        ph[r]["finish"] = timed(s.finish_step, s.sync)
        ph[r]["sum"] = sum(v for k, v in ph[r].items() if k in ("build", "frontier", "export", "early_own_traversal", "graft",
                                                                "forest_traversal", "late_copies_traversal", "finish"))
    if step:
        rows.append(ph)
last = rows[-1]
for r in range(W):
    print(json.dumps({"rank": r, **{k: (round(v, 3) if isinstance(v, float) else v) for k, v in last[r].items()}}))
print(json.dumps({"world": W, "particles_per_rank": ppr, "ic": kind, "early_permille": early_permille,
                  "max_over_ranks_sum_ms": round(max(p["sum"] for p in last), 3),
                  "mean_over_ranks_sum_ms": round(sum(p["sum"] for p in last) / W, 3),
                  "bytes_sent_mean": round(sum(p["bytes_sent"] for p in last) / W)}))
for s in sims:
    s.close()
