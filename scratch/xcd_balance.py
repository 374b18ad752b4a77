"""Interaction-list lengths per contiguous eighth of the groups (what one XCD walks)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nbody_amd as nb
for name, n, kind, theta, eps, dt in (("disk524k", 524288, "disk", 0.75, 1e-3, 1e-4), ("sphere2M", 1 << 21, "sphere", 0.6, 0.1, 1e-6),
                                      ("plummer4M", 1 << 22, "plummer", 0.6, 1e-3, 1e-6)):
    p = nb.Params(num_points=n, theta=theta, dt=dt, softening_factor=eps, morton_bits=21, traversal_group_size=64, flags=nb.FLAG_TIMING)
    with nb.Simulation(p) as sim:
        getattr(sim, "spawn_" + kind)(*(() if kind != "plummer" else (0.1, 1.0)))
        sim.step(3)
        ll = sim.get_list_len().astype(np.int64)
    g = len(ll)
    q, r = divmod(g, 8)
    bounds = [0]
    for x in range(8):
        bounds.append(bounds[-1] + q + (1 if x < r else 0))
    sums = [int(ll[bounds[x]:bounds[x + 1]].sum()) for x in range(8)]
    print(json.dumps({"config": name, "groups": g, "mean_list": float(ll.mean()), "max_list": int(ll.max()),
                      "per_xcd_share_of_mean": [round(s * 8 / ll.sum(), 3) for s in sums]}))
