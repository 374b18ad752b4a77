import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, nbody_amd as nb, oracle_lib as O
from test_gpu_fp64 import make_ic64, new_sim64
kind, n, mc, bits = "disk", 32768, 32, 10
ic = make_ic64(kind, n)
ref = O.full_build_dp(ic["x"], ic["y"], ic["z"], ic["m"], 5.0, bits, mc)
t = ref["tree"]; c = ref["com"]
for rep in range(1):
    with new_sim64(n, morton_bits=bits, max_num_codes_per_leaf=mc) as sim:
        sim.set_state(ic["x"], ic["y"], ic["z"], ic["m"], ic["vx"], ic["vy"], ic["vz"])
        gpu = sim.get_octree()
    bad = np.nonzero((gpu["comx"] != c["comx"]) | (gpu["mass"] != c["mass"]))[0]
    cnt = gpu["pe"] - gpu["pb"] + 1 if "pe" in gpu else None
    print("rep", rep, "n_oct", t["n_oct"], "bad", bad.size, bad[:20])
    for o in bad[:0]:
        print(o, "nc", t["num_children"][o], "parent", gpu["parent"][o], "gpu", gpu["comx"][o], gpu["mass"][o], "ref", c["comx"][o], c["mass"][o],
              {k: gpu[k][o] for k in gpu if hasattr(gpu[k], "shape") and gpu[k].shape[:1] == (t["n_oct"],) and k not in ("comx","comy","comz","mass")})
badset = set(bad.tolist())
cnt = gpu["points_end"] - gpu["points_begin"] + 1
for o in bad:
    fc, nc = t["first_child"][o], t["num_children"][o]
    ch = list(range(fc, fc + nc)) if nc else []
    if not any(c_ in badset for c_ in ch):
        print("lowest bad", o, "cnt", cnt[o], "nc", nc, "children", [(c_, int(cnt[c_]), int(t["num_children"][c_])) for c_ in ch], "gpu", gpu["comx"][o], gpu["mass"][o], "ref", c["comx"][o], c["mass"][o])
print("----")
par = gpu["parent"]
seen = set()
for o in bad:
    if t["num_children"][o] == 0:
        p = par[o]
        if p in seen: continue
        seen.add(p)
        fc, nc = t["first_child"][p], t["num_children"][p]
        print("parent", p, "cnt", cnt[p], "bad?", p in badset, "gp", par[p], "gpcnt", cnt[par[p]] if par[p] >= 0 else None,
              "children", [(c_, int(cnt[c_]), int(t["num_children"][c_]), c_ in badset) for c_ in range(fc, fc + nc)])
