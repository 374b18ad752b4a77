"""Own-tree step time of one octant domain of the 8 x 2,097,152 sphere (what one rank of an 8-GPU run
holds after the Morton-range decomposition), for one library build."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nbody_amd as nb
n = 8 * 2097152
ic = nb.sample_ic(nb.IC_SPHERE, n, 42)
for name, sel in (("low", (ic["x"] < 2.5) & (ic["y"] < 2.5) & (ic["z"] < 2.5)), ("high", (ic["x"] >= 2.5) & (ic["y"] >= 2.5) & (ic["z"] >= 2.5))):
    sub = {k: np.ascontiguousarray(v[sel]) for k, v in ic.items()}
    m = len(sub["x"])
    p = nb.Params(num_points=m, morton_bits=21, theta=0.6, dt=1e-6, softening_factor=0.1, flags=nb.FLAG_TIMING | nb.FLAG_TIMING_STAGES)
    with nb.Simulation(p) as s:
        s.set_state(sub["x"], sub["y"], sub["z"], sub["m"], sub["vx"], sub["vy"], sub["vz"])
        s.step(5); s.get_timings()
        s.step(20); t = s.get_timings()
        ll = s.get_list_len().astype(np.float64)
    print(json.dumps({"lib": os.path.basename(nb.LIB_PATH), "octant": name, "n": m, "traverse": round(t["ms_traverse"] / 20, 4),
                      "total": round(t["ms_total"] / 20, 4), "ll_mean": round(ll.mean(), 1), "ll_max": ll.max(), "frac>2x": float((ll > 2 * ll.mean()).mean())}))
