"""Per-stage times (ms) of a step for one library build: python scratch/stage_times.py N bits [kind]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nbody_amd as nb
n, bits = int(sys.argv[1]), int(sys.argv[2])
kind = sys.argv[3] if len(sys.argv) > 3 else "sphere"
fit = nb.FLAG_FIT_KEYS if len(sys.argv) > 4 and sys.argv[4] == "fit" else 0
if len(sys.argv) > 4 and sys.argv[4] == "nouni":
    fit = nb.FLAG_NO_UNIFORM_MASS
if kind == "disk":
    ic = nb.sample_ic(nb.IC_DISK, n, 42); kw = dict(theta=0.75, dt=1e-4, softening_factor=1e-3)
else:
    ic = nb.sample_ic(nb.IC_SPHERE, n, 42); kw = dict(theta=0.6, dt=1e-6, softening_factor=0.1)
p = nb.Params(num_points=n, morton_bits=bits, flags=nb.FLAG_TIMING | nb.FLAG_TIMING_STAGES | fit, **kw)
with nb.Simulation(p) as s:
    s.set_state(ic["x"], ic["y"], ic["z"], ic["m"], ic["vx"], ic["vy"], ic["vz"])
    s.step(5); s.get_timings()
    s.step(20); t = s.get_timings()
out = {k[3:]: round(t[k] / 20, 4) for k in ("ms_integrate", "ms_sort", "ms_build", "ms_com", "ms_traverse", "ms_total")}
out["lib"] = os.path.basename(nb.LIB_PATH); out["n"] = n; out["bits"] = bits; out["kind"] = kind; out["fit"] = bool(fit); out["inter"] = t["interactions"] // 20
print(json.dumps(out))
