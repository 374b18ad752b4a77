"""Wall-clock ms/step without stage events: python scratch/step_time.py N bits [kind]"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbody_amd as nb
n, bits = int(sys.argv[1]), int(sys.argv[2])
kind = sys.argv[3] if len(sys.argv) > 3 else "sphere"
if kind == "disk":
    ic = nb.sample_ic(nb.IC_DISK, n, 42); kw = dict(theta=0.75, dt=1e-4, softening_factor=1e-3)
else:
    ic = nb.sample_ic(nb.IC_SPHERE, n, 42); kw = dict(theta=0.6, dt=1e-6, softening_factor=0.1)
with nb.Simulation(nb.Params(num_points=n, morton_bits=bits, flags=nb.FLAG_TIMING, **kw)) as s:
    s.set_state(ic["x"], ic["y"], ic["z"], ic["m"], ic["vx"], ic["vy"], ic["vz"])
    s.step(10); s.sync(); s.get_timings()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); s.step(50); s.sync(); best = min(best, (time.perf_counter() - t0) / 50 * 1e3)
    t = s.get_timings()
print(t)
print(json.dumps({"lib": os.path.basename(nb.LIB_PATH), "n": n, "kind": kind, "ms_per_step": round(best, 4), "traverse": round(t["ms_traverse"] / 150, 4), "rest": round(best - t["ms_traverse"] / 150, 4)}))
