"""Sort of nearly sorted keys against the digit passes over everything: same permutation, same keys,
same state after a number of steps; step times.  python scratch/coh_check.py [kind] [n] [steps]"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nbody_amd as nb

kind = sys.argv[1] if len(sys.argv) > 1 else "sphere"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 21
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
kw = {"sphere": dict(theta=0.6, dt=1e-6, softening_factor=0.1),
      "disk": dict(theta=0.75, dt=1e-4, softening_factor=1e-3),
      "plummer": dict(theta=0.6, dt=1e-6, softening_factor=1e-3)}[kind]


def run(flags, timed=int(os.environ.get("COH_TIMED", "40"))):
    with nb.Simulation(nb.Params(num_points=n, morton_bits=21, flags=nb.FLAG_TIMING | nb.FLAG_TIMING_STAGES | flags, seed=42, **kw)) as s:
        if kind == "sphere":
            ic = nb.sample_ic(nb.IC_SPHERE, n, 42)
            s.set_state(ic["x"], ic["y"], ic["z"], ic["m"])
        elif kind == "plummer":
            s.spawn_plummer(0.1, 1.0)
        else:
            s.spawn_disk()
        perms = []
        for k in range(steps):
            s.step(1)
            o = s.get_octree()
            perms.append((o["perm"].copy(), o["codes_sorted"].copy()))
        st = s.get_state()
        s.get_timings()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); s.step(timed); s.sync()
            best = min(best, (time.perf_counter() - t0) / timed * 1e3)
        t = s.get_timings()
        stats = s.sort_stats()
    return perms, st, best, t, stats


pa, sa, ta, tma, sta = run(nb.FLAG_NO_COHERENT_SORT)
for name, fl in (("forced", nb.FLAG_ALWAYS_COHERENT_SORT), ("auto", 0)):
    pb, sb, tb, tmb, stb = run(fl)
    same_perm = all((a[0] == b[0]).all() and (a[1] == b[1]).all() for a, b in zip(pa, pb))
    same_state = all((sa[k].view(np.uint32) == sb[k].view(np.uint32)).all() for k in ("x", "vx", "ax", "az"))
    print(json.dumps({"kind": kind, "n": n, "mode": name, "perm_and_keys_equal": bool(same_perm), "state_bits_equal": bool(same_state),
                      "ms_per_step_full": round(ta, 4), "ms_per_step": round(tb, 4),
                      "sort_ms_full": round(tma["ms_sort"] / tma["steps"], 4), "sort_ms": round(tmb["ms_sort"] / tmb["steps"], 4),
                      "stats": stb}), flush=True)
