"""Cost of the nearly-sorted path against the number of moved elements: the sphere's collapse
speeds up from rest, so the moved share grows step by step.  python scratch/coh_curve.py [n] [steps]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbody_amd as nb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 21
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 80
for name, fl in (("full", nb.FLAG_NO_COHERENT_SORT), ("forced", nb.FLAG_ALWAYS_COHERENT_SORT)):
    with nb.Simulation(nb.Params(num_points=n, morton_bits=21, flags=nb.FLAG_TIMING | nb.FLAG_TIMING_STAGES | fl, seed=42,
                                 theta=0.6, dt=1e-6, softening_factor=0.1)) as s:
        ic = nb.sample_ic(nb.IC_SPHERE, n, 42)
        s.set_state(ic["x"], ic["y"], ic["z"], ic["m"])
        rows = []
        for k in range(steps):
            s.step(1); s.sync()
            t = s.get_timings()
            rows.append((k, s.sort_stats()["last_moved"], round(t["ms_sort"], 4)))
        print(name, json.dumps(rows[::2]))
