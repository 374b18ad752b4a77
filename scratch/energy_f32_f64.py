"""Total energy (kinetic + tree potential) of the headline cold sphere through collapse and
rebound, float handle against double handle: the comparison the reference draws in
plots/plot_energy.py ("double" / "float")."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nbody_amd as nb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 21
total = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
every = int(sys.argv[3]) if len(sys.argv) > 3 else 250
ic = nb.sample_ic(nb.IC_SPHERE, n, 42)
for name, fl in (("f32", 0), ("f64", nb.FLAG_FP64)):
    sim = nb.Simulation(nb.Params(num_points=n, theta=0.6, dt=1e-6, softening_factor=0.1, morton_bits=21, flags=fl))
    sim.set_state(ic["x"], ic["y"], ic["z"], ic["m"])
    sim.step(5)
    e0 = sum(sim.tree_energy())
    done, t0 = 5, time.perf_counter()
    while done < total:
        sim.step(every); done += every
        ke, pe = sim.tree_energy()
        st = sim.get_state()
        r = np.sqrt((st["x"] - 2.5) ** 2 + (st["y"] - 2.5) ** 2 + (st["z"] - 2.5) ** 2)
        print(json.dumps({"precision": name, "step": done, "kinetic": ke, "potential": pe,
                          "rel_change": (ke + pe - e0) / abs(e0), "r50": float(np.median(r)),
                          "finite": bool(np.isfinite(st["ax"]).all()),
                          "s_elapsed": round(time.perf_counter() - t0, 1)}), flush=True)
    sim.close()
