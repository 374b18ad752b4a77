"""Config 4 through collapse and rebound: ms/step and tree statistics along the way."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nbody_amd as nb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 21
total = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
every = int(sys.argv[3]) if len(sys.argv) > 3 else 250
bits = int(sys.argv[4]) if len(sys.argv) > 4 else 21
ic = nb.sample_ic(nb.IC_SPHERE, n, 42)
sim = nb.Simulation(nb.Params(num_points=n, theta=0.6, dt=1e-6, softening_factor=0.1, morton_bits=bits,
                              flags=nb.FLAG_TIMING))
sim.set_state(ic["x"], ic["y"], ic["z"], ic["m"])
done = 0
while done < total:
    t0 = time.perf_counter()
    for _ in range(every // 50):
        sim.step(50); sim.get_timings() if False else None
        t = sim.get_timings()
    sim.sync()
    dt = (time.perf_counter() - t0) / every * 1e3
    done += every
    st = sim.get_state()
    r = np.sqrt((st["x"] - 2.5) ** 2 + (st["y"] - 2.5) ** 2 + (st["z"] - 2.5) ** 2)
    U, M, no = sim.tree_counts()
    print(f"step {done:5d}  {dt:7.3f} ms/step  trav {t['ms_traverse']/50:7.3f}  inter/step {t['interactions']/50:.3e}  "
          f"r50 {np.median(r):.4f} r99 {np.quantile(r, 0.99):.4f}  U {U} M {M} n_oct {no}  finite {np.isfinite(st['ax']).all()}", flush=True)
