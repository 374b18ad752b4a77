import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbody_amd as nb
bits = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n = 1 << 21
ic = nb.sample_ic(nb.IC_SPHERE, n, 42)
sim = nb.Simulation(nb.Params(num_points=n, theta=0.6, dt=1e-6, softening_factor=0.1, morton_bits=bits, flags=nb.FLAG_TIMING))
sim.set_state(ic["x"], ic["y"], ic["z"], ic["m"])
sim.step(3); sim.get_timings()
sim.step(20)
t = sim.get_timings()
print({k: round(v / 20, 4) if k.startswith("ms") else v for k, v in t.items()})
sim.close()
