import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbody_amd as nb
n = 524288
ic = nb.sample_ic(nb.IC_DISK, n, 42)
sim = nb.Simulation(nb.Params(num_points=n, theta=0.75, dt=1e-4, softening_factor=1e-3, morton_bits=21))
sim.set_state(ic["x"], ic["y"], ic["z"], ic["m"], ic["vx"], ic["vy"], ic["vz"])
sim.step(20)
sim.sync()
sim.close()
