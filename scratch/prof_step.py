import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbody_amd as nb
bits = int(sys.argv[1]) if len(sys.argv) > 1 else 21
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 21
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
group = int(sys.argv[4]) if len(sys.argv) > 4 else 64
flags = nb.FLAG_FP64 if len(sys.argv) > 5 and sys.argv[5] == "f64" else 0
ic = nb.sample_ic(nb.IC_SPHERE, n, 42)
sim = nb.Simulation(nb.Params(num_points=n, theta=0.6, dt=1e-6, softening_factor=0.1, morton_bits=bits, traversal_group_size=group, flags=flags))
sim.set_state(ic["x"], ic["y"], ic["z"], ic["m"])
sim.step(steps)
sim.sync()
print("ok", sim.tree_counts())
sim.close()
