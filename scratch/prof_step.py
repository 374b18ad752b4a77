"""A few steps of one workload, for runs under rocprofv3:
python scratch/prof_step.py <bits> <n> <steps> [group] [f32|f64] [sphere|plummer|disk]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbody_amd as nb
bits = int(sys.argv[1]) if len(sys.argv) > 1 else 21
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 21
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
group = int(sys.argv[4]) if len(sys.argv) > 4 else 64
flags = nb.FLAG_FP64 if len(sys.argv) > 5 and sys.argv[5] == "f64" else 0
flags |= int(os.environ.get("NBODY_EXTRA_FLAGS", "0"))
kind = sys.argv[6] if len(sys.argv) > 6 else "sphere"
kw = {"sphere": dict(theta=0.6, dt=1e-6, softening_factor=0.1),
      "plummer": dict(theta=0.6, dt=1e-6, softening_factor=1e-3),
      "disk": dict(theta=0.75, dt=1e-4, softening_factor=1e-3)}[kind]
sim = nb.Simulation(nb.Params(num_points=n, morton_bits=bits, traversal_group_size=group, flags=flags, **kw))
if kind == "sphere":
    ic = nb.sample_ic(nb.IC_SPHERE, n, 42)
    sim.set_state(ic["x"], ic["y"], ic["z"], ic["m"])
elif kind == "plummer":
    sim.spawn_plummer(0.1, 1.0)
else:
    sim.spawn_disk()
sim.step(steps)
sim.sync()
print("ok", sim.tree_counts())
sim.close()
