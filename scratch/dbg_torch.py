import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nbody_amd as nb
mode = sys.argv[1]
print("device_count before", torch.cuda.device_count())
if mode == "first":
    torch.cuda.init(); print("torch init ok")
for i in range(int(sys.argv[2])):
    with nb.Simulation(nb.Params(num_points=4096, morton_bits=21)) as s:
        s.spawn_disk(); s.update(); s.sync()
print("sims ok")
torch.cuda.init()
print("torch init ok late", torch.zeros(3, device="cuda").sum().item())
