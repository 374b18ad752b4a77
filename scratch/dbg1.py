import sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, nbody_amd as nb, oracle_lib as O
n, bits, theta, eps, dt = 4096, 10, 0.6, 0.01, 1e-3
ic = nb.sample_ic(nb.IC_DISK, n, 42)
with nb.Simulation(nb.Params(num_points=n, morton_bits=bits, theta=theta, dt=dt, softening_factor=eps, device_id=0)) as sim:
    sim.set_state(ic["x"], ic["y"], ic["z"], ic["m"], ic["vx"], ic["vy"], ic["vz"])
    g0 = sim.get_octree()
    sim.update()
    g = sim.get_octree()
    st = sim.get_state()
ref0 = O.full_build(ic["x"], ic["y"], ic["z"], ic["m"], 5.0, bits, 32)
for k in ("comx","comy","comz","mass"):
    d = g0[k] != ref0["com"][k]
    print("spawn", k, d.sum(), "of", d.size, np.abs(g0[k]-ref0["com"][k]).max())
p = ref0["perm"]; zero = np.zeros(n, np.float32)
x, y, z, _ = O.leapfrog_pos(dt, 5.0, ref0["xs"], ref0["ys"], ref0["zs"], ic["vx"][p], ic["vy"][p], ic["vz"][p], zero, zero, zero)
ref = O.full_build(x, y, z, ref0["ms"], 5.0, bits, 32)
for k in ("comx","comy","comz","mass"):
    d = g[k] != ref["com"][k]
    print("upd", k, d.sum(), "of", d.size, np.abs(g[k]-ref["com"][k]).max())
print("parent ok", g["parent"][:10])
ax, ay, az, ll = O.traverse(ref["tree"], ref["com"], ref["xs"], ref["ys"], ref["zs"], ref["ms"], 64, theta, eps, 1.0, 5.0)
print("list_len gpu", g["list_len"][:16]); print("list_len orc", ll[:16])
print("ndiff", (g["list_len"]!=ll).sum(), "of", ll.size)
gc = {k: g[k] for k in ("comx","comy","comz","mass")}
ax2, ay2, az2, ll2 = O.traverse(ref["tree"], gc, ref["xs"], ref["ys"], ref["zs"], ref["ms"], 64, theta, eps, 1.0, 5.0)
print("with gpu com ndiff", (g["list_len"]!=ll2).sum())
err = np.sqrt((st["ax"] - ax) ** 2 + (st["ay"] - ay) ** 2 + (st["az"] - az) ** 2)
rel = (err / np.sqrt(ax.astype(np.float64) ** 2 + ay ** 2 + az ** 2))
print("rel acc max", rel.max(), "mean", rel.mean())
