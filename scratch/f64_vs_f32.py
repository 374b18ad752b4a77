import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import nbody_amd as nb
n = 1 << 21
ic = nb.sample_ic(nb.IC_SPHERE, n, 42)
res = {}
for name, fl in (("f32", 0), ("f64", nb.FLAG_FP64)):
    with nb.Simulation(nb.Params(num_points=n, morton_bits=21, theta=0.6, dt=0.0, softening_factor=0.1, flags=fl)) as sim:
        sim.set_state(ic["x"], ic["y"], ic["z"], ic["m"])
        sim.update()
        oc = sim.get_octree()
        res[name] = oc
        print(name, "U M n_oct", oc["U"], oc["M"], oc["n_oct"], "sum list_len*64", int(oc["list_len"].astype(np.int64).sum()) * 64,
              "leaf sizes mean", float(np.mean((oc["points_end"] - oc["points_begin"] + 1)[oc["num_children"] == 0])))
a, b = res["f32"], res["f64"]
print("codes equal frac", float(np.mean(a["codes_sorted"] == b["codes_sorted"])))
for lvl in range(4, 14):
    print(lvl, int(np.sum((a["level"] == lvl) & (a["num_children"] == 0))), int(np.sum((b["level"] == lvl) & (b["num_children"] == 0))))
