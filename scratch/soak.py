"""Soak: repeated builds against the oracle (COM protocol, look-back chains) and a long run."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import oracle_lib as O
import nbody_amd as nb
rng = np.random.default_rng(123)
XF = int(os.environ.get("NBODY_EXTRA_FLAGS", "0"))   # e.g. 2048: every sort by the nearly-sorted path
t0 = time.time()
bad = 0
n = 200_000
for bits in (21, 10):
    with nb.Simulation(nb.Params(num_points=n, morton_bits=bits, theta=0.6, softening_factor=0.01, flags=XF)) as sim:
        for it in range(40):
            k = rng.integers(1, 12)
            c = rng.uniform(0.5, 4.5, (k, 3))
            p = (c[rng.integers(0, k, n)] + rng.normal(0, rng.uniform(0.005, 0.3), (n, 3))).clip(0, 5).astype(np.float32)
            m = rng.uniform(0.5, 1.5, n).astype(np.float32)
            sim.set_state(p[:, 0].copy(), p[:, 1].copy(), p[:, 2].copy(), m)
            g = sim.get_octree()
            r = O.full_build(p[:, 0], p[:, 1], p[:, 2], m, 5.0, bits, 32)
            ok = (g["n_oct"] == r["tree"]["n_oct"] and np.array_equal(g["perm"], r["perm"])
                  and all(np.array_equal(g[a], r["com"][a]) for a in ("comx", "comy", "comz", "mass"))
                  and np.array_equal(g["first_child"][g["num_children"] > 0], r["tree"]["first_child"][r["tree"]["num_children"] > 0]))
            bad += not ok
            if not ok: print("MISMATCH bits", bits, "iter", it, flush=True)
    print("bits", bits, "40 builds checked, mismatches so far", bad, round(time.time() - t0, 1), "s", flush=True)
# long run
n = 1 << 20
ic = nb.sample_ic(nb.IC_PLUMMER, n, 7, p0=0.1, p1=1.0)
with nb.Simulation(nb.Params(num_points=n, morton_bits=21, theta=0.6, dt=2e-4, softening_factor=2e-3, flags=XF)) as sim:
    sim.set_state(ic["x"], ic["y"], ic["z"], ic["m"])
    sim.step(1)
    e0 = sum(sim.tree_energy())
    for blk in range(10):
        sim.step(1000)
        e = sum(sim.tree_energy())
        st = sim.get_state()
        print("steps", (blk + 1) * 1000, "E", e, "rel change", (e - e0) / abs(e0), "finite", bool(np.isfinite(st["ax"]).all()),
              "n_oct", sim.get_num_octree_nodes(), round(time.time() - t0, 1), "s", flush=True)
    print("sorts", sim.sort_stats())
print("SOAK DONE mismatches", bad)
