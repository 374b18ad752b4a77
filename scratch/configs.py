"""Times the BASELINE.json configurations that fit one GPU (not the bench line)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nbody_amd as nb

def run(name, n, kind, theta, eps, dt, bits, steps=20, group=64, fit=False, **kw):
    p = nb.Params(num_points=n, theta=theta, dt=dt, softening_factor=eps, morton_bits=bits, traversal_group_size=group,
                  flags=nb.FLAG_TIMING | (nb.FLAG_FIT_KEYS if fit else 0))
    with nb.Simulation(p) as sim:
        if kind == "disk": sim.spawn_disk()
        elif kind == "sphere": sim.spawn_sphere()
        elif kind == "plummer": sim.spawn_plummer(0.1, 1.0)
        U, M, no = sim.tree_counts()
        sim.step(3); sim.get_timings()
        t0 = time.perf_counter(); sim.step(steps); sim.sync(); t1 = time.perf_counter()
        t = sim.get_timings()
        st = sim.get_state()
    build = (t["ms_integrate"] + t["ms_sort"] + t["ms_build"] + t["ms_com"]) / steps
    print(json.dumps({"config": name, "n": n, "bits": bits, "fitted_key_domain": fit, "group": group, "ms_per_step": round((t1 - t0) * 1e3 / steps, 3),
                      "build_ms": round(build, 3), "traverse_ms": round(t["ms_traverse"] / steps, 3),
                      "interactions_per_step": t["interactions"] // steps,
                      "tflops20": round(20 * t["interactions"] / (t["ms_traverse"] * 1e-3) / 1e12, 1),
                      "U": U, "M": M, "n_oct": no, "finite": bool(np.isfinite(st["ax"]).all())}), flush=True)

which = sys.argv[1:] or ["c2", "c4", "c4m", "c5"]
for bits, group, fit in ((21, 64, False), (21, 32, False), (10, 64, False), (10, 64, True)):
    if "c2" in which: run("524k disk theta=0.75 (README ~27 ms/frame)", 524288, "disk", 0.75, 1e-3, 1e-4, bits, group=group, fit=fit)
    if "c4" in which: run("2M sphere theta=0.6 (README ~150 ms)", 1 << 21, "sphere", 0.6, 0.1, 1e-6, bits, group=group, fit=fit)
    if "c4m" in which: run("4M sphere theta=0.6 (README ~430 ms)", 1 << 22, "sphere", 0.6, 0.1, 1e-6, bits, group=group, fit=fit)
    if "c5" in which: run("16.7M Plummer theta=0.6 on ONE GPU", 1 << 24, "plummer", 0.6, 1e-3, 1e-6, bits, steps=5, group=group, fit=fit)
