"""Step time against N and theta, the axes of the reference's plots/exec-scaling.png and
exec-build-traversal-stacked.png (their initial condition is not stated; disk and sphere here)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbody_amd as nb
for kind in ("disk", "sphere"):
    eps, dt = (1e-3, 1e-4) if kind == "disk" else (0.1, 1e-6)      # the renderer's values per spawner
    for theta in (0.25, 0.5, 0.75):
        for group in (32, 64):
            row = []
            for e in range(16, 22):
                n = 1 << e
                p = nb.Params(num_points=n, theta=theta, dt=dt, softening_factor=eps, morton_bits=21,
                              traversal_group_size=group, flags=nb.FLAG_TIMING)
                with nb.Simulation(p) as sim:
                    sim.spawn_disk() if kind == "disk" else sim.spawn_sphere()
                    sim.step(3); sim.get_timings(); sim.sync()
                    t0 = time.perf_counter(); sim.step(10); sim.sync(); t1 = time.perf_counter()
                    t = sim.get_timings()
                row.append((round((t1 - t0) * 100, 3), round(t["ms_build"] / 10, 3), round(t["ms_traverse"] / 10, 3)))
            print(json.dumps({"ic": kind, "theta": theta, "group": group, "N": "2^16..2^21",
                              "ms_per_step": [r[0] for r in row], "build_ms": [r[1] for r in row],
                              "traversal_ms": [r[2] for r in row]}), flush=True)
