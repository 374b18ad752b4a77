#!/usr/bin/env python3
"""bench.py -- headline benchmark of the Barnes-Hut step on MI355X.

Workload (BASELINE.json configs[3], the README's "~150 ms/step" case): 2,097,152-particle
cold uniform sphere (r = 0.5, centred in a domain of 5.0, m = 1, eps = 0.1, dt = 1e-6,
the reference's SPAWN_SPHERE parameters), theta = 0.6, 32 codes per leaf, wave64 groups.
A step is one Simulation::update(): kick-drift + Morton, radix sort, tree build, COM,
traversal, kick.  Synthetic data, generated on the host before the timed region; every
input is resident in HBM when timing starts.

    python bench.py --gpus N --steps K --warmup W

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): weak scaling with
2,097,152 particles per GPU, i.e. one sphere of N x 2,097,152 particles.
  --parallelism let (default): domain decomposition along the Morton curve (nbody_dd_*).
      Every rank samples its own slice of the initial conditions, the ranks agree on
      Morton-key ranges of equal particle count (all-gathered key histograms) and migrate
      the particles; per step every rank builds and traverses the tree of its own range and
      imports a locally-essential tree from the others (all-gather of the top-of-tree
      nodes + one all-to-all of (x,y,z,m) records).  No rank ever holds the whole set.
  --parallelism replicated: every rank holds the whole particle set and builds the same
      tree; the traversal is sharded by Morton-contiguous group ranges and the
      accelerations are exchanged with one all-gather per step (bit-identical to 1 GPU).

Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_PER_GPU = 2_097_152
THETA = 0.6
EPS = 0.1
DT = 1e-6
FLOP_PER_INTERACTION = 20          # GPU-Gems/Nyland convention for one softened pair force
# algorithmic bytes per particle of integrate+sort+permute+build+COM, by key width: kick-drift
# and encode 36 R + 28/32 W; the digit histograms in one key read 4/8; 4/5 digit passes of
# 16/24 (key + index, read and written once; 63-bit keys are sorted on their upper 40 bits and
# the rare ties ordered in place, one more key read); permute 36 R + 64 W; unique 12/20; tree
# kernels (leaves, radix, depth, rank, emit) ~16; centre of mass 20
BUILD_BYTES_PER_PARTICLE = {10: 64 + 4 + 4 * 16 + 100 + 12 + 16 + 20,        # 30-bit keys: 280
                            21: 68 + 8 + 5 * 24 + 8 + 100 + 20 + 16 + 20}    # 63-bit keys: 360
PEAK_FP32_TFLOPS = 157.3           # MI355X FP32 vector peak
PEAK_FP64_TFLOPS = 78.6            # MI355X FP64 vector peak
PEAK_HBM_GBS = 8000.0              # MI355X HBM3E peak
# shader clock measured inside k_traverse with the profiling build (s_memtime against
# s_memrealtime over 400 back-to-back steps): 2.26-2.30 GHz on the box of profiles/r3_walk_profile.txt;
# an earlier box held 2.0 GHz (profiles/r1_walk_profile.txt) -- devices differ
SUSTAINED_CLOCK_GHZ = 2.28
REFERENCE_MS_PER_STEP = 150.0      # reference README, RTX 500 Ada, 2,097,152 particles, theta=0.6


_REAL_STDOUT = None


def quiet_stdout():
    """Libraries below print to stdout on their own (RCCL's \"Librccl path\", gloo's connection
    lines): everything written to file descriptor 1 goes to stderr until emit() restores it, so
    that the one JSON line is all that stdout ever carries."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    sys.stdout.flush()
    if _REAL_STDOUT is not None:
        os.dup2(_REAL_STDOUT, 1)
    print(json.dumps(obj), flush=True)
    if _REAL_STDOUT is not None:
        os.dup2(2, 1)


def cpu_baseline(ic, bits, steps=1, eps=EPS):
    """The host oracle (same algorithm, same parameters) timed on this machine's cores on
    `steps` steps of the full 2,097,152-particle workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    cores = O.lib().orc_num_threads()
    t0 = time.time()
    sim = O.OracleSim(ic["x"], ic["y"], ic["z"], ic["m"], bits=bits, max_codes=32, group=64,
                      theta=THETA, dt=DT, gravity=1.0, eps=eps, domain=5.0)
    spawn_s = time.time() - t0
    ms, inter, tb, tt = [], 0, 0.0, 0.0
    for _ in range(steps):
        t0 = time.time()
        sim.step()
        ms.append((time.time() - t0) * 1e3)
        tm = sim.timings()
        tb += tm["ms_build"]
        tt += tm["ms_traverse"]
        inter = sim.counts()[3]
    sim.close()
    mean = sum(ms) / len(ms)
    return {"value": round(mean, 1), "unit": "ms/step", "cores": cores, "kind": "port",
            "sample": f"{steps} step(s) of the full {len(ic['x'])}-particle workload after the "
                      f"spawn build ({spawn_s:.1f} s); every stage OpenMP (parallel stable sort, tree loops, level-parallel centre of mass, traversal)",
            "build_ms": round(tb / steps, 1), "traversal_ms": round(tt / steps, 1),
            "interactions_per_s": round(inter / (tt / steps / 1e3)) if tt > 0 else None}


def accuracy_sample(st, eps, k=4096):
    """Step-local relative acceleration error of the timed run's last step (the reference's
    metric, src/cuda/validator.cu:38-63) on k randomly chosen bodies, against the host oracle's
    float64 direct sum over all particles at the same positions."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    n = len(st["x"])
    idx = np.random.default_rng(1).choice(n, k, replace=False)
    rest = np.ones(n, bool)
    rest[idx] = False
    order = np.concatenate([idx, np.flatnonzero(rest)])       # the sampled bodies first
    t0 = time.perf_counter()
    r = O.direct_f64(st["x"][order], st["y"][order], st["z"][order], st["m"][order], eps, 1.0, 0, k)
    mean, mx = O.acc_error(st["ax"][idx], st["ay"][idx], st["az"][idx], *r)
    return {"bodies_sampled": k, "mean_rel_acc_error": float(f"{mean:.3e}"), "max_rel_acc_error": float(f"{mx:.3e}"),
            "against": f"float64 direct sum over all {n} particles at the same positions (host oracle, "
                       f"{time.perf_counter() - t0:.1f} s)",
            "note": "monopole Barnes-Hut at this theta, as in the reference; inside a homogeneous sphere the "
                    "attractions nearly cancel, which inflates a relative measure (the 32,768-particle disk of "
                    "the reference's error plot gives 5e-4 at theta = 0.5, asserted in the tests)"}


def direct_sum_block(no_cpu):
    """BASELINE.json configs[2] on one GPU, reported beside the headline value: the O(N^2)
    direct sum of 1,048,576 bodies (nbody_direct_sum_dev, the validator's kernel), timed with
    events on torch's current stream, which is the stream the kernel is launched on."""
    import numpy as np
    import torch
    import nbody_amd as nb
    n, eps, grav = 1_048_576, 0.01, 1.0
    ic = nb.sample_ic(nb.IC_SPHERE, n, 42)
    posm = np.stack([ic["x"], ic["y"], ic["z"], np.full(n, 1.0 / n, np.float32)], 1).astype(np.float32)
    dev_posm = torch.from_numpy(posm).cuda()
    out = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(3)]
    lib = nb.load_library()
    stream = torch.cuda.current_stream().cuda_stream

    def launch():
        rc = lib.nbody_direct_sum_dev(dev_posm.data_ptr(), n, dev_posm.data_ptr(), n, eps, grav,
                                      out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), 0, stream)
        if rc != 0:
            raise RuntimeError(f"nbody_direct_sum_dev failed with status {rc}")

    launch()
    torch.cuda.synchronize()
    reps = 3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    rate = float(n) * n / (ms * 1e-3)
    tf = FLOP_PER_INTERACTION * rate / 1e12
    cpu = None
    if not no_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        nt = 65536
        tc = time.perf_counter()
        O.direct_f32(posm[:, 0], posm[:, 1], posm[:, 2], posm[:, 3], eps, grav, 0, nt)
        tc = time.perf_counter() - tc
        cpu = {"value": round(nt * float(n) / tc), "unit": "interactions/s",
               "cores": O.lib().orc_num_threads(), "kind": "port",
               "sample": f"{nt} target bodies x {n} sources ({tc:.1f} s), OpenMP over targets"}
    return {"workload": "direct sum, 1,048,576-body uniform sphere r=0.5, m=1/N, eps=0.01 "
                        "(the validator's all-pairs kernel, one launch)",
            "ms_per_launch": round(ms, 3), "value": round(rate), "unit": "interactions/s",
            "finite": bool(torch.isfinite(out[0]).all().item()),
            "roofline": {"bound": "valu", "kernel": "k_direct", "achieved": round(tf, 2),
                         "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / PEAK_FP32_TFLOPS, 4),
                         "traffic": None,
                         "issue_bound_frac_at_sustained_clock": round(rate / (1024 * SUSTAINED_CLOCK_GHZ * 1e9 * 2), 4),
                         "note": "20 FLOP x N^2 / mean kernel time over 3 launches; the ceiling of the "
                                 "14-instruction pair is 2 interactions per cycle per SIMD"},
            "cpu_baseline": cpu}


def run_direct(args, world, rank, local_rank):
    """BASELINE.json configs[2]: O(N^2) direct sum of 1,048,576 bodies, bodies block-partitioned
    over the ranks, (x,y,z,m) tiles passed round a ring (nbody_amd.parallel.ring_direct_sum)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import nbody_amd as nb
    from nbody_amd import parallel
    n = 1_048_576
    eps, grav = 0.01, 1.0
    n_local = n // world
    ic = nb.sample_ic(nb.IC_SPHERE, n, 42)
    posm = np.stack([ic["x"], ic["y"], ic["z"], np.full(n, 1.0 / n, np.float32)], 1).astype(np.float32)
    dev = f"cuda:{local_rank}"
    mine = torch.from_numpy(posm[rank * n_local:(rank + 1) * n_local].copy()).to(dev)
    out = [torch.zeros(n_local, dtype=torch.float32, device=dev) for _ in range(3)]
    lib = nb.load_library()
    kern = parallel.gpu_tile_kernel(lib, *out, eps, grav)
    if world == 1:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(dev),
                                init_method=f"tcp://127.0.0.1:{29500 + os.getpid() % 1000}")

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        parallel.ring_direct_sum(mine, kern)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        parallel.ring_direct_sum(mine, kern)
    fence()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    finite = bool(torch.isfinite(out[0]).all().item())
    if rank == 0:
        K = max(args.steps, 1)
        rate = float(n) * n * K / elapsed
        tf = 20 * rate / 1e12
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            # the oracle's float32 direct sum on 131,072 target bodies against all 1,048,576 sources
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O
            nt = 131072
            tc = time.perf_counter()
            O.direct_f32(posm[:, 0], posm[:, 1], posm[:, 2], posm[:, 3], eps, grav, 0, nt)
            tc = time.perf_counter() - tc
            cpu = {"value": round(nt * float(n) / tc), "unit": "interactions/s",
                   "cores": O.lib().orc_num_threads(), "kind": "port",
                   "sample": f"{nt} target bodies x {n} sources ({tc:.1f} s), OpenMP over targets"}
        emit(({
            "metric": "interactions/s, O(N^2) direct sum", "value": round(rate), "unit": "interactions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed * 1e3 / K, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "direct sum, 1,048,576-body uniform sphere r=0.5, m=1/N, eps=0.01",
                       "parallelism": "single" if world == 1 else f"ring of {world} position tiles"},
            "finite": finite,
            "roofline": {"bound": "valu", "kernel": "k_direct", "achieved": round(tf, 2),
                         "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / PEAK_FP32_TFLOPS, 4),
                         "traffic": None, "note": "20 FLOP x N^2 / wall time incl. ring exchange"},
            "cpu_baseline": cpu}))
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=["bh", "direct"], default="bh",
                    help="bh = the headline Barnes-Hut step (default); direct = O(N^2) direct sum")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--morton-bits", type=int, default=int(os.environ.get("NBODY_MORTON_BITS", "21")),
                    help="bits per axis of the Morton key: 21 = the reference's 63-bit keys "
                         "(default), 10 = 30-bit keys")
    ap.add_argument("--no-alt-bits", action="store_true",
                    help="skip the short extra run with the other key width")
    ap.add_argument("--particles-per-gpu", type=int, default=N_PER_GPU)
    ap.add_argument("--ic", choices=["sphere", "plummer"], default="sphere",
                    help="sphere = the headline workload; plummer = cold Plummer sphere a=0.1, r<=1, "
                         "total mass 1, eps=1e-3 (16,777,216 particles at --gpus 8)")
    ap.add_argument("--parallelism", choices=["let", "replicated"], default="let",
                    help="multi-GPU scheme (ignored with one GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-direct-sum", action="store_true",
                    help="1 GPU only: skip the direct-sum block reported beside the headline value")
    ap.add_argument("--evolve-steps", type=int, default=2000,
                    help="1 GPU only: after the timed region advance this many more steps (the cold "
                         "sphere collapses and rebounds within ~300) and time 20 steps of the "
                         "clustered state, reported beside `value`; 0 = skip")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="development aid: all ranks share GPU 0 and talk over gloo (the numbers "
                         "mean nothing; it exercises the multi-rank code path on a 1-GPU box)")
    ap.add_argument("--walk-only", action="store_true",
                    help="profiling aid: skip the force arithmetic (results meaningless)")
    args = ap.parse_args()
    quiet_stdout()

    import numpy as np
    import torch
    import nbody_amd as nb

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with "
                             "python -m torch.distributed.run --nproc-per-node N\n")
        sys.exit(2)
    if not torch.cuda.is_available():
        sys.stderr.write("bench.py needs a GPU (no CPU fallback)\n")
        sys.exit(2)
    rehearse = args.rehearse_on_one_gpu
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    coll_dev = "cpu" if rehearse else f"cuda:{local_rank}"
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    if args.workload == "direct":
        run_direct(args, world, rank, local_rank)
        return

    n = args.particles_per_gpu * world
    group = 64
    ngroups = (n + group - 1) // group
    let = world > 1 and args.parallelism == "let"
    eps = EPS
    kind, p0, p1 = nb.IC_SPHERE, 0.0, 0.0
    if args.ic == "plummer":
        kind, p0, p1 = nb.IC_PLUMMER, 0.1, 1.0
        eps = 1e-3
    flags = nb.FLAG_TIMING | (nb.FLAG_DEBUG_WALK_ONLY if args.walk_only else 0)
    ic = None
    stepper = None
    if let:
        from nbody_amd import parallel
        # head-room: equal-count key ranges hold n / world particles each, within 1 %
        cap = int(args.particles_per_gpu * 1.25) + 4096
        params = nb.Params(num_points=cap, max_num_codes_per_leaf=32, traversal_group_size=group,
                           theta=THETA, dt=DT, gravity=1.0, softening_factor=eps, domain_size=5.0,
                           morton_bits=args.morton_bits, device_id=local_rank, flags=flags)
        sim = nb.Simulation(params)
        stepper = parallel.DomainDecomposition(sim, f"cuda:{local_rank}", via_host=rehearse)
        inject = os.environ.get("NBODY_BENCH_FAIL_ALL_TO_ALL")   # test aid: "<rank>:<call index>"
        if inject and int(inject.split(":")[0]) == rank:
            stepper.comm.fail_all_to_all_at = int(inject.split(":")[1])
        try:
            stepper.spawn(kind, n, p0, p1)      # own slice of the set, then key ranges + migration
        except Exception as e:   # noqa: BLE001
            sys.stderr.write(f"bench.py rank {rank}: spawn failed: {e}\n")
            sys.stderr.flush()
            os._exit(3)
        n_here = stepper.stats()["n_local"]
    else:
        ic = nb.sample_ic(kind, n, 42, p0=p0, p1=p1)
        assert ngroups % world == 0, "groups must divide evenly over the ranks"
        n_here = len(ic["x"])
        params = nb.Params(num_points=n_here, max_num_codes_per_leaf=32, traversal_group_size=group,
                           theta=THETA, dt=DT, gravity=1.0, softening_factor=eps, domain_size=5.0,
                           morton_bits=args.morton_bits, device_id=local_rank, flags=flags)
        sim = nb.Simulation(params)
        sim.set_state(ic["x"], ic["y"], ic["z"], ic["m"])
        if world > 1:
            from nbody_amd import parallel
            stepper = parallel.ShardedTraversal(parallel.GpuBackend(sim, f"cuda:{local_rank}"))
    U, M, n_oct = sim.tree_counts()

    def die(where, exc):
        # a failed collective or nbody_dd_* call: say what failed on this rank and leave at once,
        # without waiting on the peers (the launcher ends them)
        sys.stderr.write(f"bench.py rank {rank}: {where} failed: {exc}\n")
        sys.stderr.flush()
        os._exit(3)

    def one_step():
        if stepper is None:
            sim.update()
        else:
            try:
                stepper.step()
            except Exception as e:   # noqa: BLE001
                die("step", e)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    done = 0
    while done < args.warmup:
        one_step()
        done += 1
    fence()
    if args.warmup:
        sim.get_timings()  # drop warm-up records
    tim = dict(steps=0, ms_integrate=0.0, ms_sort=0.0, ms_build=0.0, ms_com=0.0, ms_traverse=0.0,
               ms_total=0.0, interactions=0)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step()
    fence()
    elapsed = time.perf_counter() - t0
    t = sim.get_timings()
    for k in tim:
        tim[k] += t[k]
    if world > 1:
        red = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(red, op=dist.ReduceOp.MAX)
        elapsed = float(red.item())
        it = torch.tensor([tim["interactions"]], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(it, op=dist.ReduceOp.SUM)
        total_inter = int(it.item())
    else:
        total_inter = int(tim["interactions"])
    st = sim.get_state()
    ok = bool(np.isfinite(st["ax"]).all() and np.isfinite(st["x"]).all())
    e_start = sum(sim.tree_energy()) if world == 1 and not args.walk_only else None
    accuracy = None
    if world == 1 and not args.no_cpu_baseline and not args.walk_only:
        accuracy = accuracy_sample(st, eps)
    evolved = None
    if world == 1 and args.evolve_steps > 0 and not args.walk_only:
        sim.step(args.evolve_steps)
        sim.get_timings()
        sim.sync()
        ta = time.perf_counter()
        sim.step(20)
        sim.sync()
        tb = time.perf_counter()
        t3 = sim.get_timings()
        s3 = sim.get_state()
        r = np.sqrt((s3["x"] - 2.5) ** 2 + (s3["y"] - 2.5) ** 2 + (s3["z"] - 2.5) ** 2)
        evolved = {"after_steps": args.warmup + args.steps + args.evolve_steps,
                   "ms_per_step": round((tb - ta) * 1e3 / 20, 4),
                   "traversal_ms": round(t3["ms_traverse"] / 20, 4),
                   "interactions_per_step": round(t3["interactions"] / 20),
                   "half_mass_radius": round(float(np.median(r)), 4),
                   "r99": round(float(np.quantile(r, 0.99)), 4),
                   "finite": bool(np.isfinite(s3["ax"]).all() and np.isfinite(s3["x"]).all())}
        e_end = sum(sim.tree_energy())
        evolved["energy"] = {"after_timed_region": e_start, "after_collapse": e_end,
                             "relative_change": round((e_end - e_start) / abs(e_start), 6),
                             "note": "kinetic + tree potential (nbody_tree_energy)"}
    let_stats = stepper.stats() if let else None
    per_rank = None
    if world > 1:
        # what every rank did, for reading a run whose aggregate looks wrong: per-rank step split
        # and, for the domain decomposition, the phases of three more steps with an event between
        # them (outside the timed region: each event costs a few microseconds of stream time)
        mine = {"rank": rank, "particles": int(n_here), "build_ms": (tim["ms_integrate"] + tim["ms_sort"] + tim["ms_build"] + tim["ms_com"]) / max(args.steps, 1),
                "traversal_ms": tim["ms_traverse"] / max(args.steps, 1), "interactions_per_step": tim["interactions"] / max(args.steps, 1)}
        if let:
            try:
                stepper.set_phase_timing(True)
                acc = None
                for _ in range(3):
                    stepper.step()
                    ph = stepper.phases()
                    acc = ph if acc is None else {k: acc[k] + v for k, v in ph.items()}
                mine["phases"] = {k: v / 3 for k, v in acc.items()}
                stepper.set_phase_timing(False)
                mine["collective_calls"] = dict(stepper.comm.calls)
            except Exception as e:   # noqa: BLE001
                die("phase-timing steps", e)
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        per_rank = gathered
    if let:
        stepper.close()
    sim.close()

    stage = None
    if world == 1 and not args.walk_only:
        # per-stage split of the build from a short extra run: each extra event costs a few
        # microseconds of stream time, so the timed region above marks only the traversal
        ps = nb.Params(**{**params.__dict__, "flags": nb.FLAG_TIMING | nb.FLAG_TIMING_STAGES})
        with nb.Simulation(ps) as s4:
            s4.set_state(ic["x"], ic["y"], ic["z"], ic["m"])
            s4.step(3)
            s4.get_timings()
            s4.step(10)
            t4 = s4.get_timings()
        stage = {k[3:]: round(t4[k] / 10, 4) for k in
                 ("ms_integrate", "ms_sort", "ms_build", "ms_com", "ms_traverse")}
    alt = None
    if world == 1 and not args.no_alt_bits and not args.walk_only:
        # the other key width on the same particles, for the record (not part of `value`)
        alt_bits = 10 if args.morton_bits == 21 else 21
        p2 = nb.Params(**{**params.__dict__, "morton_bits": alt_bits})
        with nb.Simulation(p2) as s2:
            s2.set_state(ic["x"], ic["y"], ic["z"], ic["m"])
            s2.step(3)
            s2.get_timings()
            torch.cuda.synchronize()
            ta = time.perf_counter()
            s2.step(20)
            s2.sync()
            tb = time.perf_counter()
            t2 = s2.get_timings()
        alt = {"morton_bits": alt_bits, "ms_per_step": round((tb - ta) * 1e3 / 20, 4),
               "traversal_ms": round(t2["ms_traverse"] / 20, 4),
               "build_ms": round((t2["ms_integrate"] + t2["ms_sort"] + t2["ms_build"] + t2["ms_com"]) / 20, 4),
               "interactions_per_step": round(t2["interactions"] / 20)}
    g32 = None
    if world == 1 and not args.no_alt_bits and not args.walk_only:
        # the reference renderer's default traversal_group_size (src/renderer.cpp:305), for the record
        p3 = nb.Params(**{**params.__dict__, "traversal_group_size": 32})
        with nb.Simulation(p3) as s5:
            s5.set_state(ic["x"], ic["y"], ic["z"], ic["m"])
            s5.step(3)
            s5.get_timings()
            torch.cuda.synchronize()
            ta = time.perf_counter()
            s5.step(20)
            s5.sync()
            tb = time.perf_counter()
            t5 = s5.get_timings()
        g32 = {"traversal_group_size": 32, "ms_per_step": round((tb - ta) * 1e3 / 20, 4),
               "traversal_ms": round(t5["ms_traverse"] / 20, 4),
               "interactions_per_step": round(t5["interactions"] / 20)}
    f64 = None
    if world == 1 and not args.no_alt_bits and not args.walk_only:
        # the reference's Simulation<double> on the same particles, for the record
        p6 = nb.Params(**{**params.__dict__, "flags": nb.FLAG_TIMING | nb.FLAG_FP64})
        with nb.Simulation(p6) as s6:
            s6.set_state(ic["x"], ic["y"], ic["z"], ic["m"])
            s6.step(2)
            s6.get_timings()
            torch.cuda.synchronize()
            ta = time.perf_counter()
            s6.step(10)
            s6.sync()
            tb = time.perf_counter()
            t6 = s6.get_timings()
        tf64 = FLOP_PER_INTERACTION * t6["interactions"] / (t6["ms_traverse"] * 1e-3) / 1e12
        f64 = {"dtype": "f64", "ms_per_step": round((tb - ta) * 1e3 / 10, 4),
               "traversal_ms": round(t6["ms_traverse"] / 10, 4),
               "interactions_per_step": round(t6["interactions"] / 10),
               "traversal_tflops": round(tf64, 2), "peak_fp64_vector_tflops": PEAK_FP64_TFLOPS,
               "frac": round(tf64 / PEAK_FP64_TFLOPS, 4)}
    fit = None
    if world == 1 and not args.no_alt_bits and not args.walk_only:
        # 30-bit keys taken relative to a cube fitted to the particles (NBODY_FLAG_FIT_KEYS):
        # the reference's key width at the resolution the particles need, for the record
        p7 = nb.Params(**{**params.__dict__, "morton_bits": 10, "flags": nb.FLAG_TIMING | nb.FLAG_FIT_KEYS})
        with nb.Simulation(p7) as s7:
            s7.set_state(ic["x"], ic["y"], ic["z"], ic["m"])
            s7.step(3)
            s7.get_timings()
            torch.cuda.synchronize()
            ta = time.perf_counter()
            s7.step(20)
            s7.sync()
            tb = time.perf_counter()
            t7 = s7.get_timings()
            cube = s7.key_domain()
        fit = {"morton_bits": 10, "fitted_key_domain": True, "cube_side": round(float(cube[3]), 5),
               "ms_per_step": round((tb - ta) * 1e3 / 20, 4),
               "traversal_ms": round(t7["ms_traverse"] / 20, 4),
               "build_ms": round((t7["ms_integrate"] + t7["ms_sort"] + t7["ms_build"] + t7["ms_com"]) / 20, 4),
               "interactions_per_step": round(t7["interactions"] / 20)}
    disk = None
    if world == 1 and not args.no_alt_bits and not args.walk_only:
        # BASELINE.json configs[1]: the reference's ~27 ms/frame case, for the record
        nd = 524_288
        icd = nb.sample_ic(nb.IC_DISK, nd, 42)
        pd_ = nb.Params(num_points=nd, max_num_codes_per_leaf=32, traversal_group_size=group, theta=0.75,
                        dt=1e-4, gravity=1.0, softening_factor=1e-3, domain_size=5.0,
                        morton_bits=args.morton_bits, device_id=local_rank, flags=nb.FLAG_TIMING)
        with nb.Simulation(pd_) as s8:
            s8.set_state(icd["x"], icd["y"], icd["z"], icd["m"], icd["vx"], icd["vy"], icd["vz"])
            s8.step(5)
            s8.get_timings()
            torch.cuda.synchronize()
            ta = time.perf_counter()
            s8.step(50)
            s8.sync()
            tb = time.perf_counter()
            t8 = s8.get_timings()
        disk = {"workload": "524,288-particle self-gravitating disk, theta=0.75, eps=1e-3, dt=1e-4 "
                            "(the reference's ~27 ms/frame configuration)",
                "ms_per_step": round((tb - ta) * 1e3 / 50, 4),
                "traversal_ms": round(t8["ms_traverse"] / 50, 4),
                "build_ms": round((t8["ms_integrate"] + t8["ms_sort"] + t8["ms_build"] + t8["ms_com"]) / 50, 4),
                "interactions_per_step": round(t8["interactions"] / 50),
                "traversal_tflops": round(FLOP_PER_INTERACTION * t8["interactions"] / (t8["ms_traverse"] * 1e-3) / 1e12, 2)}
    slow = None
    if world == 1 and not args.no_alt_bits and not args.walk_only and args.morton_bits == 21:
        # A system whose particles move less than the distance to their neighbours per step (cold
        # Plummer sphere, total mass 1, the reference's dt = 1e-6), stepped the way a viewer steps
        # it -- one update, then a wait -- so that every sort knows how the previous one went:
        # the sort of nearly sorted keys against the digit passes over all elements.
        ns = 2_097_152
        ics = nb.sample_ic(nb.IC_PLUMMER, ns, 42, p0=0.1, p1=1.0)
        res = {}
        for name, fl in (("digit_passes_over_all", nb.FLAG_NO_COHERENT_SORT), ("nearly_sorted_path", 0)):
            ps_ = nb.Params(num_points=ns, max_num_codes_per_leaf=32, traversal_group_size=group, theta=THETA,
                            dt=1e-6, gravity=1.0, softening_factor=1e-3, domain_size=5.0, morton_bits=21,
                            device_id=local_rank, flags=nb.FLAG_TIMING | nb.FLAG_TIMING_STAGES | fl)
            with nb.Simulation(ps_) as s9:
                s9.set_state(ics["x"], ics["y"], ics["z"], ics["m"], ics["vx"], ics["vy"], ics["vz"])
                for _ in range(4):
                    s9.update(); s9.sync()
                s9.get_timings()
                for _ in range(20):
                    s9.update(); s9.sync()
                t9 = s9.get_timings()
                st9 = s9.sort_stats()
            res[name] = {"sort_and_permute_ms": round(t9["ms_sort"] / 20, 4),
                         "build_ms": round((t9["ms_integrate"] + t9["ms_sort"] + t9["ms_build"] + t9["ms_com"]) / 20, 4),
                         "sorts": {"nearly_sorted": st9["coherent_sorts"], "all_elements": st9["full_sorts"]},
                         "moved_elements_last_step": st9["last_moved"]}
        slow = {"workload": "2,097,152-particle Plummer sphere at rest (total mass 1), theta=%.2f, eps=1e-3, dt=1e-6, "
                            "63-bit keys, one update then a wait per step" % THETA, **res,
                "note": "the library takes the shorter sort when the previous sort (finished by then) moved at most "
                        "1 element in 64, or when the last three counts it has seen moved at most 1 in 1024; the "
                        "sphere of this line's timed region starts from rest and accelerates (none moved in step 1, "
                        "a quarter of all by step 130), so its sorts are digit passes over all elements"}
    direct = None
    if world == 1 and not args.no_direct_sum and not args.walk_only:
        direct = direct_sum_block(args.no_cpu_baseline)
    traffic, traffic_src = None, None
    for name in ("r3_traffic.json", "r2_traffic.json", "r1_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                traffic = json.load(f)["k_traverse"]["bytes"]
            traffic_src = "profiles/" + name
            break
        except Exception:
            pass

    if rank == 0:
        K = max(args.steps, 1)
        ms_step = elapsed * 1e3 / K
        trav_ms = tim["ms_traverse"] / K
        build_ms = (tim["ms_integrate"] + tim["ms_sort"] + tim["ms_build"] + tim["ms_com"]) / K
        inter_step_rank0 = tim["interactions"] / K
        tflops = FLOP_PER_INTERACTION * inter_step_rank0 / (trav_ms * 1e-3) / 1e12 if trav_ms else 0.0
        build_bytes = BUILD_BYTES_PER_PARTICLE[args.morton_bits]
        build_gbs = build_bytes * n_here / (build_ms * 1e-3) / 1e9 if build_ms else 0.0
        out = {
            "metric": "ms/step (build+traversal) at 2M particles theta=0.6",
            "value": round(ms_step, 4),
            "unit": "ms/step",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4),
            "higher_is_better": False,
            "scaling": "weak",
            "vs_baseline": round(ms_step / REFERENCE_MS_PER_STEP, 5) if world == 1 else None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": ("cold uniform sphere r=0.5 in domain 5.0 (SPAWN_SPHERE), "
                                    if args.ic == "sphere" else
                                    "cold Plummer sphere a=0.1 truncated at r=1 in domain 5.0, total mass 1, ") +
                                   f"{args.particles_per_gpu} particles/GPU, theta=0.6, eps={eps:g}, "
                                   "dt=1e-6, 32 codes/leaf, group=64",
                       "particles": n, "morton_bits": args.morton_bits,
                       "parallelism": "single" if world == 1 else
                       (f"{world} Morton-key ranges of equal particle count (all-gathered key "
                        "histograms), own tree per GPU, all-gather of top-of-tree nodes, "
                        "locally-essential-tree exchange (all-to-all of pruned tree copies, grafted "
                        "behind the own tree and walked with it in one traversal)" if let else
                        f"replicated build, traversal sharded over {world} Morton ranges, "
                        "all-gather of accelerations"),
                       "tree": {"unique_codes": U, "leaves": M, "octree_nodes": n_oct}},
            "build_ms": round(build_ms, 4),
            "traversal_ms": round(trav_ms, 4),
            "stage_ms": stage,
            "stage_ms_note": "10 steps of a separate run with one HIP event per stage (whole problem "
                             "on this rank); build_ms / traversal_ms above are from the timed region",
            "interactions_per_step": round(total_inter / K),
            "interactions_per_s": round(total_inter / elapsed),
            "finite": ok,
            "baseline_note": "vs_baseline = ms/step / 150 ms (reference README, RTX 500 Ada "
                             "laptop GPU; different hardware)",
            "roofline": {"bound": "valu", "kernel": "k_traverse", "achieved": round(tflops, 2),
                         "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(tflops / PEAK_FP32_TFLOPS, 4), "traffic": traffic,
                         "traffic_note": "HBM-side bytes per launch from the committed rocprofv3 PMC "
                                         f"passes ({traffic_src}; steady-state dispatches only), not "
                                         "measured in this run",
                         "pipe": "FP32 vector ALU (v_pk_fma_f32 / v_rsq_f32); priced against the dense "
                                 "f32 peak, which is the same 157.3 TFLOP/s for MFMA and packed VALU",
                         "note": "20 FLOP x exact interaction count / mean kernel time (HIP events on "
                                 "the library stream)",
                         "issue_bound": {"cycles_per_128_interactions": 64,
                                         "peak_interactions_per_s": round(1024 * 2.4e9 * 2),
                                         "achieved_interactions_per_s": round(inter_step_rank0 / (trav_ms * 1e-3)) if trav_ms else None,
                                         "frac": round(inter_step_rank0 / (trav_ms * 1e-3) / (1024 * 2.4e9 * 2), 4) if trav_ms else None,
                                         "sustained_clock_ghz": SUSTAINED_CLOCK_GHZ,
                                         "frac_at_sustained_clock": round(inter_step_rank0 / (trav_ms * 1e-3) / (1024 * SUSTAINED_CLOCK_GHZ * 1e9 * 2), 4) if trav_ms else None,
                                         "note": "12 packed ops (4 cycles) + 2 v_rsq_f32 (8 cycles) per wave "
                                                 "per two sources: the instruction-issue floor of the force "
                                                 "arithmetic on 1024 SIMDs, at the nominal 2.4 GHz and at the "
                                                 "clock measured inside this kernel (s_memtime against "
                                                 "s_memrealtime, profiles/r3_walk_profile.txt; another box "
                                                 "measured 2.0 GHz)"}},
            "roofline_build": {"bound": "hbm", "achieved": round(build_gbs, 1), "peak": PEAK_HBM_GBS,
                               "unit": "GB/s", "frac": round(build_gbs / PEAK_HBM_GBS, 4),
                               "bytes_per_particle": build_bytes,
                               "note": f"{build_bytes} algorithmic B/particle over integrate+sort+permute+"
                                       f"build+COM with the {3 * args.morton_bits}-bit keys this run used "
                                       f"({BUILD_BYTES_PER_PARTICLE[10]} with 30-bit keys, "
                                       f"{BUILD_BYTES_PER_PARTICLE[21]} with 63-bit keys); 2.6 TB/s at 16.7 M "
                                       "particles. Only kick-drift and permute stream at the chip's rate; a digit "
                                       "pass is a chain per tile (ticket, load, rank, look-back, write) with two "
                                       "workgroups per CU, no unit busy more than a third of the time "
                                       "(profiles/r3_sort_pass_experiments.txt; launches of many tiles number them by "
                                       "workgroup index with helping instead of a same-address ticket atomic, which "
                                       "completes 15 ns apart); the waves of the tree and "
                                       "centre-of-mass kernels wait 60-85 % of their cycles on dependent loads "
                                       "(profiles/r3_build_kernels_pmc_2M_16M.txt)"},
        }
        if world > 1:
            out["rccl_world"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                                 "collectives_on": coll_dev}
            peak_i = 1024 * SUSTAINED_CLOCK_GHZ * 1e9 * 2
            ranks = []
            for pr in per_rank:
                tf = FLOP_PER_INTERACTION * pr["interactions_per_step"] / (pr["traversal_ms"] * 1e-3) / 1e12 if pr["traversal_ms"] else 0.0
                bg = build_bytes * pr["particles"] / (pr["build_ms"] * 1e-3) / 1e9 if pr["build_ms"] else 0.0
                ranks.append({"rank": pr["rank"], "particles": pr["particles"], "build_ms": round(pr["build_ms"], 4),
                              "traversal_ms": round(pr["traversal_ms"], 4),
                              "interactions_per_step": round(pr["interactions_per_step"]),
                              "roofline": {"achieved": round(tf, 2), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / PEAK_FP32_TFLOPS, 4),
                                           "frac_of_issue_bound_at_sustained_clock": round(pr["interactions_per_step"] / (pr["traversal_ms"] * 1e-3) / peak_i, 4) if pr["traversal_ms"] else None},
                              "roofline_build": {"achieved": round(bg, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(bg / PEAK_HBM_GBS, 4)}})
            out["per_rank"] = ranks
            if let and all("phases" in pr for pr in per_rank):
                keys = list(per_rank[0]["phases"].keys())
                out["phases_over_ranks"] = {k: {"min": round(min(pr["phases"][k] for pr in per_rank), 4),
                                                "mean": round(sum(pr["phases"][k] for pr in per_rank) / world, 4),
                                                "max": round(max(pr["phases"][k] for pr in per_rank), 4)} for k in keys}
                out["phases_note"] = ("mean of 3 steps after the timed region, one HIP event between phases: ms of GPU "
                                      "stream time per phase (exchange includes waiting for the slowest peer), "
                                      "ms_host_wait = host blocked on the sizes of the copies, bytes per rank and step")
                out["collective_calls_rank0"] = per_rank[0].get("collective_calls")
        if let:
            out["roofline"]["note"] += ("; rank 0's traversal launches (own tree and imported copies in one "
                                        "walk), interactions of rank 0")
            out["roofline"]["traffic"] = None
            st = let_stats
            out["let_records_per_step"] = {"sent_rank0": st["records_sent"], "received_rank0": st["records_received"],
                                           "nodes_sent_rank0": st["nodes_sent"], "nodes_received_rank0": st["nodes_received"]}
            out["domain_rank0"] = {"particles": st["n_local"], "first_key": st["splitters"][0],
                                   "end_key": st["splitters"][1]}
        if accuracy is not None:
            out["accuracy"] = accuracy
        if evolved is not None:
            out["after_collapse"] = evolved
        if alt is not None:
            out["other_key_width"] = alt
        if fit is not None:
            out["fitted_30bit_keys"] = fit
        if disk is not None:
            out["disk_524k"] = disk
        if slow is not None:
            out["slow_system_sort"] = slow
        if direct is not None:
            out["direct_sum"] = direct
        if g32 is not None:
            out["other_group_size"] = g32
        if f64 is not None:
            out["other_precision"] = f64
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ic, args.morton_bits, steps=5, eps=eps)
        emit(out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
