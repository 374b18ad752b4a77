"""The benchmark's output contract: one JSON line on stdout, nothing else, with the fields the
driver reads (metric, value, unit, n_gpus, steps, warmup, ms_per_step, higher_is_better, scaling,
vs_baseline, dtype, data, config.workload) and the roofline / cpu_baseline objects.  Run on a
small particle count so that it takes seconds."""
import json
import os
import socket
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
           "--particles-per-gpu", "131072", "--no-alt-bits", "--evolve-steps", "0", "--no-direct-sum", *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[:2000]          # library chatter must not reach stdout
    return json.loads(lines[0])


def test_bench_line_has_the_contract_fields():
    d = run_bench()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2
    assert d["unit"] == "ms/step" and d["higher_is_better"] is False and d["dtype"] == "f32"
    assert d["value"] == d["ms_per_step"] > 0 and d["finite"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "valu" and r["unit"] == "TFLOP/s" and r["kernel"] == "k_traverse"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1
    b = d["roofline_build"]
    assert b["bound"] == "hbm" and b["bytes_per_particle"] == 360 and 0 < b["frac"] < 1
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > d["value"] and c["unit"] == "ms/step"
    a = d["accuracy"]
    assert a["bodies_sampled"] == 4096 and 0 < a["mean_rel_acc_error"] < 0.1
    # the traversal cannot take longer than the step it is part of
    assert d["traversal_ms"] <= d["ms_per_step"] and d["build_ms"] > 0


def test_bench_direct_sum_workload_line():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "direct", "--steps", "1", "--warmup", "1",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[:2000]
    d = json.loads(lines[0])
    assert d["unit"] == "interactions/s" and d["higher_is_better"] is True and d["finite"] is True
    assert d["roofline"]["bound"] == "valu" and 0.3 < d["roofline"]["frac"] < 0.7


def _torchrun_bench(world, env_extra=None):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", str(world), "--steps", "3", "--warmup", "1", "--particles-per-gpu", "65536",
           "--rehearse-on-one-gpu", "--no-cpu-baseline"]
    env = dict(os.environ, **(env_extra or {}))
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    return r, time.time() - t0


def test_multi_rank_line_names_the_world_and_splits_the_step_per_rank():
    """Two ranks on this GPU over gloo (the numbers mean nothing): the line says which world and
    backend ran and carries per-rank step splits and the phases of the domain-decomposed step."""
    r, _ = _torchrun_bench(2)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[:2000]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["finite"] is True and d["value"] > 0
    assert d["rccl_world"]["world_size"] == 2 and d["rccl_world"]["backend"] == "gloo"
    assert [p["rank"] for p in d["per_rank"]] == [0, 1]
    for p in d["per_rank"]:
        assert p["particles"] > 60000 and p["build_ms"] > 0 and p["traversal_ms"] > 0
        assert 0 < p["roofline"]["frac"] < 1 and 0 < p["roofline_build"]["frac"] < 1
    ph = d["phases_over_ranks"]
    for k in ("ms_build", "ms_frontier", "ms_export", "ms_early_traversal", "ms_sizing", "ms_exchange", "ms_graft",
              "ms_traversal", "ms_total", "ms_host_wait", "ms_host_total", "bytes_frontier", "bytes_sent", "bytes_received"):
        assert ph[k]["min"] <= ph[k]["mean"] <= ph[k]["max"], k
    assert ph["ms_build"]["min"] > 0 and ph["ms_traversal"]["min"] > 0 and ph["bytes_sent"]["min"] > 0
    assert ph["ms_total"]["max"] >= ph["ms_traversal"]["max"]
    assert d["let_records_per_step"]["nodes_sent_rank0"] > 0


def test_failed_collective_ends_every_rank_with_the_message():
    """Rank 1's fifth all-to-all reports failure: both ranks leave with a non-zero status within
    seconds, the failing one with the library's and the callback's message; nothing hangs."""
    r, took = _torchrun_bench(2, {"NBODY_BENCH_FAIL_ALL_TO_ALL": "1:4"})
    assert r.returncode != 0
    assert took < 120, took
    assert "collective failed" in r.stderr and "injected failure of all_to_all call 4 on rank 1" in r.stderr, r.stderr[-3000:]
    assert not [l for l in r.stdout.splitlines() if l.strip().startswith("{")]   # no result line
