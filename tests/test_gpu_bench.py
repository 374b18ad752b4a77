"""The benchmark's output contract: one JSON line on stdout, nothing else, with the fields the
driver reads (metric, value, unit, n_gpus, steps, warmup, ms_per_step, higher_is_better, scaling,
vs_baseline, dtype, data, config.workload) and the roofline / cpu_baseline objects.  Run on a
small particle count so that it takes seconds."""
import json
import os
import subprocess
import sys

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
