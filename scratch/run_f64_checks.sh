#!/bin/bash
set -o pipefail
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 || { tail -5 gpurun_out/smoke.log; exit 1; }
tail -1 gpurun_out/smoke.log
timeout -k 10 300 python -u -m pytest tests/test_gpu_parallel.py -m gpu -x -q -k cpp --timeout 120 --timeout-method thread > gpurun_out/cpp.log 2>&1 || { tail -20 gpurun_out/cpp.log; exit 1; }
tail -1 gpurun_out/cpp.log
timeout -k 10 600 python bench.py --evolve-steps 0 > gpurun_out/bench_f64.log 2>&1 || { tail -5 gpurun_out/bench_f64.log; exit 1; }
tail -1 gpurun_out/bench_f64.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['traversal_ms'], d.get('other_precision'))"
