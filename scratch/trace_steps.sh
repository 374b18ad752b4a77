#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace statistics of a few steps of the 2M sphere, the
# 16.7M Plummer sphere and the 524k disk -> gpurun_out/trace_<tag>/{2M,16M,disk}_kernel_stats.csv
set -e
export TMPDIR=/tmp
ROOT=$PWD
TAG=${1:-cur}
OUT=$ROOT/gpurun_out/trace_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
for cfg in "2097152 sphere 10 2M" "16777216 plummer 3 16M" "524288 disk 20 disk"; do
  set -- $cfg
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_$4 -o step -- python3 $ROOT/scratch/prof_step.py 21 $1 $3 64 f32 $2 > $OUT/t_$4.log 2>&1
  cp "$(find $OUT/t_$4 -name '*kernel_stats.csv' | head -1)" $OUT/$4_kernel_stats.csv
  rm -rf $OUT/t_$4
done
cd $ROOT
python3 - $OUT <<'PY'
import csv, sys
out = sys.argv[1]
for tag in ("2M", "16M", "disk"):
    rows = list(csv.DictReader(open(f"{out}/{tag}_kernel_stats.csv")))
    print("==", tag)
    tot = 0.0
    for r in rows:
        n = r["Name"].split("(")[0].replace("void nb::", "").replace("nb::", "")
        if "traverse" in n or "rocclr_copy" in n or "permute<true>" in n or "k_morton" in n: continue
        a = float(r["AverageNs"]) / 1e3
        print(f"  {n[:44]:44s} calls {r['Calls']:>4s} avg {a:8.1f} us")
PY
