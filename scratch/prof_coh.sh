#!/bin/bash
# GPU box: rocprofv3 kernel statistics of a few steps with the nearly-sorted path forced
set -e
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/coh_prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o step -- python3 $ROOT/scratch/prof_step.py 21 ${1:-16777216} ${2:-4} 64 f32 ${3:-plummer} > $OUT/log.txt 2>&1
cp "$(find $OUT/t -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats.csv
rm -rf $OUT/t
cd $ROOT
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/coh_prof/kernel_stats.csv")))
for r in rows:
    n = r["Name"].split("(")[0].replace("void nb::", "").replace("nb::", "")
    if "traverse" in n: continue
    print(f"  {n[:56]:56s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
