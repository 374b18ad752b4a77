#!/bin/bash
set -e
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_quick
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o q -- python3 $ROOT/scratch/prof_step.py 21 2097152 20 ${1:-64} > $OUT/run.log 2>&1
cd $ROOT
f=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.15:
        print(r["Name"][:60].ljust(60), r["Calls"].rjust(5), "avg us %9.1f" % (float(r["AverageNs"]) / 1e3), "%6.2f%%" % float(r["Percentage"]))
PY
