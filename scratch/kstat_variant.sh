#!/bin/bash
# usage (GPU box): scratch/kstat_variant.sh <lib or ""> <tag>  -- rocprofv3 kernel stats of 6 steps at 2M
export TMPDIR=/tmp
ROOT=$PWD
[ -n "$1" ] && export NBODY_AMD_LIB=$ROOT/$1
OUT=$ROOT/gpurun_out/ks_$2
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- python3 $ROOT/scratch/prof_step.py 21 2097152 20 > $OUT.log 2>&1 || exit 1
cd $ROOT
cp $(find $OUT -name "*kernel_stats.csv" | head -1) gpurun_out/ks_$2.csv
rm -rf $OUT
