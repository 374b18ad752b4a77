#!/bin/bash
# kernel-trace stats and VALU counters of a 5-step double-precision run (2M sphere, theta 0.6)
set -e
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_fp64
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o f64 -- python3 $ROOT/scratch/prof_step.py 21 2097152 5 64 f64 > $OUT/run.log 2>&1
timeout -s KILL 120 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $OUT/pmc_sq -- python3 $ROOT/scratch/prof_step.py 21 2097152 3 64 f64 > $OUT/pmc.log 2>&1
cd $ROOT
cp "$(find $OUT/trace -name '*kernel_stats.csv' | head -1)" $OUT/fp64_2M_theta0.6_kernel_stats.csv
python3 scratch/pmc_summary.py $OUT/pmc_sq k64_traverse > $OUT/k64_traverse_pmc.txt
rm -rf $OUT/trace $OUT/pmc_sq
cat $OUT/fp64_2M_theta0.6_kernel_stats.csv | head -12; cat $OUT/k64_traverse_pmc.txt
