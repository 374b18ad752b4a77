#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of the default bench command and PMC
# passes of a 6-step run; leaves summaries under gpurun_out/prof_r3/.  Each rocprofv3 run is
# its own process with the program directly after `--`.
set -e
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_r3
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace21 -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --no-alt-bits --evolve-steps 0 --no-direct-sum > $OUT/bench_bits21_under_rocprof.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace10 -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --no-alt-bits --evolve-steps 0 --no-direct-sum --morton-bits 10 > $OUT/bench_bits10_under_rocprof.log 2>&1
timeout -s KILL 120 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -- python3 $ROOT/scratch/prof_step.py 21 2097152 6 > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -- python3 $ROOT/scratch/prof_step.py 21 2097152 6 > $OUT/pmc_write.log 2>&1
timeout -s KILL 120 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $OUT/pmc_sq1 -- python3 $ROOT/scratch/prof_step.py 21 2097152 6 > $OUT/pmc_sq1.log 2>&1
cd $ROOT
for b in 21 10; do
  f=$(find $OUT/trace$b -name "*kernel_stats.csv" | head -1)
  cp "$f" $OUT/bench_2M_theta0.6_bits${b}_kernel_stats.csv
done
python3 scratch/make_traffic.py $OUT/pmc_fetch $OUT/pmc_write r3 > $OUT/traffic.json
python3 scratch/pmc_summary.py $OUT/pmc_sq1 k_traverse --steady > $OUT/k_traverse_pmc_bits21.txt
rm -rf $OUT/trace21 $OUT/trace10 $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq1
ls -la $OUT
