#!/bin/bash
# Runs on the GPU box (via gpurun): SQ counters of the build kernels (digit pass, unique,
# centre of mass, ...) at 2,097,152 and 16,777,216 particles, and kernel-trace statistics of the
# other workloads.  Each rocprofv3 run is its own process with the program directly after `--`.
set -e
export TMPDIR=/tmp
ROOT=$PWD
TAG=${1:-base}
OUT=$ROOT/gpurun_out/prof_r3_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE"
SQ2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"
for cfg in "2097152 sphere 5 2M" "16777216 plummer 3 16M"; do
  set -- $cfg
  timeout -s KILL 240 rocprofv3 --output-format csv --pmc $SQ1 -d $OUT/sq1_$4 -- python3 $ROOT/scratch/prof_step.py 21 $1 $3 64 f32 $2 > $OUT/sq1_$4.log 2>&1
  timeout -s KILL 240 rocprofv3 --output-format csv --pmc $SQ2 -d $OUT/sq2_$4 -- python3 $ROOT/scratch/prof_step.py 21 $1 $3 64 f32 $2 > $OUT/sq2_$4.log 2>&1 || echo "sq2 pass failed for $4"
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$4 -o step -- python3 $ROOT/scratch/prof_step.py 21 $1 $3 64 f32 $2 > $OUT/trace_$4.log 2>&1
done
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_disk -o step -- python3 $ROOT/scratch/prof_step.py 21 524288 20 64 f32 disk > $OUT/trace_disk.log 2>&1
cd $ROOT
for k in k_sort_pass k_sort_hist k_unique k_com_t k_leaves k_radix k_depth k_emit k_permute k_integrate k_tiefix; do
  for s in 2M 16M; do
    echo "== $k, $s particles (per dispatch, steady-state steps)" >> $OUT/build_kernels_pmc.txt
    python3 scratch/pmc_summary.py $OUT/sq1_$s $k --steady >> $OUT/build_kernels_pmc.txt || true
    python3 scratch/pmc_summary.py $OUT/sq2_$s $k --steady >> $OUT/build_kernels_pmc.txt || true
  done
done
for s in 2M 16M disk; do
  f=$(find $OUT/trace_$s -name "*kernel_stats.csv" | head -1)
  cp "$f" $OUT/step_${s}_kernel_stats.csv
done
rm -rf $OUT/sq1_* $OUT/sq2_* $OUT/trace_2M $OUT/trace_16M $OUT/trace_disk
ls -la $OUT
