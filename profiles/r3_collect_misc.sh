#!/bin/bash
# Runs on the GPU box (via gpurun): the clock the chip sustains inside the traversal kernel
# (profiling build of the library), kernel statistics and SQ counters of the direct sum over
# 1,048,576 bodies, SQ counters of the double-precision traversal.  Each rocprofv3 run is its own
# process with the program directly after `--`.
set -e
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_r3_misc
rm -rf $OUT; mkdir -p $OUT
NBODY_AMD_LIB=$ROOT/scratch/prof/libnbody_prof.so timeout -k 10 300 python3 scratch/prof_step.py 21 2097152 400 > $OUT/walk.out 2> $OUT/walk.err
tail -5 $OUT/walk.err > $OUT/walk_profile.txt
cd /tmp
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_direct -o direct -- python3 $ROOT/bench.py --workload direct --steps 3 --warmup 1 --no-cpu-baseline > $OUT/direct_under_rocprof.log 2>&1
timeout -s KILL 240 rocprofv3 --output-format csv --pmc $SQ1 -d $OUT/sq_direct -- python3 $ROOT/bench.py --workload direct --steps 2 --warmup 1 --no-cpu-baseline > $OUT/sq_direct.log 2>&1
SQ64="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY"
timeout -s KILL 240 rocprofv3 --output-format csv --pmc $SQ64 -d $OUT/sq_f64 -- python3 $ROOT/scratch/prof_step.py 21 2097152 4 64 f64 > $OUT/sq_f64.log 2>&1
cd $ROOT
cp "$(find $OUT/trace_direct -name '*kernel_stats.csv' | head -1)" $OUT/direct_1M_kernel_stats.csv
python3 scratch/pmc_summary.py $OUT/sq_direct k_direct > $OUT/k_direct_pmc.txt
python3 scratch/pmc_summary.py $OUT/sq_f64 k64_traverse --steady > $OUT/k64_traverse_pmc.txt
rm -rf $OUT/trace_direct $OUT/sq_direct $OUT/sq_f64
ls -la $OUT; cat $OUT/walk_profile.txt
