#!/bin/bash
# GPU box: SQ counters of the traversal kernel in walk-only mode (tree walk and list bookkeeping
# without the force arithmetic) -> gpurun_out/pmc_walk.txt
set -e
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/pmc_walk
rm -rf $OUT; mkdir -p $OUT
cd /tmp
NBODY_EXTRA_FLAGS=256 timeout -s KILL 120 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $OUT/sq -- python3 $ROOT/scratch/prof_step.py 21 2097152 6 > $OUT/log.txt 2>&1
cd $ROOT
python3 scratch/pmc_summary.py $OUT/sq k_traverse --steady > gpurun_out/pmc_walk.txt
rm -rf $OUT
