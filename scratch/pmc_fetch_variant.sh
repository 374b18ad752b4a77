#!/bin/bash
# usage (on the GPU box): scratch/pmc_fetch_variant.sh <lib path or ""> <outdir>  -- FETCH_SIZE of k_traverse, steady state
export TMPDIR=/tmp
ROOT=$PWD
[ -n "$1" ] && export NBODY_AMD_LIB=$ROOT/$1
OUT=$ROOT/gpurun_out/$2
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -s KILL 120 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT -- python3 $ROOT/scratch/prof_step.py 21 2097152 6 > $OUT.log 2>&1 || exit 1
cd $ROOT
python3 scratch/pmc_summary.py $OUT k_traverse --steady
rm -rf $OUT
