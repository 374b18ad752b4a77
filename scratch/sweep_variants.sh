#!/bin/bash
# runs scratch/stage_times.py for every library variant under scratch/variants
for lib in scratch/variants/libnbody_*.so; do
  for cfg in "2097152 21 sphere" "524288 21 disk"; do
    NBODY_AMD_LIB=$PWD/$lib timeout -k 10 120 python scratch/stage_times.py $cfg || exit 1
  done
done
