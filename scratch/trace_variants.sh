#!/bin/bash
# Runs on the GPU box: scratch/trace_steps.sh for every library variant under scratch/variants
# (NBODY_AMD_LIB selects the library the Python binding loads), selected kernels only.
for lib in scratch/variants/libnbody_*.so; do
  tag=$(basename $lib .so); tag=${tag#libnbody_}
  export NBODY_AMD_LIB=$PWD/$lib
  echo "#### $tag"
  bash scratch/trace_steps.sh v_$tag 2>&1 | grep -E "^==|${1:-k_}" || exit 1
done
