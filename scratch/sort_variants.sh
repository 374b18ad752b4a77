#!/bin/bash
# Runs on the GPU box: the sort microbenchmark for every library variant under scratch/variants.
# (ablation variants give wrong results on purpose: their status is ignored)
for lib in scratch/variants/libnbody_*.so; do
  tag=$(basename $lib .so); tag=${tag#libnbody_}
  [ -x scratch/variants/sortbench_$tag ] || continue
  echo "== $tag"
  for cfg in "2097152 21 2" "16777216 21 2" "2097152 10 1" "524288 21 2"; do
    timeout -k 5 120 scratch/variants/sortbench_$tag $cfg; rc=$?
    if [ $rc -ne 0 ] && [ $rc -ne 2 ]; then echo "FAILED $tag $cfg rc=$rc"; exit 1; fi
  done
done
