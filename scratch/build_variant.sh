#!/bin/bash
# usage: scratch/build_variant.sh <tag> <extra hipcc flags...>   -> scratch/variants/libnbody_<tag>.so
set -e
tag=$1; shift
ROOT=$(cd $(dirname $0)/.. && pwd)
B=/tmp/nbvar_$tag; rm -rf $B; mkdir -p $B
cd $ROOT/nbody_amd/csrc
for f in api k_points k_sort k_tree k_traverse k_let k_fp64 k_reduce dd; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-result "$@" -c $f.hip -o $B/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $ROOT/scratch/variants/libnbody_$tag.so $B/*.o
echo built $tag
