#!/bin/bash
# builds scratch/prof/libnbody_prof.so = the library with -DNB_WALK_PROFILE=1
set -e
cd "$(dirname "$0")/../nbody_amd/csrc"
mkdir -p ../../scratch/prof
for f in api k_points k_sort k_tree k_traverse k_let k_fp64 k_reduce dd; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-result -DNB_WALK_PROFILE=1 -c $f.hip -o ../../scratch/prof/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../scratch/prof/libnbody_prof.so ../../scratch/prof/*.o
