#!/bin/bash
# usage: scratch/build_sortbench.sh <variant tag>  -> scratch/variants/sortbench_<tag> (runs against libnbody_<tag>.so)
set -e
tag=$1
cd $(dirname $0)/..
/opt/rocm/bin/hipcc -O2 -std=c++17 -w --offload-arch=gfx950 scratch/sortbench.cpp -Lscratch/variants -lnbody_$tag -Wl,-rpath,'$ORIGIN' -o scratch/variants/sortbench_$tag
