#!/bin/bash
# usage: bash tools/ab_build_sd.sh <tag> "<extra hipcc flags>"  -- like ab_build.sh, for semi_dense.hip
set -e
TAG=$1; FLAGS=$2
cd "$(dirname "$0")/../tadataka_amd/csrc"
make -s -j4
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off $FLAGS -c semi_dense.hip -o ../lib/obj/semi_dense_$TAG.o
OBJS=$(ls ../lib/obj/*.o | grep -v "semi_dense" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libtadataka_hip_$TAG.so $OBJS ../lib/obj/semi_dense_$TAG.o -ldl
rm -f ../lib/obj/semi_dense_$TAG.o
echo built libtadataka_hip_$TAG.so
