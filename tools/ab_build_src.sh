#!/bin/bash
# usage: bash tools/ab_build_src.sh <tag> <source.hip> "<extra hipcc flags>"
# As tools/ab_build.sh, for any one translation unit (flags of the Makefile for that file are kept).
set -e
TAG=$1; SRC=$2; FLAGS=$3
cd "$(dirname "$0")/../tadataka_amd/csrc"
make -s -j4
BASE=$(basename $SRC .hip)
EXTRA=""
case $BASE in granular|semi_dense|pyramid) EXTRA="-ffp-contract=off";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $EXTRA $FLAGS -c $SRC -o ../lib/obj/${BASE}_$TAG.o
OBJS=$(ls ../lib/obj/*.o | grep -v "/${BASE}\(_[a-z0-9]*\)\?\.o" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libtadataka_hip_$TAG.so $OBJS ../lib/obj/${BASE}_$TAG.o -ldl
rm -f ../lib/obj/${BASE}_$TAG.o
echo built libtadataka_hip_$TAG.so
