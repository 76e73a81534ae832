#!/bin/bash
# usage: bash tools/ab_build.sh <tag> "<extra hipcc flags, e.g. -DTDK_EXP_SETPRIO=1>"
# Builds an experimental copy of the library as tadataka_amd/lib/libtadataka_hip_<tag>.so (A/B runs:
# TDK_LIBRARY=$PWD/tadataka_amd/lib/libtadataka_hip_<tag>.so python tools/kbench.py).  Only dvo.hip is recompiled with the extra flags.
set -e
TAG=$1; FLAGS=$2
cd "$(dirname "$0")/../tadataka_amd/csrc"
make -s -j4
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $FLAGS -c dvo.hip -o ../lib/obj/dvo_$TAG.o
OBJS=$(ls ../lib/obj/*.o | grep -v "dvo" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libtadataka_hip_$TAG.so $OBJS ../lib/obj/dvo_$TAG.o -ldl
rm -f ../lib/obj/dvo_$TAG.o
echo built libtadataka_hip_$TAG.so
