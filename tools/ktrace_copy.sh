#!/bin/bash
# usage: bash tools/ktrace_copy.sh <tag> <command...>   -> kernel + memory-copy trace (rocprofv3, no counters)
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/ktc_$TAG; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o k -- "$@" > $OUT/run.log 2>&1
cd $ROOT
ls $OUT
