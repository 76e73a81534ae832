#!/bin/bash
# per-kernel time of the default bench (run on the GPU box)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/ktrace_${1:-x}; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
shift
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 2 "$@" > $OUT/stdout.log 2>&1
cd $ROOT
python profiles/summarize.py $OUT 2>/dev/null | head -16
grep -o '"ms_per_step": [0-9.]*' $OUT/stdout.log
