#!/bin/bash
# Kernel-level times of the semi-dense and bundle-adjustment paths (run via gpurun):
#   bash profiles/run_other_kernels.sh r01
set -u
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/other_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o other -- python $ROOT/tools/other_kernels.py > "$OUT/stdout.log" 2>&1
cd "$ROOT"
python "$ROOT/profiles/summarize_other.py" "$OUT" | tee "$OUT/summary.txt"
