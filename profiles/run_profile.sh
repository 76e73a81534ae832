#!/bin/bash
# Profiles bench.py's default workload on the GPU box.  Usage (via gpurun):
#   bash profiles/run_profile.sh <tag>        e.g. r01
# Writes raw rocprofv3 output under gpurun_out/prof_<tag>/ (scratch); the
# summaries worth keeping are copied into profiles/ by hand afterwards.
set -u
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-workloads --no-solo-pass --no-traffic-pass --min-seconds 0.2 --steps 10 --warmup 2"
cd /tmp
# 1) per-kernel time: (a) as the headline runs -- two batches in flight, one batch's pyramid beside the other's
#    estimation, so every kernel but the full-resolution evaluations is stretched ("under overlap") --
#    and (b) --single-buffer: one batch, every kernel alone on the device ("alone")
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- $BENCH > "$OUT/trace_stdout.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_single" -o bench -- $BENCH --single-buffer > "$OUT/trace_single_stdout.log" 2>&1
# 2) HBM traffic, separate passes (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2)
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o bench -- $BENCH > "$OUT/pmc_fetch_stdout.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o bench -- $BENCH > "$OUT/pmc_write_stdout.log" 2>&1
# 3) issue-side picture
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d "$OUT/pmc_sq" -o bench -- $BENCH > "$OUT/pmc_sq_stdout.log" 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d "$OUT/pmc_sq2" -o bench -- $BENCH > "$OUT/pmc_sq2_stdout.log" 2>&1
cd "$ROOT"
find "$OUT" -name "*.csv" | head -50
python "$ROOT/profiles/summarize.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
