#!/bin/bash
# Kernel-level times of the semi-dense session and the bundle-adjustment handle (run via gpurun):
#   bash profiles/run_workloads.sh r02 [sd|ba|all]
set -u
TAG=${1:-r02}
WHAT=${2:-all}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/workloads_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o wl -- python $ROOT/tools/profile_workloads.py $WHAT > "$OUT/stdout.log" 2>&1
cd "$ROOT"
f=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
echo "== $f"
python - "$f" <<'PY' | tee "$OUT/summary.txt"
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("%-60s %6s %10s %8s" % ("kernel", "calls", "avg_us", "pct"))
for r in rows:
    name = r["Name"].split("(")[0][:60]
    print("%-60s %6s %10.1f %8s" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
tail -3 "$OUT/stdout.log"
