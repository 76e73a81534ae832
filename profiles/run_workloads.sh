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
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("%-48s %7s %10s %10s %10s %7s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "pct"))
for r in rows:
    m = re.search(r'(k_\w+(<[^>]*>)?)', r["Name"])
    name = m.group(1) if m else r["Name"][:48]
    print("%-48s %7s %10.1f %10.1f %10.1f %7s" % (name, r["Calls"], float(r["AverageNs"]) / 1e3,
          float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
PY
tail -3 "$OUT/stdout.log"
