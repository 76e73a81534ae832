#!/bin/bash
# usage: bash tools/ktrace.sh <tag> <command...>   -> per-kernel stats of the command (rocprofv3 --kernel-trace --stats)
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/ktrace_$TAG; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- "$@" > $OUT/run.log 2>&1
cd $ROOT
python - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:12]:
        print(f"{r['Name'][:70]:70s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:9.1f} us total {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['Percentage']):5.1f}%")
PY
