#!/bin/bash
# usage: bash tools/pmc_kernel.sh <tag> <kernel-substring> <command...>
# PMC passes (separate, --pmc only) of one kernel of a command; prints per-launch averages.
TAG=$1; KERN=$2; shift 2
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/a -o k -- "$@" > $OUT/a.log 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS --output-format csv -d $OUT/b -o k -- "$@" > $OUT/b.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c -o k -- "$@" > $OUT/c.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/d -o k -- "$@" > $OUT/d.log 2>&1
timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum --output-format csv -d $OUT/e -o k -- "$@" > $OUT/e.log 2>&1
cd $ROOT
python - "$OUT" "$KERN" <<'PY' | tee $OUT/summary.txt
import csv, glob, collections, sys
out, kern = sys.argv[1], sys.argv[2]
for sub in "abcde":
    acc = collections.defaultdict(list)
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (out, sub), recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(f"{k:24s} {sum(v)/len(v):18.0f} n={len(v)}")
PY
