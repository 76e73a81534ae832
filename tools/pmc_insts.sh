#!/bin/bash
# usage: bash tools/pmc_insts.sh <tag> <kernel-substring> <command...>
# instruction-mix PMC passes (separate, --pmc only) of one kernel; prints per-launch averages.
TAG=$1; KERN=$2; shift 2
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmci_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $OUT/a -o k -- "$@" > $OUT/a.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --output-format csv -d $OUT/b -o k -- "$@" > $OUT/b.log 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/c -o k -- "$@" > $OUT/c.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/d -o k -- "$@" > $OUT/d.log 2>&1
cd $ROOT
python - "$OUT" "$KERN" <<'PY' | tee $OUT/summary.txt
import csv, glob, collections, sys
out, kern = sys.argv[1], sys.argv[2]
for sub in "abcd":
    acc = collections.defaultdict(list)
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (out, sub), recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not acc:
        print("pass", sub, "no data:", open("%s/%s.log" % (out, sub)).read()[-400:])
    for k, v in sorted(acc.items()):
        print(f"{k:28s} {sum(v)/len(v):18.0f} n={len(v)}")
PY
