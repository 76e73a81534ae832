#!/bin/bash
# usage: bash tools/pmc_insts_multi.sh <tag> "<kernel-substr> <kernel-substr> ..." <command...>
# instruction-mix PMC passes of several kernels of one command; per-launch averages per kernel.
TAG=$1; KERNS=$2; shift 2
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmcm_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $OUT/a -o k -- "$@" > $OUT/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --output-format csv -d $OUT/b -o k -- "$@" > $OUT/b.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/c -o k -- "$@" > $OUT/c.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/d -o k -- "$@" > $OUT/d.log 2>&1
cd $ROOT
python - "$OUT" $KERNS <<'PY' | tee $OUT/summary.txt
import csv, glob, collections, sys
out, kerns = sys.argv[1], sys.argv[2:]
table = collections.defaultdict(dict)
for sub in "abcd":
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (out, sub), recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            for k in kerns:
                if k in r["Kernel_Name"]:
                    acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in acc.items():
            table[c][k] = sum(v) / len(v)
print("%-26s" % "counter" + "".join("%18s" % k[:17] for k in kerns))
for c in sorted(table):
    print("%-26s" % c + "".join("%18.0f" % table[c].get(k, float("nan")) for k in kerns))
PY
