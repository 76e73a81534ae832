#!/bin/bash
# memory-pipe counters of the evaluation kernel (run on the GPU box)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/kprof2_${1:-x}; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
shift
run() { d=$1; shift; timeout 90 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$d -o k -- python $ROOT/tools/kbench.py --reps 5 > $OUT/$d.log 2>&1; }
run b TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE
run c TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE
run d SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU
cd $ROOT
python - <<PY
import csv,glob,collections
for sub in "abcd":
    acc=collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%sub, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_dvo_eval" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in sorted(acc.items()): print(f"{k:40s} {sum(v)/len(v):18.0f} n={len(v)}")
PY
