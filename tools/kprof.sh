#!/bin/bash
# PMC snapshot of the evaluation kernel via tools/kbench.py (run on the GPU box)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/kprof_${1:-x}; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
shift
timeout 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $OUT/a -o k -- python $ROOT/tools/kbench.py --reps 5 "$@" > $OUT/a.log 2>&1
timeout 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES --output-format csv -d $OUT/b -o k -- python $ROOT/tools/kbench.py --reps 5 "$@" > $OUT/b.log 2>&1
cd $ROOT
python - <<PY
import csv,glob,collections
for sub in ("a","b"):
    acc=collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%sub, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_dvo_eval" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in sorted(acc.items()): print(f"{k:24s} {sum(v)/len(v):16.0f} n={len(v)}")
PY
