#!/bin/bash
# usage: bash tools/pmc_pyramid.sh <tag> <case>   -- counter passes of k_pyramid_stream on the bench batch (one pass per group)
TAG=$1; CASE=${2:-skimage_all}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
CMD="python $ROOT/tools/kbench_pyramid_one.py $CASE 6"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" \
           "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_WAVE_CYCLES" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCC_EA_WRREQ_STALL_sum TCC_EA_WRREQ_sum TCC_EA_RDREQ_sum TCC_EA_WR_UNCACHED_32B_sum TCC_HIT_sum TCC_MISS_sum" \
           "TA_BUSY_avr TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -o k -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i failed: $grp" >> $OUT/failed.txt
done
cd $ROOT
python - "$OUT" <<'PY' | tee $OUT/summary.txt
import csv, glob, collections, sys
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob("%s/p*/**/*counter_collection.csv" % out, recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_pyramid_stream" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:36s} {sum(v)/len(v):18.0f} n={len(v)}")
PY
cat $OUT/failed.txt 2>/dev/null
