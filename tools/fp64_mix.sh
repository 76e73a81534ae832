#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/fp64_mix.sh
# FP64 instruction mix, busy cycles and HBM traffic of k_dvo_eval<huber> at full resolution (256 VGA
# pairs), full evaluations and error-only probes separately; rocprofv3 --pmc in passes of their own
# (no tracing next to the counters).  Writes profiles/r05_fp64_mix.json + profiles/r05_fp64_mix.txt.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/fp64mix; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for kind in full probe; do
  FLAG=""; [ $kind = probe ] && FLAG="--probe"
  CMD="python $ROOT/tools/kbench.py --reps 12 $FLAG"
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/$kind/a -o k -- $CMD > $OUT/$kind.a.log 2>&1
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/$kind/b -o k -- $CMD > $OUT/$kind.b.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$kind/c -o k -- $CMD > $OUT/$kind.c.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/$kind/d -o k -- $CMD > $OUT/$kind.d.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$kind/t -o k -- $CMD > $OUT/$kind.t.log 2>&1
done
cd $ROOT
python tools/fp64_mix_summary.py $OUT | tee profiles/r05_fp64_mix.txt
