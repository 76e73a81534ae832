#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/power_trace.sh
# Package power and shader clock sampled with rocm-smi while tools/kbench.py loops the full-resolution
# evaluation kernel (full, then probe) for 12 s each; kernel time from the same run beside it.
# Writes profiles/r03_power.txt.
OUT=profiles/r03_power.txt
echo "# rocm-smi samples (every ~0.5 s) during tools/kbench.py --seconds 12 [--probe]; idle sample first" > $OUT
sample() { rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | tr -s ' ' | tr '\n' ';'; echo; }
echo "idle: $(sample)" >> $OUT
for kind in full probe; do
  FLAG=""; [ $kind = probe ] && FLAG="--probe"
  python tools/kbench.py --seconds 12 --reps 1 $FLAG > gpurun_out/power_$kind.log 2>&1 &
  PID=$!
  sleep 4        # import + setup
  while kill -0 $PID 2>/dev/null; do echo "$kind: $(sample)" >> $OUT; sleep 0.5; done
  echo "$kind kernel: $(tail -1 gpurun_out/power_$kind.log)" >> $OUT
done
cat $OUT
