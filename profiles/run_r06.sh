#!/bin/bash
# Round 6: every profile of profiles/r06_* from the final code, one gpurun call:  bash profiles/run_r06.sh
# Raw output under gpurun_out/r06/ (scratch); the summaries are copied into profiles/ by hand.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
echo "== bench.py (the driver's command)"; timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err; python tools/show_bench.py $OUT/bench.json
echo "== kernel traces + counters of the headline"; timeout 1500 bash profiles/run_profile.sh r06 > $OUT/profile.log 2>&1; tail -5 $OUT/profile.log
echo "== pyramid counters"; timeout 900 bash tools/pmc_pyramid.sh r06 skimage_all > $OUT/pmc_pyramid.txt 2>&1; tail -3 $OUT/pmc_pyramid.txt
timeout 300 python tools/kbench_pyramid.py > $OUT/kbench_pyramid.txt 2>&1
echo "== semi-dense + BA kernel times"; timeout 900 bash profiles/run_workloads.sh r06 all > $OUT/workloads.txt 2>&1; tail -30 $OUT/workloads.txt
timeout 300 python tools/kbench_sd_warp.py > $OUT/kbench_sd_warp.txt 2>&1; timeout 300 python tools/kbench_sd_update.py > $OUT/kbench_sd_update.txt 2>&1; timeout 300 python tools/kbench_ba.py > $OUT/kbench_ba.txt 2>&1
echo "== semi-dense counters"; timeout 1500 bash tools/pmc_insts_multi.sh r06sd "k_sd_targets k_sd_gather2 k_ud_classify k_ud_estimate" python $ROOT/tools/sd_child.py 6 > $OUT/pmc_sd.txt 2>&1; tail -40 $OUT/pmc_sd.txt
echo "== robust modes"; for w in tukey student-t; do timeout 600 python bench.py --weights $w --no-workloads --no-cpu-baseline --no-traffic-pass --no-solo-pass --min-seconds 3 > $OUT/bench_$w.json 2> $OUT/bench_$w.err; python -c "
import json,sys; d=json.loads(open('$OUT/bench_$w.json').read().strip().splitlines()[-1]); print('$w', d['ms_per_step'])"; done
