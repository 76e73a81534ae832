#!/usr/bin/env python3
"""Summarises tools/fp64_mix.sh: per-launch counter averages of k_dvo_eval -> FLOPs per pixel,
FP64 rate, effective clock, HBM traffic; writes profiles/r05_fp64_mix.json (read by bench.py; r03_fp64_mix.* are the
same measurement of the round-3 kernel: 110 FP64 instructions per pixel where round 5 has ~99)."""
import collections
import csv
import glob
import json
import os
import sys

out_dir = sys.argv[1]
PX = 256 * 480 * 640
res = {}
KERNEL = {"full": "k_dvo_eval", "probe": "k_dvo_probe"}   # the error-only evaluation is a kernel of its own since round 3
for kind in ("full", "probe"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out_dir}/{kind}/[abcd]/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if KERNEL[kind] in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    avg = {k: sum(v) / len(v) for k, v in acc.items()}
    dur = []
    for f in glob.glob(f"{out_dir}/{kind}/t/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if KERNEL[kind] in r["Kernel_Name"]:
                dur.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3)
    dur_us = sum(dur) / len(dur) if dur else float("nan")
    g = lambda k: avg.get(k, float("nan"))
    insts = g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_TRANS_F64")
    flops = 64.0 * (g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_TRANS_F64")
                    + 2.0 * g("SQ_INSTS_VALU_FMA_F64"))
    clock = g("GRBM_GUI_ACTIVE") / 8.0 / (dur_us * 1e-6) / 1e9 if dur_us == dur_us else float("nan")
    # MI355X_MICROARCH.md, HBM section: FETCH_SIZE / WRITE_SIZE in passes of their own, KiB units; on gfx950
    # FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream -> the read side is doubled
    hbm = (2.0 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024.0
    res[kind] = {
        "flops_per_px": flops / PX, "fp64_insts_per_px": insts * 64.0 / PX,
        "valu_insts_per_px": g("SQ_INSTS_VALU") * 64.0 / PX,
        "kernel_us_in_profile": dur_us, "launches_averaged": len(dur),
        "tflops_in_profile": flops / (dur_us * 1e-6) / 1e12 if dur_us == dur_us else None,
        "effective_clock_ghz": clock,
        "fp64_pipe_busy": insts * 4.0 / 1024.0 / (g("GRBM_GUI_ACTIVE") / 8.0),
        "wait_any_over_wave_cycles": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"),
        "hbm_bytes_per_launch": hbm, "hbm_bytes_per_px": hbm / PX,
        "counters": avg,
    }
res["effective_clock_ghz"] = res["full"]["effective_clock_ghz"]
res["source"] = "tools/fp64_mix.sh: rocprofv3 --pmc passes over tools/kbench.py (256 pairs 640x480, huber), per-launch averages"
json.dump(res, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r05_fp64_mix.json"), "w"), indent=1)
for kind in ("full", "probe"):
    r = res[kind]
    print(f"[{kind}] {r['kernel_us_in_profile']:.1f} us/launch over {r['launches_averaged']} launches | "
          f"FP64 insts/px {r['fp64_insts_per_px']:.1f} | FLOPs/px {r['flops_per_px']:.1f} | "
          f"{r['tflops_in_profile']:.1f} TFLOP/s = {r['tflops_in_profile'] / 78.6:.3f} of 78.6 | clock {r['effective_clock_ghz']:.2f} GHz | "
          f"FP64 pipe busy {r['fp64_pipe_busy']:.2f} | WAIT_ANY/WAVE_CYCLES {r['wait_any_over_wave_cycles']:.2f} | "
          f"HBM {r['hbm_bytes_per_px']:.1f} B/px")
    for k, v in sorted(r["counters"].items()):
        print(f"    {k:28s} {v:18.0f}")
