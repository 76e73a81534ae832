#!/usr/bin/env python3
"""Per-kernel average duration of tools/other_kernels.py (rocprofv3 kernel trace)
next to the algorithmic bytes each launch has to move."""
import collections
import csv
import glob
import sys

N_PX = 640 * 480          # semi-dense maps
N_OBS = 8 * 50000         # BA observations
N_PTS, N_POSES = 50000, 8
# kernel -> (algorithmic bytes per launch, what they are)
BYTES = {
    "k_age_scatter": (N_PX * 8 + N_PX * 8, "read depth, atomicMax one u64 target per px"),
    "k_age_gather": (N_PX * 24, "read winner index + age0, write age1"),
    "k_propagate_scatter": (N_PX * 16 + N_PX * 12, "read depth+variance, link one list node per px"),
    "k_propagate_fold": (N_PX * (4 + 16) + N_PX * 16, "walk lists, write depth+variance"),
    "k_sobel": (N_PX * 24, "read image, write gx, gy"),
    "k_update_depth": (N_PX * (8 + 8 + 8 + 8 + 16 + 8) + N_PX * 24, "key/age/prior maps + gradients in; flag/depth/variance out (search reads excluded)"),
    "k_ba_projection": (N_OBS * (16 + 16 + 48 + 24) * 1, "indices in; x, A (2x6), B (2x3) out (pose/point gathers cached)"),
    "k_ba_block_reduce": (N_OBS * (16 + 16 + 24), "indices, observation, point gather per observation"),
    "k_ba_invert_V": (N_PTS * (48 + 72), "V (6 upper) in, V^-1 (3x3) out"),
    "k_ba_schur": (N_OBS * (144 + 8) + N_PTS * (72 + 24), "W (6x3) per observation, V^-1, e_b per point"),
    "k_ba_backsub": (N_OBS * (144 + 8) + N_PTS * (72 + 24 + 24), "W per observation, V^-1, e_b in; delta_b out"),
}


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0].split("<")[0]


def main():
    root = sys.argv[1]
    durs = collections.defaultdict(list)
    for f in glob.glob(root + "/trace/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            durs[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print(f'{"kernel":24s} {"calls":>6s} {"avg_us":>9s} {"alg_MB":>8s} {"GB/s":>8s}  notes')
    for k, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
        avg = sum(v) / len(v) / 1e3
        if k in BYTES:
            b, note = BYTES[k]
            print(f"{k:24s} {len(v):6d} {avg:9.1f} {b/1e6:8.1f} {b/avg/1e3:8.0f}  {note}")
        else:
            print(f"{k:24s} {len(v):6d} {avg:9.1f} {'':8s} {'':8s}")


if __name__ == "__main__":
    main()
