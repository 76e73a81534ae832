"""Kernel timeline of ONE drop-in-sized estimation (1 pair, 640x480, 3 levels, Huber) from a rocprofv3
kernel trace: python tools/single_pair_timeline.py run  (under rocprofv3 --kernel-trace), then
python tools/single_pair_timeline.py show <kernel_trace.csv>."""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import numpy as np
    from tadataka_amd import _lib, ops, synthetic
    _lib.require_gpu()
    pair = synthetic.make_pair(480, 640, seed=0)
    cam = pair["cam"]
    batch = ops.DvoBatch(1, 480, 640, n_levels=3, ratio=1.5)
    batch.set_anti_aliasing(True)
    ident = ops.pose12(np.eye(3), np.zeros(3))[None]
    for _ in range(30):
        batch.upload(0, pair["I0"], pair["D0"], pair["I1"])
        batch.build_pyramid()
        batch.estimate(cam, cam, ident, ops.W_HUBER, 20)
    _lib.call("tdk_sync")


def show(path):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    # the last call: from the last pyramid kernel on
    start = max(i for i, r in enumerate(rows) if "rescale" in r["Kernel_Name"] or "pyramid" in r["Kernel_Name"])
    while start > 0 and ("rescale" in rows[start - 1]["Kernel_Name"] or "pyramid" in rows[start - 1]["Kernel_Name"]):
        start -= 1
    t0 = int(rows[start]["Start_Timestamp"])
    prev_end = t0
    busy = 0.0
    for r in rows[start:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("(")[0][-40:]
        print(f"{(s - t0) / 1e3:8.1f} us  +gap {(s - prev_end) / 1e3:6.1f}  dur {(e - s) / 1e3:6.1f}  grid {r['Grid_Size_X']:>7s}  {name}")
        busy += (e - s) / 1e3
        prev_end = e
    print(f"span {(prev_end - t0) / 1e3:.1f} us, kernels {busy:.1f} us, {len(rows) - start} launches")


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else show(sys.argv[2])
