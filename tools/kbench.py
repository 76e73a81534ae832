#!/usr/bin/env python3
"""Micro-benchmark of the DVO evaluation kernel alone (one level, fixed pose):
used for A/B work on the kernel.  Prints Gpx/s and GB/s (24 B/px)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tadataka_amd import _lib, ops, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=256)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--mode", default="huber")
    ap.add_argument("--probe", action="store_true", help="error-only evaluations (tdk_dvo_photometric_error)")
    ap.add_argument("--seconds", type=float, default=0.0, help="loop for this long instead of --reps launches")
    args = ap.parse_args()
    _lib.require_gpu()
    B, H, W = args.pairs, args.height, args.width
    cam = synthetic.camera_for(W, H)
    poses = np.tile(ops.pose12(np.eye(3), np.zeros(3)), (B, 1))
    rng = np.random.default_rng(0)
    truth = poses.copy()
    for i in range(B):
        truth[i, :9] = synthetic.rodrigues(rng.uniform(-0.005, 0.005, 3)).ravel()
        truth[i, 9:] = rng.uniform(-0.01, 0.01, 3)
    batch = ops.DvoBatch(B, H, W, with_weight_map=(args.mode == "map"))
    batch.fill_synthetic(cam, truth, 0, 0.02)
    mode = {"none": ops.W_NONE, "huber": ops.W_HUBER, "map": ops.W_MAP}[args.mode]
    if args.probe:
        run = lambda: batch.photometric_error(0, cam, cam, truth)
    else:
        run = lambda: batch.evaluate(0, cam, cam, truth, mode)
    for _ in range(3):
        run()
    batch.set_profiling(True)
    t0 = time.perf_counter()
    reps = 0
    while reps < args.reps or time.perf_counter() - t0 < args.seconds:
        run()
        reps += 1
    args.reps = reps
    wall = time.perf_counter() - t0
    prof = batch.get_profile("probe" if args.probe else "full")
    ms = prof["total_ms"] / prof["launches"]
    px = B * H * W
    print(f"{'probe' if args.probe else 'full'} x{reps}: kernel {ms*1e3:.1f} us  {px/ms/1e6:.1f} Gpx/s  {px*24/ms/1e6:.0f} GB/s (24 B/px)  "
          f"host-loop {wall/args.reps*1e3:.3f} ms/eval")


if __name__ == "__main__":
    main()
