#!/usr/bin/env python3
"""GPU calls only (no oracle) of the semi-dense session and the bundle-adjustment
handle at the BASELINE sizes, repeated, for `rocprofv3 --kernel-trace --stats`
(profiles/run_workloads.sh)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tadataka_amd import _lib, ops, synthetic  # noqa: E402

REPS = 5
WHAT = sys.argv[1] if len(sys.argv) > 1 else "all"


def semi_dense(B=64):
    H, W = 480, 640
    sd = ops.SemiDenseSession(B, H, W, max_refframes=2)
    sd.set_params(ops.make_params(0.5, 10.0, 0.01, 0.01, 0.002, 0.02), 1.0, 10.0, 0.01)
    base = synthetic.make_semi_dense_case(H, W, seed=1)
    T10 = np.linalg.inv(base["T_wk"]) @ base["T_wr"]
    for t in range(B):
        sd.push_frame(t, base["cam"], base["ref_image"], base["T_wr"])
        sd.push_frame(t, base["cam"], base["key_image"], base["T_wk"])
        rng = np.random.default_rng(1000 + t)
        age = base["age"] if t == 0 else (rng.uniform(0, 1, (H, W)) < 0.3).astype(np.uint64)
        sd.set_maps(t, base["prior_depth"], base["prior_variance"], age)
    T10s = np.tile(T10, (B, 1, 1))
    for _ in range(REPS):
        sd.propagate(T10s, commit=False)
        sd.update_depth(commit=False)
    print("sd timing", sd.timing())
    sd.close()


def ba():
    b = synthetic.make_ba_case()
    x_obs = ops.ba_projection(b["poses"], b["points"], b["vp_idx"], b["pt_idx"], jacobians=False)
    h = ops.BundleAdjustment(len(b["poses"]), len(b["points"]), b["vp_idx"], b["pt_idx"], x_obs)
    for _ in range(REPS):
        h.block_sums(b["poses_noisy"], b["points_noisy"])
        h.sum_squared_error(b["poses_noisy"], b["points_noisy"])
        h.step(b["poses_noisy"], b["points_noisy"], 1e-3)
    h.close()


def main():
    _lib.require_gpu()
    if WHAT in ("all", "sd"):
        semi_dense()
    if WHAT in ("all", "ba"):
        ba()
    print("ok")


if __name__ == "__main__":
    main()
