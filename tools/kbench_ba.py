#!/usr/bin/env python3
"""Kernel times of the bundle-adjustment handle at 8 poses x 50 000 points (HIP events
inside the library): block sums, error-only reduce, one LM step."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tadataka_amd import _lib, ops, synthetic  # noqa: E402


def main():
    _lib.require_gpu()
    b = synthetic.make_ba_case()
    x_obs = ops.ba_projection(b["poses"], b["points"], b["vp_idx"], b["pt_idx"], jacobians=False)
    h = ops.BundleAdjustment(len(b["poses"]), len(b["points"]), b["vp_idx"], b["pt_idx"], x_obs)
    for _ in range(3):
        h.block_sums(b["poses_noisy"], b["points_noisy"])
    h.set_profiling(True)
    for _ in range(20):
        h.block_sums(b["poses_noisy"], b["points_noisy"])
        h.sum_squared_error(b["poses_noisy"], b["points_noisy"])
        h.step(b["poses_noisy"], b["points_noisy"], 1e-3)
    prof = h.get_profile()
    print(
          {k: round(v[1] / max(v[0], 1) * 1e3, 1) for k, v in prof.items()}, "us per launch")
    h.close()


if __name__ == "__main__":
    main()
