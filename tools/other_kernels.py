#!/usr/bin/env python3
"""GPU calls only (no oracle) of the semi-dense and bundle-adjustment paths at the
BASELINE sizes, repeated, for `rocprofv3 --kernel-trace --stats`
(profiles/run_other_kernels.sh turns the trace into profiles/r01_other_kernels.txt)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tadataka_amd import _lib, ops, synthetic  # noqa: E402

REPS = 5


def main():
    _lib.require_gpu()
    # semi-dense step, 640x480 (BASELINE configs[3])
    c = synthetic.make_semi_dense_case(480, 640, seed=1)
    key = (c["cam"], c["key_image"], c["T_wk"]); ref = (c["cam"], c["ref_image"], c["T_wr"])
    params = ops.make_params(0.5, 10.0, 0.01, 0.01, 0.002, 0.005)
    T10 = np.linalg.inv(c["T_wk"]) @ c["T_wr"]
    for _ in range(REPS):
        ops.increment_age(c["age"], c["cam"], c["cam"], T10, c["prior_depth"])
        ops.propagate(T10, c["cam"], c["cam"], c["prior_depth"], c["prior_variance"], 1., 10., .01)
        ops.update_depth(key, [ref], c["age"], c["prior_depth"], c["prior_variance"], params)
    # bundle adjustment, 8 poses x 50 000 points, every point seen by every pose (configs[5])
    b = synthetic.make_ba_case()
    x_obs = ops.ba_projection(b["poses"], b["points"], b["vp_idx"], b["pt_idx"], jacobians=False)
    ba = ops.BundleAdjustment(len(b["poses"]), len(b["points"]), b["vp_idx"], b["pt_idx"], x_obs)
    for _ in range(REPS):
        ops.ba_projection(b["poses_noisy"], b["points_noisy"], b["vp_idx"], b["pt_idx"])
        ops.ba_block_reduce(b["poses_noisy"], b["points_noisy"], x_obs, b["vp_idx"], b["pt_idx"])
        ba.step(b["poses_noisy"], b["points_noisy"], 1e-3)
    ba.close()
    print("ok")


if __name__ == "__main__":
    main()
