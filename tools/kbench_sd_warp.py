"""Kernel time of the semi-dense forward warp (increment_age + propagate) of bench.py's semi_dense_vga
workload: 64 VGA tracks, SURVEY 8(d) cfg3 maps; gather path (default) against the slot path (tdk_set_option(TDK_OPT_SD_WARP_GATHER, 0))."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tadataka_amd import _lib, ops, synthetic  # noqa: E402

_lib.require_gpu()
B, H, W = 64, 480, 640
for mode in ("0", "1", "0", "1"):
    ops.set_option("sd_warp_gather", int(mode))
    sd = ops.SemiDenseSession(B, H, W, max_refframes=2)
    sd.set_age_policy(False)
    sd.set_params(ops.make_params(0.5, 10.0, 0.01, 0.01, 0.002, 0.02), 1.0, 10.0, 0.01)
    base = synthetic.make_semi_dense_case(H, W, seed=1)
    T10 = np.linalg.inv(base["T_wk"]) @ base["T_wr"]
    for t in range(B):
        sd.push_frame(t, base["cam"], base["ref_image"], base["T_wr"])
        sd.push_frame(t, base["cam"], base["key_image"], base["T_wk"])
        rng = np.random.default_rng(1000 + t)
        age = (rng.uniform(0, 1, (H, W)) < 0.3).astype(np.uint64)
        sd.set_maps(t, base["depth_gt"] * rng.uniform(0.9, 1.1, (H, W)), base["prior_variance"], age)
    T10s = np.tile(T10, (B, 1, 1))
    sd.propagate(T10s, commit=False)
    ms = 0.0
    for _ in range(20):
        sd.propagate(T10s, commit=False)
        ms += sd.timing()["warp_ms"]
    print(f"sd_warp_gather={mode}: warp step {ms / 20:.3f} ms for {B} tracks ({56.0 * B * H * W / (ms / 20 * 1e-3) / 1e12:.2f} TB/s of 56 B/px), fallbacks {sd.warp_fallbacks()}")
    sd.close()
