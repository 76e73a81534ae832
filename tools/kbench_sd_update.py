"""Kernel time of update_depth (k_ud_classify + k_ud_estimate) on bench.py's semi_dense_vga workload:
64 VGA tracks, SURVEY 8(d) cfg3 maps (~30 % valid pixels, ~10 search positions per pixel)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tadataka_amd import _lib, ops, synthetic  # noqa: E402

_lib.require_gpu()
B, H, W = 64, 480, 640
sd = ops.SemiDenseSession(B, H, W, max_refframes=2)
sd.set_age_policy(False)
sd.set_params(ops.make_params(0.5, 10.0, 0.01, 0.01, 0.002, 0.02), 1.0, 10.0, 0.01)
base = synthetic.make_semi_dense_case(H, W, seed=1)
for t in range(B):
    sd.push_frame(t, base["cam"], base["ref_image"], base["T_wr"])
    sd.push_frame(t, base["cam"], base["key_image"], base["T_wk"])
    rng = np.random.default_rng(1000 + t)
    age = (rng.uniform(0, 1, (H, W)) < 0.3).astype(np.uint64)
    sd.set_maps(t, base["depth_gt"] * rng.uniform(0.9, 1.1, (H, W)), base["prior_variance"], age)
sd.update_depth(commit=False)
ms = 0.0
n = int(os.environ.get("N", "20"))
for _ in range(n):
    sd.update_depth(commit=False)
    ms += sd.timing()["update_depth_ms"]
print(f"update_depth {ms / n:.3f} ms for {B} tracks")
sd.close()
