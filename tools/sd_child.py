"""The semi_dense_vga workload of bench.py, GPU calls only, a few steps: the child process of bench.py's HBM-traffic
passes for the semi-dense kernels (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE around it) and of tools/pmc_*.sh.
usage: python tools/sd_child.py [steps] [tracks]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tadataka_amd import _lib, ops, synthetic  # noqa: E402

_lib.require_gpu()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
H, W = 480, 640
sd = ops.SemiDenseSession(B, H, W, max_refframes=2)
sd.set_age_policy(False)
sd.set_params(ops.make_params(0.5, 10.0, 0.01, 0.01, 0.002, 0.02), 1.0, 10.0, 0.01)
base = synthetic.make_semi_dense_case(H, W, seed=1)
T10 = np.linalg.inv(base["T_wk"]) @ base["T_wr"]
for t in range(B):
    sd.push_frame(t, base["cam"], base["ref_image"], base["T_wr"])
    sd.push_frame(t, base["cam"], base["key_image"], base["T_wk"])
    if t == 0:
        age, pd_ = base["age"], base["prior_depth"]
    else:
        rng = np.random.default_rng(1000 + t)
        age = (rng.uniform(0, 1, (H, W)) < 0.3).astype(np.uint64)
        pd_ = base["depth_gt"] * rng.uniform(0.9, 1.1, (H, W))
    sd.set_maps(t, pd_, base["prior_variance"], age)
T10s = np.tile(T10, (B, 1, 1))
for _ in range(steps):
    sd.propagate(T10s, commit=False)
    sd.update_depth(commit=False)
_lib.call("tdk_sync")
sd.close()
