"""One pyramid reading of tools/kbench_pyramid.py, for a kernel trace: python tools/kbench_pyramid_one.py <case> [n]
case: ideal | skimage_d0 | skimage_all | skimage_noclip"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tadataka_amd import _lib, ops, synthetic  # noqa: E402

case = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B, L, H, W = 256, 3, 480, 640
_lib.require_gpu()
cam = synthetic.camera_for(W, H)
poses = np.tile(ops.pose12(np.eye(3), np.zeros(3)), (B, 1))
batch = ops.DvoBatch(B, H, W, n_levels=L, ratio=1.5)
if case == "ideal":
    batch.set_anti_aliasing(True)
else:
    batch.set_skimage_pyramid(level0=["D0"] if case == "skimage_d0" else "all", clip=case != "skimage_noclip")
batch.fill_synthetic(cam, poses, seed0=0)
import time
for _ in range(5):
    batch.build_pyramid()
_lib.call("tdk_sync")
t0 = time.perf_counter()
for _ in range(n):
    batch.build_pyramid()
_lib.call("tdk_sync")
if len(sys.argv) > 3:
    print(f"{(time.perf_counter() - t0) / n * 1e3:.3f} ms per build")
batch.close()
