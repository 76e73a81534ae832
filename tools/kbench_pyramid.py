"""Times tdk_dvo_build_pyramid on the bench batch (256 VGA pairs x 3 arrays, 3 levels), alone on the
device: TDK_PYRAMID_STREAM=0 (tiles) against the default (streaming kernel).  Usage:
python tools/kbench_pyramid.py [pairs] [levels] [height] [width]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tadataka_amd import _lib, ops, synthetic  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 3
H = int(sys.argv[3]) if len(sys.argv) > 3 else 480
W = int(sys.argv[4]) if len(sys.argv) > 4 else 640
_lib.require_gpu()
cam = synthetic.camera_for(W, H)
poses = np.tile(ops.pose12(np.eye(3), np.zeros(3)), (B, 1))
for mode in ("0", "1", "0", "1"):
    os.environ["TDK_PYRAMID_STREAM"] = mode
    batch = ops.DvoBatch(B, H, W, n_levels=L, ratio=1.5)
    batch.set_anti_aliasing(True)
    batch.fill_synthetic(cam, poses, seed0=0)
    for _ in range(5):
        batch.build_pyramid()
    _lib.call("tdk_sync")
    n = 50
    t0 = time.perf_counter()
    for _ in range(n):
        batch.build_pyramid()
    _lib.call("tdk_sync")
    dt = (time.perf_counter() - t0) / n
    px = B * 3 * H * W
    out = sum(B * 3 * int(round(H / 1.5 ** l)) * int(round(W / 1.5 ** l)) for l in range(1, L))
    print(f"TDK_PYRAMID_STREAM={mode}: {dt * 1e3:.3f} ms per build, {(px + out) * 8 / dt / 1e12:.2f} TB/s of compulsory traffic")
    batch.close()
