#!/usr/bin/env python3
"""Time of the anti-aliased pyramid of the bench batch (256 pairs 640x480): all levels, and level 1 alone."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tadataka_amd import _lib, ops, synthetic
import bench
_lib.require_gpu()
B = 256
cam = synthetic.camera_for(640, 480)
CASES = ((3, False), (2, False), (3, True), (2, True))
if len(sys.argv) > 1:       # e.g. "3e": three levels, ndimage order only; "2e": level 1 alone
    CASES = tuple((int(a[0]), a.endswith("e")) for a in sys.argv[1:])
for levels, exact in CASES:
    bt = ops.DvoBatch(B, 480, 640, n_levels=levels, ratio=1.5)
    bt.fill_synthetic(cam, bench.true_poses(B, 0), seed0=0, noise=0.02)
    bt.set_anti_aliasing(True, exact=exact)
    for _ in range(3):
        bt.build_pyramid()
    _lib.call("tdk_sync")
    t0 = time.perf_counter()
    for _ in range(20):
        bt.build_pyramid()
    _lib.call("tdk_sync")
    print({k: v for k, v in os.environ.items() if k.startswith("TDK_AA") or k.startswith("TDK_SEP")},
          "ndimage order" if exact else "tap lists (depth: ndimage order)", "levels", levels,
          "pyramid %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
    bt.close()
