#!/usr/bin/env python3
"""dvo_stream_x256 taken apart: step time with / without the upload, upload alone, for both formats."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tadataka_amd import _lib, ops, synthetic
import bench
_lib.require_gpu()
B, H, W = 256, 480, 640
cam = synthetic.camera_for(W, H)
ident = np.tile(ops.pose12(np.eye(3), np.zeros(3)), (B, 1))
batches = []
for k in range(3):
    bt = ops.DvoBatch(B, H, W, n_levels=3, ratio=1.5)
    bt.fill_synthetic(cam, bench.true_poses(B, k * B), seed0=k * B, noise=0.02)
    batches.append(bt)
for dtype in (np.uint8, np.float64):
    pins = [ops.PinnedBuffer((B, H * W), dtype=dtype) for _ in batches]
    for p in pins:
        src = np.stack([batches[0].download(i, 0, 'I1').ravel() for i in range(0, B, 32)])
        src = np.tile(src, (32, 1))[:B]
        p.array[:] = np.clip(np.rint(src * 255.0), 0, 255).astype(np.uint8) if dtype == np.uint8 else src
    def run(upload, compute, steps=8, arrays=None):
        def step(k):
            a, b, c = batches[k % 3], batches[(k + 1) % 3], batches[(k + 2) % 3]
            if upload: c.upload_async("I1", 0, B, pins[(k + 2) % 3])
            if compute:
                b.build_pyramid(arrays)
                a.estimate(cam, cam, ident, ops.W_HUBER, 20)
        for k in range(3): step(k)
        _lib.call("tdk_sync"); t0 = time.perf_counter()
        for k in range(steps): step(k)
        _lib.call("tdk_sync")
        return (time.perf_counter() - t0) / steps * 1e3
    print(np.dtype(dtype).name, "compute only %.3f ms | upload only %.3f ms | both %.3f ms" % (run(False, True), run(True, False), run(True, True)))
    print(np.dtype(dtype).name, "pyramid of I1 only: compute only %.3f ms | both %.3f ms" % (run(False, True, arrays=("I1",)), run(True, True, arrays=("I1",))))
    for p in pins: p.close()
