#!/usr/bin/env python3
"""Where the time of one drop-in PoseChangeEstimator call goes (640x480, 3 levels)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tadataka_amd import _lib, ops, synthetic  # noqa: E402


def t(fn, n=200):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    _lib.require_gpu()
    pair = synthetic.make_pair(480, 640, seed=0)
    cam = pair["cam"]
    batch = ops.DvoBatch(1, 480, 640, n_levels=3, ratio=1.5)
    batch.set_anti_aliasing(True)
    ident = ops.pose12(np.eye(3), np.zeros(3))[None]
    up = lambda: batch.upload(0, pair["I0"], pair["D0"], pair["I1"])
    up()
    print("upload           %.3f ms" % t(up))
    print("build_pyramid    %.3f ms (async, +sync below)" % t(lambda: (batch.build_pyramid(), _lib.call("tdk_sync"))))
    batch.build_pyramid()
    print("estimate         %.3f ms" % t(lambda: batch.estimate(cam, cam, ident, ops.W_HUBER, 20)))
    full = lambda: (up(), batch.build_pyramid(), batch.estimate(cam, cam, ident, ops.W_HUBER, 20))
    print("all three        %.3f ms" % t(full))
    a = np.empty(3 * 480 * 640)
    print("numpy memcpy 7.4MB %.3f ms" % t(lambda: np.copyto(a[:480 * 640], pair["I0"].ravel())) , "(one image)")


if __name__ == "__main__":
    main()
