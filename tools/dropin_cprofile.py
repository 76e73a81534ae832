#!/usr/bin/env python3
"""cProfile of the drop-in semi-dense loop body at 640x480 (device maps, no synchronisation between calls)."""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tadataka_amd  # noqa: F401
import rust_bindings.semi_dense as rsd
from rust_bindings.camera import CameraParameters
from tadataka.matrix import inv_motion_matrix
from tadataka_amd import _lib, synthetic
H, W, n = 480, 640, 40
cam, depth0, T_w, images = synthetic.make_track(H, W, n, step=(0.01, 0.002, 0.003))
cp = CameraParameters((cam[0], cam[1]), (cam[2], cam[3]))
params = rsd.Params(0.5, 10.0, 0.01, 0.01, 0.002, 0.02)
rng = np.random.default_rng(3)
T10s = [np.dot(inv_motion_matrix(T_w[i]), T_w[i - 1]) for i in range(1, n)]


def loop(k0, k1, state):
    frame0, refframes, d0, v0, a0 = state
    for i in range(k0, k1):
        frame1 = rsd.Frame(cp, images[i], T_w[i])
        a1 = rsd.increment_age(a0, frame0.camera_params, frame1.camera_params, T10s[i - 1], d0)
        d1, v1 = rsd.propagate(T10s[i - 1], frame0.camera_params, frame1.camera_params, d0, v0, 1.0, 10.0, 0.01)
        d1, v1, f1 = rsd.update_depth(frame1, refframes, a1, d1, v1, params)
        refframes.append(frame1); d0, v0, a0 = d1, v1, a1; frame0 = frame1
    return frame0, refframes, d0, v0, a0


frame0 = rsd.Frame(cp, images[0], T_w[0])
state = (frame0, [frame0], depth0 * rng.uniform(0.9, 1.1, (H, W)), np.full((H, W), 0.05), np.zeros((H, W), dtype=np.uint64))
state = loop(1, 8, state)
_lib.call("tdk_sync")
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
state = loop(8, n, state)
_lib.call("tdk_sync")
pr.disable()
print("ms per frame (profiled) %.3f" % ((time.perf_counter() - t0) / (n - 8) * 1e3))
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
