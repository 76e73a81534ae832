#!/usr/bin/env python3
"""Host time of each call of the drop-in semi-dense loop body at 640x480 (device maps), synchronised after
every call so that the pieces add up: where the per-frame time of `semi_dense_dropin_vga` goes."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tadataka_amd  # noqa: F401
import rust_bindings.semi_dense as rsd
from rust_bindings.camera import CameraParameters
from tadataka.matrix import inv_motion_matrix
from tadataka_amd import _lib, synthetic
H, W, n = 480, 640, 12
cam, depth0, T_w, images = synthetic.make_track(H, W, n, step=(0.01, 0.002, 0.003))
cp = CameraParameters((cam[0], cam[1]), (cam[2], cam[3]))
params = rsd.Params(0.5, 10.0, 0.01, 0.01, 0.002, 0.02)
rng = np.random.default_rng(3)
frame0 = rsd.Frame(cp, images[0], T_w[0]); refframes = [frame0]
d0 = depth0 * rng.uniform(0.9, 1.1, (H, W)); v0 = np.full((H, W), 0.05); a0 = np.zeros((H, W), dtype=np.uint64)
acc = {}
def timed(name, fn):
    _lib.call("tdk_sync"); t0 = time.perf_counter(); r = fn(); _lib.call("tdk_sync")
    acc.setdefault(name, []).append(time.perf_counter() - t0); return r
for i in range(1, n):
    T10 = np.dot(inv_motion_matrix(T_w[i]), T_w[i - 1])
    frame1 = timed("Frame()", lambda: rsd.Frame(cp, images[i], T_w[i]))
    timed("frame upload (first device use)", lambda: frame1._resident())
    a1 = timed("increment_age", lambda: rsd.increment_age(a0, frame0.camera_params, frame1.camera_params, T10, d0))
    d1, v1 = timed("propagate", lambda: rsd.propagate(T10, frame0.camera_params, frame1.camera_params, d0, v0, 1.0, 10.0, 0.01))
    d1, v1, f1 = timed("update_depth", lambda: rsd.update_depth(frame1, refframes, a1, d1, v1, params))
    refframes.append(frame1); d0, v0, a0 = d1, v1, a1; frame0 = frame1
for k, v in acc.items():
    print(f"{k:36s} {np.median(v[2:]) * 1e3:8.3f} ms")
print(f"{'sum':36s} {sum(np.median(v[2:]) for v in acc.values()) * 1e3:8.3f} ms")
