"""Host-side time of the calls of one step of the stream workload (upload_async / build_pyramid / estimate), for the note on launch overheads."""
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
    bt.build_pyramid()
    batches.append(bt)
pins = [ops.PinnedBuffer((B, H * W), dtype=np.uint8) for _ in batches]
for p in pins:
    src = np.stack([batches[0].download(i, 0, 'I1').ravel() for i in range(0, B, 32)])
    src = np.tile(src, (32, 1))[:B]
    p.array[:] = np.clip(np.rint(src * 255.0), 0, 255).astype(np.uint8)
acc = {"upload": [], "pyramid": [], "estimate": []}
def step(k, rec):
    a, b, c = batches[k % 3], batches[(k + 1) % 3], batches[(k + 2) % 3]
    t0 = time.perf_counter(); c.upload_async("I1", 0, B, pins[(k + 2) % 3])
    t1 = time.perf_counter(); b.build_pyramid(("I1",))
    t2 = time.perf_counter(); a.estimate(cam, cam, ident, ops.W_HUBER, 20)
    t3 = time.perf_counter()
    if rec:
        acc["upload"].append(t1 - t0); acc["pyramid"].append(t2 - t1); acc["estimate"].append(t3 - t2)
for k in range(3): step(k, False)
_lib.call("tdk_sync"); t0 = time.perf_counter()
for k in range(12): step(k, True)
_lib.call("tdk_sync")
print("step %.3f ms" % ((time.perf_counter() - t0) / 12 * 1e3), {k: round(float(np.median(v)) * 1e3, 3) for k, v in acc.items()})

