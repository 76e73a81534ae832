"""Times tdk_dvo_build_pyramid on the bench batch (256 VGA pairs x 3 arrays, 3 levels), alone on the
device, for the pyramid readings the library offers:
    ideal / tiles     ideal constants, tdk_set_option(TDK_OPT_PYRAMID_STREAM, 0)
    ideal / stream    ideal constants, the streaming kernel (the headline of rounds 1-4)
    skimage D0        skimage to the bit, level 0 of its own for the depth map only
    skimage all       skimage to the bit, level 0 (rescale(., 1.0)) for every array -- what the reference builds
    skimage noclip    the same without clip=True's tracking
Usage: python tools/kbench_pyramid.py [pairs] [levels] [height] [width]"""
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
CASES = (("ideal / tiles", "0", None), ("ideal / stream", "1", None), ("skimage D0", "1", (["D0"], True)),
         ("skimage all", "1", ("all", True)), ("skimage noclip", "1", ("all", False)),
         ("skimage all, L0 own kernel", "3", ("all", True)), ("skimage noclip, L0 own", "3", ("all", False)))
for rep in range(2):
    for name, stream, sk in CASES:
        ops.set_option("pyramid_stream", int(stream))
        batch = ops.DvoBatch(B, H, W, n_levels=L, ratio=1.5)
        if sk is None:
            batch.set_anti_aliasing(True)
        else:
            batch.set_skimage_pyramid(level0=sk[0], clip=sk[1])
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
        out += B * bin(batch.level0_mask & 7).count("1") * H * W
        print(f"{name:28s}: {dt * 1e3:.3f} ms per build, {(px + out) * 8 / dt / 1e12:.2f} TB/s of compulsory traffic "
              f"({(px + out) * 8 / 1e9:.2f} GB)", flush=True)
        batch.close()
