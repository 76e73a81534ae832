"""One 640x480 pair through the drop-in tadataka.vo.dvo.PoseChangeEstimator (3 levels, Huber): ms per call with the
default pyramid ("skimage": to the bit, level 0 and clip included) and with the ideal-constants one."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tadataka_amd  # noqa: E402,F401
from tadataka.camera import CameraModel, CameraParameters  # noqa: E402
from tadataka.vo import dvo  # noqa: E402
from tadataka_amd import synthetic  # noqa: E402

pair = synthetic.make_pair(480, 640, seed=0)
cam = pair["cam"]
cm = CameraModel(CameraParameters(cam[0:2], cam[2:4]), distortion_model=None)
for mode in ("skimage", "ideal", "skimage", "ideal"):
    dvo.PYRAMID = mode
    est = dvo.PoseChangeEstimator(cm, cm, n_coarse_to_fine=3, max_iter=20)
    for _ in range(20):
        est(pair["I0"], pair["D0"], pair["I1"], "huber")
    t0 = time.perf_counter()
    n = 300
    for _ in range(n):
        est(pair["I0"], pair["D0"], pair["I1"], "huber")
    print(f"{mode:8s} {(time.perf_counter() - t0) / n * 1e3:.3f} ms per call")
