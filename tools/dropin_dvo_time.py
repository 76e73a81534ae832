"""ms per call of the drop-in tadataka.vo.dvo.PoseChangeEstimator (640x480, 3 levels, Huber)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tadataka_amd  # noqa
from tadataka.camera import CameraModel, CameraParameters
from tadataka.vo import dvo
from tadataka_amd import synthetic
pair = synthetic.make_pair(480, 640, seed=0)
cam = pair["cam"]
cm = CameraModel(CameraParameters(cam[0:2], cam[2:4]), distortion_model=None)
est = dvo.PoseChangeEstimator(cm, cm, n_coarse_to_fine=3, max_iter=20)
for _ in range(5):
    pose = est(pair["I0"], pair["D0"], pair["I1"], "huber")
n = 300
t0 = time.perf_counter()
for _ in range(n):
    pose = est(pair["I0"], pair["D0"], pair["I1"], "huber")
print("ms per call %.4f" % ((time.perf_counter() - t0) / n * 1e3), pose.t)
if len(sys.argv) > 1:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200):
        est(pair["I0"], pair["D0"], pair["I1"], "huber")
    pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(12)
