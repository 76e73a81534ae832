#!/usr/bin/env python3
"""End-to-end timings through the host-pointer C ABI (PCIe copies included) for
the BASELINE configs other than the headline bench, next to the CPU oracle on
the same inputs.  Prints one line per measurement; used for DESIGN.md."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tadataka_amd  # noqa: E402
from tadataka_amd import _lib, ops, synthetic  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def timeit(fn, reps=5, warm=1):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def main():
    _lib.require_gpu()
    print("device:", _lib.device_name())
    # cfg2 through the drop-in API, one pair, host buffers
    from tadataka.camera import CameraModel, CameraParameters
    from tadataka.vo.dvo import PoseChangeEstimator
    pair = synthetic.make_pair(480, 640, seed=0)
    cam = pair["cam"]
    cm = CameraModel(CameraParameters(cam[:2], cam[2:]), None)
    est = PoseChangeEstimator(cm, cm, n_coarse_to_fine=3)
    t = timeit(lambda: est(pair["I0"], pair["D0"], pair["I1"], "huber"))
    from scipy.spatial.transform import Rotation
    tc = timeit(lambda: orc.dvo_estimate(pair["I0"], pair["D0"], pair["I1"], cam, cam, "huber", 3), reps=2)
    print(f"cfg2 PoseChangeEstimator 640x480 3-level huber, host buffers in/out: gpu {t*1e3:.2f} ms, "
          f"cpu oracle {tc*1e3:.1f} ms, x{tc/t:.1f}")

    # cfg3 semi-dense step 640x480
    c = synthetic.make_semi_dense_case(480, 640, seed=1)
    key = (c["cam"], c["key_image"], c["T_wk"]); ref = (c["cam"], c["ref_image"], c["T_wr"])
    args = (0.5, 10.0, 0.01, 0.01, 0.002, 0.005)
    pg, po = ops.make_params(*args), orc.make_params(*args)
    T10 = np.linalg.inv(c["T_wk"]) @ c["T_wr"]
    age0 = c["age"]
    for name, g, o in (
        ("increment_age", lambda: ops.increment_age(age0, c["cam"], c["cam"], T10, c["prior_depth"]),
         lambda: orc.increment_age(age0, c["cam"], c["cam"], T10, c["prior_depth"])),
        ("propagate", lambda: ops.propagate(T10, c["cam"], c["cam"], c["prior_depth"], c["prior_variance"], 1., 10., .01),
         lambda: orc.propagate(T10, c["cam"], c["cam"], c["prior_depth"], c["prior_variance"], 1., 10., .01)),
        ("update_depth", lambda: ops.update_depth(key, [ref], c["age"], c["prior_depth"], c["prior_variance"], pg),
         lambda: orc.update_depth(key, [ref], c["age"], c["prior_depth"], c["prior_variance"], po)),
    ):
        tg, to = timeit(g), timeit(o, reps=3)
        print(f"cfg3 {name} 640x480 host buffers: gpu {tg*1e3:.2f} ms, cpu oracle {to*1e3:.2f} ms, x{to/tg:.2f}")

    # cfg5 BA 8 poses x 50k points
    b = synthetic.make_ba_case()
    x_true = orc.ba_projection(b["poses"], b["points"], b["vp_idx"], b["pt_idx"], jacobians=False)
    tg = timeit(lambda: ops.ba_block_reduce(b["poses_noisy"], b["points_noisy"], x_true, b["vp_idx"], b["pt_idx"]))
    to = timeit(lambda: orc.ba_block_reduce(b["poses_noisy"], b["points_noisy"], x_true, b["vp_idx"], b["pt_idx"]), reps=2)
    print(f"cfg5 BA block reduce 8x50000 (400k obs) host buffers: gpu {tg*1e3:.2f} ms, cpu oracle {to*1e3:.1f} ms, x{to/tg:.1f}")
    tg = timeit(lambda: ops.ba_projection(b["poses_noisy"], b["points_noisy"], b["vp_idx"], b["pt_idx"]))
    to = timeit(lambda: orc.ba_projection(b["poses_noisy"], b["points_noisy"], b["vp_idx"], b["pt_idx"]), reps=2)
    print(f"cfg5 BA Projection.compute+jacobians 400k obs host buffers: gpu {tg*1e3:.2f} ms, cpu oracle {to*1e3:.1f} ms, x{to/tg:.1f}")
    # cfg5, the whole Levenberg-Marquardt loop (LocalBundleAdjustment.compute) on the device
    g = ops.BundleAdjustment(len(b["poses"]), len(b["points"]), b["vp_idx"], b["pt_idx"], x_true)
    box = {}

    def solve():
        box["r"] = g.solve(b["poses_noisy"], b["points_noisy"], max_iter=10, absolute_error_threshold=1e-20,
                           relative_error_threshold=1e-6)
    tg = timeit(solve)
    errs = box["r"][2]
    print(f"cfg5 BA Levenberg-Marquardt 8x50000, {len(errs) - 1} iterations on the device: gpu {tg*1e3:.2f} ms "
          f"({tg*1e3/(len(errs)-1):.2f} ms/iteration), mean sq. error {errs[0]:.3e} -> {errs[-1]:.3e}")
    g.close()


if __name__ == "__main__":
    main()
