#!/usr/bin/env python3
"""Round-3 fixtures: the reference's own PoseChangeEstimator (imported from
/root/reference with the stubs of generate_golden.py) at the settings its examples and
tests really use.  Runs in the build container only; the .npz outputs travel.

  dvo_examples.npz  examples/dvo_pose_change.py:34-37 (n_coarse_to_fine=5) on a 240x320
                    pair; examples/semi_dense_vo.py:45-54 (n_coarse_to_fine=7, weights =
                    safe_invert(variance map)) at 640x480; Student-t / Tukey / ndarray
                    weights through the 3-level VGA pyramid (tests/vo/test_dvo.py:46-50
                    runs every weight option through PoseChangeEstimator)
  dvo_real.npz      two New-Tsukuba frames (tests/dataset/new_tsukuba/.../left/00201,
                    00205: the dataset tests/vo/test_dvo.py:24-53 uses), rgb2gray ->
                    float, analytic depth (the depth XMLs are not in the checkout), all
                    five weight options, both pyramid readings, 5 levels at 480x640 and
                    the example's half-resolution call
  dvo_holes.npz     depth maps with missing readings (zeros as sensors report them, NaNs), four weight options
  dvo_ill.npz       ill-conditioned scenes (tests/golden/scenes.py): per-update twists of
                    the reference's lstsq on J, final poses, singular values of J

Every record holds: final pose (rotvec, t), pose after each level, PhotometricError
evaluations per level.  Usage: python tests/golden/generate_golden_r3.py [--only NAME]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import generate_golden as gg          # noqa: E402  (stubs + reference import)
import scenes                         # noqa: E402
from tadataka_amd import synthetic    # noqa: E402
from oracle import oracle as orc      # noqa: E402  (pyramid stand-in of the stubs only)

TSUKUBA = "/root/reference/tests/dataset/new_tsukuba/illumination/daylight/left"


def run_pyramid(cam0, cam1, I0, D0, I1, weights, n_levels, aa, max_iter=20, ratio=1.5):
    """The reference's PoseChangeEstimator, instrumented: pose after each level and the
    number of PhotometricError evaluations per level (coarse -> fine)."""
    import tadataka.vo.dvo as dvo
    from tadataka.camera import CameraModel, CameraParameters
    cm0 = CameraModel(CameraParameters(cam0[0:2], cam0[2:4]), distortion_model=None)
    cm1 = CameraModel(CameraParameters(cam1[0:2], cam1[2:4]), distortion_model=None)
    counts, level_poses, xis = [], [], []
    orig_err_cls = dvo.PhotometricError
    orig_level = dvo._PoseChangeEstimator.__call__
    orig_solve = dvo.solve_linear_equation

    class Err(orig_err_cls):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            counts.append(0)

        def __call__(self, pose10):
            counts[-1] += 1
            return super().__call__(pose10)

    def level_call(self, I0_, D0_, I1_, pose10, weights_=None):
        pose = orig_level(self, I0_, D0_, I1_, pose10, weights_)
        level_poses.append(np.concatenate([pose.rotation.as_rotvec(), pose.t]))
        return pose

    def solve(J, r, weights_=None, **kw):
        xi = orig_solve(J, r, weights_, **kw)
        xis.append(xi.copy())
        return xi

    gg.ANTI_ALIASING[0] = aa
    dvo.PhotometricError = Err
    dvo._PoseChangeEstimator.__call__ = level_call
    dvo.solve_linear_equation = solve
    try:
        est = dvo.PoseChangeEstimator(cm0, cm1, n_coarse_to_fine=n_levels, max_iter=max_iter,
                                      layer_size_ratio=ratio)
        pose = est(I0, D0, I1, weights)
    finally:
        dvo.PhotometricError = orig_err_cls
        dvo._PoseChangeEstimator.__call__ = orig_level
        dvo.solve_linear_equation = orig_solve
        gg.ANTI_ALIASING[0] = False
    return dict(rotvec=pose.rotation.as_rotvec(), t=pose.t, evals=np.array(counts, dtype=np.int64),
                level_poses=np.array(level_poses), xis=np.array(xis).reshape(-1, 6))


def put(out, tag, rec):
    for k, v in rec.items():
        out[f"{tag}_{k}"] = v


def mode_arg(name, wmap):
    return wmap if name == "map" else (None if name == "None" else name)


def capture_examples():
    out = {}
    # examples/dvo_pose_change.py: half-resolution frames, n_coarse_to_fine=5, weights None
    pair = synthetic.make_pair(240, 320, seed=5)
    for aa in (False, True):
        for name in ("None", "huber"):
            rec = run_pyramid(pair["cam"], pair["cam"], pair["I0"], pair["D0"], pair["I1"],
                              mode_arg(name, None), 5, aa)
            put(out, f"ex5_{'aa' if aa else 'bl'}_{name}", rec)
            print("ex5", aa, name, rec["evals"], rec["t"])
    # examples/semi_dense_vo.py: 7 levels, weights = 1 / variance
    pair = synthetic.make_pair(480, 640, seed=0)
    wmap = scenes.weight_map((480, 640), seed=41)
    for aa in (False, True):
        rec = run_pyramid(pair["cam"], pair["cam"], pair["I0"], pair["D0"], pair["I1"], wmap, 7, aa)
        put(out, f"ex7_{'aa' if aa else 'bl'}_map", rec)
        print("ex7", aa, rec["evals"], rec["t"])
        # every robust option through the 3-level VGA pyramid of BASELINE configs[1]
        for name in ("student-t", "tukey", "map"):
            rec = run_pyramid(pair["cam"], pair["cam"], pair["I0"], pair["D0"], pair["I1"],
                              mode_arg(name, wmap), 3, aa)
            put(out, f"v3_{'aa' if aa else 'bl'}_{name}", rec)
            print("v3", aa, name, rec["evals"], rec["t"])
    return out


def load_rgb(path):
    from PIL import Image
    return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8))


def capture_real():
    rgb0 = load_rgb(os.path.join(TSUKUBA, "tsukuba_daylight_L_00201.png"))
    rgb1 = load_rgb(os.path.join(TSUKUBA, "tsukuba_daylight_L_00205.png"))
    out = dict(rgb0=rgb0, rgb1=rgb1)
    I0 = scenes.gray_from_rgb_u8(rgb0)
    I1 = scenes.gray_from_rgb_u8(rgb1)
    H, W = I0.shape
    D0 = scenes.tsukuba_depth(H, W)
    cam = scenes.TSUKUBA_CAM
    wmap = scenes.weight_map((H, W), seed=42)
    for aa in (False, True):
        for name in ("None", "huber", "student-t", "tukey", "map"):
            rec = run_pyramid(cam, cam, I0, D0, I1, mode_arg(name, wmap), 5, aa)
            put(out, f"full_{'aa' if aa else 'bl'}_{name}", rec)
            print("real full", aa, name, rec["evals"], rec["t"])
    # the example's get(): scale 0.5 by skimage.rescale (restated, anti-aliased), then 5 levels
    gg.ANTI_ALIASING[0] = True
    I0h, I1h, D0h = (orc.rescale(a, 0.5, anti_aliasing=True) for a in (I0, I1, D0))
    camh = cam * 0.5
    out["half_shape"] = np.array(I0h.shape)
    for name in ("None", "huber"):
        rec = run_pyramid(camh, camh, I0h, D0h, I1h, mode_arg(name, None), 5, True)
        put(out, f"half_aa_{name}", rec)
        print("real half", name, rec["evals"], rec["t"])
    return out


def capture_ill():
    """_PoseChangeEstimator at one level (per-update twists = what lstsq returned) and the
    3-level loop, on scenes whose J is rank deficient or badly conditioned."""
    import tadataka.vo.dvo as dvo
    from tadataka.camera import CameraModel, CameraParameters
    from tadataka.pose import Pose
    out = {}
    for scene in scenes.ILL_SCENES:
        pair = scenes.ill_pair(scene)
        cam = pair["cam"]
        cm = CameraModel(CameraParameters(cam[0:2], cam[2:4]), distortion_model=None)
        for name in ("None", "huber"):
            xis, svals = [], []
            orig_solve = dvo.solve_linear_equation

            def solve(J, r, weights_=None, **kw):
                xi = orig_solve(J, r, weights_, **kw)
                w = np.ones(len(r)) if weights_ is None else np.sqrt(weights_)
                svals.append(np.linalg.svd(J * w[:, None], compute_uv=False))
                xis.append(xi.copy())
                return xi

            dvo.solve_linear_equation = solve
            try:
                est = dvo._PoseChangeEstimator(cm, cm, max_iter=20)
                pose = est(pair["I0"], pair["D0"], pair["I1"], Pose.identity(), mode_arg(name, None))
            finally:
                dvo.solve_linear_equation = orig_solve
            tag = f"{scene}_{name}"
            out[f"{tag}_xis"] = np.array(xis).reshape(-1, 6)
            out[f"{tag}_svals"] = np.array(svals).reshape(-1, 6)
            out[f"{tag}_rotvec"] = pose.rotation.as_rotvec()
            out[f"{tag}_t"] = pose.t
            print("ill", scene, name, "updates", len(xis), "cond", svals[0][0] / max(svals[0][-1], 1e-300),
                  "xi0", xis[0])
            for aa in (False, True):
                rec = run_pyramid(cam, cam, pair["I0"], pair["D0"], pair["I1"], mode_arg(name, None), 3, aa)
                put(out, f"{scene}_{'aa' if aa else 'bl'}_{name}_pyr", rec)
    return out


def capture_holes():
    """Depth maps with missing readings: zeros and NaNs (tests/golden/scenes.py:holes_pair)."""
    out = {}
    for tag, fill in (("zero", 0.0), ("nan", np.nan)):
        pair = scenes.holes_pair(fill)
        for name in ("None", "huber", "tukey", "student-t"):
            for aa in (False, True):
                rec = run_pyramid(pair["cam"], pair["cam"], pair["I0"], pair["D0"], pair["I1"], mode_arg(name, None), 3, aa)
                put(out, f"{tag}_{'aa' if aa else 'bl'}_{name}", rec)
            print("holes", tag, name, rec["evals"], rec["t"], rec["rotvec"])
    return out


def main():
    gg.install_stubs()
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    jobs = {"dvo_examples.npz": capture_examples, "dvo_real.npz": capture_real, "dvo_ill.npz": capture_ill,
            "dvo_holes.npz": capture_holes}
    for fname, fn in jobs.items():
        if only and only not in fname:
            continue
        np.savez_compressed(os.path.join(HERE, fname), **fn())
        print(fname, os.path.getsize(os.path.join(HERE, fname)))


if __name__ == "__main__":
    main()
