#!/usr/bin/env python3
"""Round-4 fixture: the inputs and the reference-authored expectations of the reference's
own semi-dense integration test, tests/vo/semi_dense/test_semi_dense.py:41-135, on the
stereo pair dataset[0] of its New-Tsukuba sample (frame 00201, left = keyframe, right =
reference frame).  Runs in the build container only; the .npz travels.

  semi_dense_tsukuba.npz
      rgb_L, rgb_R      uint8 [480, 640, 3]: tests/dataset/new_tsukuba/illumination/daylight/
                        {left,right}/tsukuba_daylight_{L,R}_00201.png, alpha discarded
                        (new_tsukuba.py:86-87 discard_alpha)
      T_wk, T_wr        the 4x4 poses the reference's own loader builds: load_poses() +
                        calc_baseline_offset() + Pose(rotation, centre -/+ offset / 2).T
                        (tadataka/dataset/new_tsukuba.py:57-101,143-161), imported from
                        /root/reference and called here -- not restated
      cam               (fx, fy, ox, oy) = (615, 615, 320, 240) (new_tsukuba.py:99)
      est_params        Params of test_estimate (test_semi_dense.py:80-87)
      est_cases         rows (u_x, u_y, prior_depth, prior_variance, expected_flag): the five
                        assertions of test_estimate that need no depth map (:104-135); the
                        remaining two (:137-149) read the dataset's depth XMLs, which are not
                        in the checkout
      upd_params        Params of test_update_depth (:45-52); its maps are constants
                        (age 1, depth 200, variance 1: :66-68) and are rebuilt by the tests
      T_rk_lapack       inv(T_wr) @ T_wk with the inverse from LAPACK dgetrf + dgetri, which is
                        what ndarray_linalg::Inverse calls (src/semi_dense/semi_dense.rs:83-89);
                        computed here with scipy.linalg.lapack for the sensitivity test

The grey images are NOT stored: tests build them with scenes.gray_from_rgb_u8 (Rec.709
weights of skimage.color.rgb2gray on the uint8 data / 255).
Usage: python tests/golden/generate_golden_r4.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import generate_golden as gg          # noqa: E402  (stubs + reference import)

ROOT = "/root/reference/tests/dataset/new_tsukuba"


def load_rgb(path):
    from PIL import Image
    return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8))


def lapack_transform_rk(T_wk, T_wr):
    from scipy.linalg import lapack
    lu, piv, info = lapack.dgetrf(np.asfortranarray(T_wr))
    assert info == 0
    inv, info = lapack.dgetri(lu, piv)
    assert info == 0
    return np.ascontiguousarray(inv) @ T_wk


def capture_semi_dense_tsukuba():
    gg.install_stubs()
    # tadataka.dataset's __init__ pulls every dataset class (and with them cv2 / yaml code);
    # a bare namespace lets new_tsukuba.py itself import unmodified
    sys.modules["skimage.io"] = types.ModuleType("skimage.io")
    sys.modules["skimage.io"].imread = None
    ds = types.ModuleType("tadataka.dataset")
    ds.__path__ = ["/root/reference/tadataka/dataset"]
    sys.modules["tadataka.dataset"] = ds
    from tadataka.dataset.new_tsukuba import load_poses, calc_baseline_offset
    from tadataka.pose import Pose
    from tadataka.vo.semi_dense.flag import ResultFlag as FLAG

    rotations, positions = load_poses(os.path.join(ROOT, "groundtruth", "camera_track.txt"))
    index = 0                                        # dataset[0]: sorted paths -> frame 00201
    offset = calc_baseline_offset(rotations[index], 10.0)
    pose_wl = Pose(rotations[index], positions[index] - offset / 2.0)
    pose_wr = Pose(rotations[index], positions[index] + offset / 2.0)
    T_wk = np.ascontiguousarray(pose_wl.T)
    T_wr = np.ascontiguousarray(pose_wr.T)
    img = os.path.join(ROOT, "illumination", "daylight")
    est_cases = np.array([
        # u_x, u_y, prior depth, prior variance, expected flag       test_semi_dense.py
        [110, 400, -10.0, 10.0, FLAG.NEGATIVE_PRIOR_DEPTH],           # :104-108
        [110, 400, 0.05, 0.2, FLAG.HYPOTHESIS_OUT_OF_SERCH_RANGE],    # :110-114
        [390, 100, 2.0, 0.2, FLAG.INSUFFICIENT_GRADIENT],             # :116-120
        [0, 200, 2.0, 0.2, FLAG.KEY_OUT_OF_RANGE],                    # :122-127
        [116, 400, 2.0, 0.001, FLAG.REF_EPIPOLAR_TOO_SHORT],          # :129-134
    ], dtype=np.float64)
    out = dict(
        rgb_L=load_rgb(os.path.join(img, "left", "tsukuba_daylight_L_00201.png")),
        rgb_R=load_rgb(os.path.join(img, "right", "tsukuba_daylight_R_00201.png")),
        T_wk=T_wk, T_wr=T_wr, cam=np.array([615.0, 615.0, 320.0, 240.0]),
        est_params=np.array([0.1, 1000.0, 0.01, 0.01, 0.01, 0.2]),
        est_cases=est_cases,
        upd_params=np.array([60.0, 1000.0, 0.01, 0.01, 0.01, 0.2]),
        T_rk_lapack=lapack_transform_rk(T_wk, T_wr),
    )
    return out


def _sha(a):
    import hashlib
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


def capture_oracle_side(out):
    """What the oracle makes of the reference's inputs (frozen so that both sides prove they
    did the same work on the GPU box), and the inputs of the inverse-sensitivity test."""
    import scenes
    from scipy.spatial.transform import Rotation
    from oracle import oracle as orc
    from tadataka_amd import synthetic
    key = (out["cam"], scenes.gray_from_rgb_u8(out["rgb_L"]), out["T_wk"])
    ref = (out["cam"], scenes.gray_from_rgb_u8(out["rgb_R"]), out["T_wr"])
    H, W = key[1].shape
    p = orc.make_params(*out["upd_params"])
    d, v, f = orc.update_depth(key, [ref], np.ones((H, W), dtype=np.uint64), np.full((H, W), 200.0),
                               np.ones((H, W)), p)
    out["upd_flag_histogram"] = np.array([(f == -b).sum() for b in range(10)], dtype=np.int64)
    out["upd_sha_depth"], out["upd_sha_var"], out["upd_sha_flag"] = _sha(d), _sha(v), _sha(f)
    # SURVEY 8(d) cfg3 seen from a generic world frame: T_wk' = G T_wk, T_wr' = G T_wr.  T_rk is the
    # same transform on paper; inv4 and LAPACK now both round, differently.
    rng = np.random.default_rng(77)
    G = np.eye(4)
    G[:3, :3] = Rotation.from_rotvec(rng.uniform(-1, 1, 3)).as_matrix()
    G[:3, 3] = rng.uniform(-5, 5, 3)
    c = synthetic.make_semi_dense_case(480, 640, seed=1)
    out["moved_G"] = G
    out["moved_T_rk_lapack"] = lapack_transform_rk(G @ c["T_wk"], G @ c["T_wr"])
    out["cfg3_T_rk_lapack"] = lapack_transform_rk(c["T_wk"], c["T_wr"])
    print("tsukuba update_depth histogram", out["upd_flag_histogram"])


def main():
    out = capture_semi_dense_tsukuba()
    capture_oracle_side(out)
    path = os.path.join(HERE, "semi_dense_tsukuba.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path))
    for k, v in out.items():
        print(k, v.shape, v.dtype)


if __name__ == "__main__":
    main()
