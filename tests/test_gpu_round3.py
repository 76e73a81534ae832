"""GPU parity tests added in round 3: the reference's own coarse-to-fine loop at the
settings its examples and tests use (tests/golden/generate_golden_r3.py):

  * examples/dvo_pose_change.py   n_coarse_to_fine=5
  * examples/semi_dense_vo.py     n_coarse_to_fine=7, weights = 1 / variance map, 640x480
  * tests/vo/test_dvo.py          every weight option through the pyramid
  * real image statistics         two New-Tsukuba frames, all five weight options
  * ill-conditioned scenes        rank-deficient / badly scaled J against lstsq

Bars (north_star): 1e-6 on the recovered pose, 1e-4 relative on H / b / error sums
(held to 1e-9 where stated).  Everything goes through the C ABI."""
import os
import sys

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))

pytestmark = pytest.mark.gpu

POSE_ATOL = 1e-6
MODES = ("None", "huber", "student-t", "tukey", "map")


@pytest.fixture(scope="module")
def ops():
    from tadataka_amd import _lib, ops as o
    _lib.require_gpu()
    return o


def _pose12(T):
    return np.concatenate([T[:3, :3].ravel(), T[:3, 3]])


def _mode(ops, name):
    return ops.W_MAP if name == "map" else ops.WEIGHT_MODES[None if name == "None" else name]


def _pose_err(P12, rotvec, t):
    R = Rotation.from_rotvec(rotvec).as_matrix()
    return max(np.max(np.abs(P12[:9].reshape(3, 3) - R)), np.max(np.abs(P12[9:] - t)))


def _check_pyramid_run(ops, g, tag, I0, D0, I1, cam, name, wmap, n_levels, aa, atol=POSE_ATOL):
    """Fused coarse-to-fine call and the level-by-level chain against one reference record:
    final pose, pose after every level, PhotometricError evaluations per level."""
    H, W = I0.shape
    batch = ops.DvoBatch(1, H, W, n_levels=n_levels, ratio=1.5, with_weight_map=(name == "map"))
    batch.set_anti_aliasing(aa)
    batch.upload(0, I0, D0, I1, wmap if name == "map" else None)
    batch.build_pyramid()
    ident = _pose12(np.eye(4))[None]
    mode = _mode(ops, name)
    P, px = batch.estimate(cam, cam, ident, mode, 20)
    assert _pose_err(P[0], g[f"{tag}_rotvec"], g[f"{tag}_t"]) < atol, tag
    evals = g[f"{tag}_evals"]
    levels = list(range(n_levels - 1, -1, -1))
    shapes = [batch.level_shape(l) for l in levels]
    assert px == sum(int(e) * h * w for e, (h, w) in zip(evals, shapes)), tag
    Pl = ident
    for k, level in enumerate(levels):
        Pl, n_evals = batch.estimate_level(level, cam, cam, Pl, mode, 20)
        assert n_evals[0] == int(evals[k]), (tag, level)
        lp = g[f"{tag}_level_poses"][k]
        assert _pose_err(Pl[0], lp[:3], lp[3:]) < atol, (tag, level)
    assert np.array_equal(Pl, P)
    batch.close()
    return P[0]


# ---------------------------------------------------------------------------
# the examples' own settings
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("aa", [False, True])
@pytest.mark.parametrize("name", ["None", "huber"])
def test_example_dvo_pose_change_5_levels(ops, golden, aa, name):
    from tadataka_amd import synthetic
    g = golden("dvo_examples.npz")
    pair = synthetic.make_pair(240, 320, seed=5)
    _check_pyramid_run(ops, g, f"ex5_{'aa' if aa else 'bl'}_{name}", pair["I0"], pair["D0"], pair["I1"],
                       pair["cam"], name, None, 5, aa)


@pytest.mark.parametrize("aa", [False, True])
def test_example_semi_dense_vo_7_levels_weight_map(ops, golden, aa):
    """examples/semi_dense_vo.py:45-54 at 640x480: 7 levels (the coarsest is 42x56, its
    anti-aliasing filter has sigma 5.2 / radius 21) and weights = safe_invert(variance)."""
    import scenes
    from tadataka_amd import synthetic
    g = golden("dvo_examples.npz")
    pair = synthetic.make_pair(480, 640, seed=0)
    wmap = scenes.weight_map((480, 640), seed=41)
    _check_pyramid_run(ops, g, f"ex7_{'aa' if aa else 'bl'}_map", pair["I0"], pair["D0"], pair["I1"],
                       pair["cam"], "map", wmap, 7, aa)


@pytest.mark.parametrize("aa", [False, True])
@pytest.mark.parametrize("name", ["student-t", "tukey", "map"])
def test_cfg2_vga_3level_every_weight_option(ops, golden, aa, name):
    import scenes
    from tadataka_amd import synthetic
    g = golden("dvo_examples.npz")
    pair = synthetic.make_pair(480, 640, seed=0)
    wmap = scenes.weight_map((480, 640), seed=41)
    _check_pyramid_run(ops, g, f"v3_{'aa' if aa else 'bl'}_{name}", pair["I0"], pair["D0"], pair["I1"],
                       pair["cam"], name, wmap, 3, aa)


def test_examples_through_the_drop_in_api(ops, golden):
    """The same two calls as the examples write them: tadataka.vo.dvo.PoseChangeEstimator."""
    import scenes
    import tadataka_amd  # noqa: F401
    from tadataka.camera import CameraModel, CameraParameters
    from tadataka.numeric import safe_invert
    from tadataka.vo.dvo import PoseChangeEstimator
    from tadataka_amd import synthetic
    g = golden("dvo_examples.npz")
    pair = synthetic.make_pair(240, 320, seed=5)
    cm = CameraModel(CameraParameters(pair["cam"][0:2], pair["cam"][2:4]), distortion_model=None)
    pose = PoseChangeEstimator(cm, cm, n_coarse_to_fine=5)(pair["I0"], pair["D0"], pair["I1"], None)
    assert np.max(np.abs(pose.rotation.as_rotvec() - g["ex5_aa_None_rotvec"])) < POSE_ATOL
    assert np.max(np.abs(pose.t - g["ex5_aa_None_t"])) < POSE_ATOL
    pair = synthetic.make_pair(480, 640, seed=0)
    cm = CameraModel(CameraParameters(pair["cam"][0:2], pair["cam"][2:4]), distortion_model=None)
    wmap = scenes.weight_map((480, 640), seed=41)
    variance = 1.0 / wmap - np.finfo(np.float64).eps
    weights = safe_invert(variance)
    pose = PoseChangeEstimator(cm, cm, n_coarse_to_fine=7)(pair["I0"], pair["D0"], pair["I1"], weights)
    # safe_invert(1 / w - eps) is w up to an ulp
    assert np.max(np.abs(pose.rotation.as_rotvec() - g["ex7_aa_map_rotvec"])) < POSE_ATOL
    assert np.max(np.abs(pose.t - g["ex7_aa_map_t"])) < POSE_ATOL


# ---------------------------------------------------------------------------
# real image statistics
# ---------------------------------------------------------------------------
@pytest.fixture(scope="module")
def tsukuba(golden):
    import scenes
    g = golden("dvo_real.npz")
    I0 = scenes.gray_from_rgb_u8(g["rgb0"])
    I1 = scenes.gray_from_rgb_u8(g["rgb1"])
    D0 = scenes.tsukuba_depth(*I0.shape)
    return g, I0, D0, I1, scenes.weight_map(I0.shape, seed=42)


@pytest.mark.parametrize("aa", [False, True])
@pytest.mark.parametrize("name", MODES)
def test_new_tsukuba_frames_5_levels(ops, tsukuba, aa, name):
    """Two frames of the dataset tests/vo/test_dvo.py uses (depths in centimetres): edges,
    flat regions, specular noise -- not the analytic texture of the other fixtures."""
    import scenes
    g, I0, D0, I1, wmap = tsukuba
    P = _check_pyramid_run(ops, g, f"full_{'aa' if aa else 'bl'}_{name}", I0, D0, I1, scenes.TSUKUBA_CAM,
                           name, wmap, 5, aa)
    assert np.all(np.isfinite(P))


def test_new_tsukuba_rgb2gray_on_the_device(ops, tsukuba):
    """The device rgb2gray of the 8-bit frames equals the doubles the fixture was made from."""
    g, I0, _, I1, _ = tsukuba
    assert np.max(np.abs(ops.rgb2gray(g["rgb0"]) - I0)) < 1e-15
    assert np.max(np.abs(ops.rgb2gray(g["rgb1"]) - I1)) < 1e-15


@pytest.mark.parametrize("name", ["None", "huber"])
def test_new_tsukuba_example_half_resolution(ops, tsukuba, name):
    """examples/dvo_pose_change.py:22-31: rescale(I, 0.5), rescale(D, 0.5), camera.resize, then 5
    levels.  The half-resolution inputs are built on the device (tdk_rescale_anti_aliased)."""
    import scenes
    g, I0, D0, I1, _ = tsukuba
    I0h, D0h, I1h = (ops.rescale(a, 0.5, anti_aliasing=True) for a in (I0, D0, I1))
    assert tuple(g["half_shape"]) == I0h.shape
    _check_pyramid_run(ops, g, f"half_aa_{name}", I0h, D0h, I1h, scenes.TSUKUBA_CAM * 0.5, name, None, 5, True)


# ---------------------------------------------------------------------------
# ill-conditioned normal equations against the reference's lstsq on J
# ---------------------------------------------------------------------------
ILL = ("plane1d", "halfflat", "weaky", "weaky2", "diag2")


@pytest.mark.parametrize("scene", ILL)
@pytest.mark.parametrize("name", ["None", "huber"])
def test_ill_conditioned_scene_vs_lstsq(ops, golden, scene, name):
    """plane1d: a zero column of J (rank 5); diag2: two parallel columns of different norm
    (rank 5, the null direction is not an axis); weaky / weaky2: cond(J) 1.6e5 / 1.6e7;
    halfflat: more than half of the image without gradient.  The reference solves each
    update with gelsd on the n x 6 matrix; here the 6x6 system is solved on the device."""
    import scenes
    import tadataka_amd  # noqa: F401
    from tadataka.math import solve_normal_equations
    g = golden("dvo_ill.npz")
    pair = scenes.ill_pair(scene)
    cam = pair["cam"]
    H, W = pair["I0"].shape
    tag = f"{scene}_{name}"
    mode = _mode(ops, name)
    batch = ops.DvoBatch(1, H, W)
    batch.upload(0, pair["I0"], pair["D0"], pair["I1"])
    ident = _pose12(np.eye(4))[None]
    # first update: the twist lstsq returned for J at the identity
    ev = batch.evaluate(0, cam, cam, ident, mode)
    xi = solve_normal_equations(ops.upper21_to_matrix(ev["H"][0]), ev["b"][0], int(ev["n_update"][0]))
    xi_ref = g[f"{tag}_xis"][0]
    assert np.max(np.abs(xi - xi_ref)) < 1e-8 * max(1.0, np.max(np.abs(xi_ref))), (xi, xi_ref)
    # the device loop (its own 6x6 solve, tdk::solve6) against the reference loop
    P, n_evals = batch.estimate_level(0, cam, cam, ident, mode, 20)
    assert n_evals[0] == len(g[f"{tag}_xis"]) + 1
    assert _pose_err(P[0], g[f"{tag}_rotvec"], g[f"{tag}_t"]) < POSE_ATOL
    batch.close()
    for aa in (False, True):
        _check_pyramid_run(ops, g, f"{scene}_{'aa' if aa else 'bl'}_{name}_pyr", pair["I0"], pair["D0"],
                           pair["I1"], cam, name, None, 3, aa)


# ---------------------------------------------------------------------------
# anti-aliased pyramid at the examples' depth: 640x480 x 7 levels (radius up to 21)
# ---------------------------------------------------------------------------
def test_anti_aliased_pyramid_vga_7_levels_bit_exact(ops):
    from oracle import oracle as orc
    from tadataka_amd import synthetic
    B, H, W, L = 2, 480, 640, 7
    batch = ops.DvoBatch(B, H, W, n_levels=L, ratio=1.5, with_weight_map=True)
    pairs = []
    rng = np.random.default_rng(3)
    for i in range(B):
        pr = synthetic.make_pair(H, W, seed=90 + i)
        pr["W0"] = rng.uniform(0.05, 50.0, (H, W))
        batch.upload(i, pr["I0"], pr["D0"], pr["I1"], pr["W0"])
        pairs.append(pr)
    batch.build_pyramid()          # anti-aliased: the C-ABI default
    for level in range(1, L):
        scale = 1 / 1.5 ** level
        for i, name in ((0, "I0"), (0, "W0"), (1, "D0"), (1, "I1")):
            want = orc.rescale(pairs[i][name], scale, anti_aliasing=True)
            got = batch.download(i, level, name)
            assert got.shape == want.shape
            assert np.array_equal(got, want), (level, i, name, float(np.max(np.abs(got - want))))
    batch.close()
