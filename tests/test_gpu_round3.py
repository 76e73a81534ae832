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
def test_anti_aliased_pyramid_vga_7_levels(ops):
    """Both evaluations of the anti-aliased levels against the oracle at 640x480 x 7 levels (weights up
    to 50, so absolute errors scale): the ndimage operation order bit for bit (levels 3+ run the
    generic-radius tiles / the per-pixel kernel), the folded tap lists to the last bits; and a 7-level
    batch mixes the two (deep levels whose tiles exceed LDS stay on the ndimage-order kernels)."""
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
    want = {}
    for level in range(1, L):
        for i, name in ((0, "I0"), (0, "W0"), (1, "D0"), (1, "I1")):
            want[(level, i, name)] = orc.rescale(pairs[i][name], 1 / 1.5 ** level, anti_aliasing=True)
    batch.set_anti_aliasing(True)
    batch.build_pyramid()
    for (level, i, name), w in want.items():
        got = batch.download(i, level, name)
        assert got.shape == w.shape
        assert np.array_equal(got, w), (level, i, name, float(np.max(np.abs(got - w))))
    batch.close()


# ---------------------------------------------------------------------------
# tdk_dvo_set_student_passes(2): the IEEE-division variant of the Student-t variance fixed point
# ---------------------------------------------------------------------------
_STUDENT_EXACT_SCRIPT = """
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import b6_err, h21_err
from tadataka_amd import ops
from scipy.spatial.transform import Rotation
d = np.load("tests/golden/dvo_small.npz")
cam = d["cam"]; H, W = d["I0"].shape
batch = ops.DvoBatch(1, H, W)
batch.set_student_passes(int(sys.argv[1]))
batch.upload(0, d["I0"], d["D0"], d["I1"])
iu = np.triu_indices(6)
worst = 0.0
for k in range(int(d["s_student-t_n_updates"])):
    T = d["s_student-t_err_T"][k]
    ev = batch.evaluate(0, cam, cam, np.concatenate([T[:3, :3].ravel(), T[:3, 3]])[None], ops.W_STUDENT_T)
    assert ev["n_update"][0] == int(d[f"s_student-t_u{k}_n_valid"])
    worst = max(worst, h21_err(ev["H"][0], d[f"s_student-t_u{k}_H"][iu]),
                b6_err(ev["b"][0], d[f"s_student-t_u{k}_b"], d[f"s_student-t_u{k}_H"][iu]))
P, n = batch.estimate_level(0, cam, cam, np.concatenate([np.eye(3).ravel(), np.zeros(3)])[None], ops.W_STUDENT_T, 20)
R = Rotation.from_rotvec(d["s_student-t_final_rotvec"]).as_matrix()
perr = max(np.max(np.abs(P[0, :9].reshape(3, 3) - R)), np.max(np.abs(P[0, 9:] - d["s_student-t_final_t"])))
assert n[0] == int(d["s_student-t_n_updates"]) + 1
print("RESULT", worst, perr)
"""


@pytest.mark.parametrize("passes", ["0", "1", "2"])
def test_student_t_exact_switch(passes):
    """The arithmetic variants of the Student-t fixed point against the reference's own per-iteration normal
    equations (dvo_small.npz): tdk_dvo_set_student_passes 0 (two Taylor passes, default), 1 (nine sequential
    passes, reciprocal arithmetic), 2 (IEEE divisions, the CPU restatement's operations)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    out = subprocess.run([sys.executable, "-c", _STUDENT_EXACT_SCRIPT, passes], env=env, cwd=root, check=True,
                         capture_output=True, text=True, timeout=300)
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][-1].split()
    worst, perr = float(line[1]), float(line[2])
    assert worst < 1e-9 and perr < POSE_ATOL, (worst, perr)


# ---------------------------------------------------------------------------
# Student-t variance: ten fixed-point steps from two Taylor passes (k_student_taylor) vs the nine sequential ones
# ---------------------------------------------------------------------------
def _student_scenes():
    from golden import scenes
    from tadataka_amd import synthetic
    out = []
    pr = synthetic.make_pair(480, 640, seed=0)
    out.append(("vga", pr["I0"], pr["D0"], pr["I1"], pr["cam"]))
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "dvo_real.npz"))
    I0, I1 = scenes.gray_from_rgb_u8(g["rgb0"]), scenes.gray_from_rgb_u8(g["rgb1"])
    out.append(("tsukuba", I0, scenes.tsukuba_depth(*I0.shape), I1, scenes.TSUKUBA_CAM))
    hp = scenes.holes_pair(0.0, 480, 640)
    out.append(("holes", hp["I0"], hp["D0"], hp["I1"], hp["cam"]))
    # heavy tails: a tenth of the target frame replaced by noise -- the case the t distribution is there for
    rng = np.random.default_rng(12)
    I1o = pr["I1"].copy()
    I1o[rng.random(I1o.shape) < 0.1] = rng.random()
    out.append(("outliers", pr["I0"], pr["D0"], I1o, pr["cam"]))
    return out


def test_student_t_variance_taylor_passes(ops):
    """The variance after ten steps (weights.py:4-16) of the default scheme -- two passes that expand every step
    around a predicted iterate -- against the nine sequential passes (reciprocal and IEEE arithmetic) and against
    the fixed point iterated in NumPy on the oracle's masked residuals; every pyramid level (the coarse ones
    fit the sample entirely), several poses, real frames, depth holes, heavy tails."""
    from oracle import oracle as orc
    from tadataka_amd import synthetic
    sys.path.insert(0, os.path.dirname(__file__))
    scn = _student_scenes()
    B, H, W, L = len(scn), 480, 640, 5
    batch = ops.DvoBatch(B, H, W, n_levels=L, ratio=1.5)
    for i, (_, I0, D0, I1, _) in enumerate(scn):
        batch.upload(i, I0, D0, I1)
    batch.build_pyramid()
    cams = np.stack([np.asarray(c[4], dtype=np.float64) for c in scn])
    rng = np.random.default_rng(5)
    poses = [_pose12(np.eye(4))]
    for scale in (1.0, 6.0):
        om, t = synthetic.random_pose(rng, 0.004 * scale, 0.008 * scale)
        T = np.eye(4); T[:3, :3] = synthetic.rodrigues(om); T[:3, 3] = t
        poses.append(_pose12(T))
    worst_seq = worst_np = 0.0
    for level in range(L):
        for P in poses:
            PP = np.tile(P, (B, 1))
            got = {}
            for mode in (0, 1, 2):
                batch.set_student_passes(mode)
                ev = batch.evaluate(level, cams, cams, PP, ops.W_STUDENT_T)
                got[mode] = (batch.robust_scale(), ev)
            v0, v1, v2 = got[0][0], got[1][0], got[2][0]
            assert np.all(np.isfinite(v1)) and np.all(v1 > 0)
            worst_seq = max(worst_seq, float(np.max(np.abs(v0 - v1) / v1)), float(np.max(np.abs(v0 - v2) / v2)))
            # the normal equations the three variances lead to
            for k in ("H", "b"):
                a, b = got[0][1][k], got[1][1][k]
                assert np.max(np.abs(a - b)) <= 1e-10 * np.max(np.abs(b)), (level, k)
            assert np.array_equal(got[0][1]["n_update"], got[1][1]["n_update"])
            if level >= 2:      # NumPy fixed point on the oracle's masked residuals (small levels: seconds)
                s_cam = 1 / pow(1.5, level)
                for i, (_, I0, D0, I1, cam) in enumerate(scn[:2]):
                    l0, ld, l1 = (batch.download(i, level, n) for n in ("I0", "D0", "I1"))
                    gx, gy = orc.image_gradient(l1)
                    cam_l = np.asarray(cam) * s_cam
                    _, r, _ = orc.dvo_rows(l0, ld, l1, gx, gy, cam_l, cam_l, P[:9].reshape(3, 3), P[9:], None)
                    s, v = r * r, 1.0
                    for _ in range(10):
                        v = np.mean(s * 6.0 / (5.0 + s / v))
                    worst_np = max(worst_np, abs(v0[i] - v) / v)
    redos = batch.student_redos()
    batch.close()
    print("student-t Taylor vs sequential", worst_seq, "vs numpy", worst_np, "third passes", redos)
    assert worst_seq < 1e-12, worst_seq
    assert worst_np < 1e-11, worst_np


def test_student_t_taylor_passes_many_small_pairs(ops):
    """More pairs than one resident round of blocks (one block per pair and pass) and levels smaller than the
    sample (24x32, 16x21): the Taylor passes against the sequential ones."""
    from tadataka_amd import synthetic
    B, H, W = 1100, 24, 32
    batch = ops.DvoBatch(B, H, W, n_levels=2, ratio=1.5)
    cam = synthetic.camera_for(W, H)
    true = np.empty((B, 12))
    for i in range(B):
        om, t = synthetic.random_pose(np.random.default_rng(100 + i))
        true[i, :9] = synthetic.rodrigues(om).ravel(); true[i, 9:] = t
    batch.fill_synthetic(cam, true, seed0=5, noise=0.05)
    batch.build_pyramid()
    P = np.tile(_pose12(np.eye(4)), (B, 1))
    for level in (0, 1):
        got = []
        for mode in (0, 1):
            batch.set_student_passes(mode)
            batch.evaluate(level, cam, cam, P, ops.W_STUDENT_T)
            got.append(batch.robust_scale())
        assert np.all(got[1] > 0)
        assert np.max(np.abs(got[0] - got[1]) / got[1]) < 1e-12, level
        assert len(np.unique(got[1])) > B // 2        # the pairs are different scenes
    batch.close()


def test_student_t_degenerate_inputs(ops):
    """Identical frames (every residual zero): the reference's second step divides 0 by 0 (weights.py:13) -- the
    variance is NaN in every scheme, no hang, no exception; and an empty update mask likewise."""
    from tadataka_amd import synthetic
    pr = synthetic.make_pair(120, 160, seed=3)
    batch = ops.DvoBatch(2, 120, 160)
    batch.upload(0, pr["I1"], pr["D0"], pr["I1"])
    batch.upload(1, pr["I0"], -np.abs(pr["D0"]), pr["I1"])       # every point behind the camera
    P = np.tile(_pose12(np.eye(4)), (2, 1))
    for mode in (0, 1, 2):
        batch.set_student_passes(mode)
        ev = batch.evaluate(0, pr["cam"], pr["cam"], P, ops.W_STUDENT_T)
        v = batch.robust_scale()
        assert np.isnan(v[0]), (mode, v)
        assert ev["n_update"][1] == 0 and np.isnan(v[1]), (mode, v, ev["n_update"])
    batch.close()


# ---------------------------------------------------------------------------
# error-only entry and the SURVEY 8(d) work counters
# ---------------------------------------------------------------------------
def test_photometric_error_entry_and_counts(ops, golden):
    """tdk_dvo_photometric_error (probe body) returns the double tdk_dvo_evaluate returns for the
    same pose, and equals the reference's PhotometricError; tdk_dvo_get_counts reports n updates and
    n + 1 errors per level as the reference loop executes them."""
    from tadataka_amd import synthetic
    d = golden("dvo_small.npz")
    cam = d["cam"]
    H, W = d["I0"].shape
    batch = ops.DvoBatch(1, H, W)
    batch.upload(0, d["I0"], d["D0"], d["I1"])
    for T, val in zip(d["s_huber_err_T"], d["s_huber_err_val"]):
        P = _pose12(T)[None]
        ss, ne = batch.photometric_error(0, cam, cam, P)
        for mode in (ops.W_NONE, ops.W_HUBER):
            ev = batch.evaluate(0, cam, cam, P, mode)
            assert ss[0] == ev["sum_sq"][0] and ne[0] == ev["n_error"][0]
        assert abs(ss[0] / ne[0] - val) <= 1e-9 * abs(val)
    P, n_evals = batch.estimate_level(0, cam, cam, _pose12(np.eye(4))[None], ops.W_HUBER, 20)
    e_px, u_px = batch.counts()
    assert e_px == int(n_evals[0]) * H * W and u_px == int(d["s_huber_n_updates"]) * H * W
    batch.close()
    g = golden("dvo_vga_pyramid.npz")
    pair = synthetic.make_pair(480, 640, seed=0)
    batch = ops.DvoBatch(1, 480, 640, n_levels=3, ratio=1.5)
    batch.upload(0, pair["I0"], pair["D0"], pair["I1"])
    batch.build_pyramid()
    P, px = batch.estimate(pair["cam"], pair["cam"], _pose12(np.eye(4))[None], ops.W_HUBER, 20)
    e_px, u_px = batch.counts()
    shapes = [batch.level_shape(l) for l in (2, 1, 0)]
    evals = g["pyr_aa_huber_evals"]
    assert e_px == px == sum(int(e) * h * w for e, (h, w) in zip(evals, shapes))
    assert u_px == sum((int(e) - 1) * h * w for e, (h, w) in zip(evals, shapes))
    batch.close()


# ---------------------------------------------------------------------------
# examples/semi_dense_vo.py:152-207 through the UNCHANGED drop-in calls, maps resident on the device
# ---------------------------------------------------------------------------
def _sliding_camera_frames(H, W, n_frames):
    from tadataka_amd import synthetic
    return synthetic.make_track(H, W, n_frames)


def test_semi_dense_vo_example_loop_through_rust_bindings(ops, monkeypatch, device_maps):
    """The loop body of examples/semi_dense_vo.py (dvo -> Frame -> increment_age -> propagate ->
    update_depth -> refframes.append -> hand the maps over), written with the example's own calls
    and names, on the drop-in packages.  Bit-exact against the oracle chain fed with the same
    transforms; tracking against the oracle's coarse-to-fine loop; and no map crosses PCIe inside
    the loop: the only uploads are the new images, the only downloads the ones the test asks for."""
    import tadataka_amd  # noqa: F401
    from oracle import oracle as orc
    from rust_bindings.camera import CameraParameters
    from rust_bindings.semi_dense import Frame, Params, increment_age, propagate, update_depth
    from tadataka.camera import CameraModel
    from tadataka.matrix import inv_motion_matrix
    from tadataka.numeric import safe_invert
    from tadataka.vo.dvo import PoseChangeEstimator

    H, W, n_frames, n_levels = 96, 128, 5, 2
    cam, depth_gt0, T_w, images = _sliding_camera_frames(H, W, n_frames)
    default_depth, default_variance, uncertaintity_bias = 1.0, 10.0, 0.01
    pargs = (0.5, 10.0, 0.01, 0.01, 0.004, 0.01)
    params = Params(*pargs)
    po = orc.make_params(*pargs)

    def dvo(camera_params0, camera_params1, image0, image1, depth_map0, variance_map0):
        estimator = PoseChangeEstimator(CameraModel(camera_params0, distortion_model=None),
                                        CameraModel(camera_params1, distortion_model=None),
                                        n_coarse_to_fine=n_levels)
        weights = safe_invert(variance_map0)
        pose10 = estimator(image0, depth_map0, image1, weights)
        return pose10.T

    def calc_pose_w1(transform10, transform_w0):
        return transform_w0.dot(inv_motion_matrix(transform10))

    traffic = {"up": 0, "down": 0, "frames": 0, "dvo_host": 0}
    real_call = ops.call

    def counting_call(name, *args):
        if name == "tdk_map_download":
            traffic["down"] += 1
        elif name == "tdk_map_upload" or (name == "tdk_map_create" and args[2] is not None):
            traffic["up"] += 1
        elif name == "tdk_frame_create":
            traffic["frames"] += 1
        elif name == "tdk_dvo_upload":
            traffic["dvo_host"] += 3
        elif name == "tdk_dvo_upload_mixed":
            traffic["dvo_host"] += sum(1 for k in range(4) if args[2][k])
        return real_call(name, *args)
    monkeypatch.setattr(ops, "call", counting_call)

    camera_params0 = CameraParameters((cam[0], cam[1]), (cam[2], cam[3]))
    frame0 = Frame(camera_params0, images[0], T_w[0])
    refframes = [frame0]
    rng = np.random.default_rng(0)
    depth_map0 = depth_gt0 * rng.uniform(0.95, 1.05, (H, W))
    variance_map0 = np.full((H, W), 0.05)
    age0 = np.zeros((H, W), dtype=np.uint64)
    # the oracle's copy of the state
    o_depth, o_var, o_age = depth_map0.copy(), variance_map0.copy(), age0.copy()
    o_frames = [(cam, images[0], T_w[0])]
    checks = []

    for i in range(1, n_frames):
        camera_params1, image1 = camera_params0, images[i]
        # ---- the example's loop body (examples/semi_dense_vo.py:175-204, plot() left out) ----
        transform10 = dvo(frame0.camera_params, camera_params1,
                          frame0.image, image1, depth_map0, variance_map0)

        transform_w1 = calc_pose_w1(transform10, frame0.transform_wf)
        frame1 = Frame(camera_params1, image1, transform_w1)

        age1 = increment_age(age0, frame0.camera_params, frame1.camera_params,
                             transform10, depth_map0)

        depth_map1, variance_map1 = propagate(
            transform10, frame0.camera_params, frame1.camera_params,
            depth_map0, variance_map0,
            default_depth, default_variance, uncertaintity_bias
        )
        depth_map1, variance_map1, flag_map = update_depth(
            frame1, refframes, age1,
            depth_map1, variance_map1, params
        )
        refframes.append(frame1)

        depth_map0, variance_map0, age0 = depth_map1, variance_map1, age1
        frame0 = frame1
        # ---- end of the loop body ----
        checks.append((transform10, transform_w1, depth_map1, variance_map1, age1, flag_map))

    in_loop = dict(traffic)
    # inside the loop: the caller's initial host maps went up in frame 1 (age0 + depth_map0 into
    # increment_age, depth_map0 + variance_map0 into propagate), after that no map moves either way
    assert in_loop["up"] == 4 and in_loop["down"] == 0, in_loop
    assert in_loop["frames"] == n_frames                       # every image uploaded exactly once as a Frame
    # DVO took I1 from the host every frame (the new image) and, in frame 1 only, depth / weights too
    assert in_loop["dvo_host"] == (n_frames - 1) + 2, in_loop

    for i, (T10, T_w1, d1, v1, a1, f1) in enumerate(checks, start=1):
        rot, t = orc.dvo_estimate(images[i - 1], o_depth, images[i], cam, cam, weights=1.0 / (o_var + 1e-16),
                                  n_coarse_to_fine=n_levels, max_iter=20, anti_aliasing=True)
        assert np.max(np.abs(T10[:3, :3] - rot.as_matrix())) < POSE_ATOL
        assert np.max(np.abs(T10[:3, 3] - t)) < POSE_ATOL
        key = (cam, images[i], T_w1)
        o_depth, o_var, o_age, o_flag = orc.semi_dense_step(key, cam, o_frames, T10, o_age, o_depth, o_var, po,
                                                            default_depth, default_variance, uncertaintity_bias)
        o_frames.append(key)
        assert a1.dtype == np.uint64 and f1.dtype == np.int64 and d1.shape == (H, W)
        assert np.array_equal(a1, o_age) and np.array_equal(f1, o_flag)
        assert np.array_equal(d1, o_depth) and np.array_equal(v1, o_var)
    assert int(o_age.max()) == n_frames - 1 and int((o_flag == 0).sum()) > 500


def test_update_depth_maps_age_check_without_a_wait(ops):
    """tdk_update_depth_maps has to report an age beyond the reference frames (the reference exits there,
    semi_dense.rs:202-205).  It knows a host-side bound of every age map (maximum of an uploaded array, + 1 per
    increment_age) and reads the device flag only when that bound does not already rule the error out: the
    error is still raised whenever an age exceeds the list -- from an uploaded map, from a chain of increments --
    results are the same on both paths, and the loop of the example (ages grow with the list) never takes the
    waiting path."""
    import tadataka_amd  # noqa: F401
    from rust_bindings.camera import CameraParameters
    from rust_bindings.semi_dense import Frame, Params, increment_age, propagate, update_depth
    from tadataka.matrix import inv_motion_matrix
    from tadataka_amd import _lib
    H, W, n_frames = 96, 128, 5
    cam, depth_gt0, T_w, images = _sliding_camera_frames(H, W, n_frames)
    cp = CameraParameters((cam[0], cam[1]), (cam[2], cam[3]))
    params = Params(0.5, 10.0, 0.01, 0.01, 0.004, 0.01)
    frames = [Frame(cp, images[i], T_w[i]) for i in range(n_frames)]
    T10 = [None] + [np.dot(inv_motion_matrix(T_w[i]), T_w[i - 1]) for i in range(1, n_frames)]
    d0, v0 = depth_gt0.copy(), np.full((H, W), 0.05)

    # (1) an uploaded age map that exceeds the list: raised (bound 2 > 1 reference frame -> device check)
    with pytest.raises(_lib.TdkError) as e:
        update_depth(frames[1], [frames[0]], np.full((H, W), 2, dtype=np.uint64), d0, v0, params)
    assert e.value.status == _lib.TDK_ERR_AGE_EXCEEDS_REFFRAMES
    # ... an uploaded map whose bound exceeds the list only at pixels the check never sees is impossible: every
    # pixel with age != 0 is checked, so bound > n_ref with all ages checked in range means the bound was loose:
    ages = np.zeros((H, W), dtype=np.uint64); ages[3, 4] = 1
    d_a, v_a, f_a = (np.asarray(m) for m in update_depth(frames[1], [frames[0]], ages, d0, v0, params))
    assert f_a[3, 4] != -9 and (f_a == -9).sum() == H * W - 1

    # (2) a chain of increments with a list that stops growing: frame 2 has ages of 2 against one reference frame
    a = np.zeros((H, W), dtype=np.uint64)
    a = increment_age(a, cp, cp, T10[1], d0)
    a = increment_age(a, cp, cp, T10[2], d0)
    assert int(np.asarray(a).max()) == 2
    with pytest.raises(_lib.TdkError) as e:
        update_depth(frames[2], [frames[1]], a, d0, v0, params)
    assert e.value.status == _lib.TDK_ERR_AGE_EXCEEDS_REFFRAMES
    # a loose bound (two increments, but the second warp lands nowhere: every age is 0) is not an error
    far = T10[2].copy(); far[:3, 3] = (1e6, 0.0, 0.0)
    b = increment_age(increment_age(np.zeros((H, W), dtype=np.uint64), cp, cp, T10[1], d0), cp, cp, far, d0)
    assert int(np.asarray(b).max()) == 0
    _, _, f_b = update_depth(frames[2], [frames[1]], b, d0, v0, params)
    assert np.all(np.asarray(f_b) == -9)

    # (3) the example's loop: no device round trip for the check, same maps as with the bound unknown
    def run(force_unknown):
        a0, dm, vm, refs, out = np.zeros((H, W), dtype=np.uint64), d0, v0, [frames[0]], []
        for i in range(1, n_frames):
            a1 = increment_age(a0, cp, cp, T10[i], dm)
            d1, v1 = propagate(T10[i], cp, cp, dm, vm, 1.0, 10.0, 0.01)
            if force_unknown:       # an age map that went through the host loses nothing but its provenance
                a1 = np.asarray(a1) + np.uint64(0)
                a1[0, 0] = np.uint64(min(int(a1[0, 0]), len(refs)))
            d1, v1, f1 = update_depth(frames[i], refs, a1, d1, v1, params)
            refs.append(frames[i]); a0, dm, vm = a1, d1, v1
            out.append([np.asarray(m).copy() for m in (a1, d1, v1, f1)])
        return out
    fast, slow = run(False), run(True)
    for fa, sl in zip(fast, slow):
        for x, y in zip(fa, sl):
            assert np.array_equal(x, y, equal_nan=True)
    assert int(fast[-1][0].max()) == n_frames - 1


def test_device_map_behaves_like_an_array(ops, device_maps):
    """What a caller may do with a returned map: look at it, compute with it, write into it and hand
    it back (the device copy follows), mix it with ndarrays."""
    from rust_bindings.camera import CameraParameters
    from rust_bindings.semi_dense import increment_age
    from oracle import oracle as orc
    from tadataka_amd import synthetic
    H, W = 40, 56
    c = synthetic.make_semi_dense_case(H, W, seed=12)
    cam = c["cam"]
    cp = CameraParameters((cam[0], cam[1]), (cam[2], cam[3]))
    T10 = np.linalg.inv(c["T_wk"]) @ c["T_wr"]
    a1 = increment_age(c["age"], cp, cp, T10, c["prior_depth"])
    assert isinstance(a1, ops.DeviceMap) and a1.shape == (H, W) and a1.ndim == 2 and len(a1) == H
    ref1 = orc.increment_age(c["age"], cam, cam, T10, c["prior_depth"])
    assert np.array_equal(np.asarray(a1), ref1) and int(a1.max()) == int(ref1.max())
    assert np.array_equal(a1 + 1, ref1 + 1) and np.array_equal(a1[3:5], ref1[3:5])
    assert np.array_equal(np.where(a1 > 0, 1, 0), np.where(ref1 > 0, 1, 0))
    # edit on the host, hand back: the device copy is refreshed
    a1[0, :] = 7
    ref1[0, :] = 7
    a2 = increment_age(a1, cp, cp, T10, c["prior_depth"])
    assert np.array_equal(a2, orc.increment_age(ref1, cam, cam, T10, c["prior_depth"]))
    # the default (without tadataka_amd.enable_device_maps()): plain ndarrays, as the reference returns
    import tadataka_amd
    tadataka_amd.enable_device_maps(False)
    try:
        a3 = increment_age(c["age"], cp, cp, T10, c["prior_depth"])
        assert type(a3) is np.ndarray and np.array_equal(a3, orc.increment_age(c["age"], cam, cam, T10, c["prior_depth"]))
        assert isinstance(a3, np.ndarray)
    finally:
        tadataka_amd.enable_device_maps(True)
    with pytest.raises(TypeError):
        increment_age(c["age"].astype(np.int64), cp, cp, T10, c["prior_depth"])


@pytest.mark.parametrize("mode", ["aa", "skimage", "bilinear"])
def test_partial_pyramid_rebuild(ops, mode):
    """tdk_dvo_build_pyramid_arrays: the levels of the named arrays are rebuilt, bit for bit as the full build
    produces them, the others are left alone (a stream that replaces I1 per step rebuilds only I1)."""
    from tadataka_amd import synthetic
    B, H, W, L = 3, 120, 160, 4
    batch = ops.DvoBatch(B, H, W, n_levels=L, ratio=1.5, with_weight_map=True)
    if mode == "skimage":
        batch.set_skimage_pyramid()        # level 0 is a level of its own here: the uploads go beside it
    else:
        batch.set_anti_aliasing(mode == "aa")
    rng = np.random.default_rng(2)
    pairs = [synthetic.make_pair(H, W, seed=30 + i) for i in range(B)]
    W0s = [rng.uniform(0.1, 5.0, (H, W)) for _ in range(B)]
    for i, p in enumerate(pairs):
        batch.upload(i, p["I0"], p["D0"], p["I1"], W0s[i])
    batch.build_pyramid()
    names = ("I0", "D0", "I1", "W0")
    before = {(i, l, n): batch.download(i, l, n) for i in range(B) for l in range(1, L) for n in names}
    # new frames for I1 (and, second round, new depth + weights); nothing is rebuilt yet
    new_I1 = [synthetic.make_pair(H, W, seed=60 + i)["I1"] for i in range(B)]
    for i, p in enumerate(pairs):
        batch.upload(i, p["I0"], p["D0"], new_I1[i], W0s[i])
    batch.build_pyramid(["I1"])
    for i in range(B):
        for l in range(1, L):
            for n in ("I0", "D0", "W0"):
                assert np.array_equal(batch.download(i, l, n), before[(i, l, n)]), (i, l, n)
            assert not np.array_equal(batch.download(i, l, "I1"), before[(i, l, "I1")])
    part = {(i, l): batch.download(i, l, "I1") for i in range(B) for l in range(1, L)}
    batch.build_pyramid()                                  # the full build of the same contents
    for i in range(B):
        for l in range(1, L):
            assert np.array_equal(batch.download(i, l, "I1"), part[(i, l)]), (i, l)
    # two arrays at once, W0 among them
    new_D0 = [p["D0"] * 1.25 for p in pairs]
    new_W0 = [rng.uniform(0.1, 5.0, (H, W)) for _ in range(B)]
    for i, p in enumerate(pairs):
        batch.upload(i, p["I0"], new_D0[i], new_I1[i], new_W0[i])
    batch.build_pyramid(["D0", "W0"])
    part = {(i, l, n): batch.download(i, l, n) for i in range(B) for l in range(1, L) for n in names}
    batch.build_pyramid()
    for key, v in part.items():
        assert np.array_equal(batch.download(*key), v), key
    with pytest.raises(Exception):
        batch.build_pyramid([])                            # an empty selection is a caller's mistake
    batch.close()
    plain = ops.DvoBatch(1, H, W, n_levels=2, ratio=1.5)
    plain.upload(0, pairs[0]["I0"], pairs[0]["D0"], pairs[0]["I1"])
    with pytest.raises(Exception):
        plain.build_pyramid(["W0"])                        # no weight map in this batch
    plain.close()


def test_async_uploads_from_pinned_memory(ops):
    """tdk_dvo_upload_async / _u8: ranges of pairs from pinned memory on the copy stream, ordered
    with the batch's own stream; float64 bit for bit, 8-bit frames as x * (1 / 255) (img_as_float)."""
    from tadataka_amd import _lib, synthetic
    B, H, W = 5, 37, 53            # odd pixel count: the per-pair stride is padded
    batch = ops.DvoBatch(B, H, W)
    rng = np.random.default_rng(8)
    pairs = [synthetic.make_pair(H, W, seed=40 + i) for i in range(B)]
    for i, pr in enumerate(pairs):
        batch.upload(i, pr["I0"], pr["D0"], pr["I1"])
    new_I1 = rng.uniform(0, 1, (3, H, W))
    pin = ops.PinnedBuffer((3, H * W))
    pin.array[:] = new_I1.reshape(3, -1)
    batch.upload_async("I1", 1, 3, pin)                    # pairs 1..3
    gray = rng.integers(0, 256, (2, H, W), dtype=np.uint8)
    pin8 = ops.PinnedBuffer((2, H * W), dtype=np.uint8)
    pin8.array[:] = gray.reshape(2, -1)
    batch.upload_async("I0", 3, 2, pin8)                   # pairs 3..4
    _lib.call("tdk_sync")
    for i in range(B):
        want_I1 = new_I1[i - 1] if 1 <= i <= 3 else pairs[i]["I1"]
        want_I0 = gray[i - 3] * (1.0 / 255.0) if i >= 3 else pairs[i]["I0"]
        assert np.array_equal(batch.download(i, 0, "I1"), want_I1), i
        assert np.array_equal(batch.download(i, 0, "I0"), want_I0), i
        assert np.array_equal(batch.download(i, 0, "D0"), pairs[i]["D0"]), i
    # the estimation that follows an async upload sees the new frames (stream order, no host wait)
    cam = pairs[0]["cam"]
    ident = np.tile(_pose12(np.eye(4)), (B, 1))
    pin.array[:] = np.stack([pairs[i]["I1"].ravel() for i in (1, 2, 3)])
    batch.upload_async("I1", 1, 3, pin)
    ev = batch.evaluate(0, cam, cam, ident, ops.W_HUBER)
    single = ops.DvoBatch(1, H, W)
    single.upload(0, pairs[2]["I0"], pairs[2]["D0"], pairs[2]["I1"])
    ev1 = single.evaluate(0, cam, cam, ident[:1], ops.W_HUBER)
    assert np.array_equal(ev["H"][2], ev1["H"][0]) and ev["sum_sq"][2] == ev1["sum_sq"][0]
    single.close(); batch.close(); pin.close(); pin8.close()


def test_async_upload_is_waited_for_by_whatever_touches_the_arrays_next(ops):
    """The batch's stream waits for an asynchronous upload when it next touches the arrays, not when the upload is
    queued: a transfer long enough to still be in flight (48 VGA frames, 8-bit and float64) followed AT ONCE by a
    download, a partial pyramid build and an estimation -- each must see the new frames."""
    from tadataka_amd import _lib, synthetic
    B, H, W = 48, 480, 640
    cam = synthetic.camera_for(W, H)
    rng = np.random.default_rng(21)
    batch = ops.DvoBatch(B, H, W, n_levels=2, ratio=1.5)
    ref = ops.DvoBatch(1, H, W, n_levels=2, ratio=1.5)
    pr = synthetic.make_pair(H, W, seed=3)
    for i in range(B):
        batch.upload(i, pr["I0"], pr["D0"], pr["I1"])
    batch.build_pyramid()
    ident = np.tile(_pose12(np.eye(4)), (B, 1))
    gray = rng.integers(0, 256, (B, H * W), dtype=np.uint8)
    pin8 = ops.PinnedBuffer((B, H * W), dtype=np.uint8)
    pin8.array[:] = gray
    f64 = rng.uniform(0.0, 1.0, (B, H * W))
    pin = ops.PinnedBuffer((B, H * W))
    pin.array[:] = f64
    for which, buf, want in (("I1", pin8, gray * (1.0 / 255.0)), ("I1", pin, f64)):
        # download straight after the upload call
        batch.upload_async(which, 0, B, buf)
        assert np.array_equal(batch.download(B - 1, 0, which).ravel(), want[B - 1])
        # partial pyramid straight after the upload call
        batch.upload_async(which, 0, B, buf)
        batch.build_pyramid([which])
        ref.upload(0, pr["I0"], pr["D0"], want[B - 1].reshape(H, W))
        ref.build_pyramid()
        assert np.array_equal(batch.download(B - 1, 1, which), ref.download(0, 1, which))
        # estimation straight after the upload call (the pyramid of the new frames is in place)
        batch.upload_async(which, 0, B, buf)
        P, px = batch.estimate(cam, cam, ident, ops.W_HUBER, 3)
        _lib.call("tdk_sync")
        P_again, px_again = batch.estimate(cam, cam, ident, ops.W_HUBER, 3)     # the frames are certainly there now
        assert np.array_equal(P, P_again) and px == px_again
        P1, px1 = ref.estimate(cam, cam, ident[:1], ops.W_HUBER, 3)            # (another block partition: last bits)
        assert np.max(np.abs(P[B - 1] - P1[0])) < 1e-12
    for x in (batch, ref, pin, pin8):
        x.close()


def test_session_outlives_its_ring(ops):
    """More committed steps than max_refframes.  Saturating ages (default): the session keeps going and
    equals the oracle chain with the same rule (ages clamped to the frames the ring still holds, the
    reference list cut to the ring).  Unbounded ages: the step that first needs a dropped frame fails
    with TDK_ERR_AGE_EXCEEDS_REFFRAMES and commits nothing."""
    from oracle import oracle as orc
    from tadataka_amd import _lib, synthetic
    H, W, n_frames, R = 48, 64, 7, 2
    cam, depth0, T_w, images = synthetic.make_track(H, W, n_frames)
    pargs = (0.5, 10.0, 0.01, 0.01, 0.004, 0.01)
    pg, po = ops.make_params(*pargs), orc.make_params(*pargs)
    rng = np.random.default_rng(2)
    depth = depth0 * rng.uniform(0.95, 1.05, (H, W))
    var = np.full((H, W), 0.05)
    age = np.zeros((H, W), dtype=np.uint64)
    for saturate in (True, False):
        sd = ops.SemiDenseSession(1, H, W, max_refframes=R)
        sd.set_age_policy(saturate)
        sd.set_params(pg, *SD_DEFAULTS)
        sd.push_frame(0, cam, images[0], T_w[0])
        sd.set_maps(0, depth, var, age)
        o_depth, o_var, o_age = depth.copy(), var.copy(), age.copy()
        frames = [(cam, images[0], T_w[0])]
        failed_at = None
        for k in range(1, n_frames):
            T10 = np.linalg.inv(T_w[k]) @ T_w[k - 1]
            sd.push_frame(0, cam, images[k])
            try:
                sd.step(T10[None], T_w[k][None], commit=True)
            except _lib.TdkError as e:
                assert e.status == _lib.TDK_ERR_AGE_EXCEEDS_REFFRAMES
                failed_at = k
                break
            n_ref = min(len(frames), R)
            a1 = np.minimum(orc.increment_age(o_age, cam, cam, T10, o_depth), np.uint64(n_ref))
            d1, v1 = orc.propagate(T10, cam, cam, o_depth, o_var, *SD_DEFAULTS)
            o_depth, o_var, o_flag = orc.update_depth((cam, images[k], T_w[k]), frames[-n_ref:], a1, d1, v1, po)
            o_age = a1
            frames.append((cam, images[k], T_w[k]))
            gd, gv, ga, gf = sd.get_maps(0, with_flag=True)
            assert np.array_equal(ga, o_age) and np.array_equal(gf, o_flag), k
            assert np.array_equal(gd, o_depth) and np.array_equal(gv, o_var), k
        if saturate:
            assert failed_at is None and int(o_age.max()) == R
        else:
            assert failed_at == R + 1          # ages reach R + 1 in step R + 1; the ring holds R references
            _, _, ga = sd.get_maps(0)
            assert int(ga.max()) == R          # nothing of the failed step was committed
        sd.close()


SD_DEFAULTS = (1.0, 10.0, 0.01)


@pytest.mark.parametrize("dz", [0.0, 1.1, 4.5])
def test_forward_warp_with_many_sources_per_target(ops, dz):
    """increment_age / propagate when the warp folds many sources onto one target (the camera backs
    off by dz: 1, ~2.4 and ~10 sources per target): slots (<= 4 sources), slots + overflow chain
    (more) -- bit-exact against the oracle's raster-order loop, host-pointer entries, device maps and
    the session alike."""
    from oracle import oracle as orc
    from tadataka_amd import synthetic
    H, W = 60, 81               # odd pixel count
    rng = np.random.default_rng(17)
    cam = synthetic.camera_for(W, H)
    depth0 = 2.0 + 0.2 * rng.uniform(-1, 1, (H, W))
    var0 = rng.uniform(0.01, 0.2, (H, W))
    age0 = rng.integers(0, 5, (H, W)).astype(np.uint64)
    T10 = np.eye(4)
    T10[:3, 3] = [0.01, -0.02, dz]
    a1 = ops.increment_age(age0, cam, cam, T10, depth0)
    d1, v1 = ops.propagate(T10, cam, cam, depth0, var0, *SD_DEFAULTS)
    oa = orc.increment_age(age0, cam, cam, T10, depth0)
    od, ov = orc.propagate(T10, cam, cam, depth0, var0, *SD_DEFAULTS)
    assert np.array_equal(a1, oa) and np.array_equal(d1, od) and np.array_equal(v1, ov)
    if dz > 4:
        assert int((oa > 0).sum()) < H * W // 6      # the frame really collapsed onto few targets
    m_a = ops.increment_age_maps(age0, cam, cam, T10, depth0)
    m_d, m_v = ops.propagate_maps(T10, cam, cam, depth0, var0, *SD_DEFAULTS)
    assert np.array_equal(m_a, oa) and np.array_equal(m_d, od) and np.array_equal(m_v, ov)
    sd = ops.SemiDenseSession(3, H, W, max_refframes=1)
    sd.set_age_policy(False)
    sd.set_params(ops.make_params(0.5, 10.0, 0.01, 0.01, 0.004, 0.01), *SD_DEFAULTS)
    img = rng.uniform(0, 1, (H, W))
    for t in range(3):
        sd.push_frame(t, cam, img, np.eye(4))
        sd.push_frame(t, cam, img, np.linalg.inv(T10))
        sd.set_maps(t, depth0, var0, age0)
    sd.propagate(np.tile(T10, (3, 1, 1)), commit=False)
    for t in range(3):
        gd, gv, ga = sd.get_results(t, with_flag=False)
        assert np.array_equal(ga, oa) and np.array_equal(gd, od) and np.array_equal(gv, ov)
    sd.close()


# ---------------------------------------------------------------------------
# Tukey: sampled brackets + one pass == radix select == the in-kernel exact fallback, bit for bit
# ---------------------------------------------------------------------------
_TUKEY_SCRIPT = """
import hashlib, os, sys
import numpy as np
from tadataka_amd import ops, synthetic
h = hashlib.sha256()
fall = 0
ident = np.concatenate([np.eye(3).ravel(), np.zeros(3)])
for (H, W, B, quant) in ((480, 640, 3, None), (53, 71, 2, None), (120, 160, 4, 64), (96, 128, 2, 4)):
    batch = ops.DvoBatch(B, H, W)
    batch.set_option("tukey", int(sys.argv[1]))
    for i in range(B):
        pr = synthetic.make_pair(H, W, seed=300 + i, rot_scale=0.01, trans_scale=0.03)
        I0, I1 = pr["I0"], pr["I1"]
        if quant:            # quantised images: thousands of exactly equal residuals (tie groups)
            I0, I1 = np.round(I0 * quant) / quant, np.round(I1 * quant) / quant
        batch.upload(i, I0, pr["D0"], I1)
    cam = synthetic.camera_for(W, H)
    P = np.tile(ident, (B, 1))
    P[:, 9:] = np.linspace(-0.01, 0.01, B)[:, None]
    ev = batch.evaluate(0, cam, cam, P, ops.W_TUKEY)
    for k in ("H", "b", "sum_sq", "n_update"):
        h.update(np.ascontiguousarray(ev[k]).tobytes())
    Pl, n = batch.estimate_level(0, cam, cam, np.tile(ident, (B, 1)), ops.W_TUKEY, 5)
    h.update(Pl.tobytes()); h.update(n.tobytes())
    fall += batch.tukey_fallbacks()
    batch.close()
print("RESULT", h.hexdigest(), fall)
"""


def test_tukey_brackets_radix_and_fallback_agree_bit_for_bit():
    """The three ways to the two medians of compute_weights_tukey give the same doubles: normal equations,
    poses and evaluation counts hash alike -- on smooth frames (brackets hold), on a level smaller than
    the sample (exhaustive), on quantised frames (tie groups of thousands: the brackets overflow and the
    exact path takes over).  The default path must take the exact fallback only for the tie-heavy cases."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for mode, value in (("", "0"), ("radix", "1"), ("fallback", "2")):      # tdk_dvo_set_option(TDK_DVO_OPT_TUKEY)
        env = dict(os.environ)
        out = subprocess.run([sys.executable, "-c", _TUKEY_SCRIPT, value], env=env, cwd=root, check=True,
                             capture_output=True, text=True, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][-1].split()
        res[mode] = (line[1], int(line[2]))
    assert res[""][0] == res["radix"][0] == res["fallback"][0], res
    assert res["radix"][1] == 0 and res["fallback"][1] > res[""][1], res
    print("fallbacks of the default path:", res[""][1])


# ---------------------------------------------------------------------------
# depth maps with missing readings
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("fill", ["zero", "nan"])
@pytest.mark.parametrize("name", ["None", "huber", "tukey", "student-t"])
def test_depth_holes_vs_reference_loop(ops, golden, fill, name):
    """9 % of the depth map is 0 (what a depth sensor reports where it has no reading: the reference keeps those
    pixels -- back-projected to the origin -- in its masks, with a Jacobian that scales with 1 / t_z) or NaN
    (every comparison masks them out; the anti-aliasing prefilter spreads them).  Pose after every level and
    evaluation counts as the reference's own loop, both pyramids."""
    import scenes
    g = golden("dvo_holes.npz")
    pair = scenes.holes_pair(0.0 if fill == "zero" else np.nan)
    for aa in (False, True):
        _check_pyramid_run(ops, g, f"{fill}_{'aa' if aa else 'bl'}_{name}", pair["I0"], pair["D0"], pair["I1"], pair["cam"],
                           name, None, 3, aa)
