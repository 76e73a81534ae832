"""Pins the CPU oracle (oracle/tdk_oracle.c + oracle/oracle.py) against the
fixtures captured from the reference itself (tests/golden/generate_golden.py).
CPU only."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from oracle import oracle as orc
from tadataka_amd import synthetic
from conftest import b6_err, h21_err, rel_err

WEIGHTS = [None, "huber", "student-t", "tukey", "map"]


def _weights(d, name):
    return d["weight_map"] if name == "map" else name


@pytest.fixture(scope="module")
def small(golden):
    return golden("dvo_small.npz")


def test_dvo_rows_match_reference(small):
    d = small
    cam = d["cam"]
    GX, GY = orc.image_gradient(d["I1"])
    for k in range(int(d["s_None_n_updates"])):
        T = d["s_None_err_T"][k]
        J, r, w = orc.dvo_rows(d["I0"], d["D0"], d["I1"], GX, GY, cam, cam,
                               T[:3, :3], T[:3, 3], None)
        assert J.shape[0] == int(d[f"s_None_u{k}_n_valid"])
        assert rel_err(J, d[f"s_None_u{k}_J"]) < 1e-10
        # the residual is the un-warped I0 - I1 (F3): bit-exact
        assert np.array_equal(r, d[f"s_None_u{k}_r"])


@pytest.mark.parametrize("wname", WEIGHTS)
def test_dvo_weights_and_normal_equations(small, wname):
    d = small
    cam = d["cam"]
    key = f"s_{wname}"
    GX, GY = orc.image_gradient(d["I1"])
    for k in range(int(d[f"{key}_n_updates"])):
        T = d[f"{key}_err_T"][k]
        weights = _weights(d, wname)
        J, r, w = orc.dvo_rows(d["I0"], d["D0"], d["I1"], GX, GY, cam, cam,
                               T[:3, :3], T[:3, 3], weights)
        assert J.shape[0] == int(d[f"{key}_u{k}_n_valid"])
        assert rel_err(w, d[f"{key}_u{k}_w"]) < 1e-10
        H, b, n = orc.dvo_normal_equations(d["I0"], d["D0"], d["I1"], GX, GY, cam, cam,
                                           T[:3, :3], T[:3, 3], weights)
        # solve_linear_equation applies sqrt(w) to rows => normal eqs carry w
        Href = d[f"{key}_u{k}_H"]
        iu = np.triu_indices(6)
        assert n == J.shape[0]
        assert h21_err(H, Href[iu]) < 1e-9
        assert b6_err(b, d[f"{key}_u{k}_b"], Href[iu]) < 1e-9
        # and the lstsq solution of the reference equals the normal-eq solve
        Hm = np.zeros((6, 6)); Hm[iu] = H; Hm = Hm + Hm.T - np.diag(np.diag(Hm))
        xi = np.linalg.solve(Hm, b)
        assert np.allclose(xi, d[f"{key}_u{k}_xi"], rtol=1e-7, atol=1e-10)


@pytest.mark.parametrize("wname", WEIGHTS)
def test_photometric_error_matches_reference(small, wname):
    d = small
    cam = d["cam"]
    Ts, vals = d[f"s_{wname}_err_T"], d[f"s_{wname}_err_val"]
    for T, v in zip(Ts, vals):
        e = orc.photometric_error(d["I0"], d["D0"], d["I1"], cam, cam, T)
        assert abs(e - v) <= 1e-10 * abs(v)


@pytest.mark.parametrize("wname", WEIGHTS)
def test_dvo_level_loop_matches_reference(small, wname):
    d = small
    cam = d["cam"]
    rot, t = orc.dvo_estimate_level(d["I0"], d["D0"], d["I1"], cam, cam,
                                    Rotation.from_rotvec(np.zeros(3)), np.zeros(3),
                                    _weights(d, wname), max_iter=20)
    assert np.allclose(rot.as_rotvec(), d[f"s_{wname}_final_rotvec"], atol=1e-9)
    assert np.allclose(t, d[f"s_{wname}_final_t"], atol=1e-9)


def test_dvo_vga_matches_reference(golden):
    v = golden("dvo_vga.npz")
    pair = synthetic.make_pair(480, 640, seed=0)
    assert np.allclose(pair["omega"], v["omega_true"]) and np.allclose(pair["t"], v["t_true"])
    cam = pair["cam"]
    GX, GY = orc.image_gradient(pair["I1"])
    iu = np.triu_indices(6)
    for wname in (None, "huber"):
        key = f"v_{wname}"
        for k in range(int(v[f"{key}_n_updates"])):
            T = v[f"{key}_err_T"][k]
            H, b, n = orc.dvo_normal_equations(pair["I0"], pair["D0"], pair["I1"], GX, GY,
                                               cam, cam, T[:3, :3], T[:3, 3], wname)
            assert n == int(v[f"{key}_u{k}_n_valid"])
            assert h21_err(H, v[f"{key}_u{k}_H"][iu]) < 1e-9
            assert b6_err(b, v[f"{key}_u{k}_b"], v[f"{key}_u{k}_H"][iu]) < 1e-9
        for T, val in zip(v[f"{key}_err_T"], v[f"{key}_err_val"]):
            e = orc.photometric_error(pair["I0"], pair["D0"], pair["I1"], cam, cam, T)
            assert abs(e - val) <= 1e-10 * abs(val)


def test_dvo_pyramid_matches_reference(golden):
    p = golden("dvo_pyramid.npz")
    pair = synthetic.make_pair(120, 160, seed=4)
    for wname in (None, "huber"):
        rot, t = orc.dvo_estimate(pair["I0"], pair["D0"], pair["I1"], pair["cam"], pair["cam"],
                                  wname, n_coarse_to_fine=3, max_iter=20)
        assert np.allclose(rot.as_rotvec(), p[f"pyr_{wname}_rotvec"], atol=1e-9)
        assert np.allclose(t, p[f"pyr_{wname}_t"], atol=1e-9)


def test_pure_python_reference_numerics(golden):
    g = golden("pyref.npz")
    # np.gradient (tadataka/vo/dvo/jacobian.py:27-29)
    gx, gy = orc.image_gradient(g["grad_img"])
    assert np.array_equal(gx, g["grad_gx"]) and np.array_equal(gy, g["grad_gy"])
    # is_in_image_range (tadataka/utils.py:35-54)
    assert np.array_equal(orc.is_in_image_range(g["rng_kp"], (9, 13)), g["rng_mask"])
    # exp_se3_t_ (tadataka/se3.py:15-29)
    for xi, t in zip(g["se3_xi"], g["se3_t"]):
        assert np.allclose(orc.exp_se3_t(xi), t, rtol=0, atol=1e-14)
    # lstsq (tadataka/math.py:32-45)
    assert np.allclose(orc.solve_lstsq(g["ls_A"], g["ls_b"]), g["ls_x"], atol=1e-12)
    assert np.allclose(orc.solve_lstsq(g["ls_A"], g["ls_b"], g["ls_w"]), g["ls_xw"], atol=1e-12)


def test_ba_matches_cython_reference(golden):
    g = golden("ba_vectors.npz")
    poses, points = g["poses"], g["points"]
    n = poses.shape[0]
    idx = np.arange(n, dtype=np.int64)
    x, A, B = orc.ba_projection(poses, points, idx, idx)
    scale = lambda ref: np.maximum(np.abs(ref).max(axis=tuple(range(1, ref.ndim)), keepdims=True), 1.0)
    assert np.max(np.abs(x - g["x"]) / scale(g["x"])) < 1e-12
    assert np.max(np.abs(B - g["B"]) / scale(g["B"])) < 1e-11
    # the symbolic derivative and the analytic chain rule agree to rounding
    # everywhere, including |omega| -> 0 and |omega| = pi
    assert np.max(np.abs(A - g["A"]) / scale(g["A"])) < 1e-8
    R = np.array([orc.exp_so3(p[:3]) for p in poses])
    assert np.max(np.abs(R - g["R"])) < 1e-12


@pytest.mark.parametrize("shape, sigma", [((37, 53), (0.25, 0.2494)), ((48, 64), (0.625, 0.625)),
                                          ((9, 7), (2.03, 1.3)), ((2, 2), (0.625, 0.625)), ((5, 1), (0.625, 0.0))])
def test_gaussian_prefilter_is_scipy_ndimage(shape, sigma):
    """The anti-aliasing prefilter of skimage.transform.rescale is
    scipy.ndimage.gaussian_filter(image, sigma, mode='mirror'); scipy is importable,
    so the oracle's restatement is pinned against the real thing, bit for bit
    (same kernel weights in), down to frames smaller than the kernel."""
    from scipy import ndimage as ndi
    from scipy.ndimage._filters import _gaussian_kernel1d
    rng = np.random.default_rng(shape[0])
    img = rng.uniform(0, 1, shape)
    ref = ndi.gaussian_filter(img, sigma, mode="mirror")
    weights = [None if s <= 1e-15 else _gaussian_kernel1d(s, 0, int(4 * s + 0.5)) for s in sigma]
    assert np.array_equal(orc.gaussian_filter_mirror(img, weights[0], weights[1]), ref)
    for s in sigma:
        if s > 1e-15:                    # the C kernel weights: exp / sum as numpy computes them, to an ulp
            assert np.allclose(orc.gaussian_weights(s), _gaussian_kernel1d(s, 0, int(4 * s + 0.5)), rtol=0, atol=2e-16)


def test_anti_aliased_rescale_is_prefilter_plus_bilinear():
    """skimage/transform/_warps.py resize(): sigma = max(0, (factor - 1) / 2) per
    axis, ndimage 'mirror' Gaussian, then the order-1 warp."""
    from scipy import ndimage as ndi
    rng = np.random.default_rng(3)
    img = rng.uniform(0, 1, (61, 83))
    for level in (1, 2, 3):
        scale = 1 / 1.5 ** level
        Ho, Wo = orc.rescale_shape(img.shape, scale)
        sigma = (max(0.0, (61 / Ho - 1) / 2), max(0.0, (83 / Wo - 1) / 2))
        expected = orc.rescale(ndi.gaussian_filter(img, sigma, mode="mirror"), scale)
        got = orc.rescale(img, scale, anti_aliasing=True)
        assert got.shape == (Ho, Wo)
        assert np.allclose(got, expected, rtol=0, atol=4e-16)     # kernel weights differ by <= 1 ulp
    assert np.array_equal(orc.rescale(img, 1.0, anti_aliasing=True), img)


# ---------------------------------------------------------------------------
# round 3: the oracle's loop against the reference's own loop at the examples' settings,
# on real image statistics and on ill-conditioned scenes (generate_golden_r3.py)
# ---------------------------------------------------------------------------
def _assert_pose(rot, t, g, tag, atol=1e-8):
    assert np.allclose(rot.as_rotvec(), g[f"{tag}_rotvec"], atol=atol), tag
    assert np.allclose(t, g[f"{tag}_t"], atol=atol), tag


@pytest.mark.parametrize("aa", [False, True])
def test_example_5_levels_matches_reference(golden, aa):
    g = golden("dvo_examples.npz")
    pair = synthetic.make_pair(240, 320, seed=5)
    rot, t = orc.dvo_estimate(pair["I0"], pair["D0"], pair["I1"], pair["cam"], pair["cam"], "huber",
                              n_coarse_to_fine=5, anti_aliasing=aa)
    _assert_pose(rot, t, g, f"ex5_{'aa' if aa else 'bl'}_huber")


@pytest.mark.parametrize("name", ["None", "tukey"])
def test_new_tsukuba_half_resolution_matches_reference(golden, name):
    """Real frames; `name` tukey is only in the full-resolution fixture, so that one runs the
    5-level loop at 480x640 once (a few seconds)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import scenes
    g = golden("dvo_real.npz")
    I0 = scenes.gray_from_rgb_u8(g["rgb0"]); I1 = scenes.gray_from_rgb_u8(g["rgb1"])
    D0 = scenes.tsukuba_depth(*I0.shape)
    cam = scenes.TSUKUBA_CAM
    if name == "None":
        I0, D0, I1 = (orc.rescale(a, 0.5, anti_aliasing=True) for a in (I0, D0, I1))
        rot, t = orc.dvo_estimate(I0, D0, I1, cam * 0.5, cam * 0.5, None, n_coarse_to_fine=5, anti_aliasing=True)
        _assert_pose(rot, t, g, "half_aa_None", atol=1e-7)
    else:
        rot, t = orc.dvo_estimate(I0, D0, I1, cam, cam, name, n_coarse_to_fine=5, anti_aliasing=True)
        _assert_pose(rot, t, g, f"full_aa_{name}", atol=1e-7)


@pytest.mark.parametrize("scene", ["plane1d", "halfflat", "weaky", "weaky2", "diag2"])
def test_ill_conditioned_scenes_match_reference(golden, scene):
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import scenes
    from scipy.spatial.transform import Rotation
    g = golden("dvo_ill.npz")
    pair = scenes.ill_pair(scene)
    trace = []
    rot, t = orc.dvo_estimate_level(pair["I0"], pair["D0"], pair["I1"], pair["cam"], pair["cam"],
                                    Rotation.from_rotvec(np.zeros(3)), np.zeros(3), "huber", 20, trace)
    xis = np.array([tr["xi"] for tr in trace[1:]])
    ref = g[f"{scene}_huber_xis"]
    assert xis.shape == ref.shape
    assert np.max(np.abs(xis - ref)) < 1e-7 * max(1.0, np.max(np.abs(ref)))
    _assert_pose(rot, t, g, f"{scene}_huber", atol=1e-7)
    rot, t = orc.dvo_estimate(pair["I0"], pair["D0"], pair["I1"], pair["cam"], pair["cam"], None,
                              n_coarse_to_fine=3, anti_aliasing=True)
    _assert_pose(rot, t, g, f"{scene}_aa_None_pyr", atol=1e-7)
