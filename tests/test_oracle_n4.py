"""CPU tests of the oracle pieces added for SURVEY N4 / N1: regularize, fusion,
rgb2gray -- against the Rust #[test] inputs where they exist -- and of the
committed cfg3 fixture (the oracle must keep reproducing it)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))

from oracle import oracle as orc


def _inv(v):   # src/numeric.rs:3-5
    return 1.0 / (np.asarray(v, dtype=np.float64) + 2.220446049250313e-16)


def test_fusion_literal():
    """src/semi_dense/fusion.rs:50-89, the test's inputs and its assertions."""
    mu1 = np.array([[1.9, -2.2], [-3.8, 4.1], [-1.5, 4.5]])
    mu2 = np.array([[-4.1, -2.5], [1.2, 5.0], [6.4, 4.1]])
    var1 = np.array([[4.8, 2.2], [3.1, 6.8], [4.0, 2.1]])
    var2 = np.array([[4.2, 3.1], [0.01, 2.0], [6.0, 3.9]])
    mu0, var0 = orc.fusion_arrays(mu1, mu2, var1, var2)
    # fusion.rs:84-85 writes (v2 m1 + v1 m2) / (v1 + v2); the function computes
    # (m1 v2 + m2 v1) / (v1 + v2): products commute, same doubles
    assert np.array_equal(mu0, (var2 * mu1 + var1 * mu2) / (var1 + var2))
    assert np.array_equal(var0, (var1 * var2) / (var1 + var2))


def test_regularize_patch_literal_inputs():
    """Inputs of regularization.rs:72-115.  (That test's `expected` divides by a
    contribution count and gates on are_statically_same, which the function it
    tests never did -- the module is commented out of mod.rs:13 and the test is
    stale; the function body :5-27 is what is restated.)"""
    depth = np.array([[0., 3., 3.], [4., 1., 9.], [2., 8., 2.]])
    variance = np.array([[0., 3., 1.], [8., 2., 4.], [0., 1., 2.]])
    flag = np.array([[0, 1, 1], [1, 0, 1], [0, 1, 1]])
    idm, iv = _inv(depth), variance     # the test passes the variance map itself as the weights
    num = den = 0.0
    for y in range(3):
        for x in range(3):
            if flag[y, x] == 0:         # Flag::Success
                num = num + idm[y, x] * iv[y, x]
                den = den + iv[y, x]
    assert orc.regularize_patch(idm, iv, flag) == num / den
    assert orc.regularize_patch(idm, iv, np.ones((3, 3))) is None      # nothing contributes
    assert orc.regularize_patch(idm, np.zeros((3, 3)), flag) is None   # zero denominator


def test_regularize_literal_structure():
    """regularization.rs:117-160: regularized[0, 2] equals the patch result of the
    zero-padded neighbourhood; here with the weights `regularize` really uses
    (inverse variances, :39-40)."""
    depth = np.array([[1., 2., 4., 2.], [3., 4., 1., 9.], [1., 4., 8., 1.]])
    variance = np.array([[1., 4., 3., 5.], [3., 5., 2., 1.], [2., 4., 2., 2.]])
    flag = np.array([[1, 0, 1, 1], [1, 1, 0, 1], [0, 1, 0, 1]])
    out = orc.regularize(depth, variance, flag)
    id_patch = np.array([[0., 0., 0.], list(_inv([2., 4., 2.])), list(_inv([4., 1., 9.]))])
    iv_patch = np.array([[0., 0., 0.], list(_inv([4., 3., 5.])), list(_inv([5., 2., 1.]))])
    f_patch = np.array([[-9, -9, -9], [0, 1, 1], [1, 0, 1]])
    assert out[0, 2] == _inv(orc.regularize_patch(id_patch, iv_patch, f_patch))
    # a pixel whose whole neighbourhood failed keeps its depth
    none = np.ones_like(flag)
    assert np.array_equal(orc.regularize(depth, variance, none), depth)
    # uniform map, all Success: the weighted mean of equal values is that value
    d = np.full((5, 7), 2.0); v = np.full((5, 7), 0.5)
    assert np.allclose(orc.regularize(d, v, np.zeros((5, 7), dtype=np.int64)), 2.0, rtol=1e-14)


def test_rgb2gray_definition():
    rng = np.random.default_rng(0)
    rgb = rng.uniform(0, 1, (7, 9, 3))
    ref = rgb @ np.array([0.2125, 0.7154, 0.0721])
    assert np.max(np.abs(orc.rgb2gray(rgb) - ref)) < 3e-16
    rgba = np.concatenate([rgb, rng.uniform(0, 1, (7, 9, 1))], axis=2)
    assert np.array_equal(orc.rgb2gray(rgba), orc.rgb2gray(rgb))
    u8 = rng.integers(0, 256, (5, 4, 3)).astype(np.uint8)
    assert np.max(np.abs(orc.rgb2gray(u8) - (u8 * (1.0 / 255.0)) @ np.array([0.2125, 0.7154, 0.0721]))) < 3e-16
    assert orc.rgb2gray(np.ones((2, 2, 3)))[0, 0] == (0.2125 + 0.7154) + 0.0721


def test_cfg3_fixture_reproduced_by_oracle(golden):
    """BASELINE configs[2] at its stated size: the oracle still does the work the
    committed fixture records (flag histogram + digests of every output map)."""
    import generate_cfg3_fixture as gen
    fx = golden("semi_dense_cfg3.npz")
    out = gen.compute()
    assert np.array_equal(out["flag_histogram"], fx["flag_histogram"])
    assert int(out["flag_histogram"].sum()) == 480 * 640
    valid = 1.0 - out["flag_histogram"][9] / (480 * 640)
    assert 0.28 < valid < 0.32                                       # "~30 % valid pixels"
    for k in ("sha_age1", "sha_depth1", "sha_var1", "sha_depth", "sha_var", "sha_flag"):
        assert np.array_equal(out[k], fx[k]), k
    assert int(out["n_age1_nonzero"]) == int(fx["n_age1_nonzero"])
