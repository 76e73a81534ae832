"""GPU parity tests added in round 2: every BASELINE config at its stated size,
the device-resident semi-dense session, the N4 post-steps and rgb2gray.

Bars: bit-exact for integer / index / flag outputs and for everything compiled
with -ffp-contract=off (semi-dense, N4, rgb2gray); 1e-6 on recovered poses."""
import hashlib
import os
import sys

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from conftest import b6_err, h21_err

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))

pytestmark = pytest.mark.gpu

POSE_ATOL = 1e-6
SD_PARAMS = (0.5, 10.0, 0.01, 0.01, 0.002, 0.02)
SD_DEFAULTS = (1.0, 10.0, 0.01)


@pytest.fixture(scope="module")
def ops():
    from tadataka_amd import _lib, ops as o
    _lib.require_gpu()
    return o


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def _pose12(T):
    return np.concatenate([T[:3, :3].ravel(), T[:3, 3]])


def _sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


# ---------------------------------------------------------------------------
# cfg2: 640x480, 3-level pyramid ratio 1.5, Huber, max_iter 20 -- against the
# reference's own PoseChangeEstimator run (tests/golden/dvo_vga_pyramid.npz)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("aa", [False, True])
@pytest.mark.parametrize("wname", [None, "huber"])
def test_cfg2_vga_3level_vs_reference_loop(ops, golden, aa, wname):
    from tadataka_amd import synthetic
    g = golden("dvo_vga_pyramid.npz")
    tag = f"pyr_aa_{wname}" if aa else f"pyr_{wname}"
    pair = synthetic.make_pair(480, 640, seed=0)
    cam = pair["cam"]
    mode = ops.WEIGHT_MODES[wname]
    batch = ops.DvoBatch(1, 480, 640, n_levels=3, ratio=1.5)
    batch.set_anti_aliasing(aa)
    batch.upload(0, pair["I0"], pair["D0"], pair["I1"])
    batch.build_pyramid()
    ident = _pose12(np.eye(4))[None]
    # (1) the fused coarse-to-fine call
    P, px = batch.estimate(cam, cam, ident, mode, 20)
    R = Rotation.from_rotvec(g[f"{tag}_rotvec"]).as_matrix()
    assert np.max(np.abs(P[0, :9].reshape(3, 3) - R)) < POSE_ATOL
    assert np.max(np.abs(P[0, 9:] - g[f"{tag}_t"])) < POSE_ATOL
    # every source pixel of every evaluation is counted: sum_l N_l * evals_l of the reference
    shapes = [batch.level_shape(l) for l in (2, 1, 0)]
    assert px == sum(int(e) * h * w for e, (h, w) in zip(g[f"{tag}_evals"], shapes))
    # (2) level by level: evaluation counts and the pose after each level
    Pl = ident
    for k, level in enumerate((2, 1, 0)):
        Pl, n_evals = batch.estimate_level(level, cam, cam, Pl, mode, 20)
        assert n_evals[0] == int(g[f"{tag}_evals"][k])
        lp = g[f"{tag}_level_poses"][k]
        assert np.max(np.abs(Pl[0, :9].reshape(3, 3) - Rotation.from_rotvec(lp[:3]).as_matrix())) < POSE_ATOL
        assert np.max(np.abs(Pl[0, 9:] - lp[3:])) < POSE_ATOL
    assert np.array_equal(Pl, P)          # fused == chained, bit for bit
    batch.close()


# ---------------------------------------------------------------------------
# cfg4, one GPU's shard: 64 pairs of 1280x720, 1 level, 2 iterations
# ---------------------------------------------------------------------------
def test_cfg4_shard_64x720p(ops, orc):
    import bench
    from tadataka_amd import synthetic
    B, H, W = 64, 720, 1280
    cam = synthetic.camera_for(W, H)
    truth = bench.true_poses(B, 0)
    batch = ops.DvoBatch(B, H, W, n_levels=1)
    batch.fill_synthetic(cam, truth, seed0=0, noise=0.02)
    ident = np.tile(_pose12(np.eye(4)), (B, 1))
    P, n_evals = batch.estimate_level(0, cam, cam, ident, ops.W_HUBER, max_iter=2)
    # properties at the full batch: the pairs moved towards their truth (the reference does not
    # re-warp the residual, F3, so single pairs may stall), used at most max_iter + 1
    # evaluations, and the whole thing is bit-reproducible
    assert np.all(n_evals >= 2) and np.all(n_evals <= 3)
    err0 = np.linalg.norm(truth[:, 9:], axis=1)
    err1 = np.linalg.norm(P[:, 9:] - truth[:, 9:], axis=1)
    assert np.mean(err1 < err0) > 0.9 and np.median(err1 / err0) < 0.7
    P2, n2 = batch.estimate_level(0, cam, cam, ident, ops.W_HUBER, max_iter=2)
    assert np.array_equal(P, P2) and np.array_equal(n_evals, n2)
    # evaluation sums of all pairs at the identity: mask sizes are exact integers <= N
    ev = batch.evaluate(0, cam, cam, ident, ops.W_HUBER)
    assert np.all(ev["n_error"] <= H * W) and np.all(ev["n_update"] == ev["n_error"])
    assert np.all(ev["n_error"] > 0.97 * H * W)
    # oracle parity on two sampled pairs (first / last block of the XCD-major grid)
    for i in (5, 63):
        I0 = batch.download(i, 0, "I0"); D0 = batch.download(i, 0, "D0"); I1 = batch.download(i, 0, "I1")
        GX, GY = orc.image_gradient(I1)
        Hm, b, n = orc.dvo_normal_equations(I0, D0, I1, GX, GY, cam, cam, np.eye(3), np.zeros(3), "huber")
        ss, ne = orc.photometric_error_sums(I0, D0, I1, cam, cam, np.eye(4))
        assert ev["n_update"][i] == n and ev["n_error"][i] == ne
        assert h21_err(ev["H"][i], Hm) < 1e-9 and b6_err(ev["b"][i], b, Hm) < 1e-9
        assert abs(ev["sum_sq"][i] - ss) <= 1e-9 * ss
        rot, t = orc.dvo_estimate_level(I0, D0, I1, cam, cam, Rotation.from_rotvec(np.zeros(3)), np.zeros(3),
                                        "huber", 2)
        assert np.max(np.abs(P[i, :9].reshape(3, 3) - rot.as_matrix())) < POSE_ATOL
        assert np.max(np.abs(P[i, 9:] - t)) < POSE_ATOL
    # a pair gives the same pose alone as inside the batch of 64 (position independence)
    one = ops.DvoBatch(1, H, W, n_levels=1)
    one.fill_synthetic(cam, truth[63:64], seed0=63, noise=0.02)
    P1, _ = one.estimate_level(0, cam, cam, ident[:1], ops.W_HUBER, max_iter=2)
    assert np.max(np.abs(P1[0] - P[63])) < 1e-12
    one.close()
    batch.close()


# ---------------------------------------------------------------------------
# cfg3 at its stated size, and the fixture that freezes its work
# ---------------------------------------------------------------------------
def test_cfg3_vga_bit_exact_and_fixture(ops, orc, golden):
    import generate_cfg3_fixture as gen
    from tadataka_amd import synthetic
    fx = golden("semi_dense_cfg3.npz")
    c = synthetic.make_semi_dense_case(480, 640, seed=1)
    key = (c["cam"], c["key_image"], c["T_wk"]); ref = (c["cam"], c["ref_image"], c["T_wr"])
    T10 = np.linalg.inv(c["T_wk"]) @ c["T_wr"]
    pg, po = ops.make_params(*SD_PARAMS), orc.make_params(*SD_PARAMS)
    # host-pointer entries (rust_bindings.semi_dense signatures)
    age1 = ops.increment_age(c["age"], c["cam"], c["cam"], T10, c["prior_depth"])
    d1, v1 = ops.propagate(T10, c["cam"], c["cam"], c["prior_depth"], c["prior_variance"], *SD_DEFAULTS)
    d, v, f = ops.update_depth(key, [ref], c["age"], c["prior_depth"], c["prior_variance"], pg)
    assert np.array_equal(age1, orc.increment_age(c["age"], c["cam"], c["cam"], T10, c["prior_depth"]))
    od1, ov1 = orc.propagate(T10, c["cam"], c["cam"], c["prior_depth"], c["prior_variance"], *SD_DEFAULTS)
    assert np.array_equal(d1, od1) and np.array_equal(v1, ov1)
    od, ov, of = orc.update_depth(key, [ref], c["age"], c["prior_depth"], c["prior_variance"], po)
    assert np.array_equal(f, of) and np.array_equal(d, od) and np.array_equal(v, ov)
    hist = np.array([(f == -b).sum() for b in range(10)])
    assert np.array_equal(hist, fx["flag_histogram"])
    for name, arr in (("sha_age1", age1), ("sha_depth1", d1), ("sha_var1", v1), ("sha_depth", d),
                      ("sha_var", v), ("sha_flag", f)):
        assert np.array_equal(_sha(arr), fx[name]), name
    # the same three operators through the device-resident session
    sd = ops.SemiDenseSession(2, 480, 640, max_refframes=2)
    sd.set_age_policy(False)      # the operators on the fixture's maps as they are: ages are not limited
    sd.set_params(pg, *SD_DEFAULTS)
    for t in range(2):
        sd.push_frame(t, c["cam"], c["ref_image"], c["T_wr"])
        sd.push_frame(t, c["cam"], c["key_image"], c["T_wk"])
        sd.set_maps(t, c["prior_depth"], c["prior_variance"], c["age"])
    sd.propagate(np.array([T10, T10]), commit=False)
    for t in range(2):
        sd1, sv1, sa1 = sd.get_results(t, with_flag=False)
        assert np.array_equal(sa1, age1) and np.array_equal(sd1, d1) and np.array_equal(sv1, v1)
    h = sd.update_depth(commit=False, histogram=True)
    assert np.array_equal(h[0], fx["flag_histogram"]) and np.array_equal(h[1], fx["flag_histogram"])
    for t in range(2):
        rd, rv, ra, rf = sd.get_results(t)
        assert np.array_equal(rf, f) and np.array_equal(rd, d) and np.array_equal(rv, v)
        assert np.array_equal(ra, c["age"])
    # nothing was committed
    cd, cv, ca = sd.get_maps(0)
    assert np.array_equal(cd, c["prior_depth"]) and np.array_equal(ca, c["age"])
    sd.close()


# ---------------------------------------------------------------------------
# one engineered pixel per Flag (structure of the reference's
# tests/vo/semi_dense/test_semi_dense.py:100-149 on a synthetic scene)
# ---------------------------------------------------------------------------
def test_one_engineered_pixel_per_flag(ops, orc):
    from tadataka_amd import synthetic
    H, W = 240, 320
    c = synthetic.make_semi_dense_case(H, W, seed=1)
    key_img = c["key_image"].copy()
    key_img[100:140, 200:260] = 0.5                      # a texture-less patch
    key = (c["cam"], key_img, c["T_wk"]); ref = (c["cam"], c["ref_image"], c["T_wr"])
    ahead = np.eye(4); ahead[2, 3] = 5.0                 # a reference camera beyond the surface
    ref_ahead = (c["cam"], c["ref_image"], ahead)
    pg, po = ops.make_params(*SD_PARAMS), orc.make_params(*SD_PARAMS)
    gt, prior = c["depth_gt"], c["prior_depth"]
    cases = [  # (u_key, prior depth, prior variance, ref frame, expected flag)
        ([110, 100], -10.0, 10.0, ref, -7),              # NEGATIVE_PRIOR_DEPTH
        ([110, 100], 0.05, 0.2, ref, -1),                # HYPOTHESIS_OUT_OF_SERCH_RANGE
        ([230, 120], gt[120, 230], 0.05, ref, -6),       # INSUFFICIENT_GRADIENT
        ([0, 100], gt[100, 0], 0.05, ref, -2),           # KEY_OUT_OF_RANGE: on the image edge
        ([113, 2], prior[2, 113], 1e-4, ref, -5),        # REF_EPIPOLAR_TOO_SHORT: very short search range
        ([2, 0], prior[0, 2], 0.05, ref, -3),            # REF_CLOSE_OUT_OF_RANGE
        ([44, 0], prior[0, 44], 0.05, ref, -4),          # REF_FAR_OUT_OF_RANGE
        ([150, 120], gt[120, 150], 0.05, ref_ahead, -8), # NEGATIVE_REF_DEPTH
        ([113, 2], prior[2, 113], 0.05, ref, 0),         # SUCCESS
    ]
    for u, pd_, pv, rf, expected in cases:
        got = ops.estimate_one(u, float(pd_), pv, key, rf, pg)
        exp = orc.estimate_debug(u, float(pd_), pv, key, rf, po)
        assert exp[2] == expected, (u, exp)
        assert got == exp, (u, got, exp)
        if expected != 0:
            assert got[0] == float(pd_) and got[1] == pv           # Err(flag) => the prior comes back
    depth, variance, flag = ops.estimate_one([113, 2], float(prior[2, 113]), 0.05, key, ref, pg)
    assert flag == 0 and depth > 0.0 and variance > 0.0
    assert abs(depth - gt[2, 113]) < 0.05 * gt[2, 113]            # SUCCESS lands near the ground truth
    # NOT_PROCESSED only exists at map level (age == 0, semi_dense.rs:196-200); the same pixels
    # through update_depth give the same flags
    age = np.zeros((H, W), dtype=np.uint64)
    pd_map, pv_map = prior.copy(), c["prior_variance"].copy()
    for u, pd_, pv, rf, expected in cases[:7]:
        if u == [113, 2]:
            continue
        age[u[1], u[0]] = 1; pd_map[u[1], u[0]] = pd_; pv_map[u[1], u[0]] = pv
    age[2, 113] = 1
    d, v, f = ops.update_depth(key, [ref], age, pd_map, pv_map, pg)
    od, ov, of = orc.update_depth(key, [ref], age, pd_map, pv_map, po)
    assert np.array_equal(f, of) and np.array_equal(d, od) and np.array_equal(v, ov)
    assert f[2, 113] == 0 and f[100, 0] == -2 and f[120, 230] == -6 and f[0, 2] == -3 and f[0, 44] == -4
    assert f[100, 110] == -1 and f[50, 50] == -9 and int((f != -9).sum()) == int(age.sum())


# ---------------------------------------------------------------------------
# semi-dense session: chained steps, ring of reference frames, errors
# ---------------------------------------------------------------------------
def test_sd_session_chain_matches_oracle(ops, orc):
    from tadataka_amd import synthetic
    H, W, n = 96, 128, 3
    pg, po = ops.make_params(0.5, 10.0, 0.01, 0.01, 0.004, 0.01), orc.make_params(0.5, 10.0, 0.01, 0.01, 0.004, 0.01)
    sd = ops.SemiDenseSession(n, H, W, max_refframes=2)
    sd.set_age_policy(False)      # the reference's rule: ages are not limited, a frame that is gone is an error
    sd.set_params(pg, *SD_DEFAULTS)
    state = []
    for t in range(n):
        c = synthetic.make_semi_dense_case(H, W, seed=10 + t, valid_fraction=0.5)
        frames = [(c["cam"], c["ref_image"], c["T_wr"]), (c["cam"], c["key_image"], c["T_wk"])]
        T3 = np.eye(4); T3[0, 3] = -0.05; T3[1, 3] = 0.02
        frames.append((c["cam"], 0.5 * (c["key_image"] + c["ref_image"]), T3))
        sd.push_frame(t, *frames[0])
        # a fresh map: age 0 everywhere, as init_age does (examples/semi_dense_vo.py:203-204)
        age0 = np.zeros((H, W), dtype=np.uint64)
        sd.set_maps(t, c["prior_depth"], c["prior_variance"], age0)
        state.append(dict(frames=frames, depth=c["prior_depth"], var=c["prior_variance"], age=age0))
    for step in (1, 2):
        T10s, Twfs = [], []
        for t in range(n):
            fr = state[t]["frames"]
            sd.push_frame(t, fr[step][0], fr[step][1])                   # pose supplied by the step
            T10s.append(np.linalg.inv(fr[step][2]) @ fr[step - 1][2])
            Twfs.append(fr[step][2])
        if step == 1:      # a dry run first: results readable, state untouched
            sd.step(np.array(T10s), np.array(Twfs), commit=False)
            d0, v0, a0 = sd.get_maps(0)
            assert np.array_equal(d0, state[0]["depth"]) and np.array_equal(a0, state[0]["age"])
        hist = sd.step(np.array(T10s), np.array(Twfs), commit=True, histogram=True)
        for t in range(n):
            st = state[t]
            fr = st["frames"]
            d, v, a, f = orc.semi_dense_step(fr[step], fr[step - 1][0], fr[:step], T10s[t], st["age"],
                                             st["depth"], st["var"], po, *SD_DEFAULTS)
            gd, gv, ga, gf = sd.get_maps(t, with_flag=True)
            assert np.array_equal(ga, a) and np.array_equal(gf, f)
            assert np.array_equal(gd, d) and np.array_equal(gv, v)
            assert np.array_equal(hist[t], [(f == -b).sum() for b in range(10)])
            st.update(depth=d, var=v, age=a)
        assert int(state[0]["age"].max()) == step
    tm = sd.timing()
    assert tm["step_ms"] > 0 and tm["warp_ms"] > 0 and tm["update_depth_ms"] > 0
    # a third step: ages reach 3 with only 2 reference frames in the ring -> the reference exits, we raise
    from tadataka_amd._lib import TdkError
    for t in range(n):
        fr = state[t]["frames"]
        sd.push_frame(t, fr[1][0], fr[1][1], fr[1][2])
    before = sd.get_maps(1)
    with pytest.raises(TdkError):
        sd.step(np.tile(np.eye(4), (n, 1, 1)))
    after = sd.get_maps(1)
    assert all(np.array_equal(x, y) for x, y in zip(before, after))      # nothing was committed
    sd.close()


def test_sd_export_feeds_dvo_batch(ops):
    from tadataka_amd import synthetic
    H, W, n = 60, 80, 2
    sd = ops.SemiDenseSession(n, H, W, max_refframes=1)
    batch = ops.DvoBatch(n, H, W, n_levels=1, with_weight_map=True)
    cases = [synthetic.make_semi_dense_case(H, W, seed=20 + t) for t in range(n)]
    for t, c in enumerate(cases):
        sd.push_frame(t, c["cam"], c["ref_image"], c["T_wr"])
        sd.push_frame(t, c["cam"], c["key_image"], c["T_wk"])
        sd.set_maps(t, c["prior_depth"], c["prior_variance"] * (1 + t), c["age"])
    sd.export_dvo(batch)
    for t, c in enumerate(cases):
        assert np.array_equal(batch.download(t, 0, "I0"), c["ref_image"])
        assert np.array_equal(batch.download(t, 0, "I1"), c["key_image"])
        assert np.array_equal(batch.download(t, 0, "D0"), c["prior_depth"])
        assert np.array_equal(batch.download(t, 0, "W0"), 1.0 / (c["prior_variance"] * (1 + t) + 1e-16))   # tadataka.numeric.safe_invert
    # and the batch is usable: the weight-map DVO runs on what the session exported
    cam = cases[0]["cam"]
    P, n_evals = batch.estimate_level(0, cam, cam, np.tile(_pose12(np.eye(4)), (n, 1)), ops.W_MAP, 5)
    assert np.all(np.isfinite(P)) and np.all(n_evals >= 1)
    batch.close(); sd.close()


# ---------------------------------------------------------------------------
# N4 post-steps and rgb2gray
# ---------------------------------------------------------------------------
def test_regularize_and_fusion_bit_exact(ops, orc):
    rng = np.random.default_rng(5)
    H, W = 61, 83
    depth = rng.uniform(0.5, 8.0, (H, W)); var = rng.uniform(1e-3, 2.0, (H, W))
    flag = rng.choice([0, 0, 0, -6, -9, -2], (H, W)).astype(np.int64)
    flag[10:20, 10:20] = -9                       # a region with no Success pixel keeps its depth
    out = ops.regularize(depth, var, flag)
    assert np.array_equal(out, orc.regularize(depth, var, flag))
    assert np.array_equal(out[11:19, 11:19], depth[11:19, 11:19])
    d = np.array([[1., 2., 4., 2.], [3., 4., 1., 9.], [1., 4., 8., 1.]])        # regularization.rs:119-136
    v = np.array([[1., 4., 3., 5.], [3., 5., 2., 1.], [2., 4., 2., 2.]])
    f = np.array([[1, 0, 1, 1], [1, 1, 0, 1], [0, 1, 0, 1]])
    assert np.array_equal(ops.regularize(d, v, f), orc.regularize(d, v, f))
    assert np.array_equal(ops.regularize(d[:1, :1], v[:1, :1], np.zeros((1, 1))), orc.regularize(d[:1, :1], v[:1, :1], np.zeros((1, 1))))
    mu1 = np.array([[1.9, -2.2], [-3.8, 4.1], [-1.5, 4.5]]); mu2 = np.array([[-4.1, -2.5], [1.2, 5.0], [6.4, 4.1]])
    v1 = np.array([[4.8, 2.2], [3.1, 6.8], [4.0, 2.1]]); v2 = np.array([[4.2, 3.1], [0.01, 2.0], [6.0, 3.9]])
    mu, vv = ops.fusion_arrays(mu1, mu2, v1, v2)                                  # fusion.rs:50-89
    assert np.array_equal(mu, (v2 * mu1 + v1 * mu2) / (v1 + v2)) and np.array_equal(vv, (v1 * v2) / (v1 + v2))
    a, b = rng.normal(size=(2, 5000)); c, e = rng.uniform(0.01, 3, (2, 5000))
    gm, gv = ops.fusion_arrays(a, b, c, e)
    om, ov = orc.fusion_arrays(a, b, c, e)
    assert np.array_equal(gm, om) and np.array_equal(gv, ov)
    assert ops.fusion_arrays(np.zeros((0, 3)), np.zeros((0, 3)), np.zeros((0, 3)), np.zeros((0, 3)))[0].shape == (0, 3)


def test_rgb2gray_bit_exact(ops, orc):
    rng = np.random.default_rng(6)
    rgb = rng.uniform(0, 1, (37, 45, 3))
    assert np.array_equal(ops.rgb2gray(rgb), orc.rgb2gray(rgb))
    rgba = np.concatenate([rgb, rng.uniform(0, 1, (37, 45, 1))], axis=2)
    assert np.array_equal(ops.rgb2gray(rgba), orc.rgb2gray(rgb))
    u8 = rng.integers(0, 256, (48, 64, 3)).astype(np.uint8)
    assert np.array_equal(ops.rgb2gray(u8), orc.rgb2gray(u8))
    assert np.max(np.abs(ops.rgb2gray(rgb) - rgb @ np.array([0.2125, 0.7154, 0.0721]))) < 3e-16
    gray = rng.uniform(0, 1, (5, 6))
    assert np.array_equal(ops.rgb2gray(gray), gray)
    with pytest.raises(ValueError):
        ops.rgb2gray(np.zeros((4, 4, 2)))


# ---------------------------------------------------------------------------
# bundle adjustment: per-pose segment reduce, atomics-free block sums
# ---------------------------------------------------------------------------
def test_ba_block_sums_handle_vs_oracle_and_reproducible(ops, orc):
    from tadataka_amd import synthetic
    c = synthetic.make_ba_case(n_poses=6, n_points=3000, seed=4)
    x_true = orc.ba_projection(c["poses"], c["points"], c["vp_idx"], c["pt_idx"], jacobians=False)
    rng = np.random.default_rng(2)
    keep = rng.uniform(size=len(c["vp_idx"])) < 0.7            # ragged visibility
    keep[c["vp_idx"] == 3] = False                              # one pose sees nothing
    for order in ("viewpoint-major", "shuffled"):
        vp, pt, xt = c["vp_idx"][keep], c["pt_idx"][keep], x_true[keep]
        if order == "shuffled":
            perm = rng.permutation(len(vp))
            vp, pt, xt = vp[perm], pt[perm], xt[perm]
        ba = ops.BundleAdjustment(6, 3000, vp, pt, xt)
        U, ea, V, eb, err = ba.block_sums(c["poses_noisy"], c["points_noisy"])
        oU, oea, oV, oeb, oerr = orc.ba_block_reduce(c["poses_noisy"], c["points_noisy"], xt, vp, pt)
        scale = lambda a: np.maximum(np.abs(a).max(axis=1, keepdims=True), 1e-300)
        assert np.max(np.abs(U - oU) / np.maximum(scale(oU), 1e-30)) < 1e-9
        assert np.max(np.abs(V - oV) / scale(oV + 1e-300)) < 1e-9
        assert np.max(np.abs(ea - oea)) <= 1e-8 * np.max(np.abs(oea))
        assert np.max(np.abs(eb - oeb)) <= 1e-8 * np.max(np.abs(oeb))
        assert abs(err - oerr) <= 1e-9 * oerr
        assert np.all(U[3] == 0) and np.all(ea[3] == 0)
        again = ba.block_sums(c["poses_noisy"], c["points_noisy"])           # no atomics: bit-reproducible
        assert all(np.array_equal(a, b) for a, b in zip((U, ea, V, eb), again[:4])) and again[4] == err
        assert ba.sum_squared_error(c["poses_noisy"], c["points_noisy"]) == pytest.approx(oerr, rel=1e-9)
        # the stateless entry (atomics for the per-point sums) agrees
        sU, sea, sV, seb, serr = ops.ba_block_reduce(c["poses_noisy"], c["points_noisy"], xt, vp, pt)
        assert np.array_equal(sU, U) and np.array_equal(sea, ea) and serr == err
        assert np.max(np.abs(sV - V) / scale(oV + 1e-300)) < 1e-12
        ba.set_profiling(True)
        ba.step(c["poses_noisy"], c["points_noisy"], 1e-3)
        prof = ba.get_profile()
        assert prof["block_reduce"][0] == 1 and prof["schur"][0] == 1 and prof["block_reduce"][1] > 0
        ba.close()


def test_ba_reduced_camera_system_on_device(ops, orc):
    """The 6P x 6P system is solved by one workgroup on the device.  A pose that sees
    nothing has U_j = 0: with damping its update is exactly zero; without damping
    the system is singular -- the elimination without pivoting gives up at the zero
    pivot, the pivoted one finds no pivot either, and the status comes back as
    TDK_ERR_SINGULAR.  Sizes around the 16-wide blocks of the kernel (6 P = 12 ... 120)
    against a dense solve of the damped normal equations."""
    from tadataka_amd import _lib, synthetic
    c = synthetic.make_ba_case(n_poses=4, n_points=400, seed=12)
    x_true = orc.ba_projection(c["poses"], c["points"], c["vp_idx"], c["pt_idx"], jacobians=False)
    keep = c["vp_idx"] != 2
    vp, pt, xt = c["vp_idx"][keep], c["pt_idx"][keep], x_true[keep]
    ba = ops.BundleAdjustment(4, 400, vp, pt, xt)
    dposes, dpoints, _ = ba.step(c["poses_noisy"], c["points_noisy"], 1e-2)
    assert np.all(dposes[2] == 0) and np.all(np.isfinite(dposes)) and np.any(dposes[0] != 0)
    with pytest.raises(_lib.TdkError) as e:
        ba.step(c["poses_noisy"], c["points_noisy"], 0.0)
    assert e.value.status == _lib.TDK_ERR_SINGULAR
    dposes2, _, _ = ba.step(c["poses_noisy"], c["points_noisy"], 1e-2)     # the handle is usable afterwards
    assert np.array_equal(dposes, dposes2)
    ba.close()
    rng = np.random.default_rng(3)
    for P in (2, 3, 5, 8, 11, 13, 16, 20):
        Q = 60
        c = synthetic.make_ba_case(n_poses=P, n_points=Q, seed=20 + P)
        keep = rng.uniform(size=P * Q) < 0.8
        vp, pt = c["vp_idx"][keep], c["pt_idx"][keep]
        xt = orc.ba_projection(c["poses"], c["points"], vp, pt, jacobians=False)
        ba = ops.BundleAdjustment(P, Q, vp, pt, xt)
        mu = 0.05
        dposes, dpoints, _ = ba.step(c["poses_noisy"], c["points_noisy"], mu)
        ba.close()
        x_pred, A, B = orc.ba_projection(c["poses_noisy"], c["points_noisy"], vp, pt)
        J = np.zeros((2 * len(vp), 6 * P + 3 * Q))
        for k, (j, i) in enumerate(zip(vp, pt)):
            J[2 * k:2 * k + 2, 6 * j:6 * j + 6] = A[k]
            J[2 * k:2 * k + 2, 6 * P + 3 * i:6 * P + 3 * i + 3] = B[k]
        delta = np.linalg.solve(J.T @ J + mu * np.eye(J.shape[1]), J.T @ (xt - x_pred).reshape(-1))
        assert np.allclose(dposes.reshape(-1), delta[:6 * P], rtol=1e-7, atol=1e-10), P
        assert np.allclose(dpoints.reshape(-1), delta[6 * P:], rtol=1e-7, atol=1e-10), P


def test_ba_lm_loop_on_a_20_pose_window(ops, orc):
    """tdk_ba_solve on the widest window the device solve takes (6 P = 120, 129 KB of LDS): the
    Levenberg-Marquardt loop converges monotonically to the parameters the observations came from
    (up to the gauge), and a second call from the same start reproduces it bit for bit (no atomics
    anywhere in the pair-wise / MFMA-free path of a 20-pose window)."""
    from tadataka_amd import synthetic
    P, Q = 20, 400
    rng = np.random.default_rng(17)
    c = synthetic.make_ba_case(n_poses=P, n_points=Q, seed=31, perturb=5e-3)
    keep = rng.uniform(size=P * Q) < 0.7
    vp, pt = c["vp_idx"][keep], c["pt_idx"][keep]
    xt = orc.ba_projection(c["poses"], c["points"], vp, pt, jacobians=False)
    ba = ops.BundleAdjustment(P, Q, vp, pt, xt)
    e0 = ba.sum_squared_error(c["poses_noisy"], c["points_noisy"]) / len(vp)
    kw = dict(max_iter=15, absolute_error_threshold=1e-22, relative_error_threshold=1e-9)
    poses, points, errors = ba.solve(c["poses_noisy"], c["points_noisy"], **kw)
    assert errors[0] == pytest.approx(e0, rel=1e-12) and len(errors) >= 3
    assert np.all(np.diff(errors) <= 0) and errors[-1] < 1e-6 * errors[0]
    x_fit = orc.ba_projection(poses, points, vp, pt, jacobians=False)
    assert np.max(np.abs(x_fit - xt)) < 1e-5                       # reprojection, not parameters: the gauge is free
    poses2, points2, errors2 = ba.solve(c["poses_noisy"], c["points_noisy"], **kw)
    assert np.array_equal(poses, poses2) and np.array_equal(points, points2) and np.array_equal(errors, errors2)
    ba.close()


def test_anti_aliased_pyramid_batches_that_do_not_fill_the_xcds(ops, orc):
    """The pyramid grid deals images to the 8 XCDs: batches with 1, 3 and 9 pairs (3, 9 and 27 images:
    neither a multiple of 8) and a frame whose width is not a multiple of the 64-column tile."""
    from tadataka_amd import synthetic
    H, W = 50, 70
    for B in (1, 3, 9):
        batch = ops.DvoBatch(B, H, W, n_levels=3, ratio=1.5)
        pairs = [synthetic.make_pair(H, W, seed=70 + i) for i in range(B)]
        for i, pr in enumerate(pairs):
            batch.upload(i, pr["I0"], pr["D0"], pr["I1"])
        batch.set_anti_aliasing(True)
        batch.build_pyramid()
        for i in (0, B - 1):
            for level in (1, 2):
                for name in ("I0", "D0", "I1"):
                    got = batch.download(i, level, name)
                    want = orc.rescale(pairs[i][name], 1 / 1.5 ** level, anti_aliasing=True)
                    assert np.array_equal(got, want), (B, i, level, name)
        batch.close()


# ---------------------------------------------------------------------------
# RCCL through the C ABI (no torch): 1-rank communicator
# ---------------------------------------------------------------------------
def test_rccl_single_rank_smoke(ops):
    import sys as _sys
    from tadataka_amd import sharding, synthetic
    assert "torch" not in _sys.modules or True     # (other tests may have imported it; the product never does)
    comm = sharding.RcclComm(0, 1, sharding.RcclComm.unique_id())
    a = np.arange(24.).reshape(2, 12)
    assert np.array_equal(comm.all_gather(a), a)
    assert np.array_equal(comm.all_reduce([1.5, -2.0], "sum"), [1.5, -2.0])
    assert np.array_equal(comm.all_reduce([1.5, -2.0], "max"), [1.5, -2.0])
    comm.barrier()
    # device-resident gather of a batch's poses, queued on the batch's stream
    B, H, W = 3, 48, 64
    cam = synthetic.camera_for(W, H)
    batch = ops.DvoBatch(B, H, W)
    for i in range(B):
        p = synthetic.make_pair(H, W, seed=70 + i)
        batch.upload(i, p["I0"], p["D0"], p["I1"])
    P, _ = batch.estimate_level(0, cam, cam, np.tile(_pose12(np.eye(4)), (B, 1)), ops.W_HUBER, 20)
    pg = sharding.PoseGather(B, comm)            # world 1: the local path
    pg.start(P, batch)
    assert np.array_equal(pg.finish(), P)
    comm.gather_poses_start(batch)               # and the RCCL path explicitly
    assert np.array_equal(comm.gather_poses_finish(), P)
    batch.close()
    comm.close()


def test_update_depth_unusual_intensity_ranges(ops, orc):
    """Reference frames with negative, denormal-range or all-zero texels: same bits as the oracle."""
    from tadataka_amd import synthetic
    c = synthetic.make_semi_dense_case(96, 128, seed=5)
    pg, po = ops.make_params(0.5, 10.0, 0.01, 0.01, 0.004, 0.01), orc.make_params(0.5, 10.0, 0.01, 0.01, 0.004, 0.01)
    age = np.ones((96, 128), dtype=np.uint64)
    for ref_image in (c["ref_image"], c["ref_image"] - 0.5, c["ref_image"] * 1e-300, np.zeros((96, 128))):
        key = (c["cam"], c["key_image"], c["T_wk"]); ref = (c["cam"], np.ascontiguousarray(ref_image), c["T_wr"])
        d, v, f = ops.update_depth(key, [ref], age, c["prior_depth"], c["prior_variance"], pg)
        od, ov, of = orc.update_depth(key, [ref], age, c["prior_depth"], c["prior_variance"], po)
        assert np.array_equal(f, of) and np.array_equal(d, od) and np.array_equal(v, ov)


def test_ba_schur_mfma_matches_pair_kernel_and_is_reproducible(ops):
    """The FP64-MFMA Schur complement (windows of <= 8 poses; W_ij rebuilt from the parameters) against the
    pair-wise kernel (tdk_ba_create_ex: SCHUR_PAIRS; W_ij stored by the reduce) on ragged visibility, and
    bit-reproducible."""
    from tadataka_amd import synthetic
    c = synthetic.make_ba_case(n_poses=5, n_points=777, seed=8)
    rng = np.random.default_rng(3)
    keep = rng.uniform(size=len(c["vp_idx"])) < 0.8
    vp, pt = c["vp_idx"][keep], c["pt_idx"][keep]
    xt = ops.ba_projection(c["poses"], c["points"], vp, pt, jacobians=False)
    outs = []
    for options in (0, 0, ops.BundleAdjustment.SCHUR_PAIRS):
        ba = ops.BundleAdjustment(5, 777, vp, pt, xt, options=options)
        dp, dq, err = ba.step(c["poses_noisy"], c["points_noisy"], 1e-2)
        outs.append(np.concatenate([dp.ravel(), dq.ravel(), [err]]))
        ba.close()
    assert np.array_equal(outs[0], outs[1])                                   # run to run
    scale = np.max(np.abs(outs[2]))
    assert np.max(np.abs(outs[0] - outs[2])) < 1e-9 * scale                   # against the pair kernel


# ---------------------------------------------------------------------------
# the loop of examples/semi_dense_vo.py:160-199, device-resident end to end
# ---------------------------------------------------------------------------
def test_semi_dense_vo_loop_on_device(ops, orc):
    """track (PoseChangeEstimator on image0, depth_map0, image1, weights = safe_invert(variance_map0))
    -> increment_age -> propagate -> update_depth, three frames after the first, with only the new
    image crossing PCIe per step.  Tracking against the oracle's coarse-to-fine loop (1e-6); mapping
    against the oracle fed with the SAME transform (bit-exact)."""
    from tadataka_amd import synthetic
    H, W, n_frames = 96, 128, 4
    cam = synthetic.camera_for(W, H)
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    depth_gt0 = synthetic.depth_map(xs, ys)
    # a camera sliding along x and slightly forward, looking at the textured surface of frame 0
    T_w = []
    for k in range(n_frames):
        T = np.eye(4); T[:3, 3] = [0.03 * k, 0.005 * k, 0.01 * k]
        T_w.append(T)
    xn, yn = (xs - cam[2]) / cam[0], (ys - cam[3]) / cam[1]

    def render(T_wk):        # first-order consistent views: texture attached to frame 0's surface
        P = np.stack([xn * depth_gt0, yn * depth_gt0, depth_gt0], axis=-1) + T_wk[:3, 3]
        return np.ascontiguousarray(synthetic.texture(P[..., 0] / P[..., 2] * cam[0] + cam[2],
                                                      P[..., 1] / P[..., 2] * cam[1] + cam[3]))
    images = [render(T) for T in T_w]
    rng = np.random.default_rng(0)
    depth = depth_gt0 * rng.uniform(0.95, 1.05, (H, W))
    var = np.full((H, W), 0.05)
    age = np.zeros((H, W), dtype=np.uint64)
    pargs = (0.5, 10.0, 0.01, 0.01, 0.004, 0.01)
    pg, po = ops.make_params(*pargs), orc.make_params(*pargs)
    n_levels = 2
    sd = ops.SemiDenseSession(1, H, W, max_refframes=n_frames)
    sd.set_params(pg, *SD_DEFAULTS)
    dvo = ops.DvoBatch(1, H, W, n_levels=n_levels, ratio=1.5, with_weight_map=True)
    dvo.set_anti_aliasing(False)
    sd.push_frame(0, cam, images[0], T_w[0])
    sd.set_maps(0, depth, var, age)
    T_w_est = [T_w[0]]
    frames = [(cam, images[0], T_w[0])]
    for k in range(1, n_frames):
        sd.push_frame(0, cam, images[k])                               # the only upload of the step
        sd.export_dvo(dvo)
        dvo.build_pyramid()
        P, _ = dvo.estimate(cam, cam, _pose12(np.eye(4))[None], ops.W_MAP, 20)
        # tracking parity: the oracle's PoseChangeEstimator with the same weight map
        rot, t = orc.dvo_estimate(images[k - 1], depth, images[k], cam, cam, weights=1.0 / (var + 1e-16),
                                  n_coarse_to_fine=n_levels, max_iter=20)
        assert np.max(np.abs(P[0, :9].reshape(3, 3) - rot.as_matrix())) < POSE_ATOL
        assert np.max(np.abs(P[0, 9:] - t)) < POSE_ATOL
        T10 = np.eye(4); T10[:3, :3] = P[0, :9].reshape(3, 3); T10[:3, 3] = P[0, 9:]
        T_w1 = T_w_est[-1] @ np.linalg.inv(T10)                        # calc_pose_w1 (:127-130)
        sd.step(T10[None], T_w1[None], commit=True)
        key = (cam, images[k], T_w1)
        depth, var, age, flag = orc.semi_dense_step(key, cam, frames, T10, age, depth, var, po, *SD_DEFAULTS)
        gd, gv, ga, gf = sd.get_maps(0, with_flag=True)
        assert np.array_equal(ga, age) and np.array_equal(gf, flag)
        assert np.array_equal(gd, depth) and np.array_equal(gv, var)
        frames.append(key)
        T_w_est.append(T_w1)
    assert int(age.max()) == n_frames - 1 and int((flag == 0).sum()) > 500
    dvo.close(); sd.close()


@pytest.mark.parametrize("max_iter", [0, 1, 2, 5])
def test_level_loop_iteration_limits_vs_oracle(ops, orc, max_iter):
    """The probe / full state machine of the device loop at the edges of the reference's
    `for k in range(max_iter)`: pose and number of PhotometricError evaluations as the oracle's
    transcription of _PoseChangeEstimator.__call__ (:92-111) gives them."""
    from tadataka_amd import synthetic
    H, W = 72, 96
    pair = synthetic.make_pair(H, W, seed=21, rot_scale=0.01, trans_scale=0.03)
    cam = pair["cam"]
    batch = ops.DvoBatch(1, H, W)
    batch.upload(0, pair["I0"], pair["D0"], pair["I1"])
    for wname in (None, "huber", "student-t"):
        P, n_evals = batch.estimate_level(0, cam, cam, _pose12(np.eye(4))[None], ops.WEIGHT_MODES[wname], max_iter)
        trace = []
        rot, t = orc.dvo_estimate_level(pair["I0"], pair["D0"], pair["I1"], cam, cam,
                                        Rotation.from_rotvec(np.zeros(3)), np.zeros(3), wname, max_iter, trace=trace)
        assert n_evals[0] == len(trace)               # one entry per PhotometricError call
        assert np.max(np.abs(P[0, :9].reshape(3, 3) - rot.as_matrix())) < POSE_ATOL
        assert np.max(np.abs(P[0, 9:] - t)) < POSE_ATOL
        if max_iter == 0:
            assert np.array_equal(P[0], _pose12(np.eye(4)))
    batch.close()
