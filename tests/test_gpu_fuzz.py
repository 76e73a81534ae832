"""Randomised parity soak: every kernel family of the HIP path against the CPU oracle on random
shapes (down to 2 x 2), random poses (small, large, behind the camera), degenerate depths (0, negative,
NaN, huge), ragged graphs -- the inputs nobody wrote a fixture for.

TDK_FUZZ_N cases per family (default 2 000: about a minute of the regular `pytest -m gpu` run; the soaks of
profiles/r05_fuzz.txt ran 4000 - 6000 on two dozen seeds, ~3.3 min per seed on the GPU box).  TDK_FUZZ_SEED moves the
whole sequence (default: rotates with the built library, tests/conftest.py: fuzz_settings; printed in the header
and the last lines of the run), TDK_FUZZ_BIG=1 draws production-size frames and batches for the batch families.  Bars as everywhere: bit-exact for the
parity-granular operators, the pyramid and the semi-dense maps; 1e-9 per entry on the normal equations;
1e-6 on poses."""
import os

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from conftest import b6_err, fuzz_settings, h21_err

pytestmark = pytest.mark.gpu

N_CASES, SEED = fuzz_settings()
RTOL_SUMS = 1e-9


@pytest.fixture(scope="module")
def ops():
    from tadataka_amd import _lib, ops as o
    _lib.require_gpu()
    return o


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def _pose12(R, t):
    return np.concatenate([np.asarray(R).ravel(), np.asarray(t).ravel()])


def _T(R, t):
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return T


def _random_pose(rng):
    kind = rng.integers(0, 6)
    if kind == 0:
        return np.eye(3), np.zeros(3)
    scale_r = [0.01, 0.01, 0.1, 0.5, 3.0][kind - 1]
    scale_t = [0.02, 0.2, 0.1, 1.0, 3.0][kind - 1]
    R = Rotation.from_rotvec(rng.uniform(-scale_r, scale_r, 3)).as_matrix()
    return R, rng.uniform(-scale_t, scale_t, 3)


def _random_scene(rng, H, W):
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    a, b, c = rng.uniform(2, 9, 3)
    I1 = 0.5 + 0.25 * np.sin(xx / a) * np.cos(yy / b) + 0.2 * np.sin((xx + yy) / c) + 0.02 * rng.uniform(-1, 1, (H, W))
    I0 = 0.5 + 0.25 * np.sin((xx + 0.3) / a) * np.cos(yy / b) + 0.2 * np.sin((xx + yy) / c) + 0.05 * rng.uniform(-1, 1, (H, W))
    D0 = 2.0 + 0.3 * np.sin(xx / 11) + 0.2 * np.cos(yy / 7) + rng.uniform(-0.05, 0.05, (H, W))
    dirt = rng.integers(0, 5)
    if dirt == 1:      # what a depth sensor reports
        D0[rng.random((H, W)) < 0.1] = 0.0
    elif dirt == 2:
        D0[rng.random((H, W)) < 0.05] = np.nan
        D0[rng.random((H, W)) < 0.05] = -1.0
    elif dirt == 3:
        D0[rng.random((H, W)) < 0.03] = 1e300
        D0[rng.random((H, W)) < 0.03] = 1e-300
    if rng.random() < 0.15:    # outliers beyond Huber's k
        I0[rng.random((H, W)) < 0.05] += rng.choice([-3.0, 3.0])
    f = rng.uniform(0.5, 2.0) * max(H, W)
    cam = np.array([f, f * rng.uniform(0.9, 1.1), W / 2 + rng.uniform(-1, 1), H / 2 + rng.uniform(-1, 1)])
    return I0, D0, I1, cam


def _same(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


# ---------------------------------------------------------------------------
# fused DVO evaluation and the error-only probe
# ---------------------------------------------------------------------------
def test_fuzz_dvo_evaluate(ops, orc):
    rng = np.random.default_rng(1000 + SEED)
    worst = 0.0
    for case in range(N_CASES):
        H, W = (int(v) for v in rng.integers(2, 97, 2))
        if rng.random() < 0.2:
            H, W = int(rng.integers(2, 5)), int(rng.integers(2, 40))
        I0, D0, I1, cam = _random_scene(rng, H, W)
        wmap = rng.uniform(0.0, 2.0, (H, W))
        R, t = _random_pose(rng)
        GX, GY = orc.image_gradient(I1)
        batch = ops.DvoBatch(1, H, W, with_weight_map=True)
        batch.upload(0, I0, D0, I1, wmap)
        P = _pose12(R, t)[None]
        info = f"case {case}: {H}x{W} pose {Rotation.from_matrix(R).as_rotvec()} {t}"
        s_ref, n_ref = orc.photometric_error_sums(I0, D0, I1, cam, cam, _T(R, t))
        ss, ne = batch.photometric_error(0, cam, cam, P)
        assert ne[0] == n_ref, info
        if n_ref and np.isfinite(s_ref):
            assert abs(ss[0] - s_ref) <= RTOL_SUMS * abs(s_ref), info
        for wname in (None, "huber", "map", "student-t", "tukey"):
            weights = wmap if wname == "map" else wname
            mode = ops.W_MAP if wname == "map" else ops.WEIGHT_MODES[wname]
            ev = batch.evaluate(0, cam, cam, P, mode)
            assert ev["n_error"][0] == n_ref, (info, wname)
            if n_ref and np.isfinite(s_ref):
                assert abs(ev["sum_sq"][0] - s_ref) <= RTOL_SUMS * abs(s_ref), (info, wname)
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                Href, bref, M = orc.dvo_normal_equations(I0, D0, I1, GX, GY, cam, cam, R, t, weights)
            assert ev["n_update"][0] == M, (info, wname)
            if M < 12 and wname in ("student-t", "tukey"):
                continue        # the statistics of a handful of residuals (a zero MAD, ...) are not the point here
            if M == 0 or not (np.all(np.isfinite(Href)) and np.all(np.isfinite(bref))):
                continue
            if np.min(Href[[0, 6, 11, 15, 18, 20]]) <= 0.0:
                continue
            e1 = h21_err(ev["H"][0], Href)
            e2 = b6_err(ev["b"][0], bref, Href)
            worst = max(worst, e1, e2)
            assert e1 < RTOL_SUMS and e2 < RTOL_SUMS, (info, wname, e1, e2)
        batch.close()
    print(f"dvo evaluate: {N_CASES} cases, worst per-entry error {worst:.2e}")


def test_fuzz_dvo_estimate_small_scenes(ops, orc):
    """The device Gauss-Newton loop against the oracle's reference-structured loop (lstsq on J, the same accept rule)
    on random small pairs with a known pose, one and two pyramid levels, ideal-constants pyramid on both sides."""
    from tadataka_amd import synthetic
    rng = np.random.default_rng(2000 + SEED)
    n = max(2, N_CASES // 3)
    mism = 0
    for case in range(n):
        H, W = int(rng.integers(24, 80)), int(rng.integers(32, 100))
        pair = synthetic.make_pair(H, W, seed=int(rng.integers(0, 1 << 30)))
        levels = int(rng.integers(1, 4))
        wname = [None, "huber", "student-t", "tukey"][int(rng.integers(0, 4))]
        cam = pair["cam"]
        if rng.random() < 0.7:      # a camera whose scaled parameters are not exact in binary (the pair's own is f = 525 W / 640, o = W / 2)
            cam = cam * rng.uniform(0.97, 1.03, 4) + np.array([0, 0, rng.uniform(-1, 1), rng.uniform(-1, 1)])
        # a prior other than the identity (what a tracker hands over) and the iteration limits of the reference's tests
        max_iter = int(rng.choice([20, 20, 5, 2, 1]))
        R0, t0 = np.eye(3), np.zeros(3)
        if rng.random() < 0.4:
            R0 = Rotation.from_rotvec(rng.uniform(-0.004, 0.004, 3)).as_matrix()
            t0 = rng.uniform(-0.008, 0.008, 3)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            Rr, tr = orc.dvo_estimate(pair["I0"], pair["D0"], pair["I1"], cam, cam, wname, n_coarse_to_fine=levels,
                                      max_iter=max_iter, anti_aliasing=True, rotation=Rotation.from_matrix(R0), t=t0)[:2]
        batch = ops.DvoBatch(1, H, W, n_levels=levels)
        batch.upload(0, pair["I0"], pair["D0"], pair["I1"])
        batch.build_pyramid()
        P, _ = batch.estimate(cam, cam, _pose12(R0, t0)[None], ops.WEIGHT_MODES[wname], max_iter)
        batch.close()
        Rr = Rr.as_matrix() if hasattr(Rr, "as_matrix") else np.asarray(Rr)
        d = max(np.max(np.abs(P[0, :9].reshape(3, 3) - Rr)), np.max(np.abs(P[0, 9:] - tr)))
        if d >= 1e-6:
            mism += 1
        assert d < 1e-6, (case, H, W, levels, wname, max_iter, d)
    print(f"dvo estimate: {n} scenes, {mism} beyond 1e-6")


# ---------------------------------------------------------------------------
# skimage.transform.rescale on the device
# ---------------------------------------------------------------------------
def test_fuzz_rescale_skimage_bit_exact(ops, orc):
    from tadataka_amd import rescale_plan
    rng = np.random.default_rng(3000 + SEED)
    for case in range(N_CASES):
        H, W = (int(v) for v in rng.integers(1, 75, 2))
        scale = float(rng.choice([1 / 1.5, 1 / 2.25, 0.5, 1.0, rng.uniform(0.08, 1.0), rng.uniform(1.0, 2.5)]))
        aa = bool(rng.integers(0, 2))
        clip = bool(rng.integers(0, 2))
        img = rng.uniform(0, 1, (H, W))
        kind = rng.integers(0, 6)
        if kind == 1:
            img[:] = rng.uniform(0.5, 3.0)                       # a plateau: clip=True decides the last bit
        elif kind == 2:
            img = np.round(img * 4) / 4                          # plateaus at the extremes
        elif kind == 3 and H * W > 1:
            img[rng.integers(0, H), rng.integers(0, W)] = np.nan
        elif kind == 4:
            img *= 1e5
        if min(np.round(H * scale), np.round(W * scale)) < 1:
            continue                                             # scikit-image raises for an empty output
        out_shape = ops.rescale_shape(img.shape, scale)
        plan = rescale_plan.resize_plan(img.shape, out_shape, aa)
        got = ops.rescale(img, scale, anti_aliasing=aa, mode="skimage", plan=plan, clip=clip)
        want = orc.rescale_skimage(img, scale, plan=plan, anti_aliasing=aa, clip=clip)
        assert got.shape == want.shape, (case, H, W, scale)
        assert _same(got, want), (case, H, W, scale, aa, clip, int(kind), np.nanmax(np.abs(got - want)))


def test_fuzz_batch_pyramid_bit_exact(ops, orc):
    """The batch pyramid (streaming kernel / tiles / general kernel, level 0 through rescale(., 1.0), clip) on
    random shapes, level counts and batch sizes: every level of every array against the oracle's rescale."""
    from tadataka_amd import rescale_plan
    rng = np.random.default_rng(3500 + SEED)
    n = max(2, N_CASES // 3)
    big = os.environ.get("TDK_FUZZ_BIG") == "1"      # frames of several strips / row segments, batches that fill the chip
    for case in range(n):
        H, W = int(rng.integers(8, 130)), int(rng.integers(8, 300))
        if big:
            H, W = int(rng.integers(100, 520)), int(rng.integers(200, 900))
        levels = int(rng.integers(1, 5))
        ratio = float(rng.choice([1.5, 2.0, 1.3]))
        while min(H, W) / ratio ** (levels - 1) < 3:
            levels -= 1
        B = int(rng.choice([1, 2, 9, 40]))
        if big:
            B = int(rng.choice([3, 30, 90]))
        batch = ops.DvoBatch(B, H, W, n_levels=levels, ratio=ratio)
        plans = rescale_plan.level_plans((H, W), levels, ratio, True)
        batch.set_skimage_pyramid(plans)
        frames = []
        for p in range(B):
            I0, D0, I1, _ = _random_scene(rng, H, W)
            if rng.random() < 0.3:
                D0[:] = 2.0
            batch.upload(p, I0, D0, I1)
            frames.append((I0, D0, I1))
        ops.set_option("pyramid_stream", int(rng.choice([0, 1, 2])))
        batch.build_pyramid()
        ops.set_option("pyramid_stream", 1)
        for p in sorted(set([0, B - 1, int(rng.integers(0, B))])):
            for l in range(levels):
                for k, name in enumerate(("I0", "D0", "I1")):
                    got = batch.download(p, l, name)
                    want = orc.rescale_skimage(frames[p][k], 1.0 / ratio ** l, plan=plans[l], anti_aliasing=True, clip=True)
                    assert _same(got, want), (case, H, W, levels, ratio, B, p, l, name)
        batch.close()


# ---------------------------------------------------------------------------
# granular operators: bit-exact, special values included
# ---------------------------------------------------------------------------
def test_fuzz_granular_bit_exact(ops, orc):
    rng = np.random.default_rng(4000 + SEED)
    specials = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 5e-324, 1e-310, 1e308, -1e308, 1.0, -1.0])
    for case in range(N_CASES):
        n = int(rng.integers(1, 3000))
        cam = np.array([rng.uniform(100, 900), rng.uniform(100, 900), rng.uniform(0, 640), rng.uniform(0, 480)])
        kp = rng.uniform(-1e3, 1e3, (n, 2))
        P = rng.uniform(-5, 5, (n, 3))
        xs = rng.uniform(-2, 2, (n, 2))
        d = rng.uniform(-1, 6, n)
        for a in (kp, P, xs, d):
            m = rng.random(a.shape) < 0.05
            a[m] = rng.choice(specials, int(m.sum()))
        R, t = _random_pose(rng)
        T = _T(R, t)
        assert _same(ops.normalize(kp, cam), orc.normalize(kp, cam)), case
        assert _same(ops.unnormalize(kp, cam), orc.unnormalize(kp, cam)), case
        assert _same(ops.project_vecs(P), orc.project_vecs(P)), case
        assert _same(ops.inv_project_vecs(xs, d), orc.inv_project_vecs(xs, d)), case
        assert _same(ops.transform(T, P), orc.transform(T, P)), case
        a, b = ops.warp_vecs(T, xs, d)
        c, e = orc.warp_vecs(T, xs, d)
        assert _same(a, c) and _same(b, e), case
        for i in range(min(n - 1, 40)):                           # src/triangulation.rs:8-39, one correspondence per call
            g0, o0 = ops.calc_depth0(T, xs[i], xs[i + 1]), orc.calc_depth0(T, xs[i], xs[i + 1])
            assert g0 == o0 or (g0 != g0 and o0 != o0), (case, i, g0, o0)
        H, W = (int(v) for v in rng.integers(2, 60, 2))
        img = rng.uniform(-1, 1, (H, W))
        m = rng.random(img.shape) < 0.05
        img[m] = rng.choice(specials, int(m.sum()))
        c2 = np.column_stack([rng.uniform(0, W - 1, 500), rng.uniform(0, H - 1, 500)])
        c2[:50] = np.floor(c2[:50])
        c2[50] = [W - 1, H - 1]
        g_, o_ = ops.interpolation(img, c2), orc.interpolation(img, c2)
        assert _same(g_, o_) and np.array_equal(np.signbit(g_), np.signbit(o_)), case   # the sign of a zero included
        gx, gy = ops.image_gradient(img)
        ogx, ogy = orc.image_gradient(img)
        assert _same(gx, ogx) and _same(gy, ogy), case
        sx, sy = ops.sobel(img)
        osx, osy = orc.sobel(img)
        assert _same(sx, osx) and _same(sy, osy), case


# ---------------------------------------------------------------------------
# semi-dense maps: bit-exact
# ---------------------------------------------------------------------------
def test_fuzz_semi_dense_warp_bit_exact(ops, orc):
    rng = np.random.default_rng(5000 + SEED)
    for case in range(N_CASES):
        H, W = (int(v) for v in rng.integers(2, 120, 2))
        f = rng.uniform(0.4, 1.5) * max(H, W)
        cam0 = np.array([f, f, W / 2, H / 2])
        cam1 = cam0 * rng.choice([1.0, 1.0, 0.5, 1.3]) if rng.random() < 0.5 else cam0
        kind = rng.integers(0, 5)
        if kind == 0:
            R, t = np.eye(3), np.array([rng.uniform(-0.3, 0.3), 0, 0])          # stereo
        elif kind == 1:
            R, t = np.eye(3), np.array([0, 0, rng.uniform(0.5, 2.0)])           # zoom out: many sources per target
        elif kind == 2:
            R, t = np.eye(3), np.array([0, 0, -rng.uniform(0.2, 1.0)])          # zoom in
        else:
            R, t = _random_pose(rng)
        T10 = _T(R, t)
        depth0 = rng.uniform(0.8, 4.0, (H, W))
        if rng.random() < 0.3:
            depth0 = np.full((H, W), 2.0) + rng.uniform(-0.01, 0.01, (H, W))
        if rng.random() < 0.2:
            depth0[rng.random((H, W)) < 0.1] = rng.choice([0.0, -1.0, np.nan])
        var0 = rng.uniform(0.01, 1.0, (H, W))
        age0 = rng.integers(0, 5, (H, W)).astype(np.uint64)
        for gather in (1, 0):
            ops.set_option("sd_warp_gather", gather)
            a = ops.increment_age(age0, cam0, cam1, T10, depth0)
            d1, v1 = ops.propagate(T10, cam0, cam1, depth0, var0, 1.5, 10.0, 0.01)
            ops.set_option("sd_warp_gather", 1)
            assert np.array_equal(a, orc.increment_age(age0, cam0, cam1, T10, depth0)), (case, H, W, int(kind), gather)
            od, ov = orc.propagate(T10, cam0, cam1, depth0, var0, 1.5, 10.0, 0.01)
            assert _same(d1, od) and _same(v1, ov), (case, H, W, int(kind), gather)


def test_fuzz_update_depth_bit_exact(ops, orc):
    rng = np.random.default_rng(6000 + SEED)
    n = max(2, N_CASES // 2)
    for case in range(n):
        H, W = int(rng.integers(8, 90)), int(rng.integers(8, 120))
        f = rng.uniform(0.6, 1.4) * max(H, W)
        cam = np.array([f, f, W / 2, H / 2])
        n_ref = int(rng.integers(1, 5))
        yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)

        def image():
            a, b = rng.uniform(1.5, 6, 2)
            return 0.5 + 0.3 * np.sin(xx / a) * np.cos(yy / b) + 0.1 * rng.uniform(-1, 1, (H, W))

        T_wk = _T(*_random_pose(rng)) if rng.random() < 0.5 else np.eye(4)
        key = (cam, image(), T_wk)
        refs = []
        for r in range(n_ref):
            dT = _T(Rotation.from_rotvec(rng.uniform(-0.02, 0.02, 3)).as_matrix(), rng.uniform(-0.15, 0.15, 3))
            if rng.random() < 0.15:
                dT = np.eye(4)                                    # identical poses: the epipole is at infinity / zero baseline
            refs.append((cam, image(), T_wk @ dT))
        age = rng.integers(0, n_ref + 1, (H, W)).astype(np.uint64)
        prior_depth = rng.uniform(0.6, 6.0, (H, W))
        prior_var = rng.uniform(1e-3, 0.5, (H, W))
        if rng.random() < 0.3:
            prior_depth[rng.random((H, W)) < 0.05] = rng.choice([0.0, -1.0])
            prior_var[rng.random((H, W)) < 0.05] = rng.choice([0.0, -0.1])
        pa = (rng.uniform(0.2, 1.0), rng.uniform(4, 12), rng.uniform(0.001, 0.1), rng.uniform(0.001, 0.1),
              rng.uniform(0.3, 2.0) / f, rng.uniform(0.0, 0.1))
        g = ops.update_depth(key, refs, age, prior_depth, prior_var, ops.make_params(*pa))
        o = orc.update_depth(key, refs, age, prior_depth, prior_var, orc.make_params(*pa))
        assert np.array_equal(g[2], o[2]), (case, H, W, n_ref, np.argwhere(g[2] != o[2])[:5])
        assert _same(g[0], o[0]) and _same(g[1], o[1]), (case, H, W, n_ref)


# ---------------------------------------------------------------------------
# bundle adjustment: block sums on ragged graphs
# ---------------------------------------------------------------------------
def test_fuzz_ba_block_sums(ops, orc):
    rng = np.random.default_rng(7000 + SEED)
    worst = 0.0
    for case in range(N_CASES):
        nP = int(rng.integers(1, 24))
        nQ = int(rng.integers(1, 3000))
        poses = np.column_stack([rng.uniform(-0.3, 0.3, (nP, 3)), rng.uniform(-1, 1, (nP, 3))])
        if rng.random() < 0.3:
            poses[rng.integers(0, nP), :3] = 0.0                  # |omega| = 0: the epsilon branch of Rodrigues
        points = np.column_stack([rng.uniform(-5, 5, (nQ, 2)), rng.uniform(4, 12, nQ)])
        vis = rng.random((nP, nQ)) < rng.choice([1.0, 0.6, 0.1])
        if rng.random() < 0.5 and nP > 1:
            vis[rng.integers(0, nP)] = False                      # a pose nobody observes from
        # (an observation whose point lies almost in the camera's plane is conditioned like 1 / z': both sides are
        #  right to rounding and 1e-9 apart -- keep the depths the reference's scenes have)
        for j in range(nP):
            zc = (orc.exp_so3(poses[j, :3]) @ points.T)[2] + poses[j, 5]
            vis[j, zc < 1.0] = False
        vp, pt = np.nonzero(vis)
        if rng.random() < 0.3:                                    # point-major order instead of pose-major
            o = np.lexsort((vp, pt))
            vp, pt = vp[o], pt[o]
        if len(vp) == 0:
            continue
        x_true = orc.ba_projection(poses, points, vp, pt, jacobians=False) + rng.normal(0, 1e-3, (len(vp), 2))
        g = ops.ba_block_reduce(poses, points, x_true, vp, pt)
        o = orc.ba_block_reduce(poses, points, x_true, vp, pt)
        for name, a, b in zip("U ea V eb".split(), g[:4], o[:4]):
            scale = max(np.max(np.abs(b)), 1e-300)
            e = np.max(np.abs(a - b)) / scale
            worst = max(worst, e)
            assert e < RTOL_SUMS, (case, nP, nQ, name, e)
        assert abs(g[4] - o[4]) <= RTOL_SUMS * abs(o[4]), case
        ba = ops.BundleAdjustment(nP, nQ, vp, pt, x_true)
        s = ba.block_sums(poses, points)
        ba.close()
        for name, a, b in zip("U ea V eb".split(), s[:4], o[:4]):
            scale = max(np.max(np.abs(b)), 1e-300)
            assert np.max(np.abs(np.asarray(a).reshape(b.shape) - b)) / scale < RTOL_SUMS, (case, nP, nQ, name, "handle")
    print(f"ba block sums: worst relative error {worst:.2e}")


# ---------------------------------------------------------------------------
# batches: several pairs with their own cameras and poses, evaluated at a random pyramid level
# ---------------------------------------------------------------------------
def test_fuzz_dvo_batches_at_pyramid_levels(ops, orc):
    rng = np.random.default_rng(8000 + SEED)
    n = max(2, N_CASES // 6)
    worst = 0.0
    big = os.environ.get("TDK_FUZZ_BIG") == "1"      # production sizes: the planner's long blocks, many blocks per pair
    for case in range(n):
        H, W = int(rng.integers(30, 200)), int(rng.integers(30, 330))
        B = int(rng.choice([2, 3, 8, 9, 17]))
        if big:
            H, W = int(rng.integers(300, 500)), int(rng.integers(400, 700))
            B = int(rng.choice([9, 40, 70]))
        levels = int(rng.integers(1, 4))
        while min(H, W) / 1.5 ** (levels - 1) < 8:
            levels -= 1
        batch = ops.DvoBatch(B, H, W, n_levels=levels, with_weight_map=True)
        cams, poses, Ts = [], [], []
        for p in range(B):
            I0, D0, I1, cam = _random_scene(rng, H, W)
            batch.upload(p, I0, D0, I1, rng.uniform(0.1, 2.0, (H, W)))
            R, t = _random_pose(rng)
            cams.append(cam); poses.append(_pose12(R, t)); Ts.append((R, t))
        batch.build_pyramid()
        cams = np.array(cams); poses = np.array(poses)
        level = int(rng.integers(0, levels))
        scale = 1.0 / 1.5 ** level
        wname = [None, "huber", "map", "student-t", "tukey"][int(rng.integers(0, 5))]
        mode = ops.W_MAP if wname == "map" else ops.WEIGHT_MODES[wname]
        ev = batch.evaluate(level, cams, cams, poses, mode)
        ss, ne = batch.photometric_error(level, cams, cams, poses)
        for p in (range(B) if not big else sorted(set(int(v) for v in rng.integers(0, B, 6)))):
            I0, D0, I1, W0 = (batch.download(p, level, k) for k in ("I0", "D0", "I1", "W0"))
            cam = cams[p] * scale
            R, t = Ts[p]
            s_ref, n_ref = orc.photometric_error_sums(I0, D0, I1, cam, cam, _T(R, t))
            assert ne[p] == n_ref and ev["n_error"][p] == n_ref, (case, p)
            if n_ref and np.isfinite(s_ref):
                assert abs(ss[p] - s_ref) <= RTOL_SUMS * abs(s_ref), (case, p)
                assert abs(ev["sum_sq"][p] - s_ref) <= RTOL_SUMS * abs(s_ref), (case, p)
            GX, GY = orc.image_gradient(I1)
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                Href, bref, M = orc.dvo_normal_equations(I0, D0, I1, GX, GY, cam, cam, R, t, W0 if wname == "map" else wname)
            assert ev["n_update"][p] == M, (case, p, wname)
            if M < 12 or not (np.all(np.isfinite(Href)) and np.all(np.isfinite(bref))):
                continue
            if np.min(Href[[0, 6, 11, 15, 18, 20]]) <= 0.0:
                continue
            e1, e2 = h21_err(ev["H"][p], Href), b6_err(ev["b"][p], bref, Href)
            worst = max(worst, e1, e2)
            assert e1 < RTOL_SUMS and e2 < RTOL_SUMS, (case, p, H, W, B, level, wname, e1, e2)
        batch.close()
    print(f"dvo batches: {n} batches, worst per-entry error {worst:.2e}")


# ---------------------------------------------------------------------------
# the semi-dense session: chained steps of several tracks against the oracle chain
# ---------------------------------------------------------------------------
def test_fuzz_sd_session_chain(ops, orc):
    rng = np.random.default_rng(9000 + SEED)
    n_cases = max(1, N_CASES // 8)
    defaults = (1.5, 10.0, 0.01)
    big = os.environ.get("TDK_FUZZ_BIG") == "1"
    for case in range(n_cases):
        H, W = int(rng.integers(12, 110)), int(rng.integers(12, 150))
        n_tracks = int(rng.choice([1, 2, 5]))
        if big:
            H, W, n_tracks = int(rng.integers(300, 480)), int(rng.integers(400, 640)), int(rng.choice([3, 9]))
        n_steps = int(rng.integers(1, 4))
        f = rng.uniform(0.6, 1.3) * max(H, W)
        cam = np.array([f, f, W / 2, H / 2])
        pa = (0.3, 12.0, rng.uniform(0.005, 0.05), rng.uniform(0.005, 0.05), rng.uniform(0.5, 1.5) / f, rng.uniform(0.0, 0.05))
        pg, po = ops.make_params(*pa), orc.make_params(*pa)
        sd = ops.SemiDenseSession(n_tracks, H, W, max_refframes=n_steps)
        sd.set_age_policy(False)
        sd.set_params(pg, *defaults)
        ops.set_option("sd_warp_gather", int(rng.integers(0, 2)))
        yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
        state = []
        for t in range(n_tracks):
            a, b = rng.uniform(1.5, 6, 2)
            frames, T = [], np.eye(4)
            for s in range(n_steps + 1):
                img = 0.5 + 0.3 * np.sin((xx + 0.7 * s) / a) * np.cos(yy / b) + 0.05 * rng.uniform(-1, 1, (H, W))
                frames.append((cam, img, T.copy()))
                T = T @ _T(Rotation.from_rotvec(rng.uniform(-0.01, 0.01, 3)).as_matrix(), rng.uniform(-0.08, 0.08, 3))
            depth = rng.uniform(0.8, 5.0, (H, W))
            var = rng.uniform(0.01, 0.3, (H, W))
            age = (rng.random((H, W)) < rng.uniform(0.1, 0.9)).astype(np.uint64) * 0      # init_age: zeros
            sd.push_frame(t, *frames[0])
            sd.set_maps(t, depth, var, age)
            state.append(dict(frames=frames, depth=depth, var=var, age=age))
        for step in range(1, n_steps + 1):
            T10s, Twfs = [], []
            for t in range(n_tracks):
                fr = state[t]["frames"]
                sd.push_frame(t, fr[step][0], fr[step][1])
                T10s.append(np.linalg.inv(fr[step][2]) @ fr[step - 1][2])
                Twfs.append(fr[step][2])
            sd.step(np.array(T10s), np.array(Twfs), commit=True)
            for t in range(n_tracks):
                st = state[t]
                fr = st["frames"]
                d, v, a, f_ = orc.semi_dense_step(fr[step], fr[step - 1][0], fr[:step], T10s[t], st["age"], st["depth"],
                                                  st["var"], po, *defaults)
                gd, gv, ga, gf = sd.get_maps(t, with_flag=True)
                assert np.array_equal(ga, a) and np.array_equal(gf, f_), (case, H, W, t, step)
                assert _same(gd, d) and _same(gv, v), (case, H, W, t, step)
                st.update(depth=d, var=v, age=a)
        ops.set_option("sd_warp_gather", 1)
        sd.close()


# ---------------------------------------------------------------------------
# bundle adjustment: one damped step against the dense normal equations
# ---------------------------------------------------------------------------
def test_fuzz_ba_damped_step(ops, orc):
    rng = np.random.default_rng(10000 + SEED)
    n = max(2, N_CASES // 4)
    for case in range(n):
        P = int(rng.integers(2, 22))
        Q = int(rng.integers(8, 90))
        poses = np.column_stack([rng.uniform(-0.2, 0.2, (P, 3)), rng.uniform(-1, 1, (P, 3))])
        points = np.column_stack([rng.uniform(-4, 4, (Q, 2)), rng.uniform(4, 12, Q)])
        vis = rng.random((P, Q)) < rng.choice([1.0, 0.8, 0.5])
        vis[:, vis.sum(0) == 0] = True                      # every point is seen
        vis[vis.sum(1) == 0, :] = True                      # every pose sees something (mu > 0 anyway)
        for j in range(P):
            zc = (orc.exp_so3(poses[j, :3]) @ points.T)[2] + poses[j, 5]
            vis[j, zc < 1.0] = False
        vp, pt = np.nonzero(vis)
        xt = orc.ba_projection(poses, points, vp, pt, jacobians=False)
        pn = poses + rng.normal(0, 1e-3, poses.shape)
        qn = points + rng.normal(0, 1e-3, points.shape)
        mu = float(rng.choice([1e-3, 0.05, 1.0, 30.0]))
        options = int(rng.choice([0, 0, ops.BundleAdjustment.SCHUR_PAIRS, ops.BundleAdjustment.SCHUR_GENERAL,
                                  ops.BundleAdjustment.SOLVE_HOST, ops.BundleAdjustment.SOLVE_PIVOTED]))
        ba = ops.BundleAdjustment(P, Q, vp, pt, xt, options=options)
        dposes, dpoints, err = ba.step(pn, qn, mu)
        ba.close()
        x_pred, A, B = orc.ba_projection(pn, qn, vp, pt)
        J = np.zeros((2 * len(vp), 6 * P + 3 * Q))
        for k, (j, i) in enumerate(zip(vp, pt)):
            J[2 * k:2 * k + 2, 6 * j:6 * j + 6] = A[k]
            J[2 * k:2 * k + 2, 6 * P + 3 * i:6 * P + 3 * i + 3] = B[k]
        r = (xt - x_pred).reshape(-1)
        delta = np.linalg.solve(J.T @ J + mu * np.eye(J.shape[1]), J.T @ r)
        assert err == pytest.approx(float(r @ r), rel=1e-10), (case, P, Q)
        assert np.allclose(dposes.reshape(-1), delta[:6 * P], rtol=1e-6, atol=1e-10), (case, P, Q, mu, options)
        assert np.allclose(dpoints.reshape(-1), delta[6 * P:], rtol=1e-6, atol=1e-10), (case, P, Q, mu, options)


# ---------------------------------------------------------------------------
# array-level operators: robust weights (tadataka/robust/weights.py restated in NumPy), weighted normal equations,
# the N4 post-steps, rgb2gray, estimate_debug_
# ---------------------------------------------------------------------------
def _weights_numpy(r, mode):
    if mode == "huber":
        w = np.ones(r.shape); a = np.abs(r); m = a > 1.345
        w[m] = 1.345 / a[m]
        return w
    if mode == "student-t":
        s = r * r
        v = 1.0
        for _ in range(10):
            v = np.mean(s * (6.0 / (5.0 + s / v)))
        return np.sqrt(6.0 / (5.0 + s / v))
    sigma = 1.4826 * np.median(np.abs(r - np.median(r)))
    x = r / sigma
    w = np.zeros(r.shape)
    m = np.abs(x) <= 4.6851
    w[m] = (1 - (x[m] / 4.6851) ** 2) ** 2
    return w


def test_fuzz_array_level_operators(ops, orc):
    rng = np.random.default_rng(11000 + SEED)
    for case in range(N_CASES):
        n = int(rng.choice([1, 2, 3, 7, 64, 255, 256, 257, 1000, 4097, 20011]))
        kind = rng.integers(0, 5)
        r = rng.normal(0, rng.uniform(0.01, 2.0), n)
        if kind == 1:
            r = np.round(r * 8) / 8                               # ties: medians of repeated values
        elif kind == 2 and n > 4:
            r[rng.random(n) < 0.1] *= 50.0                        # heavy tail
        elif kind == 3:
            r = np.abs(r) + 0.5                                   # one-sided
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for mode in ("huber", "student-t", "tukey"):
                want = _weights_numpy(r, mode)
                got = ops.robust_weights(r, ops.WEIGHT_MODES[mode])
                if mode == "huber":
                    assert np.array_equal(got, want), (case, n, mode)
                elif np.all(np.isfinite(want)):
                    assert np.allclose(got, want, rtol=1e-11, atol=1e-14), (case, n, mode, int(kind), np.max(np.abs(got - want)))
        p = int(rng.integers(1, 9))
        A = rng.normal(0, 1, (n, p)) * rng.uniform(0.1, 10, p)
        b = rng.normal(0, 1, n)
        w = rng.uniform(0, 2, n) if rng.random() < 0.7 else None
        M, g = ops.weighted_normal_equations(A, b, w)
        Aw = A if w is None else A * w[:, None]
        Mr, gr = Aw.T @ A, Aw.T @ b
        sc = np.sqrt(np.outer(np.diag(Mr), np.diag(Mr))) + 1e-300
        assert np.max(np.abs(M - Mr) / sc) < RTOL_SUMS, (case, n, p)
        assert np.max(np.abs(g - gr)) <= RTOL_SUMS * max(np.max(np.sqrt(np.diag(Mr)) * np.sqrt(np.sum((b if w is None else b * np.sqrt(w)) ** 2))), 1e-300), (case, n, p)
        # N4 post-steps and colour conversion: bit-exact
        H, W = (int(v) for v in rng.integers(1, 70, 2))
        depth = rng.uniform(0.5, 5, (H, W)); var = rng.uniform(0.01, 1, (H, W))
        flag = rng.choice([0, 0, 0, -1, -6, -9], (H, W)).astype(np.int64)
        assert _same(ops.regularize(depth, var, flag), orc.regularize(depth, var, flag)), (case, H, W, "regularize")
        m2, v2 = rng.uniform(0.5, 5, (H, W)), rng.uniform(0.01, 1, (H, W))
        a1, a2 = ops.fusion_arrays(depth, m2, var, v2)
        o1, o2 = orc.fusion_arrays(depth, m2, var, v2)
        assert _same(a1, o1) and _same(a2, o2), (case, H, W, "fusion")
        ch = int(rng.choice([3, 4]))
        rgb8 = rng.integers(0, 256, (H, W, ch), dtype=np.uint8)
        assert _same(ops.rgb2gray(rgb8), orc.rgb2gray(rgb8)), (case, H, W, "rgb2gray u8")
        rgbf = rng.uniform(0, 1, (H, W, ch))
        assert _same(ops.rgb2gray(rgbf), orc.rgb2gray(rgbf)), (case, H, W, "rgb2gray f64")


def test_fuzz_estimate_one_pixel(ops, orc):
    """rust_bindings.semi_dense.estimate_debug_ (tdk_estimate_one): one pixel's (depth, variance, flag), bit for bit."""
    rng = np.random.default_rng(12000 + SEED)
    n = max(4, N_CASES * 4)
    H, W = 60, 80
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    scene = None
    for case in range(n):
        if case % 16 == 0:
            f = rng.uniform(0.7, 1.3) * W
            cam = np.array([f, f, W / 2 + rng.uniform(-1, 1), H / 2 + rng.uniform(-1, 1)])
            a, b = rng.uniform(1.5, 6, 2)
            key_img = 0.5 + 0.3 * np.sin(xx / a) * np.cos(yy / b) + 0.08 * rng.uniform(-1, 1, (H, W))
            ref_img = 0.5 + 0.3 * np.sin((xx + 1.3) / a) * np.cos(yy / b) + 0.08 * rng.uniform(-1, 1, (H, W))
            T_wk = _T(*_random_pose(rng)) if rng.random() < 0.5 else np.eye(4)
            dT = _T(Rotation.from_rotvec(rng.uniform(-0.03, 0.03, 3)).as_matrix(), rng.uniform(-0.2, 0.2, 3))
            pa = (rng.uniform(0.2, 1.0), rng.uniform(4, 12), rng.uniform(0.001, 0.1), rng.uniform(0.001, 0.1),
                  rng.uniform(0.3, 2.0) / f, rng.uniform(0.0, 0.1))
            scene = ((cam, key_img, T_wk), (cam, ref_img, T_wk @ dT), ops.make_params(*pa), orc.make_params(*pa))
        key, ref, pg, po = scene
        u = np.array([rng.integers(-1, W + 1), rng.integers(-1, H + 1)], dtype=np.int64)
        if not (0 <= u[0] < W and 0 <= u[1] < H):
            continue                                    # the reference indexes the image with it
        pd_ = float(rng.choice([rng.uniform(0.3, 8.0), 0.0, -1.0]))
        pv = float(rng.choice([rng.uniform(1e-4, 1.0), 0.0]))
        g = ops.estimate_one(u, pd_, pv, key, ref, pg)
        o = orc.estimate_debug(u, pd_, pv, key, ref, po)
        assert g[2] == o[2] and _same(np.array(g[:2]), np.array(o[:2])), (case, u, pd_, pv, g, o)


# ---------------------------------------------------------------------------
# the drop-in PoseChangeEstimator with its default pyramid (skimage to the bit, level 0 and clip included)
# ---------------------------------------------------------------------------
@pytest.mark.skimage_pyramid
def test_fuzz_dropin_pose_change_estimator(orc):
    import warnings
    import tadataka_amd  # noqa: F401
    import tadataka.vo.dvo as dvo
    from tadataka.camera import CameraModel, CameraParameters
    from tadataka_amd import synthetic
    assert dvo.PYRAMID == "skimage"
    rng = np.random.default_rng(13000 + SEED)
    n = max(2, N_CASES // 6)
    for case in range(n):
        H, W = int(rng.integers(30, 140)), int(rng.integers(40, 180))
        levels = int(rng.integers(1, 6))
        ratio = float(rng.choice([1.5, 1.5, 2.0, 1.3]))
        while min(H, W) / ratio ** (levels - 1) < 10:
            levels -= 1
        pair = synthetic.make_pair(H, W, seed=int(rng.integers(0, 1 << 30)))
        cam = pair["cam"] * rng.uniform(0.97, 1.03, 4) + np.array([0, 0, rng.uniform(-1, 1), rng.uniform(-1, 1)])
        opt = [None, "huber", "student-t", "tukey", "map"][int(rng.integers(0, 5))]
        weights = rng.uniform(0.2, 2.0, (H, W)) if opt == "map" else opt
        D0 = pair["D0"].copy()
        # missing readings -- not at ratio 1.3: its prefilter (sigma 0.15, side weights 2e-10) turns a zero next to a
        # depth of 2 into a depth of 1e-9, a handful of Jacobian rows 1e9 times larger than the rest, cond(J) 3e8: lstsq
        # on J (the reference) still resolves the translation, normal equations in double cannot (docs/HISTORY.md 11b, limits)
        # ... nor with a single level: the identity-scale warp of level 0 turns a zero into a depth of ~1e-13, and at the
        # identity prior that is z itself: cond(J) 2.8e12, where lstsq too is accurate to cond * eps = 6e-4 at best
        if ratio > 1.4 and levels > 1 and rng.random() < 0.25:
            D0[rng.random((H, W)) < 0.05] = 0.0
        cm = CameraModel(CameraParameters(cam[0:2], cam[2:4]), distortion_model=None)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            est = dvo.PoseChangeEstimator(cm, cm, n_coarse_to_fine=levels, max_iter=20)
            est.layer_size_ratio = ratio
            pose = est(pair["I0"], D0, pair["I1"], weights)
            rot, t = orc.dvo_estimate(pair["I0"], D0, pair["I1"], cam, cam, weights, levels, 20, ratio, pyramid="skimage")
        d = max(np.max(np.abs(pose.rotation.as_matrix() - rot.as_matrix())), np.max(np.abs(pose.t - t)))
        assert d < 1e-6, (case, H, W, levels, ratio, opt, d)


# ---------------------------------------------------------------------------
# the drop-in semi-dense loop through rust_bindings.semi_dense, ndarrays or device-resident maps, with the caller
# editing maps between the calls
# ---------------------------------------------------------------------------
def test_fuzz_dropin_semi_dense_loop(ops, orc):
    import tadataka_amd
    from rust_bindings.camera import CameraParameters
    from rust_bindings.semi_dense import Frame, Params, increment_age, propagate, update_depth
    rng = np.random.default_rng(14000 + SEED)
    n_cases = max(2, N_CASES // 4)
    defaults = (1.2, 8.0, 0.02)
    for case in range(n_cases):
        H, W = int(rng.integers(10, 100)), int(rng.integers(10, 130))
        n_frames = int(rng.integers(2, 5))
        f = rng.uniform(0.6, 1.3) * max(H, W)
        cam = np.array([f, f, W / 2 + rng.uniform(-1, 1), H / 2 + rng.uniform(-1, 1)])
        cp = CameraParameters((cam[0], cam[1]), (cam[2], cam[3]))
        pa = (0.3, 12.0, rng.uniform(0.005, 0.05), rng.uniform(0.005, 0.05), rng.uniform(0.5, 1.5) / f, rng.uniform(0.0, 0.05))
        params, po = Params(*pa), orc.make_params(*pa)
        yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
        a, b = rng.uniform(1.5, 6, 2)
        previous = tadataka_amd.enable_device_maps(bool(rng.integers(0, 2)))
        try:
            T = np.eye(4)
            img = 0.5 + 0.3 * np.sin(xx / a) * np.cos(yy / b) + 0.05 * rng.uniform(-1, 1, (H, W))
            frame0 = Frame(cp, img, T)
            refframes = [frame0]
            depth0 = rng.uniform(0.8, 5.0, (H, W)); var0 = rng.uniform(0.01, 0.3, (H, W))
            age0 = np.zeros((H, W), dtype=np.uint64)
            o_depth, o_var, o_age = depth0.copy(), var0.copy(), age0.copy()
            o_frames = [(cam, img, T.copy())]
            for s in range(1, n_frames):
                T10 = _T(Rotation.from_rotvec(rng.uniform(-0.01, 0.01, 3)).as_matrix(), rng.uniform(-0.08, 0.08, 3))
                T = T @ np.linalg.inv(T10)
                img = 0.5 + 0.3 * np.sin((xx + 0.7 * s) / a) * np.cos(yy / b) + 0.05 * rng.uniform(-1, 1, (H, W))
                frame1 = Frame(cp, img, T)
                age1 = increment_age(age0, frame0.camera_params, frame1.camera_params, T10, depth0)
                depth1, var1 = propagate(T10, frame0.camera_params, frame1.camera_params, depth0, var0, *defaults)
                depth1, var1, flag = update_depth(frame1, refframes, age1, depth1, var1, params)
                refframes.append(frame1)
                key = (cam, img, T.copy())
                d, v, a_, f_ = orc.semi_dense_step(key, cam, o_frames, T10, o_age, o_depth, o_var, po, *defaults)
                o_frames.append(key)
                assert np.array_equal(np.asarray(age1), a_) and np.array_equal(np.asarray(flag), f_), (case, H, W, s)
                assert _same(np.asarray(depth1), d) and _same(np.asarray(var1), v), (case, H, W, s)
                o_depth, o_var, o_age = d.copy(), v.copy(), a_.copy()
                # the caller edits a map before handing it on (the example masks by flag): three ways of writing
                edit = rng.integers(0, 4)
                m = rng.random((H, W)) < 0.1
                if edit == 1:
                    depth1[m] = 1.5
                    o_depth[m] = 1.5
                elif edit == 2:
                    np.asarray(var1)[m] = 0.25
                    o_var[m] = 0.25
                elif edit == 3:
                    age1 = np.array(age1, dtype=np.uint64)
                    age1[m] = 0
                    o_age[m] = 0
                depth0, var0, age0, frame0 = depth1, var1, age1, frame1
        finally:
            tadataka_amd.enable_device_maps(previous)


# ---------------------------------------------------------------------------
# bundle adjustment: the Levenberg-Marquardt loop on the device against the reference's loop on dense matrices
# ---------------------------------------------------------------------------
def test_fuzz_ba_lm_loop(ops, orc):
    """tdk_ba_solve against LocalBundleAdjustment.compute / lm_update (reference local_ba.py:91-134) carried out on the
    host with the oracle's Jacobians and dense damped normal equations: the same accepted errors and parameters."""
    rng = np.random.default_rng(15000 + SEED)
    n = max(2, N_CASES // 6)
    total_iters = 0
    for case in range(n):
        P, Q = int(rng.integers(2, 9)), int(rng.integers(10, 60))
        poses = np.column_stack([rng.uniform(-0.2, 0.2, (P, 3)), rng.uniform(-1, 1, (P, 3))])
        points = np.column_stack([rng.uniform(-4, 4, (Q, 2)), rng.uniform(5, 12, Q)])
        vis = rng.random((P, Q)) < rng.choice([1.0, 0.8])
        vis[:, vis.sum(0) < 2] = True
        vis[vis.sum(1) < 4, :] = True
        vp, pt = np.nonzero(vis)
        xt = orc.ba_projection(poses, points, vp, pt, jacobians=False)
        p0 = poses + rng.normal(0, 2e-3, poses.shape)
        q0 = points + rng.normal(0, 1e-2, points.shape)
        max_iter = int(rng.integers(1, 7))
        mu0, nu = float(rng.choice([1.0, 1e-2])), float(rng.choice([100.0, 10.0]))
        kw = dict(absolute_error_threshold=1e-14, relative_error_threshold=1e-9)

        def error(pp, qq):
            x = orc.ba_projection(pp, qq, vp, pt, jacobians=False)
            return float(np.mean(np.sum((xt - x) ** 2, axis=1)))

        def update(pp, qq, mu):
            x, A, B = orc.ba_projection(pp, qq, vp, pt)
            J = np.zeros((2 * len(vp), 6 * P + 3 * Q))
            for k, (j, i) in enumerate(zip(vp, pt)):
                J[2 * k:2 * k + 2, 6 * j:6 * j + 6] = A[k]
                J[2 * k:2 * k + 2, 6 * P + 3 * i:6 * P + 3 * i + 3] = B[k]
            d = np.linalg.solve(J.T @ J + mu * np.eye(J.shape[1]), J.T @ (xt - x).reshape(-1))
            return d[:6 * P].reshape(P, 6), d[6 * P:].reshape(Q, 3)

        pp, qq, mu = p0.copy(), q0.copy(), mu0
        errors = [error(pp, qq)]
        for _ in range(max_iter):
            e0 = errors[-1]
            done = False
            for trial_mu in (mu / nu, mu):
                dp, dq = update(pp, qq, trial_mu)
                e1 = error(pp + dp, qq + dq)
                if e1 < e0:
                    pp, qq, mu, done = pp + dp, qq + dq, trial_mu, True
                    break
            if not done:
                e1, trial_mu = np.inf, mu
                for _guard in range(40):
                    trial_mu *= nu
                    dp, dq = update(pp, qq, trial_mu)
                    e1 = error(pp + dp, qq + dq)
                    if not e1 > e0:
                        break
                pp, qq, mu = pp + dp, qq + dq, trial_mu
            rel = abs((e0 - e1) / e1)
            errors.append(e1)
            if e1 < kw["absolute_error_threshold"] or rel < kw["relative_error_threshold"]:
                break
        ba = ops.BundleAdjustment(P, Q, vp, pt, xt)
        gp, gq, gerr = ba.solve(p0, q0, max_iter=max_iter, initial_mu=mu0, nu=nu, **kw)
        ba.close()
        assert len(gerr) == len(errors), (case, P, Q, len(gerr), len(errors), gerr, errors)
        assert np.allclose(gerr, errors, rtol=1e-6, atol=1e-18), (case, P, Q, gerr, errors)
        # the gauge is free and the damping falls to 1e-10 in a converging two-pose window (cond(H) 8e11 there: parameters
        # 1e-7 apart along the gauge directions): the reprojections are what the two loops must agree on
        x_dev = orc.ba_projection(gp, gq, vp, pt, jacobians=False)
        x_host = orc.ba_projection(pp, qq, vp, pt, jacobians=False)
        assert np.max(np.abs(x_dev - x_host)) < 1e-9, (case, P, Q)
        assert np.allclose(gp, pp, rtol=1e-4, atol=1e-6) and np.allclose(gq, qq, rtol=1e-4, atol=1e-6), (case, P, Q)
        total_iters += len(errors) - 1
    print(f"ba lm loop: {n} windows, {total_iters} accepted iterations compared")


# ---------------------------------------------------------------------------
# the data paths of a batch: host uploads, pinned float64 / 8-bit uploads on the copy stream, partial pyramid rebuilds
# ---------------------------------------------------------------------------
def test_fuzz_batch_data_paths(ops, orc):
    """Random sequences of upload / upload_async (float64 and 8-bit) / build_pyramid(subset) on one batch against a
    shadow copy on the host: every level of every array is what the last build of THAT array made of the frame that was
    current then (levels of an array that was not rebuilt stay as they were), level 0 is always the current frame."""
    rng = np.random.default_rng(16000 + SEED)
    n_cases = max(1, N_CASES // 6)
    names = ("I0", "D0", "I1")
    for case in range(n_cases):
        H, W = int(rng.integers(16, 90)), int(rng.integers(16, 120))
        B = int(rng.choice([1, 3, 9]))
        L = int(rng.integers(1, 4))
        while min(H, W) / 1.5 ** (L - 1) < 6:
            L -= 1
        batch = ops.DvoBatch(B, H, W, n_levels=L)
        frames = [[None] * 3 for _ in range(B)]
        levels = [[[None] * 3 for _ in range(B)] for _ in range(L)]
        for p in range(B):
            I0, D0, I1, _ = _random_scene(rng, H, W)
            batch.upload(p, I0, D0, I1)
            frames[p] = [I0, D0, I1]

        def rebuild(which):
            batch.build_pyramid(None if which is None else [names[k] for k in which])
            for k in (range(3) if which is None else which):
                for p in range(B):
                    for l in range(1, L):
                        levels[l][p][k] = orc.rescale(frames[p][k], 1 / 1.5 ** l, anti_aliasing=True)
        rebuild(None)
        pin64 = ops.PinnedBuffer((B, H, W), np.float64)
        pin8 = ops.PinnedBuffer((B, H, W), np.uint8)
        for step in range(int(rng.integers(3, 9))):
            op = rng.integers(0, 5)
            if op == 0:                                           # a pair replaced from host arrays
                p = int(rng.integers(0, B))
                I0, D0, I1, _ = _random_scene(rng, H, W)
                batch.upload(p, I0, D0, I1)
                frames[p] = [I0, D0, I1]
            elif op == 1:                                         # one array of a range of pairs from pinned float64
                k = int(rng.integers(0, 3)); first = int(rng.integers(0, B)); n = int(rng.integers(1, B - first + 1))
                new = rng.uniform(0.5, 3.0, (n, H, W))
                pin64.array[:n] = new
                batch.upload_async(names[k], first, n, pin64)
                for i in range(n):
                    frames[first + i][k] = new[i].copy()
                ops.call("tdk_sync")                              # the pinned buffer is reused below
            elif op == 2:                                         # 8-bit frames (I0 or I1), converted as img_as_float does
                k = int(rng.choice([0, 2])); first = int(rng.integers(0, B)); n = int(rng.integers(1, B - first + 1))
                new = rng.integers(0, 256, (n, H, W), dtype=np.uint8)
                pin8.array[:n] = new
                batch.upload_async(names[k], first, n, pin8)
                for i in range(n):
                    frames[first + i][k] = new[i] * (1.0 / 255.0)
                ops.call("tdk_sync")
            elif op == 3:
                rebuild(sorted(set(int(v) for v in rng.integers(0, 3, int(rng.integers(1, 3))))))
            else:
                rebuild(None)
            # look at a few (pair, level, array) cells
            for _ in range(4):
                p, l, k = int(rng.integers(0, B)), int(rng.integers(0, L)), int(rng.integers(0, 3))
                want = frames[p][k] if l == 0 else levels[l][p][k]
                assert _same(batch.download(p, l, names[k]), want), (case, step, int(op), p, l, names[k])
        batch.close(); pin64.close(); pin8.close()


# ---------------------------------------------------------------------------
# sessions that outlive their ring of reference frames (saturating ages), several tracks
# ---------------------------------------------------------------------------
def test_fuzz_sd_session_ring(ops, orc):
    rng = np.random.default_rng(17000 + SEED)
    n_cases = max(1, N_CASES // 8)
    defaults = (1.5, 10.0, 0.01)
    for case in range(n_cases):
        H, W = int(rng.integers(12, 90)), int(rng.integers(12, 120))
        n_tracks = int(rng.choice([1, 3]))
        R = int(rng.integers(1, 4))
        n_steps = R + int(rng.integers(1, 4))
        f = rng.uniform(0.6, 1.3) * max(H, W)
        cam = np.array([f, f, W / 2, H / 2])
        pa = (0.3, 12.0, rng.uniform(0.005, 0.05), rng.uniform(0.005, 0.05), rng.uniform(0.5, 1.5) / f, rng.uniform(0.0, 0.05))
        pg, po = ops.make_params(*pa), orc.make_params(*pa)
        sd = ops.SemiDenseSession(n_tracks, H, W, max_refframes=R)
        sd.set_age_policy(True)
        sd.set_params(pg, *defaults)
        yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
        state = []
        for t in range(n_tracks):
            a, b = rng.uniform(1.5, 6, 2)
            frames, T = [], np.eye(4)
            for s in range(n_steps + 1):
                img = 0.5 + 0.3 * np.sin((xx + 0.7 * s) / a) * np.cos(yy / b) + 0.05 * rng.uniform(-1, 1, (H, W))
                frames.append((cam, img, T.copy()))
                T = T @ _T(Rotation.from_rotvec(rng.uniform(-0.01, 0.01, 3)).as_matrix(), rng.uniform(-0.05, 0.05, 3))
            depth = rng.uniform(0.8, 5.0, (H, W)); var = rng.uniform(0.01, 0.3, (H, W))
            age = np.zeros((H, W), dtype=np.uint64)
            sd.push_frame(t, *frames[0])
            sd.set_maps(t, depth, var, age)
            state.append(dict(frames=frames, depth=depth, var=var, age=age))
        for step in range(1, n_steps + 1):
            T10s, Twfs = [], []
            for t in range(n_tracks):
                fr = state[t]["frames"]
                sd.push_frame(t, fr[step][0], fr[step][1])
                T10s.append(np.linalg.inv(fr[step][2]) @ fr[step - 1][2])
                Twfs.append(fr[step][2])
            sd.step(np.array(T10s), np.array(Twfs), commit=True)
            n_ref = min(step, R)
            for t in range(n_tracks):
                st = state[t]
                fr = st["frames"]
                a1 = np.minimum(orc.increment_age(st["age"], cam, cam, T10s[t], st["depth"]), np.uint64(n_ref))
                d1, v1 = orc.propagate(T10s[t], cam, cam, st["depth"], st["var"], *defaults)
                d, v, f_ = orc.update_depth(fr[step], fr[step - n_ref:step], a1, d1, v1, po)
                gd, gv, ga, gf = sd.get_maps(t, with_flag=True)
                assert np.array_equal(ga, a1) and np.array_equal(gf, f_), (case, H, W, R, t, step)
                assert _same(gd, d) and _same(gv, v), (case, H, W, R, t, step)
                st.update(depth=d, var=v, age=a1)
        sd.close()


# ---------------------------------------------------------------------------
# the drop-in warp and metric classes (tadataka.warp, tadataka.metric)
# ---------------------------------------------------------------------------
def test_fuzz_dropin_warp_and_metric(orc):
    import warnings
    import tadataka_amd  # noqa: F401
    from tadataka.camera import CameraModel, CameraParameters
    from tadataka.metric import PhotometricError, photometric_error
    from tadataka.pose import Pose
    from tadataka.warp import LocalWarp2D, Warp2D
    rng = np.random.default_rng(18000 + SEED)
    for case in range(max(2, N_CASES // 2)):
        H, W = (int(v) for v in rng.integers(4, 90, 2))
        I0, D0, I1, cam0 = _random_scene(rng, H, W)
        D0 = np.where(np.isfinite(D0) & (D0 > 1e-3) & (D0 < 1e3), D0, 2.0)      # the array-level warp is not about holes
        cam1 = cam0 * rng.uniform(0.9, 1.1, 4) if rng.random() < 0.5 else cam0
        cm0 = CameraModel(CameraParameters(cam0[0:2], cam0[2:4]), distortion_model=None)
        cm1 = CameraModel(CameraParameters(cam1[0:2], cam1[2:4]), distortion_model=None)
        R, t = _random_pose(rng)
        if np.max(np.abs(t)) > 1.5:
            t = t / 3
        pose10 = Pose(Rotation.from_matrix(R), t)
        T10 = np.array(pose10.T)            # (the Rotation object re-orthonormalises R: its matrix is what both sides get)
        n = int(rng.integers(1, 400))
        us = np.column_stack([rng.uniform(0, W - 1, n), rng.uniform(0, H - 1, n)])
        ds = rng.uniform(0.5, 6.0, n)
        # LocalWarp2D = unnormalize(warp_vecs(T10, normalize(us), d)): bit for bit through the granular operators
        us1, d1 = LocalWarp2D(cm0, cm1, pose10)(us, ds)
        xs1, od1 = orc.warp_vecs(T10, orc.normalize(us, cam0), ds)
        assert _same(us1, orc.unnormalize(xs1, cam1)) and _same(d1, od1), (case, "LocalWarp2D")
        # Warp2D between world poses: the same points through two transforms (rounding apart)
        Rw, tw = _random_pose(rng)
        pose_w0 = Pose(Rotation.from_matrix(Rw), tw)
        pose_w1 = pose_w0 * pose10.inv() if hasattr(pose10, "inv") else None
        if pose_w1 is not None:
            us1w, d1w = Warp2D(cm0, cm1, pose_w0, pose_w1)(us, ds)
            ok = np.isfinite(us1).all(axis=1) & (np.abs(d1) > 1e-6)
            assert np.allclose(us1w[ok], us1[ok], rtol=1e-9, atol=1e-7 * max(H, W)) and np.allclose(d1w[ok], d1[ok], rtol=1e-9, atol=1e-9), (case, "Warp2D")
        # photometric error: function and class
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = orc.photometric_error(I0, D0, I1, cam0, cam1, T10)
        got_f = photometric_error(LocalWarp2D(cm0, cm1, pose10), I0, D0, I1)
        got_c = PhotometricError(cm0, cm1, I0, D0, I1)(pose10)
        for got in (got_f, got_c):
            assert (np.isnan(got) and np.isnan(want)) or abs(got - want) <= 1e-9 * abs(want), (case, got, want)
        # (not at the identity: through two world poses the relative transform is the identity up to 1e-17, and the
        #  border pixels on the inclusive mask edge change sides -- in the reference's own Warp2D just as well)
        if pose_w1 is not None and not np.array_equal(T10, np.eye(4)):
            got_w = photometric_error(Warp2D(cm0, cm1, pose_w0, pose_w1), I0, D0, I1)
            if np.isfinite(want):
                assert abs(got_w - want) <= 1e-6 * abs(want) + 1e-12, (case, got_w, want)


# ---------------------------------------------------------------------------
# the array-level calc_pose_update of the drop-in tadataka.vo.dvo (points already in frame 1)
# ---------------------------------------------------------------------------
def test_fuzz_dropin_calc_pose_update(orc):
    """vo/dvo/__init__.py:46-70 restated with the oracle's operators (mask, bilinear samples of the gradient maps,
    jacobian.py:8-24, lstsq on sqrt(w) J) against the device reduction + 6 x 6 solve of the drop-in function."""
    import warnings
    import tadataka_amd  # noqa: F401
    from tadataka.camera import CameraModel, CameraParameters
    from tadataka.vo.dvo import calc_pose_update
    rng = np.random.default_rng(19000 + SEED)
    for case in range(max(2, N_CASES // 2)):
        H, W = (int(v) for v in rng.integers(6, 80, 2))
        f = rng.uniform(0.6, 1.5) * max(H, W)
        cam = np.array([f, f * rng.uniform(0.95, 1.05), W / 2 + rng.uniform(-1, 1), H / 2 + rng.uniform(-1, 1)])
        cm = CameraModel(CameraParameters(cam[0:2], cam[2:4]), distortion_model=None)
        n = int(rng.integers(30, 3000))
        z = rng.uniform(0.8, 5.0, n)
        P1 = np.column_stack([rng.uniform(-0.7, 0.7, n) * z * W / f, rng.uniform(-0.7, 0.7, n) * z * H / f, z])
        P1[rng.random(n) < 0.05, 2] *= -1.0                        # behind the camera
        GX, GY = rng.normal(0, 0.1, (H, W)), rng.normal(0, 0.1, (H, W))
        r = rng.normal(0, 0.05, n)
        if rng.random() < 0.3:
            r[rng.random(n) < 0.1] += 3.0
        opt = [None, "huber", "student-t", "tukey", "map"][int(rng.integers(0, 5))]
        wmap = rng.uniform(0.1, 2.0, n)
        us1 = orc.unnormalize(orc.project_vecs(P1), cam)
        mask = orc.is_in_image_range(us1, (H, W)) & (P1[:, 2] > 0)
        got = calc_pose_update(cm, r, GX, GY, P1, wmap if opt == "map" else opt)
        if not mask.any():
            assert got is None, case
            continue
        gx, gy = orc.interpolation(GX, us1[mask]), orc.interpolation(GY, us1[mask])
        x, y, zz = P1[mask, 0], P1[mask, 1], P1[mask, 2]
        fgx, fgy = cam[0] * gx, cam[1] * gy
        z2, xy = zz * zz, x * y
        J = np.column_stack((fgx / zz, fgy / zz, -(fgx * x + fgy * y) / z2, -(fgx * xy + fgy * (z2 + y * y)) / z2,
                             (fgx * (z2 + x * x) + fgy * xy) / z2, (-fgx * y + fgy * x) / zz))
        rm = r[mask]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            w = None if opt is None else wmap[mask] if opt == "map" else _weights_numpy(rm, opt)
        if w is not None and not np.all(np.isfinite(w)):
            continue
        want = orc.solve_lstsq(J, rm, w)
        s = np.linalg.svd(J if w is None else J * np.sqrt(w)[:, None], compute_uv=False)
        if s[-1] < 1e-5 * s[0] or mask.sum() < 12:
            continue                                              # a rank-deficient draw: other rules apply
        assert got is not None and np.allclose(got, want, rtol=1e-7, atol=1e-9 * np.max(np.abs(want))), (case, opt, got, want)


# ---------------------------------------------------------------------------
# the device loop's state machine: heterogeneous batches (pairs that stop at different evaluations, empty masks)
# ---------------------------------------------------------------------------
def test_fuzz_dvo_estimate_heterogeneous_batches(ops, orc):
    """Pairs of one batch finish their levels at different evaluations, some never start (no valid depth), some diverge
    at once (unrelated second frame): every pair's pose against the oracle's loop run on that pair alone."""
    import warnings
    from tadataka_amd import synthetic
    rng = np.random.default_rng(20000 + SEED)
    n = max(1, N_CASES // 12)
    for case in range(n):
        H, W = int(rng.integers(24, 70)), int(rng.integers(32, 90))
        B = int(rng.choice([2, 5, 9, 20]))
        levels = int(rng.integers(1, 4))
        wname = [None, "huber", "student-t", "tukey"][int(rng.integers(0, 4))]
        max_iter = int(rng.choice([20, 20, 3]))
        batch = ops.DvoBatch(B, H, W, n_levels=levels)
        pairs, cams, kinds = [], [], []
        for p in range(B):
            pair = synthetic.make_pair(H, W, seed=int(rng.integers(0, 1 << 30)))
            kind = int(rng.choice([0, 0, 0, 1, 2, 3]))
            I0, D0, I1 = pair["I0"], pair["D0"].copy(), pair["I1"]
            if kind == 1:
                D0[:] = np.nan                                    # nothing to warp: the reference warns and returns the prior
            elif kind == 2:
                I1 = rng.uniform(0, 1, (H, W))                    # unrelated frame: the first candidate is rejected
            elif kind == 3:
                I1 = I0.copy()                                    # identical frames
            cam = pair["cam"] * rng.uniform(0.98, 1.02, 4) + np.array([0, 0, rng.uniform(-1, 1), rng.uniform(-1, 1)])
            batch.upload(p, I0, D0, I1)
            pairs.append((I0, D0, I1)); cams.append(cam); kinds.append(kind)
        batch.build_pyramid()
        cams = np.array(cams)
        prior = np.tile(_pose12(np.eye(3), np.zeros(3)), (B, 1))
        P, _ = batch.estimate(cams, cams, prior, ops.WEIGHT_MODES[wname], max_iter)
        warned = batch.warnings()
        for p in range(B):
            I0, D0, I1 = pairs[p]
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                try:
                    Rr, tr = orc.dvo_estimate(I0, D0, I1, cams[p], cams[p], wname, n_coarse_to_fine=levels,
                                              max_iter=max_iter, anti_aliasing=True)[:2]
                except np.linalg.LinAlgError:
                    continue        # identical frames under Tukey / Student-t: 0 / 0 weights, the reference's lstsq raises
            d = max(np.max(np.abs(P[p, :9].reshape(3, 3) - Rr.as_matrix())), np.max(np.abs(P[p, 9:] - tr)))
            assert d < 1e-6, (case, p, kinds[p], H, W, B, levels, wname, max_iter, d)
            if kinds[p] == 1:
                assert warned[p], (case, p)
        batch.close()
