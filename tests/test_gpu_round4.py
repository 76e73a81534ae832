"""GPU parity tests added in round 4.

* The reference's own semi-dense integration test (tests/vo/semi_dense/test_semi_dense.py:41-135)
  replayed through the drop-in `rust_bindings.semi_dense` API on its own New-Tsukuba stereo pair
  (fixture tests/golden/semi_dense_tsukuba.npz, built by tests/golden/generate_golden_r4.py from
  the reference's loader).

Bars: flags / depth / variance bit-exact against the oracle; the five reference-authored flags as
the reference's test asserts them."""
import hashlib
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))

import tadataka_amd  # noqa: F401,E402   (puts the drop-in packages on sys.path)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from tadataka_amd import _lib, ops as o
    _lib.require_gpu()
    return o


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def _sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


# ---------------------------------------------------------------------------
# tests/vo/semi_dense/test_semi_dense.py on dataset[0] of the New-Tsukuba sample
# ---------------------------------------------------------------------------
@pytest.fixture(scope="module")
def tsukuba(golden):
    import scenes
    g = golden("semi_dense_tsukuba.npz")
    return g, scenes.gray_from_rgb_u8(g["rgb_L"]), scenes.gray_from_rgb_u8(g["rgb_R"])


def _frames(g, key_image, ref_image):
    """The set-up of test_update_depth / test_estimate (:54-59, :89-94)."""
    from rust_bindings.camera import CameraParameters
    from rust_bindings.semi_dense import Frame
    fx, fy, ox, oy = g["cam"]
    key_camera_params = CameraParameters((fx, fy), (ox, oy))
    ref_camera_params = CameraParameters((fx, fy), (ox, oy))
    keyframe = Frame(key_camera_params, key_image, g["T_wk"])
    refframe = Frame(ref_camera_params, ref_image, g["T_wr"])
    return keyframe, refframe


def test_reference_test_estimate(ops, orc, tsukuba):
    """test_estimate (:76-135): the five assertions whose inputs are in the checkout, with the
    reference's literal priors and its own FLAG names."""
    from rust_bindings.semi_dense import Params, estimate_debug_
    from tadataka.vo.semi_dense.flag import ResultFlag as FLAG
    g, key_image, ref_image = tsukuba
    keyframe, refframe = _frames(g, key_image, ref_image)
    params = Params(min_depth=0.1, max_depth=1000.0, geo_coeff=0.01, photo_coeff=0.01,
                    ref_step_size=0.01, min_gradient=0.2)
    assert np.array_equal(g["est_params"], [0.1, 1000.0, 0.01, 0.01, 0.01, 0.2])

    def estimate(u_key, prior_depth, prior_variance):
        return estimate_debug_(u_key, prior_depth, prior_variance, keyframe, refframe, params)

    depth, variance, flag = estimate(np.array([110, 400]), -10.0, 10.0)
    assert flag == FLAG.NEGATIVE_PRIOR_DEPTH
    depth, variance, flag = estimate(np.array([110, 400]), 0.05, 0.2)
    assert flag == FLAG.HYPOTHESIS_OUT_OF_SERCH_RANGE
    depth, variance, flag = estimate(np.array([390, 100]), 2.0, 0.2)
    assert flag == FLAG.INSUFFICIENT_GRADIENT
    depth, variance, flag = estimate(np.array([0, 200]), 2.0, 0.2)        # u_key is on the image edge
    assert flag == FLAG.KEY_OUT_OF_RANGE
    depth, variance, flag = estimate(np.array([116, 400]), 2.0, 0.001)    # very short search range
    assert flag == FLAG.REF_EPIPOLAR_TOO_SHORT

    # the same rows as stored by the generator, against the oracle, value for value
    po = orc.make_params(*g["est_params"])
    key, ref = (g["cam"], key_image, g["T_wk"]), (g["cam"], ref_image, g["T_wr"])
    for ux, uy, pd_, pv_, expected in g["est_cases"]:
        u = np.array([int(ux), int(uy)])
        got = estimate(u, pd_, pv_)
        assert got[2] == int(expected)
        assert got == orc.estimate_debug(u, pd_, pv_, key, ref, po)


def test_reference_test_update_depth(ops, orc, tsukuba):
    """test_update_depth (:41-73): update_depth over the whole real frame -- ages 1, depth 200 (cm),
    variance 1, Params(60, 1000, ...) -- bit-exact against the oracle, histogram and digests frozen."""
    from rust_bindings.semi_dense import Params, update_depth
    g, key_image, ref_image = tsukuba
    keyframe, refframe = _frames(g, key_image, ref_image)
    params = Params(min_depth=60.0, max_depth=1000.0, geo_coeff=0.01, photo_coeff=0.01,
                    ref_step_size=0.01, min_gradient=0.2)
    shape = key_image.shape
    age_map = np.ones(shape, dtype=np.uint64)
    prior_depth = 200.0 * np.ones(shape, dtype=np.float64)
    prior_variance = np.ones(shape, dtype=np.float64)
    depth, variance, flag = (np.asarray(m) for m in update_depth(keyframe, [refframe, ], age_map, prior_depth,
                                                                 prior_variance, params))
    key, ref = (g["cam"], key_image, g["T_wk"]), (g["cam"], ref_image, g["T_wr"])
    od, ov, of = orc.update_depth(key, [ref], age_map, prior_depth, prior_variance, orc.make_params(*g["upd_params"]))
    assert np.array_equal(flag, of)
    assert np.array_equal(depth, od) and np.array_equal(variance, ov)
    hist = np.array([(flag == -b).sum() for b in range(10)])
    assert np.array_equal(hist, g["upd_flag_histogram"]) and hist[0] == 32595
    for name, arr in (("upd_sha_depth", depth), ("upd_sha_var", variance), ("upd_sha_flag", flag)):
        assert np.array_equal(_sha(arr), g[name]), name
    # the host-pointer entry (tdk_update_depth) and the single-pixel entry agree with the map path
    d2, v2, f2 = ops.update_depth(key, [ref], age_map, prior_depth, prior_variance, ops.make_params(*g["upd_params"]))
    assert np.array_equal(f2, flag) and np.array_equal(d2, depth) and np.array_equal(v2, variance)
    pg = ops.make_params(*g["upd_params"])
    ys, xs = np.nonzero(flag == 0)
    for k in range(0, len(ys), max(1, len(ys) // 25)):
        one = ops.estimate_one([int(xs[k]), int(ys[k])], 200.0, 1.0, key, ref, pg)
        assert one == (depth[ys[k], xs[k]], variance[ys[k], xs[k]], 0)


# ---------------------------------------------------------------------------
# DeviceMap: host and device copy cannot diverge (advisor finding / review item 6)
# ---------------------------------------------------------------------------
def _edit_cases():
    def via_asarray(m, mask, x):
        np.asarray(m)[mask] = x

    def via_slice(m, mask, x):
        m[10:20][:] = x

    def via_fill(m, mask, x):
        m.fill(x)

    def via_copyto(m, mask, x):
        np.copyto(m, np.where(mask, x, np.asarray(m).copy()))

    def via_putmask(m, mask, x):
        np.putmask(m, mask, x)

    def via_transpose(m, mask, x):
        m.T[:, 10:20] = x                   # rows 10:20 of the map through a transposed view

    def via_rows(m, mask, x):
        for k, row in enumerate(m):
            if 10 <= k < 20:
                row[:] = x

    def via_flat(m, mask, x):
        m.flat[10 * m.shape[1]:20 * m.shape[1]] = x

    def via_ufunc_out(m, mask, x):
        np.multiply(m, 0.0, out=m)
        np.add(m, x, out=m)

    def via_setitem(m, mask, x):
        m[mask] = x
    return [via_asarray, via_slice, via_fill, via_copyto, via_putmask, via_transpose, via_rows, via_flat,
            via_ufunc_out, via_setitem]


@pytest.mark.parametrize("edit", _edit_cases(), ids=lambda f: f.__name__)
def test_device_map_edits_reach_update_depth(ops, orc, edit, device_maps):
    """Everything that is legal on the ndarray the reference returns -- np.asarray(m)[mask] = x,
    m[a:b][:] = x, m.fill(x), np.copyto(m, ...), views, iteration -- followed by handing the map back
    into update_depth: the result is the oracle's on the edited map."""
    from rust_bindings.camera import CameraParameters
    from rust_bindings.semi_dense import Frame, Params, propagate, update_depth
    from tadataka_amd import synthetic
    H, W = 96, 128
    c = synthetic.make_semi_dense_case(H, W, seed=21, valid_fraction=0.6)
    cam = c["cam"]
    cp = CameraParameters((cam[0], cam[1]), (cam[2], cam[3]))
    pargs = (0.5, 10.0, 0.01, 0.01, 0.004, 0.01)
    keyframe, refframe = Frame(cp, c["key_image"], c["T_wk"]), Frame(cp, c["ref_image"], c["T_wr"])
    # a depth map as the loop has it: returned by a previous call, resident on the device
    T10 = np.eye(4)
    depth_map, variance_map = propagate(T10, cp, cp, c["prior_depth"], c["prior_variance"], 1.0, 10.0, 0.0)
    assert isinstance(depth_map, ops.DeviceMap)
    twin = orc.propagate(T10, cam, cam, c["prior_depth"], c["prior_variance"], 1.0, 10.0, 0.0)[0]
    assert np.array_equal(depth_map, twin)
    rng = np.random.default_rng(5)
    mask = rng.uniform(0, 1, (H, W)) < 0.3
    edit(depth_map, mask, 2.5)
    edit(twin, mask, 2.5)
    assert not np.array_equal(twin, c["prior_depth"])
    d, v, f = update_depth(keyframe, [refframe], c["age"], depth_map, variance_map, Params(*pargs))
    key, ref = (cam, c["key_image"], c["T_wk"]), (cam, c["ref_image"], c["T_wr"])
    od, ov, of = orc.update_depth(key, [ref], c["age"], twin, np.asarray(variance_map), orc.make_params(*pargs))
    assert np.array_equal(f, of) and np.array_equal(d, od) and np.array_equal(v, ov)


def test_device_map_escape_tracking(ops, orc, monkeypatch, device_maps):
    """Once a writable reference to the host copy has left the map it is `escaped` for good: every device use
    re-sends the host copy, so later writes through the reference are seen (no reference counting is consulted:
    whether the reference is still alive is not knowable portably); read-only looks (np.array_equal, m.max(),
    arithmetic) never cost an upload; copies are maps of their own; the caller's ndarray is not aliased."""
    import copy
    from rust_bindings.camera import CameraParameters
    from rust_bindings.semi_dense import increment_age
    from tadataka_amd import synthetic
    H, W = 40, 56
    c = synthetic.make_semi_dense_case(H, W, seed=12)
    cam = c["cam"]
    cp = CameraParameters((cam[0], cam[1]), (cam[2], cam[3]))
    T10 = np.linalg.inv(c["T_wk"]) @ c["T_wr"]
    uploads = []
    real_call = ops.call

    def counting_call(name, *args):
        if name == "tdk_map_upload":
            uploads.append(name)
        return real_call(name, *args)
    monkeypatch.setattr(ops, "call", counting_call)

    depth = ops.DeviceMap.of(c["prior_depth"], np.float64)
    a1 = increment_age(c["age"], cp, cp, T10, depth)
    twin = orc.increment_age(c["age"], cam, cam, T10, c["prior_depth"])
    # read-only looks: no upload afterwards
    assert np.array_equal(a1, twin) and int(a1.max()) == int(twin.max()) and np.array_equal(a1 + 1, twin + 1)
    increment_age(a1, cp, cp, T10, depth)
    assert uploads == []
    # a live reference: writes through it after a device use are still seen
    held = np.asarray(a1)
    held[0, :] = 7; twin[0, :] = 7
    assert np.array_equal(increment_age(a1, cp, cp, T10, depth), orc.increment_age(twin, cam, cam, T10, c["prior_depth"]))
    held[1, :] = 9; twin[1, :] = 9
    assert np.array_equal(increment_age(a1, cp, cp, T10, depth), orc.increment_age(twin, cam, cam, T10, c["prior_depth"]))
    assert len(uploads) == 2
    del held
    increment_age(a1, cp, cp, T10, depth)            # the map stays escaped: one upload per device use
    assert len(uploads) == 3
    import inspect
    assert "getrefcount" not in inspect.getsource(ops.DeviceMap)
    # copies own their buffer; destroying one leaves the other intact
    b = copy.copy(a1); d = copy.deepcopy(a1)
    assert b._h.value != a1._h.value and d._h.value != a1._h.value
    b[2, :] = 1
    del b
    assert np.array_equal(a1, twin) and np.array_equal(d, twin)
    # the array a map was made from stays the caller's
    src = c["prior_depth"].copy()
    m = ops.DeviceMap.of(src, np.float64)
    src[:] = 0.0
    assert np.array_equal(m, c["prior_depth"])
    # a frame's image handed out as a map: editing the map does not touch the frame
    from rust_bindings.semi_dense import Frame
    frame = Frame(cp, c["key_image"], c["T_wk"])
    img = frame.image
    np.asarray(img)[:] = 0.0
    assert float(np.asarray(img).max()) == 0.0 and np.array_equal(frame.image, c["key_image"])


# ---------------------------------------------------------------------------
# the row-streaming pyramid kernel (k_pyramid_stream): bit-identical with the tiles and the oracle
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("shape,levels", [((480, 640), 3), ((480, 640), 2), ((120, 160), 3), ((97, 131), 3),
                                          ((720, 1280), 3), ((250, 249), 3), ((251, 497), 5)])
def test_streaming_pyramid_is_bit_identical(ops, orc, monkeypatch, shape, levels):
    """tdk_set_option(TDK_OPT_PYRAMID_STREAM, 2) forces the streaming kernel for the first one / two levels of any
    batch (by default it takes batches of >= 256 strips); 0 keeps the tiled kernel.  Every level of every
    array must come out bit for bit the same, and equal to the oracle's anti-aliased rescale: one
    strip (W < 249), three strips (VGA), strips that do not divide the width, odd heights, a frame
    whose last chunk is partial, deeper pyramids whose later levels stay on the tiles."""
    from tadataka_amd import synthetic
    H, W = shape
    B = 2
    rng = np.random.default_rng(11)
    pairs = []
    for i in range(B):
        pr = synthetic.make_pair(H, W, seed=60 + i)
        pr["W0"] = rng.uniform(0.05, 50.0, (H, W))
        pairs.append(pr)
    got = {}
    for mode in ("0", "2"):
        ops.set_option("pyramid_stream", int(mode))
        batch = ops.DvoBatch(B, H, W, n_levels=levels, ratio=1.5, with_weight_map=True)
        batch.set_anti_aliasing(True)
        for i, pr in enumerate(pairs):
            batch.upload(i, pr["I0"], pr["D0"], pr["I1"], pr["W0"])
        batch.build_pyramid()
        got[mode] = {(l, i, n): batch.download(i, l, n) for l in range(1, levels) for i in range(B)
                     for n in ("I0", "D0", "I1", "W0")}
        batch.close()
    for key, tiles in got["0"].items():
        assert np.array_equal(got["2"][key], tiles), (key, float(np.max(np.abs(got["2"][key] - tiles))))
    for (l, i, n) in ((1, 0, "D0"), (1, 1, "W0"), (2, 0, "I1"), (2, 1, "D0")):
        if l < levels:
            want = orc.rescale(pairs[i][n], 1 / 1.5 ** l, anti_aliasing=True)
            assert np.array_equal(got["2"][(l, i, n)], want), (l, i, n)


# ---------------------------------------------------------------------------
# multi-GPU path, as far as one GPU goes
# ---------------------------------------------------------------------------
def test_pose_gather_alternating_batches_with_host_collectives(ops):
    """The ordering code of csrc/comm.hip that a real multi-GPU bench run takes, with a 1-rank RCCL
    communicator: device-resident pose gathers queued on the streams of TWO batches in turn (bench.py
    keeps two in flight), the gather of step k collected after step k + 1's estimation, and host-buffer
    collectives (the bench's MAX / SUM reductions and its barrier) on the communicator's own stream in
    between -- every gather must return the poses of the batch it was started on."""
    from tadataka_amd import sharding, synthetic
    comm = sharding.RcclComm(0, 1, sharding.RcclComm.unique_id())
    B, H, W = 4, 60, 80
    cam = synthetic.camera_for(W, H)
    ident = np.tile(ops.pose12(np.eye(3), np.zeros(3)), (B, 1))
    batches = []
    for k in range(2):
        bt = ops.DvoBatch(B, H, W, n_levels=2, ratio=1.5)
        for i in range(B):
            p = synthetic.make_pair(H, W, seed=200 + 10 * k + i)
            bt.upload(i, p["I0"], p["D0"], p["I1"])
        bt.build_pyramid()
        batches.append(bt)
    gather = sharding.PoseGather(B, comm)
    gather.comm = comm                     # world 1 would take the local shortcut: force the RCCL path
    expected, collected = [], []
    for step in range(6):
        bt = batches[step % 2]
        poses, _ = bt.estimate(cam, cam, ident, ops.W_HUBER, 20)
        if gather.pending:
            collected.append(gather.finish())
        assert np.array_equal(comm.all_reduce([float(step), 2.0], "max"), [float(step), 2.0])
        gather.start(poses, bt)
        expected.append(poses.copy())
        comm.barrier()                     # a host-buffer collective while the gather is in flight
        assert np.array_equal(comm.all_reduce([1.0], "sum"), [1.0])
    collected.append(gather.finish())
    assert len(collected) == len(expected) == 6
    for got, want in zip(collected, expected):
        assert got.shape == (B, 12) and np.array_equal(got, want)
    assert not np.array_equal(expected[0], expected[1])          # the two batches hold different pairs
    with pytest.raises(Exception):
        comm.gather_poses_finish()                               # nothing in flight
    for bt in batches:
        bt.close()
    comm.close()


def test_bench_dry_ranks_runs_the_multi_rank_bookkeeping():
    """bench.py --dry-ranks 3: three worker processes share the GPU and exchange through files; the
    script's own step() / timed_block() / PoseGather run with world 3 (sharded seeds, two batches in
    flight, gathered poses checked against the truth of every rank inside bench.py).  Also the
    cfg4 workload definition (1280x720, 1 level, 2 iterations) at a reduced pair count."""
    import json
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TDK_RENDEZVOUS_KEY="pytest_dry_%d" % os.getpid(), TDK_FILECOMM_TIMEOUT="60")
    p = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--dry-ranks", "3", "--pairs", "6", "--steps", "3",
                        "--warmup", "1", "--min-seconds", "0", "--no-cpu-baseline", "--no-workloads", "--height", "120",
                        "--width", "160"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=180)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 3 and out["rccl_ranks"] == 0 and "dry run" in out["exchange"]
    assert out["config"]["pairs_per_gpu"] == 6 and out["max_translation_error"] < 5e-3
    assert abs(out["frame_pairs_per_s"] - 3 * 6 * out["steps"] * out["timed_blocks"] / out["timed_seconds"]) < 1e-6
    p = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--config", "cfg4", "--pairs", "4", "--steps", "2",
                        "--warmup", "1", "--min-seconds", "0", "--no-cpu-baseline", "--no-workloads"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=180)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["config"]["height"] == 720 and out["config"]["levels"] == 1 and out["config"]["max_iter"] == 2
    assert out["config"]["name"].startswith("cfg4")


# ---------------------------------------------------------------------------
# the forward warp as a gather (k_sd_targets / k_sd_gather) against the slot path and the oracle
# ---------------------------------------------------------------------------
def _warp_case(H, W, seed, kind):
    """Maps and a transform: 'stereo' = SURVEY cfg3's x-baseline (disparity spread ~20 px, no y motion),
    'motion' = a small general motion (box of a few pixels in both axes), 'zoom' = a strong zoom-out that
    folds many sources onto every target (box far beyond the gather's window: the slot path)."""
    from tadataka_amd import synthetic
    c = synthetic.make_semi_dense_case(H, W, seed=seed, valid_fraction=0.5)
    rng = np.random.default_rng(seed)
    T10 = np.eye(4)
    if kind == "stereo":
        T10 = np.linalg.inv(c["T_wr"]) @ c["T_wk"]
    elif kind == "motion":
        T10[:3, :3] = synthetic.rodrigues(rng.uniform(-0.01, 0.01, 3))
        T10[:3, 3] = rng.uniform(-0.03, 0.03, 3)
    else:
        T10[2, 3] = 4.0
    age0 = rng.integers(0, 4, (H, W)).astype(np.uint64)
    var0 = rng.uniform(0.01, 0.2, (H, W))
    return c, T10, age0, var0


@pytest.mark.parametrize("kind", ["stereo", "motion", "zoom"])
@pytest.mark.parametrize("shape", [(480, 640), (97, 131), (60, 64)])
def test_forward_warp_gather_equals_slot_path_and_oracle(ops, orc, monkeypatch, kind, shape):
    """increment_age and propagate: the gather path (default), the slot path (TDK_OPT_SD_WARP_GATHER = 0) and the
    oracle give the same bits -- for a stereo baseline, a small general motion and a zoom-out whose
    displacement box exceeds the gather's window (there the default path IS the slot path: it falls back
    on the device); frame sizes that are not multiples of the 64 x 4 tiles included."""
    H, W = shape
    c, T10, age0, var0 = _warp_case(H, W, 31, kind)
    cam = c["cam"]
    want_age = orc.increment_age(age0, cam, cam, T10, c["prior_depth"])
    want_d, want_v = orc.propagate(T10, cam, cam, c["prior_depth"], var0, 1.0, 10.0, 0.01)
    # 1 = k_sd_targets + k_sd_gather2 (default), 0 = slots
    for mode in ("1", "0"):
        ops.set_option("sd_warp_gather", int(mode))
        got_age = ops.increment_age(age0, cam, cam, T10, c["prior_depth"])
        got_d, got_v = ops.propagate(T10, cam, cam, c["prior_depth"], var0, 1.0, 10.0, 0.01)
        assert np.array_equal(got_age, want_age), (mode, kind)
        assert np.array_equal(got_d, want_d) and np.array_equal(got_v, want_v), (mode, kind)
    assert int((want_age > 0).sum()) > 0


def test_forward_warp_more_than_four_sources_inside_the_window(ops, orc, monkeypatch):
    """A zoom-out by 0.47 on an 8 x 32 frame: the displacement box still fits every gather kernel's window,
    but most targets collect five or more sources -- the plain-scan branch of the gathers (more sources than
    slots) against the oracle.  The test counts the sources per target itself to be sure the branch runs."""
    from tadataka_amd import synthetic
    H, W = 8, 32
    c = synthetic.make_semi_dense_case(H, W, seed=77, valid_fraction=0.5)
    rng = np.random.default_rng(77)
    cam = c["cam"]
    d0 = np.full((H, W), 2.0) * rng.uniform(0.98, 1.02, (H, W))
    T10 = np.eye(4)
    T10[2, 3] = 2.0 * (1.0 / 0.47 - 1.0)
    fx, fy, ox, oy = [float(v) for v in np.asarray(cam).ravel()[:4]]
    ys, xs = np.mgrid[0:H, 0:W].astype(float)
    z1 = d0 + T10[2, 3]
    u = ((xs - ox) / fx * d0) / z1 * fx + ox
    v = ((ys - oy) / fy * d0) / z1 * fy + oy
    ok = (u >= 0) & (u <= W - 1) & (v >= 0) & (v <= H - 1)
    tg = (v[ok].astype(int) * W + u[ok].astype(int))
    assert np.bincount(tg).max() >= 5, "the case no longer folds more than four sources onto a target"
    age0 = rng.integers(0, 4, (H, W)).astype(np.uint64)
    var0 = rng.uniform(0.01, 0.2, (H, W))
    want_age = orc.increment_age(age0, cam, cam, T10, d0)
    want_d, want_v = orc.propagate(T10, cam, cam, d0, var0, 1.0, 10.0, 0.01)
    sd = ops.SemiDenseSession(1, H, W, max_refframes=2)
    sd.set_age_policy(False)
    sd.set_params(ops.make_params(0.5, 10.0, 0.01, 0.01, 0.004, 0.01), 1.0, 10.0, 0.01)
    sd.push_frame(0, cam, c["ref_image"], c["T_wr"])
    sd.push_frame(0, cam, c["key_image"], c["T_wk"])
    for mode in ("1", "0"):
        ops.set_option("sd_warp_gather", int(mode))
        got_age = ops.increment_age(age0, cam, cam, T10, d0)
        got_d, got_v = ops.propagate(T10, cam, cam, d0, var0, 1.0, 10.0, 0.01)
        assert np.array_equal(got_age, want_age), mode
        assert np.array_equal(got_d, want_d) and np.array_equal(got_v, want_v), mode
        sd.set_maps(0, d0, var0, age0)
        before = sd.warp_fallbacks()
        sd.propagate(T10[None], commit=False)
        if mode != "0":
            assert sd.warp_fallbacks() == before, f"mode {mode}: the box left the gather's window"
        d1, v1, a1 = sd.get_results(0, with_flag=False)
        assert np.array_equal(a1, want_age) and np.array_equal(d1, want_d) and np.array_equal(v1, want_v), mode
    sd.close()


def test_session_counts_warp_fallbacks(ops, orc, monkeypatch):
    """The session's fused increment_age + propagate: tracks with a small displacement box take the gather,
    a track with a zoom-out takes the slot path in the same launch; both bit-exact, the counter says which."""
    ops.set_option("sd_warp_gather", 1)
    H, W, n = 96, 128, 3
    pargs = (0.5, 10.0, 0.01, 0.01, 0.004, 0.01)
    sd = ops.SemiDenseSession(n, H, W, max_refframes=2)
    sd.set_age_policy(False)
    sd.set_params(ops.make_params(*pargs), 1.0, 10.0, 0.01)
    cases, T10s = [], []
    for t, kind in enumerate(("stereo", "zoom", "motion")):
        c, T10, age0, var0 = _warp_case(H, W, 40 + t, kind)
        sd.push_frame(t, c["cam"], c["ref_image"], c["T_wr"])
        sd.push_frame(t, c["cam"], c["key_image"], c["T_wk"])
        sd.set_maps(t, c["prior_depth"], var0, age0)
        cases.append((c, age0, var0)); T10s.append(T10)
    assert sd.warp_fallbacks() == 0
    sd.propagate(np.array(T10s), commit=False)
    assert sd.warp_fallbacks() == 1                      # the zoom-out track
    for t, (c, age0, var0) in enumerate(cases):
        d1, v1, a1 = sd.get_results(t, with_flag=False)
        assert np.array_equal(a1, orc.increment_age(age0, c["cam"], c["cam"], T10s[t], c["prior_depth"]))
        od, ov = orc.propagate(T10s[t], c["cam"], c["cam"], c["prior_depth"], var0, 1.0, 10.0, 0.01)
        assert np.array_equal(d1, od) and np.array_equal(v1, ov)
    sd.close()


# ---------------------------------------------------------------------------
# the search's five quotients by one refined reciprocal (shared_quotient, csrc/semi_dense.hip) against IEEE divisions
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("special", ["finite", "inf_nan"])
def test_update_depth_mixed_magnitude_reference_frames(ops, orc, special):
    """Reference frames whose texels mix ordinary intensities with zeros, negative zeros, denormals, values far
    outside [2^-401, 2^400) and (second case) Inf / NaN, texel by texel, so that search windows hold every mixture
    of "plain" and other numerators: flags, depths and variances equal the oracle's (IEEE divisions on the CPU)."""
    from tadataka_amd import synthetic
    H, W = 96, 128
    c = synthetic.make_semi_dense_case(H, W, seed=5)
    pg, po = ops.make_params(0.5, 10.0, 0.01, 0.01, 0.004, 0.01), orc.make_params(0.5, 10.0, 0.01, 0.01, 0.004, 0.01)
    age = np.ones((H, W), dtype=np.uint64)
    rng = np.random.default_rng(9)
    scales = [1.0, 1.0, 1.0, 1e-130, 1e-200, 1e-310, 0.0, -0.0, 1e150, -1.0, 3e-122, 2.6e120]
    if special == "inf_nan":
        scales += [np.inf, np.nan, -np.inf]
    with np.errstate(all="ignore"):
        for density in (0.02, 0.3, 1.0):
            pick = rng.integers(0, len(scales), (H, W))
            mult = np.asarray(scales)[pick]
            mult = np.where(rng.uniform(size=(H, W)) < density, mult, 1.0)
            ref_image = np.ascontiguousarray(c["ref_image"] * mult)
            key = (c["cam"], c["key_image"], c["T_wk"]); ref = (c["cam"], ref_image, c["T_wr"])
            d, v, f = ops.update_depth(key, [ref], age, c["prior_depth"], c["prior_variance"], pg)
            od, ov, of = orc.update_depth(key, [ref], age, c["prior_depth"], c["prior_variance"], po)
            assert np.array_equal(f, of), (special, density, int((f != of).sum()))
            assert np.array_equal(d, od, equal_nan=True) and np.array_equal(v, ov, equal_nan=True), (special, density)
            assert int((of == 0).sum()) > 0 or density == 1.0
