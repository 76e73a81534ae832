"""GPU tests added in round 6.

  * the clip slots of a batch pyramid after a build of FEWER arrays (advisor, round 5: "left clean" was a flag, not a
    count -- a partial build followed by a full one read slots nobody had initialised)
  * the streaming pyramid kernel on strips / segments / waves that mix the three tap runs of the identity-scale level
    (stream_level0_chunk: uniform fast paths + the general form), bit for bit against the oracle

Everything goes through the C ABI."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
sys.path.insert(0, os.path.dirname(__file__))

pytestmark = [pytest.mark.gpu]


@pytest.fixture(scope="module")
def ops():
    from tadataka_amd import _lib, ops as o
    _lib.require_gpu()
    return o


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def _plane(shape, value):
    """A constant plane: filtered image and blends round to either side of the constant, so clip=True changes outputs
    at BOTH bounds (measured with the oracle: 0.1 -> 1434 outputs raised / 640 lowered at level 1, 0.7 -> 205 / 140 at
    level 2)."""
    return np.full(shape, value)


@pytest.mark.parametrize("stream", [0, 2])
def test_partial_build_then_full_build_with_clip(ops, orc, stream):
    """build_pyramid(['I1']) (3 clip slots, left clean by k_clip_small) and then build_pyramid() (9 slots): every level of
    every array equals skimage's with clip=True -- incl. the arrays whose slots the first build never touched and whose
    clip acts at the lower bound (an uninitialised slot decodes to NaN bounds: the clip would be skipped)."""
    from tadataka_amd import rescale_plan
    ops.set_option("pyramid_stream", stream)
    H, W, L = 96, 128, 3
    plans = rescale_plan.level_plans((H, W), L)
    imgs = {"I0": _plane((H, W), 0.8631789223498866), "D0": _plane((H, W), 0.1), "I1": _plane((H, W), 0.7)}
    acts = 0
    for name, img in imgs.items():
        for l in range(L):
            a = orc.rescale_skimage(img, 1 / 1.5 ** l, plans[l], clip=True)
            b = orc.rescale_skimage(img, 1 / 1.5 ** l, plans[l], clip=False)
            acts += int(not np.array_equal(a, b))
    assert acts >= 3, "the scene must make clip=True act, or the test proves nothing"
    for order in (("partial", "full"), ("full", "partial", "full")):
        batch = ops.DvoBatch(1, H, W, n_levels=L, ratio=1.5)
        batch.set_skimage_pyramid(plans, level0="all", clip=True)
        batch.upload(0, imgs["I0"], imgs["D0"], imgs["I1"])
        for what in order:
            if what == "partial":
                batch.build_pyramid(["I1"])
            else:
                batch.build_pyramid()
        for name, img in imgs.items():
            for l in range(L):
                want = orc.rescale_skimage(img, 1 / 1.5 ** l, plans[l], clip=True)
                got = batch.download(0, l, name)
                assert np.array_equal(got, want), (order, name, l, float(np.max(np.abs(got - want))))
        # a change of the rescale options between builds (another slot layout) must not reuse "clean" either
        batch.set_skimage_pyramid(plans, level0=["D0"], clip=True)
        batch.build_pyramid(["D0"])
        batch.set_skimage_pyramid(plans, level0="all", clip=True)
        batch.build_pyramid()
        for name, img in imgs.items():
            for l in range(L):
                want = orc.rescale_skimage(img, 1 / 1.5 ** l, plans[l], clip=True)
                assert np.array_equal(batch.download(0, l, name), want), (order, name, l, "after option change")
        batch.close()
    ops.set_option("pyramid_stream", 1)


@pytest.mark.parametrize("shape", [(120, 160), (480, 640), (97, 333), (64, 500)])
def test_streaming_level0_tap_runs_bit_for_bit(ops, orc, shape):
    """rescale(., 1.0)'s estimated map a o + b puts an axis into three runs -- taps (o, o + 1), o alone, (o - 1, o) -- and
    the streaming kernel takes a specialised path where a wave's columns and a chunk's rows lie inside one run, the
    general form elsewhere.  Maps with the runs in different places (offsets of either sign, scales an ulp off 1) and
    the interpreter's own plan: level 0 and the two shrinking levels bit for bit against the oracle, clip included."""
    from tadataka_amd import rescale_plan
    ops.set_option("pyramid_stream", 2)
    H, W = shape
    L = 3
    base = rescale_plan.level_plans((H, W), L)
    rng = np.random.default_rng(H * 1000 + W)
    eps = np.finfo(float).eps
    maps = [base[0]["map"],
            np.array([1.0, 3e-14, 1.0, -2e-14]),
            np.array([1.0 + eps, -1e-13, 1.0 - eps / 2, 4e-14]),
            np.array([1.0 - eps / 2, 2e-12, 1.0 + eps, -3e-12]),
            np.array([1.0, 0.0, 1.0, 0.0]),
            np.array([1.0, 0.25, 1.0, -0.25])]
    B = 3
    frames = [{k: (rng.random((H, W)) if k != "D0" else rng.uniform(0.5, 4.0, (H, W))) for k in ("I0", "D0", "I1")}
              for _ in range(B)]
    frames[1]["D0"] = np.full((H, W), 0.8631789223498866)          # a plane: clip acts
    frames[2]["I1"] = np.minimum(frames[2]["I1"] * 1.7, 1.0)       # a saturated plateau
    for m in maps:
        plans = [dict(base[0], map=m)] + base[1:]
        batch = ops.DvoBatch(B, H, W, n_levels=L, ratio=1.5)
        batch.set_skimage_pyramid(plans, level0="all", clip=True)
        for i, fr in enumerate(frames):
            batch.upload(i, fr["I0"], fr["D0"], fr["I1"])
        batch.build_pyramid()
        for i, fr in enumerate(frames):
            for name in ("I0", "D0", "I1"):
                for l in range(L):
                    want = orc.rescale_skimage(fr[name], 1 / 1.5 ** l, plans[l], clip=True)
                    got = batch.download(i, l, name)
                    assert np.array_equal(got, want), (m.tolist(), i, name, l, float(np.max(np.abs(got - want))))
        batch.close()
    ops.set_option("pyramid_stream", 1)


@pytest.mark.skimage_pyramid
def test_dropin_on_float32_and_integer_frames_against_the_dtype_fixture(ops, golden):
    """float32 frames are widened: the drop-in's pose is the reference's pose on the widened values to 1e-6 (and so
    ~1e-5 from the reference's single-precision pose: the measured, documented gap of INTEGRATION.md).  Integer frames
    are refused, in the estimator and in the scikit-image stand-in's filtered rescale."""
    import tadataka_amd  # noqa: F401
    import tadataka.vo.dvo as dvo
    from skimage.transform import rescale
    from tadataka.camera import CameraModel, CameraParameters
    from tadataka_amd import rescale_plan, synthetic
    g = golden("skimage_dtypes.npz")
    pair = synthetic.make_pair(480, 640, seed=0)
    cm = CameraModel(CameraParameters(pair["cam"][0:2], pair["cam"][2:4]), distortion_model=None)
    f32 = {k: pair[k].astype(np.float32) for k in ("I0", "D0", "I1")}
    u8 = np.clip(np.round(pair["I0"] * 255), 0, 255).astype(np.uint8)
    dvo.PYRAMID_PLANS = lambda shape, n, ratio: rescale_plan.recorded_level_plans(g, (int(shape[0]), int(shape[1])), n, ratio)
    try:
        est = dvo.PoseChangeEstimator(cm, cm, n_coarse_to_fine=3, max_iter=20)
        for name in ("None", "huber"):
            pose = est(f32["I0"], f32["D0"], f32["I1"], None if name == "None" else name)
            err = max(np.max(np.abs(pose.rotation.as_rotvec() - g[f"dvo_f32_widened_{name}_rotvec"])),
                      np.max(np.abs(pose.t - g[f"dvo_f32_widened_{name}_t"])))
            assert err < 1e-6, (name, err)
            assert 1e-6 < np.max(np.abs(pose.t - g[f"dvo_f32_{name}_t"])) < 5e-5      # the single-precision reference
        with pytest.raises(TypeError):
            est(u8, pair["D0"], u8)
    finally:
        dvo.PYRAMID_PLANS = None
    with pytest.raises(NotImplementedError):
        rescale(u8, 1 / 1.5)
    assert rescale(u8, 1.0).shape == u8.shape and rescale(u8, 1 / 1.5, anti_aliasing=False).shape == (320, 427)
