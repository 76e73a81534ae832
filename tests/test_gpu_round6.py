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
