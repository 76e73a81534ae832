#!/opt/conda/bin/python3.9
"""Round-6 fixture: what the REAL scikit-image 0.18.3 / the reference's PoseChangeEstimator do with frames that are not
float64 -- measured, so that the drop-in's two documented differences rest on data:

  float32 frames   skimage keeps single precision through rescale (float32 Gaussian output, a float32 instance of the
                   bilinear warp, a float32 estimated map); the drop-in widens to float64.  Recorded: the reference's
                   pose on float32 frames and on the same frames widened to float64 (configs[1], 3 levels, None / Huber)
                   and rescale(float32) itself next to rescale(float64) of a small image.
  integer frames   skimage runs the anti-aliasing prefilter IN the integer dtype (scipy.ndimage.gaussian_filter keeps
                   it: the filtered image is quantised) and converts afterwards; the drop-in refuses integer frames
                   (TypeError).  Recorded: rescale(uint8) against rescale(img_as_float(uint8)) and the reference's pose on
                   uint8 images against its pose on img_as_float of them.

Runs under /opt/conda/bin/python3.9 only (like generate_golden_skimage.py, whose helpers it uses); writes
tests/golden/skimage_dtypes.npz.  Nothing of skimage or of the reference travels, only arrays."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import generate_golden_skimage as gs             # noqa: E402  (imports the real skimage first, then the reference stubs)
import generate_golden_r3 as g3                  # noqa: E402
from tadataka_amd import synthetic               # noqa: E402
from skimage import img_as_float                 # noqa: E402


def main():
    out = {}
    gs.use_real_skimage()
    rng = np.random.default_rng(606)
    # --- rescale itself, small images ---
    img = rng.random((48, 64))
    for ki, s in enumerate((1.0, 1 / 1.5, 1 / 2.25)):
        r64 = gs.checked_rescale(img, s)                 # (records the plan of this shape, too)
        r32 = gs.real_rescale(img.astype(np.float32), s)
        assert r32.dtype == np.float32, r32.dtype
        out[f"f32_rescale_{ki}"] = r32
        out[f"f64_rescale_{ki}"] = r64
    out["f32_in"] = img
    u8 = rng.integers(0, 256, (48, 64)).astype(np.uint8)
    out["u8_in"] = u8
    for ki, s in enumerate((1.0, 1 / 1.5, 1 / 2.25)):
        ru = gs.real_rescale(u8, s)
        rf = gs.real_rescale(img_as_float(u8), s)
        assert ru.dtype == np.float64
        out[f"u8_rescale_{ki}"] = ru
        out[f"u8_as_float_rescale_{ki}"] = rf
        print("rescale(u8) vs rescale(img_as_float(u8)), scale %.3f: max |diff| = %.3e" % (s, np.max(np.abs(ru - rf))), flush=True)
    # --- the reference's estimator on configs[1] ---
    pair = synthetic.make_pair(480, 640, seed=0)
    for name in ("None", "huber"):
        mode = g3.mode_arg(name, None)
        r64 = gs.run(pair["cam"], pair["I0"], pair["D0"], pair["I1"], mode, 3)
        f32 = {k: pair[k].astype(np.float32) for k in ("I0", "D0", "I1")}
        r32 = gs.run(pair["cam"], f32["I0"], f32["D0"], f32["I1"], mode, 3)
        # the SAME single-precision values, widened: what the drop-in computes from float32 frames
        w = {k: f32[k].astype(np.float64) for k in f32}
        rw = gs.run(pair["cam"], w["I0"], w["D0"], w["I1"], mode, 3)
        for tag, rec in (("f64", r64), ("f32", r32), ("f32_widened", rw)):
            out[f"dvo_{tag}_{name}_rotvec"] = rec["rotvec"]
            out[f"dvo_{tag}_{name}_t"] = rec["t"]
            out[f"dvo_{tag}_{name}_evals"] = np.asarray(rec["evals"])
        print(name, "f32 vs widened: |dt| %.3e |drot| %.3e; widened vs f64 frames: |dt| %.3e" % (
            np.max(np.abs(r32["t"] - rw["t"])), np.max(np.abs(r32["rotvec"] - rw["rotvec"])),
            np.max(np.abs(rw["t"] - r64["t"]))), flush=True)
        # 8-bit images (depth stays float): the reference on uint8 frames and on img_as_float of them
        q = {k: np.clip(np.round(pair[k] * 255.0), 0, 255).astype(np.uint8) for k in ("I0", "I1")}
        ru = gs.run(pair["cam"], q["I0"], pair["D0"], q["I1"], mode, 3)
        rf = gs.run(pair["cam"], img_as_float(q["I0"]), pair["D0"], img_as_float(q["I1"]), mode, 3)
        for tag, rec in (("u8", ru), ("u8_as_float", rf)):
            out[f"dvo_{tag}_{name}_rotvec"] = rec["rotvec"]
            out[f"dvo_{tag}_{name}_t"] = rec["t"]
            out[f"dvo_{tag}_{name}_evals"] = np.asarray(rec["evals"])
        print(name, "u8 frames vs img_as_float(u8): |dt| %.3e |drot| %.3e" % (
            np.max(np.abs(ru["t"] - rf["t"])), np.max(np.abs(ru["rotvec"] - rf["rotvec"]))), flush=True)
    gs.store_plans(out)
    np.savez_compressed(os.path.join(HERE, "skimage_dtypes.npz"), **out)
    print("wrote skimage_dtypes.npz", len(out), "arrays")


if __name__ == "__main__":
    main()
