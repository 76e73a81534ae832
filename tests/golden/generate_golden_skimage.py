#!/opt/conda/bin/python3.9
"""Round-5 fixtures: the pyramid -- and BASELINE configs[1] end to end -- pinned against a REAL
scikit-image.  The build container's second interpreter (/opt/conda/bin/python3.9: scikit-image
0.18.3, numpy 1.26.4, scipy 1.7.1) has what the default python3 lacks; this script runs THERE
and only there (the .npz outputs travel, nothing of skimage or of the reference does):

    /opt/conda/bin/python3.9 tests/golden/generate_golden_skimage.py [--only rescale|dvo|seeds]

  skimage_rescale.npz   (i)  skimage.transform.rescale(image, s) and rescale(image, s,
        anti_aliasing=False) for 8 shapes x the pyramid scales 1 / 1.5^k (k = 0: the identity
        scale the reference also sends through rescale, tadataka/vo/dvo/__init__.py:144-148) and
        0.5, on a random image and on a smooth depth-like map; a constant image, an image with a
        saturated plateau at its maximum (where clip=True acts), an 8-bit image / 255, an image
        with one NaN; rgb2gray of uint8 RGB / RGBA.  Beside every output: the PLAN the generating
        interpreter produced for that (shape, scale) -- the affine map resize() estimates by SVD
        and scipy.ndimage's Gaussian kernels -- because those two are products of the installed
        NumPy / LAPACK / libm, not of the algorithm (oracle/tdk_oracle.c: orc_rescale_skimage).
        Small outputs are stored whole, large ones as SHA-256 + a few rows.
  skimage_dvo.npz       (ii) the reference's own PoseChangeEstimator (imported from /root/reference
        with the stubs of generate_golden.py, `rescale` being the REAL skimage.transform.rescale):
        BASELINE configs[1] (seed-0 VGA pair, 3 levels) with every weight option, the examples'
        5- and 7-level settings, the New-Tsukuba pair (full and half resolution), the holes and
        ill-conditioned scenes; per record final pose, pose after every level, PhotometricError
        evaluations per level.
  skimage_seeds.npz     (ii) 40 further seeds at 120x160 and 32 at 480x640 (None and Huber).

The generator ASSERTS, for every plan it stores, that the oracle's restatement fed with that plan
reproduces skimage's output bit for bit -- a fixture that would not be reproducible is never written.
"""
import hashlib
import os
import sys
import warnings

warnings.filterwarnings("ignore")

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))

import skimage                                   # noqa: E402  the real one, before any stub
import skimage.color                             # noqa: E402
import skimage.transform                         # noqa: E402

REAL = {k: v for k, v in sys.modules.items() if k == "skimage" or k.startswith("skimage.")}
real_rescale = skimage.transform.rescale
real_rgb2gray = skimage.color.rgb2gray
assert skimage.__version__ == "0.18.3", skimage.__version__

sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
import generate_golden as gg                     # noqa: E402
import generate_golden_r3 as g3                  # noqa: E402
import scenes                                    # noqa: E402
from tadataka_amd import synthetic               # noqa: E402
from oracle import oracle as orc                 # noqa: E402  (plan restatement + self-check only)

PLANS = {}


def plan_key(in_shape, out_shape):
    return f"plan_{in_shape[0]}x{in_shape[1]}_{out_shape[0]}x{out_shape[1]}"


def plan_for(in_shape, out_shape):
    """The plan THIS interpreter's NumPy gives (skimage's own calls, restated in oracle.py)."""
    key = plan_key(in_shape, out_shape)
    if key not in PLANS:
        PLANS[key] = orc.skimage_plan(in_shape, out_shape)
    return PLANS[key]


def store_plans(out, keys=None):
    for key, p in PLANS.items():
        if keys is not None and key not in keys:
            continue
        out[key + "_map"] = p["map"]
        out[key + "_wr"] = np.zeros(0) if p["wr"] is None else p["wr"]
        out[key + "_wc"] = np.zeros(0) if p["wc"] is None else p["wc"]


def checked_rescale(image, scale, **kw):
    """skimage.transform.rescale, with the assertion that the oracle + this interpreter's plan is bit-identical."""
    o = real_rescale(image, scale, **kw)
    if image.ndim == 2 and image.dtype == np.float64:
        plan = plan_for(image.shape, o.shape)
        aa = kw.get("anti_aliasing", True)
        mine = orc.rescale_skimage(image, scale, plan if aa else dict(plan, wr=None, wc=None))
        same = np.array_equal(mine, o, equal_nan=True)
        assert same, ("oracle restatement differs from skimage", image.shape, scale, np.nanmax(np.abs(mine - o)))
    return o


def put_array(out, tag, a):
    a = np.ascontiguousarray(a)
    out[tag + "_shape"] = np.array(a.shape)
    out[tag + "_sha"] = np.frombuffer(hashlib.sha256(a.tobytes()).digest(), dtype=np.uint8)
    if a.size <= 6000:
        out[tag] = a
    else:                                        # large outputs: the digest decides, a few rows say where it went wrong
        out[tag + "_rows"] = a[::max(7, a.shape[0] // 6)].copy()


def depth_like(shape, seed):
    h, w = shape
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    rng = np.random.default_rng(seed)
    return 2.0 + 0.3 * np.sin(x / 40.0 * 640 / w) + 0.2 * np.cos(y / 30.0 * 480 / h) + 0.001 * rng.random(shape)


def capture_rescale():
    out = {}
    shapes = [(48, 64), (60, 80), (37, 53), (96, 128), (120, 160), (240, 320), (480, 640), (720, 1280)]
    scales = [1.0, 1 / 1.5, 1 / 1.5 ** 2, 1 / 1.5 ** 3, 0.5]
    cases = []
    for si, shape in enumerate(shapes):
        rng = np.random.default_rng(1000 + si)
        img = rng.random(shape)
        dep = depth_like(shape, 2000 + si)
        for ki, s in enumerate(scales):
            if shape == (720, 1280) and ki not in (0, 1):
                continue
            tag = f"r{si}_{ki}"
            cases.append((tag, shape, s))
            put_array(out, tag + "_aa", checked_rescale(img, s))
            put_array(out, tag + "_bl", checked_rescale(img, s, anti_aliasing=False))
            put_array(out, tag + "_dep", checked_rescale(dep, s))
        out[f"r{si}_seed"] = np.array([1000 + si, 2000 + si])
    out["case_tags"] = np.array([c[0] for c in cases])
    out["case_shapes"] = np.array([c[1] for c in cases])
    out["case_scales"] = np.array([c[2] for c in cases])
    # clip=True at work and the special values
    rng = np.random.default_rng(77)
    special = {
        "const": np.full((48, 64), 0.8631789223498866),       # a plane: clip=True acts at both levels
        "plateau": np.minimum(rng.random((60, 80)) * 1.6, 1.0) * 0.8631789223498866,   # 40 % of the pixels on the maximum
        "u8": rng.integers(0, 256, (60, 80)).astype(np.float64) / 255.0,
        "u8sat": np.minimum(rng.integers(0, 400, (60, 80)), 255).astype(np.float64) / 255.0,
        "planes": np.where(np.arange(80)[None, :] < 40, 2.0, 3.7) * np.ones((60, 1)),
    }
    nan_img = rng.random((48, 64))
    nan_img[20, 30] = np.nan
    special["nan"] = nan_img
    for name, img in special.items():
        out[f"sp_{name}_in"] = img
        for ki, s in enumerate([1.0, 1 / 1.5, 1 / 1.5 ** 2]):
            o = checked_rescale(img, s)
            out[f"sp_{name}_{ki}"] = o
            # would the result differ without clip=True?  (recorded: the tests name the cases where it acts)
            plan = plan_for(img.shape, o.shape)
            noclip = orc.rescale_skimage(img, s, plan, clip=False)
            out[f"sp_{name}_{ki}_clip_acts"] = np.array(int(not np.array_equal(noclip, o, equal_nan=True)))
    # rgb2gray (examples/dvo_pose_change.py:24, examples/semi_dense_vo.py:65)
    rgb = rng.integers(0, 256, (60, 80, 3), dtype=np.uint8)
    rgba = rng.integers(0, 256, (30, 40, 4), dtype=np.uint8)
    out["rgb"] = rgb
    out["rgb_gray"] = real_rgb2gray(rgb)
    out["rgba"] = rgba
    out["rgba_gray"] = real_rgb2gray(skimage.color.rgba2rgb(rgba))
    store_plans(out)
    out["versions"] = np.array([skimage.__version__, np.__version__, __import__("scipy").__version__])
    return out


# ---------------------------------------------------------------------------
# (ii) the reference's PoseChangeEstimator on the real rescale
# ---------------------------------------------------------------------------
def use_real_skimage():
    gg.install_stubs()
    sys.modules.update(REAL)                     # the real skimage back in place of the stubs
    import tadataka.vo.dvo as dvo
    dvo.rescale = checked_rescale                # == skimage.transform.rescale, self-checked
    return dvo


def run(cam, I0, D0, I1, weights, n_levels, max_iter=20):
    rec = g3.run_pyramid(cam, cam, I0, D0, I1, weights, n_levels, True, max_iter=max_iter)
    rec.pop("xis")
    return rec


def level_shapes(shape, n_levels, ratio=1.5):
    return [orc.rescale_shape(shape, 1 / ratio ** l) for l in range(n_levels)]


def capture_dvo():
    use_real_skimage()
    out = {}
    PLANS.clear()
    # BASELINE configs[1]: the seed-0 VGA pair, 3 levels, every weight option of tests/vo/test_dvo.py:46-50
    pair = synthetic.make_pair(480, 640, seed=0)
    wmap = scenes.weight_map((480, 640), seed=41)
    for name in ("None", "huber", "student-t", "tukey", "map"):
        rec = run(pair["cam"], pair["I0"], pair["D0"], pair["I1"], g3.mode_arg(name, wmap), 3)
        g3.put(out, f"v3_{name}", rec)
        print("v3", name, rec["evals"], rec["rotvec"], rec["t"], flush=True)
    # examples/semi_dense_vo.py:45-54: 7 levels, weights = 1 / variance
    rec = run(pair["cam"], pair["I0"], pair["D0"], pair["I1"], wmap, 7)
    g3.put(out, "ex7_map", rec)
    print("ex7", rec["evals"], rec["t"], flush=True)
    # examples/dvo_pose_change.py:34-37: 5 levels on a 240x320 pair
    pair5 = synthetic.make_pair(240, 320, seed=5)
    for name in ("None", "huber"):
        rec = run(pair5["cam"], pair5["I0"], pair5["D0"], pair5["I1"], g3.mode_arg(name, None), 5)
        g3.put(out, f"ex5_{name}", rec)
        print("ex5", name, rec["evals"], rec["t"], flush=True)
    # the small pyramid case of dvo_pyramid.npz
    pairs = synthetic.make_pair(120, 160, seed=4)
    for name in ("None", "huber"):
        rec = run(pairs["cam"], pairs["I0"], pairs["D0"], pairs["I1"], g3.mode_arg(name, None), 3)
        g3.put(out, f"s3_{name}", rec)
        print("s3", name, rec["evals"], rec["t"], flush=True)
    # New-Tsukuba frames (tests/vo/test_dvo.py:24-53's dataset), 5 levels, full and half resolution
    rgb0 = g3.load_rgb(os.path.join(g3.TSUKUBA, "tsukuba_daylight_L_00201.png"))
    rgb1 = g3.load_rgb(os.path.join(g3.TSUKUBA, "tsukuba_daylight_L_00205.png"))
    I0 = scenes.gray_from_rgb_u8(rgb0)
    I1 = scenes.gray_from_rgb_u8(rgb1)
    out["tsukuba_gray_sha"] = np.frombuffer(hashlib.sha256(I0.tobytes() + I1.tobytes()).digest(), dtype=np.uint8)
    H, W = I0.shape
    D0 = scenes.tsukuba_depth(H, W)
    cam = scenes.TSUKUBA_CAM
    wm = scenes.weight_map((H, W), seed=42)
    for name in ("None", "huber", "student-t", "tukey", "map"):
        rec = run(cam, I0, D0, I1, g3.mode_arg(name, wm), 5)
        g3.put(out, f"tsu_full_{name}", rec)
        print("tsukuba full", name, rec["evals"], rec["t"], flush=True)
    I0h, I1h, D0h = (checked_rescale(a, 0.5) for a in (I0, I1, D0))   # the example's get(): rescale(., 0.5)
    out["tsu_half_shape"] = np.array(I0h.shape)
    for name in ("None", "huber"):
        rec = run(cam * 0.5, I0h, D0h, I1h, g3.mode_arg(name, None), 5)
        g3.put(out, f"tsu_half_{name}", rec)
        print("tsukuba half", name, rec["evals"], rec["t"], flush=True)
    # depth maps with missing readings, ill-conditioned scenes (tests/golden/scenes.py)
    for tag, fill in (("zero", 0.0), ("nan", np.nan)):
        p = scenes.holes_pair(fill)
        for name in ("None", "huber", "tukey", "student-t"):
            rec = run(p["cam"], p["I0"], p["D0"], p["I1"], g3.mode_arg(name, None), 3)
            g3.put(out, f"holes_{tag}_{name}", rec)
            print("holes", tag, name, rec["evals"], rec["t"], flush=True)
    for scene in scenes.ILL_SCENES:
        p = scenes.ill_pair(scene)
        for name in ("None", "huber"):
            rec = run(p["cam"], p["I0"], p["D0"], p["I1"], g3.mode_arg(name, None), 3)
            g3.put(out, f"ill_{scene}_{name}", rec)
            print("ill", scene, name, rec["evals"], rec["t"], flush=True)
    store_plans(out)
    out["versions"] = np.array([skimage.__version__, np.__version__, __import__("scipy").__version__])
    return out


def capture_seeds():
    use_real_skimage()
    out = {}
    PLANS.clear()
    for (h, w, n_seeds, names) in ((120, 160, 40, ("None", "huber")), (480, 640, 32, ("huber",))):
        for seed in range(100, 100 + n_seeds):
            pair = synthetic.make_pair(h, w, seed=seed)
            for name in names:
                rec = run(pair["cam"], pair["I0"], pair["D0"], pair["I1"], g3.mode_arg(name, None), 3)
                g3.put(out, f"p{h}_{seed}_{name}", rec)
            print("seed", h, seed, rec["evals"], rec["t"], flush=True)
        out[f"p{h}_seeds"] = np.arange(100, 100 + n_seeds)
    store_plans(out)
    out["versions"] = np.array([skimage.__version__, np.__version__, __import__("scipy").__version__])
    return out


def main():
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    jobs = {"skimage_rescale.npz": capture_rescale, "skimage_dvo.npz": capture_dvo, "skimage_seeds.npz": capture_seeds}
    for fname, fn in jobs.items():
        if only and only not in fname:
            continue
        np.savez_compressed(os.path.join(HERE, fname), **fn())
        print(fname, os.path.getsize(os.path.join(HERE, fname)), flush=True)


if __name__ == "__main__":
    main()
