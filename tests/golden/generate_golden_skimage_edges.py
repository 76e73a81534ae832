#!/opt/conda/bin/python3.9
"""Edge cases of skimage.transform.rescale that the randomised soak (tests/test_gpu_fuzz.py) found the device
handling differently from the oracle -- pinned here against the REAL scikit-image 0.18.3 of the build
container's conda interpreter, so that the oracle's reading of them is not an assumption:

    /opt/conda/bin/python3.9 tests/golden/generate_golden_skimage_edges.py   ->  skimage_rescale_edges.npz

  * an output axis of ONE pixel: resize() estimates its affine map from a degenerate corner set, the scale of
    that axis comes out as exactly 0 and only the offset is used;
  * a sample position that is an integer (11 x 71 -> 7 x 47: row map 11/7 o + 2/7 hits 5.0): the second tap is
    ceil(p) = floor(p), the pixel next to it is never read -- visible with an Inf / NaN there;
  * clip=True without the prefilter on an image whose only NaN is tapped by no output: numpy.clip's bounds are
    the image's min / max, i.e. NaN, i.e. every output is NaN;
  * the same three with anti_aliasing on / off and clip on / off.
Every record: image, scale, flags, the plan this interpreter produced, skimage's output.  As in
generate_golden_skimage.py the generator asserts that the oracle fed with the plan reproduces skimage bit
for bit before it writes anything."""
import os
import sys
import warnings

warnings.filterwarnings("ignore")
import numpy as np
import skimage
from skimage.transform import rescale

assert skimage.__version__ == "0.18.3", skimage.__version__
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as orc                 # noqa: E402


def main():
    rng = np.random.default_rng(20260930)
    out = {}
    names = []

    def add(name, img, scale, aa, clip):
        ref = rescale(img, scale, anti_aliasing=aa, clip=clip)
        plan = orc.skimage_plan(img.shape, ref.shape, aa)
        got = orc.rescale_skimage(img, scale, plan=plan, anti_aliasing=aa, clip=clip)
        assert got.shape == ref.shape and np.array_equal(got, ref, equal_nan=True), (name, aa, clip)
        key = f"{name}_aa{int(aa)}_clip{int(clip)}"
        names.append(key)
        out[key + "_image"] = img
        out[key + "_scale"] = np.float64(scale)
        out[key + "_flags"] = np.array([int(aa), int(clip)])
        out[key + "_map"] = plan["map"]
        out[key + "_wr"] = np.zeros(0) if plan["wr"] is None else plan["wr"]
        out[key + "_wc"] = np.zeros(0) if plan["wc"] is None else plan["wc"]
        out[key + "_out"] = ref

    for aa in (True, False):
        for clip in (True, False):
            # one-pixel output axes
            add("one_row", rng.uniform(0, 1, (2, 45)), 2 / 3, aa, clip)
            add("one_row_b", rng.uniform(0, 1, (3, 67)), 1 / 2.25, aa, clip)
            add("one_col", rng.uniform(0, 1, (45, 2)), 2 / 3, aa, clip)
            add("one_px", rng.uniform(0, 1, (2, 2)), 0.5, aa, clip)
            # integer sample positions, finite and with non-finite pixels around them
            base = rng.uniform(0, 1, (11, 71))
            add("int_pos", base, 2 / 3, aa, clip)
            for tag, bad in (("nan", np.nan), ("inf", np.inf), ("ninf", -np.inf)):
                img = base.copy()
                img[6, 40] = bad               # the row below the integer position y = 5.0 of output row 3
                add("int_pos_" + tag, img, 2 / 3, aa, clip)
                img = base.copy()
                img[5, 40] = bad               # on it
                add("int_pos_on_" + tag, img, 2 / 3, aa, clip)
            # scale 1/3: the estimated map is 3 o + 1 up to a few ulp -- whichever positions come out as exact
            # integers in this interpreter read one pixel, the others two
            for tag, bad in (("nan", np.nan), ("inf", np.inf)):
                img = rng.uniform(0, 1, (12, 15))
                img[5, :] = bad                # below the positions y = 4 of output row 1
                img[:, 8] = bad                # right of the positions x = 7 of output column 2
                add("third_" + tag, img, 1 / 3, aa, clip)
            img = rng.uniform(0, 1, (23, 28))
            img[12, 9] = np.nan
            add("int_pos_wide", img, 0.19614148867970432, aa, clip)
            # a NaN that no tap of a sparse (unfiltered) sampling sees
            img = rng.uniform(0, 1, (34, 26))
            img[20, 13] = np.nan
            add("untapped_nan", img, 0.1734810006610749, aa, clip)
            img = rng.uniform(0, 1, (53, 22))
            img[1, 1] = np.nan
            add("untapped_nan_b", img, 0.14779952916856445, aa, clip)
            img = rng.uniform(0, 1, (40, 40))
            img[17, 22] = np.inf
            add("sparse_inf", img, 0.2, aa, clip)
    # img_as_float of the integer types (what rgb2gray / rescale do to an 8-bit frame first): a product with the rounded
    # reciprocal of the type's maximum, not a division -- 24 of the 256 uint8 values differ in the last bit
    from skimage import img_as_float
    from skimage.color import rgb2gray
    for dt in (np.uint8, np.int8, np.uint16, np.int16):
        info = np.iinfo(dt)
        x = np.arange(info.min, info.max + 1, dtype=dt)
        if x.size > 256:
            x = x[rng.integers(0, x.size, 400)]
            x[:2] = (info.min, info.max)
        out[f"as_float_{np.dtype(dt).name}_in"] = x
        out[f"as_float_{np.dtype(dt).name}_out"] = img_as_float(x)
    gray8 = rng.integers(0, 256, (12, 16), dtype=np.uint8)
    out["gray_u8_in"] = gray8
    out["gray_u8_as_float"] = img_as_float(gray8)
    out["names"] = np.array(names)
    path = os.path.join(HERE, "skimage_rescale_edges.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(names), "records", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
