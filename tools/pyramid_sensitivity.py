"""How much of the recovered pose is decided by the last bits of skimage.transform.rescale -- i.e. how
reproducible the REFERENCE's own PoseChangeEstimator is across NumPy / LAPACK builds.  CPU only (oracle loop,
pinned to 1e-16 against the reference on real scikit-image 0.18.3 by tests/test_oracle_skimage.py).

For the 72 seeded pairs of tests/golden/skimage_seeds.npz (40 at 120x160, 32 at 640x480; 3 levels, Huber,
max_iter 20) the pose of the reference run on the real skimage (scikit-image 0.18.3 / numpy 1.26.4 / scipy
1.7.1, the build container's /opt/conda interpreter) is compared with the same loop on variants of the pyramid:

  A  the generator's plans (estimated affine maps + scipy kernels recorded in the fixture)            -> parity
  B  THIS interpreter's plans: the same skimage / scipy source on another NumPy build (LAPACK, exp)   -> what the
     reference itself would return here
  C  the generator's plans with every map entry moved by ONE ulp
  D  ideal constants (rounds 1-4): (i + 0.5) f - 0.5, libm kernels, level 0 = the frame, no clip
  E  as A for the depth map only; the images (I0, I1) as in D
  F  as A, except that level 0 of the IMAGES is the frame itself (bench.py --pyramid skimage-depth-level0)

Usage: python tools/pyramid_sensitivity.py > profiles/r05_pyramid_sensitivity.txt"""
import os
import sys

import numpy as np
from scipy.spatial.transform import Rotation

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import oracle as orc            # noqa: E402
from tadataka_amd import synthetic          # noqa: E402


def loop(pair, weights, n_levels, rescale_of):
    """PoseChangeEstimator.__call__ with one rescale function per array."""
    rotation, t = Rotation.from_rotvec(np.zeros(3)), np.zeros(3)
    cam = np.asarray(pair["cam"], dtype=np.float64)
    for level in reversed(range(n_levels)):
        scale = 1 / pow(1.5, level)
        rotation, t = orc.dvo_estimate_level(rescale_of["I0"](pair["I0"], scale), rescale_of["D0"](pair["D0"], scale),
                                             rescale_of["I1"](pair["I1"], scale), cam * scale, cam * scale,
                                             rotation, t, weights, 20)
    return np.concatenate([rotation.as_rotvec(), t])


def main():
    g = np.load(os.path.join(REPO, "tests", "golden", "skimage_seeds.npz"))
    fixture = orc.fixture_plans(g)

    def with_plans(plans):
        def f(image, scale):
            return orc.rescale_skimage(image, scale, plans(image.shape, orc.rescale_shape(image.shape, scale)))
        return f

    def own(in_shape, out_shape):
        return orc.skimage_plan(in_shape, out_shape)

    def one_ulp(in_shape, out_shape):
        p = dict(fixture(in_shape, out_shape))
        p["map"] = np.nextafter(p["map"], np.inf)
        return p

    def ideal(image, scale):
        return image if scale == 1.0 else orc.rescale(image, scale, anti_aliasing=True)

    exact = with_plans(fixture)

    def exact_but_level0(image, scale):
        return image if scale == 1.0 else exact(image, scale)

    variants = {
        "A generator's plans": {k: exact for k in ("I0", "D0", "I1")},
        "B this interpreter's plans": {k: with_plans(own) for k in ("I0", "D0", "I1")},
        "C generator's plans + 1 ulp": {k: with_plans(one_ulp) for k in ("I0", "D0", "I1")},
        "D ideal constants": {k: ideal for k in ("I0", "D0", "I1")},
        "E depth exact, images ideal": {"I0": ideal, "D0": exact, "I1": ideal},
        "F exact, images' level 0 = frame": {"I0": exact_but_level0, "D0": exact, "I1": exact_but_level0},
    }
    import numpy
    import scipy
    print("pose gap |rotvec, t|_max against the reference's PoseChangeEstimator on real scikit-image 0.18.3 "
          f"(fixture interpreter: {' / '.join(str(v) for v in g['versions'])}; this one: numpy {numpy.__version__}, "
          f"scipy {scipy.__version__})")
    print("3 levels, ratio 1.5, max_iter 20, weights='huber'; the north_star bar on the pose is 1e-6\n")
    for (h, w) in ((120, 160), (480, 640)):
        seeds = [int(s) for s in g[f"p{h}_seeds"]]
        gaps = {name: [] for name in variants}
        for seed in seeds:
            pair = synthetic.make_pair(h, w, seed=seed)
            want = np.concatenate([g[f"p{h}_{seed}_huber_rotvec"], g[f"p{h}_{seed}_huber_t"]])
            for name, fns in variants.items():
                gaps[name].append(float(np.max(np.abs(loop(pair, "huber", 3, fns) - want))))
        print(f"{w}x{h}, {len(seeds)} pairs (seeds {seeds[0]}..{seeds[-1]})")
        print(f"  {'pyramid':36s} {'median':>10s} {'max':>10s} {'> 1e-6':>8s} {'> 1e-5':>8s}")
        for name, v in gaps.items():
            v = np.array(v)
            print(f"  {name:36s} {np.median(v):10.2e} {v.max():10.2e} {int((v > 1e-6).sum()):5d}/{len(v):<2d} "
                  f"{int((v > 1e-5).sum()):5d}/{len(v):<2d}")
        print()
    # the plans themselves
    print("the plans of 640x480 -> level, generator's interpreter vs this one (ax, bx, ay, by):")
    for level in range(3):
        out_shape = orc.rescale_shape((480, 640), 1 / 1.5 ** level)
        a, b = fixture((480, 640), out_shape)["map"], own((480, 640), out_shape)["map"]
        print(f"  level {level} {out_shape}: " + "  ".join(f"{float(x).hex()} | {float(y).hex()}" for x, y in zip(a, b)))
        print(f"           |difference| = " + "  ".join(f"{abs(x - y):.2e}" for x, y in zip(a, b)))


if __name__ == "__main__":
    main()
