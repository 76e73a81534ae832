"""The interpreter-dependent constants of skimage.transform.rescale / resize.

The reference builds every pyramid level -- level 0 at scale 1.0 included -- with
`skimage.transform.rescale(image, scale)` (tadataka/vo/dvo/__init__.py:144-148).  Two
ingredients of that call are not determined by the algorithm but by the NumPy / LAPACK /
libm of the interpreter it runs in:

  * the affine map of the bilinear warp: `resize()` ESTIMATES it from three corner
    correspondences (`AffineTransform.estimate`: Hartley normalisation, `numpy.linalg.svd`,
    `numpy.linalg.inv`; skimage/transform/_warps.py:156-176, _geometric.py:18-69, 652-702)
    instead of using `factor` and `factor / 2 - 1 / 2`; the estimate is a few ulp off in the
    scale and ~1e-13 off in the offset, differently on every LAPACK build;
  * the Gaussian kernels of the anti-aliasing prefilter: `numpy.exp(-0.5 / sigma**2 * x**2)`
    normalised by its `.sum()` (scipy/ndimage/filters.py `_gaussian_kernel1d`), where NumPy's
    SIMD `exp` and pairwise sum differ from libm's in the last bit.

And the last bits matter: at the identity prior the right / bottom border of every level projects
exactly onto the inclusive mask boundary, so a depth level that differs by one ulp moves a few
hundred pixels across it and the recovered pose by ~1e-5.

This module makes the SAME NumPy calls skimage and scipy make, in the same order, so that on any
interpreter the device pyramid is what `skimage.transform.rescale` would return there
(verified bit for bit against scikit-image 0.18.3: tests/golden/skimage_rescale.npz, whose
generator ran these restatements next to the real package).  The kernels of csrc/pyramid.hip
take the result as arguments (`tdk_rescale_skimage`, `tdk_dvo_set_level_plan`).
"""
import math
from functools import lru_cache

import numpy as np

__all__ = ["resize_map", "gaussian_kernel", "resize_plan", "level_plans", "rescale_shape", "recorded_level_plans"]


def rescale_shape(shape, scale):
    """rescale(): output_shape = np.round(scale * shape) (skimage/transform/_warps.py:286)."""
    return (int(np.round(shape[0] * scale)), int(np.round(shape[1] * scale)))


def _center_and_normalize_points(points):
    centroid = np.mean(points, axis=0)
    rms = math.sqrt(np.sum((points - centroid) ** 2) / points.shape[0])
    norm_factor = math.sqrt(2) / rms
    matrix = np.array([[norm_factor, 0, -norm_factor * centroid[0]],
                       [0, norm_factor, -norm_factor * centroid[1]],
                       [0, 0, 1]])
    pointsh = np.vstack([points.T, np.ones((points.shape[0]),)])
    new_pointsh = (matrix @ pointsh).T
    new_points = new_pointsh[:, :2]
    new_points[:, 0] /= new_pointsh[:, 2]
    new_points[:, 1] /= new_pointsh[:, 2]
    return matrix, new_points


def resize_map(in_shape, out_shape):
    """(ax, bx, ay, by): output (row oy, column ox) of resize() samples (ay * oy + by, ax * ox + bx)."""
    rows, cols = float(out_shape[0]), float(out_shape[1])
    factors = np.asarray(in_shape, dtype=float) / np.asarray([rows, cols], dtype=float)
    if rows == 1 and cols == 1:
        return np.array([1.0, in_shape[1] / 2.0 - 0.5, 1.0, in_shape[0] / 2.0 - 0.5])
    src = np.array([[1, 1], [1, rows], [cols, rows]]) - 1
    dst = np.zeros(src.shape, dtype=np.double)
    dst[:, 0] = factors[1] * (src[:, 0] + 0.5) - 0.5
    dst[:, 1] = factors[0] * (src[:, 1] + 0.5) - 0.5
    src_matrix, s = _center_and_normalize_points(src)
    dst_matrix, d = _center_and_normalize_points(dst)
    xs, ys, xd, yd = s[:, 0], s[:, 1], d[:, 0], d[:, 1]
    n = s.shape[0]
    A = np.zeros((n * 2, 9))
    A[:n, 0] = xs
    A[:n, 1] = ys
    A[:n, 2] = 1
    A[:n, 6] = -xd * xs
    A[:n, 7] = -xd * ys
    A[n:, 3] = xs
    A[n:, 4] = ys
    A[n:, 5] = 1
    A[n:, 6] = -yd * xs
    A[n:, 7] = -yd * ys
    A[:n, 8] = xd
    A[n:, 8] = yd
    coeffs = list(range(6))                      # AffineTransform._coeffs
    A = A[:, coeffs + [8]]
    _, _, V = np.linalg.svd(A)
    H = np.zeros((3, 3))
    H.flat[coeffs + [8]] = -V[-1, :-1] / V[-1, -1]
    H[2, 2] = 1
    H = np.linalg.inv(dst_matrix) @ H @ src_matrix
    return np.array([H[0, 0], H[0, 2], H[1, 1], H[1, 2]])


def gaussian_kernel(sigma):
    """gaussian_filter1d's kernel (truncate = 4), or None where gaussian_filter skips the axis."""
    if not sigma > 1e-15:
        return None
    radius = int(4.0 * float(sigma) + 0.5)
    sigma2 = sigma * sigma
    x = np.arange(-radius, radius + 1)
    phi_x = np.exp(-0.5 / sigma2 * x ** 2)
    return np.ascontiguousarray((phi_x / phi_x.sum())[::-1])


@lru_cache(maxsize=256)
def _plan(in_shape, out_shape, anti_aliasing):
    factors = np.asarray(in_shape, dtype=float) / np.asarray(out_shape, dtype=float)
    sigma = np.maximum(0, (factors - 1) / 2)
    return {"map": resize_map(in_shape, out_shape),
            "wr": gaussian_kernel(sigma[0]) if anti_aliasing else None,
            "wc": gaussian_kernel(sigma[1]) if anti_aliasing else None}


def resize_plan(in_shape, out_shape, anti_aliasing=True):
    """{'map': (ax, bx, ay, by), 'wr': kernel over rows or None, 'wc': ... columns} of
    skimage.transform.resize(image, out_shape) for an image of in_shape (cached per shape)."""
    p = _plan((int(in_shape[0]), int(in_shape[1])), (int(out_shape[0]), int(out_shape[1])), bool(anti_aliasing))
    return dict(p)


def level_plans(shape, n_levels, ratio=1.5, anti_aliasing=True):
    """Plans of the levels 0 .. n_levels - 1 of PoseChangeEstimator's pyramid (scale 1 / ratio**level)."""
    return [resize_plan(shape, rescale_shape(shape, 1 / pow(ratio, level)), anti_aliasing)
            for level in range(n_levels)]


def recorded_level_plans(records, shape, n_levels, ratio=1.5):
    """The plans another interpreter produced, from records `plan_{H}x{W}_{Ho}x{Wo}_{map,wr,wc}` (a mapping, e.g. an
    opened tests/golden/skimage_*.npz): with them the device pyramid equals what scikit-image returned THERE, bit for
    bit -- how bench.py and the tests hold results against the reference run on a real scikit-image."""
    plans = []
    for level in range(n_levels):
        ho, wo = rescale_shape(shape, 1 / pow(ratio, level))
        key = f"plan_{int(shape[0])}x{int(shape[1])}_{ho}x{wo}"
        plans.append({"map": records[key + "_map"], "wr": records[key + "_wr"], "wc": records[key + "_wc"]})
    return plans
