"""CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

NumPy-structured restatement of one DVO iteration, array pass by array pass the
way the reference's Python runs it (CPU baseline (i): "the
apples-to-apples reference CPU path"): every intermediate of the reference is
materialised here too (coordinate grid, normalised points, P0, P1, projected
coordinates, the mask compaction of every array, the M x 6 Jacobian, lstsq on
it).  The compiled leaves of the reference (rust_bindings.*, _normalizer) are
single vectorised NumPy expressions -- if anything faster than the per-row Rust
loops they stand for, so timing this does not flatter the GPU.

Only tests/ and bench.py's cpu_baseline leg import this module.

Reference lines followed:
  one_iteration     <- tadataka/vo/dvo/__init__.py:93-110 (body of the Gauss-Newton loop)
  calc_pose_update  <- tadataka/vo/dvo/__init__.py:46-70
  calc_jacobian     <- tadataka/vo/dvo/jacobian.py:8-24
  photometric_error <- tadataka/metric.py:13-39, tadataka/warp.py:78-88
  image_coordinates <- tadataka/coordinates.py:7-19
  interpolation     <- tadataka/interpolation/__init__.py:13-29, src/interpolation.rs:9-43
  weights           <- tadataka/robust/weights.py:38-43 (Huber)
  solve             <- tadataka/math.py:17-19,32-45 (lstsq on sqrt(w)-scaled rows)
"""
import numpy as np


def image_coordinates(shape):
    h, w = shape[0:2]
    xs, ys = np.meshgrid(np.arange(w), np.arange(h))
    return np.column_stack((xs.flatten(), ys.flatten()))


def normalize(us, cam):
    return (np.asarray(us, dtype=np.float64) - cam[2:4]) / cam[0:2]


def unnormalize(xs, cam):
    return xs * cam[0:2] + cam[2:4]


def inv_pi(xs, depths):
    return np.column_stack((xs[:, 0] * depths, xs[:, 1] * depths, depths))


def pi(P):
    return P[:, 0:2] / (P[:, [2]] + 1e-16)


def transform(R, t, P):
    return np.dot(R, P.T).T + t


def is_in_image_range(us, shape):
    h, w = shape[0:2]
    xs, ys = us[:, 0], us[:, 1]
    return np.logical_and(np.logical_and(0 <= xs, xs <= w - 1), np.logical_and(0 <= ys, ys <= h - 1))


def interpolation(image, C):
    """Bilinear samples with the reference's integer-coordinate short cuts folded
    into clamped upper indices (value-identical for finite images)."""
    cx, cy = C[:, 0], C[:, 1]
    lx, ly = np.floor(cx), np.floor(cy)
    lxi, lyi = lx.astype(np.int64), ly.astype(np.int64)
    uxi = np.minimum(lxi + 1, image.shape[1] - 1)
    uyi = np.minimum(lyi + 1, image.shape[0] - 1)
    ux, uy = lx + 1.0, ly + 1.0
    return (image[lyi, lxi] * (ux - cx) * (uy - cy) + image[lyi, uxi] * (cx - lx) * (uy - cy) +
            image[uyi, lxi] * (ux - cx) * (cy - ly) + image[uyi, uxi] * (cx - lx) * (cy - ly))


def calc_image_gradient(image):
    DY, DX = np.gradient(image)
    return DX, DY


def calc_jacobian(focal_length, didx, didy, P):
    fx, fy = focal_length
    fgx, fgy = fx * didx, fy * didy
    x, y, z = P[:, 0], P[:, 1], P[:, 2]
    z2 = z * z
    xy = x * y
    return np.column_stack((fgx / z, fgy / z, -(fgx * x + fgy * y) / (z * z),
                            -(fgx * xy + fgy * (z2 + y * y)) / z2,
                            (fgx * (z2 + x * x) + fgy * xy) / z2, (-fgx * y + fgy * x) / z))


def compute_weights_huber(residuals, k=1.345):
    r = np.abs(residuals)
    w = np.ones(r.shape)
    big = r > k
    w[big] = k / r[big]
    return w


def solve_linear_equation(J, r, weights=None):
    if weights is None:
        return np.linalg.lstsq(J, r, rcond=None)[0]
    sw = np.sqrt(weights)
    return np.linalg.lstsq(J * sw.reshape(-1, 1), r * sw, rcond=None)[0]


def calc_pose_update(cam1, residuals, GX1, GY1, P1, weights):
    us1 = unnormalize(pi(P1), cam1)
    mask = is_in_image_range(us1, GX1.shape) & (P1[:, 2] > 0)
    if not np.any(mask):
        return None
    r = residuals[mask]
    p1 = P1[mask]
    gx1 = interpolation(GX1, us1[mask])
    gy1 = interpolation(GY1, us1[mask])
    J = calc_jacobian(cam1[0:2], gx1, gy1, p1)
    if weights is None:
        return solve_linear_equation(J, r)
    if weights == "huber":
        return solve_linear_equation(J, r, compute_weights_huber(r))
    raise ValueError(f"No such weights '{weights}'")


def photometric_error(I0, D0, I1, cam0, cam1, T10):
    us0 = image_coordinates(D0.shape)
    xs0 = normalize(us0, cam0)
    P1 = transform(T10[:3, :3], T10[:3, 3], inv_pi(xs0, D0.flatten()))
    us1 = unnormalize(pi(P1), cam1)
    mask = is_in_image_range(us1, D0.shape)
    m0 = us0[mask]
    i0 = I0[m0[:, 1], m0[:, 0]]
    i1 = interpolation(I1, us1[mask])
    d = i0 - i1
    return np.mean(d * d)


class Level(object):
    """What _PoseChangeEstimator.__call__ prepares once per level (:83-90)."""
    def __init__(self, I0, D0, I1, cam0, cam1):
        self.I0, self.D0, self.I1 = I0, D0, I1
        self.cam0, self.cam1 = np.asarray(cam0, dtype=np.float64), np.asarray(cam1, dtype=np.float64)
        us0 = image_coordinates(I0.shape)
        self.P0 = inv_pi(normalize(us0, self.cam0), D0.flatten())
        self.GX1, self.GY1 = calc_image_gradient(I1)
        self.residuals = (I0 - I1).flatten()


def one_iteration(level, T10, weights="huber"):
    """One pass of the loop body: calc_pose_update at T10, then the photometric
    error (evaluated at T10: the candidate composition is host-side noise).
    Returns (xi, error)."""
    P1 = transform(T10[:3, :3], T10[:3, 3], level.P0)
    xi = calc_pose_update(level.cam1, level.residuals, level.GX1, level.GY1, P1, weights)
    err = photometric_error(level.I0, level.D0, level.I1, level.cam0, level.cam1, T10)
    return xi, err
