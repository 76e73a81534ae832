"""tadataka.vo.dvo.jacobian (reference tadataka/vo/dvo/jacobian.py:8-29).

Inside the estimator these two steps are fused into the device evaluation
kernel (the N x 6 Jacobian is never materialised); the functions are kept for
callers that want the arrays."""
import numpy as np

from tadataka_amd import ops


def calc_jacobian(focal_length, didx, didy, P):
    """Rows [fgx/z, fgy/z, -(fgx x + fgy y)/z^2, -(fgx xy + fgy (z^2+y^2))/z^2,
    (fgx (z^2+x^2) + fgy xy)/z^2, (-fgx y + fgy x)/z], fg* = f* * didx/didy
    (Kerl 2012); twist order [v, omega].  Host-side elementwise helper."""
    fx, fy = focal_length
    fgx, fgy = fx * np.asarray(didx), fy * np.asarray(didy)
    x, y, z = P[:, 0], P[:, 1], P[:, 2]
    z2, xy = z * z, x * y
    J = np.empty((P.shape[0], 6))
    J[:, 0] = fgx / z
    J[:, 1] = fgy / z
    J[:, 2] = -(fgx * x + fgy * y) / (z * z)
    J[:, 3] = -(fgx * xy + fgy * (z2 + y * y)) / z2
    J[:, 4] = (fgx * (z2 + x * x) + fgy * xy) / z2
    J[:, 5] = (-fgx * y + fgy * x) / z
    return J


def calc_image_gradient(image):
    """np.gradient of the image, returned as (DX, DY); computed on the device."""
    return ops.image_gradient(image)
