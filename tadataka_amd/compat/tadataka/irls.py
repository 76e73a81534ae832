"""tadataka.irls (reference tadataka/irls.py:186-218): Huber IRLS `fit(X, y)`
-- OLS start, MAD scale, then weighted least squares with w = huber(r / scale).
Every X^T W X / X^T W y reduction runs on the device
(tdk_weighted_normal_equations); the p x p solves stay on the host.  The
reference's only call site uses p = 3 (flow_estimation.py:10-12)."""
import numpy as np

from tadataka.math import solve_normal_equations
from tadataka_amd import ops


def mad(a, c=0.6744897501960817, axis=0):
    """median(|a| / c), c = Phi^-1(3/4): the residuals are centred on zero, not on
    their median (irls.py:43-62)."""
    return np.median(np.abs(np.asarray(a)) / c, axis=axis)


def huber_weights(z, t=1.345):
    """HuberT.weights (irls.py:137-159): 1 inside |z| <= t, t / |z| outside."""
    z = np.asarray(z, dtype=np.float64)
    absz = np.abs(z)
    return np.where(absz <= t, 1.0, t / np.where(absz > 0, absz, 1.0))


def _wls(X, y, w=None):
    M, g = ops.weighted_normal_equations(X, y, w)
    return solve_normal_equations(M, g)


def fit(X, y, max_iter=100):
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    params = _wls(X, y)
    resid = y - X @ params
    scale = mad(resid)
    for _ in range(max_iter):
        if scale == 0.0:
            break
        params = _wls(X, y, huber_weights(resid / scale))
        resid = y - X @ params
        scale = mad(resid)
    return params
