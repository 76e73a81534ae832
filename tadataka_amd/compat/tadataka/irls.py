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


class HuberT(object):
    """The M-estimator object `fit` takes (irls.py:69-175; statsmodels' HuberT): tuning constant t, and for
    z = residual / scale the criterion rho, its derivative psi, the IRLS weights psi(z) / z and psi's derivative."""

    def __init__(self, t=1.345):
        self.t = t

    def _subset(self, z):
        return np.abs(np.asarray(z)) <= self.t

    def rho(self, z):
        z = np.asarray(z)
        return np.where(self._subset(z), 0.5 * z ** 2, np.abs(z) * self.t - 0.5 * self.t ** 2)

    def psi(self, z):
        z = np.asarray(z)
        return np.where(self._subset(z), z, self.t * np.sign(z))

    def weights(self, z):
        return huber_weights(z, self.t)

    def psi_deriv(self, z):
        return self._subset(z)

    def __call__(self, z):
        return self.rho(z)


class Residual(object):
    """y - X params (irls.py:178-183)."""

    def __init__(self, X, y):
        self.X, self.y = X, y

    def compute(self, params):
        return self.y - self.X.dot(params)


def _wls(X, y, w=None):
    M, g = ops.weighted_normal_equations(X, y, w)
    return solve_normal_equations(M, g)


def least_squares(X, y):
    """irls.py:186-188 (lstsq there; the p x p normal equations of the device reduction here)."""
    return _wls(np.asarray(X, dtype=np.float64), np.asarray(y, dtype=np.float64))


def weighted_least_squares(X, y, weights):
    """irls.py:191-196."""
    return _wls(np.asarray(X, dtype=np.float64), np.asarray(y, dtype=np.float64), np.asarray(weights, dtype=np.float64))


def fit(X, y, max_iter=100, M=None):
    """irls.py:199-218; M: an object with .weights(z) (default HuberT())."""
    M = HuberT() if M is None else M
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    params = _wls(X, y)
    resid = y - X @ params
    scale = mad(resid)
    for _ in range(max_iter):
        if scale == 0.0:
            break
        params = _wls(X, y, M.weights(resid / scale))
        resid = y - X @ params
        scale = mad(resid)
    return params
