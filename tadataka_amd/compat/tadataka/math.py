"""tadataka.math (reference tadataka/math.py:5-45).

solve_linear_equation(A, b, weights) minimises ||sqrt(W)(A x - b)||.  The
reference scales the n x p matrix row by row and hands it to LAPACK's gelsd;
here the n-row reduction A^T W A, A^T W b runs on the device
(tdk_weighted_normal_equations) and only the p x p system is solved on the host.
The solve is minimum-norm, like lstsq, when the system is rank deficient.  Known
limit (there is deliberately no CPU fallback that would run lstsq on the n rows):
normal equations square the condition number, so for cond(A) beyond ~1e6 (low
texture, planar scenes) directions that gelsd would still keep fall under the
relative eigenvalue cut-off 1e-13 and are truncated.  method="cg" runs scipy's
conjugate gradient on the reduced p x p system, which is the system the reference
hands to it as well."""
import numpy as np

from tadataka_amd import ops


def weighted_mean(x, w):
    assert(x.shape == w.shape)
    s = w.sum()
    if s == 0:
        raise ValueError("Sum of weights is zero")
    return (x * w).sum() / s


def solve_normal_equations(M, g):
    """x with M x = g for symmetric PSD M; pseudo-inverse on a relative
    eigenvalue cut-off when M is singular."""
    M = np.asarray(M, dtype=np.float64)
    scale = np.sqrt(np.where(np.diag(M) > 0, np.diag(M), 1.0))
    Ms = M / np.outer(scale, scale)
    lam, V = np.linalg.eigh(Ms)
    keep = lam > 1e-13 * max(lam.max(), 0.0)
    coeff = np.zeros_like(lam)
    coeff[keep] = (V.T @ (g / scale))[keep] / lam[keep]
    return (V @ coeff) / scale


def solve_linear_equation(A, b, weights=None, method="lstsq", **kwargs):
    A = np.asarray(A, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert(A.shape[0] == b.shape[0])
    if weights is not None:
        assert(A.shape[0] == weights.shape[0])
    if method not in ("lstsq", "cg"):
        raise ValueError(f"No such method '{method}'")
    if A.shape[1] > 8:
        raise ValueError("the device reduction supports at most 8 unknowns")
    # the n-row reduction on the device, the p x p solve on the host
    M, g = ops.weighted_normal_equations(A, b, weights)
    if method == "cg":
        # the reference runs scipy's cg on exactly this reduced system (math.py:21-23)
        from scipy.sparse import linalg
        x, _ = linalg.cg(M, g, **kwargs)
        return x
    return solve_normal_equations(M, g)
