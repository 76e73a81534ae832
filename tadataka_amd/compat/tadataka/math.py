"""tadataka.math (reference tadataka/math.py:5-45).

solve_linear_equation(A, b, weights) minimises ||sqrt(W)(A x - b)||.  The
reference scales the n x p matrix row by row and hands it to LAPACK's gelsd;
here the n-row reduction A^T W A, A^T W b runs on the device
(tdk_weighted_normal_equations) and only the p x p system is solved on the host.
The solve is minimum-norm, like lstsq, when the system is rank deficient
(tests/golden/dvo_ill.npz: the reference's lstsq twists on rank-deficient and badly
scaled scenes are reproduced to 1e-9).  Known limit (there is deliberately no CPU
fallback that would run lstsq on the n rows): normal equations square the condition
number.  Ill-conditioning that is a column scale (one weak gradient direction; cond(A)
2e7 in the fixture) is removed by the Jacobi scaling of the p x p system; what remains
out of reach is cond(A) beyond ~1e6 AFTER column scaling, where directions gelsd would
still keep fall under the relative eigenvalue cut-off 1e-13 and are truncated.  method="cg" runs scipy's
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


def solve_normal_equations(M, g, n_rows=0):
    """x with M x = g for symmetric PSD M.  When M is singular: the minimum-norm
    least-squares solution, as lstsq returns for a rank-deficient A (the rank is decided
    on the Jacobi-scaled matrix with a relative eigenvalue cut-off, the norm is taken in
    the original coordinates).  Same algorithm as tdk::solve6 (csrc/tdk_math.h)."""
    M = np.array(M, dtype=np.float64)
    # columns of A whose norm is below gelsd's rcond * sigma_max, rcond = eps * max(n, p)
    # (numpy's default): dropped there whatever the other columns are; zeroed here before
    # the scaling would turn their rounding noise into a unit column
    d = np.diag(M)
    cut = (np.finfo(np.float64).eps * max(n_rows, M.shape[0])) ** 2
    tiny = ~(d > cut * d.max())
    M[tiny, :] = 0.0
    M[:, tiny] = 0.0
    scale = np.sqrt(np.where(np.diag(M) > 0, np.diag(M), 1.0))
    Ms = M / np.outer(scale, scale)
    lam, V = np.linalg.eigh(Ms)
    keep = lam > 1e-13 * max(lam.max(), 0.0)
    coeff = np.zeros_like(lam)
    coeff[keep] = (V.T @ (g / scale))[keep] / lam[keep]
    x = (V @ coeff) / scale
    if not keep.all():
        # null space of M in the original coordinates; remove x's component in it
        N, _ = np.linalg.qr(V[:, ~keep] / scale[:, None])
        x = x - N @ (N.T @ x)
    return x


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
    return solve_normal_equations(M, g, A.shape[0])
