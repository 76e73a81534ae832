"""tadataka.robust.weights (reference tadataka/robust/weights.py:4-43): robust
IRLS weights of a residual vector, computed on the device -- including the
global statistics (Student-t: 10 fixed-point variance iterations; Tukey: two
medians by radix select)."""
import numpy as np

from tadataka_amd import ops


def _as_vector(r):
    r = np.asarray(r, dtype=np.float64)
    return r, r.reshape(-1)


def compute_weights_huber(r, k=1.345):
    r, flat = _as_vector(r)
    return ops.robust_weights(flat, ops.W_HUBER, k).reshape(r.shape)


def compute_weights_student_t(r, nu=5, n_iter=10):
    """NB: returns the square ROOT of the Student-t weight, as the reference does."""
    r, flat = _as_vector(r)
    return ops.robust_weights(flat, ops.W_STUDENT_T, nu, n_iter).reshape(r.shape)


def compute_weights_tukey(r, beta=4.6851, c=1.4826):
    r, flat = _as_vector(r)
    return ops.robust_weights(flat, ops.W_TUKEY, beta, c).reshape(r.shape)


def tukey(x, beta):
    x = np.asarray(x, dtype=np.float64)
    w = np.zeros(x.shape)
    mask = np.abs(x) <= beta
    w[mask] = np.power(1 - np.power(x[mask] / beta, 2), 2)
    return w


def median_absolute_deviation(x):
    return np.median(np.abs(x - np.median(x)))
