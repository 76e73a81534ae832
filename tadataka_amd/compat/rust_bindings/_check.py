import numpy as np


def f64(a, ndim, name):
    """rust-numpy 0.9 refuses anything but an ndarray of the exact dtype
    (SURVEY §8b); mirror that instead of converting silently."""
    if not isinstance(a, np.ndarray) or a.dtype != np.float64:
        raise TypeError(f"{name} must be a numpy.ndarray of dtype float64")
    if a.ndim != ndim:
        raise TypeError(f"{name} must be {ndim}-dimensional")
    return a


def typed(a, dtype, ndim, name):
    if not isinstance(a, np.ndarray) or a.dtype != np.dtype(dtype):
        raise TypeError(f"{name} must be a numpy.ndarray of dtype {np.dtype(dtype).name}")
    if a.ndim != ndim:
        raise TypeError(f"{name} must be {ndim}-dimensional")
    return a
