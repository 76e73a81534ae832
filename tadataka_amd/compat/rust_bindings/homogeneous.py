"""rust_bindings.homogeneous (src/py/homogeneous.rs:6-44): append / drop the
trailing 1.  Pure data movement, no arithmetic."""
import numpy as np

from rust_bindings._check import f64


def to_homogeneous_vec(x):
    return np.append(f64(x, 1, "x"), 1.0)


def to_homogeneous_vecs(xs):
    xs = f64(xs, 2, "xs")
    return np.hstack([xs, np.ones((xs.shape[0], 1))])


def from_homogeneous_vec(x):
    return f64(x, 1, "x")[:-1].copy()


def from_homogeneous_vecs(xs):
    return f64(xs, 2, "xs")[:, :-1].copy()
