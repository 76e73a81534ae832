"""tadataka.projection (reference tadataka/projection.py:10-26)."""
import numpy as np

from rust_bindings import projection

EPSILON = 1e-16


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def pi(P):
    """Project 3D points onto the normalized image plane: xy / (z + 1e-16)."""
    P = _f64(P)
    if P.ndim == 1:
        return projection.project_vec(P)
    return projection.project_vecs(P)


def inv_pi(xs, depths):
    """Back-projection from the normalized image plane: [x, y, 1] * depth."""
    xs = _f64(xs)
    if xs.ndim == 1:
        return projection.inv_project_vec(xs, depths)
    return projection.inv_project_vecs(xs, _f64(depths))


class PerspectiveProjection(object):
    def __init__(self, camera_parameters):
        self.camera_parameters = camera_parameters

    def compute(self, P):
        K = self.camera_parameters.matrix
        return pi(np.dot(K, P.T).T)
