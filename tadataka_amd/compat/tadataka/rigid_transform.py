"""tadataka.rigid_transform (reference tadataka/rigid_transform.py:111-123 for
the two functions on the hot path, plus the small batched helpers)."""
import numpy as np

from rust_bindings import transform as _transform


def transform_se3(T10, P0):
    """4x4 transform applied to [N,3] points on the device (src/transform.rs:9-28)."""
    T10 = np.ascontiguousarray(T10, dtype=np.float64)
    P0 = np.ascontiguousarray(P0, dtype=np.float64)
    if P0.ndim == 1:
        return _transform.transform(T10, P0.reshape(1, 3))[0]
    return _transform.transform(T10, P0)


def transform(R, t, P):
    """R p + t for one point [3] or points [N,3]."""
    assert(R.shape == (3, 3))
    assert(t.shape == (3,))
    T = np.identity(4)
    T[0:3, 0:3] = R
    T[0:3, 3] = t
    return transform_se3(T, P)


def inv_transform(R, t, P):
    return transform(R.T, -np.dot(R.T, t), P)


def transform_each(rotations, translations, points):
    assert(rotations.shape[0] == translations.shape[0] == points.shape[0])
    return np.einsum('ijk,ik->ij', rotations, points) + translations


def transform_all(rotations, translations, points):
    assert(rotations.shape[0] == translations.shape[0])
    rotated = np.einsum('ijk,lk->ilj', rotations, points)
    return rotated + translations[:, np.newaxis, :]


def inv_transform_all(rotations, translations, points):
    inv_rotations = np.swapaxes(rotations, 1, 2)
    inv_translations = -np.einsum('ijk,ik->ij', inv_rotations, translations)
    return transform_all(inv_rotations, inv_translations, points)


def rotate_each(rotations, points):
    assert(rotations.shape[0] == points.shape[0])
    return np.einsum('ijk,ik->ij', rotations, points)


class Transform(object):
    """q = s R p + t."""
    def __init__(self, R, t, s=1.0):
        self.R, self.t, self.s = R, t, s

    def __call__(self, P):
        return self.s * np.dot(self.R, P.T).T + self.t
