"""tadataka.matrix (reference tadataka/matrix.py:8-81): 4x4 rigid-motion
bookkeeping on the host (a handful of 3x3 products per frame)."""
import numpy as np

from rust_bindings import homogeneous


def get_rotation(T):
    return T[0:3, 0:3]


def get_translation(T):
    return T[0:3, 3]


def get_rotation_translation(T):
    return get_rotation(T), get_translation(T)


def motion_matrix(R, t):
    T = np.zeros((4, 4))
    T[0:3, 0:3] = R
    T[0:3, 3] = t
    T[3, 3] = 1
    return T


def inv_motion_matrix(T):
    R, t = get_rotation_translation(T)
    return motion_matrix(R.T, -np.dot(R.T, t))


def calc_relative_transform(T_wa, T_wb):
    """T_ab = inv(T_wa) T_wb."""
    return np.dot(inv_motion_matrix(T_wa), T_wb)


def homogeneous_matrix(A, b):
    if A.shape[0] != A.shape[1]:
        raise ValueError("'A' must be a square matrix")
    if A.shape[0] != b.shape[0]:
        raise ValueError("Number of rows of 'A' must match "
                         "the number of elements of 'b'")
    d = A.shape[0]
    W = np.identity(d + 1)
    W[0:d, 0:d] = A
    W[0:d, d] = b
    return W


def to_homogeneous(X):
    X = np.ascontiguousarray(X, dtype=np.float64)
    if X.ndim == 1:
        return homogeneous.to_homogeneous_vec(X)
    return homogeneous.to_homogeneous_vecs(X)


def from_homogeneous(X):
    if X.ndim == 1:
        return X[0:X.shape[0] - 1]
    return X[:, 0:X.shape[1] - 1]


def homogeneous_transformation(X, T):
    Y = np.dot(T, to_homogeneous(X).T).T
    return from_homogeneous(Y)
