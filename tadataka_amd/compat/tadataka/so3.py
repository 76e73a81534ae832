"""tadataka.so3 (reference tadataka/so3.py:28-46): rotation-vector helpers."""
import numpy as np
from scipy.spatial.transform import Rotation

EPSILON = 1e-16


def is_rotation_matrix(R):
    assert(R.shape[0] == R.shape[1])
    return (np.isclose(np.dot(R, R.T), np.identity(3)).all() and
            np.isclose(np.linalg.det(R), 1.0))


def tangent_so3(v):
    """[v]x, the skew matrix with [v]x p = v x p."""
    x, y, z = v
    return np.array([[0, -z, y],
                     [z, 0, -x],
                     [-y, x, 0]])


def exp_so3(rotvec):
    return Rotation.from_rotvec(rotvec).as_matrix()


def log_so3(R):
    return Rotation.from_matrix(R).as_rotvec()
