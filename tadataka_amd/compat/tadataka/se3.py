"""tadataka.se3 (reference tadataka/se3.py:15-64): the 6-vector twist
xi = [v, omega] <-> 4x4 motion.  Host-side, 6 numbers per frame pair; the same
formulas run inside the device Gauss-Newton loop (tdk_math.h exp_se3_t)."""
import numpy as np

from tadataka.so3 import exp_so3, log_so3, tangent_so3

EPSILON = 1e-16


def normalize(omega):
    theta = np.linalg.norm(omega)
    if theta == 0:
        return np.zeros(len(omega)), 0
    return omega / theta, theta


def _V(theta, K):
    I = np.eye(3)
    KK = np.dot(K, K)
    if theta < EPSILON:
        return I + K * theta / 2 + KK * pow(theta, 2) / 6
    return I + (1 - np.cos(theta)) / theta * K + (theta - np.sin(theta)) / theta * KK


def exp_se3_t_(xi):
    """Translation part of exp(xi): V(omega) v, K built from the unit axis."""
    v, rotvec = xi[:3], xi[3:]
    axis, theta = normalize(rotvec)
    return np.dot(_V(theta, tangent_so3(axis)), v)


def exp_se3(xi):
    G = np.identity(4)
    G[0:3, 0:3] = exp_so3(xi[3:])
    G[0:3, 3] = exp_se3_t_(xi)
    return G


def log_se3(G):
    R, t = G[0:3, 0:3], G[0:3, 3]
    axis, theta = normalize(log_so3(R))
    if theta == 0:
        return np.concatenate((t, axis * theta))
    K = tangent_so3(axis)
    beta = 1 - theta * np.sin(theta) / (2 * (1 - np.cos(theta)))
    V_inv = np.eye(3) - theta / 2 * K + beta * np.dot(K, K)
    return np.concatenate((V_inv.dot(t), axis * theta))


def get_rotation(G):
    return G[0:3, 0:3]


def get_translation(G):
    return G[0:3, 3]
