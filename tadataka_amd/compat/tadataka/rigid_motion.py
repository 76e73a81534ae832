"""tadataka.rigid_motion: similarity transform between two point sets in the least-squares sense
(reference: tadataka/rigid_motion.py:25-107, the method of Zinsser et al.; imported by examples/plot.py:12,
which both examples import).  Host-side and tiny -- nothing here touches the device."""
import numpy as np


def calculate_rotation(X, Y):
    """R minimising sum |R x_i - y_i|^2 for centred X, Y (rigid_motion.py:25-30): with
    X^T Y = U S V^T, R = V U^T (no determinant correction, as in the reference)."""
    U, _, VT = np.linalg.svd(np.dot(X.T, Y))
    return np.dot(VT.T, U.T)


def calculate_scaling(X, Y, R):
    """s = sum_i y_i^T R x_i / sum_i x_i^T x_i (rigid_motion.py:33-36)."""
    return np.sum(Y * np.dot(X, R.T)) / np.sum(X * X)


def calculate_translation(s, R, p, q):
    return q - s * np.dot(R, p)


class LeastSquaresRigidMotion(object):
    """`R, t, s = LeastSquaresRigidMotion(P, Q).solve()` such that `s R p_i + t ~ q_i`."""
    def __init__(self, P, Q):
        if P.shape != Q.shape:
            raise ValueError("P and Q must be the same shape")
        self.n_features = P.shape[1]
        self.P = P
        self.Q = Q

    def solve(self):
        mean_p = np.mean(self.P, axis=0)
        mean_q = np.mean(self.Q, axis=0)
        X = self.P - mean_p
        Y = self.Q - mean_q
        R = calculate_rotation(X, Y)
        s = calculate_scaling(X, Y, R)
        return R, calculate_translation(s, R, mean_p, mean_q), s
