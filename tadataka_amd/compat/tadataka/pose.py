"""tadataka.pose (reference tadataka/pose.py:19-75): Pose = SciPy Rotation + t.

`WorldPose` does not exist in the reference at this revision although
examples/dvo_pose_change.py:6 imports it; it is provided as an
alias of Pose.  The essential-matrix / PnP helpers of the reference need
OpenCV and belong to the feature-based front end (out of scope): they are
importable and raise at call time."""
import numpy as np
from scipy.spatial.transform import Rotation

from tadataka.matrix import motion_matrix
from tadataka.se3 import exp_se3_t_

min_correspondences = 6


class Pose(object):
    """`rotation` (a SciPy Rotation, as in the reference) is built on first access when the pose came
    from a rotation MATRIX -- every result of the device loop does, and constructing a Rotation costs
    ~25 us, a sixth of a 640x480 estimation."""

    def __init__(self, rotation, translation):
        assert(isinstance(rotation, Rotation))
        self._rotation = rotation
        self._R = None
        self.t = translation

    @classmethod
    def _from_R(cls, R, t):
        obj = cls.__new__(cls)
        obj._rotation = None
        obj._R = R
        obj.t = t
        return obj

    @property
    def rotation(self):
        if self._rotation is None:
            # from here on the Rotation is the single source of truth, as in the reference (from_matrix
            # re-orthonormalises: R and rotation.as_matrix() may differ in the last bits)
            self._rotation, self._R = Rotation.from_matrix(self._R), None
        return self._rotation

    @rotation.setter
    def rotation(self, value):
        assert(isinstance(value, Rotation))
        self._rotation, self._R = value, None

    @property
    def R(self):
        return self._R.copy() if self._R is not None else self._rotation.as_matrix()

    @property
    def T(self):
        return motion_matrix(self.R, self.t)

    def __str__(self):
        fmt = lambda values: ' '.join("{: .3f}".format(v) for v in values)
        return ("rotvec = [ " + fmt(self.rotation.as_rotvec()) +
                " ]  t = [ " + fmt(self.t) + " ]")

    @classmethod
    def identity(cls):
        return cls._from_R(np.eye(3), np.zeros(3))

    @classmethod
    def from_se3(cls, xi):
        return cls(Rotation.from_rotvec(xi[3:]), exp_se3_t_(xi))

    @classmethod
    def from_matrix(cls, T):
        """From a 4x4 (or 12-double {R, t}) rigid motion, e.g. a device result."""
        T = np.asarray(T, dtype=np.float64)
        if T.size == 12:
            return cls._from_R(T[:9].reshape(3, 3).copy(), T[9:].copy())
        return cls._from_R(T[0:3, 0:3].copy(), T[0:3, 3].copy())

    def inv(self):
        return Pose(*convert_coordinate(self.rotation, self.t))

    def __mul__(self, other):
        rotation = self.rotation                    # (first: R below is then this Rotation's matrix)
        return Pose(rotation * other.rotation,
                    np.dot(self.R, other.t) + self.t)

    def __eq__(self, other):
        return (np.isclose(self.rotation.as_rotvec(),
                           other.rotation.as_rotvec()).all() and
                np.isclose(self.t, other.t).all())


WorldPose = Pose


def convert_coordinate(rotation, t):
    inv_rotation = rotation.inv()
    return inv_rotation, -np.dot(inv_rotation.as_matrix(), t)


def calc_reprojection_threshold(keypoints, k=2.0):
    center = np.mean(keypoints, axis=0, keepdims=True)
    rms = np.sqrt(np.mean(np.sum(np.power(keypoints - center, 2), axis=1)))
    return k * rms / keypoints.shape[0]


def _feature_based_only(name):
    def stub(*args, **kwargs):
        raise NotImplementedError(
            f"tadataka.pose.{name} belongs to the feature-based front end "
            "(needs OpenCV); it is outside the MI355X hot-path build")
    stub.__name__ = name
    return stub


solve_pnp = _feature_based_only("solve_pnp")
n_triangulated = _feature_based_only("n_triangulated")
triangulation_indices = _feature_based_only("triangulation_indices")
select_valid_pose = _feature_based_only("select_valid_pose")
pose_change_from_stereo = _feature_based_only("pose_change_from_stereo")
estimate_pose_change = _feature_based_only("estimate_pose_change")
