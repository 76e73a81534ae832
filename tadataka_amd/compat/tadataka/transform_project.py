"""tadataka.transform_project: MI355X stand-in for the Cython module built from
sympy-generated C (reference tadataka/transform_project.pyx:22-50,
tadataka/so3_codegen.py:48-87).  pose = [omega(3), t(3)].

These are the single-observation entry points the reference exposes; the batch
path (tadataka.local_ba.Projection) evaluates all observations in one launch."""
import numpy as np

from tadataka_amd import ops

_ZERO = np.zeros(1, dtype=np.int64)


def _check(pose, point):
    pose = np.asarray(pose)
    point = np.asarray(point)
    if pose.dtype != np.float64 or point.dtype != np.float64 or pose.ndim != 1 or point.ndim != 1:
        raise ValueError("Buffer dtype mismatch, expected 1-D float64 arrays")
    return pose, point


def transform_project(pose, point):
    pose, point = _check(pose, point)
    return ops.ba_projection(pose[None], point[None], _ZERO, _ZERO, jacobians=False)[0]


def pose_jacobian(pose, point):
    pose, point = _check(pose, point)
    return ops.ba_projection(pose[None], point[None], _ZERO, _ZERO)[1][0]


def point_jacobian(pose, point):
    pose, point = _check(pose, point)
    return ops.ba_projection(pose[None], point[None], _ZERO, _ZERO)[2][0]


def exp_so3(rotvec):
    rotvec = np.asarray(rotvec)
    if rotvec.dtype != np.float64 or rotvec.ndim != 1:
        raise ValueError("Buffer dtype mismatch, expected a 1-D float64 array")
    return ops.ba_exp_so3(rotvec[None])[0]
