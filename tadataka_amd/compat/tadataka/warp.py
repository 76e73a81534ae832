"""tadataka.warp (reference tadataka/warp.py:10-88): 2D <-> 3D warps between two
camera frames.  The per-point arithmetic runs on the device."""
import numpy as np

from rust_bindings import warp as _warp
from tadataka.decorator import allow_1d
from tadataka.matrix import calc_relative_transform
from tadataka.pose import Pose
from tadataka.projection import inv_pi, pi
from tadataka.rigid_transform import transform_se3


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def warp2d_(T10, xs0, depths0):
    """Normalized-plane warp with depth: rust_bindings.warp.warp_vecs."""
    return _warp.warp_vecs(_f64(T10), _f64(xs0), _f64(depths0))


def warp3d(T_w0, T_w1, P0):
    return transform_se3(calc_relative_transform(T_w1, T_w0), P0)


def warp2d(T_wa, T_wb, xs, depths):
    return pi(warp3d(T_wa, T_wb, inv_pi(xs, depths)))


class Warp3D(object):
    def __init__(self, pose_w0, pose_w1):
        assert(isinstance(pose_w0, Pose))
        assert(isinstance(pose_w1, Pose))
        self.T_w0 = pose_w0.T
        self.T_w1 = pose_w1.T

    @allow_1d(which_argument=1)
    def __call__(self, P):
        return warp3d(self.T_w0, self.T_w1, _f64(P))


def warp_depth(warp, xs0, depths0):
    P1 = warp(inv_pi(xs0, depths0))
    return pi(P1), P1[:, 2]


class Warp2D(object):
    """Image-plane warp between two world-posed cameras."""
    def __init__(self, camera_model0, camera_model1, pose_w0, pose_w1):
        self.camera_model0 = camera_model0
        self.camera_model1 = camera_model1
        self.warp3d = Warp3D(pose_w0, pose_w1)

    def __call__(self, us0, depths0):
        xs0 = self.camera_model0.normalize(us0)
        xs1, depths1 = warp_depth(self.warp3d, xs0, depths0)
        return self.camera_model1.unnormalize(xs1), depths1


class LocalWarp2D(object):
    """Image-plane warp for a relative pose pose10 (frame 0 -> frame 1)."""
    def __init__(self, camera_model0, camera_model1, pose10):
        self.camera_model0 = camera_model0
        self.camera_model1 = camera_model1
        self.T10 = pose10.T

    def __call__(self, us0, depths0):
        xs0 = self.camera_model0.normalize(us0)
        xs1, depths1 = warp2d_(self.T10, xs0, depths0)
        return self.camera_model1.unnormalize(xs1), depths1
