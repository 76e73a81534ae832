"""tadataka.metric (reference tadataka/metric.py:8-39): photometric error of a
relative pose.  The reference materialises the pixel grid, warps it, masks and
interpolates in separate passes; here one fused, error-only device pass over the
frame pair (tdk_dvo_photometric_error) returns the masked sum of squares and the count."""
import numpy as np

from tadataka_amd import ops


def calc_error_(v1, v2):
    d = v1 - v2
    return np.mean(d * d)


def _evaluate(batch, camera_model0, camera_model1, T10):
    ss, ne = batch.photometric_error(0, ops.camera_vec(camera_model0), ops.camera_vec(camera_model1),
                                     ops.pose12(T10[0:3, 0:3], T10[0:3, 3])[None])
    n = int(ne[0])
    if n == 0:
        return float("nan")            # np.mean of an empty selection
    return float(ss[0]) / n


def photometric_error(warp, gray_image0, depth_map0, gray_image1):
    """mean((I0[u0] - I1<warp(u0)>)^2) over the pixels that land inside image 1.
    `warp` is a LocalWarp2D (it carries the two camera models and T10)."""
    h, w = depth_map0.shape
    if hasattr(warp, "T10"):
        T10 = warp.T10
    elif hasattr(warp, "warp3d"):       # a Warp2D between two world poses: frame 0 -> world -> frame 1 as one transform
        T10 = np.linalg.inv(warp.warp3d.T_w1) @ warp.warp3d.T_w0
    else:
        raise TypeError("photometric_error needs a LocalWarp2D or a Warp2D (the fused device pass takes the cameras "
                        "and the relative pose, not a callable)")
    batch = ops.DvoBatch(1, h, w)
    try:
        batch.upload(0, gray_image0, depth_map0, gray_image1)
        return _evaluate(batch, warp.camera_model0, warp.camera_model1, T10)
    finally:
        batch.close()


class PhotometricError(object):
    """Frames stay resident on the device; each call is one fused evaluation."""
    def __init__(self, camera_model0, camera_model1, I0, D0, I1):
        self.camera_model0 = camera_model0
        self.camera_model1 = camera_model1
        self.I0, self.D0, self.I1 = I0, D0, I1
        self._batch = ops.DvoBatch(1, I0.shape[0], I0.shape[1])
        self._batch.upload(0, I0, D0, I1)

    def __call__(self, pose10):
        return _evaluate(self._batch, self.camera_model0, self.camera_model1, pose10.T)
