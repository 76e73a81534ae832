"""tadataka.vo.dvo (reference tadataka/vo/dvo/__init__.py:26-150): direct visual
odometry by Gauss-Newton on the photometric error (Kerl 2012).

Same classes, arguments and behaviour as the reference; the work is done by the
fused device kernels of libtadataka_hip.so:

  * _PoseChangeEstimator / PoseChangeEstimator keep (I0, D0, I1[, W0]) resident
    on the MI355X, build the pyramid there, and run the whole accept/reject
    loop on the device (tdk_dvo_estimate_level / tdk_dvo_estimate) -- for every
    weight option, including the Student-t / Tukey global statistics;
  * calc_pose_update works on the arrays the reference passes
    (tdk_dvo_pose_update) and returns the twist.

Note: as in the reference the residual I0 - I1 is NOT re-warped
between iterations; only the Jacobian moves with the pose.
"""
import warnings

import numpy as np

from tadataka.math import solve_normal_equations
from tadataka.pose import Pose
from tadataka.robust.weights import (compute_weights_huber, compute_weights_student_t,
                                     compute_weights_tukey)
from tadataka.vo.dvo.jacobian import calc_image_gradient, calc_jacobian  # noqa: F401
from tadataka_amd import ops


def calc_error(r, weights=None):
    if weights is None:
        return np.dot(r, r)
    return np.dot(r * weights, r)


def compute_weights(name, residuals):
    if name == "tukey":
        return compute_weights_tukey(residuals)
    if name == "student-t":
        return compute_weights_student_t(residuals)
    if name == "huber":
        return compute_weights_huber(residuals)
    raise ValueError(f"No such weights '{name}'")


def level_to_scale(level, layer_size_ratio):
    return 1 / pow(layer_size_ratio, level)


def _check_weights_name(weights):
    if isinstance(weights, str) and weights not in ("huber", "student-t", "tukey"):
        raise ValueError(f"No such weights '{weights}'")


def calc_pose_update(camera_model1, residuals, GX1, GY1, P1, weights):
    """One Gauss-Newton step xi = argmin ||sqrt(W)(J xi - r)|| for points P1
    (already in frame 1).  Returns None if no point projects into the image with
    positive depth."""
    assert(GX1.shape == GY1.shape)
    _check_weights_name(weights)
    cam1 = ops.camera_vec(camera_model1)
    if weights is None:
        H, b, n = ops.dvo_pose_update(cam1, residuals, GX1, GY1, P1, ops.W_NONE)
    elif isinstance(weights, str) and weights == "huber":
        H, b, n = ops.dvo_pose_update(cam1, residuals, GX1, GY1, P1, ops.W_HUBER)
    elif isinstance(weights, str):
        # Student-t / Tukey need statistics of the MASKED residuals first
        mask = _update_mask(cam1, P1, GX1.shape)
        if not np.any(mask):
            return None
        w = np.zeros(len(residuals))
        w[mask] = compute_weights(weights, np.asarray(residuals, dtype=np.float64)[mask])
        H, b, n = ops.dvo_pose_update(cam1, residuals, GX1, GY1, P1, ops.W_MAP, w)
    else:
        H, b, n = ops.dvo_pose_update(cam1, residuals, GX1, GY1, P1, ops.W_MAP,
                                      np.asarray(weights, dtype=np.float64).reshape(-1))
    if n == 0:
        return None
    return solve_normal_equations(H, b, n)


def _update_mask(cam1, P1, shape):
    """in_range(unnormalize(pi(P1))) & z > 0, evaluated by the device operators."""
    from tadataka.utils import is_in_image_range
    us1 = ops.unnormalize(ops.project_vecs(P1), cam1)
    return is_in_image_range(us1, shape) & (np.asarray(P1)[:, 2] > 0)


def _is_map(weights):
    return isinstance(weights, (np.ndarray, ops.DeviceMap))


def _fused_mode(weights):
    """Weight mode of the fused device loop."""
    if _is_map(weights):
        return ops.W_MAP
    return ops.WEIGHT_MODES[weights]      # None / "huber" / "student-t" / "tukey"


_IDENTITY12 = ops.pose12(np.eye(3), np.zeros(3))[None]


def _pose12(pose):
    if pose is None:
        return _IDENTITY12
    return ops.pose12(pose.R, pose.t)[None]


# The reference builds its pyramid with skimage.transform.rescale(image, scale) -- every level, level 0
# at scale 1.0 included (:144-148; scikit-image pinned to 0.16.2, setup.py:117).  PYRAMID selects what the
# device builds:
#   "skimage"  (default) what scikit-image returns on THIS interpreter, to the bit: anti-aliasing prefilter,
#              the affine map resize() estimates by SVD, scipy's Gaussian kernels, clip=True, level 0 through
#              rescale(., 1.0) -- tadataka_amd/rescale_plan.py makes skimage's own NumPy calls; pinned against
#              scikit-image 0.18.3 (tests/golden/skimage_*.npz).  PYRAMID_PLANS, if set, is a callable
#              (shape, n_levels, ratio) -> list of plans recorded on another interpreter (the fixtures').
#   "ideal"    the same pipeline with ideal constants (sample positions (i + 0.5) * factor - 0.5, libm
#              kernels, level 0 = the frame itself, no clip): platform-independent, within the spread that
#              two NumPy builds show between themselves, not bit-identical with any of them.
#   "bilinear" ideal constants without the prefilter (anti_aliasing=False).
import tadataka_amd
PYRAMID = "skimage" if tadataka_amd.PYRAMID_ANTI_ALIASING else "bilinear"
PYRAMID_PLANS = None


# Device batches are kept between calls (one per shape / pyramid / weight-map
# configuration, most recent two): creating one costs ~4 ms of allocations and a
# stream, an estimation of one 640x480 pair ~0.5 ms.  Not thread-safe, like the
# rest of this module.
_BATCHES = {}


def _batch_for(shape, n_levels, ratio, with_weight_map, pyramid=None):
    key = (int(shape[0]), int(shape[1]), int(n_levels), float(ratio), bool(with_weight_map), pyramid,
           PYRAMID_PLANS)
    batch = _BATCHES.pop(key, None)
    if batch is None:
        batch = ops.DvoBatch(1, key[0], key[1], n_levels=key[2], ratio=key[3], with_weight_map=key[4])
        if pyramid == "skimage":
            plans = PYRAMID_PLANS(shape, n_levels, ratio) if PYRAMID_PLANS is not None else None
            batch.set_skimage_pyramid(plans)
        else:
            batch.set_anti_aliasing(pyramid != "bilinear")
    _BATCHES[key] = batch          # most recently used last
    while len(_BATCHES) > 2:
        _BATCHES.pop(next(iter(_BATCHES))).close()
    return batch


class _PoseChangeEstimator(object):
    """Gauss-Newton at one resolution."""
    def __init__(self, camera_model0, camera_model1, max_iter):
        self.camera_model0 = camera_model0
        self.camera_model1 = camera_model1
        self.max_iter = max_iter

    def __call__(self, I0, D0, I1, pose10, weights=None):
        _check_weights_name(weights)
        has_map = _is_map(weights)
        batch = _batch_for(I0.shape, 1, 1.5, has_map)
        batch.upload(0, I0, D0, I1, weights if has_map else None)
        return _estimate_level(batch, 0, self.camera_model0, self.camera_model1, pose10,
                               weights, self.max_iter)


def _estimate_level(batch, level, camera_model0, camera_model1, pose10, weights, max_iter):
    cam0, cam1 = ops.camera_vec(camera_model0), ops.camera_vec(camera_model1)
    mode = _fused_mode(weights)
    P, n_evals = batch.estimate_level(level, cam0, cam1, _pose12(pose10), mode, max_iter)
    _warn_if_too_large(batch)
    return Pose.from_matrix(P[0])


def _warn_if_too_large(batch):
    # the reference warns whenever calc_pose_update returns None (:97-100): at any iteration, at
    # every pyramid level.  The device loop records that per pair.
    if batch.warnings().any():
        warnings.warn("Camera pose change is too large.", RuntimeWarning)


def _reject_integer_frames(**arrays):
    """Every array the reference hands to PoseChangeEstimator goes through skimage.transform.rescale.  For a float image
    that is what this port reproduces to the bit; for an INTEGER image skimage first runs the anti-aliasing prefilter in
    the image's own integer type (scipy.ndimage.gaussian_filter keeps the dtype: the filtered values are quantised) and
    only then scales to [0, 1] -- rescale(u8, s) != rescale(img_as_float(u8), s) for every s < 1 on scikit-image 0.18.3.
    That pipeline is not reproduced here, and taking the integers as floats (0 .. 255) would silently be another problem
    altogether, so integer frames are refused with the conversion the examples apply (examples/dvo_pose_change.py:22-31)."""
    for name, a in arrays.items():
        if a is None or isinstance(a, ops.DeviceMap):
            continue
        kind = np.asarray(a).dtype.kind
        if kind in "uib":
            raise TypeError(f"{name} has dtype {np.asarray(a).dtype}: pass float images (skimage.img_as_float / rgb2gray, as the "
                            "reference's examples do); the reference's handling of integer frames -- an integer-valued "
                            "prefilter inside skimage.transform.rescale -- is not reproduced by this port")


class PoseChangeEstimator(object):
    """Coarse-to-fine DVO: levels n_coarse_to_fine-1 ... 0 at scale
    1 / layer_size_ratio**level, each level starting from the previous result."""
    def __init__(self, camera_model0, camera_model1,
                 n_coarse_to_fine=5, max_iter=20, layer_size_ratio=1.5):
        self.n_coarse_to_fine = n_coarse_to_fine
        self.max_iter = max_iter
        self.layer_size_ratio = layer_size_ratio
        self.camera_model0 = camera_model0
        self.camera_model1 = camera_model1

    def __call__(self, I0, D0, I1, weights=None, pose10=None):
        assert(I0.shape == D0.shape == I1.shape)
        assert(np.ndim(I0) == 2)
        assert(np.ndim(D0) == 2)
        assert(np.ndim(I1) == 2)
        _check_weights_name(weights)
        has_map = _is_map(weights)
        _reject_integer_frames(I0=I0, D0=D0, I1=I1, weights=weights if has_map else None)
        batch = _batch_for(I0.shape, self.n_coarse_to_fine, self.layer_size_ratio, has_map, PYRAMID)
        batch.upload(0, I0, D0, I1, weights if has_map else None)
        batch.build_pyramid()
        cam0 = ops.camera_vec(self.camera_model0)
        cam1 = ops.camera_vec(self.camera_model1)
        P, _ = batch.estimate(cam0, cam1, _pose12(pose10), _fused_mode(weights), self.max_iter)
        _warn_if_too_large(batch)
        return Pose.from_matrix(P[0])
