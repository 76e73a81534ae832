"""rust_bindings.semi_dense (src/py/semi_dense.rs:35-246): Frame, Params,
increment_age, propagate, update_depth, estimate_debug_ on the MI355X.

By default every call returns plain ndarrays, exactly as the reference's extension module
does (one download per returned map).  examples/semi_dense_vo.py:182-199 hands every map
one of these functions returns straight into the next call (and into PoseChangeEstimator
as depth map and 1 / variance); after `tadataka_amd.enable_device_maps()` the functions
return `tadataka_amd.ops.DeviceMap`s instead: array-likes that live in HBM and are
downloaded the first time somebody looks at them (np.asarray, indexing, arithmetic,
plotting ...).  Passed back in, they are used where they are; plain ndarrays are accepted
either way (and uploaded).  Frame.image is such a map too then, backed by the image the
frame keeps on the device."""
import numpy as np

from rust_bindings._check import f64, typed
from rust_bindings.camera import CameraParameters
from tadataka_amd import ops
from tadataka_amd._lib import TDK_ERR_NO_DEVICE, TdkError


LAZY_MAPS = False          # tadataka_amd.enable_device_maps()


def _out(*maps):
    if not LAZY_MAPS:
        maps = tuple(np.asarray(m) for m in maps)
    return maps[0] if len(maps) == 1 else maps


def _map_arg(a, dtype, name):
    """An ndarray of the exact dtype (rust-numpy's rule) or a DeviceMap of it."""
    if isinstance(a, ops.DeviceMap):
        if a.dtype != np.dtype(dtype):
            raise TypeError(f"{name} must have dtype {np.dtype(dtype).name}")
        return a
    return typed(a, dtype, 2, name)


def _camera(camera_params):
    """Duck-typed like camera_params_from_py (src/py/semi_dense.rs:19-33)."""
    f = camera_params.focal_length
    o = camera_params.offset
    f64(f, 1, "camera_params.focal_length"); f64(o, 1, "camera_params.offset")
    return np.array([f[0], f[1], o[0], o[1]])


class Frame(object):
    """Frame(camera_params, image, transform_wf); takes a snapshot of its inputs
    (src/py/semi_dense.rs:53-91 copies them).  The snapshot of the image is taken ON THE
    DEVICE: the constructor uploads it (tdk_frame) and keeps no host copy -- `image` downloads
    it when somebody asks.  (Without a GPU the frame keeps a host copy: it is a value type.)"""

    def __init__(self, camera_params, image, transform):
        self._cam = _camera(camera_params)
        img = f64(image, 2, "image")
        self._shape = img.shape
        self._transform = f64(transform, 2, "transform").copy()
        self._dev, self._host_image = None, None
        try:
            self._dev = ops.DeviceFrame(img)
        except TdkError as e:
            if e.status != TDK_ERR_NO_DEVICE:
                raise
            self._host_image = img.copy()

    @property
    def camera_params(self):
        return CameraParameters(self._cam[0:2], self._cam[2:4])

    def _image_copy(self):
        """A fresh host array of the image (what the reference's getter returns)."""
        if self._host_image is not None:
            return self._host_image.copy()
        return self._dev.download()

    @property
    def image(self):
        if not LAZY_MAPS or self._dev is None:
            return self._image_copy()
        # the frame's image where it lives on the device; the host side is fetched on first look
        return ops.DeviceMap(self._shape, np.float64, owner=self)

    def _device_image_ptr(self):
        return self._resident()[1].device_ptr()

    @property
    def transform_wf(self):
        return self._transform.copy()

    def _as_tuple(self):
        return (self._cam, self._image_copy(), self._transform)

    def _resident(self):
        """(camera, device-resident image, T_wf)."""
        if self._dev is None:
            self._dev = ops.DeviceFrame(self._host_image)      # raises without a GPU
        return (self._cam, self._dev, self._transform)


class Params(object):
    """Params(min_depth, max_depth, geo_coeff, photo_coeff, ref_step_size,
    min_gradient) (src/py/semi_dense.rs:93-108)."""

    def __init__(self, min_depth, max_depth, geo_coeff, photo_coeff, ref_step_size, min_gradient):
        self._c = ops.make_params(min_depth, max_depth, geo_coeff, photo_coeff, ref_step_size,
                                  min_gradient)


def increment_age(age_map0, camera_params0, camera_params1, transform10, depth_map0):
    _map_arg(age_map0, np.uint64, "age_map0"); f64(transform10, 2, "transform10")
    _map_arg(depth_map0, np.float64, "depth_map0")
    if age_map0.shape != depth_map0.shape:
        raise ValueError("age_map0 and depth_map0 must have the same shape")   # age.rs:13 assert
    return _out(ops.increment_age_maps(age_map0, _camera(camera_params0), _camera(camera_params1),
                                       transform10, depth_map0))


def propagate(transform10, camera_params0, camera_params1, depth_map0, variance_map0,
              default_depth, default_variance, uncertaintity_bias):
    f64(transform10, 2, "transform10"); _map_arg(depth_map0, np.float64, "depth_map0")
    _map_arg(variance_map0, np.float64, "variance_map0")
    return _out(*ops.propagate_maps(transform10, _camera(camera_params0), _camera(camera_params1),
                                    depth_map0, variance_map0, default_depth, default_variance,
                                    uncertaintity_bias))


def update_depth(keyframe, refframes, age_map, prior_depth, prior_variance, params):
    """Returns (depth, variance, flag) (src/py/semi_dense.rs:182-186)."""
    _map_arg(age_map, np.uint64, "age_map"); _map_arg(prior_depth, np.float64, "prior_depth")
    _map_arg(prior_variance, np.float64, "prior_variance")
    shape = age_map.shape
    if prior_depth.shape != shape or prior_variance.shape != shape or keyframe._shape != shape:
        raise ValueError("maps and keyframe image must share one shape")   # semi_dense.rs:168-173
    if any(r._shape != shape for r in refframes):
        raise ValueError("reference frames must have the key frame's shape")
    # frames and maps stay on the device: the example passes an ever longer refframes list
    # (examples/semi_dense_vo.py:199) and the maps of the previous call; nothing is uploaded again
    return _out(*ops.update_depth_maps(keyframe._resident(), [r._resident() for r in refframes], age_map,
                                       prior_depth, prior_variance, params._c))


def estimate_debug_(u_key, prior_depth, prior_variance, keyframe, refframe, params):
    """Single pixel; returns (depth, variance, flag) (src/py/semi_dense.rs:126-155)."""
    typed(u_key, np.int64, 1, "u_key")
    return ops.estimate_one(u_key, prior_depth, prior_variance, keyframe._as_tuple(),
                            refframe._as_tuple(), params._c)
