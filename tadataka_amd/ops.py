"""NumPy-facing wrappers over the C ABI (tadataka_amd/_lib.py).

Everything here runs on the MI355X through libtadataka_hip.so; there is no CPU
implementation behind these functions.
"""
import ctypes as C

import numpy as np

from tadataka_amd import _lib
from tadataka_amd._lib import (W_HUBER, W_MAP, W_NONE, W_STUDENT_T, W_TUKEY, SemiDenseParams,
                               c_double_p, c_int64_p, c_int_p, c_uint64_p, call)

WEIGHT_MODES = {None: W_NONE, "huber": W_HUBER, "student-t": W_STUDENT_T, "tukey": W_TUKEY}


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _p(a):
    return a.ctypes.data_as(c_double_p)


def camera_vec(camera):
    """Accepts (fx, fy, ox, oy) or an object with .focal_length/.offset (duck
    typing as src/py/semi_dense.rs:19-33)."""
    if hasattr(camera, "focal_length") and hasattr(camera, "offset"):
        return _f64(np.concatenate([np.asarray(camera.focal_length, dtype=np.float64),
                                    np.asarray(camera.offset, dtype=np.float64)]), (4,))
    if hasattr(camera, "camera_parameters"):
        return camera_vec(camera.camera_parameters)
    return _f64(camera, (4,))


# ---- parity-granular operators ---------------------------------------------
def normalize(keypoints, camera):
    kp = _f64(keypoints).reshape(-1, 2)
    out = np.empty_like(kp)
    cam = camera_vec(camera)
    call("tdk_normalize", _p(kp), kp.shape[0], _p(cam), _p(out))
    return out


def unnormalize(keypoints, camera):
    kp = _f64(keypoints).reshape(-1, 2)
    out = np.empty_like(kp)
    cam = camera_vec(camera)
    call("tdk_unnormalize", _p(kp), kp.shape[0], _p(cam), _p(out))
    return out


def project_vecs(points):
    P = _f64(points).reshape(-1, 3)
    out = np.empty((P.shape[0], 2))
    call("tdk_project_vecs", _p(P), P.shape[0], _p(out))
    return out


def inv_project_vecs(xs, depths):
    xs = _f64(xs).reshape(-1, 2)
    d = _f64(depths).reshape(-1)
    if d.shape[0] != xs.shape[0]:
        raise ValueError("xs and depths must have the same length")
    out = np.empty((xs.shape[0], 3))
    call("tdk_inv_project_vecs", _p(xs), _p(d), xs.shape[0], _p(out))
    return out


def transform(T, points):
    T = _f64(T, (4, 4))
    P = _f64(points).reshape(-1, 3)
    out = np.empty_like(P)
    call("tdk_transform", _p(T), _p(P), P.shape[0], _p(out))
    return out


def warp_vecs(T10, xs, depths):
    T10 = _f64(T10, (4, 4))
    xs = _f64(xs).reshape(-1, 2)
    d = _f64(depths).reshape(-1)
    if d.shape[0] != xs.shape[0]:
        raise ValueError("xs and depths must have the same length")
    oxs = np.empty_like(xs)
    od = np.empty_like(d)
    call("tdk_warp_vecs", _p(T10), _p(xs), _p(d), xs.shape[0], _p(oxs), _p(od))
    return oxs, od


def interpolation(image, coordinates):
    image = _f64(image)
    if image.ndim != 2:
        raise ValueError("Image have to be a two dimensional array")
    c = _f64(coordinates).reshape(-1, 2)
    out = np.empty(c.shape[0])
    call("tdk_interpolation", _p(image), image.shape[0], image.shape[1], _p(c), c.shape[0], _p(out))
    return out


def calc_depth0(T10, x0, x1):
    T10 = _f64(T10, (4, 4)); x0 = _f64(x0, (2,)); x1 = _f64(x1, (2,))
    out = C.c_double()
    call("tdk_calc_depth0", _p(T10), _p(x0), _p(x1), C.byref(out))
    return float(out.value)


def image_gradient(image):
    image = _f64(image)
    gx = np.empty_like(image); gy = np.empty_like(image)
    call("tdk_image_gradient", _p(image), image.shape[0], image.shape[1], _p(gx), _p(gy))
    return gx, gy


def rescale_shape(shape, scale):
    return (max(1, int(np.round(shape[0] * scale))), max(1, int(np.round(shape[1] * scale))))


def _plan_args(plan):
    m = _f64(plan["map"], (4,))
    wr = None if plan.get("wr") is None or len(plan["wr"]) == 0 else _f64(plan["wr"])
    wc = None if plan.get("wc") is None or len(plan["wc"]) == 0 else _f64(plan["wc"])
    return (m, wr, wc), (_p(m), None if wr is None else _p(wr), 0 if wr is None else len(wr) // 2,
                         None if wc is None else _p(wc), 0 if wc is None else len(wc) // 2)


def rescale(image, scale, anti_aliasing=False, mode="ideal", plan=None, clip=True):
    """skimage.transform.rescale(image, scale[, anti_aliasing]) of a 2-D float64 image.

    mode="skimage": what scikit-image returns on THIS interpreter, to the bit (the estimated affine map and
    scipy's Gaussian kernels from tadataka_amd.rescale_plan, clip=True) -- or on the interpreter a `plan`
    was recorded on.  mode="ideal" (the default of this low-level wrapper): ideal sample positions
    (i + 0.5) * factor - 0.5, libm kernels, no clip -- within ~1e-13 of the former."""
    return resize(image, rescale_shape(np.shape(image), scale), anti_aliasing, mode, plan, clip)


def resize(image, output_shape, anti_aliasing=False, mode="ideal", plan=None, clip=True):
    """skimage.transform.resize to an explicit (height, width); see rescale()."""
    image = _f64(image)
    Ho, Wo = int(output_shape[0]), int(output_shape[1])
    out = np.empty((Ho, Wo))
    if plan is None and mode == "ideal":
        call("tdk_rescale_anti_aliased" if anti_aliasing else "tdk_rescale", _p(image), image.shape[0],
             image.shape[1], _p(out), Ho, Wo)
        return out
    if plan is None:
        from . import rescale_plan
        plan = rescale_plan.resize_plan(image.shape, (Ho, Wo), anti_aliasing)
    elif not anti_aliasing:
        plan = dict(plan, wr=None, wc=None)
    keep, args = _plan_args(plan)
    call("tdk_rescale_skimage", _p(image), image.shape[0], image.shape[1], _p(out), Ho, Wo, *args, 1 if clip else 0)
    return out


LIBRARY_OPTIONS = {"pyramid_stream": 0, "sd_warp_gather": 1}


def set_option(name, value):
    """tdk_set_option (library-wide): "pyramid_stream" 1 (default) streaming kernel for batches that fill the chip /
    0 LDS tiles always / 2 streaming kernel always; "sd_warp_gather" 1 (default) gather kernels with the slot path as
    device-side fallback / 0 slot path for every track.  Bit-identical either way."""
    call("tdk_set_option", LIBRARY_OPTIONS[name], int(value))


# ---- DVO batch ----------------------------------------------------------------
def pose12(R, t):
    return np.concatenate([_f64(R, (9,)), _f64(t, (3,))])


class DvoBatch(object):
    """Device-resident batch of frame pairs (tdk_dvo)."""

    def __init__(self, n_pairs, height, width, n_levels=1, ratio=1.5, with_weight_map=False):
        self.n_pairs, self.height, self.width, self.n_levels = n_pairs, height, width, n_levels
        self.ratio = float(ratio)
        self.level0_mask = 0
        self.with_weight_map = bool(with_weight_map)
        self._h = C.c_void_p()
        call("tdk_dvo_create", n_pairs, height, width, n_levels, float(ratio),
             int(self.with_weight_map), C.byref(self._h))

    def close(self):
        if self._h:
            call("tdk_dvo_destroy", self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, pair, I0, D0, I1, weight_map=None):
        shape = (self.height, self.width)
        arrays = (I0, D0, I1, weight_map)
        if any(isinstance(a, DeviceMap) for a in arrays):
            # maps the previous mapping step left on the device are copied there
            host = (C.c_void_p * 4)(); dev = (C.c_void_p * 4)()
            keep = []
            for k, a in enumerate(arrays):
                if a is None:
                    continue
                if isinstance(a, DeviceMap):
                    if a.shape != shape or a.dtype != np.float64:
                        raise ValueError("map shape / dtype does not match the batch")
                    dev[k] = a.device_ptr()
                else:
                    h = _f64(a, shape)
                    keep.append(h)
                    host[k] = h.ctypes.data
            call("tdk_dvo_upload_mixed", self._h, pair, host, dev)
            return
        I0 = _f64(I0, shape); D0 = _f64(D0, shape); I1 = _f64(I1, shape)
        w = None if weight_map is None else _f64(weight_map, shape)
        call("tdk_dvo_upload", self._h, pair, _p(I0), _p(D0), _p(I1), None if w is None else _p(w))

    def upload_async(self, which, first_pair, n_pairs, pinned):
        """Queues the upload of one array of a range of pairs from a PinnedBuffer; does not wait."""
        fn = "tdk_dvo_upload_async_u8" if pinned.dtype == np.uint8 else "tdk_dvo_upload_async"
        call(fn, self._h, {"I0": 0, "D0": 1, "I1": 2, "W0": 3}[which], first_pair, n_pairs, pinned.ptr)

    def fill_synthetic(self, camera, poses12, seed0=0, noise=0.02):
        cam = camera_vec(camera)
        P = _f64(poses12, (self.n_pairs, 12))
        call("tdk_dvo_fill_synthetic", self._h, _p(cam), _p(P), C.c_uint64(seed0), float(noise))

    def set_anti_aliasing(self, enabled):
        """Levels WITHOUT a plan: anti-aliased (the default) or plain bilinear, at the ideal sample positions."""
        call("tdk_dvo_set_anti_aliasing", self._h, 1 if enabled else 0)

    def set_skimage_pyramid(self, plans=None, level0="all", clip=True, anti_aliasing=True):
        """Every level as skimage.transform.rescale returns it, to the bit: level 0 through rescale(., 1.0) like
        the reference (tadataka/vo/dvo/__init__.py:144-148), the estimated affine maps, scipy's kernels, clip=True.
        plans: None = this interpreter's (tadataka_amd.rescale_plan.level_plans), or a list of n_levels dicts
        {'map', 'wr', 'wc'} recorded elsewhere (a fixture).  level0: "all", or an iterable of "I0" / "D0" / "I1" /
        "W0" -- the arrays whose level 0 is built (the others keep the uploaded frame as level 0), or None."""
        from . import rescale_plan
        if plans is None:
            plans = rescale_plan.level_plans((self.height, self.width), self.n_levels, self.ratio, anti_aliasing)
        if len(plans) != self.n_levels:
            raise ValueError("one plan per pyramid level")
        for level, plan in enumerate(plans):
            if plan is None:
                call("tdk_dvo_set_level_plan", self._h, level, None, None, 0, None, 0)
                continue
            keep, args = _plan_args(plan if anti_aliasing else dict(plan, wr=None, wc=None))
            call("tdk_dvo_set_level_plan", self._h, level, *args)
        names = {"I0": 0, "D0": 1, "I1": 2, "W0": 3}
        if level0 == "all":
            mask = 15 if self.with_weight_map else 7
        else:
            mask = 0
            for name in (level0 or ()):
                mask |= 1 << names[name]
        call("tdk_dvo_set_rescale_options", self._h, mask, 1 if clip else 0)
        self.level0_mask = mask

    def set_ideal_pyramid(self):
        """Back to the C ABI's defaults: ideal sample positions, libm kernels, level 0 = the frame, no clip."""
        for level in range(self.n_levels):
            call("tdk_dvo_set_level_plan", self._h, level, None, None, 0, None, 0)
        call("tdk_dvo_set_rescale_options", self._h, 0, 0)
        self.level0_mask = 0

    def build_pyramid(self, arrays=None):
        """Levels 1 .. n_levels - 1 (and level 0 where set_skimage_pyramid asked for it) of every array, or of the
        named ones only: arrays = iterable of "I0", "D0", "I1", "W0" (a stream that replaces I1 per step rebuilds
        only that)."""
        if arrays is None:
            call("tdk_dvo_build_pyramid", self._h)
            return
        mask = 0
        for name in arrays:
            mask |= 1 << {"I0": 0, "D0": 1, "I1": 2, "W0": 3}[name]
        call("tdk_dvo_build_pyramid_arrays", self._h, mask)

    def level_shape(self, level):
        h, w = C.c_int(), C.c_int()
        call("tdk_dvo_level_shape", self._h, level, C.byref(h), C.byref(w))
        return h.value, w.value

    def download(self, pair, level, which):
        out = np.empty(self.level_shape(level))
        call("tdk_dvo_download", self._h, pair, level, {"I0": 0, "D0": 1, "I1": 2, "W0": 3}[which], _p(out))
        return out

    def _cams(self, camera):
        cam = np.asarray(camera, dtype=np.float64)
        if cam.ndim == 1:
            cam = np.tile(camera_vec(cam), (self.n_pairs, 1))
        return _f64(cam, (self.n_pairs, 4))

    def evaluate(self, level, camera0, camera1, poses12, weight_mode=W_NONE):
        """Returns dict(H [n,21], b [n,6], n_update [n], sum_sq [n], n_error [n])."""
        n = self.n_pairs
        c0, c1 = self._cams(camera0), self._cams(camera1)
        P = _f64(poses12, (n, 12))
        H = np.empty((n, 21)); b = np.empty((n, 6)); ss = np.empty(n)
        nu = np.empty(n, dtype=np.int64); ne = np.empty(n, dtype=np.int64)
        call("tdk_dvo_evaluate", self._h, level, _p(c0), _p(c1), _p(P), weight_mode, _p(H), _p(b),
             nu.ctypes.data_as(c_int64_p), _p(ss), ne.ctypes.data_as(c_int64_p))
        return dict(H=H, b=b, n_update=nu, sum_sq=ss, n_error=ne)

    def photometric_error(self, level, camera0, camera1, poses12):
        """Error-only pass: (sum_sq [n], n_error [n]) of PhotometricError at poses12."""
        n = self.n_pairs
        c0, c1 = self._cams(camera0), self._cams(camera1)
        P = _f64(poses12, (n, 12))
        ss = np.empty(n); ne = np.empty(n, dtype=np.int64)
        call("tdk_dvo_photometric_error", self._h, level, _p(c0), _p(c1), _p(P), _p(ss),
             ne.ctypes.data_as(c_int64_p))
        return ss, ne

    def estimate_level(self, level, camera0, camera1, poses12, weight_mode=W_NONE, max_iter=20):
        n = self.n_pairs
        c0, c1 = self._cams(camera0), self._cams(camera1)
        P = _f64(poses12, (n, 12)).copy()
        ne = np.zeros(n, dtype=np.int32)
        call("tdk_dvo_estimate_level", self._h, level, _p(c0), _p(c1), _p(P), weight_mode, max_iter,
             ne.ctypes.data_as(c_int_p))
        return P, ne

    def estimate(self, camera0, camera1, poses12, weight_mode=W_NONE, max_iter=20):
        """Coarse-to-fine over the pyramid built by build_pyramid()."""
        n = self.n_pairs
        c0, c1 = self._cams(camera0), self._cams(camera1)
        P = _f64(poses12, (n, 12)).copy()
        px = C.c_int64(0)
        call("tdk_dvo_estimate", self._h, _p(c0), _p(c1), _p(P), weight_mode, max_iter, C.byref(px))
        return P, int(px.value)

    def warnings(self):
        """bool[n_pairs]: the last estimate() / estimate_level() met an empty update mask
        ("Camera pose change is too large" in the reference)."""
        f = np.zeros(self.n_pairs, dtype=np.int32)
        call("tdk_dvo_get_warnings", self._h, f.ctypes.data_as(c_int_p))
        return f != 0

    def counts(self):
        """(error_pixels, update_pixels) of the last estimate() / estimate_level(): source pixels of
        every PhotometricError evaluation and of every calc_pose_update, summed over pairs and levels."""
        e = C.c_int64(); u = C.c_int64()
        call("tdk_dvo_get_counts", self._h, C.byref(e), C.byref(u))
        return int(e.value), int(u.value)

    def tukey_fallbacks(self):
        """Pairs whose Tukey medians were redone by the exact radix select since the batch was created."""
        v = C.c_int64()
        call("tdk_dvo_get_tukey_fallbacks", self._h, C.byref(v))
        return int(v.value)

    OPTIONS = {"chain": 0, "tukey": 1}

    def set_option(self, name, value):
        """tdk_dvo_set_option: "chain" 1 (default) the whole coarse-to-fine chain of a small batch is queued at once /
        0 the host drives it level by level; "tukey" 0 (default) sampled brackets / 1 radix select / 2 brackets with the
        exact fallback forced.  Same results either way."""
        call("tdk_dvo_set_option", self._h, self.OPTIONS[name], int(value))

    def set_student_passes(self, mode):
        """0: Taylor passes (default); 1: nine sequential passes; 2: nine passes with IEEE divisions."""
        call("tdk_dvo_set_student_passes", self._h, int(mode))

    def robust_scale(self):
        """Student-t variance / Tukey c * MAD of each pair's last robust evaluation."""
        out = np.empty(self.n_pairs, dtype=np.float64)
        call("tdk_dvo_get_robust_scale", self._h, _p(out))
        return out

    def student_redos(self):
        """Pairs whose Student-t variance needed a third Taylor pass since the batch was created."""
        v = C.c_int64()
        call("tdk_dvo_get_student_redos", self._h, C.byref(v))
        return int(v.value)

    def student_fallbacks(self):
        """Pairs whose third Taylor pass had not converged either and that took the nine sequential steps."""
        v = C.c_int64()
        call("tdk_dvo_get_student_fallbacks", self._h, C.byref(v))
        return int(v.value)

    def set_profiling(self, enabled, all_levels=False):
        """HIP events around the evaluation launches of the finest level (all_levels: of every level)."""
        call("tdk_dvo_set_profiling", self._h, 2 if (enabled and all_levels) else int(bool(enabled)))

    def get_profile(self, kind="full", level=0):
        """Evaluation launches of a level since set_profiling(True): kind 'full' (only full
        evaluations), 'probe' (only error-only probes of candidates), 'mixed'."""
        n = C.c_int64(); ms = C.c_double(); px = C.c_int64()
        call("tdk_dvo_get_profile_level", self._h, int(level), {"full": 0, "probe": 1, "mixed": 2}[kind], C.byref(n),
             C.byref(ms), C.byref(px))
        return dict(launches=int(n.value), total_ms=float(ms.value), pixels=int(px.value))


class PinnedBuffer(object):
    """Page-locked host memory as a float64 or uint8 ndarray (tdk_pinned_alloc): the source of
    DvoBatch.upload_async."""

    def __init__(self, shape, dtype=np.float64):
        self.shape = tuple(int(v) for v in shape)
        self.dtype = np.dtype(dtype)
        assert self.dtype in (np.dtype(np.float64), np.dtype(np.uint8))
        n = int(np.prod(self.shape))
        self.ptr = C.c_void_p()
        call("tdk_pinned_alloc", C.c_size_t(n * self.dtype.itemsize), C.byref(self.ptr))
        ctype = c_double_p if self.dtype == np.float64 else C.POINTER(C.c_uint8)
        self.array = np.ctypeslib.as_array(C.cast(self.ptr, ctype), shape=(n,)).reshape(self.shape)

    def close(self):
        if self.ptr:
            self.array = None
            call("tdk_pinned_free", self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def upper21_to_matrix(H21):
    H = np.zeros((6, 6))
    H[np.triu_indices(6)] = H21
    return H + H.T - np.diag(np.diag(H))


# ---- least-squares pieces -----------------------------------------------------------
def weighted_normal_equations(A, b, w=None):
    """(A^T W A [p,p], A^T W b [p]) reduced on the device; p <= 8."""
    A = _f64(A); b = _f64(b).reshape(-1)
    if A.ndim != 2 or A.shape[0] != b.shape[0]:
        raise ValueError("A must be [n, p] and b [n]")
    n, p = A.shape
    wv = None if w is None else _f64(w, (n,))
    tri = np.empty(p * (p + 1) // 2); g = np.empty(p)
    call("tdk_weighted_normal_equations", _p(A), _p(b), None if wv is None else _p(wv), n, p,
         _p(tri), _p(g))
    M = np.zeros((p, p))
    M[np.triu_indices(p)] = tri
    return M + M.T - np.diag(np.diag(M)), g


def dvo_pose_update(camera1, residuals, GX1, GY1, P1, weight_mode=W_NONE, weights=None):
    """Normal equations of calc_pose_update on explicit arrays: (H [6,6], b [6], n_valid)."""
    cam = camera_vec(camera1)
    GX1 = _f64(GX1); GY1 = _f64(GY1, GX1.shape)
    P1 = _f64(P1).reshape(-1, 3)
    n = P1.shape[0]
    r = _f64(residuals, (n,))
    wv = None if weights is None else _f64(weights, (n,))
    H21 = np.empty(21); b = np.empty(6); nv = C.c_int64()
    call("tdk_dvo_pose_update", _p(cam), _p(r), _p(GX1), _p(GY1), GX1.shape[0], GX1.shape[1], _p(P1), n,
         weight_mode, None if wv is None else _p(wv), _p(H21), _p(b), C.byref(nv))
    return upper21_to_matrix(H21), b, int(nv.value)


def robust_weights(residuals, mode, p0=None, p1=None):
    """compute_weights_{huber,student_t,tukey}; (p0, p1) = (k, -) | (nu, n_iter) | (beta, c),
    None = the reference defaults."""
    r = _f64(residuals).reshape(-1)
    w = np.empty_like(r)
    if p0 is None and p1 is None:
        call("tdk_robust_weights", _p(r), r.shape[0], mode, _p(w))
    else:
        d0, d1 = {W_HUBER: (1.345, 0.0), W_STUDENT_T: (5.0, 10.0), W_TUKEY: (4.6851, 1.4826)}[mode]
        call("tdk_robust_weights_ex", _p(r), r.shape[0], mode, float(d0 if p0 is None else p0),
             float(d1 if p1 is None else p1), _p(w))
    return w


# ---- semi-dense -------------------------------------------------------------------
def make_params(min_depth, max_depth, geo_coeff, photo_coeff, ref_step_size, min_gradient):
    return SemiDenseParams(float(min_depth), float(max_depth), float(geo_coeff), float(photo_coeff),
                           float(ref_step_size), float(min_gradient))


def sobel(image):
    image = _f64(image)
    gx = np.empty_like(image); gy = np.empty_like(image)
    call("tdk_sobel", _p(image), image.shape[0], image.shape[1], _p(gx), _p(gy))
    return gx, gy


def increment_age(age0, camera0, camera1, T10, depth0):
    age0 = np.ascontiguousarray(age0, dtype=np.uint64)
    H, W = age0.shape
    c0, c1 = camera_vec(camera0), camera_vec(camera1)
    T10 = _f64(T10, (4, 4)); d0 = _f64(depth0, (H, W))
    age1 = np.empty_like(age0)
    call("tdk_increment_age", age0.ctypes.data_as(c_uint64_p), H, W, _p(c0), _p(c1), _p(T10), _p(d0),
         age1.ctypes.data_as(c_uint64_p))
    return age1


def propagate(T10, camera0, camera1, depth0, variance0, default_depth, default_variance,
              uncertaintity_bias):
    d0 = _f64(depth0)
    H, W = d0.shape
    v0 = _f64(variance0, (H, W))
    c0, c1 = camera_vec(camera0), camera_vec(camera1)
    T10 = _f64(T10, (4, 4))
    d1 = np.empty_like(d0); v1 = np.empty_like(d0)
    call("tdk_propagate", _p(T10), _p(c0), _p(c1), _p(d0), _p(v0), H, W, float(default_depth),
         float(default_variance), float(uncertaintity_bias), _p(d1), _p(v1))
    return d1, v1


def update_depth(key, refs, age, prior_depth, prior_variance, params):
    """key = (camera, image, T_wf); refs = list of the same.  Returns
    (depth, variance, flag) in the order of src/py/semi_dense.rs:182-186."""
    kc = camera_vec(key[0]); ki = _f64(key[1]); kT = _f64(key[2], (4, 4))
    H, W = ki.shape
    n_ref = len(refs)
    rc = _f64([camera_vec(r[0]) for r in refs] if n_ref else np.zeros((0, 4)), (n_ref, 4))
    ri = _f64([r[1] for r in refs] if n_ref else np.zeros((0, H, W)), (n_ref, H, W))
    rT = _f64([r[2] for r in refs] if n_ref else np.zeros((0, 4, 4)), (n_ref, 4, 4))
    age = np.ascontiguousarray(age, dtype=np.uint64).reshape(H, W)
    pd_ = _f64(prior_depth, (H, W)); pv = _f64(prior_variance, (H, W))
    depth = np.empty((H, W)); var = np.empty((H, W)); flag = np.empty((H, W), dtype=np.int64)
    call("tdk_update_depth", _p(kc), _p(ki), _p(kT), n_ref, _p(rc), _p(ri), _p(rT),
         age.ctypes.data_as(c_uint64_p), _p(pd_), _p(pv), H, W, C.byref(params), _p(depth), _p(var),
         flag.ctypes.data_as(c_int64_p))
    return depth, var, flag


class DeviceFrame(object):
    """An image resident on the device (tdk_frame): what rust_bindings.semi_dense.Frame holds."""

    def __init__(self, image):
        img = _f64(image)
        self.shape = img.shape
        self._h = C.c_void_p()
        call("tdk_frame_create", _p(img), img.shape[0], img.shape[1], C.byref(self._h))

    def device_ptr(self):
        p = C.c_void_p()
        call("tdk_frame_device_ptr", self._h, C.byref(p))
        return p.value

    def download(self):
        out = np.empty(self.shape)
        call("tdk_frame_download", self._h, _p(out))
        return out

    def close(self):
        if self._h:
            call("tdk_frame_destroy", self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def update_depth_frames(key, refs, age, prior_depth, prior_variance, params):
    """update_depth with device-resident frames: key = (camera, DeviceFrame, T_wf), refs = list of the
    same.  Returns (depth, variance, flag)."""
    kc = camera_vec(key[0]); kT = _f64(key[2], (4, 4))
    H, W = key[1].shape
    n_ref = len(refs)
    rc = _f64([camera_vec(r[0]) for r in refs] if n_ref else np.zeros((0, 4)), (n_ref, 4))
    rT = _f64([r[2] for r in refs] if n_ref else np.zeros((0, 4, 4)), (n_ref, 4, 4))
    handles = (C.c_void_p * max(n_ref, 1))(*[r[1]._h for r in refs])
    age = np.ascontiguousarray(age, dtype=np.uint64).reshape(H, W)
    pd_ = _f64(prior_depth, (H, W)); pv = _f64(prior_variance, (H, W))
    depth = np.empty((H, W)); var = np.empty((H, W)); flag = np.empty((H, W), dtype=np.int64)
    call("tdk_update_depth_frames", _p(kc), key[1]._h, _p(kT), n_ref, _p(rc), handles, _p(rT),
         age.ctypes.data_as(c_uint64_p), _p(pd_), _p(pv), C.byref(params), _p(depth), _p(var),
         flag.ctypes.data_as(c_int64_p))
    return depth, var, flag


# ndarray attributes that cannot hand out a writable alias of the buffer (they return scalars, tuples or
# fresh arrays): reading them leaves the device copy valid.  Everything else -- views (T, flat, real,
# reshape, ravel, squeeze, view, ...), mutating methods (fill, sort, put, itemset, partition, ...), ctypes /
# __array_interface__ -- counts as a writable reference that escaped.
# (astype(copy=False), conj / conjugate of a real array return the array ITSELF; clip, round, choose, dot, cumsum,
# cumprod, take accept an `out` positionally: none of them is in the list)
_MAP_PURE_ATTRS = frozenset((
    "all", "any", "argmax", "argmin", "argsort", "compress", "copy", "dump", "dumps", "flatten", "item", "itemsize",
    "max", "mean", "min", "nonzero", "prod", "ptp", "repeat", "searchsorted", "std", "strides", "sum", "tobytes",
    "tofile", "tolist", "tostring", "trace", "var"))


# NumPy functions that neither write into their array arguments nor return views of them (unless given
# out=): running them on the host copy leaves the device copy valid.  Anything not listed is treated as if
# it could (np.copyto, np.putmask, np.reshape, np.transpose, np.ravel, np.split, ...).
_MAP_PURE_FUNCS = frozenset((
    "all", "allclose", "amax", "amin", "any", "argmax", "argmin", "argsort", "argwhere", "array_equal",
    "array_equiv", "average", "bincount", "clip", "concatenate", "copy", "count_nonzero", "cumsum", "diff",
    "dot", "flatnonzero", "histogram", "isclose", "max", "mean", "median", "min", "nanmax", "nanmean",
    "nanmedian", "nanmin", "nansum", "nonzero", "percentile", "prod", "ptp", "quantile", "round", "sort",
    "stack", "std", "sum", "unique", "var", "where", "zeros_like", "ones_like", "empty_like", "full_like"))


def _host_args(x, pure, top=False):
    """DeviceMaps -> their host arrays inside the positional / keyword arguments of a NumPy function.  Of the
    positional arguments of a `pure` function only the FIRST is read-only for certain: several of them accept an
    `out` array positionally (np.clip(a, lo, hi, out), np.round(a, decimals, out), np.sum(a, axis, dtype, out))."""
    if isinstance(x, DeviceMap):
        return x._materialise() if pure else x._expose()
    if isinstance(x, (list, tuple)):
        if top:
            return type(x)(_host_args(y, pure and k == 0) for k, y in enumerate(x))
        return type(x)(_host_args(y, pure) for y in x)
    if isinstance(x, dict):
        return {k: _host_args(v, pure and k != "out") for k, v in x.items()}
    return x


class DeviceMap(np.lib.mixins.NDArrayOperatorsMixin):
    """An H x W map that lives on the device (tdk_map) and is downloaded when somebody looks at it.

    It is what rust_bindings.semi_dense.increment_age / propagate / update_depth return and accept:
    the loop of examples/semi_dense_vo.py:182-199 hands every map it gets straight into the next
    call, so nothing crosses PCIe there.  Towards NumPy it behaves as an array-like: np.asarray(m),
    m[...], arithmetic and ufuncs, NumPy functions (np.copyto, np.where, np.stack, ...), every ndarray
    attribute (m.shape, m.copy(), m.max(), m.fill(0), ...) work and materialise the host copy once.
    It is not an ndarray subclass: an ndarray's memory can be read by C code without any hook that
    could wait for the download.

    Host and device copy cannot diverge: writes through the map (m[mask] = v, ufunc out=, m.fill)
    and every *writable reference to the host copy that leaves the object* -- np.asarray(m), a slice
    m[a:b], m.T, iteration, a bound method -- mark the map `escaped`; an escaped map's host copy is
    sent to the device again before EVERY later device use (whether the reference is still alive is not
    something a Python object can know portably: no reference counting is consulted), so
    `np.asarray(depth_map)[mask] = 0` followed by update_depth behaves as it does with the ndarrays
    the reference returns.  Read-only looks (np.array_equal, m.max(), arithmetic, m.copy()) never escape a map.  A map that borrows a frame's image (Frame.image) becomes a map of its own
    at that point; the frame keeps its image, as the reference's getter returns a copy."""

    __array_priority__ = 100.0

    def __init__(self, shape, dtype, host=None, owner=None):
        self.shape = (int(shape[0]), int(shape[1]))
        self.dtype = np.dtype(dtype)
        assert self.dtype.itemsize == 8
        self._host = None            # materialised host copy
        self._escaped = False        # the host copy was written to, or a writable reference to it is out
        self._owner = owner          # a Frame whose device image this map borrows (_device_image_ptr())
        self._h = C.c_void_p()
        self._ptr = None
        if owner is None:
            src = None
            if host is not None:
                # uploaded synchronously and not kept: the caller's array stays the caller's (a later
                # write to it must not reach this map), and a look at the map downloads it
                host = np.ascontiguousarray(host, dtype=self.dtype).reshape(self.shape)
                src = host.ctypes.data_as(C.c_void_p)
            call("tdk_map_create", self.shape[0], self.shape[1], src, C.byref(self._h))
        elif host is not None:
            self._host = host

    # -- device side ---------------------------------------------------------------------------
    @classmethod
    def empty(cls, shape, dtype):
        return cls(shape, dtype)

    @classmethod
    def of(cls, a, dtype):
        """`a` as a DeviceMap of `dtype`: itself if it is one, uploaded if it is an ndarray."""
        if isinstance(a, DeviceMap):
            if a.dtype != np.dtype(dtype):
                raise TypeError(f"map has dtype {a.dtype}, expected {np.dtype(dtype)}")
            return a            # handle() / device_ptr() bring the device copy up to date
        return cls(np.shape(a), dtype, host=a)

    def handle(self):
        self._refresh_device()
        if self._owner is not None:
            raise TypeError("a map that borrows a frame's image has no tdk_map handle")
        return self._h

    def device_ptr(self):
        self._refresh_device()
        if self._owner is not None:
            return self._owner._device_image_ptr()
        if self._ptr is None:
            p = C.c_void_p()
            call("tdk_map_device_ptr", self._h, C.byref(p))
            self._ptr = p.value
        return self._ptr

    def _refresh_device(self):
        """Before every device use: if the host copy may have been written to, it goes up again."""
        if not self._escaped:
            return
        if self._owner is not None:
            # the frame's image is not ours to change: from here on this is a map of its own
            h = C.c_void_p()
            call("tdk_map_create", self.shape[0], self.shape[1], self._host.ctypes.data_as(C.c_void_p), C.byref(h))
            self._h, self._owner, self._ptr = h, None, None
        else:
            call("tdk_map_upload", self._h, self._host.ctypes.data_as(C.c_void_p))

    def safe_invert(self, epsilon=1e-16):
        """1 / (self + epsilon) on the device (tadataka.numeric.safe_invert)."""
        out = DeviceMap(self.shape, np.float64)
        call("tdk_map_safe_invert", self.handle(), float(epsilon), out._h)
        return out

    def close(self):
        if self._h:
            call("tdk_map_destroy", self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __copy__(self):
        """A map of its own (a shallow copy of the attributes would destroy one handle twice)."""
        return DeviceMap(self.shape, self.dtype, host=self._materialise())

    def __deepcopy__(self, memo):
        return self.__copy__()

    def __reduce__(self):
        return (np.array, (self._materialise(),))       # pickles as the ndarray it stands for

    # -- host side -----------------------------------------------------------------------------
    def _materialise(self):
        """The host copy, for reading inside this module (the reference does not leave it)."""
        if self._host is None:
            if self._owner is not None:
                self._host = self._owner._image_copy()
            else:
                out = np.empty(self.shape, dtype=self.dtype)
                call("tdk_map_download", self._h, out.ctypes.data_as(C.c_void_p))
                self._host = out
        return self._host

    def _expose(self):
        """The host copy as a writable array that leaves the object (or is written to here)."""
        a = self._materialise()
        self._escaped = True
        return a

    _writable = _expose

    def __array__(self, dtype=None, copy=None):
        if dtype is not None and np.dtype(dtype) != self.dtype:
            return self._materialise().astype(dtype)
        if copy:
            return self._materialise().copy()
        return self._expose()

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        inputs = tuple(x._materialise() if isinstance(x, DeviceMap) else x for x in inputs)
        if "out" in kwargs:
            kwargs["out"] = tuple(x._expose() if isinstance(x, DeviceMap) else x for x in kwargs["out"])
        return getattr(ufunc, method)(*inputs, **kwargs)

    def __array_function__(self, func, types, args, kwargs):
        """np.copyto(m, ...), np.putmask(m, ...), np.where(m > 0, ...), np.stack([m1, m2]), ...: the
        function runs on the host copies; it may write into them or return views of them."""
        name = getattr(func, "__name__", "")
        if name in ("ndim", "shape", "size") and len(args) == 1 and not kwargs and isinstance(args[0], DeviceMap):
            return getattr(args[0], name)               # no reason to fetch the data for these
        pure = name in _MAP_PURE_FUNCS
        return func(*_host_args(args, pure, top=True), **_host_args(kwargs, pure))

    ndim = 2

    @property
    def size(self):
        return self.shape[0] * self.shape[1]

    @property
    def nbytes(self):
        return self.size * 8

    def __len__(self):
        return self.shape[0]

    def __iter__(self):
        return iter(self._expose())

    def __getitem__(self, key):
        r = self._materialise()[key]
        if isinstance(r, np.ndarray) and r.base is not None:        # a view: writes to it land in the host copy
            self._escaped = True
        return r

    def __setitem__(self, key, value):
        self._expose()[key] = value._materialise() if isinstance(value, DeviceMap) else value

    def __getattr__(self, name):           # everything else an ndarray has (copy, max, flatten, T, fill, ...)
        if name.startswith("_"):
            raise AttributeError(name)
        host = self._materialise() if name in _MAP_PURE_ATTRS else self._expose()
        return getattr(host, name)

    def __repr__(self):
        state = "host copy present" if self._host is not None else "on the device"
        return f"DeviceMap(shape={self.shape}, dtype={self.dtype}, {state})"


def increment_age_maps(age0, camera0, camera1, T10, depth0):
    """increment_age with device-resident maps in and out (tdk_increment_age_maps)."""
    a0 = DeviceMap.of(age0, np.uint64); d0 = DeviceMap.of(depth0, np.float64)
    if a0.shape != d0.shape:
        raise ValueError("age_map0 and depth_map0 must have the same shape")
    c0, c1 = camera_vec(camera0), camera_vec(camera1)
    T10 = _f64(T10, (4, 4))
    a1 = DeviceMap.empty(a0.shape, np.uint64)
    call("tdk_increment_age_maps", a0.handle(), _p(c0), _p(c1), _p(T10), d0.handle(), a1._h)
    return a1


def propagate_maps(T10, camera0, camera1, depth0, variance0, default_depth, default_variance,
                   uncertaintity_bias):
    d0 = DeviceMap.of(depth0, np.float64); v0 = DeviceMap.of(variance0, np.float64)
    if d0.shape != v0.shape:
        raise ValueError("depth_map0 and variance_map0 must have the same shape")
    c0, c1 = camera_vec(camera0), camera_vec(camera1)
    T10 = _f64(T10, (4, 4))
    d1 = DeviceMap.empty(d0.shape, np.float64); v1 = DeviceMap.empty(d0.shape, np.float64)
    call("tdk_propagate_maps", _p(T10), _p(c0), _p(c1), d0.handle(), v0.handle(), float(default_depth),
         float(default_variance), float(uncertaintity_bias), d1._h, v1._h)
    return d1, v1


def update_depth_maps(key, refs, age, prior_depth, prior_variance, params):
    """update_depth with device-resident frames AND maps: key = (camera, DeviceFrame, T_wf), refs =
    list of the same.  Returns (depth, variance, flag) as DeviceMaps."""
    kc = camera_vec(key[0]); kT = _f64(key[2], (4, 4))
    shape = key[1].shape
    n_ref = len(refs)
    rc = _f64([camera_vec(r[0]) for r in refs] if n_ref else np.zeros((0, 4)), (n_ref, 4))
    rT = _f64([r[2] for r in refs] if n_ref else np.zeros((0, 4, 4)), (n_ref, 4, 4))
    handles = (C.c_void_p * max(n_ref, 1))(*[r[1]._h for r in refs])
    a = DeviceMap.of(age, np.uint64); pd_ = DeviceMap.of(prior_depth, np.float64)
    pv = DeviceMap.of(prior_variance, np.float64)
    depth = DeviceMap.empty(shape, np.float64); var = DeviceMap.empty(shape, np.float64)
    flag = DeviceMap.empty(shape, np.int64)
    call("tdk_update_depth_maps", _p(kc), key[1]._h, _p(kT), n_ref, _p(rc), handles, _p(rT), a.handle(),
         pd_.handle(), pv.handle(), C.byref(params), depth._h, var._h, flag._h)
    return depth, var, flag


def estimate_one(u_key, prior_depth, prior_variance, key, ref, params):
    u = np.ascontiguousarray(u_key, dtype=np.int64).reshape(2)
    kc = camera_vec(key[0]); ki = _f64(key[1]); kT = _f64(key[2], (4, 4))
    rc = camera_vec(ref[0]); ri = _f64(ref[1]); rT = _f64(ref[2], (4, 4))
    H, W = ki.shape
    d = C.c_double(); v = C.c_double(); f = C.c_int64()
    call("tdk_estimate_one", u.ctypes.data_as(c_int64_p), float(prior_depth), float(prior_variance),
         _p(kc), _p(ki), _p(kT), _p(rc), _p(ri), _p(rT), H, W, C.byref(params), C.byref(d), C.byref(v),
         C.byref(f))
    return float(d.value), float(v.value), int(f.value)


def regularize(depth_map, variance_map, flag_map):
    """regularize (src/semi_dense/regularization.rs:29-64): regularized DEPTH map."""
    d = _f64(depth_map)
    H, W = d.shape
    v = _f64(variance_map, (H, W))
    f = np.ascontiguousarray(flag_map, dtype=np.int64).reshape(H, W)
    out = np.empty_like(d)
    call("tdk_regularize", _p(d), _p(v), f.ctypes.data_as(c_int64_p), H, W, _p(out))
    return out


def fusion_arrays(mu1, mu2, var1, var2):
    """fusion_arrays (src/semi_dense/fusion.rs:13-42): (mu, var), elementwise."""
    m1 = _f64(mu1)
    m2 = _f64(mu2, m1.shape); v1 = _f64(var1, m1.shape); v2 = _f64(var2, m1.shape)
    mu = np.empty_like(m1); var = np.empty_like(m1)
    call("tdk_fusion_arrays", _p(m1), _p(m2), _p(v1), _p(v2), m1.size, _p(mu), _p(var))
    return mu, var


def rgb2gray(image):
    """skimage.color.rgb2gray as the examples call it: [H,W,3|4] float or uint8 -> [H,W] float64;
    two-dimensional input is returned as float64 with its VALUES unchanged (scikit-image 0.16.2 passes
    grey images through)."""
    a = np.asarray(image)
    if a.ndim == 2:
        return _f64(a)
    if a.ndim != 3 or a.shape[2] not in (3, 4):
        raise ValueError("the input array must have a shape == (.., ..,[ ..,] 3)), got " + str(a.shape))
    H, W, ch = a.shape
    out = np.empty((H, W))
    if a.dtype == np.uint8:
        a = np.ascontiguousarray(a)
        call("tdk_rgb2gray_u8", a.ctypes.data_as(C.POINTER(C.c_uint8)), H, W, ch, _p(out))
    else:
        a = _f64(a)
        call("tdk_rgb2gray", _p(a), H, W, ch, _p(out))
    return out


class SemiDenseSession(object):
    """Device-resident semi-dense mapping session over a batch of tracks (tdk_sd):
    per step increment_age -> propagate -> update_depth for every track, the maps
    and the frame rings staying in HBM (examples/semi_dense_vo.py:182-199)."""

    def __init__(self, n_tracks, height, width, max_refframes=4):
        self.n_tracks, self.height, self.width = int(n_tracks), int(height), int(width)
        self.max_refframes = int(max_refframes)
        self._h = C.c_void_p()
        call("tdk_sd_create", self.n_tracks, self.height, self.width, self.max_refframes, C.byref(self._h))

    def close(self):
        if self._h:
            call("tdk_sd_destroy", self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_age_policy(self, saturate):
        """False (default, the reference's rule): ages grow without bound as in
        src/semi_dense/age.rs:28 and a step whose ages exceed the ring raises
        TdkError(TDK_ERR_AGE_EXCEEDS_REFFRAMES) -- for good, so size the ring for the track.
        True (opt-in): ages saturate at the ring size (a pixel tracked for longer than
        max_refframes steps keeps using the oldest retained frame; identical to the reference for
        tracks of at most max_refframes + 1 frames, different beyond)."""
        call("tdk_sd_set_age_policy", self._h, int(bool(saturate)))

    def warp_fallbacks(self):
        """(track, step) forward warps of this session that took the slot path instead of the gather."""
        v = C.c_int64()
        call("tdk_sd_get_warp_fallbacks", self._h, C.byref(v))
        return int(v.value)

    def set_params(self, params, default_depth, default_variance, uncertaintity_bias):
        call("tdk_sd_set_params", self._h, C.byref(params), float(default_depth), float(default_variance),
             float(uncertaintity_bias))

    def set_maps(self, track, depth=None, variance=None, age=None):
        shape = (self.height, self.width)
        d = None if depth is None else _f64(depth, shape)
        v = None if variance is None else _f64(variance, shape)
        a = None if age is None else np.ascontiguousarray(age, dtype=np.uint64).reshape(shape)
        call("tdk_sd_set_maps", self._h, track, None if d is None else _p(d), None if v is None else _p(v),
             None if a is None else a.ctypes.data_as(c_uint64_p))

    def _read(self, fn, track, with_flag):
        shape = (self.height, self.width)
        d = np.empty(shape); v = np.empty(shape); a = np.empty(shape, dtype=np.uint64)
        f = np.empty(shape, dtype=np.int64) if with_flag else None
        call(fn, self._h, track, _p(d), _p(v), a.ctypes.data_as(c_uint64_p),
             None if f is None else f.ctypes.data_as(c_int64_p))
        return (d, v, a, f) if with_flag else (d, v, a)

    def get_maps(self, track, with_flag=False):
        """(depth, variance, age[, flag]) of the session's current state."""
        return self._read("tdk_sd_get_maps", track, with_flag)

    def get_results(self, track, with_flag=True):
        """(depth, variance, age[, flag]) of the last call, committed or not."""
        return self._read("tdk_sd_get_results", track, with_flag)

    def push_frame(self, track, camera, image, transform_wf=None):
        cam = camera_vec(camera)
        img = _f64(image, (self.height, self.width))
        T = None if transform_wf is None else _f64(transform_wf, (4, 4))
        call("tdk_sd_push_frame", self._h, track, _p(cam), _p(img), None if T is None else _p(T))

    def step(self, transforms10, key_transforms_wf=None, commit=True, histogram=False):
        """One mapping step for every track.  Returns the per-track flag histogram
        [n_tracks, 10] (flags 0, -1, ..., -9) if asked for."""
        T10 = _f64(transforms10, (self.n_tracks, 4, 4))
        Tw = None if key_transforms_wf is None else _f64(key_transforms_wf, (self.n_tracks, 4, 4))
        hist = np.zeros((self.n_tracks, 10), dtype=np.int64) if histogram else None
        call("tdk_sd_step", self._h, _p(T10), None if Tw is None else _p(Tw), int(bool(commit)),
             None if hist is None else hist.ctypes.data_as(c_int64_p))
        return hist

    def propagate(self, transforms10, commit=True):
        """increment_age + propagate on the current maps."""
        T10 = _f64(transforms10, (self.n_tracks, 4, 4))
        call("tdk_sd_propagate", self._h, _p(T10), int(bool(commit)))

    def update_depth(self, key_transforms_wf=None, commit=True, histogram=False):
        """update_depth with the current maps as (age, prior depth, prior variance)."""
        Tw = None if key_transforms_wf is None else _f64(key_transforms_wf, (self.n_tracks, 4, 4))
        hist = np.zeros((self.n_tracks, 10), dtype=np.int64) if histogram else None
        call("tdk_sd_update_depth", self._h, None if Tw is None else _p(Tw), int(bool(commit)),
             None if hist is None else hist.ctypes.data_as(c_int64_p))
        return hist

    def export_dvo(self, batch):
        """Fills `batch` (a DvoBatch with one pair per track) on the device."""
        call("tdk_sd_export_dvo", self._h, batch._h)

    def timing(self):
        """dict(warp_ms, update_depth_ms, step_ms) of the last step (HIP events)."""
        ms = np.empty(3)
        call("tdk_sd_get_timing", self._h, _p(ms))
        return dict(warp_ms=float(ms[0]), update_depth_ms=float(ms[1]), step_ms=float(ms[2]))


# ---- bundle adjustment ------------------------------------------------------------------
def ba_projection(poses, points, viewpoint_indices, point_indices, jacobians=True):
    poses = _f64(poses).reshape(-1, 6); points = _f64(points).reshape(-1, 3)
    vp = np.ascontiguousarray(viewpoint_indices, dtype=np.int64)
    pt = np.ascontiguousarray(point_indices, dtype=np.int64)
    n = vp.shape[0]
    x = np.empty((n, 2))
    A = np.empty((n, 2, 6)) if jacobians else None
    B = np.empty((n, 2, 3)) if jacobians else None
    call("tdk_ba_projection", _p(poses), poses.shape[0], _p(points), points.shape[0],
         vp.ctypes.data_as(c_int64_p), pt.ctypes.data_as(c_int64_p), n, _p(x),
         _p(A) if jacobians else None, _p(B) if jacobians else None)
    return (x, A, B) if jacobians else x


def ba_exp_so3(rotvecs):
    r = _f64(rotvecs).reshape(-1, 3)
    R = np.empty((r.shape[0], 3, 3))
    call("tdk_ba_exp_so3", _p(r), r.shape[0], _p(R))
    return R


def ba_block_reduce(poses, points, x_true, viewpoint_indices, point_indices):
    poses = _f64(poses).reshape(-1, 6); points = _f64(points).reshape(-1, 3)
    vp = np.ascontiguousarray(viewpoint_indices, dtype=np.int64)
    pt = np.ascontiguousarray(point_indices, dtype=np.int64)
    n = vp.shape[0]
    xt = _f64(x_true, (n, 2))
    nP, nQ = poses.shape[0], points.shape[0]
    U = np.empty((nP, 21)); ea = np.empty((nP, 6)); V = np.empty((nQ, 6)); eb = np.empty((nQ, 3))
    err = C.c_double()
    call("tdk_ba_block_reduce", _p(poses), nP, _p(points), nQ, _p(xt), vp.ctypes.data_as(c_int64_p),
         pt.ctypes.data_as(c_int64_p), n, _p(U), _p(ea), _p(V), _p(eb), C.byref(err))
    return U, ea, V, eb, float(err.value)


class BundleAdjustment(object):
    """Device-resident observation graph for sparse bundle adjustment (tdk_ba)."""

    # tdk_ba_create_ex options: kernels that serve other shapes, forced onto this one (tests)
    SCHUR_PAIRS, SCHUR_GENERAL, SOLVE_HOST, SOLVE_PIVOTED = 1, 2, 4, 8

    def __init__(self, n_poses, n_points, viewpoint_indices, point_indices, x_true, options=0):
        vp = np.ascontiguousarray(viewpoint_indices, dtype=np.int64)
        pt = np.ascontiguousarray(point_indices, dtype=np.int64)
        self.n = vp.shape[0]
        xt = _f64(x_true, (self.n, 2))
        self.n_poses, self.n_points = int(n_poses), int(n_points)
        self._h = C.c_void_p()
        call("tdk_ba_create_ex", self.n_poses, self.n_points, vp.ctypes.data_as(c_int64_p),
             pt.ctypes.data_as(c_int64_p), _p(xt), self.n, int(options), C.byref(self._h))

    def close(self):
        if self._h:
            call("tdk_ba_destroy", self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sum_squared_error(self, poses, points):
        poses = _f64(poses, (self.n_poses, 6)); points = _f64(points, (self.n_points, 3))
        err = C.c_double()
        call("tdk_ba_error", self._h, _p(poses), _p(points), C.byref(err))
        return float(err.value)

    def block_sums(self, poses, points, per_point=True):
        """(U [P,21], ea [P,6], V [Q,6], eb [Q,3], sum ||e||^2): the sums of ba_block_reduce on
        this graph, atomics-free (bit-reproducible).  per_point=False leaves V and eb on the
        device (returned as None): only the per-pose sums cross the bus."""
        poses = _f64(poses, (self.n_poses, 6)); points = _f64(points, (self.n_points, 3))
        U = np.empty((self.n_poses, 21)); ea = np.empty((self.n_poses, 6))
        V = np.empty((self.n_points, 6)) if per_point else None
        eb = np.empty((self.n_points, 3)) if per_point else None
        err = C.c_double()
        call("tdk_ba_block_sums", self._h, _p(poses), _p(points), _p(U), _p(ea),
             None if V is None else _p(V), None if eb is None else _p(eb), C.byref(err))
        return U, ea, V, eb, float(err.value)

    KERNELS = ("block_reduce", "error_reduce", "point_sums", "schur", "backsub", "rcs_solve")

    def set_profiling(self, enabled):
        call("tdk_ba_set_profiling", self._h, int(bool(enabled)))

    def get_profile(self):
        """{kernel: (launches, total_ms)} since set_profiling(True)."""
        n = np.zeros(6, dtype=np.int64); ms = np.zeros(6)
        call("tdk_ba_get_profile", self._h, n.ctypes.data_as(c_int64_p), _p(ms))
        return {k: (int(n[i]), float(ms[i])) for i, k in enumerate(self.KERNELS)}

    def solve(self, poses, points, max_iter=200, initial_mu=1.0, nu=100.0,
              absolute_error_threshold=1e-8, relative_error_threshold=1e-6):
        """Levenberg-Marquardt loop on the device.  Returns (poses [P,6], points [Q,3],
        errors): errors[0] is the initial mean squared error, errors[k] the one
        accepted by iteration k - 1."""
        poses = np.array(_f64(poses, (self.n_poses, 6)))
        points = np.array(_f64(points, (self.n_points, 3)))
        hist = np.zeros(int(max_iter) + 1)
        n_iter = C.c_int()
        call("tdk_ba_solve", self._h, _p(poses), _p(points), int(max_iter), float(initial_mu), float(nu),
             float(absolute_error_threshold), float(relative_error_threshold), _p(hist), C.byref(n_iter))
        return poses, points, hist[:n_iter.value + 1]

    def step(self, poses, points, mu):
        """(dposes [P,6], dpoints [Q,3], sum ||e||^2 at the input parameters)."""
        poses = _f64(poses, (self.n_poses, 6)); points = _f64(points, (self.n_points, 3))
        dposes = np.empty_like(poses); dpoints = np.empty_like(points)
        err = C.c_double()
        call("tdk_ba_step", self._h, _p(poses), _p(points), float(mu), _p(dposes), _p(dpoints), C.byref(err))
        return dposes, dpoints, float(err.value)

