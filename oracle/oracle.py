"""CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front-end of oracle/liboracle.so (the plain-C restatement of the
reference hot path, oracle/tdk_oracle.c) plus a NumPy mirror of the reference's
Python DVO orchestration.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; tadataka_amd never does.

Reference lines followed by the Python parts:
  * dvo_estimate_level  <- tadataka/vo/dvo/__init__.py:79-111 (_PoseChangeEstimator)
  * dvo_estimate        <- tadataka/vo/dvo/__init__.py:114-150 (PoseChangeEstimator)
  * exp_se3_t           <- tadataka/se3.py:15-29
  * solve (lstsq)       <- tadataka/math.py:17-19,32-45
"""
import ctypes as C
import os
import subprocess

import numpy as np
from scipy.spatial.transform import Rotation

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

W_NONE, W_HUBER, W_STUDENT_T, W_TUKEY, W_MAP = 0, 1, 2, 3, 4
WEIGHT_MODES = {None: W_NONE, "huber": W_HUBER, "student-t": W_STUDENT_T,
                "tukey": W_TUKEY}

_dp = C.POINTER(C.c_double)
_u64p = C.POINTER(C.c_uint64)
_i64p = C.POINTER(C.c_int64)
_u8p = C.POINTER(C.c_uint8)


class Params(C.Structure):
    _fields_ = [("inv_depth_min", C.c_double), ("inv_depth_max", C.c_double),
                ("geo_coeff", C.c_double), ("photo_coeff", C.c_double),
                ("ref_step_size", C.c_double), ("min_gradient", C.c_double)]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "tdk_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def build_native(out_dir=None):
    """The same C restatement compiled `-O3 -march=native` (contraction allowed) on
    the box it is timed on -- SURVEY section 8(d)'s stronger CPU baseline (ii).  Not
    bit-identical to the exact build; it is only ever TIMED (bench.py)."""
    import tempfile
    out_dir = out_dir or tempfile.gettempdir()
    so = os.path.join(out_dir, "liboracle_native_%d.so" % os.getuid())
    src = os.path.join(_HERE, "tdk_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-std=c11", "-shared", "-o", so, src,
                               "-lm"], stdout=subprocess.DEVNULL)
    return so


def use_library(path=None):
    """Switch the loaded shared object (None: the exact -O2 build)."""
    global _LIB
    _LIB = None
    if path is not None:
        _LIB = _load(path)


def _load(path):
    L = C.CDLL(path)
    L.orc_calc_depth0.restype = C.c_double
    L.orc_ba_block_reduce.restype = C.c_double
    for name in ("orc_dvo_rows", "orc_dvo_normal_equations",
                 "orc_photometric_error", "orc_estimate_debug"):
        getattr(L, name).restype = C.c_int64
    return L


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _load(build())
    return _LIB


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _cam(cam):
    """cam = (fx, fy, ox, oy) as float64[4]."""
    return _d(np.asarray(cam, dtype=np.float64).reshape(4))


# ---- per-point geometry --------------------------------------------------
def normalize(kp, cam):
    kp, p = _d(kp); cam, pc = _cam(cam)
    out = np.empty_like(kp)
    lib().orc_normalize(p, C.c_int64(kp.shape[0]), pc, out.ctypes.data_as(_dp))
    return out


def unnormalize(kp, cam):
    kp, p = _d(kp); cam, pc = _cam(cam)
    out = np.empty_like(kp)
    lib().orc_unnormalize(p, C.c_int64(kp.shape[0]), pc, out.ctypes.data_as(_dp))
    return out


def project_vecs(P):
    P, p = _d(P)
    out = np.empty((P.shape[0], 2))
    lib().orc_project_vecs(p, C.c_int64(P.shape[0]), out.ctypes.data_as(_dp))
    return out


def inv_project_vecs(xs, depths):
    xs, p = _d(xs); depths, pd = _d(depths)
    out = np.empty((xs.shape[0], 3))
    lib().orc_inv_project_vecs(p, pd, C.c_int64(xs.shape[0]), out.ctypes.data_as(_dp))
    return out


def transform(T, P):
    T, pt = _d(T); P, p = _d(P)
    out = np.empty_like(P)
    lib().orc_transform(pt, p, C.c_int64(P.shape[0]), out.ctypes.data_as(_dp))
    return out


def warp_vecs(T10, xs, depths):
    T10, pt = _d(T10); xs, p = _d(xs); depths, pd = _d(depths)
    oxs = np.empty_like(xs); od = np.empty_like(depths)
    lib().orc_warp_vecs(pt, p, pd, C.c_int64(xs.shape[0]),
                        oxs.ctypes.data_as(_dp), od.ctypes.data_as(_dp))
    return oxs, od


def interpolation(image, coords):
    image, pi_ = _d(image); coords, pc = _d(coords)
    out = np.empty(coords.shape[0])
    rc = lib().orc_interpolation(pi_, C.c_int(image.shape[0]), C.c_int(image.shape[1]),
                                 pc, C.c_int64(coords.shape[0]), out.ctypes.data_as(_dp))
    if rc != 0:
        raise ValueError("Coordinates out of image range")
    return out


def calc_depth0(T10, x0, x1):
    T10, pt = _d(T10); x0, p0 = _d(x0); x1, p1 = _d(x1)
    return float(lib().orc_calc_depth0(pt, p0, p1))


def is_in_image_range(kp, shape):
    kp, p = _d(kp)
    out = np.empty(kp.shape[0], dtype=np.uint8)
    lib().orc_is_in_image_range(p, C.c_int64(kp.shape[0]), C.c_int(shape[0]),
                                C.c_int(shape[1]), out.ctypes.data_as(_u8p))
    return out.astype(bool)


# ---- DVO -----------------------------------------------------------------
def image_gradient(I):
    I, p = _d(I)
    GX = np.empty_like(I); GY = np.empty_like(I)
    lib().orc_image_gradient(p, C.c_int(I.shape[0]), C.c_int(I.shape[1]),
                             GX.ctypes.data_as(_dp), GY.ctypes.data_as(_dp))
    return GX, GY


def _weight_args(weights):
    if isinstance(weights, np.ndarray):
        w0, pw = _d(weights)
        return W_MAP, w0, pw
    if weights not in WEIGHT_MODES:
        raise ValueError(f"No such weights '{weights}'")
    return WEIGHT_MODES[weights], None, None


def dvo_rows(I0, D0, I1, GX1, GY1, cam0, cam1, R, t, weights=None):
    """Masked (J, r, w) of one calc_pose_update call."""
    I0, p0 = _d(I0); D0, pd = _d(D0); I1, p1 = _d(I1)
    GX1, pgx = _d(GX1); GY1, pgy = _d(GY1)
    cam0, pc0 = _cam(cam0); cam1, pc1 = _cam(cam1)
    R, pr = _d(R); t, pt = _d(t)
    mode, w0, pw = _weight_args(weights)
    H, W = I0.shape
    N = H * W
    J = np.empty((N, 6)); r = np.empty(N); w = np.empty(N)
    M = lib().orc_dvo_rows(p0, pd, p1, pgx, pgy, pw, C.c_int(H), C.c_int(W), pc0, pc1,
                           pr, pt, C.c_int(mode), J.ctypes.data_as(_dp),
                           r.ctypes.data_as(_dp), w.ctypes.data_as(_dp))
    return J[:M], r[:M], w[:M]


def dvo_normal_equations(I0, D0, I1, GX1, GY1, cam0, cam1, R, t, weights=None):
    I0, p0 = _d(I0); D0, pd = _d(D0); I1, p1 = _d(I1)
    GX1, pgx = _d(GX1); GY1, pgy = _d(GY1)
    cam0, pc0 = _cam(cam0); cam1, pc1 = _cam(cam1)
    R, pr = _d(R); t, pt = _d(t)
    mode, w0, pw = _weight_args(weights)
    H, W = I0.shape
    Hm = np.empty(21); b = np.empty(6)
    M = lib().orc_dvo_normal_equations(p0, pd, p1, pgx, pgy, pw, C.c_int(H), C.c_int(W),
                                       pc0, pc1, pr, pt, C.c_int(mode),
                                       Hm.ctypes.data_as(_dp), b.ctypes.data_as(_dp))
    return Hm, b, int(M)


def photometric_error_sums(I0, D0, I1, cam0, cam1, T10):
    I0, p0 = _d(I0); D0, pd = _d(D0); I1, p1 = _d(I1)
    cam0, pc0 = _cam(cam0); cam1, pc1 = _cam(cam1)
    T10, pt = _d(T10)
    s = C.c_double(0.0)
    n = lib().orc_photometric_error(p0, pd, p1, C.c_int(I0.shape[0]), C.c_int(I0.shape[1]),
                                    pc0, pc1, pt, C.byref(s))
    return float(s.value), int(n)


def photometric_error(I0, D0, I1, cam0, cam1, T10):
    s, n = photometric_error_sums(I0, D0, I1, cam0, cam1, T10)
    return s / n if n > 0 else float("nan")


def rescale_shape(shape, scale):
    return (int(np.round(shape[0] * scale)), int(np.round(shape[1] * scale)))


def rescale(image, scale, anti_aliasing=False):
    image, p = _d(image)
    Ho, Wo = rescale_shape(image.shape, scale)
    out = np.empty((Ho, Wo))
    fn = lib().orc_rescale_anti_aliased if anti_aliasing else lib().orc_rescale_bilinear
    fn(p, C.c_int(image.shape[0]), C.c_int(image.shape[1]), out.ctypes.data_as(_dp), C.c_int(Ho), C.c_int(Wo))
    return out


# ---------------------------------------------------------------------------
# skimage.transform.rescale, faithfully (scikit-image 0.18.3; orc_rescale_skimage in
# tdk_oracle.c).  The two interpreter-dependent ingredients -- the affine map that
# resize() ESTIMATES by SVD and scipy.ndimage's Gaussian kernels -- are computed here
# with the very NumPy calls skimage / scipy make (so on any interpreter they are what
# skimage would get there), or come from a fixture that recorded the generator's.
# ---------------------------------------------------------------------------
def _center_and_normalize_points(points):
    """skimage/transform/_geometric.py:18-69 (Hartley normalisation), call for call."""
    import math
    centroid = np.mean(points, axis=0)
    rms = math.sqrt(np.sum((points - centroid) ** 2) / points.shape[0])
    norm_factor = math.sqrt(2) / rms
    matrix = np.array([[norm_factor, 0, -norm_factor * centroid[0]],
                       [0, norm_factor, -norm_factor * centroid[1]],
                       [0, 0, 1]])
    pointsh = np.vstack([points.T, np.ones((points.shape[0]),)])
    new_pointsh = (matrix @ pointsh).T
    new_points = new_pointsh[:, :2]
    new_points[:, 0] /= new_pointsh[:, 2]
    new_points[:, 1] /= new_pointsh[:, 2]
    return matrix, new_points


def skimage_resize_map(in_shape, out_shape):
    """(ax, bx, ay, by) of skimage.transform.resize's warp: _warps.py:156-176 (three corner
    correspondences, AffineTransform.estimate, off-diagonals zeroed) with
    ProjectiveTransform.estimate (_geometric.py:652-702) restated call for call."""
    rows, cols = float(out_shape[0]), float(out_shape[1])
    factors = np.asarray(in_shape, dtype=float) / np.asarray([rows, cols], dtype=float)
    if rows == 1 and cols == 1:
        return np.array([1.0, in_shape[1] / 2.0 - 0.5, 1.0, in_shape[0] / 2.0 - 0.5])
    src = np.array([[1, 1], [1, rows], [cols, rows]]) - 1
    dst = np.zeros(src.shape, dtype=np.double)
    dst[:, 0] = factors[1] * (src[:, 0] + 0.5) - 0.5
    dst[:, 1] = factors[0] * (src[:, 1] + 0.5) - 0.5
    src_matrix, s = _center_and_normalize_points(src)
    dst_matrix, d = _center_and_normalize_points(dst)
    xs, ys, xd, yd = s[:, 0], s[:, 1], d[:, 0], d[:, 1]
    n = s.shape[0]
    A = np.zeros((n * 2, 9))
    A[:n, 0] = xs; A[:n, 1] = ys; A[:n, 2] = 1; A[:n, 6] = -xd * xs; A[:n, 7] = -xd * ys
    A[n:, 3] = xs; A[n:, 4] = ys; A[n:, 5] = 1; A[n:, 6] = -yd * xs; A[n:, 7] = -yd * ys
    A[:n, 8] = xd; A[n:, 8] = yd
    coeffs = list(range(6))                      # AffineTransform._coeffs
    A = A[:, coeffs + [8]]
    _, _, V = np.linalg.svd(A)
    Hm = np.zeros((3, 3))
    Hm.flat[coeffs + [8]] = -V[-1, :-1] / V[-1, -1]
    Hm[2, 2] = 1
    Hm = np.linalg.inv(dst_matrix) @ Hm @ src_matrix
    return np.array([Hm[0, 0], Hm[0, 2], Hm[1, 1], Hm[1, 2]])


def scipy_gaussian_kernel(sigma):
    """scipy.ndimage gaussian_filter1d's kernel (truncate 4), or None where gaussian_filter skips the axis."""
    if not sigma > 1e-15:
        return None
    radius = int(4.0 * float(sigma) + 0.5)
    sigma2 = sigma * sigma
    x = np.arange(-radius, radius + 1)
    phi_x = np.exp(-0.5 / sigma2 * x ** 2)
    return np.ascontiguousarray((phi_x / phi_x.sum())[::-1])


def skimage_plan(in_shape, out_shape, anti_aliasing=True):
    """Everything of resize() that depends on shapes only: {'map', 'wr', 'wc'}."""
    factors = np.asarray(in_shape, dtype=float) / np.asarray(out_shape, dtype=float)
    sigma = np.maximum(0, (factors - 1) / 2)
    return {"map": skimage_resize_map(in_shape, out_shape),
            "wr": scipy_gaussian_kernel(sigma[0]) if anti_aliasing else None,
            "wc": scipy_gaussian_kernel(sigma[1]) if anti_aliasing else None}


def rescale_skimage(image, scale, plan=None, anti_aliasing=True, clip=True):
    """skimage.transform.rescale(image, scale) of a 2-D float64 image, to the bit given the plan."""
    image, p = _d(image)
    Ho, Wo = rescale_shape(image.shape, scale)
    if plan is None:
        plan = skimage_plan(image.shape, (Ho, Wo), anti_aliasing)
    out = np.empty((Ho, Wo))
    m = np.ascontiguousarray(plan["map"], dtype=np.float64)
    wr = None if plan.get("wr") is None or len(plan["wr"]) == 0 else np.ascontiguousarray(plan["wr"], dtype=np.float64)
    wc = None if plan.get("wc") is None or len(plan["wc"]) == 0 else np.ascontiguousarray(plan["wc"], dtype=np.float64)
    lib().orc_rescale_skimage(
        p, C.c_int(image.shape[0]), C.c_int(image.shape[1]), out.ctypes.data_as(_dp), C.c_int(Ho), C.c_int(Wo),
        m.ctypes.data_as(_dp),
        None if wr is None else wr.ctypes.data_as(_dp), C.c_int(0 if wr is None else len(wr) // 2),
        None if wc is None else wc.ctypes.data_as(_dp), C.c_int(0 if wc is None else len(wc) // 2),
        C.c_int(1 if clip else 0))
    return out


def gaussian_weights(sigma, radius=None):
    radius = lib().orc_gaussian_radius(C.c_double(sigma)) if radius is None else radius
    w = np.empty(2 * radius + 1)
    lib().orc_gaussian_weights(C.c_double(sigma), C.c_int(radius), w.ctypes.data_as(_dp))
    return w


def gaussian_filter_mirror(image, weights_rows, weights_cols):
    """scipy.ndimage.gaussian_filter(image, sigma, mode='mirror') for given 1-D kernels (None: skip the axis)."""
    image, p = _d(image)
    out = np.empty_like(image)
    wr = None if weights_rows is None else np.ascontiguousarray(weights_rows, dtype=np.float64)
    wc = None if weights_cols is None else np.ascontiguousarray(weights_cols, dtype=np.float64)
    lib().orc_gaussian_filter_mirror(
        p, C.c_int(image.shape[0]), C.c_int(image.shape[1]),
        None if wr is None else wr.ctypes.data_as(_dp), C.c_int(0 if wr is None else len(wr) // 2),
        None if wc is None else wc.ctypes.data_as(_dp), C.c_int(0 if wc is None else len(wc) // 2),
        out.ctypes.data_as(_dp))
    return out


def tangent_so3(v):
    return np.array([[0., -v[2], v[1]], [v[2], 0., -v[0]], [-v[1], v[0], 0.]])


def exp_se3_t(xi):
    """tadataka/se3.py:15-29 -- K is built from the *normalised* rotvec."""
    v, rotvec = xi[:3], xi[3:]
    theta = np.linalg.norm(rotvec)
    omega = np.zeros(3) if theta == 0 else rotvec / theta
    K = tangent_so3(omega)
    I = np.eye(3)
    if theta < 1e-16:
        V = I + K * theta / 2 + np.dot(K, K) * pow(theta, 2) / 6
    else:
        V = (I + (1 - np.cos(theta)) / theta * K +
             (theta - np.sin(theta)) / theta * np.dot(K, K))
    return np.dot(V, v)


def motion_matrix(R, t):
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return T


def solve_lstsq(J, r, w=None):
    """tadataka/math.py:32-45 with method='lstsq'."""
    if w is None:
        return np.linalg.lstsq(J, r, rcond=None)[0]
    sw = np.sqrt(w)
    return np.linalg.lstsq(J * sw.reshape(-1, 1), r * sw, rcond=None)[0]


def dvo_estimate_level(I0, D0, I1, cam0, cam1, rotation, t, weights=None,
                       max_iter=20, trace=None):
    """_PoseChangeEstimator.__call__ (tadataka/vo/dvo/__init__.py:79-111).

    rotation is a scipy Rotation, t a 3-vector (the reference's Pose fields).
    Returns (rotation, t).  `trace`, if a list, receives one dict per
    iteration with the intermediates."""
    GX1, GY1 = image_gradient(I1)
    prev_error = photometric_error(I0, D0, I1, cam0, cam1,
                                   motion_matrix(rotation.as_matrix(), t))
    if trace is not None:
        trace.append({"error0": prev_error})
    for _ in range(max_iter):
        J, r, w = dvo_rows(I0, D0, I1, GX1, GY1, cam0, cam1,
                           rotation.as_matrix(), t, weights)
        if J.shape[0] == 0:
            return rotation, t
        xi = solve_lstsq(J, r, None if weights is None else w)
        drot = Rotation.from_rotvec(xi[3:])
        cand_rot = drot * rotation
        cand_t = np.dot(drot.as_matrix(), t) + exp_se3_t(xi)
        curr_error = photometric_error(I0, D0, I1, cam0, cam1,
                                       motion_matrix(cand_rot.as_matrix(), cand_t))
        if trace is not None:
            trace.append({"n_valid": J.shape[0], "xi": xi, "error": curr_error})
        if curr_error > prev_error:
            break
        prev_error = curr_error
        rotation, t = cand_rot, cand_t
    return rotation, t


def dvo_estimate(I0, D0, I1, cam0, cam1, weights=None, n_coarse_to_fine=5,
                 max_iter=20, layer_size_ratio=1.5, rotation=None, t=None, anti_aliasing=False,
                 pyramid=None, plans=None, level_poses=None):
    """PoseChangeEstimator.__call__ (tadataka/vo/dvo/__init__.py:125-150).

    pyramid="skimage": every level -- level 0 / scale 1.0 included, as the reference does -- through
    rescale_skimage (scikit-image 0.18.3 to the bit, given the plans: a callable (in_shape, out_shape)
    -> plan, or None for this interpreter's own).  Otherwise the idealised readings of earlier rounds:
    level 0 is the input itself, the others sample at (i + 0.5) * factor - 0.5, plain bilinear or
    (anti_aliasing=True) behind the Gaussian prefilter."""
    if pyramid == "skimage":
        def rescale(image, scale):
            out_shape = rescale_shape(image.shape, scale)
            plan = plans(image.shape, out_shape) if plans is not None else None
            return rescale_skimage(image, scale, plan)
    else:
        def rescale(image, scale, _aa=anti_aliasing):
            return globals()["rescale"](image, scale, anti_aliasing=_aa)
    rotation = Rotation.from_rotvec(np.zeros(3)) if rotation is None else rotation
    t = np.zeros(3) if t is None else t
    cam0 = np.asarray(cam0, dtype=np.float64); cam1 = np.asarray(cam1, dtype=np.float64)
    for level in reversed(range(n_coarse_to_fine)):
        scale = 1 / pow(layer_size_ratio, level)
        W0 = rescale(weights, scale) if isinstance(weights, np.ndarray) else weights
        rotation, t = dvo_estimate_level(
            rescale(I0, scale), rescale(D0, scale), rescale(I1, scale),
            cam0 * scale, cam1 * scale, rotation, t, W0, max_iter)
        if level_poses is not None:
            level_poses.append(np.concatenate([rotation.as_rotvec(), t]))
    return rotation, t


def fixture_plans(npz):
    """(in_shape, out_shape) -> plan from the plan_* records of a tests/golden/skimage_*.npz."""
    def get(in_shape, out_shape):
        key = f"plan_{in_shape[0]}x{in_shape[1]}_{out_shape[0]}x{out_shape[1]}"
        return {"map": npz[key + "_map"], "wr": npz[key + "_wr"], "wc": npz[key + "_wc"]}
    return get


# ---- semi-dense ------------------------------------------------------------
def make_params(min_depth, max_depth, geo_coeff, photo_coeff, ref_step_size,
                min_gradient):
    p = Params()
    lib().orc_make_params(C.c_double(min_depth), C.c_double(max_depth),
                          C.c_double(geo_coeff), C.c_double(photo_coeff),
                          C.c_double(ref_step_size), C.c_double(min_gradient),
                          C.byref(p))
    return p


def sobel(img):
    img, p = _d(img)
    gx = np.empty_like(img); gy = np.empty_like(img)
    lib().orc_sobel(p, C.c_int(img.shape[0]), C.c_int(img.shape[1]),
                    gx.ctypes.data_as(_dp), gy.ctypes.data_as(_dp))
    return gx, gy


def increment_age(age0, cam0, cam1, T10, depth0):
    age0 = np.ascontiguousarray(age0, dtype=np.uint64)
    cam0, pc0 = _cam(cam0); cam1, pc1 = _cam(cam1)
    T10, pt = _d(T10); depth0, pd = _d(depth0)
    age1 = np.empty_like(age0)
    lib().orc_increment_age(age0.ctypes.data_as(_u64p), C.c_int(age0.shape[0]),
                            C.c_int(age0.shape[1]), pc0, pc1, pt, pd,
                            age1.ctypes.data_as(_u64p))
    return age1


def propagate(T10, cam0, cam1, depth0, var0, default_depth, default_variance,
              uncertaintity_bias):
    cam0, pc0 = _cam(cam0); cam1, pc1 = _cam(cam1)
    T10, pt = _d(T10); depth0, pd = _d(depth0); var0, pv = _d(var0)
    depth1 = np.empty_like(depth0); var1 = np.empty_like(var0)
    lib().orc_propagate(pt, pc0, pc1, pd, pv, C.c_int(depth0.shape[0]),
                        C.c_int(depth0.shape[1]), C.c_double(default_depth),
                        C.c_double(default_variance), C.c_double(uncertaintity_bias),
                        depth1.ctypes.data_as(_dp), var1.ctypes.data_as(_dp))
    return depth1, var1


def transform_rk(T_wk, T_wr):
    T_wk, pk = _d(T_wk); T_wr, pr = _d(T_wr)
    out = np.empty((4, 4))
    lib().orc_transform_rk(pk, pr, out.ctypes.data_as(_dp))
    return out


def estimate_debug(u_key, prior_depth, prior_variance, key, ref, params):
    """key / ref = (cam[4], image, T_wf[4,4])."""
    u = np.ascontiguousarray(u_key, dtype=np.int64)
    kc, pkc = _cam(key[0]); ki, pki = _d(key[1]); kT, pkT = _d(key[2])
    rc, prc = _cam(ref[0]); ri, pri = _d(ref[1]); rT, prT = _d(ref[2])
    od = C.c_double(); ov = C.c_double()
    f = lib().orc_estimate_debug(u.ctypes.data_as(_i64p), C.c_double(prior_depth),
                                 C.c_double(prior_variance), pkc, pki, pkT, prc, pri,
                                 prT, C.c_int(ki.shape[0]), C.c_int(ki.shape[1]),
                                 C.byref(params), C.byref(od), C.byref(ov))
    return float(od.value), float(ov.value), int(f)


def update_depth(key, refs, age, prior_depth, prior_variance, params, T_rks=None):
    """key = (cam, image, T); refs = list of (cam, image, T).
    Returns (depth, variance, flag) like src/py/semi_dense.rs:182-186.
    T_rks (n_ref x 4 x 4): use these instead of orc_transform_rk's inv(T_wr) T_wk --
    the sensitivity tests pass the LAPACK inverse the reference calls (semi_dense.rs:83-89)."""
    kc, pkc = _cam(key[0]); ki, pki = _d(key[1]); kT, pkT = _d(key[2])
    H, W = ki.shape
    n_ref = len(refs)
    rcams = np.ascontiguousarray([np.asarray(r[0], dtype=np.float64) for r in refs]).reshape(n_ref, 4)
    rimgs = np.ascontiguousarray([r[1] for r in refs], dtype=np.float64).reshape(n_ref, H, W)
    rTs = np.ascontiguousarray([r[2] for r in refs], dtype=np.float64).reshape(n_ref, 4, 4)
    age = np.ascontiguousarray(age, dtype=np.uint64)
    pd_, ppd = _d(prior_depth); pv_, ppv = _d(prior_variance)
    depth = np.empty((H, W)); var = np.empty((H, W)); flag = np.empty((H, W), dtype=np.int64)
    if T_rks is not None:
        T_rks = np.ascontiguousarray(T_rks, dtype=np.float64).reshape(n_ref, 4, 4)
    ptrk = T_rks.ctypes.data_as(_dp) if T_rks is not None else None
    rc = lib().orc_update_depth_trk(pkc, pki, pkT, C.c_int(n_ref), rcams.ctypes.data_as(_dp),
                                rimgs.ctypes.data_as(_dp), rTs.ctypes.data_as(_dp), ptrk,
                                age.ctypes.data_as(_u64p), ppd, ppv, C.c_int(H), C.c_int(W),
                                C.byref(params), depth.ctypes.data_as(_dp),
                                var.ctypes.data_as(_dp), flag.ctypes.data_as(_i64p))
    if rc != 0:
        raise RuntimeError("Age exceeds the refframe size")
    return depth, var, flag


# ---- semi-dense post-steps, colour conversion ------------------------
def regularize_patch(inv_depth, inv_variance, flag):
    """regularize_patch (regularization.rs:5-27) on 3x3 arrays; None if nothing contributes."""
    a, pa = _d(np.asarray(inv_depth, dtype=np.float64).reshape(9))
    b, pb = _d(np.asarray(inv_variance, dtype=np.float64).reshape(9))
    f = np.ascontiguousarray(flag, dtype=np.int64).reshape(9)
    out = C.c_double()
    ok = lib().orc_regularize_patch(pa, pb, f.ctypes.data_as(_i64p), C.byref(out))
    return float(out.value) if ok else None


def regularize(depth, variance, flag):
    depth, pd = _d(depth); variance, pv = _d(variance)
    flag = np.ascontiguousarray(flag, dtype=np.int64)
    out = np.empty_like(depth)
    lib().orc_regularize(pd, pv, flag.ctypes.data_as(_i64p), C.c_int(depth.shape[0]),
                         C.c_int(depth.shape[1]), out.ctypes.data_as(_dp))
    return out


def fusion_arrays(mu1, mu2, var1, var2):
    mu1, p1 = _d(mu1); mu2, p2 = _d(mu2); var1, q1 = _d(var1); var2, q2 = _d(var2)
    mu = np.empty_like(mu1); var = np.empty_like(mu1)
    lib().orc_fusion_arrays(p1, p2, q1, q2, C.c_int64(mu1.size), mu.ctypes.data_as(_dp),
                            var.ctypes.data_as(_dp))
    return mu, var


def rgb2gray(rgb):
    rgb = np.asarray(rgb)
    if rgb.dtype == np.uint8:
        rgb = rgb * (1.0 / 255.0)      # img_as_float: np.multiply(image, 1. / imax_in)
    rgb, p = _d(rgb)
    H, W, ch = rgb.shape
    out = np.empty((H, W))
    lib().orc_rgb2gray(p, C.c_int64(H * W), C.c_int(ch), out.ctypes.data_as(_dp))
    return out


def semi_dense_step(key, prev_cam, refs, T10, age0, depth0, var0, params, default_depth,
                    default_variance, uncertaintity_bias):
    """One mapping step of examples/semi_dense_vo.py:182-199: increment_age ->
    propagate -> update_depth.  key = (cam, image, T_wf) of the newest frame,
    refs = earlier frames, oldest first.  Returns (depth, variance, age, flag)."""
    age1 = increment_age(age0, prev_cam, key[0], T10, depth0)
    d1, v1 = propagate(T10, prev_cam, key[0], depth0, var0, default_depth, default_variance,
                       uncertaintity_bias)
    d, v, f = update_depth(key, refs, age1, d1, v1, params)
    return d, v, age1, f


# ---- bundle adjustment -------------------------------------------------------
def exp_so3(rotvec):
    r, p = _d(rotvec)
    out = np.empty((3, 3))
    lib().orc_exp_so3(p, out.ctypes.data_as(_dp))
    return out


def ba_transform_project(pose, point):
    a, pa = _d(pose); b, pb = _d(point)
    out = np.empty(2)
    lib().orc_ba_transform_project(pa, pb, out.ctypes.data_as(_dp))
    return out


def ba_pose_jacobian(pose, point):
    a, pa = _d(pose); b, pb = _d(point)
    out = np.empty((2, 6))
    lib().orc_ba_pose_jacobian(pa, pb, out.ctypes.data_as(_dp))
    return out


def ba_point_jacobian(pose, point):
    a, pa = _d(pose); b, pb = _d(point)
    out = np.empty((2, 3))
    lib().orc_ba_point_jacobian(pa, pb, out.ctypes.data_as(_dp))
    return out


def ba_projection(poses, points, vp_idx, pt_idx, jacobians=True):
    poses, pp = _d(poses); points, pq = _d(points)
    vp = np.ascontiguousarray(vp_idx, dtype=np.int64)
    pt = np.ascontiguousarray(pt_idx, dtype=np.int64)
    n = vp.shape[0]
    x = np.empty((n, 2))
    A = np.empty((n, 2, 6)) if jacobians else None
    B = np.empty((n, 2, 3)) if jacobians else None
    lib().orc_ba_projection(pp, pq, vp.ctypes.data_as(_i64p), pt.ctypes.data_as(_i64p),
                            C.c_int64(n), x.ctypes.data_as(_dp),
                            A.ctypes.data_as(_dp) if jacobians else None,
                            B.ctypes.data_as(_dp) if jacobians else None)
    return (x, A, B) if jacobians else x


def ba_block_reduce(poses, points, x_true, vp_idx, pt_idx):
    poses, pp = _d(poses); points, pq = _d(points); x_true, px = _d(x_true)
    vp = np.ascontiguousarray(vp_idx, dtype=np.int64)
    pt = np.ascontiguousarray(pt_idx, dtype=np.int64)
    nP, nQ = poses.shape[0], points.shape[0]
    U = np.empty((nP, 21)); ea = np.empty((nP, 6)); V = np.empty((nQ, 6)); eb = np.empty((nQ, 3))
    err = lib().orc_ba_block_reduce(pp, C.c_int64(nP), pq, C.c_int64(nQ), px,
                                    vp.ctypes.data_as(_i64p), pt.ctypes.data_as(_i64p),
                                    C.c_int64(vp.shape[0]), U.ctypes.data_as(_dp),
                                    ea.ctypes.data_as(_dp), V.ctypes.data_as(_dp),
                                    eb.ctypes.data_as(_dp))
    return U, ea, V, eb, float(err)
