"""ctypes binding of libtadataka_hip.so (include/tadataka_hip.h).

There is no CPU fallback: if the shared library is missing, or a compute entry
is called without a usable MI355X, the call raises.  Nothing here imports or
calls the test oracle.
"""
import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libtadataka_hip.so")

c_double_p = C.POINTER(C.c_double)
c_int64_p = C.POINTER(C.c_int64)
c_uint64_p = C.POINTER(C.c_uint64)
c_int_p = C.POINTER(C.c_int)


class SemiDenseParams(C.Structure):
    """tdk_semi_dense_params (Params.new, src/py/semi_dense.rs:93-108)."""
    _fields_ = [("min_depth", C.c_double), ("max_depth", C.c_double),
                ("geo_coeff", C.c_double), ("photo_coeff", C.c_double),
                ("ref_step_size", C.c_double), ("min_gradient", C.c_double)]


class TdkError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"libtadataka_hip: status {status}: {message}")
        self.status = status


TDK_OK = 0
TDK_ERR_INVALID_ARGUMENT = -1
TDK_ERR_HIP = -2
TDK_ERR_OUT_OF_RANGE = -3
TDK_ERR_AGE_EXCEEDS_REFFRAMES = -4
TDK_ERR_NO_DEVICE = -5
TDK_ERR_SINGULAR = -6

W_NONE, W_HUBER, W_STUDENT_T, W_TUKEY, W_MAP = 0, 1, 2, 3, 4

_d, _i, _i64, _u64, _vp = c_double_p, C.c_int, C.c_int64, C.c_uint64, C.c_void_p

# name -> argtypes; every entry of include/tadataka_hip.h returning tdk_status
PROTOTYPES = {
    "tdk_device_count": [c_int_p],
    "tdk_set_device": [_i],
    "tdk_get_device": [c_int_p],
    "tdk_sync": [],
    "tdk_device_name": [C.c_char_p, _i],
    "tdk_pinned_alloc": [C.c_size_t, C.POINTER(_vp)],
    "tdk_pinned_free": [_vp],
    "tdk_dvo_upload_mixed": [_vp, _i, C.POINTER(_vp), C.POINTER(_vp)],
    "tdk_dvo_upload_async": [_vp, _i, _i, _i, _vp],
    "tdk_dvo_upload_async_u8": [_vp, _i, _i, _i, _vp],
    "tdk_map_create": [_i, _i, _vp, C.POINTER(_vp)],
    "tdk_map_destroy": [_vp],
    "tdk_map_upload": [_vp, _vp],
    "tdk_map_download": [_vp, _vp],
    "tdk_map_shape": [_vp, c_int_p, c_int_p],
    "tdk_map_device_ptr": [_vp, C.POINTER(_vp)],
    "tdk_frame_device_ptr": [_vp, C.POINTER(_vp)],
    "tdk_frame_download": [_vp, _d],
    "tdk_map_safe_invert": [_vp, C.c_double, _vp],
    "tdk_increment_age_maps": [_vp, _d, _d, _d, _vp, _vp],
    "tdk_propagate_maps": [_d, _d, _d, _vp, _vp, C.c_double, C.c_double, C.c_double, _vp, _vp],
    "tdk_update_depth_maps": [_vp, _vp, _d, _i, _d, C.POINTER(_vp), _d, _vp, _vp, _vp,
                              C.POINTER(SemiDenseParams), _vp, _vp, _vp],
    "tdk_normalize": [_d, _i64, _d, _d],
    "tdk_unnormalize": [_d, _i64, _d, _d],
    "tdk_project_vecs": [_d, _i64, _d],
    "tdk_inv_project_vecs": [_d, _d, _i64, _d],
    "tdk_transform": [_d, _d, _i64, _d],
    "tdk_warp_vecs": [_d, _d, _d, _i64, _d, _d],
    "tdk_interpolation": [_d, _i, _i, _d, _i64, _d],
    "tdk_calc_depth0": [_d, _d, _d, _d],
    "tdk_image_gradient": [_d, _i, _i, _d, _d],
    "tdk_rescale": [_d, _i, _i, _d, _i, _i],
    "tdk_rescale_anti_aliased": [_d, _i, _i, _d, _i, _i],
    "tdk_debug_check_canaries": [c_int_p],
    "tdk_set_option": [_i, _i],
    "tdk_dvo_set_option": [_vp, _i, _i],
    "tdk_rescale_skimage": [_d, _i, _i, _d, _i, _i, _d, _d, _i, _d, _i, _i],
    "tdk_dvo_set_anti_aliasing": [_vp, _i],
    "tdk_dvo_set_level_plan": [_vp, _i, _d, _d, _i, _d, _i],
    "tdk_dvo_set_rescale_options": [_vp, C.c_uint, _i],
    "tdk_dvo_create": [_i, _i, _i, _i, C.c_double, _i, C.POINTER(_vp)],
    "tdk_dvo_destroy": [_vp],
    "tdk_dvo_upload": [_vp, _i, _d, _d, _d, _d],
    "tdk_dvo_fill_synthetic": [_vp, _d, _d, _u64, C.c_double],
    "tdk_dvo_build_pyramid": [_vp],
    "tdk_dvo_build_pyramid_arrays": [_vp, C.c_uint],
    "tdk_dvo_download": [_vp, _i, _i, _i, _d],
    "tdk_dvo_level_shape": [_vp, _i, c_int_p, c_int_p],
    "tdk_dvo_evaluate": [_vp, _i, _d, _d, _d, _i, _d, _d, c_int64_p, _d, c_int64_p],
    "tdk_dvo_estimate_level": [_vp, _i, _d, _d, _d, _i, _i, c_int_p],
    "tdk_dvo_photometric_error": [_vp, _i, _d, _d, _d, _d, c_int64_p],
    "tdk_dvo_estimate": [_vp, _d, _d, _d, _i, _i, c_int64_p],
    "tdk_dvo_get_stream": [_vp, C.POINTER(_vp)],
    "tdk_dvo_get_warnings": [_vp, c_int_p],
    "tdk_dvo_get_counts": [_vp, c_int64_p, c_int64_p],
    "tdk_dvo_get_tukey_fallbacks": [_vp, c_int64_p],
    "tdk_dvo_get_student_redos": [_vp, c_int64_p],
    "tdk_dvo_get_student_fallbacks": [_vp, c_int64_p],
    "tdk_dvo_set_student_passes": [_vp, _i],
    "tdk_dvo_get_robust_scale": [_vp, _d],
    "tdk_dvo_set_profiling": [_vp, _i],
    "tdk_dvo_get_profile": [_vp, c_int64_p, _d, c_int64_p],
    "tdk_dvo_get_profile_kind": [_vp, _i, c_int64_p, _d, c_int64_p],
    "tdk_dvo_get_profile_level": [_vp, _i, _i, c_int64_p, _d, c_int64_p],
    "tdk_weighted_normal_equations": [_d, _d, _d, _i64, _i, _d, _d],
    "tdk_dvo_pose_update": [_d, _d, _d, _d, _i, _i, _d, _i64, _i, _d, _d, _d, c_int64_p],
    "tdk_robust_weights": [_d, _i64, _i, _d],
    "tdk_robust_weights_ex": [_d, _i64, _i, C.c_double, C.c_double, _d],
    "tdk_increment_age": [c_uint64_p, _i, _i, _d, _d, _d, _d, c_uint64_p],
    "tdk_propagate": [_d, _d, _d, _d, _d, _i, _i, C.c_double, C.c_double, C.c_double, _d, _d],
    "tdk_update_depth": [_d, _d, _d, _i, _d, _d, _d, c_uint64_p, _d, _d, _i, _i,
                         C.POINTER(SemiDenseParams), _d, _d, c_int64_p],
    "tdk_frame_create": [_d, _i, _i, C.POINTER(_vp)],
    "tdk_frame_destroy": [_vp],
    "tdk_update_depth_frames": [_d, _vp, _d, _i, _d, C.POINTER(_vp), _d, c_uint64_p, _d, _d,
                                C.POINTER(SemiDenseParams), _d, _d, c_int64_p],
    "tdk_estimate_one": [c_int64_p, C.c_double, C.c_double, _d, _d, _d, _d, _d, _d, _i, _i,
                         C.POINTER(SemiDenseParams), _d, _d, c_int64_p],
    "tdk_sobel": [_d, _i, _i, _d, _d],
    "tdk_regularize": [_d, _d, c_int64_p, _i, _i, _d],
    "tdk_fusion_arrays": [_d, _d, _d, _d, _i64, _d, _d],
    "tdk_rgb2gray": [_d, _i, _i, _i, _d],
    "tdk_rgb2gray_u8": [C.POINTER(C.c_uint8), _i, _i, _i, _d],
    "tdk_sd_create": [_i, _i, _i, _i, C.POINTER(_vp)],
    "tdk_sd_destroy": [_vp],
    "tdk_sd_set_age_policy": [_vp, _i],
    "tdk_sd_get_warp_fallbacks": [_vp, c_int64_p],
    "tdk_sd_set_params": [_vp, C.POINTER(SemiDenseParams), C.c_double, C.c_double, C.c_double],
    "tdk_sd_set_maps": [_vp, _i, _d, _d, c_uint64_p],
    "tdk_sd_get_maps": [_vp, _i, _d, _d, c_uint64_p, c_int64_p],
    "tdk_sd_get_results": [_vp, _i, _d, _d, c_uint64_p, c_int64_p],
    "tdk_sd_push_frame": [_vp, _i, _d, _d, _d],
    "tdk_sd_step": [_vp, _d, _d, _i, c_int64_p],
    "tdk_sd_propagate": [_vp, _d, _i],
    "tdk_sd_update_depth": [_vp, _d, _i, c_int64_p],
    "tdk_sd_export_dvo": [_vp, _vp],
    "tdk_sd_get_timing": [_vp, _d],
    "tdk_ba_projection": [_d, _i64, _d, _i64, c_int64_p, c_int64_p, _i64, _d, _d, _d],
    "tdk_ba_exp_so3": [_d, _i64, _d],
    "tdk_ba_block_reduce": [_d, _i64, _d, _i64, _d, c_int64_p, c_int64_p, _i64, _d, _d, _d, _d, _d],
    "tdk_ba_create": [_i64, _i64, c_int64_p, c_int64_p, _d, _i64, C.POINTER(_vp)],
    "tdk_ba_create_ex": [_i64, _i64, c_int64_p, c_int64_p, _d, _i64, C.c_uint, C.POINTER(_vp)],
    "tdk_ba_destroy": [_vp],
    "tdk_ba_error": [_vp, _d, _d, _d],
    "tdk_ba_step": [_vp, _d, _d, C.c_double, _d, _d, _d],
    "tdk_ba_block_sums": [_vp, _d, _d, _d, _d, _d, _d, _d],
    "tdk_ba_set_profiling": [_vp, _i],
    "tdk_ba_get_profile": [_vp, c_int64_p, _d],
    "tdk_comm_available": [],
    "tdk_comm_unique_id": [C.POINTER(C.c_uint8)],
    "tdk_comm_create": [C.POINTER(C.c_uint8), _i, _i, C.POINTER(_vp)],
    "tdk_comm_destroy": [_vp],
    "tdk_comm_rank": [_vp, c_int_p, c_int_p],
    "tdk_comm_all_gather": [_vp, _d, _i64, _d],
    "tdk_comm_all_reduce": [_vp, _d, _i64, _i],
    "tdk_comm_barrier": [_vp],
    "tdk_dvo_gather_poses_start": [_vp, _vp],
    "tdk_dvo_gather_poses_finish": [_vp, _d],
    "tdk_ba_solve": [_vp, _d, _d, _i, C.c_double, C.c_double, C.c_double, C.c_double, _d, C.POINTER(C.c_int)],
}

_lib = None


def load():
    """Returns the loaded library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("TDK_LIBRARY") or LIB_PATH      # TDK_LIBRARY: another build of the same ABI (make asan)
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or `make -C tadataka_amd/csrc`).  There is no CPU fallback.")
    # HIP maps all streams of a process onto a few hardware queues (4 by default) and a stream that waits for an
    # event of another stream holds up every stream behind it in its queue.  Each DvoBatch owns a stream, uploads
    # and the stateless operators have theirs: give them queues of their own unless the caller decided otherwise.
    # (Read by the HIP runtime when it starts: no effect if something else initialised HIP in this process first.)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    lib = C.CDLL(path)
    lib.tdk_version.restype = C.c_char_p
    lib.tdk_last_error.restype = C.c_char_p
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)      # AttributeError here = header/library mismatch
        fn.argtypes = argtypes
        fn.restype = C.c_int
    _lib = lib
    return lib


def check(status):
    if status == TDK_OK:
        return
    msg = load().tdk_last_error().decode("utf-8", "replace")
    if status == TDK_ERR_OUT_OF_RANGE:
        raise ValueError(msg)        # what tadataka.interpolation raises
    raise TdkError(status, msg)


# ctypes releases the GIL around every foreign call.  The C library serialises its entries itself (one process-wide
# recursive mutex, include/tadataka_hip.h); this lock additionally keeps a call and the reading of its status /
# tdk_last_error() together for all Python threads of the process.
_call_lock = threading.RLock()


def call(name, *args):
    with _call_lock:
        check(getattr(load(), name)(*args))


def device_count():
    n = C.c_int(0)
    call("tdk_device_count", C.byref(n))
    return n.value


def require_gpu():
    if device_count() <= 0:
        raise TdkError(TDK_ERR_NO_DEVICE, "no MI355X / HIP device visible; there is no CPU fallback")


def device_name():
    buf = C.create_string_buffer(256)
    call("tdk_device_name", buf, 256)
    return buf.value.decode()
