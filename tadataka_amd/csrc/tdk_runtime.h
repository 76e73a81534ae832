// tdk_runtime.h -- process-wide HIP state of libtadataka_hip.so: one stream,
// error reporting, and a small grow-only device scratch pool so the
// parity-granular entry points do not hipMalloc on every call.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <mutex>

#include "../../include/tadataka_hip.h"

namespace tdk {

void set_error(const char *fmt, ...);
hipStream_t stream();
// A second stream for uploads of arrays the caller owns: such a call has to wait for ITS copy before it returns
// (the array may go away), and on this stream that wait does not include the kernels queued on stream().  The
// destination must not be in use on stream() (a fresh allocation, or the caller has ordered it).
hipStream_t upload_stream();
tdk_status ensure_device();
int option(int which);   // tdk_set_option
// One process-wide recursive mutex taken by EVERY entry of the C ABI (TDK_API_GUARD at its top): the library keeps
// process-wide state (one stream and grow-only scratch pools for the stateless entries, the allocation registry) and
// a handle must not be used from two threads at once, so concurrent callers are serialised here instead of being
// asked to do it themselves.  (tdk_last_error() stays per thread.)
std::recursive_mutex &api_mutex();

// Grow-only device buffers, indexed by slot; contents are undefined between calls.
constexpr int kScratchSlots = 16;
tdk_status scratch(int slot, size_t bytes, void **ptr);

// Pinned host staging buffer (grow-only) for small D2H results.
tdk_status pinned(int slot, size_t bytes, void **ptr);

// pyramid.hip: skimage.transform.rescale on the device.  A level's sample positions and Gaussian kernels either
// come from the host (a "plan": the affine map skimage's resize() estimates and scipy.ndimage's kernels, both
// products of the caller's NumPy -- tdk_dvo_set_level_plan, tdk_rescale_skimage) or are the ideal ones.
struct AxisMap {
    double a, b;   // position of output index o: a * o + b (one product, one sum) ...
    double s;      // ... or, ideal: (o + 0.5) * s - 0.5 with s = n_in / n_out
    int ideal;
};
AxisMap ideal_axis(int n_in, int n_out);
AxisMap affine_axis(double a, double b);
struct PyramidLevelDesc {
    double *dst[4];
    int64_t stride;
    int H, W;
    AxisMap mx, my;          // columns, rows
    const double *wr, *wc;   // HOST: scipy's kernels, 2 R + 1 entries (centre at [R]); null = axis not filtered
    int Rr, Rc;
};
// ideal maps and (anti_aliasing) libm kernels for a level of an H x W source; `storage` holds the kernels:
// 2 * (2 * pyramid_max_radius() + 1) doubles that must outlive the launch
void ideal_level_plan(PyramidLevelDesc *lv, int H, int W, bool anti_aliasing, double *storage);
int pyramid_max_radius();
size_t pyramid_weight_doubles(int n_out);
size_t pyramid_clip_bytes(int64_t n_images, int n_out);
// every level of `levels` for `n_arrays` arrays of `batch` images in a few launches (see pyramid.hip)
tdk_status launch_pyramid(const double *const *srcs, int n_arrays, int H, int W, int64_t src_stride, int n_out,
                          const PyramidLevelDesc *levels, int batch, double *weights, bool upload_weights,
                          void *clip_slots, int stream_mode, hipStream_t stream, int *slots_clean = nullptr);

// dvo.hip: the full-resolution input arrays of a DVO batch, for producers that fill it on the device (tdk_sd_export_dvo)
struct DvoLevel0 {
    double *I0, *D0, *I1, *W0;   // [n_pairs][stride]; W0 is null without a weight map
    double *poses;               // [n_pairs][12] last accepted poses of the device loop
    int64_t stride;
    int H, W, n_pairs;
    hipStream_t stream;
};
tdk_status dvo_level0(tdk_dvo *h, DvoLevel0 *out);

// Device allocations of the whole library go through these (the hipMalloc / hipFree macros below), so that
// TDK_DEBUG_CANARY=1 can put every one of them between two 4 KiB red zones of 0xFF bytes (a NaN pattern as doubles):
// the kernels rely on documented padding (stream loads overrun a block's range by a few hundred pixels)
// and an allocation-size regression would otherwise be a silent out-of-bounds access.  The zones are verified by
// tdk_debug_check_canaries(), by tdk_sync() and when an allocation is freed; a violation is reported with the name of
// the allocation (the pointer expression and source line of its hipMalloc) and makes those calls return TDK_ERR_HIP.
hipError_t dev_malloc(void **ptr, size_t bytes, const char *what, const char *file, int line);
hipError_t dev_free(void *ptr);
bool canaries_enabled();
tdk_status check_canaries();

// Other translation units keep process-wide device state of their own (the map pool and the staging ring of
// semi_dense.hip); a hook registered here runs when tdk_set_device leaves a device, BEFORE its streams go away.
void on_device_release(void (*hook)());

}  // namespace tdk

#define TDK_HIP(call)                                                                   \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            tdk::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),       \
                           __FILE__, __LINE__);                                         \
            return TDK_ERR_HIP;                                                         \
        }                                                                               \
    } while (0)

#define TDK_TRY(call)                     \
    do {                                  \
        tdk_status s_ = (call);           \
        if (s_ != TDK_OK) return s_;      \
    } while (0)

#define TDK_REQUIRE(cond, msg)                                   \
    do {                                                         \
        if (!(cond)) {                                           \
            tdk::set_error("invalid argument: %s", msg);         \
            return TDK_ERR_INVALID_ARGUMENT;                     \
        }                                                        \
    } while (0)

#ifndef TDK_RUNTIME_IMPL
#define hipMalloc(ptr, bytes) tdk::dev_malloc((void **)(ptr), (bytes), #ptr, __FILE__, __LINE__)
#define hipFree(ptr) tdk::dev_free((void *)(ptr))
#endif

#define TDK_API_GUARD std::lock_guard<std::recursive_mutex> tdk_api_guard_(tdk::api_mutex())

// Launch check: catches bad configurations right after the <<<>>>.
#define TDK_LAUNCH_CHECK() TDK_HIP(hipGetLastError())
