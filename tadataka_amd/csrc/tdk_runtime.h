// tdk_runtime.h -- process-wide HIP state of libtadataka_hip.so: one stream,
// error reporting, and a small grow-only device scratch pool so the
// parity-granular entry points do not hipMalloc on every call.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/tadataka_hip.h"

namespace tdk {

void set_error(const char *fmt, ...);
hipStream_t stream();
// A second stream for uploads of arrays the caller owns: such a call has to wait for ITS copy before it returns
// (the array may go away), and on this stream that wait does not include the kernels queued on stream().  The
// destination must not be in use on stream() (a fresh allocation, or the caller has ordered it).
hipStream_t upload_stream();
tdk_status ensure_device();

// Grow-only device buffers, indexed by slot; contents are undefined between calls.
constexpr int kScratchSlots = 16;
tdk_status scratch(int slot, size_t bytes, void **ptr);

// Pinned host staging buffer (grow-only) for small D2H results.
tdk_status pinned(int slot, size_t bytes, void **ptr);

// granular.hip: bilinear rescale of `batch` images laid out with the given strides
tdk_status launch_rescale(const double *src, int H, int W, double *dst, int Ho, int Wo, int batch,
                          int64_t src_stride, int64_t dst_stride, hipStream_t stream);

// granular.hip: every pyramid level of `n_arrays` arrays in one launch.
// mode 0: one thread per output pixel, blocks ordered so that all levels of one
// (pair, array) are dispatched together (level 0 is re-read from the Infinity
// Cache, not HBM); mode 1: level-0 tiles staged in LDS, one pass.
struct PyramidLevelDesc {
    double *dst[4];
    int64_t stride;
    int H, W;
};
tdk_status launch_pyramid(const double *const *srcs, int n_arrays, int H, int W, int64_t src_stride,
                          int n_out, const PyramidLevelDesc *levels, int batch, int mode, hipStream_t stream);
// the same levels with skimage's anti-aliasing prefilter (Gaussian, sigma = (factor - 1) / 2 per axis)
tdk_status launch_pyramid_aa(const double *const *srcs, int n_arrays, int H, int W, int64_t src_stride, int n_out,
                             const PyramidLevelDesc *levels, int batch, double *weights, bool upload_weights,
                             hipStream_t stream, unsigned skip_mask = 0u);   // bit l: level l is built elsewhere
size_t pyramid_aa_weight_doubles(int n_out);

// pyramid_sep.hip: the same levels as one separable resampling filter per axis (FMA chains over tap
// lists built on the host; last-bit differences from the ndimage operation order).  A plan holds the
// tap lists of one pyramid geometry on the device; levels it cannot take (enlarged axes, tiles beyond
// LDS) are left to launch_pyramid_aa -- pyramid_sep_mask() has a bit per level it builds.
struct PyramidSepPlan;
tdk_status pyramid_sep_create(int H, int W, int n_out, const int *Ho, const int *Wo, hipStream_t stream,
                              PyramidSepPlan **out, unsigned *handled_mask);
tdk_status pyramid_sep_destroy(PyramidSepPlan *p);
bool pyramid_sep_matches(const PyramidSepPlan *p, int H, int W, int n_out, const PyramidLevelDesc *levels);
unsigned pyramid_sep_mask(const PyramidSepPlan *p);
tdk_status launch_pyramid_sep(PyramidSepPlan *p, const double *const *srcs, int n_arrays, int64_t src_stride,
                              const PyramidLevelDesc *levels, int batch, hipStream_t stream);

// dvo.hip: level-0 arrays of a DVO batch, for producers that fill it on the device (tdk_sd_export_dvo)
struct DvoLevel0 {
    double *I0, *D0, *I1, *W0;   // [n_pairs][stride]; W0 is null without a weight map
    double *poses;               // [n_pairs][12] last accepted poses of the device loop
    int64_t stride;
    int H, W, n_pairs;
    hipStream_t stream;
};
tdk_status dvo_level0(tdk_dvo *h, DvoLevel0 *out);

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

// Launch check: catches bad configurations right after the <<<>>>.
#define TDK_LAUNCH_CHECK() TDK_HIP(hipGetLastError())
