/*
 * tadataka_hip.h -- C ABI of libtadataka_hip.so, the MI355X (gfx950) native
 * replacement of Tadataka's per-pixel DVO / semi-dense / BA hot path.
 *
 * This is the drop-in boundary: the reference reaches the same functionality
 * through its compiled extension modules (pyo3 crate `rust_bindings`,
 * pybind11 `tadataka.camera._normalizer`, Cython `tadataka.transform_project`).
 * Each entry point below names the reference interface it replaces
 * (file:line in the reference checkout).  INTEGRATION.md shows the binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - extern "C", POD arguments only, caller-allocated outputs.
 *   - every function returns a tdk_status (0 = ok, < 0 = error); nothing
 *     throws, aborts or exits.  tdk_last_error() gives a message.
 *   - arrays are C-contiguous float64 unless stated; images are [row=y][col=x];
 *     coordinates are (x, y) pairs; camera = {fx, fy, ox, oy}; transforms are
 *     row-major 4x4; poses for DVO are 12 doubles {R row-major (9), t (3)}.
 *   - pointers are HOST pointers unless the name ends in `_dev`.
 *   - calls are serialised on one HIP stream per process; they block until the
 *     result is in the output buffers unless stated otherwise.
 *   - thread-safe by serialisation (since round 5): the library keeps process-wide state (one stream and
 *     grow-only device / pinned scratch pools for the stateless entry points) and a handle (tdk_dvo, tdk_sd,
 *     tdk_ba, tdk_comm) cannot serve two callers at once, so EVERY entry takes one process-wide recursive
 *     mutex for its duration.  Concurrent callers are correct, not concurrent: the reference's extension
 *     modules hold the GIL for the whole call (single-threaded by construction); a binding that releases the
 *     GIL (ctypes does) needs nothing more.  tdk_last_error() is per thread.
 */
#ifndef TADATAKA_HIP_H
#define TADATAKA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int tdk_status;
enum {
    TDK_OK = 0,
    TDK_ERR_INVALID_ARGUMENT = -1,
    TDK_ERR_HIP = -2,            /* a HIP runtime call failed */
    TDK_ERR_OUT_OF_RANGE = -3,   /* coordinates outside the image (ValueError in the reference) */
    TDK_ERR_AGE_EXCEEDS_REFFRAMES = -4, /* reference: process::exit(1), semi_dense.rs:202-205 */
    TDK_ERR_NO_DEVICE = -5,
    TDK_ERR_SINGULAR = -6
};

/* ---- runtime ------------------------------------------------------------ */
const char *tdk_version(void);
const char *tdk_last_error(void);
tdk_status tdk_device_count(int *count);
tdk_status tdk_set_device(int device);
tdk_status tdk_get_device(int *device);
tdk_status tdk_sync(void);
/* Library-wide options (the stateless entries and every handle read them at each call).
 *   TDK_OPT_PYRAMID_STREAM   which kernel builds the first two shrinking pyramid levels: 1 (default) the streaming
 *                            kernel for batches that fill the chip (>= 256 full-height strips), LDS tiles otherwise;
 *                            0 always the tiles; 2 always the streaming kernel; 3 as 1, but level 0 by the streaming
 *                            kernel of its own that one-level pyramids use (k_level0_stream) instead of fused into the
 *                            pass over the source (same time on the bench batch).  Bit-identical either way.
 *   TDK_OPT_SD_WARP_GATHER   increment_age / propagate: 1 (default) the gather kernels, with the slot path as the
 *                            device-side fallback of tracks whose displacement box exceeds the gather's window;
 *                            0 the slot path for every track.  Bit-identical either way. */
enum { TDK_OPT_PYRAMID_STREAM = 0, TDK_OPT_SD_WARP_GATHER = 1 };
tdk_status tdk_set_option(int option, int value);
/* Debugging aid.  With TDK_DEBUG_CANARY=1 in the environment (read once, at the first allocation) every device
 * allocation of the library sits between two 4 KiB red zones of 0xFF bytes; this call (and tdk_sync, and every
 * destroy) verifies them and returns TDK_ERR_HIP, the damaged allocation named in tdk_last_error(), if a kernel or
 * a copy wrote outside an allocation.  n_allocations (optional): live allocations, -1 when the mode is off. */
tdk_status tdk_debug_check_canaries(int *n_allocations);
tdk_status tdk_device_name(char *buf, int buflen);
/* Page-locked host memory for the asynchronous uploads (tdk_dvo_upload_async). */
tdk_status tdk_pinned_alloc(size_t bytes, void **out);
tdk_status tdk_pinned_free(void *ptr);

/* ---- parity-granular per-point operators (1:1 with rust_bindings.*) ------- */
/* tadataka.camera._normalizer.normalize/unnormalize (tadataka/camera/_normalizer.cpp:12-27),
 * src/camera.rs:32-63 */
tdk_status tdk_normalize(const double *keypoints, int64_t n, const double *camera, double *out);
tdk_status tdk_unnormalize(const double *keypoints, int64_t n, const double *camera, double *out);
/* rust_bindings.projection.project_vecs / inv_project_vecs (src/py/projection.rs:7-41) */
tdk_status tdk_project_vecs(const double *points, int64_t n, double *out);
tdk_status tdk_inv_project_vecs(const double *xs, const double *depths, int64_t n, double *out);
/* rust_bindings.transform.transform (src/py/transform.rs:6-15) */
tdk_status tdk_transform(const double *transform10, const double *points, int64_t n, double *out);
/* rust_bindings.warp.warp_vecs (src/py/warp.rs:51-67) */
tdk_status tdk_warp_vecs(const double *transform10, const double *xs, const double *depths,
                         int64_t n, double *out_xs, double *out_depths);
/* rust_bindings.interpolation.interpolation behind tadataka.interpolation.interpolation
 * (src/py/interpolation.rs:6-15, tadataka/interpolation/__init__.py:13-29).
 * Returns TDK_ERR_OUT_OF_RANGE where the Python wrapper raises ValueError. */
tdk_status tdk_interpolation(const double *image, int height, int width,
                             const double *coordinates, int64_t m, double *out);
/* rust_bindings.triangulation.calc_depth0 (src/py/triangulation.rs:7-19) */
tdk_status tdk_calc_depth0(const double *transform10, const double *x0, const double *x1, double *depth);
/* tadataka.vo.dvo.jacobian.calc_image_gradient = np.gradient (jacobian.py:27-29); out (DX, DY) */
tdk_status tdk_image_gradient(const double *image, int height, int width, double *gx, double *gy);
/* skimage.transform.rescale(image, scale) / resize(image, (out_height, out_width)) of a 2-D float64 image as
 * the reference calls it for every pyramid level (tadataka/vo/dvo/__init__.py:144-148; order 1, mode 'reflect',
 * anti_aliasing and clip as given), bit for bit what scikit-image 0.18.3 returns -- pinned against that package
 * run in the build container (tests/golden/skimage_rescale.npz).  Two ingredients of skimage's pipeline are
 * products of the caller's NumPy / LAPACK rather than of the algorithm and are therefore ARGUMENTS:
 *   map[4] = (ax, bx, ay, by): output (row oy, column ox) samples the filtered image at (ay * oy + by, ax * ox + bx);
 *            resize() estimates this map by SVD from three corner correspondences (skimage/transform/_warps.py:156-176),
 *            a few ulp / 1e-13 off factor and factor / 2 - 1 / 2, differently on every LAPACK build;
 *   w_rows / w_cols: scipy.ndimage's Gaussian kernels (2 * radius + 1 doubles, numpy.exp(-0.5 / sigma^2 * x^2) / sum);
 *            radius 0 = that axis is not filtered (sigma <= 1e-15, or anti_aliasing=False).
 * tadataka_amd/rescale_plan.py computes both with the NumPy calls skimage and scipy make.  clip: skimage's
 * clip=True (outputs clipped to the extremes of the filtered image). */
tdk_status tdk_rescale_skimage(const double *image, int height, int width, double *out, int out_height, int out_width,
                               const double *map, const double *w_rows, int radius_rows, const double *w_cols,
                               int radius_cols, int clip);
/* The same pipeline with the IDEAL constants instead of a plan -- sample positions (i + 0.5) * factor - 0.5,
 * kernels from libm's exp, no clip -- without (tdk_rescale) and with (tdk_rescale_anti_aliased) the Gaussian
 * prefilter.  What a caller without NumPy gets; within ~1e-13 of skimage, not bit-identical with it. */
tdk_status tdk_rescale(const double *image, int height, int width, double *out,
                       int out_height, int out_width);
tdk_status tdk_rescale_anti_aliased(const double *image, int height, int width, double *out,
                                    int out_height, int out_width);

/* ---- DVO: device-resident batch of frame pairs (new, fused) ---------------
 * Replaces, for all pairs at once, the per-iteration body of
 * _PoseChangeEstimator.__call__ (tadataka/vo/dvo/__init__.py:79-111):
 * calc_pose_update (:46-70) + PhotometricError (tadataka/metric.py:13-39). */
typedef struct tdk_dvo tdk_dvo;

enum { TDK_W_NONE = 0, TDK_W_HUBER = 1, TDK_W_STUDENT_T = 2, TDK_W_TUKEY = 3, TDK_W_MAP = 4 };

/* Allocates device storage for n_pairs pairs of height x width frames
 * (I0, D0, I1 and, if with_weight_map, W0) plus an n_levels-deep pyramid with
 * scale 1/ratio^level. */
tdk_status tdk_dvo_create(int n_pairs, int height, int width, int n_levels, double ratio,
                          int with_weight_map, tdk_dvo **out);
tdk_status tdk_dvo_destroy(tdk_dvo *h);
/* Host -> device copy of one pair (level 0).  weight_map may be NULL. */
tdk_status tdk_dvo_upload(tdk_dvo *h, int pair, const double *I0, const double *D0,
                          const double *I1, const double *weight_map);
/* As tdk_dvo_upload with every array taken from the host OR from the device: for k = 0..3
 * (I0, D0, I1, W0) device4[k] (a device pointer: tdk_map_device_ptr / tdk_frame_device_ptr) is
 * copied on the device if non-NULL, else host4[k] is uploaded if non-NULL, else the array is left
 * as it is.  What the drop-in PoseChangeEstimator calls when examples/semi_dense_vo.py:44-53 hands
 * it the depth map and 1 / variance that the previous mapping step left on the device. */
tdk_status tdk_dvo_upload_mixed(tdk_dvo *h, int pair, const double *const *host4,
                                const double *const *device4);
/* Asynchronous upload of one array (which = 0 I0, 1 D0, 2 I1, 3 W0) of pairs
 * [first_pair, first_pair + n_pairs) from PINNED host memory (tdk_pinned_alloc) laid out
 * [n_pairs][height * width]: the copy runs on the batch's copy stream, behind the work already
 * queued on the batch and ahead of whatever is queued next; the call does not wait.  For a
 * streaming consumer: the next frames of one batch arrive while another batch is estimated. */
tdk_status tdk_dvo_upload_async(tdk_dvo *h, int which, int first_pair, int n_pairs,
                                const double *pinned_host);
/* The same for 8-bit grey frames as a camera delivers them: [n_pairs][height * width] bytes in
 * pinned memory, converted on the device to the float64 image skimage's img_as_float /
 * rgb2gray-of-grey gives: x * (1 / 255), the product with the rounded reciprocal (skimage/util/dtype.py), which
 * is not x / 255 for 24 of the 256 values.  An eighth of the PCIe bytes of the float64 hand-over. */
tdk_status tdk_dvo_upload_async_u8(tdk_dvo *h, int which, int first_pair, int n_pairs,
                                   const uint8_t *pinned_host);
/* Fills every pair on the device from the analytic synthetic scene of
 * tadataka_amd/synthetic.py (bench inputs "generated on device"): pair i uses
 * poses12[i] as ground truth and seed seed0+i for the noise. */
tdk_status tdk_dvo_fill_synthetic(tdk_dvo *h, const double *camera, const double *poses12,
                                  uint64_t seed0, double noise);
/* Builds pyramid levels 1..n_levels-1 (and level 0 of the arrays of tdk_dvo_set_rescale_options) on the device
 * from the uploaded frames. */
tdk_status tdk_dvo_build_pyramid(tdk_dvo *h);
/* The same for a subset of the arrays: bit 0 I0, bit 1 D0, bit 2 I1, bit 3 W0.  A consumer of a stream of frames
 * that replaces only I1 of every pair per step (tdk_dvo_upload_async*) rebuilds a third of the pyramid. */
tdk_status tdk_dvo_build_pyramid_arrays(tdk_dvo *h, unsigned int arrays);
/* Device -> host copy of one array of one pair/level: which = 0 I0, 1 D0, 2 I1, 3 W0. */
tdk_status tdk_dvo_download(tdk_dvo *h, int pair, int level, int which, double *out);
/* Levels WITHOUT a plan (below): with (1, the default) or without (0) the anti-aliasing prefilter, at the
 * ideal sample positions (tdk_rescale_anti_aliased / tdk_rescale). */
tdk_status tdk_dvo_set_anti_aliasing(tdk_dvo *h, int enabled);
/* skimage.transform.rescale to the bit for pyramid level `level` (0 .. n_levels - 1): the map and kernels of
 * tdk_rescale_skimage for (full-resolution shape -> shape of that level).  map == NULL removes the plan.
 * Takes effect at the next tdk_dvo_build_pyramid*. */
tdk_status tdk_dvo_set_level_plan(tdk_dvo *h, int level, const double *map, const double *w_rows, int radius_rows,
                                  const double *w_cols, int radius_cols);
/* level0_arrays (bit 0 I0, bit 1 D0, bit 2 I1, bit 3 W0): the arrays whose LEVEL 0 is built by the pyramid too --
 * the reference sends level 0 through rescale(image, 1.0) like every other level, and skimage's estimated map
 * for scale 1.0 is a few ulp from the identity, so that level differs from the frame by ~1e-13 at almost every
 * pixel (the uploaded frames are then kept beside level 0).  clip: skimage's clip=True for every level.
 * Defaults: 0, 0 (level 0 is the uploaded frame itself, nothing is clipped). */
tdk_status tdk_dvo_set_rescale_options(tdk_dvo *h, unsigned int level0_arrays, int clip);
tdk_status tdk_dvo_level_shape(tdk_dvo *h, int level, int *height, int *width);

/* One evaluation per pair at `level`:
 *   normal equations of calc_pose_update at poses12[pair]:
 *       Hout[pair][21] (upper triangle, row-major), bout[pair][6], n_update[pair]
 *   photometric error at the same pose: sum_sq[pair], n_error[pair]
 * cameras are [n_pairs][4] at FULL resolution (scaled per level internally as
 * tadataka.camera.resize does, camera/model.py:69-74).  Any output may be NULL. */
tdk_status tdk_dvo_evaluate(tdk_dvo *h, int level, const double *camera0, const double *camera1,
                            const double *poses12, int weight_mode, double *Hout, double *bout,
                            int64_t *n_update, double *sum_sq, int64_t *n_error);
/* PhotometricError.__call__ / photometric_error (tadataka/metric.py:13-39) for one pose per
 * pair: sum_sq[pair] = sum of (I0[u0] - I1<warp(u0)>)^2 over the pixels that land inside
 * image 1, n_error[pair] their count (the error is sum_sq / n_error).  Error-only pass: no
 * gradients, no Jacobian, no normal equations (the "probe" body of the evaluation kernel:
 * 24 B/px read, ~45 % of the arithmetic of tdk_dvo_evaluate).  Either output may be NULL. */
tdk_status tdk_dvo_photometric_error(tdk_dvo *h, int level, const double *camera0,
                                     const double *camera1, const double *poses12,
                                     double *sum_sq, int64_t *n_error);
/* Gauss-Newton loop of one level for all pairs, accept/reject on the device.
 * poses12 is updated in place; n_evals[pair] (optional) = evaluations used. */
tdk_status tdk_dvo_estimate_level(tdk_dvo *h, int level, const double *camera0,
                                  const double *camera1, double *poses12, int weight_mode,
                                  int max_iter, int *n_evals);
/* Coarse-to-fine over all levels (PoseChangeEstimator.__call__, :125-150). */
tdk_status tdk_dvo_estimate(tdk_dvo *h, const double *camera0, const double *camera1,
                            double *poses12, int weight_mode, int max_iter, int64_t *pixel_evals);
/* too_large[pair] = 1 if some evaluation of the last tdk_dvo_estimate /
 * tdk_dvo_estimate_level call found an empty update mask for that pair -- where the
 * reference warns "Camera pose change is too large." and returns the pose it had
 * (vo/dvo/__init__.py:97-100), at any iteration and any pyramid level. */
tdk_status tdk_dvo_get_warnings(tdk_dvo *h, int *too_large);
/* Work of the last tdk_dvo_estimate / tdk_dvo_estimate_level call in SURVEY 8(d)'s units, summed
 * over pairs and levels (source pixels of the level): error_pixels counts PhotometricError
 * evaluations (metric.py:36-39; n + 1 per level and pair), update_pixels counts calc_pose_update
 * calls whose normal equations were formed and solved (vo/dvo/__init__.py:46-70; n per level and
 * pair).  One "DVO iter" of the metric = one update + one error.  Either pointer may be NULL. */
tdk_status tdk_dvo_get_counts(tdk_dvo *h, int64_t *error_pixels, int64_t *update_pixels);
/* Diagnostic of the Tukey weights (weights.py:21-35): the two medians of an evaluation come from sampled
 * brackets + one pass over the residuals; a pair whose order statistics fall outside its brackets (or whose
 * bracket overflows) is redone by an exact radix select over its residual map.  *pairs = how often that
 * happened since the batch was created (the results are the same doubles either way). */
tdk_status tdk_dvo_get_tukey_fallbacks(tdk_dvo *h, int64_t *pairs);
/* Diagnostic of the Student-t weights (weights.py:4-16): fixed-point steps 2..10 of the variance come from two
 * passes over the residuals that expand every step around a predicted iterate (sample, then first pass); a pair
 * whose prediction was too far off for the remainder bound takes a third pass.  *pairs = how often that
 * happened since the batch was created. */
tdk_status tdk_dvo_get_student_redos(tdk_dvo *h, int64_t *pairs);
/* ... and a pair whose third pass still moved an expansion point by more than 1e-4 (a small update mask with gross
 * outliers: the sample's sequence tens of per cent off) takes the nine remaining steps one after the other, as
 * weights.py:13-16 does.  *pairs = how often that happened since the batch was created. */
tdk_status tdk_dvo_get_student_fallbacks(tdk_dvo *h, int64_t *pairs);
/* How the Student-t variance is iterated: 0 (default) the Taylor passes above; 1 the nine sequential passes
 * over the residuals (one per fixed-point step, reciprocal arithmetic); 2 the nine passes with IEEE divisions
 * (the CPU restatement's operations).  Defaults from TDK_STUDENT=sequential / TDK_STUDENT_EXACT=1 at creation. */
tdk_status tdk_dvo_set_student_passes(tdk_dvo *h, int mode);
/* Per-batch options.  Defaults are what the library would choose; the alternatives give the same results (the
 * tests compare them in-process) and exist to A/B a path or to force a rare one.
 *   TDK_DVO_OPT_CHAIN   1 (default): a batch of up to 2^22 pixels queues its whole coarse-to-fine chain at once and
 *                       waits once (speculative level chain); 0: the host drives level by level, round by round
 *   TDK_DVO_OPT_TUKEY   0 (default): the two medians of Tukey's scale from sampled brackets + one pass; 1: by radix
 *                       select; 2: brackets, but every pair takes the exact fallback of a failed bracket */
enum { TDK_DVO_OPT_CHAIN = 0, TDK_DVO_OPT_TUKEY = 1 };
tdk_status tdk_dvo_set_option(tdk_dvo *h, int option, int value);
/* The robust scale the last Student-t / Tukey evaluation of each pair used: the variance after ten steps
 * (weights.py:16) or c * MAD (weights.py:34).  scale: n_pairs doubles (host). */
tdk_status tdk_dvo_get_robust_scale(tdk_dvo *h, double *scale);
/* The hipStream_t every launch and copy of this batch is queued on.  Each batch
 * owns its stream: calls on different batches overlap on the device (e.g. the
 * HBM-bound pyramid of one batch under the FP64-bound estimation of another). */
tdk_status tdk_dvo_get_stream(tdk_dvo *h, void **stream_out);
/* Per-launch timing of the full-resolution evaluation kernel with HIP events
 * recorded inside the library, on its own stream, around each launch:
 * launches, summed milliseconds and summed source pixels since enabling. */
tdk_status tdk_dvo_set_profiling(tdk_dvo *h, int enabled);
tdk_status tdk_dvo_get_profile(tdk_dvo *h, int64_t *launches, double *total_ms, int64_t *pixels);
/* The device loop first PROBES a candidate pose (error only) and forms the normal
 * equations only for accepted candidates -- as the reference computes n updates and n + 1
 * errors per level.  kind 0: launches that evaluated only in full (what tdk_dvo_get_profile
 * returns), 1: only probes, 2: both kinds of pairs in one launch. */
tdk_status tdk_dvo_get_profile_kind(tdk_dvo *h, int kind, int64_t *launches, double *total_ms,
                                    int64_t *pixels);
/* tdk_dvo_set_profiling(h, 2) times the evaluation launches of EVERY pyramid level (1: the
 * finest level only); this returns the buckets of one level (kinds as above; pixels = that
 * level's source pixels x pairs evaluated).  Meant for a single batch alone on the device: with
 * another batch's pyramid running beside it the coarse-level launches are stretched. */
tdk_status tdk_dvo_get_profile_level(tdk_dvo *h, int level, int kind, int64_t *launches, double *total_ms,
                                     int64_t *pixels);

/* ---- least-squares pieces at the reference's own granularity -------------- */
/* A^T W A (upper triangle, row-major, p(p+1)/2) and A^T W b (p) of an n x p
 * system, p <= 8; w may be NULL.  The reduction behind
 * tadataka.math.solve_linear_equation (tadataka/math.py:32-45) and each IRLS
 * step of tadataka.irls.fit (tadataka/irls.py:186-218). */
tdk_status tdk_weighted_normal_equations(const double *A, const double *b, const double *w,
                                         int64_t n, int p, double *AtWA, double *AtWb);
/* calc_pose_update (tadataka/vo/dvo/__init__.py:46-70) on the arrays the
 * reference passes: residuals [n], GX1/GY1 [H,W], P1 [n,3]; weight_mode none /
 * huber / map (weights [n], indexed like residuals).  Returns the normal
 * equations and the mask count; n_valid == 0 is the reference's `return None`. */
tdk_status tdk_dvo_pose_update(const double *camera1, const double *residuals, const double *GX1,
                               const double *GY1, int height, int width, const double *P1,
                               int64_t n, int weight_mode, const double *weights, double *H21,
                               double *b6, int64_t *n_valid);
/* compute_weights_{huber,student_t,tukey} (tadataka/robust/weights.py:4-43) of a
 * residual vector: mode = TDK_W_HUBER / TDK_W_STUDENT_T / TDK_W_TUKEY. */
tdk_status tdk_robust_weights(const double *residuals, int64_t m, int mode, double *weights);
/* The same with the reference's keyword parameters: huber (p0 = k), student-t
 * (p0 = nu, p1 = n_iter), tukey (p0 = beta, p1 = c). */
tdk_status tdk_robust_weights_ex(const double *residuals, int64_t m, int mode, double p0, double p1,
                                 double *weights);

/* ---- semi-dense (rust_bindings.semi_dense) --------------------------------- */
typedef struct {
    double min_depth, max_depth, geo_coeff, photo_coeff, ref_step_size, min_gradient;
} tdk_semi_dense_params; /* Params.new, src/py/semi_dense.rs:93-108 */

/* increment_age (src/py/semi_dense.rs:35-51 -> src/semi_dense/age.rs:6-32) */
tdk_status tdk_increment_age(const uint64_t *age0, int height, int width, const double *camera0,
                             const double *camera1, const double *transform10,
                             const double *depth0, uint64_t *age1);
/* propagate (src/py/semi_dense.rs:189-218 -> src/semi_dense/propagation.rs:48-92) */
tdk_status tdk_propagate(const double *transform10, const double *camera0, const double *camera1,
                         const double *depth0, const double *variance0, int height, int width,
                         double default_depth, double default_variance, double uncertaintity_bias,
                         double *depth1, double *variance1);
/* update_depth (src/py/semi_dense.rs:157-187 -> src/semi_dense/semi_dense.rs:160-234).
 * ref_* hold n_ref frames back to back; outputs in Python order (depth, variance, flag). */
tdk_status tdk_update_depth(const double *key_camera, const double *key_image,
                            const double *key_transform_wf, int n_ref, const double *ref_cameras,
                            const double *ref_images, const double *ref_transforms_wf,
                            const uint64_t *age, const double *prior_depth,
                            const double *prior_variance, int height, int width,
                            const tdk_semi_dense_params *params, double *depth, double *variance,
                            int64_t *flag);
/* Frame (src/py/semi_dense.rs:53-91) keeps its image for as long as it lives and update_depth
 * borrows its refframes -- the example appends every frame to the list it passes
 * (examples/semi_dense_vo.py:199).  tdk_frame is that object on the device: the image is uploaded
 * once, tdk_update_depth_frames takes the frames by handle (same arithmetic and outputs as
 * tdk_update_depth; camera and pose still travel as small host arrays), so a call moves the
 * three maps in and the three results out and nothing per reference frame. */
typedef struct tdk_frame tdk_frame;
tdk_status tdk_frame_create(const double *image, int height, int width, tdk_frame **out);
tdk_status tdk_frame_destroy(tdk_frame *f);
tdk_status tdk_frame_download(const tdk_frame *f, double *image);   /* Frame.image getter */
tdk_status tdk_update_depth_frames(const double *key_camera, const tdk_frame *key_frame,
                                   const double *key_transform_wf, int n_ref,
                                   const double *ref_cameras, const tdk_frame *const *ref_frames,
                                   const double *ref_transforms_wf, const uint64_t *age,
                                   const double *prior_depth, const double *prior_variance,
                                   const tdk_semi_dense_params *params, double *depth,
                                   double *variance, int64_t *flag);
/* Device-resident maps.  The loop of examples/semi_dense_vo.py:182-199 hands every map a call
 * returns straight into the next call (age1 -> update_depth, depth_map1 / variance_map1 ->
 * update_depth -> next frame's increment_age / propagate / dvo).  A tdk_map is an H x W array of
 * 8-byte elements (float64, uint64 or int64 -- the caller knows which) that lives on the device;
 * the three operators below are the host-pointer entries above with maps in and out: same
 * arithmetic, asynchronous on the library stream, nothing crosses PCIe until tdk_map_download.
 * rust_bindings.semi_dense returns such maps wrapped as lazily downloaded array objects.
 *   tdk_map_create        host may be NULL (contents undefined until written)
 *   tdk_map_destroy       the buffer is recycled for the next map of the same size
 *   tdk_map_safe_invert   out = 1 / (v + epsilon)  (tadataka/numeric.py:1-2: the DVO weights of
 *                         examples/semi_dense_vo.py:52), out may be v itself
 *   tdk_update_depth_maps an age beyond n_ref must be reported by the call (TDK_ERR_AGE_EXCEEDS_REFFRAMES: the
 *                         reference exits the process there, semi_dense.rs:202-205).  The library keeps, on the
 *                         host, an upper bound of every map read as uint64 -- the maximum of an uploaded array,
 *                         + 1 per tdk_increment_age_maps -- and waits for the device's verdict only if that bound
 *                         exceeds n_ref: the loop of the example, whose refframes grow with the ages, never waits.
 *                         (A map written through tdk_map_device_ptr by the caller keeps its old bound: upload
 *                         through tdk_map_upload, or pass n_ref generously, if ages are edited that way.)
 *   tdk_frame_create      copies on its own stream and waits for that copy only, not for queued kernels */
typedef struct tdk_map tdk_map;
tdk_status tdk_map_create(int height, int width, const void *host, tdk_map **out);
tdk_status tdk_map_destroy(tdk_map *m);
tdk_status tdk_map_upload(tdk_map *m, const void *host);
tdk_status tdk_map_download(const tdk_map *m, void *host);
tdk_status tdk_map_shape(const tdk_map *m, int *height, int *width);
tdk_status tdk_map_device_ptr(const tdk_map *m, void **ptr);
tdk_status tdk_frame_device_ptr(const tdk_frame *f, void **ptr);
tdk_status tdk_map_safe_invert(const tdk_map *v, double epsilon, tdk_map *out);
tdk_status tdk_increment_age_maps(const tdk_map *age0, const double *camera0, const double *camera1,
                                  const double *transform10, const tdk_map *depth0, tdk_map *age1);
tdk_status tdk_propagate_maps(const double *transform10, const double *camera0, const double *camera1,
                              const tdk_map *depth0, const tdk_map *variance0, double default_depth,
                              double default_variance, double uncertaintity_bias, tdk_map *depth1,
                              tdk_map *variance1);
tdk_status tdk_update_depth_maps(const double *key_camera, const tdk_frame *key_frame,
                                 const double *key_transform_wf, int n_ref, const double *ref_cameras,
                                 const tdk_frame *const *ref_frames, const double *ref_transforms_wf,
                                 const tdk_map *age, const tdk_map *prior_depth,
                                 const tdk_map *prior_variance, const tdk_semi_dense_params *params,
                                 tdk_map *depth, tdk_map *variance, tdk_map *flag);
/* estimate_debug_ (src/py/semi_dense.rs:126-155) */
tdk_status tdk_estimate_one(const int64_t *u_key, double prior_depth, double prior_variance,
                            const double *key_camera, const double *key_image,
                            const double *key_transform_wf, const double *ref_camera,
                            const double *ref_image, const double *ref_transform_wf, int height,
                            int width, const tdk_semi_dense_params *params, double *depth,
                            double *variance, int64_t *flag);
/* Sobel maps of ImageGradient::new (src/semi_dense/gradient.rs:11-15) */
tdk_status tdk_sobel(const double *image, int height, int width, double *gx, double *gy);

/* Semi-dense post-steps.  Disabled in the reference's Python surface
 * (src/semi_dense/mod.rs:13, src/py/semi_dense.rs:220-233) but called by
 * examples/semi_dense_vo.py:82-88.
 * regularize (src/semi_dense/regularization.rs:29-64): 3x3 inverse-variance
 * weighted mean of the inverse depths of the Success (flag == 0) neighbours,
 * returned as a depth; pixels without a contributing neighbour keep their depth. */
tdk_status tdk_regularize(const double *depth, const double *variance, const int64_t *flag,
                          int height, int width, double *regularized);
/* fusion_arrays (src/semi_dense/fusion.rs:13-42): elementwise Gaussian fusion
 * mu = (mu1 var2 + mu2 var1) / (var1 + var2), var = var1 var2 / (var1 + var2). */
tdk_status tdk_fusion_arrays(const double *mu1, const double *mu2, const double *var1,
                             const double *var2, int64_t n, double *mu, double *var);
/* skimage.color.rgb2gray as the examples call it (examples/dvo_pose_change.py:22-31,
 * examples/semi_dense_vo.py:62-66): 0.2125 R + 0.7154 G + 0.0721 B of the first
 * three of `channels` interleaved channels; rgb is float64 [height][width][channels]
 * (tdk_rgb2gray) or uint8 multiplied by 1/255 first, as img_as_float does (tdk_rgb2gray_u8). */
tdk_status tdk_rgb2gray(const double *rgb, int height, int width, int channels, double *gray);
tdk_status tdk_rgb2gray_u8(const uint8_t *rgb, int height, int width, int channels, double *gray);

/* ---- semi-dense: device-resident session over a batch of tracks (new, fused) ----
 * One step of the mapping loop of examples/semi_dense_vo.py:182-199 for n_tracks
 * independent sequences at once, with every map and frame resident in HBM:
 *     age1            = increment_age(age0, cam0, cam1, T10, depth0)
 *     depth1, var1    = propagate(T10, cam0, cam1, depth0, var0, defaults...)
 *     depth, var, flag = update_depth(newest frame, earlier frames, age1, depth1, var1, params)
 * Frames live in a per-track ring of max_refframes + 1 images; the newest pushed
 * frame is the key frame of the step, the one before it is "frame 0" of the warp
 * and the frame `age` steps back is the reference frame of a pixel
 * (refframes[len - age], src/semi_dense/semi_dense.rs:207).
 *
 * The ring is the one place where a session departs from the reference's loop, which appends
 * every frame to an unbounded `refframes` list (examples/semi_dense_vo.py:199, "TODO remove
 * unused reference frames") while increment_age lets ages grow without bound (age.rs:28).  With
 * a bounded ring a pixel tracked for more than max_refframes steps would ask for a frame that is
 * gone.  tdk_sd_set_age_policy chooses what happens then:
 *   saturate = 0 (default since round 4: the reference's behaviour)
 *                           ages are not limited; a step in which some age exceeds the ring fails
 *                           with TDK_ERR_AGE_EXCEEDS_REFFRAMES (what update_depth does when
 *                           age > len(refframes), semi_dense.rs:202-205) and commits nothing --
 *                           every later step fails too, so size max_refframes for the whole track.
 *   saturate = 1 (opt-in)   increment_age saturates at the number of reference frames the next
 *                           update_depth will see, min(frames - 1, max_refframes): such a pixel
 *                           keeps using the OLDEST frame of the ring -- results then differ from the
 *                           reference's for tracks longer than max_refframes + 1 frames (identical
 *                           up to there). */
typedef struct tdk_sd tdk_sd;
tdk_status tdk_sd_create(int n_tracks, int height, int width, int max_refframes, tdk_sd **out);
tdk_status tdk_sd_set_age_policy(tdk_sd *h, int saturate);
/* The forward warp of a step (increment_age + propagate) runs as a gather over a small window of sources
 * where the track's displacement field fits one (see csrc/semi_dense.hip: k_sd_targets / k_sd_gather2) and
 * through per-target slot lists where it does not; same results either way.  tracks: how many (track, step)
 * warps of this session have taken the slot path so far.  TDK_SD_GATHER=0 disables the gather. */
tdk_status tdk_sd_get_warp_fallbacks(tdk_sd *h, int64_t *tracks);
tdk_status tdk_sd_destroy(tdk_sd *h);
/* Params.new (src/py/semi_dense.rs:93-108) + the three scalars of propagate */
tdk_status tdk_sd_set_params(tdk_sd *h, const tdk_semi_dense_params *params, double default_depth,
                             double default_variance, double uncertaintity_bias);
/* Host -> device / device -> host copies of one track's maps; any pointer may be NULL. */
tdk_status tdk_sd_set_maps(tdk_sd *h, int track, const double *depth, const double *variance,
                           const uint64_t *age);
tdk_status tdk_sd_get_maps(tdk_sd *h, int track, double *depth, double *variance, uint64_t *age,
                           int64_t *flag);
/* Frame.new (src/py/semi_dense.rs:53-66): appends a frame to the track's ring
 * (the oldest one is dropped when the ring is full).  transform_wf may be NULL and
 * supplied by the step that makes this frame its key frame. */
tdk_status tdk_sd_push_frame(tdk_sd *h, int track, const double *camera, const double *image,
                             const double *transform_wf);
/* The step described above for every track.  transforms10 [n_tracks][16] maps the
 * previous frame to the newest one; key_transforms_wf [n_tracks][16] (may be NULL
 * if given to tdk_sd_push_frame) is the pose of the newest frame.  commit = 0
 * leaves the session's maps untouched (the results can still be read with
 * tdk_sd_get_results); flag_histogram (optional) receives, per track, the number of
 * pixels with flag 0, -1, ..., -9.  TDK_ERR_AGE_EXCEEDS_REFFRAMES (nothing is
 * committed) where the reference exits the process. */
tdk_status tdk_sd_step(tdk_sd *h, const double *transforms10, const double *key_transforms_wf,
                       int commit, int64_t *flag_histogram);
/* The two halves of the step on their own, on the session's current maps:
 * tdk_sd_propagate = increment_age + propagate (results: age, depth, variance);
 * tdk_sd_update_depth = update_depth with the current (age, depth, variance) as
 * (age_map, prior_depth, prior_variance) and the newest frame as key frame. */
tdk_status tdk_sd_propagate(tdk_sd *h, const double *transforms10, int commit);
tdk_status tdk_sd_update_depth(tdk_sd *h, const double *key_transforms_wf, int commit,
                               int64_t *flag_histogram);
/* Outputs of the last step / propagate / update_depth call whether committed or not. */
tdk_status tdk_sd_get_results(tdk_sd *h, int track, double *depth, double *variance,
                              uint64_t *age, int64_t *flag);
/* Feeds the DVO batch of the same step on the device (examples/semi_dense_vo.py:44-53):
 * pair t gets I0 = the track's previous frame, D0 = its depth map, I1 = its newest
 * frame and, if the batch has a weight map, W0 = tadataka.numeric.safe_invert(variance)
 * = 1 / (variance + 1e-16). */
tdk_status tdk_sd_export_dvo(tdk_sd *h, tdk_dvo *batch);
/* Kernel times of the last step, measured with HIP events on the session's stream:
 * ms[0] scatter + fold (increment_age + propagate), ms[1] update_depth (classify +
 * estimate), ms[2] the whole step. */
tdk_status tdk_sd_get_timing(tdk_sd *h, double *ms3);

/* ---- bundle adjustment (tadataka.transform_project, tadataka/local_ba.py) -- */
/* Projection.compute / .jacobians over n observations (local_ba.py:23-39);
 * x_pred [n][2], A [n][2][6], B [n][2][3]; any output may be NULL. */
tdk_status tdk_ba_projection(const double *poses, int64_t n_poses, const double *points,
                             int64_t n_points, const int64_t *viewpoint_indices,
                             const int64_t *point_indices, int64_t n, double *x_pred, double *A,
                             double *B);
/* transform_project.exp_so3 (transform_project.pyx:46-50) for n rotation vectors */
tdk_status tdk_ba_exp_so3(const double *rotvecs, int64_t n, double *R);
/* Fused residual + Jacobian + block reduce (the sums sparseba.SBA.compute
 * starts from, call site local_ba.py:74-77): U [n_poses][21], ea [n_poses][6],
 * V [n_points][6], eb [n_points][3], err = sum ||x_true - x_pred||^2. */
tdk_status tdk_ba_block_reduce(const double *poses, int64_t n_poses, const double *points,
                               int64_t n_points, const double *x_true,
                               const int64_t *viewpoint_indices, const int64_t *point_indices,
                               int64_t n, double *U, double *ea, double *V, double *eb,
                               double *err);

/* Sparse bundle-adjustment step on a fixed observation graph: what
 * LocalBundleAdjustment.calc_update / calc_error obtain from the third-party
 * sparseba.SBA.compute in the reference (tadataka/local_ba.py:72-86).  The
 * handle keeps the index arrays, x_true and the per-point observation lists on
 * the device.  tdk_ba_step returns the Levenberg-Marquardt update for damping mu
 * (added to the diagonals of U_j and V_i) via the Schur complement on the pose
 * block, plus sum ||x_true - x_pred||^2 at the input parameters. */
typedef struct tdk_ba tdk_ba;
tdk_status tdk_ba_create(int64_t n_poses, int64_t n_points, const int64_t *viewpoint_indices,
                         const int64_t *point_indices, const double *x_true, int64_t n,
                         tdk_ba **out);
/* The same with options (bits): which of the library's kernels serve the handle is normally decided by its shape;
 * these bits force the alternatives (which exist for other shapes) onto any shape -- the tests run them on small
 * windows against the dense solve.
 *   TDK_BA_OPT_SCHUR_PAIRS    Schur complement by the pair-wise kernel (dense observation table) instead of the FP64
 *                             MFMA kernel that windows of <= 8 poses take; W_ij is then stored, not rebuilt
 *   TDK_BA_OPT_SCHUR_GENERAL  ... by the general kernel (per-point observation lists, atomics): what graphs too large
 *                             for the dense table take
 *   TDK_BA_OPT_SOLVE_HOST     the reduced camera system is solved on the host (what windows beyond 20 poses do)
 *   TDK_BA_OPT_SOLVE_PIVOTED  device solve: skip the attempt without pivoting */
enum { TDK_BA_OPT_SCHUR_PAIRS = 1, TDK_BA_OPT_SCHUR_GENERAL = 2, TDK_BA_OPT_SOLVE_HOST = 4, TDK_BA_OPT_SOLVE_PIVOTED = 8 };
tdk_status tdk_ba_create_ex(int64_t n_poses, int64_t n_points, const int64_t *viewpoint_indices,
                            const int64_t *point_indices, const double *x_true, int64_t n_observations,
                            unsigned int options, tdk_ba **out);
tdk_status tdk_ba_destroy(tdk_ba *h);
tdk_status tdk_ba_error(tdk_ba *h, const double *poses, const double *points, double *sum_sq);
/* The block sums of tdk_ba_block_reduce on the handle's graph, without atomics:
 * bit-reproducible for any observation order.  Any output may be NULL. */
tdk_status tdk_ba_block_sums(tdk_ba *h, const double *poses, const double *points, double *U,
                             double *ea, double *V, double *eb, double *sum_sq);
/* Per-kernel timing with HIP events on the library stream.  Index: 0 block reduce
 * (Jacobians + per-pose sums), 1 error-only reduce, 2 per-point sums, 3 Schur
 * complement, 4 back-substitution, 5 reduced camera system (device solve);
 * launches[6] and total_ms[6] since enabling. */
tdk_status tdk_ba_set_profiling(tdk_ba *h, int enabled);
tdk_status tdk_ba_get_profile(tdk_ba *h, int64_t *launches, double *total_ms);
tdk_status tdk_ba_step(tdk_ba *h, const double *poses, const double *points, double mu,
                       double *dposes, double *dpoints, double *sum_sq);
/* The whole Levenberg-Marquardt loop of LocalBundleAdjustment.compute
 * (tadataka/local_ba.py:91-134: damping trials mu/nu, mu, mu nu, ... per
 * iteration; stop on mean squared error < absolute_error_threshold or relative
 * change < relative_error_threshold) with poses [n_poses][6] and points
 * [n_points][3] resident on the device; both are updated in place.
 * error_history (optional, max_iter + 1 doubles): [0] the initial mean squared
 * error, [k] the error accepted by iteration k - 1; n_iter (optional): iterations run. */
tdk_status tdk_ba_solve(tdk_ba *h, double *poses, double *points, int max_iter, double initial_mu,
                        double nu, double absolute_error_threshold,
                        double relative_error_threshold, double *error_history, int *n_iter);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI ---------------------------
 * The reference has no distributed code; independent frame pairs shard across
 * ranks with no data-path collective and the recovered poses are all-gathered
 * (SURVEY section 8(e)).  librccl.so is opened on the first call.  Rank 0 creates
 * a 128-byte unique id (ncclGetUniqueId) and hands it to the other ranks out of
 * band (tadataka_amd/sharding.py uses a file next to the rendezvous port);
 * tdk_comm_create is collective (ncclCommInitRank) on the current device.
 * RANKS MUST BE SEPARATE PROCESSES: every entry of this library holds one
 * process-wide lock for its whole duration, the blocking collectives included
 * (tdk_comm_create, _all_gather, _all_reduce, _barrier,
 * tdk_dvo_gather_poses_finish) -- two ranks hosted as threads of one process
 * would deadlock (the first waits, lock held, for a peer that cannot enter). */
typedef struct tdk_comm tdk_comm;
tdk_status tdk_comm_available(void);   /* TDK_OK if librccl can be opened and has the symbols used here */
tdk_status tdk_comm_unique_id(uint8_t *id128);
tdk_status tdk_comm_create(const uint8_t *id128, int rank, int world, tdk_comm **out);
tdk_status tdk_comm_destroy(tdk_comm *c);
tdk_status tdk_comm_rank(tdk_comm *c, int *rank, int *world);
/* Host-buffer collectives on float64 (staged through the device; blocking):
 * recv holds world * count doubles in rank order; op: 0 sum, 1 max. */
tdk_status tdk_comm_all_gather(tdk_comm *c, const double *send, int64_t count, double *recv);
tdk_status tdk_comm_all_reduce(tdk_comm *c, double *values, int64_t count, int op);
tdk_status tdk_comm_barrier(tdk_comm *c);
/* ncclAllGather of the batch's device-resident poses [n_pairs][12] (the result of the
 * last tdk_dvo_estimate / _estimate_level), queued on the batch's own stream right
 * behind the estimation; _finish waits for it and copies [world * n_pairs][12] out.
 * One gather in flight per communicator. */
tdk_status tdk_dvo_gather_poses_start(tdk_dvo *h, tdk_comm *c);
tdk_status tdk_dvo_gather_poses_finish(tdk_comm *c, double *poses_all);

#ifdef __cplusplus
}
#endif
#endif /* TADATAKA_HIP_H */
