/*
 * tdk_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Plain-C restatement of the Tadataka per-pixel DVO / semi-dense / BA hot
 * path.  Every function cites the reference file:line it follows (paths are
 * relative to the reference checkout).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product path
 * (tadataka_amd, libtadataka_hip.so) never links or calls it.
 *
 * Conventions: all arrays are C-contiguous float64; images are [row=y][col=x];
 * coordinates are (x, y) pairs; cam = {fx, fy, ox, oy}; T = row-major 4x4.
 */
#ifndef TDK_ORACLE_H
#define TDK_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- per-point geometry ------------------------------------------------ */
void orc_normalize(const double *kp, int64_t n, const double *cam, double *out);
void orc_unnormalize(const double *kp, int64_t n, const double *cam, double *out);
void orc_project_vecs(const double *P, int64_t n, double *out);
void orc_inv_project_vecs(const double *xs, const double *depths, int64_t n, double *out);
void orc_transform(const double *T, const double *P, int64_t n, double *out);
void orc_warp_vecs(const double *T10, const double *xs, const double *depths,
                   int64_t n, double *out_xs, double *out_depths);
int orc_interpolation(const double *image, int H, int W, const double *coords,
                      int64_t m, double *out);
double orc_calc_depth0(const double *T10, const double *x0, const double *x1);
void orc_is_in_image_range(const double *kp, int64_t n, int H, int W, uint8_t *mask);

/* ---- DVO ---------------------------------------------------------------- */
void orc_image_gradient(const double *I, int H, int W, double *GX, double *GY);

/* weight modes for orc_dvo_* (tadataka/vo/dvo/__init__.py:32-39,63-70) */
enum { ORC_W_NONE = 0, ORC_W_HUBER = 1, ORC_W_STUDENT_T = 2, ORC_W_TUKEY = 3,
       ORC_W_MAP = 4 };

/* Masked Jacobian rows of one calc_pose_update call.  Jout (N x 6), rout (N),
 * wout (N) are caller-allocated at full size N = H*W; returns M = number of
 * masked pixels written. */
int64_t orc_dvo_rows(const double *I0, const double *D0, const double *I1,
                     const double *GX1, const double *GY1, const double *W0,
                     int H, int W, const double *cam0, const double *cam1,
                     const double *R, const double *t, int weight_mode,
                     double *Jout, double *rout, double *wout);

/* Same pass, reduced to the weighted normal equations: Hout[21] upper
 * triangle row-major of sum w J^T J, bout[6] = sum w J^T r. */
int64_t orc_dvo_normal_equations(const double *I0, const double *D0,
                                 const double *I1, const double *GX1,
                                 const double *GY1, const double *W0, int H,
                                 int W, const double *cam0, const double *cam1,
                                 const double *R, const double *t,
                                 int weight_mode, double *Hout, double *bout);

/* tadataka/metric.py:13-27.  Returns the number of in-range pixels, writes the
 * sum of squared differences (mean = sum / count). */
int64_t orc_photometric_error(const double *I0, const double *D0,
                              const double *I1, int H, int W,
                              const double *cam0, const double *cam1,
                              const double *T10, double *sum_sq);

/* Pyramid level (the build's stand-in for skimage.transform.rescale, parity
 * unpinned -- see DESIGN.md).  Output shape is (Ho, Wo). */
void orc_rescale_bilinear(const double *src, int H, int W, double *dst, int Ho, int Wo);
/* skimage.transform.rescale with anti_aliasing=True (its 0.15+ default): Gaussian
 * prefilter (scipy.ndimage.gaussian_filter, mode 'mirror') + the bilinear warp */
void orc_gaussian_weights(double sigma, int radius, double *w);
int orc_gaussian_radius(double sigma);
void orc_gaussian_filter_mirror(const double *src, int H, int W, const double *wr, int Rr,
                                const double *wc, int Rc, double *dst);
void orc_rescale_anti_aliased(const double *src, int H, int W, double *dst, int Ho, int Wo);
/* skimage.transform.rescale (0.18.3) with the interpreter-dependent constants as inputs: map = (ax, bx, ay, by) the
 * estimated affine map, wr / wc scipy's Gaussian kernels (2 R + 1 entries, NULL = axis not filtered), clip as clip=True */
void orc_rescale_skimage(const double *src, int H, int W, double *dst, int Ho, int Wo, const double *map,
                         const double *wr, int Rr, const double *wc, int Rc, int clip);

/* ---- semi-dense ---------------------------------------------------------- */
typedef struct {
    double inv_depth_min;  /* inv(max_depth) */
    double inv_depth_max;  /* inv(min_depth) */
    double geo_coeff;
    double photo_coeff;
    double ref_step_size;
    double min_gradient;
} orc_params;

void orc_make_params(double min_depth, double max_depth, double geo_coeff,
                     double photo_coeff, double ref_step_size,
                     double min_gradient, orc_params *out);
void orc_sobel(const double *img, int H, int W, double *gx, double *gy);
void orc_increment_age(const uint64_t *age0, int H, int W, const double *cam0,
                       const double *cam1, const double *T10,
                       const double *depth0, uint64_t *age1);
void orc_propagate(const double *T10, const double *cam0, const double *cam1,
                   const double *depth0, const double *var0, int H, int W,
                   double default_depth, double default_variance,
                   double uncertaintity_bias, double *depth1, double *var1);
void orc_transform_rk(const double *T_wk, const double *T_wr, double *T_rk);
/* One pixel (src/py/semi_dense.rs:126-155).  Returns the flag. */
int64_t orc_estimate_debug(const int64_t *u_key, double prior_depth,
                           double prior_variance, const double *key_cam,
                           const double *key_image, const double *key_T,
                           const double *ref_cam, const double *ref_image,
                           const double *ref_T, int H, int W,
                           const orc_params *params, double *out_depth,
                           double *out_variance);
/* src/semi_dense/semi_dense.rs:160-234.  Returns 0, or -1 if some age exceeds
 * n_ref (the reference calls process::exit(1) there). */
int orc_update_depth(const double *key_cam, const double *key_image,
                     const double *key_T, int n_ref, const double *ref_cams,
                     const double *ref_images, const double *ref_Ts,
                     const uint64_t *age, const double *prior_depth,
                     const double *prior_variance, int H, int W,
                     const orc_params *params, double *out_depth,
                     double *out_variance, int64_t *out_flag);
int orc_update_depth_trk(const double *key_cam, const double *key_image,
                         const double *key_T, int n_ref, const double *ref_cams,
                         const double *ref_images, const double *ref_Ts,
                         const double *T_rks_in,
                         const uint64_t *age, const double *prior_depth,
                         const double *prior_variance, int H, int W,
                         const orc_params *params, double *out_depth,
                         double *out_variance, int64_t *out_flag);

/* ---- semi-dense post-steps and colour conversion ---------------- */
int orc_regularize_patch(const double *inv_depth, const double *inv_variance,
                         const int64_t *flag, double *out);
void orc_regularize(const double *depth, const double *variance,
                    const int64_t *flag, int H, int W, double *out);
void orc_fusion_arrays(const double *mu1, const double *mu2, const double *var1,
                       const double *var2, int64_t n, double *mu, double *var);
void orc_rgb2gray(const double *rgb, int64_t n, int channels, double *out);

/* ---- bundle adjustment ---------------------------------------------------- */
void orc_exp_so3(const double *rotvec, double *R);
void orc_ba_transform_project(const double *pose, const double *point, double *out);
void orc_ba_pose_jacobian(const double *pose, const double *point, double *out);
void orc_ba_point_jacobian(const double *pose, const double *point, double *out);
void orc_ba_projection(const double *poses, const double *points,
                       const int64_t *vp_idx, const int64_t *pt_idx, int64_t n,
                       double *x_pred, double *A, double *B);
/* Per-viewpoint / per-point normal-equation blocks of the SBA formulation
 * (call site tadataka/local_ba.py:74-77).  U[n_poses][21], ea[n_poses][6],
 * V[n_points][6], eb[n_points][3]; returns sum of squared residuals. */
double orc_ba_block_reduce(const double *poses, int64_t n_poses,
                           const double *points, int64_t n_points,
                           const double *x_true, const int64_t *vp_idx,
                           const int64_t *pt_idx, int64_t n, double *U,
                           double *ea, double *V, double *eb);

#ifdef __cplusplus
}
#endif
#endif
