// semi_dense.hip -- rust_bindings.semi_dense on the MI355X: increment_age,
// propagate, update_depth / estimate_debug_ and the Sobel maps they use.
//
// Compiled with -ffp-contract=off: the warped target pixel is an *index* and the
// per-pixel result a discrete flag, so every + - * / sqrt is kept as one IEEE
// rounding in the reference's operation order (bit-exact against the oracle).
//
// The two forward-warp scatters are order dependent in the reference (a raster
// loop): increment_age keeps the LAST raster writer (src/semi_dense/age.rs:18-29)
// and propagate folds colliding sources SEQUENTIALLY in raster order with a
// non-associative rule (src/semi_dense/propagation.rs:21-46,59-82).  Both are
// reproduced deterministically:
//   * age:       atomicMax of the source raster index per target, then a gather;
//   * propagate: sources are threaded onto a per-target linked list with
//                atomicExch; one thread per target then folds its list in
//                increasing source index (selection over the short list).
#include "tdk_math.h"
#include "tdk_runtime.h"

#include <math.h>
#include <string.h>

#include <vector>

namespace {

using tdk::Cam;

constexpr int kBlock = 256;

inline int grid_for(int64_t n) {
    int64_t g = (n + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if (g > 1 << 20) g = 1 << 20;
    return (int)g;
}

struct Mat4 {
    double m[16];
};

// ---- Sobel (src/gradient.rs:4-26, src/convolution.rs:29-52) -----------------
__global__ __launch_bounds__(kBlock) void k_sobel(const double *__restrict__ img, int H, int W,
                                                  double *__restrict__ gx, double *__restrict__ gy) {
    const double kx[9] = {1., 0., -1., 2., 0., -2., 1., 0., -1.};
    const double ky[9] = {1., 2., 1., 0., 0., 0., -1., -2., -1.};
    int64_t N = (int64_t)H * W;
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < N; i += (int64_t)gridDim.x * kBlock) {
        int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
        double sx = 0.0, sy = 0.0;
        if (y >= 1 && y <= H - 2 && x >= 1 && x <= W - 2) {
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int b = 0; b < 3; b++) {
                    double v = img[(int64_t)(y - 1 + a) * W + (x - 1 + b)];
                    sx += kx[3 * a + b] * v;
                    sy += ky[3 * a + b] * v;
                }
        }
        gx[i] = sx;
        gy[i] = sy;
    }
}

// ---- increment_age (src/semi_dense/age.rs:6-32) -------------------------------
__global__ __launch_bounds__(kBlock) void k_age_scatter(int H, int W, Cam c0, Cam c1, Mat4 T,
                                                        const double *__restrict__ depth0,
                                                        int *__restrict__ winner) {
    int N = H * W;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += gridDim.x * kBlock) {
        int y0 = i / W, x0 = i - y0 * W;
        double px, py, d1;
        tdk::perspective_warp(T.m, c0, c1, (double)x0, (double)y0, depth0[i], px, py, d1);
        if (!tdk::in_range(px, py, H, W)) continue;
        int x1 = (int)px, y1 = (int)py;  // `as usize`: truncation
        atomicMax(&winner[y1 * W + x1], i + 1);
    }
}

__global__ __launch_bounds__(kBlock) void k_age_gather(int N, const uint64_t *__restrict__ age0,
                                                       const int *__restrict__ winner,
                                                       uint64_t *__restrict__ age1) {
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += gridDim.x * kBlock) {
        int w = winner[i];
        age1[i] = w > 0 ? age0[w - 1] + 1 : 0;
    }
}

// ---- propagate (src/semi_dense/propagation.rs) ----------------------------------
__device__ __forceinline__ double propagate_variance(double depth0, double depth1, double variance0,
                                                     double uncertaintity) {
    double ratio = tdk::safe_inv(depth1) / tdk::safe_inv(depth0);  // :16-18
    double r2 = ratio * ratio;
    return (r2 * r2) * variance0 + uncertaintity;
}

__device__ __forceinline__ bool is_statically_same(double id1, double id2, double variance) {
    double ds = (id1 - id2) * (id1 - id2);  // src/semi_dense/stat.rs:5-15
    double fs = 2.0 * 2.0;
    return ds <= fs * variance;
}

// handle_collision (:21-46) with fusion (src/semi_dense/fusion.rs:3-11)
__device__ __forceinline__ void handle_collision(double depth_a, double depth_b, double var_a, double var_b,
                                                 double &d, double &v) {
    double ida = tdk::safe_inv(depth_a), idb = tdk::safe_inv(depth_b);
    if (is_statically_same(ida, idb, var_a) && is_statically_same(ida, idb, var_b)) {
        double vs = var_a + var_b;
        double mu = (ida * var_b + idb * var_a) / vs;
        double var = (var_a * var_b) / vs;
        d = tdk::safe_inv(mu);
        v = var;
        return;
    }
    if (depth_a < depth_b) { d = depth_a; v = var_a; }
    else { d = depth_b; v = var_b; }
}

__global__ __launch_bounds__(kBlock) void k_propagate_scatter(int H, int W, Cam c0, Cam c1, Mat4 T,
                                                              const double *__restrict__ depth0,
                                                              const double *__restrict__ var0, double bias,
                                                              double *__restrict__ d1a,
                                                              double *__restrict__ v1a,
                                                              int *__restrict__ head,
                                                              int *__restrict__ next) {
    int N = H * W;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += gridDim.x * kBlock) {
        int y0 = i / W, x0 = i - y0 * W;
        double d0 = depth0[i];
        double ux, uy, d1;
        tdk::perspective_warp(T.m, c0, c1, (double)x0, (double)y0, d0, ux, uy, d1);
        if (!tdk::in_range(ux, uy, H, W)) { next[i] = -2; continue; }
        d1a[i] = d1;
        v1a[i] = propagate_variance(d0, d1, var0[i], bias);
        int t = (int)uy * W + (int)ux;
        next[i] = atomicExch(&head[t], i);
    }
}

__global__ __launch_bounds__(kBlock) void k_propagate_fold(int N, const int *__restrict__ head,
                                                           const int *__restrict__ next,
                                                           const double *__restrict__ d1a,
                                                           const double *__restrict__ v1a,
                                                           double default_depth, double default_variance,
                                                           double *__restrict__ depth1,
                                                           double *__restrict__ var1) {
    for (int t = blockIdx.x * kBlock + threadIdx.x; t < N; t += gridDim.x * kBlock) {
        int h = head[t];
        double d = default_depth, v = default_variance;
        int last = -1;
        bool have = false;
        while (true) {
            // next source in raster order: smallest list entry greater than `last`
            int best = 0x7fffffff;
            for (int j = h; j >= 0; j = next[j])
                if (j > last && j < best) best = j;
            if (best == 0x7fffffff) break;
            if (!have) { d = d1a[best]; v = v1a[best]; have = true; }
            else {
                double nd, nv;
                handle_collision(d1a[best], d, v1a[best], v, nd, nv);
                d = nd; v = nv;
            }
            last = best;
        }
        depth1[t] = d;
        var1[t] = v;
    }
}

// ---- update_depth / estimate (src/semi_dense/semi_dense.rs) ----------------------
struct RefConst {           // per reference frame, precomputed on the host
    double T_rk[16];        // inv(T_wr) T_wk (:83-89)
    double cam[4];
    double e_key[2];        // calc_key_epipole (epipolar.rs:9-20)
    double pt_rk[2];        // project(t_rk), for geo_var (variance.rs:45-52)
    const double *image;
};

struct EstParams {
    double vmin, vmax;      // inv_depth_range
    double geo_coeff, photo_coeff, ref_step, min_gradient;
};

__device__ __forceinline__ double norm2(double a, double b) { return sqrt(a * a + b * b); }

__device__ __forceinline__ void vnormalize2(double &a, double &b) {  // src/vector.rs:4-11
    double n = norm2(a, b);
    if (n == 0.) return;
    a = a / n;
    b = b / n;
}

__device__ __forceinline__ int check_args(double inv_depth, double variance, double vmin, double vmax) {
    if (inv_depth <= 0.) return -7;  // hypothesis.rs:15-37
    double mn = inv_depth - 2.0 * variance, mx = inv_depth + 2.0 * variance;
    if (mx <= vmin || vmax <= mn) return -1;
    return 0;
}

__device__ __forceinline__ double clampd(double v, double mn, double mx) {  // src/cmp.rs:3-12
    if (v < mn) return mn;
    if (v > mx) return mx;
    return v;
}

constexpr int kMaxRefSamples = 1 << 22;

// Exact-branch bilinear (src/interpolation.rs:9-43) through the clamped form.
__device__ __forceinline__ double sample(const double *img, int H, int W, double x, double y) {
    return tdk::bilinear(img, H, W, x, y);
}

// estimate (:91-158).  Returns the Flag (0 = Success) and writes (inv_depth, variance).
__device__ int estimate(double ukx, double uky, double prior_id, double prior_var, const Cam &kc,
                        const double *__restrict__ key_image, const RefConst &rf, int H, int W,
                        const double *__restrict__ gx, const double *__restrict__ gy,
                        const EstParams &pr, double &out_id, double &out_var) {
    const double *T = rf.T_rk;
    const Cam rc{rf.cam[0], rf.cam[1], rf.cam[2], rf.cam[3]};
    // prior.range() (hypothesis.rs:54-61) -> depth_search_range (depth.rs:25-30)
    double rmin = clampd(prior_id - 2.0 * prior_var, pr.vmin, pr.vmax);
    double rmax = clampd(prior_id + 2.0 * prior_var, pr.vmin, pr.vmax);
    double min_depth = tdk::safe_inv(rmax), max_depth = tdk::safe_inv(rmin);

    double xk, yk;
    tdk::normalize(kc, ukx, uky, xk, yk);

    // step_ratio (:27-40), calc_ref_depth (depth.rs:6-15)
    double key_depth0 = tdk::safe_inv(prior_id);
    double ref_depth = ((T[8] * (xk * key_depth0) + T[9] * (yk * key_depth0)) + T[10] * (1.0 * key_depth0)) + T[11];
    if (ref_depth <= 0.) return -8;
    double ratio = prior_id / tdk::safe_inv(ref_depth);
    double key_step = ratio * pr.ref_step;

    // calc_ref_ends (:51-60)
    double xmin_x, xmin_y, xmax_x, xmax_y, dtmp;
    tdk::warp(T, xk, yk, min_depth, xmin_x, xmin_y, dtmp);
    tdk::warp(T, xk, yk, max_depth, xmax_x, xmax_y, dtmp);
    double rdx = xmax_x - xmin_x, rdy = xmax_y - xmin_y;

    // calc_key_direction (:42-49)
    double kdx = xk - rf.e_key[0], kdy = yk - rf.e_key[1];
    if (!(rdx * kdx + rdy * kdy > 0.)) { kdx = -kdx; kdy = -kdy; }

    // key_coordinates (epipolar.rs:22-36) -> unnormalize -> all_in_range
    vnormalize2(kdx, kdy);
    double key_I[5];
    {
        double ux[5], uy[5];
#pragma unroll
        for (int i = 0; i < 5; i++) {
            double s = key_step * (double)(i - 2);
            tdk::unnormalize(kc, xk + s * kdx, yk + s * kdy, ux[i], uy[i]);
        }
#pragma unroll
        for (int i = 0; i < 5; i++)
            if (!tdk::in_range(ux[i], uy[i], H, W)) return -2;
#pragma unroll
        for (int i = 0; i < 5; i++) key_I[i] = sample(key_image, H, W, ux[i], uy[i]);
    }
    double g2 = 0.0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        double d = key_I[i + 1] - key_I[i];
        g2 += d * d;
    }
    double key_gradient = sqrt(g2);
    if (key_gradient < pr.min_gradient) return -6;

    // ref_coordinates (epipolar.rs:38-54), check_us_ref (:62-81)
    double rnorm = norm2(rdx, rdy);
    double dirx = rdx / (rnorm + tdk::kEps16), diry = rdy / (rnorm + tdk::kEps16);
    double nf = rnorm / pr.ref_step;
    int n = 0;
    if (nf >= 0.) n = nf < (double)kMaxRefSamples ? (int)nf : kMaxRefSamples;
    if (n < 5) return -5;
    {
        double ux, uy;
        tdk::unnormalize(rc, xmin_x + (0.0 * pr.ref_step) * dirx, xmin_y + (0.0 * pr.ref_step) * diry, ux, uy);
        if (!tdk::in_range(ux, uy, H, W)) return -3;
        double s = (double)(n - 1) * pr.ref_step;
        tdk::unnormalize(rc, xmin_x + s * dirx, xmin_y + s * diry, ux, uy);
        if (!tdk::in_range(ux, uy, H, W)) return -4;
    }

    // intensities::search (intensities.rs:6-36): sliding 5-window, first minimum
    double kn[5];
    {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 5; i++) s += key_I[i] * key_I[i];
        double nn = sqrt(s);
#pragma unroll
        for (int i = 0; i < 5; i++) kn[i] = (nn == 0.) ? key_I[i] : key_I[i] / nn;
    }
    double w0 = 0, w1 = 0, w2 = 0, w3 = 0, w4 = 0;
    double min_err = INFINITY;
    int argmin = 0;
    for (int i = 0; i < n; i++) {
        double s = (double)i * pr.ref_step, ux, uy;
        tdk::unnormalize(rc, xmin_x + s * dirx, xmin_y + s * diry, ux, uy);
        w0 = w1; w1 = w2; w2 = w3; w3 = w4;
        w4 = sample(rf.image, H, W, ux, uy);
        if (i < 4) continue;
        double q = ((((w0 * w0 + w1 * w1) + w2 * w2) + w3 * w3) + w4 * w4);
        double sn = sqrt(q);
        double a0 = w0, a1 = w1, a2 = w2, a3 = w3, a4 = w4;
        if (sn != 0.) { a0 = w0 / sn; a1 = w1 / sn; a2 = w2 / sn; a3 = w3 / sn; a4 = w4 / sn; }
        double d0 = a0 - kn[0], d1 = a1 - kn[1], d2 = a2 - kn[2], d3 = a3 - kn[3], d4 = a4 - kn[4];
        double e = ((((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3) + d4 * d4);
        if (e < min_err) { min_err = e; argmin = i - 4; }
    }
    argmin += 2;

    // calc_key_depth (depth.rs:17-23)
    double sa = (double)argmin * pr.ref_step;
    double key_depth = tdk::calc_depth0(T, xk, yk, xmin_x + sa * dirx, xmin_y + sa * diry);

    // calc_alpha (variance.rs:54-105)
    double adx = rdx, ady = rdy;
    vnormalize2(adx, ady);
    double xrx, xry;
    tdk::warp(T, xk, yk, key_depth, xrx, xry, dtmp);
    int ai = fabs(adx) > fabs(ady) ? 0 : 1;
    double alpha;
    {
        const double *ri = &T[4 * ai], *rz = &T[8];
        double ti = T[4 * ai + 3], tz = T[11];
        double rzy = (rz[0] * xk + rz[1] * yk) + rz[2] * 1.0;
        double riy = (ri[0] * xk + ri[1] * yk) + ri[2] * 1.0;
        double dd = rzy * ti - riy * tz;
        double nn = (ai == 0 ? xrx : xry) * tz - ti;
        alpha = (ai == 0 ? adx : ady) * dd / (nn * nn);
    }

    // geo_var (variance.rs:30-52) with ImageGradient::get (gradient.rs:17-25)
    double edx = xk - rf.pt_rk[0], edy = yk - rf.pt_rk[1];
    vnormalize2(edx, edy);
    double igx = sample(gx, H, W, ukx, uky), igy = sample(gy, H, W, ukx, uky);
    vnormalize2(igx, igy);
    double p = edx * igx + edy * igy;
    double geo = (p == 0.) ? 1. / tdk::kEps16 : 1. / (p * p);
    double photo = 2. / (key_gradient / key_step);  // variance.rs:26-28, semi_dense.rs:153
    double a2 = alpha * alpha;
    double gg = pr.geo_coeff * pr.geo_coeff, pp = pr.photo_coeff * pr.photo_coeff;
    double variance = a2 * (gg * geo + pp * photo);  // variance.rs:15-24

    double id = tdk::safe_inv(key_depth);
    int f = check_args(id, variance, pr.vmin, pr.vmax);
    if (f) return f;
    out_id = id;
    out_var = variance;
    return 0;
}

// update_depth raster loop (:186-229), one thread per pixel.
__global__ __launch_bounds__(kBlock) void k_update_depth(Cam kc, const double *__restrict__ key_image,
                                                         const double *__restrict__ gx,
                                                         const double *__restrict__ gy, int n_ref,
                                                         const RefConst *__restrict__ refs,
                                                         const uint64_t *__restrict__ age,
                                                         const double *__restrict__ prior_depth,
                                                         const double *__restrict__ prior_var, int H, int W,
                                                         EstParams pr, double *__restrict__ out_depth,
                                                         double *__restrict__ out_var,
                                                         int64_t *__restrict__ out_flag) {
    int N = H * W;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += gridDim.x * kBlock) {
        uint64_t a = age[i];
        double d = prior_depth[i], v = prior_var[i];
        if (a == 0) {
            out_depth[i] = d; out_var[i] = v; out_flag[i] = -9;  // NotProcessed
            continue;
        }
        double pid = tdk::safe_inv(d);
        int f = check_args(pid, v, pr.vmin, pr.vmax);
        if (f) {
            out_depth[i] = d; out_var[i] = v; out_flag[i] = f;
            continue;
        }
        int y = i / W, x = i - y * W;
        const RefConst &rf = refs[n_ref - (int)a];
        double id = pid, var = v;
        f = estimate((double)x, (double)y, pid, v, kc, key_image, rf, H, W, gx, gy, pr, id, var);
        if (f) { id = pid; var = v; }  // Err(flag) => (prior, flag)
        out_depth[i] = tdk::safe_inv(id);
        out_var[i] = var;
        out_flag[i] = f;
    }
}

__global__ void k_estimate_one(Cam kc, const double *key_image, const double *gx, const double *gy,
                               const RefConst *refs, double ukx, double uky, double pid, double pvar, int H,
                               int W, EstParams pr, double *out /*[id, var, flag]*/) {
    double id = 0, var = 0;
    int f = estimate(ukx, uky, pid, pvar, kc, key_image, refs[0], H, W, gx, gy, pr, id, var);
    out[0] = id;
    out[1] = var;
    out[2] = (double)f;
}

// ---- host helpers ---------------------------------------------------------------

// General 4x4 inverse, Gauss-Jordan with partial pivoting (the reference calls
// LAPACK through ndarray-linalg, src/semi_dense/semi_dense.rs:83-89).
int inv4(const double *A, double *Ainv) {
    double M[4][8];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) { M[i][j] = A[4 * i + j]; M[i][4 + j] = (i == j); }
    for (int c = 0; c < 4; c++) {
        int p = c;
        for (int r = c + 1; r < 4; r++) if (fabs(M[r][c]) > fabs(M[p][c])) p = r;
        if (M[p][c] == 0.0) return -1;
        if (p != c) for (int j = 0; j < 8; j++) { double s = M[c][j]; M[c][j] = M[p][j]; M[p][j] = s; }
        double piv = M[c][c];
        for (int j = 0; j < 8; j++) M[c][j] /= piv;
        for (int r = 0; r < 4; r++) {
            if (r == c) continue;
            double f = M[r][c];
            if (f == 0.0) continue;
            for (int j = 0; j < 8; j++) M[r][j] -= f * M[c][j];
        }
    }
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) Ainv[4 * i + j] = M[i][4 + j];
    return 0;
}

tdk_status make_ref_const(const double *T_wk, const double *T_wr, const double *cam, const double *image_dev,
                          RefConst *rc) {
    double T_rw[16];
    if (inv4(T_wr, T_rw) != 0) {
        tdk::set_error("reference frame transform is singular");
        return TDK_ERR_SINGULAR;
    }
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0.0;
            for (int k = 0; k < 4; k++) s += T_rw[4 * i + k] * T_wk[4 * k + j];
            rc->T_rk[4 * i + j] = s;
        }
    for (int k = 0; k < 4; k++) rc->cam[k] = cam[k];
    // e_key = project(R_wk^T (t_wr - t_wk))
    double dt[3] = {T_wr[3] - T_wk[3], T_wr[7] - T_wk[7], T_wr[11] - T_wk[11]};
    double pe[3];
    for (int i = 0; i < 3; i++) pe[i] = (T_wk[i] * dt[0] + T_wk[4 + i] * dt[1]) + T_wk[8 + i] * dt[2];
    tdk::project(pe[0], pe[1], pe[2], rc->e_key[0], rc->e_key[1]);
    tdk::project(rc->T_rk[3], rc->T_rk[7], rc->T_rk[11], rc->pt_rk[0], rc->pt_rk[1]);
    rc->image = image_dev;
    return TDK_OK;
}

EstParams est_params(const tdk_semi_dense_params *p) {
    EstParams e;
    e.vmin = tdk::safe_inv(p->max_depth);  // src/py/semi_dense.rs:103
    e.vmax = tdk::safe_inv(p->min_depth);
    e.geo_coeff = p->geo_coeff;
    e.photo_coeff = p->photo_coeff;
    e.ref_step = p->ref_step_size;
    e.min_gradient = p->min_gradient;
    return e;
}

Cam cam_of(const double *c) { return Cam{c[0], c[1], c[2], c[3]}; }

Mat4 mat_of(const double *T) {
    Mat4 m;
    for (int i = 0; i < 16; i++) m.m[i] = T[i];
    return m;
}

tdk_status h2d(int slot, const void *host, size_t bytes, void **dev) {
    TDK_TRY(tdk::scratch(slot, bytes, dev));
    if (bytes) TDK_HIP(hipMemcpyAsync(*dev, host, bytes, hipMemcpyHostToDevice, tdk::stream()));
    return TDK_OK;
}

tdk_status check_image_dims(int H, int W) {
    TDK_REQUIRE(H >= 1 && W >= 1 && (int64_t)H * W < (1ll << 30), "bad image size");
    return TDK_OK;
}

}  // namespace

extern "C" {

tdk_status tdk_sobel(const double *image, int H, int W, double *gx, double *gy) {
    TDK_REQUIRE(image && gx && gy, "null pointer");
    TDK_TRY(check_image_dims(H, W));
    size_t bytes = (size_t)H * W * 8;
    void *d_img, *d_gx, *d_gy;
    TDK_TRY(h2d(0, image, bytes, &d_img));
    TDK_TRY(tdk::scratch(1, bytes, &d_gx));
    TDK_TRY(tdk::scratch(2, bytes, &d_gy));
    k_sobel<<<grid_for((int64_t)H * W), kBlock, 0, tdk::stream()>>>((const double *)d_img, H, W, (double *)d_gx,
                                                                    (double *)d_gy);
    TDK_LAUNCH_CHECK();
    TDK_HIP(hipMemcpyAsync(gx, d_gx, bytes, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipMemcpyAsync(gy, d_gy, bytes, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipStreamSynchronize(tdk::stream()));
    return TDK_OK;
}

tdk_status tdk_increment_age(const uint64_t *age0, int H, int W, const double *camera0, const double *camera1,
                             const double *T10, const double *depth0, uint64_t *age1) {
    TDK_REQUIRE(age0 && camera0 && camera1 && T10 && depth0 && age1, "null pointer");
    TDK_TRY(check_image_dims(H, W));
    int N = H * W;
    void *d_age0, *d_depth, *d_winner, *d_age1;
    TDK_TRY(h2d(0, age0, (size_t)N * 8, &d_age0));
    TDK_TRY(h2d(1, depth0, (size_t)N * 8, &d_depth));
    TDK_TRY(tdk::scratch(2, (size_t)N * 4, &d_winner));
    TDK_TRY(tdk::scratch(3, (size_t)N * 8, &d_age1));
    TDK_HIP(hipMemsetAsync(d_winner, 0, (size_t)N * 4, tdk::stream()));
    k_age_scatter<<<grid_for(N), kBlock, 0, tdk::stream()>>>(H, W, cam_of(camera0), cam_of(camera1), mat_of(T10),
                                                             (const double *)d_depth, (int *)d_winner);
    TDK_LAUNCH_CHECK();
    k_age_gather<<<grid_for(N), kBlock, 0, tdk::stream()>>>(N, (const uint64_t *)d_age0, (const int *)d_winner,
                                                            (uint64_t *)d_age1);
    TDK_LAUNCH_CHECK();
    TDK_HIP(hipMemcpyAsync(age1, d_age1, (size_t)N * 8, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipStreamSynchronize(tdk::stream()));
    return TDK_OK;
}

tdk_status tdk_propagate(const double *T10, const double *camera0, const double *camera1, const double *depth0,
                         const double *variance0, int H, int W, double default_depth, double default_variance,
                         double uncertaintity_bias, double *depth1, double *variance1) {
    TDK_REQUIRE(T10 && camera0 && camera1 && depth0 && variance0 && depth1 && variance1, "null pointer");
    TDK_TRY(check_image_dims(H, W));
    int N = H * W;
    size_t b8 = (size_t)N * 8, b4 = (size_t)N * 4;
    void *d_d0, *d_v0, *d_d1a, *d_v1a, *d_head, *d_next, *d_d1, *d_v1;
    TDK_TRY(h2d(0, depth0, b8, &d_d0));
    TDK_TRY(h2d(1, variance0, b8, &d_v0));
    TDK_TRY(tdk::scratch(2, b8, &d_d1a));
    TDK_TRY(tdk::scratch(3, b8, &d_v1a));
    TDK_TRY(tdk::scratch(4, b4, &d_head));
    TDK_TRY(tdk::scratch(5, b4, &d_next));
    TDK_TRY(tdk::scratch(6, b8, &d_d1));
    TDK_TRY(tdk::scratch(7, b8, &d_v1));
    TDK_HIP(hipMemsetAsync(d_head, 0xff, b4, tdk::stream()));  // -1: empty list
    k_propagate_scatter<<<grid_for(N), kBlock, 0, tdk::stream()>>>(
        H, W, cam_of(camera0), cam_of(camera1), mat_of(T10), (const double *)d_d0, (const double *)d_v0,
        uncertaintity_bias, (double *)d_d1a, (double *)d_v1a, (int *)d_head, (int *)d_next);
    TDK_LAUNCH_CHECK();
    k_propagate_fold<<<grid_for(N), kBlock, 0, tdk::stream()>>>(N, (const int *)d_head, (const int *)d_next,
                                                                (const double *)d_d1a, (const double *)d_v1a,
                                                                default_depth, default_variance, (double *)d_d1,
                                                                (double *)d_v1);
    TDK_LAUNCH_CHECK();
    TDK_HIP(hipMemcpyAsync(depth1, d_d1, b8, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipMemcpyAsync(variance1, d_v1, b8, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipStreamSynchronize(tdk::stream()));
    return TDK_OK;
}

tdk_status tdk_update_depth(const double *key_camera, const double *key_image, const double *key_T, int n_ref,
                            const double *ref_cameras, const double *ref_images, const double *ref_Ts,
                            const uint64_t *age, const double *prior_depth, const double *prior_variance, int H,
                            int W, const tdk_semi_dense_params *params, double *depth, double *variance,
                            int64_t *flag) {
    TDK_REQUIRE(key_camera && key_image && key_T && age && prior_depth && prior_variance && params && depth &&
                    variance && flag && n_ref >= 0,
                "bad argument");
    TDK_REQUIRE(n_ref == 0 || (ref_cameras && ref_images && ref_Ts), "null reference frames");
    TDK_TRY(check_image_dims(H, W));
    int N = H * W;
    // the reference exits the process if some age exceeds len(refframes) (:202-205)
    for (int i = 0; i < N; i++)
        if (age[i] > (uint64_t)n_ref) {
            tdk::set_error("Age exceeds the refframe size");
            return TDK_ERR_AGE_EXCEEDS_REFFRAMES;
        }
    size_t b8 = (size_t)N * 8;
    void *d_key, *d_gx, *d_gy, *d_refs_img, *d_age, *d_pd, *d_pv, *d_od, *d_ov, *d_of, *d_rc;
    TDK_TRY(h2d(0, key_image, b8, &d_key));
    TDK_TRY(tdk::scratch(1, b8, &d_gx));
    TDK_TRY(tdk::scratch(2, b8, &d_gy));
    TDK_TRY(h2d(3, ref_images, b8 * (size_t)n_ref, &d_refs_img));
    TDK_TRY(h2d(4, age, b8, &d_age));
    TDK_TRY(h2d(5, prior_depth, b8, &d_pd));
    TDK_TRY(h2d(6, prior_variance, b8, &d_pv));
    TDK_TRY(tdk::scratch(7, b8, &d_od));
    TDK_TRY(tdk::scratch(8, b8, &d_ov));
    TDK_TRY(tdk::scratch(9, b8, &d_of));
    std::vector<RefConst> rcs((size_t)(n_ref > 0 ? n_ref : 1));
    for (int r = 0; r < n_ref; r++)
        TDK_TRY(make_ref_const(key_T, ref_Ts + 16 * r, ref_cameras + 4 * r,
                               (const double *)d_refs_img + (size_t)r * N, &rcs[r]));
    TDK_TRY(h2d(10, rcs.data(), sizeof(RefConst) * rcs.size(), &d_rc));
    k_sobel<<<grid_for(N), kBlock, 0, tdk::stream()>>>((const double *)d_key, H, W, (double *)d_gx, (double *)d_gy);
    TDK_LAUNCH_CHECK();
    k_update_depth<<<grid_for(N), kBlock, 0, tdk::stream()>>>(
        cam_of(key_camera), (const double *)d_key, (const double *)d_gx, (const double *)d_gy, n_ref,
        (const RefConst *)d_rc, (const uint64_t *)d_age, (const double *)d_pd, (const double *)d_pv, H, W,
        est_params(params), (double *)d_od, (double *)d_ov, (int64_t *)d_of);
    TDK_LAUNCH_CHECK();
    TDK_HIP(hipMemcpyAsync(depth, d_od, b8, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipMemcpyAsync(variance, d_ov, b8, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipMemcpyAsync(flag, d_of, b8, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipStreamSynchronize(tdk::stream()));  // rcs must outlive the H2D copy
    return TDK_OK;
}

tdk_status tdk_estimate_one(const int64_t *u_key, double prior_depth, double prior_variance,
                            const double *key_camera, const double *key_image, const double *key_T,
                            const double *ref_camera, const double *ref_image, const double *ref_T, int H, int W,
                            const tdk_semi_dense_params *params, double *depth, double *variance, int64_t *flag) {
    TDK_REQUIRE(u_key && key_camera && key_image && key_T && ref_camera && ref_image && ref_T && params &&
                    depth && variance && flag,
                "null pointer");
    TDK_TRY(check_image_dims(H, W));
    EstParams pr = est_params(params);
    *depth = prior_depth;
    *variance = prior_variance;
    // check_args first (src/py/semi_dense.rs:137-142), on the host: two comparisons
    double pid = tdk::safe_inv(prior_depth);
    if (pid <= 0.) { *flag = -7; return TDK_OK; }
    {
        double mn = pid - 2.0 * prior_variance, mx = pid + 2.0 * prior_variance;
        if (mx <= pr.vmin || pr.vmax <= mn) { *flag = -1; return TDK_OK; }
    }
    size_t b8 = (size_t)H * W * 8;
    void *d_key, *d_gx, *d_gy, *d_ref, *d_rc, *d_out;
    TDK_TRY(h2d(0, key_image, b8, &d_key));
    TDK_TRY(tdk::scratch(1, b8, &d_gx));
    TDK_TRY(tdk::scratch(2, b8, &d_gy));
    TDK_TRY(h2d(3, ref_image, b8, &d_ref));
    RefConst rc;
    TDK_TRY(make_ref_const(key_T, ref_T, ref_camera, (const double *)d_ref, &rc));
    TDK_TRY(h2d(10, &rc, sizeof(RefConst), &d_rc));
    TDK_TRY(tdk::scratch(7, 3 * 8, &d_out));
    k_sobel<<<grid_for((int64_t)H * W), kBlock, 0, tdk::stream()>>>((const double *)d_key, H, W, (double *)d_gx,
                                                                    (double *)d_gy);
    TDK_LAUNCH_CHECK();
    k_estimate_one<<<1, 1, 0, tdk::stream()>>>(cam_of(key_camera), (const double *)d_key, (const double *)d_gx,
                                               (const double *)d_gy, (const RefConst *)d_rc, (double)u_key[0],
                                               (double)u_key[1], pid, prior_variance, H, W, pr, (double *)d_out);
    TDK_LAUNCH_CHECK();
    double out[3];
    TDK_HIP(hipMemcpyAsync(out, d_out, sizeof(out), hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipStreamSynchronize(tdk::stream()));
    *flag = (int64_t)out[2];
    if (*flag == 0) {
        *depth = tdk::safe_inv(out[0]);
        *variance = out[1];
    }
    return TDK_OK;
}

}  // extern "C"
