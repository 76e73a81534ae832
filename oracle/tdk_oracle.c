/*
 * tdk_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Plain-C, single-threaded restatement of the Tadataka DVO / semi-dense / BA
 * hot path, written from the reference's arithmetic (file:line cited per
 * function; paths relative to the reference checkout).  It is the checker the
 * HIP path is compared against and the timed "port" CPU baseline.  Nothing in
 * tadataka_amd/ may call into this file.
 *
 * Pinned against: the reference's own Rust #[test] / pytest literals, the
 * reference's _bilinear.cpp compiled as oracle/_ref/libref_bilinear.so, and
 * fixtures (tests/golden/, made by tests/golden/generate_golden.py in the
 * build container) captured from the reference's unmodified Python DVO
 * orchestration and from its sympy-generated Cython transform_project.
 * and, since round 5, fixtures captured from the reference's PoseChangeEstimator run
 * on the REAL scikit-image 0.18.3 (tests/golden/generate_golden_skimage.py under the
 * build container's /opt/conda interpreter): orc_rescale_skimage is bit-identical with
 * skimage.transform.rescale given the interpreter-dependent constants it takes as
 * inputs, the whole coarse-to-fine loop reproduces the reference to 1e-16.
 * NOT pinned by any reference output (no Rust toolchain, missing dataset depth maps):
 * the numeric results of semi-dense estimate/update_depth beyond the Rust unit-test
 * literals and the reference's own flag assertions -- "parity unpinned" for those,
 * see DESIGN.md 3; and sparseba's damping convention (the package is absent).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off so that +,-,*,/ and
 * sqrt are evaluated exactly as written, one IEEE rounding each).
 */
#include "tdk_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define EPS16 1e-16                   /* src/projection.rs:4, so3_codegen.py:45 */
#define EPSM 2.220446049250313e-16    /* f64::EPSILON, src/numeric.rs:1 */

/* ========================================================================
 * Per-point geometry
 * ===================================================================== */

/* src/camera.rs:36-41, tadataka/camera/_normalizer.cpp:16-17 */
static inline void normalize1(const double *cam, double ux, double uy,
                              double *x, double *y) {
    *x = (ux - cam[2]) / cam[0];
    *y = (uy - cam[3]) / cam[1];
}

/* src/camera.rs:43-48, tadataka/camera/_normalizer.cpp:25-26 */
static inline void unnormalize1(const double *cam, double x, double y,
                                double *ux, double *uy) {
    *ux = x * cam[0] + cam[2];
    *uy = y * cam[1] + cam[3];
}

void orc_normalize(const double *kp, int64_t n, const double *cam, double *out) {
    for (int64_t i = 0; i < n; i++)
        normalize1(cam, kp[2 * i], kp[2 * i + 1], &out[2 * i], &out[2 * i + 1]);
}

void orc_unnormalize(const double *kp, int64_t n, const double *cam, double *out) {
    for (int64_t i = 0; i < n; i++)
        unnormalize1(cam, kp[2 * i], kp[2 * i + 1], &out[2 * i], &out[2 * i + 1]);
}

/* src/projection.rs:11-14 */
static inline void project1(const double *p, double *x, double *y) {
    double z = p[2] + EPS16;
    *x = p[0] / z;
    *y = p[1] / z;
}

/* src/projection.rs:16-18 (to_homogeneous(x) * depth) */
static inline void inv_project1(double x, double y, double d, double *p) {
    p[0] = x * d;
    p[1] = y * d;
    p[2] = 1.0 * d;
}

void orc_project_vecs(const double *P, int64_t n, double *out) {
    for (int64_t i = 0; i < n; i++)
        project1(&P[3 * i], &out[2 * i], &out[2 * i + 1]);
}

void orc_inv_project_vecs(const double *xs, const double *depths, int64_t n,
                          double *out) {
    for (int64_t i = 0; i < n; i++)
        inv_project1(xs[2 * i], xs[2 * i + 1], depths[i], &out[3 * i]);
}

/* src/transform.rs:17-23: (T . [p;1])[0:3], k accumulated 0..3 */
static inline void transform1(const double *T, const double *p, double *q) {
    for (int i = 0; i < 3; i++) {
        const double *r = &T[4 * i];
        q[i] = ((r[0] * p[0] + r[1] * p[1]) + r[2] * p[2]) + r[3] * 1.0;
    }
}

void orc_transform(const double *T, const double *P, int64_t n, double *out) {
    for (int64_t i = 0; i < n; i++) transform1(T, &P[3 * i], &out[3 * i]);
}

/* src/warp.rs:11-29 (1-D), :31-50 (N-D) */
static inline void warp1(const double *T10, double x0, double y0, double d0,
                         double *x1, double *y1, double *d1) {
    double p0[3], p1[3];
    inv_project1(x0, y0, d0, p0);
    transform1(T10, p0, p1);
    project1(p1, x1, y1);
    *d1 = p1[2];
}

void orc_warp_vecs(const double *T10, const double *xs, const double *depths,
                   int64_t n, double *out_xs, double *out_depths) {
    for (int64_t i = 0; i < n; i++)
        warp1(T10, xs[2 * i], xs[2 * i + 1], depths[i], &out_xs[2 * i],
              &out_xs[2 * i + 1], &out_depths[i]);
}

/* src/warp.rs:70-86 PerspectiveWarp (1-D) */
static inline void perspective_warp1(const double *T10, const double *cam0,
                                     const double *cam1, double u0x, double u0y,
                                     double d0, double *u1x, double *u1y,
                                     double *d1) {
    double x0, y0, x1, y1;
    normalize1(cam0, u0x, u0y, &x0, &y0);
    warp1(T10, x0, y0, d0, &x1, &y1, d1);
    unnormalize1(cam1, x1, y1, u1x, u1y);
}

/* src/image_range.rs:11-17, tadataka/utils.py:35-44 */
static inline int in_range1(double x, double y, int H, int W) {
    double h = (double)H, w = (double)W;
    return 0. <= x && x <= w - 1. && 0. <= y && y <= h - 1.;
}

void orc_is_in_image_range(const double *kp, int64_t n, int H, int W,
                           uint8_t *mask) {
    for (int64_t i = 0; i < n; i++)
        mask[i] = (uint8_t)in_range1(kp[2 * i], kp[2 * i + 1], H, W);
}

/* src/interpolation.rs:9-43 (= tadataka/interpolation/_bilinear.cpp:5-35) */
static inline double interpolate1(const double *image, int W, double cx,
                                  double cy) {
    double lx = floor(cx), ly = floor(cy);
    int64_t lxi = (int64_t)lx, lyi = (int64_t)ly;

    if (lx == cx && ly == cy) return image[lyi * W + lxi];

    double ux = lx + 1.0, uy = ly + 1.0;
    int64_t uxi = (int64_t)ux, uyi = (int64_t)uy;

    if (lx == cx)
        return image[lyi * W + lxi] * (ux - cx) * (uy - cy) +
               image[uyi * W + lxi] * (ux - cx) * (cy - ly);

    if (ly == cy)
        return image[lyi * W + lxi] * (ux - cx) * (uy - cy) +
               image[lyi * W + uxi] * (cx - lx) * (uy - cy);

    return image[lyi * W + lxi] * (ux - cx) * (uy - cy) +
           image[lyi * W + uxi] * (cx - lx) * (uy - cy) +
           image[uyi * W + lxi] * (ux - cx) * (cy - ly) +
           image[uyi * W + uxi] * (cx - lx) * (cy - ly);
}

/* tadataka/interpolation/__init__.py:13-29: range check first (ValueError in
 * the reference -> return -1 here), then src/py/interpolation.rs:6-15. */
int orc_interpolation(const double *image, int H, int W, const double *coords,
                      int64_t m, double *out) {
    for (int64_t i = 0; i < m; i++)
        if (!in_range1(coords[2 * i], coords[2 * i + 1], H, W)) return -1;
    for (int64_t i = 0; i < m; i++)
        out[i] = interpolate1(image, W, coords[2 * i], coords[2 * i + 1]);
    return 0;
}

/* src/triangulation.rs:8-39 */
double orc_calc_depth0(const double *T10, const double *x0, const double *x1) {
    int i = fabs(T10[3]) > fabs(T10[7]) ? 0 : 1;
    const double *ri = &T10[4 * i], *rz = &T10[8];
    double ti = T10[4 * i + 3], tz = T10[11];
    double y0[3] = {x0[0], x0[1], 1.0};
    double n = ti - tz * x1[i];
    double rzy = (rz[0] * y0[0] + rz[1] * y0[1]) + rz[2] * y0[2];
    double riy = (ri[0] * y0[0] + ri[1] * y0[1]) + ri[2] * y0[2];
    double d = rzy * x1[i] - riy;
    return n / (d + EPS16);
}

/* ========================================================================
 * DVO (tadataka/vo/dvo/__init__.py, jacobian.py, metric.py, robust/weights.py)
 * ===================================================================== */

/* np.gradient as used by tadataka/vo/dvo/jacobian.py:27-29; returns (DX, DY) */
void orc_image_gradient(const double *I, int H, int W, double *GX, double *GY) {
    for (int y = 0; y < H; y++) {
        const double *row = &I[(int64_t)y * W];
        double *g = &GX[(int64_t)y * W];
        if (W == 1) { g[0] = 0.0; continue; }
        g[0] = (row[1] - row[0]) / 1.0;
        for (int x = 1; x < W - 1; x++) g[x] = (row[x + 1] - row[x - 1]) / 2.0;
        g[W - 1] = (row[W - 1] - row[W - 2]) / 1.0;
    }
    for (int x = 0; x < W; x++) {
        if (H == 1) { GY[x] = 0.0; continue; }
        GY[x] = (I[(int64_t)W + x] - I[x]) / 1.0;
        for (int y = 1; y < H - 1; y++)
            GY[(int64_t)y * W + x] =
                (I[(int64_t)(y + 1) * W + x] - I[(int64_t)(y - 1) * W + x]) / 2.0;
        GY[(int64_t)(H - 1) * W + x] =
            (I[(int64_t)(H - 1) * W + x] - I[(int64_t)(H - 2) * W + x]) / 1.0;
    }
}

static int cmp_double(const void *a, const void *b) {
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

/* np.median */
static double median_of(const double *v, int64_t n, double *scratch) {
    memcpy(scratch, v, (size_t)n * sizeof(double));
    qsort(scratch, (size_t)n, sizeof(double), cmp_double);
    if (n % 2 == 1) return scratch[n / 2];
    return (scratch[n / 2 - 1] + scratch[n / 2]) / 2.0;
}

/* tadataka/robust/weights.py:4-43 on the masked residual vector r (M).
 * Writes the weight vector exactly as compute_weights returns it. */
static void robust_weights(int mode, const double *r, int64_t M, double *w) {
    if (mode == ORC_W_HUBER) { /* :38-43 */
        const double k = 1.345;
        for (int64_t i = 0; i < M; i++) {
            double a = fabs(r[i]);
            w[i] = a > k ? k / a : 1.0;
        }
    } else if (mode == ORC_W_STUDENT_T) { /* :4-18 */
        const double nu = 5.0;
        double variance = 1.0;
        for (int it = 0; it < 10; it++) {
            double acc = 0.0;
            for (int64_t i = 0; i < M; i++) {
                double s = r[i] * r[i];
                acc += s * ((nu + 1.0) / (nu + s / variance));
            }
            variance = acc / (double)M;
        }
        for (int64_t i = 0; i < M; i++) {
            double s = r[i] * r[i];
            w[i] = sqrt((nu + 1.0) / (nu + s / variance));
        }
    } else if (mode == ORC_W_TUKEY) { /* :21-35 */
        const double beta = 4.6851, c = 1.4826;
        double *scratch = (double *)malloc((size_t)M * sizeof(double));
        double *dev = (double *)malloc((size_t)M * sizeof(double));
        double med = median_of(r, M, scratch);
        for (int64_t i = 0; i < M; i++) dev[i] = fabs(r[i] - med);
        double sigma = c * median_of(dev, M, scratch);
        for (int64_t i = 0; i < M; i++) {
            double x = r[i] / sigma;
            if (fabs(x) <= beta) {
                double q = x / beta;
                double u = 1.0 - q * q;
                w[i] = u * u;
            } else {
                w[i] = 0.0;
            }
        }
        free(scratch);
        free(dev);
    }
}

/* One calc_pose_update pass (tadataka/vo/dvo/__init__.py:46-70) preceded by
 * the per-level precomputation of _PoseChangeEstimator.__call__ (:86-90,94):
 *   us0 -> normalize (cam0) -> inv_pi(., D0) = P0;  P1 = R P0 + t
 *   us1 = unnormalize_cam1(pi(P1));  mask = in_range(us1) & (P1z > 0)
 *   J = calc_jacobian(f1, interp(GX1, us1), interp(GY1, us1), P1) (jacobian.py:8-24)
 *   r = (I0 - I1) at the SOURCE pixel (never re-warped, :90). */
int64_t orc_dvo_rows(const double *I0, const double *D0, const double *I1,
                     const double *GX1, const double *GY1, const double *W0,
                     int H, int W, const double *cam0, const double *cam1,
                     const double *R, const double *t, int weight_mode,
                     double *Jout, double *rout, double *wout) {
    int64_t M = 0;
    const double fx = cam1[0], fy = cam1[1];
    for (int y0 = 0; y0 < H; y0++) {
        for (int x0 = 0; x0 < W; x0++) {
            int64_t idx = (int64_t)y0 * W + x0;
            double xn, yn, P0[3], P1[3];
            normalize1(cam0, (double)x0, (double)y0, &xn, &yn);
            inv_project1(xn, yn, D0[idx], P0);
            /* tadataka/rigid_transform.py:115-123: np.dot(R, P.T).T + t */
            for (int i = 0; i < 3; i++)
                P1[i] = ((R[3 * i] * P0[0] + R[3 * i + 1] * P0[1]) +
                         R[3 * i + 2] * P0[2]) + t[i];
            double px, py, u1x, u1y;
            project1(P1, &px, &py);
            unnormalize1(cam1, px, py, &u1x, &u1y);
            if (!(in_range1(u1x, u1y, H, W) && P1[2] > 0)) continue;

            double gx = interpolate1(GX1, W, u1x, u1y);
            double gy = interpolate1(GY1, W, u1x, u1y);
            double fgx = fx * gx, fgy = fy * gy;
            double x = P1[0], yy = P1[1], z = P1[2];
            double z2 = z * z, xy = x * yy;
            double *J = &Jout[6 * M];
            J[0] = fgx / z;
            J[1] = fgy / z;
            J[2] = -(fgx * x + fgy * yy) / (z * z);
            J[3] = -(fgx * xy + fgy * (z2 + yy * yy)) / z2;
            J[4] = (fgx * (z2 + x * x) + fgy * xy) / z2;
            J[5] = (-fgx * yy + fgy * x) / z;
            rout[M] = I0[idx] - I1[idx];
            wout[M] = (weight_mode == ORC_W_MAP) ? W0[idx] : 1.0;
            M++;
        }
    }
    if (weight_mode == ORC_W_HUBER || weight_mode == ORC_W_STUDENT_T ||
        weight_mode == ORC_W_TUKEY)
        robust_weights(weight_mode, rout, M, wout);
    return M;
}

/* tadataka/math.py:32-45: rows scaled by sqrt(w) then lstsq  ==  normal
 * equations sum w J^T J xi = sum w J^T r. */
int64_t orc_dvo_normal_equations(const double *I0, const double *D0,
                                 const double *I1, const double *GX1,
                                 const double *GY1, const double *W0, int H,
                                 int W, const double *cam0, const double *cam1,
                                 const double *R, const double *t,
                                 int weight_mode, double *Hout, double *bout) {
    int64_t N = (int64_t)H * W;
    double *J = (double *)malloc((size_t)N * 6 * sizeof(double));
    double *r = (double *)malloc((size_t)N * sizeof(double));
    double *w = (double *)malloc((size_t)N * sizeof(double));
    int64_t M = orc_dvo_rows(I0, D0, I1, GX1, GY1, W0, H, W, cam0, cam1, R, t,
                             weight_mode, J, r, w);
    for (int k = 0; k < 21; k++) Hout[k] = 0.0;
    for (int k = 0; k < 6; k++) bout[k] = 0.0;
    for (int64_t m = 0; m < M; m++) {
        const double *j = &J[6 * m];
        /* solve_linear_equation takes sqrt(weights) of whatever
         * compute_weights returned, then squares it again in A^T A */
        double sw = sqrt(w[m]);
        double ww = sw * sw;
        int k = 0;
        for (int a = 0; a < 6; a++) {
            for (int b = a; b < 6; b++) Hout[k++] += ww * j[a] * j[b];
            bout[a] += ww * j[a] * r[m];
        }
    }
    free(J);
    free(r);
    free(w);
    return M;
}

/* tadataka/metric.py:13-27 with LocalWarp2D (tadataka/warp.py:78-88) */
int64_t orc_photometric_error(const double *I0, const double *D0,
                              const double *I1, int H, int W,
                              const double *cam0, const double *cam1,
                              const double *T10, double *sum_sq) {
    int64_t count = 0;
    double acc = 0.0;
    for (int y0 = 0; y0 < H; y0++) {
        for (int x0 = 0; x0 < W; x0++) {
            int64_t idx = (int64_t)y0 * W + x0;
            double u1x, u1y, d1;
            perspective_warp1(T10, cam0, cam1, (double)x0, (double)y0, D0[idx],
                              &u1x, &u1y, &d1);
            if (!in_range1(u1x, u1y, H, W)) continue; /* no z test: metric.py:22 */
            double d = I0[idx] - interpolate1(I1, W, u1x, u1y);
            acc += d * d;
            count++;
        }
    }
    *sum_sq = acc;
    return count;
}

/* _shared/interpolation.pxd coord_map, mode 'R' ("reflect" = numpy.pad 'reflect':
 * d c b | a b c d | c b a) */
static inline int64_t skimage_reflect(int64_t dim, int64_t coord) {
    int64_t cmax = dim - 1;
    if (dim == 1) return 0;
    if (coord < 0) {
        if (((-coord) / cmax) % 2 != 0) return cmax - ((-coord) % cmax);
        return (-coord) % cmax;
    }
    if (coord > cmax) {
        if ((coord / cmax) % 2 != 0) return cmax - (coord % cmax);
        return coord % cmax;
    }
    return coord;
}

/* skimage's warp of an (already filtered) image f: _warp_fast / bilinear_interpolation
 * (transform/_warps_cy.pyx, _shared/interpolation.pxd).  Sample positions: map != NULL
 * -> col = ax * ox + bx, row = ay * oy + by (_transform_metric: a product and a sum);
 * map == NULL -> the IDEAL positions (o + 0.5) * (in / out) - 0.5. */
static void warp_bilinear(const double *f, int H, int W, double *dst, int Ho, int Wo, const double *map) {
    const double sy = (double)H / (double)Ho, sx = (double)W / (double)Wo;
    for (int oy = 0; oy < Ho; oy++) {
        const double r = map ? map[2] * (double)oy + map[3] : ((double)oy + 0.5) * sy - 0.5;
        const int64_t minr = (int64_t)floor(r), maxr = (int64_t)ceil(r);
        const double dr = r - (double)minr;
        const int64_t r0 = skimage_reflect(H, minr), r1 = skimage_reflect(H, maxr);
        for (int ox = 0; ox < Wo; ox++) {
            const double c = map ? map[0] * (double)ox + map[1] : ((double)ox + 0.5) * sx - 0.5;
            const int64_t minc = (int64_t)floor(c), maxc = (int64_t)ceil(c);
            const double dc = c - (double)minc;
            const int64_t c0 = skimage_reflect(W, minc), c1 = skimage_reflect(W, maxc);
            const double top = (1 - dc) * f[r0 * W + c0] + dc * f[r0 * W + c1];
            const double bottom = (1 - dc) * f[r1 * W + c0] + dc * f[r1 * W + c1];
            dst[(int64_t)oy * Wo + ox] = (1 - dr) * top + dr * bottom;
        }
    }
}

/* rescale(image, scale, anti_aliasing=False) at the ideal sample positions, no clip */
void orc_rescale_bilinear(const double *src, int H, int W, double *dst, int Ho,
                          int Wo) {
    warp_bilinear(src, H, W, dst, Ho, Wo, NULL);
}

/* ------------------------------------------------------------------------
 * Anti-aliased rescale with the IDEAL constants (sample positions, libm kernels, no
 * clip): what skimage.transform.rescale does with its defaults (tadataka/vo/dvo/
 * __init__.py:144-148 calls it as rescale(image, scale)) up to the interpreter-
 * dependent last bits that orc_rescale_skimage below takes as inputs
 * (skimage/transform/_warps.py: resize()):
 *     factors = input_shape / output_shape            (per axis)
 *     sigma   = max(0, (factors - 1) / 2)
 *     image   = scipy.ndimage.gaussian_filter(image, sigma, mode='mirror')
 *               (skimage mode 'reflect' -> ndimage 'mirror'; truncate = 4)
 *     out     = bilinear warp with dst -> src: (i + 0.5) * factor - 0.5
 * The Gaussian part is pinned against scipy.ndimage itself in
 * tests/test_oracle_golden.py, the whole pipeline against scikit-image 0.18.3 in
 * tests/test_oracle_skimage.py.
 * --------------------------------------------------------------------- */

/* scipy.ndimage._filters._gaussian_kernel1d, order 0: exp(-0.5 / sigma^2 * x^2) / sum */
void orc_gaussian_weights(double sigma, int radius, double *w) {
    double sigma2 = sigma * sigma, sum = 0.0;
    for (int i = -radius; i <= radius; i++) {
        w[i + radius] = exp(-0.5 / sigma2 * (double)(i * i));
        sum += w[i + radius];
    }
    for (int i = 0; i <= 2 * radius; i++) w[i] = w[i] / sum;
}

/* int(truncate * sigma + 0.5), truncate = 4 (gaussian_filter1d) */
int orc_gaussian_radius(double sigma) { return (int)(4.0 * sigma + 0.5); }

/* ndimage 'mirror': d c b | a b c d | c b a */
static int mirror_idx(int64_t i, int n) {
    if (n == 1) return 0;
    int64_t p = 2 * ((int64_t)n - 1);
    i %= p;
    if (i < 0) i += p;
    if (i >= n) i = p - i;
    return (int)i;
}

/* One axis of scipy.ndimage.correlate1d with a symmetric kernel (ni_filters.c,
 * the `symmetric > 0` branch): centre tap first, then the pairs from the
 * outermost inwards, (x[-j] + x[+j]) * w[j].  w has 2 * radius + 1 entries. */
static double symmetric_tap(const double *line, int n, int64_t stride, int i, const double *w, int radius) {
    double tmp = line[(int64_t)i * stride] * w[radius];
    for (int j = -radius; j < 0; j++)
        tmp += (line[(int64_t)mirror_idx((int64_t)i + j, n) * stride] +
                line[(int64_t)mirror_idx((int64_t)i - j, n) * stride]) * w[radius + j];
    return tmp;
}

/* gaussian_filter(src, (sigma_r, sigma_c), mode='mirror'): axis 0 first, then
 * axis 1; an axis with sigma <= 1e-15 is skipped (scipy does the same). */
void orc_gaussian_filter_mirror(const double *src, int H, int W, const double *wr, int Rr,
                                const double *wc, int Rc, double *dst) {
    double *tmp = (double *)malloc(sizeof(double) * (size_t)H * W);
    if (wr) {
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) tmp[(int64_t)y * W + x] = symmetric_tap(src + x, H, W, y, wr, Rr);
    } else {
        memcpy(tmp, src, sizeof(double) * (size_t)H * W);
    }
    if (wc) {
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) dst[(int64_t)y * W + x] = symmetric_tap(tmp + (int64_t)y * W, W, 1, x, wc, Rc);
    } else {
        memcpy(dst, tmp, sizeof(double) * (size_t)H * W);
    }
    free(tmp);
}

void orc_rescale_anti_aliased(const double *src, int H, int W, double *dst, int Ho, int Wo) {
    double sr = ((double)H / (double)Ho - 1.0) / 2.0, sc = ((double)W / (double)Wo - 1.0) / 2.0;
    if (sr < 0.0) sr = 0.0;
    if (sc < 0.0) sc = 0.0;
    int Rr = orc_gaussian_radius(sr), Rc = orc_gaussian_radius(sc);
    double *wr = NULL, *wc = NULL;
    if (sr > 1e-15) { wr = (double *)malloc(sizeof(double) * (2 * Rr + 1)); orc_gaussian_weights(sr, Rr, wr); }
    if (sc > 1e-15) { wc = (double *)malloc(sizeof(double) * (2 * Rc + 1)); orc_gaussian_weights(sc, Rc, wc); }
    double *f = (double *)malloc(sizeof(double) * (size_t)H * W);
    orc_gaussian_filter_mirror(src, H, W, wr, Rr, wc, Rc, f);
    warp_bilinear(f, H, W, dst, Ho, Wo, NULL);
    free(f); free(wr); free(wc);
}

/* ------------------------------------------------------------------------
 * skimage.transform.rescale as scikit-image 0.18.3 executes it for a 2-D float64
 * image (order 1, mode 'reflect', clip=True: what tadataka/vo/dvo/__init__.py:
 * 144-148 calls for EVERY level, level 0 / scale 1.0 included), restated from the
 * installed package's sources (transform/_warps.py resize() :91-185, warp()
 * :826-930, _clip_warp_output; transform/_warps_cy.pyx _warp_fast /
 * _transform_metric; _shared/interpolation.pxd bilinear_interpolation /
 * coord_map mode 'R') and PINNED against that package run in the build container
 * (tests/golden/generate_golden_skimage.py -> tests/golden/skimage_*.npz).
 *
 * Two quantities in that pipeline are products of the interpreter's NumPy /
 * LAPACK / libm rather than of the algorithm, so they are INPUTS here:
 *   map = (ax, bx, ay, by): resize() ESTIMATES its affine map from three corner
 *         correspondences (AffineTransform.estimate: Hartley normalisation,
 *         numpy.linalg.svd, numpy.linalg.inv) instead of using
 *         factor, factor/2 - 1/2; the estimate is a few ulp off in the scale and
 *         ~1e-13 off in the offset, differently on every LAPACK build.  Sample
 *         positions are col = ax * ox + bx, row = ay * oy + by (a product and a
 *         sum, each rounded: _transform_metric).
 *   wr / wc: scipy.ndimage's Gaussian kernels, numpy.exp(-0.5 / sigma^2 * x^2)
 *         normalised by their numpy sum (2 R + 1 entries; NULL = that axis has
 *         sigma <= 1e-15 and is not filtered).  numpy's SIMD exp and pairwise
 *         sum differ from libm's exp and a sequential sum in the last bit.
 * oracle.py (skimage_plan) and the product's host code (tadataka_amd/
 * rescale_plan.py) compute both with the same NumPy calls skimage / scipy make,
 * or take them from a fixture that recorded what the generating interpreter got.
 * --------------------------------------------------------------------- */

/* numpy.clip(x, lo, hi) == minimum(maximum(x, lo), hi), NaN-propagating */
static inline double np_clip(double x, double lo, double hi) {
    double m = (x != x || lo != lo) ? NAN : (x > lo ? x : lo);
    return (m != m || hi != hi) ? NAN : (m < hi ? m : hi);
}

void orc_rescale_skimage(const double *src, int H, int W, double *dst, int Ho, int Wo, const double *map,
                         const double *wr, int Rr, const double *wc, int Rc, int clip) {
    /* resize(): image = ndi.gaussian_filter(image, sigma, mode='mirror') -- a plain copy when no axis is filtered */
    double *f = (double *)malloc(sizeof(double) * (size_t)H * W);
    orc_gaussian_filter_mirror(src, H, W, wr, Rr, wc, Rc, f);
    /* _clip_warp_output: bounds are the min / max of the image handed to warp(), i.e. of the FILTERED image
     * (ndarray.min / .max: NaN if any element is NaN) */
    double lo = f[0], hi = f[0];
    int has_nan = 0;
    for (int64_t i = 0; i < (int64_t)H * W; i++) {
        if (f[i] != f[i]) has_nan = 1;
        if (f[i] < lo) lo = f[i];
        if (f[i] > hi) hi = f[i];
    }
    if (has_nan) lo = hi = NAN;
    warp_bilinear(f, H, W, dst, Ho, Wo, map);
    if (clip)
        for (int64_t i = 0; i < (int64_t)Ho * Wo; i++) dst[i] = np_clip(dst[i], lo, hi);
    free(f);
}

/* ========================================================================
 * Semi-dense (src/semi_dense/ *.rs)
 * ===================================================================== */

/* src/numeric.rs:3-5, src/semi_dense/numeric.rs:17-27 */
static inline double safe_inv(double v) { return 1. / (v + EPSM); }

/* src/py/semi_dense.rs:93-108 */
void orc_make_params(double min_depth, double max_depth, double geo_coeff,
                     double photo_coeff, double ref_step_size,
                     double min_gradient, orc_params *out) {
    out->inv_depth_min = safe_inv(max_depth);
    out->inv_depth_max = safe_inv(min_depth);
    out->geo_coeff = geo_coeff;
    out->photo_coeff = photo_coeff;
    out->ref_step_size = ref_step_size;
    out->min_gradient = min_gradient;
}

/* src/gradient.rs:4-26 + src/convolution.rs:29-52: correlation with the 3x3
 * kernels, valid region written at a 1-px offset, border left at zero.
 * (kernel * window).sum() runs row-major over the 3x3 window. */
void orc_sobel(const double *img, int H, int W, double *gx, double *gy) {
    static const double kx[9] = {1., 0., -1., 2., 0., -2., 1., 0., -1.};
    static const double ky[9] = {1., 2., 1., 0., 0., 0., -1., -2., -1.};
    memset(gx, 0, (size_t)H * W * sizeof(double));
    memset(gy, 0, (size_t)H * W * sizeof(double));
    for (int y = 0; y + 2 < H; y++) {
        for (int x = 0; x + 2 < W; x++) {
            double sx = 0.0, sy = 0.0;
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) {
                    double v = img[(int64_t)(y + a) * W + (x + b)];
                    sx += kx[3 * a + b] * v;
                    sy += ky[3 * a + b] * v;
                }
            gx[(int64_t)(y + 1) * W + (x + 1)] = sx;
            gy[(int64_t)(y + 1) * W + (x + 1)] = sy;
        }
    }
}

/* src/semi_dense/age.rs:6-32 */
void orc_increment_age(const uint64_t *age0, int H, int W, const double *cam0,
                       const double *cam1, const double *T10,
                       const double *depth0, uint64_t *age1) {
    memset(age1, 0, (size_t)H * W * sizeof(uint64_t));
    for (int y0 = 0; y0 < H; y0++) {
        for (int x0 = 0; x0 < W; x0++) {
            int64_t idx = (int64_t)y0 * W + x0;
            double qx, qy, q1x, q1y, d1, px, py;
            normalize1(cam0, (double)x0, (double)y0, &qx, &qy);
            warp1(T10, qx, qy, depth0[idx], &q1x, &q1y, &d1);
            unnormalize1(cam1, q1x, q1y, &px, &py);
            if (!in_range1(px, py, H, W)) continue;
            int64_t x1 = (int64_t)px, y1 = (int64_t)py; /* `as usize` truncation */
            age1[y1 * W + x1] = age0[idx] + 1;
        }
    }
}

/* src/semi_dense/propagation.rs:9-19 */
static inline double propagate_variance(double depth0, double depth1,
                                        double variance0, double uncertaintity) {
    double ratio = safe_inv(depth1) / safe_inv(depth0);
    double r2 = ratio * ratio; /* powi(4) */
    return (r2 * r2) * variance0 + uncertaintity;
}

/* src/semi_dense/stat.rs:5-27 */
static inline int is_statically_same(double id1, double id2, double variance) {
    double ds = (id1 - id2) * (id1 - id2);
    double fs = 2.0 * 2.0;
    return ds <= fs * variance;
}

/* src/semi_dense/propagation.rs:21-46 + fusion.rs:3-11 */
static inline void handle_collision(double depth_a, double depth_b, double var_a,
                                    double var_b, double *d, double *v) {
    double ida = safe_inv(depth_a), idb = safe_inv(depth_b);
    if (is_statically_same(ida, idb, var_a) && is_statically_same(ida, idb, var_b)) {
        double vs = var_a + var_b;
        double mu = (ida * var_b + idb * var_a) / vs;
        double var = (var_a * var_b) / vs;
        *d = safe_inv(mu);
        *v = var;
        return;
    }
    if (depth_a < depth_b) { *d = depth_a; *v = var_a; }
    else { *d = depth_b; *v = var_b; }
}

/* src/semi_dense/propagation.rs:48-92.  The HashMap is replaced by a dense
 * "touched" map; the fold order (source raster order) is what matters. */
void orc_propagate(const double *T10, const double *cam0, const double *cam1,
                   const double *depth0, const double *var0, int H, int W,
                   double default_depth, double default_variance,
                   double uncertaintity_bias, double *depth1, double *var1) {
    int64_t N = (int64_t)H * W;
    uint8_t *touched = (uint8_t *)calloc((size_t)N, 1);
    for (int64_t i = 0; i < N; i++) { depth1[i] = default_depth; var1[i] = default_variance; }
    for (int y0 = 0; y0 < H; y0++) {
        for (int x0 = 0; x0 < W; x0++) {
            int64_t idx = (int64_t)y0 * W + x0;
            double d0 = depth0[idx];
            double u1x, u1y, d1a;
            perspective_warp1(T10, cam0, cam1, (double)x0, (double)y0, d0, &u1x,
                              &u1y, &d1a);
            if (!in_range1(u1x, u1y, H, W)) continue;
            double v1a = propagate_variance(d0, d1a, var0[idx], uncertaintity_bias);
            int64_t t = (int64_t)u1y * W + (int64_t)u1x;
            if (touched[t]) {
                double d, v;
                handle_collision(d1a, depth1[t], v1a, var1[t], &d, &v);
                depth1[t] = d; var1[t] = v;
            } else {
                touched[t] = 1; depth1[t] = d1a; var1[t] = v1a;
            }
        }
    }
    free(touched);
}

/* General 4x4 inverse by Gauss-Jordan with partial pivoting (the reference
 * calls LAPACK via ndarray-linalg, src/semi_dense/semi_dense.rs:83-89). */
static int inv4(const double *A, double *Ainv) {
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

/* src/semi_dense/semi_dense.rs:83-89 */
void orc_transform_rk(const double *T_wk, const double *T_wr, double *T_rk) {
    double T_rw[16];
    inv4(T_wr, T_rw);
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0.0;
            for (int k = 0; k < 4; k++) s += T_rw[4 * i + k] * T_wk[4 * k + j];
            T_rk[4 * i + j] = s;
        }
}

static inline double norm2(double a, double b) { return sqrt(a * a + b * b); }

/* src/vector.rs:4-11 */
static inline void vnormalize2(double *a, double *b) {
    double n = norm2(*a, *b);
    if (n == 0.) return;
    *a = *a / n; *b = *b / n;
}

/* src/semi_dense/hypothesis.rs:15-37 */
static inline int check_args(double inv_depth, double variance, double vmin,
                             double vmax) {
    if (inv_depth <= 0.) return -7; /* NegativePriorDepth */
    double mn = inv_depth - 2.0 * variance, mx = inv_depth + 2.0 * variance;
    if (mx <= vmin || vmax <= mn) return -1; /* HypothesisOutOfSerchRange */
    return 0;
}

/* src/cmp.rs:3-12 */
static inline double clampd(double v, double mn, double mx) {
    if (v < mn) return mn;
    if (v > mx) return mx;
    return v;
}

/* Hypothesis::range (src/semi_dense/hypothesis.rs:54-61) */
static inline void hypothesis_range(double inv_depth, double variance,
                                    double vmin, double vmax, double *rmin,
                                    double *rmax) {
    *rmin = clampd(inv_depth - 2.0 * variance, vmin, vmax);
    *rmax = clampd(inv_depth + 2.0 * variance, vmin, vmax);
}

/* calc_ref_depth (src/semi_dense/depth.rs:6-15) */
static inline double calc_ref_depth(const double *T_rk, double xk, double yk,
                                    double depth_key) {
    double pk[3];
    inv_project1(xk, yk, depth_key, pk);
    return ((T_rk[8] * pk[0] + T_rk[9] * pk[1]) + T_rk[10] * pk[2]) + T_rk[11];
}

/* step_ratio (src/semi_dense/semi_dense.rs:27-40) */
static inline int step_ratio(const double *T_rk, double xk, double yk,
                             double key_inv_depth, double *ratio) {
    double key_depth = safe_inv(key_inv_depth);
    double ref_depth = calc_ref_depth(T_rk, xk, yk, key_depth);
    if (ref_depth <= 0.) return -8; /* NegativeRefDepth */
    *ratio = key_inv_depth / safe_inv(ref_depth);
    return 0;
}

/* calc_key_epipole (src/semi_dense/epipolar.rs:9-20): project(R_wk^T (t_wr - t_wk)) */
static inline void key_epipole(const double *T_wk, const double *T_wr,
                               double *ex, double *ey) {
    double dt[3] = {T_wr[3] - T_wk[3], T_wr[7] - T_wk[7], T_wr[11] - T_wk[11]};
    double pe[3];
    for (int i = 0; i < 3; i++)
        pe[i] = (T_wk[i] * dt[0] + T_wk[4 + i] * dt[1]) + T_wk[8 + i] * dt[2];
    project1(pe, ex, ey);
}

/* key_coordinates (src/semi_dense/epipolar.rs:22-36) */
static inline void key_coordinates(double dirx, double diry, double xk,
                                   double yk, double step, double *out /*[10]*/) {
    static const double steps[5] = {-2., -1., 0., 1., 2.};
    vnormalize2(&dirx, &diry);
    for (int i = 0; i < 5; i++) {
        double s = step * steps[i];
        out[2 * i] = xk + s * dirx;
        out[2 * i + 1] = yk + s * diry;
    }
}

#define ORC_MAX_REF_SAMPLES (1 << 22)

/* ref_coordinates (src/semi_dense/epipolar.rs:38-54): n and the unit step */
static inline int64_t ref_line(double rdx, double rdy, double step,
                               double *dirx, double *diry) {
    double rnorm = norm2(rdx, rdy);
    *dirx = rdx / (rnorm + EPS16);
    *diry = rdy / (rnorm + EPS16);
    double nf = rnorm / step; /* `as usize`: truncation, NaN/negative -> 0 */
    if (!(nf >= 0.)) return 0;
    return nf < (double)ORC_MAX_REF_SAMPLES ? (int64_t)nf : ORC_MAX_REF_SAMPLES;
}

static inline void ref_coordinate(double xmin_x, double xmin_y, double dirx,
                                  double diry, double step, int64_t i,
                                  double *x, double *y) {
    double s = (double)i * step;
    *x = xmin_x + s * dirx;
    *y = xmin_y + s * diry;
}

/* intensities::search (src/semi_dense/intensities.rs:6-36): argmin over
 * windows of the squared distance between L2-normalised k-vectors, first
 * minimum wins; the returned index is argmin + k/2. */
static int64_t search_intensities(const double *seq, int64_t n,
                                  const double *kernel, int k) {
    double kn[16];
    double s = 0.0;
    for (int i = 0; i < k; i++) s += kernel[i] * kernel[i];
    double nn = sqrt(s);
    for (int i = 0; i < k; i++) kn[i] = (nn == 0.) ? kernel[i] : kernel[i] / nn;
    double min_err = INFINITY;
    int64_t argmin = 0;
    for (int64_t i = 0; i + k <= n; i++) {
        double q = 0.0;
        for (int j = 0; j < k; j++) q += seq[i + j] * seq[i + j];
        double sn = sqrt(q);
        double e = 0.0;
        for (int j = 0; j < k; j++) {
            double a = (sn == 0.) ? seq[i + j] : seq[i + j] / sn;
            double d = a - kn[j];
            e += d * d;
        }
        if (e < min_err) { min_err = e; argmin = i; }
    }
    return argmin + k / 2;
}

/* alpha_ (src/semi_dense/variance.rs:54-70) */
static inline double alpha_(double xk, double yk, double x_ref_i,
                            double direction_i, const double *ri,
                            const double *rz, double ti, double tz) {
    double rzy = (rz[0] * xk + rz[1] * yk) + rz[2] * 1.0;
    double riy = (ri[0] * xk + ri[1] * yk) + ri[2] * 1.0;
    double d = rzy * ti - riy * tz;
    double n = x_ref_i * tz - ti;
    return direction_i * d / (n * n);
}

/* calc_alpha_ (src/semi_dense/variance.rs:83-95) */
static inline double calc_alpha_(const double *T_rk, double xk, double yk,
                                 double dx, double dy, double prior_depth) {
    double xrx, xry, d1;
    warp1(T_rk, xk, yk, prior_depth, &xrx, &xry, &d1);
    int i = fabs(dx) > fabs(dy) ? 0 : 1;
    return alpha_(xk, yk, i == 0 ? xrx : xry, i == 0 ? dx : dy, &T_rk[4 * i],
                  &T_rk[8], T_rk[4 * i + 3], T_rk[11]);
}

/* geo_var_ (src/semi_dense/variance.rs:30-43) */
static inline double geo_var_(double dx, double dy, double gx, double gy) {
    vnormalize2(&dx, &dy);
    vnormalize2(&gx, &gy);
    double p = dx * gx + dy * gy;
    if (p == 0.) return 1. / EPS16;
    return 1. / (p * p);
}

/* calc_variance (src/semi_dense/variance.rs:15-24) */
static inline double calc_variance(double alpha, double geo_var,
                                   double photo_var, double geo_coeff,
                                   double photo_coeff) {
    double a2 = alpha * alpha;
    double g2 = geo_coeff * geo_coeff;
    double p2 = photo_coeff * photo_coeff;
    return a2 * (g2 * geo_var + p2 * photo_var);
}

/* src/semi_dense/semi_dense.rs:91-158.  prior_inv_depth/prior_variance form
 * the prior Hypothesis; gx/gy are the Sobel maps of the key image.  Returns
 * 0 and writes (inv_depth, variance) on success, else the negative Flag. */
static int estimate(double ukx, double uky, double prior_inv_depth,
                    double prior_variance, const double *key_cam,
                    const double *key_image, const double *T_wk,
                    const double *ref_cam, const double *ref_image,
                    const double *T_wr, const double *T_rk, int H, int W,
                    const double *gx, const double *gy, const orc_params *pr,
                    double *out_inv_depth, double *out_variance) {
    const double vmin = pr->inv_depth_min, vmax = pr->inv_depth_max;

    /* prior.range() -> depth_search_range (depth.rs:25-30) */
    double rmin, rmax;
    hypothesis_range(prior_inv_depth, prior_variance, vmin, vmax, &rmin, &rmax);
    double min_depth = safe_inv(rmax), max_depth = safe_inv(rmin);

    double xk, yk;
    normalize1(key_cam, ukx, uky, &xk, &yk);

    double ratio;
    int f = step_ratio(T_rk, xk, yk, prior_inv_depth, &ratio);
    if (f) return f;
    double key_step = ratio * pr->ref_step_size;

    /* calc_ref_ends (:51-60) */
    double xmin_x, xmin_y, xmax_x, xmax_y, dtmp;
    warp1(T_rk, xk, yk, min_depth, &xmin_x, &xmin_y, &dtmp);
    warp1(T_rk, xk, yk, max_depth, &xmax_x, &xmax_y, &dtmp);
    double rdx = xmax_x - xmin_x, rdy = xmax_y - xmin_y;

    double ex, ey;
    key_epipole(T_wk, T_wr, &ex, &ey);

    /* calc_key_direction (:42-49) */
    double kdx = xk - ex, kdy = yk - ey;
    if (!(rdx * kdx + rdy * kdy > 0.)) { kdx = -kdx; kdy = -kdy; }

    /* key samples -> unnormalize -> all_in_range (:119-124) */
    double xs_key[10], ukxs[5], ukys[5];
    key_coordinates(kdx, kdy, xk, yk, key_step, xs_key);
    for (int i = 0; i < 5; i++)
        unnormalize1(key_cam, xs_key[2 * i], xs_key[2 * i + 1], &ukxs[i], &ukys[i]);
    for (int i = 0; i < 5; i++)
        if (!in_range1(ukxs[i], ukys[i], H, W)) return -2; /* KeyOutOfRange */

    /* key intensities and the gradient gate (:126-134) */
    double key_I[5];
    for (int i = 0; i < 5; i++) key_I[i] = interpolate1(key_image, W, ukxs[i], ukys[i]);
    double g2 = 0.0;
    for (int i = 0; i < 4; i++) { double d = key_I[i + 1] - key_I[i]; g2 += d * d; }
    double key_gradient = sqrt(g2);
    if (key_gradient < pr->min_gradient) return -6; /* InsufficientGradient */

    /* ref samples (:137-139) and check_us_ref (:62-81) */
    double dirx, diry, ux, uy, x, y;
    int64_t n = ref_line(rdx, rdy, pr->ref_step_size, &dirx, &diry);
    if (n < 5) return -5; /* RefEpipolarTooShort */
    ref_coordinate(xmin_x, xmin_y, dirx, diry, pr->ref_step_size, 0, &x, &y);
    unnormalize1(ref_cam, x, y, &ux, &uy);
    if (!in_range1(ux, uy, H, W)) return -3; /* RefCloseOutOfRange */
    ref_coordinate(xmin_x, xmin_y, dirx, diry, pr->ref_step_size, n - 1, &x, &y);
    unnormalize1(ref_cam, x, y, &ux, &uy);
    if (!in_range1(ux, uy, H, W)) return -4; /* RefFarOutOfRange */

    /* ref intensities and the search (:142-145) */
    double *ref_I = (double *)malloc((size_t)n * sizeof(double));
    for (int64_t i = 0; i < n; i++) {
        ref_coordinate(xmin_x, xmin_y, dirx, diry, pr->ref_step_size, i, &x, &y);
        unnormalize1(ref_cam, x, y, &ux, &uy);
        ref_I[i] = interpolate1(ref_image, W, ux, uy);
    }
    int64_t argmin = search_intensities(ref_I, n, key_I, 5);
    free(ref_I);

    /* calc_key_depth (depth.rs:17-23) */
    double xr[2];
    ref_coordinate(xmin_x, xmin_y, dirx, diry, pr->ref_step_size, argmin, &xr[0], &xr[1]);
    double x_key[2] = {xk, yk};
    double key_depth = orc_calc_depth0(T_rk, x_key, xr);

    /* calc_alpha (variance.rs:97-105): direction re-derived from the ends */
    double adx = rdx, ady = rdy;
    vnormalize2(&adx, &ady);
    double alpha = calc_alpha_(T_rk, xk, yk, adx, ady, key_depth);

    /* geo_var (variance.rs:45-52) with ImageGradient::get (gradient.rs:17-25) */
    double t_rk[3] = {T_rk[3], T_rk[7], T_rk[11]};
    double px, py;
    project1(t_rk, &px, &py);
    double geo = geo_var_(xk - px, yk - py, interpolate1(gx, W, ukx, uky),
                          interpolate1(gy, W, ukx, uky));
    /* photo_var (variance.rs:26-28) of key_gradient / key_step_size (:153) */
    double photo = 2. / (key_gradient / key_step);
    double variance = calc_variance(alpha, geo, photo, pr->geo_coeff, pr->photo_coeff);

    double id = safe_inv(key_depth);
    f = check_args(id, variance, vmin, vmax);
    if (f) return f;
    *out_inv_depth = id;
    *out_variance = variance;
    return 0;
}

/* ---- test hooks: the helpers above, callable one by one so that the Rust
 * #[test] literals of the reference can be replayed against them ---------- */
double orc_t_safe_inv(double v) { return safe_inv(v); }
int orc_t_check_args(double id, double var, double vmin, double vmax) { return check_args(id, var, vmin, vmax); }
void orc_t_hypothesis_range(double id, double var, double vmin, double vmax, double *out) { hypothesis_range(id, var, vmin, vmax, &out[0], &out[1]); }
double orc_t_calc_ref_depth(const double *T_rk, const double *x_key, double d) { return calc_ref_depth(T_rk, x_key[0], x_key[1], d); }
int orc_t_step_ratio(const double *T_rk, const double *x_key, double key_inv_depth, double *ratio) { return step_ratio(T_rk, x_key[0], x_key[1], key_inv_depth, ratio); }
void orc_t_ref_ends(const double *T_rk, const double *x_key, double dmin, double dmax, double *out) {
    double d;
    warp1(T_rk, x_key[0], x_key[1], dmin, &out[0], &out[1], &d);
    warp1(T_rk, x_key[0], x_key[1], dmax, &out[2], &out[3], &d);
}
void orc_t_key_epipole(const double *T_wk, const double *T_wr, double *out) { key_epipole(T_wk, T_wr, &out[0], &out[1]); }
void orc_t_key_coordinates(const double *dir, const double *x_key, double step, double *out) { key_coordinates(dir[0], dir[1], x_key[0], x_key[1], step, out); }
int64_t orc_t_ref_coordinates(const double *x_min, const double *dir, double step, double *out, int64_t cap) {
    double dx, dy;
    int64_t n = ref_line(dir[0], dir[1], step, &dx, &dy);
    for (int64_t i = 0; i < n && i < cap; i++)
        ref_coordinate(x_min[0], x_min[1], dx, dy, step, i, &out[2 * i], &out[2 * i + 1]);
    return n;
}
/* check_us_ref (src/semi_dense/semi_dense.rs:62-81) */
int orc_t_check_us_ref(const double *us_ref, int64_t n, int64_t us_key_size, int H, int W) {
    if (n < us_key_size) return -5;
    if (!in_range1(us_ref[0], us_ref[1], H, W)) return -3;
    if (!in_range1(us_ref[2 * (n - 1)], us_ref[2 * (n - 1) + 1], H, W)) return -4;
    return 0;
}
int64_t orc_t_search(const double *seq, int64_t n, const double *kernel, int k) { return search_intensities(seq, n, kernel, k); }
double orc_t_alpha(const double *x_key, double x_ref_i, double direction_i, const double *ri, const double *rz, double ti, double tz) { return alpha_(x_key[0], x_key[1], x_ref_i, direction_i, ri, rz, ti, tz); }
double orc_t_calc_alpha(const double *T_rk, const double *x_key, const double *dir, double prior_depth) { return calc_alpha_(T_rk, x_key[0], x_key[1], dir[0], dir[1], prior_depth); }
double orc_t_geo_var(const double *dir, const double *grad) { return geo_var_(dir[0], dir[1], grad[0], grad[1]); }
double orc_t_calc_variance(double alpha, double geo, double photo, double geo_coeff, double photo_coeff) { return calc_variance(alpha, geo, photo, geo_coeff, photo_coeff); }
double orc_t_propagate_variance(double d0, double d1, double v0, double u) { return propagate_variance(d0, d1, v0, u); }
void orc_t_handle_collision(double da, double db, double va, double vb, double *out) { handle_collision(da, db, va, vb, &out[0], &out[1]); }

/* src/py/semi_dense.rs:126-155 */
int64_t orc_estimate_debug(const int64_t *u_key, double prior_depth,
                           double prior_variance, const double *key_cam,
                           const double *key_image, const double *key_T,
                           const double *ref_cam, const double *ref_image,
                           const double *ref_T, int H, int W,
                           const orc_params *params, double *out_depth,
                           double *out_variance) {
    *out_depth = prior_depth;
    *out_variance = prior_variance;
    double pid = safe_inv(prior_depth);
    int f = check_args(pid, prior_variance, params->inv_depth_min, params->inv_depth_max);
    if (f) return f;
    double *gx = (double *)malloc((size_t)H * W * sizeof(double));
    double *gy = (double *)malloc((size_t)H * W * sizeof(double));
    orc_sobel(key_image, H, W, gx, gy);
    double T_rk[16], id, var;
    orc_transform_rk(key_T, ref_T, T_rk);
    f = estimate((double)u_key[0], (double)u_key[1], pid, prior_variance, key_cam,
                 key_image, key_T, ref_cam, ref_image, ref_T, T_rk, H, W, gx, gy,
                 params, &id, &var);
    free(gx);
    free(gy);
    if (f) return f;
    *out_depth = safe_inv(id);
    *out_variance = var;
    return 0;
}

/* src/semi_dense/semi_dense.rs:160-234 */
int orc_update_depth(const double *key_cam, const double *key_image,
                     const double *key_T, int n_ref, const double *ref_cams,
                     const double *ref_images, const double *ref_Ts,
                     const uint64_t *age, const double *prior_depth,
                     const double *prior_variance, int H, int W,
                     const orc_params *params, double *out_depth,
                     double *out_variance, int64_t *out_flag) {
    return orc_update_depth_trk(key_cam, key_image, key_T, n_ref, ref_cams, ref_images,
                                ref_Ts, NULL, age, prior_depth, prior_variance, H, W,
                                params, out_depth, out_variance, out_flag);
}

/* The same loop with T_rk = inv(T_wr) T_wk supplied by the caller (n_ref x 16, or NULL
 * for orc_transform_rk): the sensitivity tests feed the inverse LAPACK's dgetrf + dgetri
 * give -- what ndarray_linalg::Inverse calls at src/semi_dense/semi_dense.rs:83-89 --
 * to count how many flags / depths the last bits of the 4x4 inverse decide. */
int orc_update_depth_trk(const double *key_cam, const double *key_image,
                         const double *key_T, int n_ref, const double *ref_cams,
                         const double *ref_images, const double *ref_Ts,
                         const double *T_rks_in,
                         const uint64_t *age, const double *prior_depth,
                         const double *prior_variance, int H, int W,
                         const orc_params *params, double *out_depth,
                         double *out_variance, int64_t *out_flag) {
    int64_t N = (int64_t)H * W;
    for (int64_t i = 0; i < N; i++)
        if (age[i] > (uint64_t)n_ref) return -1;
    double *gx = (double *)malloc((size_t)N * sizeof(double));
    double *gy = (double *)malloc((size_t)N * sizeof(double));
    orc_sobel(key_image, H, W, gx, gy);
    double *T_rks = (double *)malloc((size_t)(n_ref > 0 ? n_ref : 1) * 16 * sizeof(double));
    for (int r = 0; r < n_ref; r++) {
        if (T_rks_in) memcpy(&T_rks[16 * r], &T_rks_in[16 * r], 16 * sizeof(double));
        else orc_transform_rk(key_T, &ref_Ts[16 * r], &T_rks[16 * r]);
    }

    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            int64_t idx = (int64_t)y * W + x;
            uint64_t a = age[idx];
            double d = prior_depth[idx], v = prior_variance[idx];
            out_depth[idx] = d;
            out_variance[idx] = v;
            if (a == 0) { out_flag[idx] = -9; continue; } /* NotProcessed */
            int r = n_ref - (int)a;
            double pid = safe_inv(d);
            int f = check_args(pid, v, params->inv_depth_min, params->inv_depth_max);
            if (f) { out_flag[idx] = f; continue; }
            double id, var;
            f = estimate((double)x, (double)y, pid, v, key_cam, key_image, key_T,
                         &ref_cams[4 * r], &ref_images[(int64_t)r * N],
                         &ref_Ts[16 * r], &T_rks[16 * r], H, W, gx, gy, params,
                         &id, &var);
            if (f) { id = pid; var = v; } /* (prior, flag) */
            out_depth[idx] = safe_inv(id);
            out_variance[idx] = var;
            out_flag[idx] = f;
        }
    }
    free(gx);
    free(gy);
    free(T_rks);
    return 0;
}

/* ========================================================================
 * Semi-dense post-steps and colour conversion
 * ======================================================================== */

/* regularize_patch (src/semi_dense/regularization.rs:5-27): inverse-variance
 * weighted mean of the inverse depths of the Success pixels of a 3x3 patch,
 * accumulated in raster order.  Returns 0 and leaves *out alone if no pixel
 * contributes (`None`). */
int orc_regularize_patch(const double *inv_depth, const double *inv_variance,
                         const int64_t *flag, double *out) {
    double numerator = 0.0, denominator = 0.0;
    for (int k = 0; k < 9; k++) {
        if (flag[k] == 0) { /* Flag::Success, src/semi_dense/flag.rs:4 */
            numerator = numerator + inv_depth[k] * inv_variance[k];
            denominator = denominator + inv_variance[k];
        }
    }
    if (denominator == 0.0) return 0;
    *out = numerator / denominator;
    return 1;
}

/* regularize (src/semi_dense/regularization.rs:29-64): the maps are padded by
 * one pixel with (0, 0, NotProcessed); out = inv(patch mean) or the input depth. */
void orc_regularize(const double *depth, const double *variance,
                    const int64_t *flag, int H, int W, double *out) {
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            double id[9], iv[9];
            int64_t f[9];
            for (int dy = 0; dy < 3; dy++) {
                for (int dx = 0; dx < 3; dx++) {
                    int yy = y + dy - 1, xx = x + dx - 1, k = 3 * dy + dx;
                    if (yy < 0 || yy >= H || xx < 0 || xx >= W) {
                        id[k] = 0.0; iv[k] = 0.0; f[k] = -9;
                    } else {
                        id[k] = safe_inv(depth[(int64_t)yy * W + xx]);
                        iv[k] = safe_inv(variance[(int64_t)yy * W + xx]);
                        f[k] = flag[(int64_t)yy * W + xx];
                    }
                }
            }
            double r;
            if (orc_regularize_patch(id, iv, f, &r)) out[(int64_t)y * W + x] = safe_inv(r);
            else out[(int64_t)y * W + x] = depth[(int64_t)y * W + x];
        }
    }
}

/* fusion_arrays (src/semi_dense/fusion.rs:3-42), elementwise */
void orc_fusion_arrays(const double *mu1, const double *mu2, const double *var1,
                       const double *var2, int64_t n, double *mu, double *var) {
    for (int64_t i = 0; i < n; i++) {
        double v = var1[i] + var2[i];
        mu[i] = (mu1[i] * var2[i] + mu2[i] * var1[i]) / v;
        var[i] = (var1[i] * var2[i]) / v;
    }
}

/* skimage.color.rgb2gray (0.16.2; examples/dvo_pose_change.py:22-31): luma
 * 0.2125 R + 0.7154 G + 0.0721 B of the first three channels of a float image.
 * Third-party, restated from its published definition: parity unpinned in the
 * last ulp (the library sums through a BLAS dot whose order is unspecified). */
void orc_rgb2gray(const double *rgb, int64_t n, int channels, double *out) {
    for (int64_t i = 0; i < n; i++) {
        const double *p = rgb + i * channels;
        out[i] = (0.2125 * p[0] + 0.7154 * p[1]) + 0.0721 * p[2];
    }
}

/* ========================================================================
 * Bundle adjustment per observation (tadataka/so3_codegen.py:48-87,
 * tadataka/transform_project.pyx:22-50, tadataka/local_ba.py:14-39)
 *
 * pose = [omega(3), t(3)].  The reference differentiates, symbolically,
 *   theta = || omega + 1e-16 ||,  K = [omega]x / theta,
 *   R = I + sin(theta) K + (1 - cos(theta)) K K,   q = R p + t,
 *   x = q_xy / (q_z + 1e-16)
 * so the Jacobians below are the analytic derivatives of exactly that
 * expression (including d theta / d omega_k = (omega_k + 1e-16) / theta).
 * ===================================================================== */

typedef struct {
    double theta, A, B, dA, dB; /* A = sin/theta, B = (1-cos)/theta^2 */
    double th[3];               /* d theta / d omega_k */
} rod_t;

static inline void rodrigues_coeffs(const double *w, rod_t *c) {
    double e0 = w[0] + EPS16, e1 = w[1] + EPS16, e2 = w[2] + EPS16;
    double theta = sqrt((e0 * e0 + e1 * e1) + e2 * e2);
    double s = sin(theta), co = cos(theta);
    c->theta = theta;
    c->A = s / theta;
    c->B = (1. - co) / (theta * theta);
    c->dA = (co * theta - s) / (theta * theta);
    c->dB = (s * theta - 2. * (1. - co)) / (theta * theta * theta);
    c->th[0] = e0 / theta; c->th[1] = e1 / theta; c->th[2] = e2 / theta;
}

static inline void cross3(const double *a, const double *b, double *o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

/* so3_codegen.py:48-56, transform_project.pyx:46-50 */
void orc_exp_so3(const double *w, double *R) {
    rod_t c;
    rodrigues_coeffs(w, &c);
    /* W = [w]x ; W^2 = w w^T - |w|^2 I (with the unperturbed w) */
    double W[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double W2[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0.0;
            for (int k = 0; k < 3; k++) s += W[3 * i + k] * W[3 * k + j];
            W2[3 * i + j] = s;
        }
    for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0 ? 1.0 : 0.0) + c.A * W[i] + c.B * W2[i];
}

static inline void ba_q(const double *pose, const double *p, const rod_t *c,
                        double *q, double *wxp, double *wwxp) {
    const double *w = pose;
    cross3(w, p, wxp);
    cross3(w, wxp, wwxp);
    for (int i = 0; i < 3; i++) q[i] = (p[i] + c->A * wxp[i] + c->B * wwxp[i]) + pose[3 + i];
}

/* so3_codegen.py:63-66 */
void orc_ba_transform_project(const double *pose, const double *point, double *out) {
    rod_t c;
    double q[3], a[3], b[3];
    rodrigues_coeffs(pose, &c);
    ba_q(pose, point, &c, q, a, b);
    double z = q[2] + EPS16;
    out[0] = q[0] / z;
    out[1] = q[1] / z;
}

/* so3_codegen.py:80-81 (x.jacobian(pose)), row-major 2x6 */
void orc_ba_pose_jacobian(const double *pose, const double *point, double *out) {
    rod_t c;
    double q[3], wxp[3], wwxp[3];
    rodrigues_coeffs(pose, &c);
    ba_q(pose, point, &c, q, wxp, wwxp);
    double z = q[2] + EPS16;
    double iz = 1. / z;
    double dxq[2][3] = {{iz, 0., -q[0] * iz * iz}, {0., iz, -q[1] * iz * iz}};
    const double *w = pose;
    for (int k = 0; k < 3; k++) {
        double ek[3] = {0, 0, 0};
        ek[k] = 1.0;
        double ekxp[3], ek_wxp[3], w_ekxp[3], dq[3];
        cross3(ek, point, ekxp);   /* G_k p */
        cross3(ek, wxp, ek_wxp);   /* G_k W p */
        cross3(w, ekxp, w_ekxp);   /* W G_k p */
        for (int i = 0; i < 3; i++)
            dq[i] = c.dA * c.th[k] * wxp[i] + c.A * ekxp[i] +
                    c.dB * c.th[k] * wwxp[i] + c.B * (ek_wxp[i] + w_ekxp[i]);
        for (int r = 0; r < 2; r++)
            out[6 * r + k] = dxq[r][0] * dq[0] + dxq[r][1] * dq[1] + dxq[r][2] * dq[2];
    }
    for (int r = 0; r < 2; r++)
        for (int k = 0; k < 3; k++) out[6 * r + 3 + k] = dxq[r][k];
}

/* so3_codegen.py:83-84 (x.jacobian(point)), row-major 2x3 */
void orc_ba_point_jacobian(const double *pose, const double *point, double *out) {
    rod_t c;
    double q[3], wxp[3], wwxp[3], R[9];
    rodrigues_coeffs(pose, &c);
    ba_q(pose, point, &c, q, wxp, wwxp);
    orc_exp_so3(pose, R);
    double z = q[2] + EPS16;
    double iz = 1. / z;
    double dxq[2][3] = {{iz, 0., -q[0] * iz * iz}, {0., iz, -q[1] * iz * iz}};
    for (int r = 0; r < 2; r++)
        for (int k = 0; k < 3; k++)
            out[3 * r + k] = dxq[r][0] * R[k] + dxq[r][1] * R[3 + k] + dxq[r][2] * R[6 + k];
}

/* tadataka/local_ba.py:23-39: gather by (viewpoint, point) index pairs */
void orc_ba_projection(const double *poses, const double *points,
                       const int64_t *vp_idx, const int64_t *pt_idx, int64_t n,
                       double *x_pred, double *A, double *B) {
    for (int64_t k = 0; k < n; k++) {
        const double *pose = &poses[6 * vp_idx[k]], *pt = &points[3 * pt_idx[k]];
        if (x_pred) orc_ba_transform_project(pose, pt, &x_pred[2 * k]);
        if (A) orc_ba_pose_jacobian(pose, pt, &A[12 * k]);
        if (B) orc_ba_point_jacobian(pose, pt, &B[6 * k]);
    }
}

/* Block sums the SBA solve (sparseba.SBA.compute, call site local_ba.py:77)
 * starts from: e = x_true - x_pred, U_j = sum_i A^T A, ea_j = sum_i A^T e,
 * V_i = sum_j B^T B, eb_i = sum_j B^T e. */
double orc_ba_block_reduce(const double *poses, int64_t n_poses,
                           const double *points, int64_t n_points,
                           const double *x_true, const int64_t *vp_idx,
                           const int64_t *pt_idx, int64_t n, double *U,
                           double *ea, double *V, double *eb) {
    memset(U, 0, (size_t)n_poses * 21 * sizeof(double));
    memset(ea, 0, (size_t)n_poses * 6 * sizeof(double));
    memset(V, 0, (size_t)n_points * 6 * sizeof(double));
    memset(eb, 0, (size_t)n_points * 3 * sizeof(double));
    double err = 0.0;
    for (int64_t k = 0; k < n; k++) {
        int64_t j = vp_idx[k], i = pt_idx[k];
        const double *pose = &poses[6 * j], *pt = &points[3 * i];
        double x[2], A[12], B[6];
        orc_ba_transform_project(pose, pt, x);
        orc_ba_pose_jacobian(pose, pt, A);
        orc_ba_point_jacobian(pose, pt, B);
        double e0 = x_true[2 * k] - x[0], e1 = x_true[2 * k + 1] - x[1];
        err += e0 * e0 + e1 * e1;
        int m = 0;
        for (int a = 0; a < 6; a++) {
            for (int b = a; b < 6; b++)
                U[21 * j + m++] += A[a] * A[b] + A[6 + a] * A[6 + b];
            ea[6 * j + a] += A[a] * e0 + A[6 + a] * e1;
        }
        m = 0;
        for (int a = 0; a < 3; a++) {
            for (int b = a; b < 3; b++)
                V[6 * i + m++] += B[a] * B[b] + B[3 + a] * B[3 + b];
            eb[3 * i + a] += B[a] * e0 + B[3 + a] * e1;
        }
    }
    return err;
}
