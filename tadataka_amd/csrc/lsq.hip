// lsq.hip -- least-squares building blocks at the reference's own granularity:
//   * tdk_weighted_normal_equations : A^T W A / A^T W b for an n x p system, the
//     reduction behind tadataka.math.solve_linear_equation (math.py:32-45) and
//     every IRLS step of tadataka.irls.fit (irls.py:186-218);
//   * tdk_dvo_pose_update           : calc_pose_update (vo/dvo/__init__.py:46-70)
//     on explicit arrays (residuals, GX1, GY1, P1) as the reference passes them;
//   * tdk_robust_weights            : compute_weights_{huber,student_t,tukey}
//     (robust/weights.py:4-43) of a residual vector, including the global
//     statistics (10 fixed-point variance iterations; medians by radix select).
#include "tdk_math.h"
#include "tdk_runtime.h"

#include <math.h>
#include <string.h>

namespace {

using tdk::Cam;

constexpr int kBlock = 256;
constexpr int kMaxP = 8;
constexpr int kAccPad = 48;
constexpr double kHuberK = 1.345;

inline int grid_for(int64_t n, int per_thread = 1) {
    int64_t g = (n + (int64_t)kBlock * per_thread - 1) / ((int64_t)kBlock * per_thread);
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;
    return (int)g;
}

// block-wide sum of NACC per-thread values -> partials[blockIdx.x][kAccPad]
template <int NACC>
__device__ __forceinline__ void block_reduce_store(double *acc, double *partials) {
    __shared__ double red[kBlock / 64][kAccPad];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NACC; k++) {
        double s = acc[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < NACC) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; w++) s += red[w][threadIdx.x];
        partials[(int64_t)blockIdx.x * kAccPad + threadIdx.x] = s;
    }
}

// fixed-order sum of the per-block partials (bit-reproducible)
__global__ __launch_bounds__(kBlock) void k_finish(const double *__restrict__ partials, int nblk, int nacc,
                                                   double *__restrict__ out) {
    __shared__ double red[kBlock / 64][64];
    const int k = threadIdx.x & 63, g = threadIdx.x >> 6;
    double s = 0.0;
    if (k < nacc)
        for (int b = g; b < nblk; b += kBlock / 64) s += partials[(int64_t)b * kAccPad + k];
    red[g][k] = s;
    __syncthreads();
    if (threadIdx.x < nacc) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < kBlock / 64; i++) t += red[i][threadIdx.x];
        out[threadIdx.x] = t;
    }
}

template <int P>
__global__ __launch_bounds__(kBlock) void k_normal_equations(const double *__restrict__ A,
                                                             const double *__restrict__ b,
                                                             const double *__restrict__ w, int64_t n,
                                                             double *__restrict__ partials) {
    constexpr int NT = P * (P + 1) / 2;
    double acc[NT + P];
#pragma unroll
    for (int i = 0; i < NT + P; i++) acc[i] = 0.0;
    for (int64_t r = blockIdx.x * (int64_t)kBlock + threadIdx.x; r < n; r += (int64_t)gridDim.x * kBlock) {
        double a[P];
#pragma unroll
        for (int j = 0; j < P; j++) a[j] = A[r * P + j];
        double wr = w ? w[r] : 1.0, br = b[r];
        int k = 0;
#pragma unroll
        for (int i = 0; i < P; i++) {
            double wa = wr * a[i];
#pragma unroll
            for (int j = i; j < P; j++) acc[k++] += wa * a[j];
            acc[NT + i] += wa * br;
        }
    }
    block_reduce_store<NT + P>(acc, partials);
}

// calc_pose_update on explicit arrays.  P1 [n,3], residuals [n], GX1/GY1 [H,W],
// weights [n] (TDK_W_MAP) -- all indexed by SOURCE pixel, as the reference's
// `weights.flatten()[mask]` / `residuals[mask]` are.
template <int WMODE>
__global__ __launch_bounds__(kBlock) void k_pose_update(Cam c1, const double *__restrict__ residuals,
                                                        const double *__restrict__ GX,
                                                        const double *__restrict__ GY, int H, int W,
                                                        const double *__restrict__ P1, int64_t n,
                                                        const double *__restrict__ weights,
                                                        double *__restrict__ partials) {
    double acc[28];
#pragma unroll
    for (int i = 0; i < 28; i++) acc[i] = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        double x = P1[3 * i], y = P1[3 * i + 1], z = P1[3 * i + 2];
        double u, v;
        {
#pragma clang fp contract(off)
            // written out here: an inlined helper would keep its own
            // (contracting) floating-point mode
            double zz = z + tdk::kEps16;
            u = (x / zz) * c1.fx + c1.ox;
            v = (y / zz) * c1.fy + c1.oy;
        }
        if (!(tdk::in_range(u, v, H, W) && z > 0)) continue;
        double gx = tdk::bilinear(GX, H, W, u, v), gy = tdk::bilinear(GY, H, W, u, v);
        double fgx = c1.fx * gx, fgy = c1.fy * gy;
        double z2 = z * z, xy = x * y;
        double J[6];
        J[0] = fgx / z;
        J[1] = fgy / z;
        J[2] = -(fgx * x + fgy * y) / (z * z);
        J[3] = -(fgx * xy + fgy * (z2 + y * y)) / z2;
        J[4] = (fgx * (z2 + x * x) + fgy * xy) / z2;
        J[5] = (-fgx * y + fgy * x) / z;
        double r = residuals[i], wr = 1.0;
        if (WMODE == TDK_W_HUBER) {
            double ar = fabs(r);
            wr = ar > kHuberK ? kHuberK / ar : 1.0;
        } else if (WMODE == TDK_W_MAP) {
            wr = weights[i];
        }
        int k = 0;
#pragma unroll
        for (int a = 0; a < 6; a++) {
            double wj = wr * J[a];
#pragma unroll
            for (int b = a; b < 6; b++) acc[k++] += wj * J[b];
            acc[21 + a] += wj * r;
        }
        acc[27] += 1.0;
    }
    block_reduce_store<28>(acc, partials);
}

// ---- robust weights ------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_huber(const double *__restrict__ r, int64_t m, double k,
                                                  double *__restrict__ w) {
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kBlock) {
        double a = fabs(r[i]);
        w[i] = a > k ? k / a : 1.0;
    }
}

// one fixed-point step of compute_weights_student_t: sum s (nu+1)/(nu + s/var)
__global__ __launch_bounds__(kBlock) void k_student_t_step(const double *__restrict__ r, int64_t m,
                                                           const double *__restrict__ variance, double nu,
                                                           double *__restrict__ partials) {
    const double var = *variance;
    double acc[1] = {0.0};
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kBlock) {
        double s = r[i] * r[i];
        acc[0] += s * ((nu + 1.0) / (nu + s / var));
    }
    block_reduce_store<1>(acc, partials);
}

__global__ void k_student_t_update(const double *__restrict__ sum, int64_t m, double *__restrict__ variance) {
    *variance = sum[0] / (double)m;
}

__global__ __launch_bounds__(kBlock) void k_student_t_weights(const double *__restrict__ r, int64_t m,
                                                              const double *__restrict__ variance, double nu,
                                                              double *__restrict__ w) {
    const double var = *variance;
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kBlock) {
        double s = r[i] * r[i];
        w[i] = sqrt((nu + 1.0) / (nu + s / var));   // note: sqrt of the weight (weights.py:18)
    }
}

// Order statistics by MSD radix select on the order-preserving 64-bit image of
// a double.  One pass = histogram of the next 8 bits among keys that match the
// prefix found so far, then a single thread picks the bin holding rank k.
__device__ __forceinline__ uint64_t ordered_key(double v) {
    uint64_t b = (uint64_t)__double_as_longlong(v);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}

__device__ __forceinline__ double key_to_double(uint64_t k) {
    uint64_t b = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

struct SelectState {
    uint64_t prefix;   // bits found so far (high bits)
    uint64_t rank;     // rank of the wanted element among keys matching the prefix
};

// value source: r[i] (mode 0) or |r[i] - *center| (mode 1, for the MAD)
__device__ __forceinline__ double select_value(const double *r, int64_t i, int mode, const double *center) {
    return mode == 0 ? r[i] : fabs(r[i] - *center);
}

__global__ __launch_bounds__(kBlock) void k_select_hist(const double *__restrict__ r, int64_t m, int mode,
                                                        const double *__restrict__ center,
                                                        const SelectState *__restrict__ st, int pass,
                                                        unsigned long long *__restrict__ hist) {
    __shared__ unsigned int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int shift = 56 - 8 * pass;
    const uint64_t prefix = st->prefix;
    const uint64_t mask = pass == 0 ? 0ull : (~0ull << (shift + 8));
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kBlock) {
        uint64_t k = ordered_key(select_value(r, i, mode, center));
        if ((k & mask) == prefix) atomicAdd(&h[(k >> shift) & 0xff], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

__global__ void k_select_pick(unsigned long long *__restrict__ hist, SelectState *__restrict__ st, int pass,
                              double *__restrict__ out) {
    const int shift = 56 - 8 * pass;
    uint64_t rank = st->rank, cum = 0;
    int bin = 255;
    for (int b = 0; b < 256; b++) {
        uint64_t c = hist[b];
        if (rank < cum + c) { bin = b; break; }
        cum += c;
    }
    st->rank = rank - cum;
    st->prefix |= ((uint64_t)bin) << shift;
    for (int b = 0; b < 256; b++) hist[b] = 0;
    if (pass == 7) *out = key_to_double(st->prefix);
}

__global__ void k_select_init(SelectState *st, uint64_t rank) {
    st->prefix = 0;
    st->rank = rank;
}

// median = mean of the two middle order statistics for even m (np.median)
__global__ void k_median_combine(const double *lo, const double *hi, double *out) { *out = (*lo + *hi) / 2.0; }

__global__ void k_scale(const double *in, double factor, double *out) { *out = factor * *in; }

__global__ __launch_bounds__(kBlock) void k_tukey_weights(const double *__restrict__ r, int64_t m,
                                                          const double *__restrict__ sigma, double beta,
                                                          double *__restrict__ w) {
    const double s = *sigma;
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kBlock) {
        double x = r[i] / s;
        double q = x / beta, u = 1.0 - q * q;
        w[i] = fabs(x) <= beta ? u * u : 0.0;
    }
}

tdk_status h2d(int slot, const void *host, size_t bytes, void **dev) {
    TDK_TRY(tdk::scratch(slot, bytes, dev));
    if (bytes) TDK_HIP(hipMemcpyAsync(*dev, host, bytes, hipMemcpyHostToDevice, tdk::stream()));
    return TDK_OK;
}

// k-th smallest (0-based) of select_value(r, ., mode, center) -> *out (device)
tdk_status device_select(const double *d_r, int64_t m, int mode, const double *d_center, uint64_t rank,
                         SelectState *d_st, unsigned long long *d_hist, double *d_out) {
    k_select_init<<<1, 1, 0, tdk::stream()>>>(d_st, rank);
    TDK_LAUNCH_CHECK();
    for (int pass = 0; pass < 8; pass++) {
        k_select_hist<<<grid_for(m, 4), kBlock, 0, tdk::stream()>>>(d_r, m, mode, d_center, d_st, pass, d_hist);
        TDK_LAUNCH_CHECK();
        k_select_pick<<<1, 1, 0, tdk::stream()>>>(d_hist, d_st, pass, d_out);
        TDK_LAUNCH_CHECK();
    }
    return TDK_OK;
}

// np.median of the selected values -> *d_out (device); d_tmp holds two doubles
tdk_status device_median(const double *d_r, int64_t m, int mode, const double *d_center, SelectState *d_st,
                         unsigned long long *d_hist, double *d_tmp, double *d_out) {
    if (m % 2 == 1) return device_select(d_r, m, mode, d_center, (uint64_t)(m / 2), d_st, d_hist, d_out);
    TDK_TRY(device_select(d_r, m, mode, d_center, (uint64_t)(m / 2 - 1), d_st, d_hist, d_tmp));
    TDK_TRY(device_select(d_r, m, mode, d_center, (uint64_t)(m / 2), d_st, d_hist, d_tmp + 1));
    k_median_combine<<<1, 1, 0, tdk::stream()>>>(d_tmp, d_tmp + 1, d_out);
    TDK_LAUNCH_CHECK();
    return TDK_OK;
}

}  // namespace

extern "C" {

tdk_status tdk_weighted_normal_equations(const double *A, const double *b, const double *w, int64_t n, int p,
                                         double *AtWA, double *AtWb) {
    TDK_API_GUARD;
    TDK_REQUIRE(p >= 1 && p <= kMaxP, "p must be in [1, 8]");
    TDK_REQUIRE(n >= 0 && AtWA && AtWb && (n == 0 || (A && b)), "bad argument");
    const int nt = p * (p + 1) / 2, nacc = nt + p;
    if (n == 0) {
        memset(AtWA, 0, sizeof(double) * nt);
        memset(AtWb, 0, sizeof(double) * p);
        return tdk::ensure_device();
    }
    void *d_A, *d_b, *d_w = nullptr, *d_part, *d_out;
    TDK_TRY(h2d(0, A, (size_t)n * p * 8, &d_A));
    TDK_TRY(h2d(1, b, (size_t)n * 8, &d_b));
    if (w) TDK_TRY(h2d(2, w, (size_t)n * 8, &d_w));
    const int nblk = grid_for(n, 4);
    TDK_TRY(tdk::scratch(3, (size_t)nblk * kAccPad * 8, &d_part));
    TDK_TRY(tdk::scratch(4, kAccPad * 8, &d_out));
#define TDK_NE(PP)                                                                                        \
    case PP:                                                                                              \
        k_normal_equations<PP><<<nblk, kBlock, 0, tdk::stream()>>>((const double *)d_A, (const double *)d_b, \
                                                                   (const double *)d_w, n, (double *)d_part); \
        break;
    switch (p) {
        TDK_NE(1) TDK_NE(2) TDK_NE(3) TDK_NE(4) TDK_NE(5) TDK_NE(6) TDK_NE(7) TDK_NE(8)
    }
#undef TDK_NE
    TDK_LAUNCH_CHECK();
    k_finish<<<1, kBlock, 0, tdk::stream()>>>((const double *)d_part, nblk, nacc, (double *)d_out);
    TDK_LAUNCH_CHECK();
    double out[kAccPad];
    TDK_HIP(hipMemcpyAsync(out, d_out, sizeof(double) * nacc, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipStreamSynchronize(tdk::stream()));
    memcpy(AtWA, out, sizeof(double) * nt);
    memcpy(AtWb, out + nt, sizeof(double) * p);
    return TDK_OK;
}

tdk_status tdk_dvo_pose_update(const double *camera1, const double *residuals, const double *GX1,
                               const double *GY1, int H, int W, const double *P1, int64_t n, int weight_mode,
                               const double *weights, double *H21, double *b6, int64_t *n_valid) {
    TDK_API_GUARD;
    TDK_REQUIRE(camera1 && residuals && GX1 && GY1 && P1 && H21 && b6 && n_valid, "null pointer");
    TDK_REQUIRE(H >= 1 && W >= 1 && n >= 0, "bad size");
    TDK_REQUIRE(weight_mode == TDK_W_NONE || weight_mode == TDK_W_HUBER || weight_mode == TDK_W_MAP,
                "weight mode must be none, huber or map (student-t / tukey: tdk_robust_weights first)");
    TDK_REQUIRE(weight_mode != TDK_W_MAP || weights, "weights is NULL");
    if (n == 0) {
        memset(H21, 0, sizeof(double) * 21);
        memset(b6, 0, sizeof(double) * 6);
        *n_valid = 0;
        return tdk::ensure_device();
    }
    void *d_r, *d_gx, *d_gy, *d_p, *d_w = nullptr, *d_part, *d_out;
    TDK_TRY(h2d(0, residuals, (size_t)n * 8, &d_r));
    TDK_TRY(h2d(1, GX1, (size_t)H * W * 8, &d_gx));
    TDK_TRY(h2d(2, GY1, (size_t)H * W * 8, &d_gy));
    TDK_TRY(h2d(3, P1, (size_t)n * 24, &d_p));
    if (weight_mode == TDK_W_MAP) TDK_TRY(h2d(4, weights, (size_t)n * 8, &d_w));
    const int nblk = grid_for(n, 4);
    TDK_TRY(tdk::scratch(5, (size_t)nblk * kAccPad * 8, &d_part));
    TDK_TRY(tdk::scratch(6, kAccPad * 8, &d_out));
    Cam c1{camera1[0], camera1[1], camera1[2], camera1[3]};
#define TDK_PU(WM)                                                                                           \
    k_pose_update<WM><<<nblk, kBlock, 0, tdk::stream()>>>(c1, (const double *)d_r, (const double *)d_gx,     \
                                                          (const double *)d_gy, H, W, (const double *)d_p, n, \
                                                          (const double *)d_w, (double *)d_part)
    if (weight_mode == TDK_W_NONE) TDK_PU(TDK_W_NONE);
    else if (weight_mode == TDK_W_HUBER) TDK_PU(TDK_W_HUBER);
    else TDK_PU(TDK_W_MAP);
#undef TDK_PU
    TDK_LAUNCH_CHECK();
    k_finish<<<1, kBlock, 0, tdk::stream()>>>((const double *)d_part, nblk, 28, (double *)d_out);
    TDK_LAUNCH_CHECK();
    double out[28];
    TDK_HIP(hipMemcpyAsync(out, d_out, sizeof(out), hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipStreamSynchronize(tdk::stream()));
    memcpy(H21, out, sizeof(double) * 21);
    memcpy(b6, out + 21, sizeof(double) * 6);
    *n_valid = (int64_t)out[27];
    return TDK_OK;
}

tdk_status tdk_robust_weights(const double *residuals, int64_t m, int mode, double *weights) {
    TDK_API_GUARD;
    // the reference defaults: k = 1.345 | nu = 5, n_iter = 10 | beta = 4.6851, c = 1.4826
    const double p0 = mode == TDK_W_HUBER ? kHuberK : (mode == TDK_W_STUDENT_T ? 5.0 : 4.6851);
    const double p1 = mode == TDK_W_STUDENT_T ? 10.0 : 1.4826;
    return tdk_robust_weights_ex(residuals, m, mode, p0, p1, weights);
}

tdk_status tdk_robust_weights_ex(const double *residuals, int64_t m, int mode, double p0, double p1,
                                 double *weights) {
    TDK_API_GUARD;
    TDK_REQUIRE(m >= 0 && (m == 0 || (residuals && weights)), "bad argument");
    TDK_REQUIRE(mode == TDK_W_HUBER || mode == TDK_W_STUDENT_T || mode == TDK_W_TUKEY,
                "mode must be huber, student-t or tukey");
    TDK_REQUIRE(mode != TDK_W_STUDENT_T || (p1 >= 0.0 && p1 <= 1e6 && p1 == floor(p1)), "n_iter must be a non-negative integer");
    if (m == 0) return tdk::ensure_device();
    void *d_r, *d_w, *d_part, *d_s;
    TDK_TRY(h2d(0, residuals, (size_t)m * 8, &d_r));
    TDK_TRY(tdk::scratch(1, (size_t)m * 8, &d_w));
    const int nblk = grid_for(m, 4);
    TDK_TRY(tdk::scratch(2, (size_t)nblk * kAccPad * 8, &d_part));
    // small state block: [0] variance/sigma, [1] sum, [2..3] tmp, [4] median, [5] mad, then SelectState, hist
    TDK_TRY(tdk::scratch(3, 64 * 8 + 256 * 8, &d_s));
    double *s = (double *)d_s;
    SelectState *st = (SelectState *)(s + 8);
    unsigned long long *hist = (unsigned long long *)(s + 64);
    const double *r = (const double *)d_r;
    double *w = (double *)d_w;
    if (mode == TDK_W_HUBER) {
        k_huber<<<grid_for(m, 4), kBlock, 0, tdk::stream()>>>(r, m, p0, w);
        TDK_LAUNCH_CHECK();
    } else if (mode == TDK_W_STUDENT_T) {
        double one = 1.0;
        TDK_HIP(hipMemcpyAsync(s, &one, 8, hipMemcpyHostToDevice, tdk::stream()));
        for (int it = 0; it < (int)p1; it++) {
            k_student_t_step<<<nblk, kBlock, 0, tdk::stream()>>>(r, m, s, p0, (double *)d_part);
            TDK_LAUNCH_CHECK();
            k_finish<<<1, kBlock, 0, tdk::stream()>>>((const double *)d_part, nblk, 1, s + 1);
            TDK_LAUNCH_CHECK();
            k_student_t_update<<<1, 1, 0, tdk::stream()>>>(s + 1, m, s);
            TDK_LAUNCH_CHECK();
        }
        k_student_t_weights<<<grid_for(m, 4), kBlock, 0, tdk::stream()>>>(r, m, s, p0, w);
        TDK_LAUNCH_CHECK();
    } else {
        TDK_HIP(hipMemsetAsync(hist, 0, 256 * 8, tdk::stream()));
        TDK_TRY(device_median(r, m, 0, nullptr, st, hist, s + 2, s + 4));   // median(r)
        TDK_TRY(device_median(r, m, 1, s + 4, st, hist, s + 2, s + 5));     // median(|r - median|)
        k_scale<<<1, 1, 0, tdk::stream()>>>(s + 5, p1, s);                    // sigma_mad = c * MAD
        TDK_LAUNCH_CHECK();
        k_tukey_weights<<<grid_for(m, 4), kBlock, 0, tdk::stream()>>>(r, m, s, p0, w);
        TDK_LAUNCH_CHECK();
    }
    TDK_HIP(hipMemcpyAsync(weights, d_w, (size_t)m * 8, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipStreamSynchronize(tdk::stream()));
    return TDK_OK;
}

}  // extern "C"
