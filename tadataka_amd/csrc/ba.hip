// ba.hip -- bundle-adjustment per-observation arithmetic on the MI355X:
// transform_project and its pose / point Jacobians (the reference's
// sympy-generated C behind tadataka.transform_project, so3_codegen.py:48-87,
// called once per observation from a Python loop in tadataka/local_ba.py:23-39),
// and the fused residual + Jacobian + block reduction that produces the sums
// sparseba.SBA.compute starts from (call site local_ba.py:74-77).
//
// pose = [omega(3), t(3)].  The reference differentiates symbolically
//   theta = ||omega + 1e-16||,  K = [omega]x / theta,
//   R = I + sin(theta) K + (1 - cos(theta)) K K,  q = R p + t,  x = q_xy / (q_z + 1e-16)
// and the Jacobians here are the analytic derivatives of exactly that expression
// (d theta / d omega_k = (omega_k + 1e-16) / theta included).
#include "tdk_math.h"
#include "tdk_runtime.h"

#include <math.h>

namespace {

constexpr int kBlock = 256;
constexpr int kPoseAcc = 28;  // 21 U + 6 ea + err
constexpr int kPoseAccPad = 32;

inline int grid_for(int64_t n) {
    int64_t g = (n + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if (g > 1 << 20) g = 1 << 20;
    return (int)g;
}

struct Rod {
    double A, B, dA, dB, th[3];
};

__device__ __forceinline__ void rodrigues_coeffs(const double *w, Rod &c) {
    double e0 = w[0] + tdk::kEps16, e1 = w[1] + tdk::kEps16, e2 = w[2] + tdk::kEps16;
    double theta = sqrt((e0 * e0 + e1 * e1) + e2 * e2);
    double s, co;
    sincos(theta, &s, &co);
    double it = 1.0 / theta;
    c.A = s * it;
    c.B = (1. - co) * it * it;
    c.dA = (co * theta - s) * it * it;
    c.dB = (s * theta - 2. * (1. - co)) * it * it * it;
    c.th[0] = e0 * it; c.th[1] = e1 * it; c.th[2] = e2 * it;
}

__device__ __forceinline__ void cross3(const double *a, const double *b, double *o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

// x (2), optionally A = dx/dpose (2x6 row-major) and B = dx/dpoint (2x3 row-major)
template <bool JAC>
__device__ __forceinline__ void project_observation(const double *pose, const double *p, double *x, double *A,
                                                    double *B) {
    Rod c;
    rodrigues_coeffs(pose, c);
    const double *w = pose;
    double wxp[3], wwxp[3], q[3];
    cross3(w, p, wxp);
    cross3(w, wxp, wwxp);
#pragma unroll
    for (int i = 0; i < 3; i++) q[i] = (p[i] + c.A * wxp[i] + c.B * wwxp[i]) + pose[3 + i];
    double iz = 1.0 / (q[2] + tdk::kEps16);
    x[0] = q[0] * iz;
    x[1] = q[1] * iz;
    if (!JAC) return;
    double dxq[2][3] = {{iz, 0., -q[0] * iz * iz}, {0., iz, -q[1] * iz * iz}};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double ek[3] = {0., 0., 0.};
        ek[k] = 1.0;
        double ekxp[3], ek_wxp[3], w_ekxp[3], dq[3];
        cross3(ek, p, ekxp);     // G_k p
        cross3(ek, wxp, ek_wxp); // G_k W p
        cross3(w, ekxp, w_ekxp); // W G_k p
#pragma unroll
        for (int i = 0; i < 3; i++)
            dq[i] = c.dA * c.th[k] * wxp[i] + c.A * ekxp[i] + c.dB * c.th[k] * wwxp[i] +
                    c.B * (ek_wxp[i] + w_ekxp[i]);
#pragma unroll
        for (int r = 0; r < 2; r++) A[6 * r + k] = dxq[r][0] * dq[0] + dxq[r][1] * dq[1] + dxq[r][2] * dq[2];
    }
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int k = 0; k < 3; k++) A[6 * r + 3 + k] = dxq[r][k];
    // R columns: R e_k = e_k + A (w x e_k) + B (w x (w x e_k))
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double ek[3] = {0., 0., 0.};
        ek[k] = 1.0;
        double a[3], b[3], col[3];
        cross3(w, ek, a);
        cross3(w, a, b);
#pragma unroll
        for (int i = 0; i < 3; i++) col[i] = ek[i] + c.A * a[i] + c.B * b[i];
#pragma unroll
        for (int r = 0; r < 2; r++) B[3 * r + k] = dxq[r][0] * col[0] + dxq[r][1] * col[1] + dxq[r][2] * col[2];
    }
}

__global__ __launch_bounds__(kBlock) void k_ba_projection(const double *__restrict__ poses,
                                                          const double *__restrict__ points,
                                                          const int64_t *__restrict__ vp,
                                                          const int64_t *__restrict__ pt, int64_t n,
                                                          double *__restrict__ x_pred, double *__restrict__ Aout,
                                                          double *__restrict__ Bout) {
    for (int64_t k = blockIdx.x * (int64_t)kBlock + threadIdx.x; k < n; k += (int64_t)gridDim.x * kBlock) {
        double pose[6], p[3], x[2], A[12], B[6];
        const double *ps = poses + 6 * vp[k];
        const double *pp = points + 3 * pt[k];
#pragma unroll
        for (int i = 0; i < 6; i++) pose[i] = ps[i];
#pragma unroll
        for (int i = 0; i < 3; i++) p[i] = pp[i];
        if (Aout != nullptr || Bout != nullptr) project_observation<true>(pose, p, x, A, B);
        else project_observation<false>(pose, p, x, A, B);
        if (x_pred) { x_pred[2 * k] = x[0]; x_pred[2 * k + 1] = x[1]; }
        if (Aout)
#pragma unroll
            for (int i = 0; i < 12; i++) Aout[12 * k + i] = A[i];
        if (Bout)
#pragma unroll
            for (int i = 0; i < 6; i++) Bout[6 * k + i] = B[i];
    }
}

__global__ __launch_bounds__(kBlock) void k_ba_exp_so3(const double *__restrict__ rotvecs, int64_t n,
                                                       double *__restrict__ R) {
    for (int64_t k = blockIdx.x * (int64_t)kBlock + threadIdx.x; k < n; k += (int64_t)gridDim.x * kBlock) {
        double w[3] = {rotvecs[3 * k], rotvecs[3 * k + 1], rotvecs[3 * k + 2]};
        Rod c;
        rodrigues_coeffs(w, c);
#pragma unroll
        for (int j = 0; j < 3; j++) {
            double ej[3] = {0., 0., 0.};
            ej[j] = 1.0;
            double a[3], b[3];
            cross3(w, ej, a);
            cross3(w, a, b);
#pragma unroll
            for (int i = 0; i < 3; i++) R[9 * k + 3 * i + j] = ej[i] + c.A * a[i] + c.B * b[i];
        }
    }
}

// Fused residual + Jacobians + block sums.  grid = (chunks, n_poses): block
// (c, j) walks observation chunk c and accumulates, in registers, only the
// observations of viewpoint j -- so the per-pose sums U_j / ea_j need no atomics
// and are bit-reproducible for any observation order; a chunk that holds no
// observation of j (the common case for the viewpoint-major order np.where
// produces) is skipped after two index reads.  The per-point sums V_i / eb_i are
// scattered with f64 atomics (few writers per point).
__global__ __launch_bounds__(kBlock) void k_ba_block_reduce(const double *__restrict__ poses,
                                                            const double *__restrict__ points,
                                                            const double *__restrict__ x_true,
                                                            const int64_t *__restrict__ vp,
                                                            const int64_t *__restrict__ pt, int64_t n,
                                                            int64_t chunk, int sorted_by_viewpoint,
                                                            double *__restrict__ V, double *__restrict__ eb,
                                                            double *__restrict__ partials) {
    const int64_t j = blockIdx.y;
    const int64_t start = blockIdx.x * chunk;
    const int64_t end = min(n, start + chunk);
    double acc[kPoseAcc];
#pragma unroll
    for (int i = 0; i < kPoseAcc; i++) acc[i] = 0.0;

    bool skip = false;
    if (sorted_by_viewpoint) skip = (vp[start] > j) || (vp[end - 1] < j);
    if (!skip) {
        double pose[6];
#pragma unroll
        for (int i = 0; i < 6; i++) pose[i] = poses[6 * j + i];
        for (int64_t k = start + threadIdx.x; k < end; k += kBlock) {
            if (vp[k] != j) continue;
            int64_t ip = pt[k];
            double p[3] = {points[3 * ip], points[3 * ip + 1], points[3 * ip + 2]};
            double x[2], A[12], B[6];
            project_observation<true>(pose, p, x, A, B);
            double e0 = x_true[2 * k] - x[0], e1 = x_true[2 * k + 1] - x[1];
            acc[27] += e0 * e0 + e1 * e1;
            int m = 0;
#pragma unroll
            for (int a = 0; a < 6; a++) {
#pragma unroll
                for (int b = a; b < 6; b++) acc[m++] += A[a] * A[b] + A[6 + a] * A[6 + b];
                acc[21 + a] += A[a] * e0 + A[6 + a] * e1;
            }
            m = 0;
#pragma unroll
            for (int a = 0; a < 3; a++) {
#pragma unroll
                for (int b = a; b < 3; b++) atomicAdd(&V[6 * ip + m++], B[a] * B[b] + B[3 + a] * B[3 + b]);
                atomicAdd(&eb[3 * ip + a], B[a] * e0 + B[3 + a] * e1);
            }
        }
    }
    __shared__ double red[kBlock / 64][kPoseAccPad];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < kPoseAcc; i++) {
        double s = acc[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) red[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < kPoseAcc) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; w++) s += red[w][threadIdx.x];
        partials[((int64_t)j * gridDim.x + blockIdx.x) * kPoseAccPad + threadIdx.x] = s;
    }
}

__global__ __launch_bounds__(kBlock) void k_ba_finish(const double *__restrict__ partials, int nchunks,
                                                      double *__restrict__ U, double *__restrict__ ea,
                                                      double *__restrict__ err_per_pose) {
    const int j = blockIdx.x;
    __shared__ double red[kBlock / 32][kPoseAccPad];
    const int k = threadIdx.x & 31, g = threadIdx.x >> 5;
    double s = 0.0;
    if (k < kPoseAcc)
        for (int b = g; b < nchunks; b += kBlock / 32) s += partials[((int64_t)j * nchunks + b) * kPoseAccPad + k];
    red[g][k] = s;
    __syncthreads();
    if (threadIdx.x < kPoseAcc) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < kBlock / 32; i++) t += red[i][threadIdx.x];
        if (threadIdx.x < 21) U[21 * j + threadIdx.x] = t;
        else if (threadIdx.x < 27) ea[6 * j + threadIdx.x - 21] = t;
        else err_per_pose[j] = t;
    }
}

tdk_status h2d(int slot, const void *host, size_t bytes, void **dev) {
    TDK_TRY(tdk::scratch(slot, bytes, dev));
    if (bytes) TDK_HIP(hipMemcpyAsync(*dev, host, bytes, hipMemcpyHostToDevice, tdk::stream()));
    return TDK_OK;
}

tdk_status check_indices(const int64_t *vp, const int64_t *pt, int64_t n, int64_t n_poses, int64_t n_points,
                         int *sorted) {
    bool s = true;
    for (int64_t k = 0; k < n; k++) {
        if (vp[k] < 0 || vp[k] >= n_poses || pt[k] < 0 || pt[k] >= n_points) {
            tdk::set_error("observation %lld has an index out of range", (long long)k);
            return TDK_ERR_INVALID_ARGUMENT;
        }
        if (k > 0 && vp[k] < vp[k - 1]) s = false;
    }
    if (sorted) *sorted = s ? 1 : 0;
    return TDK_OK;
}

}  // namespace

extern "C" {

tdk_status tdk_ba_projection(const double *poses, int64_t n_poses, const double *points, int64_t n_points,
                             const int64_t *vp, const int64_t *pt, int64_t n, double *x_pred, double *A,
                             double *B) {
    TDK_REQUIRE(n >= 0 && n_poses >= 0 && n_points >= 0, "negative size");
    if (n == 0) return tdk::ensure_device();
    TDK_REQUIRE(poses && points && vp && pt, "null pointer");
    TDK_TRY(check_indices(vp, pt, n, n_poses, n_points, nullptr));
    void *d_poses, *d_points, *d_vp, *d_pt, *d_x = nullptr, *d_A = nullptr, *d_B = nullptr;
    TDK_TRY(h2d(0, poses, (size_t)n_poses * 48, &d_poses));
    TDK_TRY(h2d(1, points, (size_t)n_points * 24, &d_points));
    TDK_TRY(h2d(2, vp, (size_t)n * 8, &d_vp));
    TDK_TRY(h2d(3, pt, (size_t)n * 8, &d_pt));
    if (x_pred) TDK_TRY(tdk::scratch(4, (size_t)n * 16, &d_x));
    if (A) TDK_TRY(tdk::scratch(5, (size_t)n * 96, &d_A));
    if (B) TDK_TRY(tdk::scratch(6, (size_t)n * 48, &d_B));
    k_ba_projection<<<grid_for(n), kBlock, 0, tdk::stream()>>>((const double *)d_poses, (const double *)d_points,
                                                               (const int64_t *)d_vp, (const int64_t *)d_pt, n,
                                                               (double *)d_x, (double *)d_A, (double *)d_B);
    TDK_LAUNCH_CHECK();
    if (x_pred) TDK_HIP(hipMemcpyAsync(x_pred, d_x, (size_t)n * 16, hipMemcpyDeviceToHost, tdk::stream()));
    if (A) TDK_HIP(hipMemcpyAsync(A, d_A, (size_t)n * 96, hipMemcpyDeviceToHost, tdk::stream()));
    if (B) TDK_HIP(hipMemcpyAsync(B, d_B, (size_t)n * 48, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipStreamSynchronize(tdk::stream()));
    return TDK_OK;
}

tdk_status tdk_ba_exp_so3(const double *rotvecs, int64_t n, double *R) {
    TDK_REQUIRE(n >= 0 && (n == 0 || (rotvecs && R)), "bad argument");
    if (n == 0) return tdk::ensure_device();
    void *d_in, *d_out;
    TDK_TRY(h2d(0, rotvecs, (size_t)n * 24, &d_in));
    TDK_TRY(tdk::scratch(1, (size_t)n * 72, &d_out));
    k_ba_exp_so3<<<grid_for(n), kBlock, 0, tdk::stream()>>>((const double *)d_in, n, (double *)d_out);
    TDK_LAUNCH_CHECK();
    TDK_HIP(hipMemcpyAsync(R, d_out, (size_t)n * 72, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipStreamSynchronize(tdk::stream()));
    return TDK_OK;
}

tdk_status tdk_ba_block_reduce(const double *poses, int64_t n_poses, const double *points, int64_t n_points,
                               const double *x_true, const int64_t *vp, const int64_t *pt, int64_t n, double *U,
                               double *ea, double *V, double *eb, double *err) {
    TDK_REQUIRE(n >= 0 && n_poses >= 1 && n_points >= 1 && n_poses <= 65535, "bad sizes");
    TDK_REQUIRE(poses && points && U && ea && V && eb && err && (n == 0 || (x_true && vp && pt)), "null pointer");
    int sorted = 0;
    TDK_TRY(check_indices(vp, pt, n, n_poses, n_points, &sorted));
    void *d_poses, *d_points, *d_xt, *d_vp, *d_pt, *d_U, *d_ea, *d_V, *d_eb, *d_part, *d_err;
    TDK_TRY(h2d(0, poses, (size_t)n_poses * 48, &d_poses));
    TDK_TRY(h2d(1, points, (size_t)n_points * 24, &d_points));
    TDK_TRY(h2d(2, x_true, (size_t)n * 16, &d_xt));
    TDK_TRY(h2d(3, vp, (size_t)n * 8, &d_vp));
    TDK_TRY(h2d(4, pt, (size_t)n * 8, &d_pt));
    TDK_TRY(tdk::scratch(5, (size_t)n_poses * 21 * 8, &d_U));
    TDK_TRY(tdk::scratch(6, (size_t)n_poses * 6 * 8, &d_ea));
    TDK_TRY(tdk::scratch(7, (size_t)n_points * 6 * 8, &d_V));
    TDK_TRY(tdk::scratch(8, (size_t)n_points * 3 * 8, &d_eb));
    TDK_TRY(tdk::scratch(10, (size_t)n_poses * 8, &d_err));
    int64_t chunk = 2048;
    int64_t nchunks = n > 0 ? (n + chunk - 1) / chunk : 1;
    if (nchunks > 4096) {
        nchunks = 4096;
        chunk = (n + nchunks - 1) / nchunks;
        nchunks = (n + chunk - 1) / chunk;
    }
    TDK_TRY(tdk::scratch(9, (size_t)n_poses * nchunks * kPoseAccPad * 8, &d_part));
    TDK_HIP(hipMemsetAsync(d_V, 0, (size_t)n_points * 6 * 8, tdk::stream()));
    TDK_HIP(hipMemsetAsync(d_eb, 0, (size_t)n_points * 3 * 8, tdk::stream()));
    if (n > 0) {
        dim3 grid((unsigned)nchunks, (unsigned)n_poses);
        k_ba_block_reduce<<<grid, kBlock, 0, tdk::stream()>>>(
            (const double *)d_poses, (const double *)d_points, (const double *)d_xt, (const int64_t *)d_vp,
            (const int64_t *)d_pt, n, chunk, sorted, (double *)d_V, (double *)d_eb, (double *)d_part);
        TDK_LAUNCH_CHECK();
    } else {
        TDK_HIP(hipMemsetAsync(d_part, 0, (size_t)n_poses * nchunks * kPoseAccPad * 8, tdk::stream()));
    }
    k_ba_finish<<<(unsigned)n_poses, kBlock, 0, tdk::stream()>>>((const double *)d_part, (int)nchunks, (double *)d_U,
                                                                 (double *)d_ea, (double *)d_err);
    TDK_LAUNCH_CHECK();
    TDK_HIP(hipMemcpyAsync(U, d_U, (size_t)n_poses * 21 * 8, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipMemcpyAsync(ea, d_ea, (size_t)n_poses * 6 * 8, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipMemcpyAsync(V, d_V, (size_t)n_points * 6 * 8, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipMemcpyAsync(eb, d_eb, (size_t)n_points * 3 * 8, hipMemcpyDeviceToHost, tdk::stream()));
    void *stage;
    TDK_TRY(tdk::pinned(3, (size_t)n_poses * 8, &stage));
    TDK_HIP(hipMemcpyAsync(stage, d_err, (size_t)n_poses * 8, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipStreamSynchronize(tdk::stream()));
    double e = 0.0;
    for (int64_t j = 0; j < n_poses; j++) e += ((const double *)stage)[j];
    *err = e;
    return TDK_OK;
}

}  // extern "C"
