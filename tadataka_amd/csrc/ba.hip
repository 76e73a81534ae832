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
#include "tdk_wave.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <utility>
#include <vector>

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
__device__ __forceinline__ void project_observation(const double *pose, const Rod &c, const double *p, double *x,
                                                    double *A, double *B) {
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

template <bool JAC>
__device__ __forceinline__ void project_observation(const double *pose, const double *p, double *x, double *A,
                                                    double *B) {
    Rod c;
    rodrigues_coeffs(pose, c);
    project_observation<JAC>(pose, c, p, x, A, B);
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

// Fused residual + Jacobians + block sums over per-pose SEGMENTS.  The
// observations are listed by viewpoint once per graph (CSR: obs_sorted, or the
// identity for the viewpoint-major order np.where produces); every block takes
// one segment of at most `seg_len` observations of ONE pose, so every block
// works, the pose lives in SGPRs and the per-pose sums U_j / ea_j need no
// atomics and are bit-reproducible for any observation order: the 28
// accumulators go through the transposed wave reduction, LDS across the four
// waves, one partial per block; k_ba_finish_seg adds a pose's partials in segment
// order.
//   MODE_ATOMIC_V : V_i / eb_i scattered with f64 atomics (stateless entry point)
//   MODE_STORE_B  : B_ij and the residual stored per observation (Bobs, [8][n]) for
//                   k_ba_point_sums
//   MODE_STORE_BW : ... and W_ij = A^T B (Wobs, [18][n]) for the Schur complement
//   MODE_ERROR    : sum ||x_true - x_pred||^2 only -- no Jacobians (the damping
//                   trials of the Levenberg-Marquardt loop, tdk_ba_error)
enum { MODE_ATOMIC_V = 0, MODE_STORE_B = 1, MODE_STORE_BW = 2, MODE_ERROR = 3 };

struct BaSeg {
    int pose, start, end, slot;   // [start, end) of the pose-sorted list; slot = index among the pose's segments
};

template <int MODE>
__global__ __launch_bounds__(kBlock) void k_ba_reduce_seg(const double *__restrict__ poses,
                                                          const double *__restrict__ points,
                                                          const double *__restrict__ x_true,
                                                          const int *__restrict__ obs_sorted,
                                                          const int *__restrict__ pt32,
                                                          const BaSeg *__restrict__ segs,
                                                          const int *__restrict__ seg_ptr, int64_t n,
                                                          double *__restrict__ V, double *__restrict__ eb,
                                                          double *__restrict__ Wobs, double *__restrict__ Bobs,
                                                          double *__restrict__ partials, int *__restrict__ ticket,
                                                          double *__restrict__ U, double *__restrict__ ea,
                                                          double *__restrict__ err_per_pose) {
    const BaSeg sg = segs[blockIdx.x];
    const int j = sg.pose;
    double acc[kPoseAcc];
#pragma unroll
    for (int i = 0; i < kPoseAcc; i++) acc[i] = 0.0;
    double pose[6];
#pragma unroll
    for (int i = 0; i < 6; i++) pose[i] = poses[6 * j + i];
    Rod rod;   // depends on the pose only: once per block, not per observation
    rodrigues_coeffs(pose, rod);
    for (int q = sg.start + (int)threadIdx.x; q < sg.end; q += kBlock) {
        const int k = obs_sorted ? obs_sorted[q] : q;
        const int ip = pt32[k];
        const double p[3] = {points[3 * (int64_t)ip], points[3 * (int64_t)ip + 1], points[3 * (int64_t)ip + 2]};
        double x[2], A[12], B[6];
        if (MODE == MODE_ERROR) project_observation<false>(pose, rod, p, x, A, B);
        else project_observation<true>(pose, rod, p, x, A, B);
        const double e0 = x_true[2 * (int64_t)k] - x[0], e1 = x_true[2 * (int64_t)k + 1] - x[1];
        acc[27] += e0 * e0 + e1 * e1;
        if (MODE == MODE_ERROR) continue;
        int m = 0;
#pragma unroll
        for (int a = 0; a < 6; a++) {
#pragma unroll
            for (int b = a; b < 6; b++) acc[m++] += A[a] * A[b] + A[6 + a] * A[6 + b];
            acc[21 + a] += A[a] * e0 + A[6 + a] * e1;
        }
        if (MODE == MODE_ATOMIC_V) {
            m = 0;
#pragma unroll
            for (int a = 0; a < 3; a++) {
#pragma unroll
                for (int b = a; b < 3; b++) atomicAdd(&V[6 * (int64_t)ip + m++], B[a] * B[b] + B[3 + a] * B[3 + b]);
                atomicAdd(&eb[3 * (int64_t)ip + a], B[a] * e0 + B[3 + a] * e1);
            }
        }
        if (MODE == MODE_STORE_B || MODE == MODE_STORE_BW) {
#pragma unroll
            for (int a = 0; a < 6; a++) Bobs[a * n + k] = B[a];
            Bobs[6 * n + k] = e0;
            Bobs[7 * n + k] = e1;
        }
        if (MODE == MODE_STORE_BW) {   // W_ij = A^T B (6x3), kept for the Schur complement
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int b = 0; b < 3; b++) Wobs[(3 * a + b) * n + k] = A[a] * B[b] + A[6 + a] * B[3 + b];
        }
    }
    __shared__ double red[kBlock / 64][kPoseAccPad];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (MODE == MODE_ERROR) {
        double v = acc[27];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) red[wave][27] = v;
    } else {
        const double v = tdk::wave_sum_transposed(acc);
        if ((lane & 1) == 0) red[wave][lane >> 1] = v;
    }
    __syncthreads();
    const int first = seg_ptr[j];
    const int t0 = MODE == MODE_ERROR ? 27 : 0;
    if ((int)threadIdx.x >= t0 && threadIdx.x < kPoseAcc) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; w++) v += red[w][threadIdx.x];
        partials[(int64_t)(first + sg.slot) * kPoseAccPad + threadIdx.x] = v;
    }
}

// Fixed-order sum of a pose's segment partials (bit-reproducible).  A separate launch
// on the same stream, not a "last block done" tail inside k_ba_reduce_seg: a device-scope
// release fence per block writes the XCD's L2 back (buffer_wbl2) and, at ~1000 blocks
// that each stored a share of B / W, that cost more than the whole reduction
// (measured 214 us at 1563 blocks against 45 us at 196).
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_ba_finish_seg(const double *__restrict__ partials,
                                                          const int *__restrict__ seg_ptr, double *__restrict__ U,
                                                          double *__restrict__ ea,
                                                          double *__restrict__ err_per_pose) {
    const int j = blockIdx.x;
    const int first = seg_ptr[j], nseg = seg_ptr[j + 1] - first;
    const int t0 = MODE == MODE_ERROR ? 27 : 0;
    const int kk = threadIdx.x & 31, g = threadIdx.x >> 5;
    double v = 0.0;
    if (kk >= t0 && kk < kPoseAcc)
        for (int b = g; b < nseg; b += kBlock / 32) v += partials[(int64_t)(first + b) * kPoseAccPad + kk];
    __shared__ double fin[kBlock / 32][kPoseAccPad];
    fin[g][kk] = v;
    __syncthreads();
    if (threadIdx.x < kPoseAcc) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < kBlock / 32; i++) t += fin[i][threadIdx.x];
        if (threadIdx.x == 27) err_per_pose[j] = t;
        else if (MODE != MODE_ERROR) {
            if (threadIdx.x < 21) U[21 * j + threadIdx.x] = t;
            else ea[6 * j + threadIdx.x - 21] = t;
        }
    }
}

// ---------------------------------------------------------------------------
// Sparse bundle adjustment step (Lourakis & Argyros' SBA, the formulation the
// reference delegates to the third-party `sparseba` package, call site
// tadataka/local_ba.py:72,77): with U_j, V_i, W_ij = A_ij^T B_ij and the
// gradients ea_j, eb_i of the block reduce, damped by mu on the diagonals,
//   Y_ij = W_ij V*_i^-1,   S_jk = delta_jk U*_j - sum_i Y_ij W_ik^T,
//   e_j  = ea_j - sum_i Y_ij eb_i,   S da = e,
//   db_i = V*_i^-1 (eb_i - sum_j W_ij^T da_j).
// ---------------------------------------------------------------------------
constexpr int kMaxLdsPoses = 12;   // (6 * 12)^2 doubles = 41 KB of LDS for the private S

// V_i = sum_j B_ij^T B_ij (upper triangle, 6) and eb_i = sum_j B_ij^T e_ij: one thread
// per point walks its observation list in increasing observation index -- no
// atomics, bit-reproducible.  Arrays per point are structure-of-arrays ([6][Q], [3][Q]).
__global__ __launch_bounds__(kBlock) void k_ba_point_sums(const int64_t *__restrict__ row_ptr,
                                                          const int64_t *__restrict__ obs_of_point,
                                                          const double *__restrict__ Bobs, int64_t n,
                                                          int64_t n_points, double *__restrict__ V,
                                                          double *__restrict__ eb) {
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n_points; i += (int64_t)gridDim.x * kBlock) {
        double v[6] = {0., 0., 0., 0., 0., 0.}, g[3] = {0., 0., 0.};
        for (int64_t pa = row_ptr[i]; pa < row_ptr[i + 1]; pa++) {
            const int64_t k = obs_of_point[pa];
            double B[6];
#pragma unroll
            for (int a = 0; a < 6; a++) B[a] = Bobs[a * n + k];
            const double e0 = Bobs[6 * n + k], e1 = Bobs[7 * n + k];
            int m = 0;
#pragma unroll
            for (int a = 0; a < 3; a++) {
#pragma unroll
                for (int b = a; b < 3; b++) v[m++] += B[a] * B[b] + B[3 + a] * B[3 + b];
                g[a] += B[a] * e0 + B[3 + a] * e1;
            }
        }
#pragma unroll
        for (int m = 0; m < 6; m++) V[m * n_points + i] = v[m];
#pragma unroll
        for (int a = 0; a < 3; a++) eb[a * n_points + i] = g[a];
    }
}

// W_ij = A_ij^T B_ij from the parameters instead of from memory (round 4).  The Schur complement and the
// back-substitution used to READ W (18 doubles per observation, written by the block reduce): 58 MB written
// and twice 58 MB read per damping trial of the 8 x 50 000 window, ~35 us of a 120 us trial at HBM speed --
// while recomputing the two Jacobians of an observation is ~300 FP64 operations, 3 us for all 400 000.  So the
// reduce stores only B (for the per-point sums) and these kernels rebuild W where they need it, with the
// pose-only part of the Rodrigues formula precomputed per pose (rod8: A, B, dA, dB, th[3]) by the first block
// of k_ba_invert_V, which runs before both.  TDK_BA_W=stored keeps the stored W (and is what the pair-wise
// and general Schur kernels always use).
__device__ __forceinline__ void recompute_W(const double *__restrict__ poses, const double *__restrict__ rod8, int j,
                                            const double *__restrict__ points, int64_t i, double *W) {
    double pose[6], p[3], x[2], A[12], B[6];
#pragma unroll
    for (int k = 0; k < 6; k++) pose[k] = poses[6 * j + k];
    Rod c;
    c.A = rod8[8 * j]; c.B = rod8[8 * j + 1]; c.dA = rod8[8 * j + 2]; c.dB = rod8[8 * j + 3];
    c.th[0] = rod8[8 * j + 4]; c.th[1] = rod8[8 * j + 5]; c.th[2] = rod8[8 * j + 6];
#pragma unroll
    for (int k = 0; k < 3; k++) p[k] = points[3 * i + k];
    project_observation<true>(pose, c, p, x, A, B);
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) W[3 * a + b] = A[a] * B[b] + A[6 + a] * B[3 + b];
}

// inverse of the damped symmetric 3x3 V (upper triangle in, upper triangle out); block 0 also leaves the
// pose-only Rodrigues coefficients of the CURRENT poses for the kernels that rebuild W (rod8 != nullptr)
__global__ __launch_bounds__(kBlock) void k_ba_invert_V(const double *__restrict__ V, double mu, int64_t n_points,
                                                        double *__restrict__ Vinv, const double *__restrict__ poses,
                                                        int n_poses, double *__restrict__ rod8) {
    if (rod8 != nullptr && blockIdx.x == 0 && (int)threadIdx.x < n_poses) {
        double w[3] = {poses[6 * threadIdx.x], poses[6 * threadIdx.x + 1], poses[6 * threadIdx.x + 2]};
        Rod c;
        rodrigues_coeffs(w, c);
        double *o = rod8 + 8 * threadIdx.x;
        o[0] = c.A; o[1] = c.B; o[2] = c.dA; o[3] = c.dB; o[4] = c.th[0]; o[5] = c.th[1]; o[6] = c.th[2]; o[7] = 0.0;
    }
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n_points; i += (int64_t)gridDim.x * kBlock) {
        const int64_t Q = n_points;
        double a = V[i] + mu, b = V[Q + i], c = V[2 * Q + i];
        double d = V[3 * Q + i] + mu, e = V[4 * Q + i], f = V[5 * Q + i] + mu;
        double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
        double det = a * c00 + b * c01 + c * c02;
        double id = 1.0 / det;
        Vinv[i] = c00 * id;
        Vinv[Q + i] = c01 * id;
        Vinv[2 * Q + i] = c02 * id;
        Vinv[3 * Q + i] = (a * f - c * c) * id;
        Vinv[4 * Q + i] = (b * c - a * e) * id;
        Vinv[5 * Q + i] = (a * d - b * b) * id;
    }
}

// Schur complement by pose pair.  Block (chunk, pair) takes one block (ja <= jb)
// of the upper block triangle of S and a range of points; a thread looks up the
// two observations of its point in the dense table obs_at[pose][point] (-1: not
// seen), forms Y = W_a V^-1 and adds Y W_b^T to 36 register accumulators (plus
// Y eb for the 6 entries of e_ja on diagonal pairs).  No atomics: per-block
// partials, summed in block order by k_ba_schur_finish, so S is bit-reproducible.
// W is read as structure of arrays, consecutive points -> consecutive
// observations for the usual pose-major order, i.e. coalesced; it is re-read
// once per pair, from L2 / Infinity Cache after the first pass.
constexpr int kSchurAcc = 42, kSchurAccPad = 48;

__global__ __launch_bounds__(kBlock) void k_ba_schur_pairs(const int *__restrict__ obs_at,
                                                           const double *__restrict__ Wobs,
                                                           const double *__restrict__ Vinv,
                                                           const double *__restrict__ eb, int64_t n,
                                                           int64_t n_points, int n_poses, int64_t chunk,
                                                           double *__restrict__ partials) {
    // pair index -> (ja, jb), row-major over the upper triangle
    int ja = 0, rem = blockIdx.y;
    while (rem >= n_poses - ja) { rem -= n_poses - ja; ja++; }
    const int jb = ja + rem;
    const int64_t Q = n_points;
    const int64_t start = blockIdx.x * chunk, end = min(Q, start + chunk);
    double acc[kSchurAcc];
#pragma unroll
    for (int i = 0; i < kSchurAcc; i++) acc[i] = 0.0;
    for (int64_t i = start + threadIdx.x; i < end; i += kBlock) {
        const int ka = obs_at[(int64_t)ja * Q + i], kb = obs_at[(int64_t)jb * Q + i];
        if (ka < 0 || kb < 0) continue;
        const double vi[6] = {Vinv[i], Vinv[Q + i], Vinv[2 * Q + i], Vinv[3 * Q + i], Vinv[4 * Q + i], Vinv[5 * Q + i]};
        double Y[18];
#pragma unroll
        for (int r = 0; r < 6; r++) {
            const double w0 = Wobs[(3 * r) * n + ka], w1 = Wobs[(3 * r + 1) * n + ka], w2 = Wobs[(3 * r + 2) * n + ka];
            Y[3 * r] = w0 * vi[0] + w1 * vi[1] + w2 * vi[2];
            Y[3 * r + 1] = w0 * vi[1] + w1 * vi[3] + w2 * vi[4];
            Y[3 * r + 2] = w0 * vi[2] + w1 * vi[4] + w2 * vi[5];
        }
#pragma unroll
        for (int c = 0; c < 6; c++) {
            const double w0 = Wobs[(3 * c) * n + kb], w1 = Wobs[(3 * c + 1) * n + kb], w2 = Wobs[(3 * c + 2) * n + kb];
#pragma unroll
            for (int r = 0; r < 6; r++) acc[6 * r + c] += Y[3 * r] * w0 + Y[3 * r + 1] * w1 + Y[3 * r + 2] * w2;
        }
        if (ja == jb) {
            const double e0 = eb[i], e1 = eb[Q + i], e2 = eb[2 * Q + i];
#pragma unroll
            for (int r = 0; r < 6; r++) acc[36 + r] += Y[3 * r] * e0 + Y[3 * r + 1] * e1 + Y[3 * r + 2] * e2;
        }
    }
    __shared__ double red[kBlock / 64][kSchurAccPad];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < kSchurAcc; i++) {
        double v = acc[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < kSchurAcc) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; w++) v += red[w][threadIdx.x];
        partials[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * kSchurAccPad + threadIdx.x] = v;
    }
}

// S(ja, jb) = -sum of the chunk partials (fixed order); e_ja likewise on diagonal pairs
__global__ void k_ba_schur_finish(const double *__restrict__ partials, int nchunks, int n_poses, int dim,
                                  double *__restrict__ S, double *__restrict__ evec) {
    int ja = 0, rem = blockIdx.x;
    while (rem >= n_poses - ja) { rem -= n_poses - ja; ja++; }
    const int jb = ja + rem;
    const int t = threadIdx.x;
    if (t >= kSchurAcc) return;
    double v = 0.0;
    for (int b = 0; b < nchunks; b++) v += partials[((int64_t)blockIdx.x * nchunks + b) * kSchurAccPad + t];
    if (t < 36) S[(size_t)(6 * ja + t / 6) * dim + 6 * jb + t % 6] = -v;
    else if (ja == jb) evec[6 * ja + t - 36] = -v;
}

// ---------------------------------------------------------------------------
// Schur complement as a symmetric rank-k update on the FP64 matrix cores.
//
// S - blockdiag(U*) = -sum_i W_i V*_i^-1 W_i^T with W_i the (6 P) x 3 stack of the W_ij
// of point i.  With the Cholesky factor V*_i^-1 = L_i L_i^T this is -Z Z^T,
// Z = [W_1 L_1 | W_2 L_2 | ...] of shape (6 P) x (3 Q): one dense contraction over
// 150 000 columns for the 8 x 50 000 window -- GEMM-shaped, unlike anything on the DVO
// path, so it goes to v_mfma_f64_16x16x4_f64.  (The FP64 MFMA rate equals the vector
// rate on this part; the gain is that every W_ij is read from memory ONCE and then
// served from LDS to all pose pairs, where the pair-wise kernel re-reads it P times
// through L2: 99-113 us -> measured below.)
//
// A block stages a chunk of kSchurPC points: thread (pose j, point i) loads W_ij
// (18 coalesced plane reads), factors V*_i^-1 and writes the three columns of
// W_ij L_i into the LDS tile Z[16 RB][3 kSchurPC]; rows beyond 6 P and points beyond
// Q are zero.  Each wave then takes every fourth K-step of 4 columns: ONE LDS read
// per 16-row block gives the operand for that block both as A (rows) and as B
// (columns: the update is symmetric), and the RB (RB + 1) / 2 upper tiles are
// accumulated in 4 registers each.  Blocks loop over chunks with the accumulators
// live (persistent grid), the four waves are added in wave order and one partial
// per block goes out; k_ba_schur_mfma_finish adds the partials in block order -- no
// atomics, bit-reproducible.  e_j -= sum_i W_ij (V*_i^-1 e_b,i) rides along: 6
// values per thread, reduced over the chunk in LDS in a fixed order.
// ---------------------------------------------------------------------------
constexpr int kSchurPC = 32;                   // points per chunk
constexpr int kSchurK = 3 * kSchurPC;          // columns of Z per chunk
constexpr int kSchurPitch = kSchurK + 4;       // 100 doubles: the 64 operand lanes spread over all banks, 2 per bank
typedef double schur_acc_t __attribute__((ext_vector_type(4)));

template <int RB, bool RECOMP>
__global__ __launch_bounds__(kBlock) void k_ba_schur_mfma(const int *__restrict__ obs_at,
                                                          const double *__restrict__ Wobs,
                                                          const double *__restrict__ Vinv,
                                                          const double *__restrict__ eb, int64_t n,
                                                          int64_t n_points, int n_poses, int n_chunks,
                                                          double *__restrict__ partials,
                                                          const double *__restrict__ poses,
                                                          const double *__restrict__ rod8,
                                                          const double *__restrict__ points) {
    constexpr int kTiles = RB * (RB + 1) / 2;
    constexpr int kRows = 16 * RB;
    __shared__ double Zs[kRows * kSchurPitch];
    __shared__ double Es[kRows * kSchurPC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t Q = n_points;
    const int j = threadIdx.x / kSchurPC, il = threadIdx.x - j * kSchurPC;   // fill role: (pose, point of the chunk)
    schur_acc_t acc[kTiles];
#pragma unroll
    for (int t = 0; t < kTiles; t++) acc[t] = (schur_acc_t){0.0, 0.0, 0.0, 0.0};
    double e_acc = 0.0;                            // threads 0 .. 6 P - 1: row threadIdx.x of e
    for (int chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        __syncthreads();                           // the previous chunk's tile has been consumed
        for (int k = threadIdx.x; k < kRows * kSchurPitch; k += kBlock) Zs[k] = 0.0;
        for (int k = threadIdx.x; k < kRows * kSchurPC; k += kBlock) Es[k] = 0.0;
        __syncthreads();
        const int64_t i = (int64_t)chunk * kSchurPC + il;
        if (j < n_poses && i < Q) {
            const int ko = obs_at[(int64_t)j * Q + i];
            if (ko >= 0) {
                const double v00 = Vinv[i], v01 = Vinv[Q + i], v02 = Vinv[2 * Q + i];
                const double v11 = Vinv[3 * Q + i], v12 = Vinv[4 * Q + i], v22 = Vinv[5 * Q + i];
                // V*^-1 = L L^T (3x3 Cholesky)
                const double l00 = sqrt(v00), l10 = v01 / l00, l20 = v02 / l00;
                const double l11 = sqrt(v11 - l10 * l10), l21 = (v12 - l20 * l10) / l11;
                const double l22 = sqrt(v22 - l20 * l20 - l21 * l21);
                const double b0 = eb[i], b1 = eb[Q + i], b2 = eb[2 * Q + i];
                const double u0 = v00 * b0 + v01 * b1 + v02 * b2;   // V*^-1 e_b
                const double u1 = v01 * b0 + v11 * b1 + v12 * b2;
                const double u2 = v02 * b0 + v12 * b1 + v22 * b2;
                double Wr[18];
                if (RECOMP) recompute_W(poses, rod8, j, points, i, Wr);
#pragma unroll
                for (int a = 0; a < 6; a++) {
                    const double w0 = RECOMP ? Wr[3 * a] : Wobs[(3 * a) * n + ko];
                    const double w1 = RECOMP ? Wr[3 * a + 1] : Wobs[(3 * a + 1) * n + ko];
                    const double w2 = RECOMP ? Wr[3 * a + 2] : Wobs[(3 * a + 2) * n + ko];
                    double *z = &Zs[(6 * j + a) * kSchurPitch + 3 * il];
                    z[0] = w0 * l00 + w1 * l10 + w2 * l20;
                    z[1] = w1 * l11 + w2 * l21;
                    z[2] = w2 * l22;
                    Es[(6 * j + a) * kSchurPC + il] = w0 * u0 + w1 * u1 + w2 * u2;
                }
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < 6 * n_poses) {      // e: the chunk's points in order
            double v = 0.0;
#pragma unroll 8
            for (int k = 0; k < kSchurPC; k++) v += Es[threadIdx.x * kSchurPC + k];
            e_acc += v;
        }
        for (int ks = wave; ks < kSchurK / 4; ks += kBlock / 64) {
            double op[RB];
#pragma unroll
            for (int rb = 0; rb < RB; rb++) op[rb] = Zs[(16 * rb + (lane & 15)) * kSchurPitch + 4 * ks + (lane >> 4)];
            int t = 0;
#pragma unroll
            for (int r = 0; r < RB; r++)
#pragma unroll
                for (int c = r; c < RB; c++) {
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[r], op[c], acc[t], 0, 0, 0);
                    t++;
                }
        }
    }
    // the four waves' accumulators, added in wave order through LDS (reuses the Z tile)
    __syncthreads();
    double *sum = Zs;                              // [kTiles][4][64]
    for (int w = 0; w < kBlock / 64; w++) {
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < kTiles; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    double *p = &sum[(t * 4 + r) * 64 + lane];
                    *p = (w == 0 ? 0.0 : *p) + acc[t][r];
                }
        }
        __syncthreads();
    }
    double *out = partials + (size_t)blockIdx.x * (kTiles * 256 + 64);
    for (int k = threadIdx.x; k < kTiles * 256; k += kBlock) out[k] = sum[k];
    if (threadIdx.x < 64) out[kTiles * 256 + threadIdx.x] = (int)threadIdx.x < 6 * n_poses ? e_acc : 0.0;
}

// S(row, col) = -(sum of the block partials, in block order), written symmetrically;
// e(row) likewise.  Tile t = (r, c), r <= c; register reg of lane l holds
// D[row = (l >> 4) + 4 reg][col = l & 15] (the f64 MFMA C/D layout).
template <int RB>
__global__ __launch_bounds__(kBlock) void k_ba_schur_mfma_finish(const double *__restrict__ partials, int n_blocks,
                                                                 int dim, double *__restrict__ S,
                                                                 double *__restrict__ evec) {
    constexpr int kTiles = RB * (RB + 1) / 2;
    constexpr int kPer = kTiles * 256 + 64;
    // 32 values per block of 256 threads: slice s of 8 adds blocks s, s + 8, ... (its loads are
    // independent: several in flight), then the 8 slices are added in slice order
    __shared__ double part[8][32];
    const int kk = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int k = blockIdx.x * 32 + kk;
    double v = 0.0;
    if (k < kPer) {
#pragma unroll 8
        for (int b = sl; b < n_blocks; b += 8) v += partials[(size_t)b * kPer + k];
    }
    part[sl][kk] = v;
    __syncthreads();
    if (sl != 0 || k >= kPer) return;
    v = part[0][kk];
#pragma unroll
    for (int q = 1; q < 8; q++) v += part[q][kk];
    if (k >= kTiles * 256) {
        const int row = k - kTiles * 256;
        if (row < dim) evec[row] = -v;
        return;
    }
    const int t = k >> 8, reg = (k >> 6) & 3, l = k & 63;
    int r = 0, rem = t;
    while (rem >= RB - r) { rem -= RB - r; r++; }
    const int c = r + rem;
    const int row = 16 * r + (l >> 4) + 4 * reg, col = 16 * c + (l & 15);
    if (row >= dim || col >= dim) return;
    if (r == c && row > col) return;               // the diagonal tiles hold both triangles: keep the upper one
    S[(size_t)row * dim + col] = -v;
    S[(size_t)col * dim + row] = -v;
}

// General fallback (too many poses x points for the dense observation table, or
// duplicate observations): one thread per point walks the point's observation
// list (CSR) and subtracts Y_ij W_ik^T from the (j, k) block of S for every
// pair of its observers and Y_ij eb_i from e_j, with atomics -- into an
// LDS-private copy of S per block when it fits, else into global memory.
template <bool LDS_S>
__global__ __launch_bounds__(kBlock) void k_ba_schur(const int64_t *__restrict__ row_ptr,
                                                     const int64_t *__restrict__ obs_of_point,
                                                     const int64_t *__restrict__ vp,
                                                     const double *__restrict__ Wobs,
                                                     const double *__restrict__ Vinv,
                                                     const double *__restrict__ eb, int64_t n,
                                                     int64_t n_points, int dim, double *__restrict__ S,
                                                     double *__restrict__ evec) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *Sl = reinterpret_cast<double *>(smem);
    if (LDS_S) {
        for (int i = threadIdx.x; i < dim * dim + dim; i += kBlock) Sl[i] = 0.0;
        __syncthreads();
    }
    double *Sacc = LDS_S ? Sl : S;
    double *eacc = LDS_S ? Sl + dim * dim : evec;
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n_points; i += (int64_t)gridDim.x * kBlock) {
        const int64_t b0 = row_ptr[i], b1 = row_ptr[i + 1];
        const int64_t Q = n_points;
        const double vi[6] = {Vinv[i], Vinv[Q + i], Vinv[2 * Q + i], Vinv[3 * Q + i], Vinv[4 * Q + i], Vinv[5 * Q + i]};
        const double ebi[3] = {eb[i], eb[Q + i], eb[2 * Q + i]};
        for (int64_t pa = b0; pa < b1; pa++) {
            const int64_t ka = obs_of_point[pa];
            const int ja = (int)vp[ka];
            double Y[18];
#pragma unroll
            for (int r = 0; r < 6; r++) {
                double w0 = Wobs[(3 * r) * n + ka], w1 = Wobs[(3 * r + 1) * n + ka], w2 = Wobs[(3 * r + 2) * n + ka];
                Y[3 * r] = w0 * vi[0] + w1 * vi[1] + w2 * vi[2];
                Y[3 * r + 1] = w0 * vi[1] + w1 * vi[3] + w2 * vi[4];
                Y[3 * r + 2] = w0 * vi[2] + w1 * vi[4] + w2 * vi[5];
            }
#pragma unroll
            for (int r = 0; r < 6; r++)
                atomicAdd(&eacc[6 * ja + r], -(Y[3 * r] * ebi[0] + Y[3 * r + 1] * ebi[1] + Y[3 * r + 2] * ebi[2]));
            for (int64_t pb = b0; pb < b1; pb++) {
                const int64_t kb = obs_of_point[pb];
                const int jb = (int)vp[kb];
                if (jb < ja) continue;   // upper block triangle; mirrored on the host
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    double w0 = Wobs[(3 * c) * n + kb], w1 = Wobs[(3 * c + 1) * n + kb], w2 = Wobs[(3 * c + 2) * n + kb];
#pragma unroll
                    for (int r = 0; r < 6; r++)
                        atomicAdd(&Sacc[(6 * ja + r) * dim + 6 * jb + c],
                                  -(Y[3 * r] * w0 + Y[3 * r + 1] * w1 + Y[3 * r + 2] * w2));
                }
            }
        }
    }
    if (LDS_S) {
        __syncthreads();
        for (int i = threadIdx.x; i < dim * dim; i += kBlock)
            if (Sl[i] != 0.0) atomicAdd(&S[i], Sl[i]);
        for (int i = threadIdx.x; i < dim; i += kBlock)
            if (Sl[dim * dim + i] != 0.0) atomicAdd(&evec[i], Sl[dim * dim + i]);
    }
}

template <bool RECOMP>
__global__ __launch_bounds__(kBlock) void k_ba_backsub(const int64_t *__restrict__ row_ptr,
                                                       const int64_t *__restrict__ obs_of_point,
                                                       const int64_t *__restrict__ vp,
                                                       const double *__restrict__ Wobs,
                                                       const double *__restrict__ Vinv,
                                                       const double *__restrict__ eb,
                                                       const double *__restrict__ da, int64_t n,
                                                       int64_t n_points, const double *__restrict__ points,
                                                       double *__restrict__ db, double *__restrict__ cpoints,
                                                       const double *__restrict__ poses,
                                                       const double *__restrict__ rod8) {
    const int64_t Q = n_points;
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n_points; i += (int64_t)gridDim.x * kBlock) {
        double g[3] = {eb[i], eb[Q + i], eb[2 * Q + i]};
        for (int64_t pa = row_ptr[i]; pa < row_ptr[i + 1]; pa++) {
            const int64_t k = obs_of_point[pa];
            const int jp = (int)vp[k];
            const double *d = da + 6 * jp;
            if (RECOMP) {
                double Wr[18];
                recompute_W(poses, rod8, jp, points, i, Wr);
#pragma unroll
                for (int r = 0; r < 6; r++) {
                    g[0] -= Wr[3 * r] * d[r];
                    g[1] -= Wr[3 * r + 1] * d[r];
                    g[2] -= Wr[3 * r + 2] * d[r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 6; r++) {
                    g[0] -= Wobs[(3 * r) * n + k] * d[r];
                    g[1] -= Wobs[(3 * r + 1) * n + k] * d[r];
                    g[2] -= Wobs[(3 * r + 2) * n + k] * d[r];
                }
            }
        }
        const double v[6] = {Vinv[i], Vinv[Q + i], Vinv[2 * Q + i], Vinv[3 * Q + i], Vinv[4 * Q + i], Vinv[5 * Q + i]};
        const double d0 = v[0] * g[0] + v[1] * g[1] + v[2] * g[2];
        const double d1 = v[1] * g[0] + v[3] * g[1] + v[4] * g[2];
        const double d2 = v[2] * g[0] + v[4] * g[1] + v[5] * g[2];
        db[3 * i] = d0; db[3 * i + 1] = d1; db[3 * i + 2] = d2;
        // the candidate of the damping trial: points + db
        cpoints[3 * i] = points[3 * i] + d0;
        cpoints[3 * i + 1] = points[3 * i + 1] + d1;
        cpoints[3 * i + 2] = points[3 * i + 2] + d2;
    }
}

// dense solve of the reduced camera system (host, dimension 6 * n_poses) by
// Gaussian elimination with partial pivoting: S is positive definite in exact
// arithmetic, but with small damping and near-degenerate points its computed
// Schur complement can lose definiteness by rounding, which LU tolerates.
int dense_solve(std::vector<double> &M, std::vector<double> &rhs, int n) {
    for (int c = 0; c < n; c++) {
        int p = c;
        double best = fabs(M[(size_t)c * n + c]);
        for (int r = c + 1; r < n; r++) {
            double v = fabs(M[(size_t)r * n + c]);
            if (v > best) { best = v; p = r; }
        }
        if (!(best > 0.0)) return -1;
        if (p != c) {
            for (int k = 0; k < n; k++) std::swap(M[(size_t)c * n + k], M[(size_t)p * n + k]);
            std::swap(rhs[(size_t)c], rhs[(size_t)p]);
        }
        const double piv = M[(size_t)c * n + c];
        for (int r = c + 1; r < n; r++) {
            const double f = M[(size_t)r * n + c] / piv;
            if (f == 0.0) continue;
            for (int k = c; k < n; k++) M[(size_t)r * n + k] -= f * M[(size_t)c * n + k];
            rhs[(size_t)r] -= f * rhs[(size_t)c];
        }
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = rhs[(size_t)i];
        for (int k = i + 1; k < n; k++) s -= M[(size_t)i * n + k] * rhs[(size_t)k];
        rhs[(size_t)i] = s / M[(size_t)i * n + i];
    }
    return 0;
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

// Observation lists by viewpoint and their split into block-sized segments (see k_ba_reduce_seg).
struct BaPlan {
    std::vector<int> obs_sorted;   // empty: the observations already are viewpoint-major
    std::vector<int> pt32;
    std::vector<BaSeg> segs;
    std::vector<int> seg_ptr;      // [n_poses + 1]
};

void ba_make_plan(const int64_t *vp, const int64_t *pt, int64_t n, int64_t n_poses, bool sorted, BaPlan *plan) {
    plan->pt32.resize((size_t)n);
    for (int64_t k = 0; k < n; k++) plan->pt32[(size_t)k] = (int)pt[k];
    std::vector<int64_t> pose_ptr((size_t)n_poses + 1, 0);
    for (int64_t k = 0; k < n; k++) pose_ptr[(size_t)vp[k] + 1]++;
    for (int64_t j = 0; j < n_poses; j++) pose_ptr[(size_t)j + 1] += pose_ptr[(size_t)j];
    plan->obs_sorted.clear();
    if (!sorted) {   // counting sort by viewpoint, observation order kept inside a pose
        plan->obs_sorted.resize((size_t)n);
        std::vector<int64_t> cursor(pose_ptr.begin(), pose_ptr.end() - 1);
        for (int64_t k = 0; k < n; k++) plan->obs_sorted[(size_t)cursor[(size_t)vp[k]]++] = (int)k;
    }
    // aim at >= ~1024 blocks (4 per CU) without going below one observation per thread
    int64_t len = (n / 1024 + kBlock - 1) / kBlock * kBlock;
    if (len < kBlock) len = kBlock;
    if (len > 8 * kBlock) len = 8 * kBlock;
    plan->segs.clear();
    plan->seg_ptr.assign((size_t)n_poses + 1, 0);
    for (int64_t j = 0; j < n_poses; j++) {
        const int64_t b0 = pose_ptr[(size_t)j], b1 = pose_ptr[(size_t)j + 1];
        int slot = 0;
        // a pose without observations still gets one (empty) segment: its block writes the zeros
        for (int64_t q = b0; q < b1 || slot == 0; q += len) {
            BaSeg sg;
            sg.pose = (int)j; sg.start = (int)q; sg.end = (int)(q + len < b1 ? q + len : b1); sg.slot = slot++;
            plan->segs.push_back(sg);
        }
        plan->seg_ptr[(size_t)j + 1] = (int)plan->segs.size();
    }
}

template <int MODE>
void launch_reduce_seg(const double *d_poses, const double *d_points, const double *d_xt, const int *d_obs_sorted,
                       const int *d_pt32, const BaSeg *d_segs, const int *d_seg_ptr, int n_segs, int n_poses, int64_t n,
                       double *d_V, double *d_eb, double *d_W, double *d_Be, double *d_part, int *d_ticket,
                       double *d_U, double *d_ea, double *d_err, hipStream_t stream) {
    k_ba_reduce_seg<MODE><<<(unsigned)n_segs, kBlock, 0, stream>>>(d_poses, d_points, d_xt, d_obs_sorted, d_pt32, d_segs,
                                                                   d_seg_ptr, n, d_V, d_eb, d_W, d_Be, d_part,
                                                                   d_ticket, d_U, d_ea, d_err);
    k_ba_finish_seg<MODE><<<(unsigned)n_poses, kBlock, 0, stream>>>(d_part, d_seg_ptr, d_U, d_ea, d_err);
}

}  // namespace

extern "C" {

tdk_status tdk_ba_projection(const double *poses, int64_t n_poses, const double *points, int64_t n_points,
                             const int64_t *vp, const int64_t *pt, int64_t n, double *x_pred, double *A,
                             double *B) {
    TDK_API_GUARD;
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
    TDK_API_GUARD;
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
    TDK_API_GUARD;
    TDK_REQUIRE(n >= 0 && n_poses >= 1 && n_points >= 1 && n_poses <= 65535, "bad sizes");
    TDK_REQUIRE(n < (1ll << 31) && n_points < (1ll << 31), "more than 2^31 observations or points");
    TDK_REQUIRE(poses && points && U && ea && V && eb && err && (n == 0 || (x_true && vp && pt)), "null pointer");
    int sorted = 0;
    TDK_TRY(check_indices(vp, pt, n, n_poses, n_points, &sorted));
    BaPlan plan;
    ba_make_plan(vp, pt, n, n_poses, sorted != 0, &plan);
    void *d_poses, *d_points, *d_xt, *d_pt, *d_U, *d_ea, *d_V, *d_eb, *d_part, *d_err, *d_segs, *d_sptr, *d_tick;
    void *d_sorted = nullptr;
    TDK_TRY(h2d(0, poses, (size_t)n_poses * 48, &d_poses));
    TDK_TRY(h2d(1, points, (size_t)n_points * 24, &d_points));
    TDK_TRY(h2d(2, x_true, (size_t)n * 16, &d_xt));
    TDK_TRY(h2d(3, plan.pt32.data(), (size_t)n * 4, &d_pt));
    if (!plan.obs_sorted.empty()) TDK_TRY(h2d(4, plan.obs_sorted.data(), (size_t)n * 4, &d_sorted));
    TDK_TRY(tdk::scratch(5, (size_t)n_poses * 21 * 8, &d_U));
    TDK_TRY(tdk::scratch(6, (size_t)n_poses * 6 * 8, &d_ea));
    TDK_TRY(tdk::scratch(7, (size_t)n_points * 6 * 8, &d_V));
    TDK_TRY(tdk::scratch(8, (size_t)n_points * 3 * 8, &d_eb));
    TDK_TRY(tdk::scratch(9, plan.segs.size() * kPoseAccPad * 8, &d_part));
    TDK_TRY(tdk::scratch(10, (size_t)n_poses * 8, &d_err));
    TDK_TRY(h2d(11, plan.segs.data(), plan.segs.size() * sizeof(BaSeg), &d_segs));
    TDK_TRY(h2d(12, plan.seg_ptr.data(), plan.seg_ptr.size() * sizeof(int), &d_sptr));
    TDK_TRY(tdk::scratch(13, (size_t)n_poses * sizeof(int), &d_tick));
    TDK_HIP(hipMemsetAsync(d_tick, 0, (size_t)n_poses * sizeof(int), tdk::stream()));
    TDK_HIP(hipMemsetAsync(d_V, 0, (size_t)n_points * 6 * 8, tdk::stream()));
    TDK_HIP(hipMemsetAsync(d_eb, 0, (size_t)n_points * 3 * 8, tdk::stream()));
    launch_reduce_seg<MODE_ATOMIC_V>((const double *)d_poses, (const double *)d_points, (const double *)d_xt,
                                     (const int *)d_sorted, (const int *)d_pt, (const BaSeg *)d_segs,
                                     (const int *)d_sptr, (int)plan.segs.size(), (int)n_poses, n, (double *)d_V, (double *)d_eb,
                                     nullptr, nullptr, (double *)d_part, (int *)d_tick, (double *)d_U,
                                     (double *)d_ea, (double *)d_err, tdk::stream());
    TDK_LAUNCH_CHECK();
    TDK_HIP(hipMemcpyAsync(U, d_U, (size_t)n_poses * 21 * 8, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipMemcpyAsync(ea, d_ea, (size_t)n_poses * 6 * 8, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipMemcpyAsync(V, d_V, (size_t)n_points * 6 * 8, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipMemcpyAsync(eb, d_eb, (size_t)n_points * 3 * 8, hipMemcpyDeviceToHost, tdk::stream()));
    void *stage;
    TDK_TRY(tdk::pinned(3, (size_t)n_poses * 8, &stage));
    TDK_HIP(hipMemcpyAsync(stage, d_err, (size_t)n_poses * 8, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipStreamSynchronize(tdk::stream()));   // `plan` must outlive the H2D copies
    double e = 0.0;
    for (int64_t j = 0; j < n_poses; j++) e += ((const double *)stage)[j];
    *err = e;
    return TDK_OK;
}

}  // extern "C"

enum { BA_K_REDUCE = 0, BA_K_ERROR = 1, BA_K_POINT_SUMS = 2, BA_K_SCHUR = 3, BA_K_BACKSUB = 4, BA_K_SOLVE = 5,
       BA_K_COUNT = 6 };

struct BaSums { double *U, *ea, *V, *eb, *W; };

struct tdk_ba {
    int64_t n_poses, n_points, n;
    unsigned options;     // TDK_BA_OPT_* of tdk_ba_create_ex
    int sorted;
    int n_segs;
    int *d_obs_sorted, *d_pt32, *d_seg_ptr, *d_ticket;   // pose-sorted observation list (NULL: identity), segments
    BaSeg *d_segs;
    double *d_poses, *d_points, *d_xt;
    int64_t *d_vp, *d_pt, *d_row_ptr, *d_obs;
    // per-kernel timing with HIP events (tdk_ba_set_profiling)
    bool profiling;
    std::vector<hipEvent_t> ev_pool;
    std::vector<int> ev_kind;
    size_t ev_used;
    double prof_ms[BA_K_COUNT];
    int64_t prof_launches[BA_K_COUNT];
    // block sums of one reduce: per pose U (21), ea (6); per point V (6), eb (3) as structure of
    // arrays; W_ij (18) per observation.  `cur` belongs to the current parameters; `alt` (allocated
    // by the first tdk_ba_solve) receives the sums at the candidate of a damping trial, so that an
    // accepted candidate needs no second pass and a rejected one leaves `cur` intact.
    BaSums cur, alt;
    double *d_part, *d_err, *d_Vinv, *d_S, *d_e, *d_da, *d_db;
    double *d_cposes, *d_cpoints;   // candidate parameters: poses + da, points + db
    bool dev_solve;                 // the reduced camera system is solved on the device (6 P <= kDevSolveMaxDim)
    double *d_Be;       // [8][n]: B_ij (2x3) and the residual of every observation
    // Schur complement by pose pair (k_ba_schur_pairs): dense observation table
    int *d_obs_at;      // [n_poses][n_points] observation index or -1; NULL -> atomics fallback
    double *d_spart;    // [pairs][point chunks][kSchurAccPad]
    double *d_mpart;    // [mfma_blocks][tiles * 256 + 64] partials of k_ba_schur_mfma (NULL: more than 8 poses)
    double *d_rod;      // [n_poses][8] pose-only Rodrigues coefficients of the current poses (recompute_W)
    int mfma_blocks;
    int64_t pchunk, npchunks;
};

namespace {

// event pair around the launches of one kernel family, only when profiling is on
struct BaTimer {
    tdk_ba *h;
    hipEvent_t e1;
    BaTimer(tdk_ba *h_, int kind) : h(h_), e1(nullptr) {
        if (!h->profiling) return;
        while (h->ev_pool.size() < h->ev_used + 2) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return;
            h->ev_pool.push_back(e);
            h->ev_kind.push_back(0);
        }
        hipEvent_t e0 = h->ev_pool[h->ev_used];
        e1 = h->ev_pool[h->ev_used + 1];
        h->ev_kind[h->ev_used] = kind;
        h->ev_used += 2;
        (void)hipEventRecord(e0, tdk::stream());
    }
    ~BaTimer() {
        if (e1) (void)hipEventRecord(e1, tdk::stream());
    }
};

// after a stream synchronisation: fold the recorded event pairs into the per-kernel sums
void ba_collect_profile(tdk_ba *h) {
    for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, h->ev_pool[i], h->ev_pool[i + 1]) != hipSuccess) continue;
        h->prof_ms[h->ev_kind[i]] += ms;
        h->prof_launches[h->ev_kind[i]] += 1;
    }
    h->ev_used = 0;
}

// W_ij rebuilt from the parameters by the Schur / back-substitution kernels instead of stored by the reduce:
// whenever the MFMA Schur kernel applies (windows of up to 8 poses with a dense observation table; d_mpart is
// only allocated then).  The other Schur kernels read W from memory.
bool ba_recompute_w(const tdk_ba *h) { return h->d_obs_at != nullptr && h->d_mpart != nullptr; }

// what a reduce computes at parameters that are already on the device
enum { REDUCE_ERROR = 0, REDUCE_SUMS = 1, REDUCE_STEP = 2 };
//   REDUCE_ERROR: the residual sum only
//   REDUCE_SUMS : per-pose and per-point block sums U, ea, V, eb
//   REDUCE_STEP : ... and W_ij per observation for the Schur complement
// Enqueues the kernels; the sums go to `into` (h->cur or h->alt), the per-pose
// squared errors to h->d_err.  Nothing is waited for.
tdk_status ba_reduce_launch(tdk_ba *h, const double *d_poses, const double *d_points, int what, const BaSums &into) {
    {
        BaTimer t(h, what == REDUCE_ERROR ? BA_K_ERROR : BA_K_REDUCE);
#define BA_LAUNCH(MODE)                                                                                          \
    launch_reduce_seg<MODE>(d_poses, d_points, h->d_xt, h->d_obs_sorted, h->d_pt32, h->d_segs, h->d_seg_ptr,      \
                            h->n_segs, (int)h->n_poses, h->n, nullptr, nullptr, into.W, h->d_Be, h->d_part,       \
                            h->d_ticket, into.U, into.ea, h->d_err, tdk::stream())
        if (what == REDUCE_ERROR) BA_LAUNCH(MODE_ERROR);
        else if (what == REDUCE_SUMS) BA_LAUNCH(MODE_STORE_B);
        else if (ba_recompute_w(h)) BA_LAUNCH(MODE_STORE_B);   // W is rebuilt where it is needed (recompute_W)
        else BA_LAUNCH(MODE_STORE_BW);
#undef BA_LAUNCH
        TDK_LAUNCH_CHECK();
    }
    if (what != REDUCE_ERROR) {
        BaTimer t(h, BA_K_POINT_SUMS);
        k_ba_point_sums<<<grid_for(h->n_points), kBlock, 0, tdk::stream()>>>(h->d_row_ptr, h->d_obs, h->d_Be, h->n,
                                                                             h->n_points, into.V, into.eb);
        TDK_LAUNCH_CHECK();
    }
    return TDK_OK;
}

// Waits for the stream and returns what the last reduce / solve left in h->d_err:
// the sum of the per-pose squared errors and the status word of k_ba_rcs_solve.
tdk_status ba_wait(tdk_ba *h, double *err, bool *singular) {
    void *stage;
    const size_t words = (size_t)h->n_poses + 1;
    TDK_TRY(tdk::pinned(3, words * 8, &stage));
    TDK_HIP(hipMemcpyAsync(stage, h->d_err, words * 8, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipStreamSynchronize(tdk::stream()));
    if (h->profiling) ba_collect_profile(h);
    double sum = 0.0;
    for (int64_t j = 0; j < h->n_poses; j++) sum += ((const double *)stage)[j];
    *err = sum;
    if (singular) *singular = ((const double *)stage)[h->n_poses] != 0.0;
    return TDK_OK;
}

tdk_status ba_reduce_dev(tdk_ba *h, const double *d_poses, const double *d_points, int what, double *err) {
    TDK_TRY(ba_reduce_launch(h, d_poses, d_points, what, h->cur));
    return ba_wait(h, err, nullptr);
}

tdk_status ba_upload(tdk_ba *h, const double *poses, const double *points) {
    TDK_HIP(hipMemcpyAsync(h->d_poses, poses, (size_t)h->n_poses * 48, hipMemcpyHostToDevice, tdk::stream()));
    TDK_HIP(hipMemcpyAsync(h->d_points, points, (size_t)h->n_points * 24, hipMemcpyHostToDevice, tdk::stream()));
    return TDK_OK;
}

tdk_status ba_reduce(tdk_ba *h, const double *poses, const double *points, int what, double *err) {
    TDK_TRY(ba_upload(h, poses, points));
    return ba_reduce_dev(h, h->d_poses, h->d_points, what, err);
}

// ---------------------------------------------------------------------------
// Reduced camera system on the device (local BA windows, 6 P <= kDevSolveMaxDim):
// one workgroup, the augmented matrix [S | e] in LDS, Gaussian elimination with
// partial pivoting -- S is positive definite in exact arithmetic, but with small
// damping and near-degenerate points its computed Schur complement can lose
// definiteness by rounding, which LU tolerates -- and back-substitution by one
// wave.  Rows are never swapped: a row that has served as pivot is final and
// drops out (`done` masks), so within one column step every element is read
// and written by its owner thread only, the pivot row is read-only, and one
// barrier per column is enough.  Every wave finds the pivot by itself (DPP
// maximum + ballot; the lowest row among equals, as dense_solve).  S arrives as
// the Schur kernels leave it (-sum Y W^T; diagonal blocks whole, of the others
// the block triangle above the diagonal is authoritative); the kernel adds
// U*_j = U_j + mu I to the diagonal blocks and ea to the right-hand side, writes
// da, the candidate poses + da, and a status word (1.0: singular).
// ---------------------------------------------------------------------------
constexpr int kDevSolveMaxDim = 120;   // padded to [128][129] doubles = 129 KB of the 160 KB of LDS

// value of lane l (uniform) in every lane: v_readlane, no LDS round trip
__device__ __forceinline__ double lane_value(double x, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_max(double x) {
    x = fmax(x, tdk::dpp_move<0xB1>(x));    // quad_perm [1,0,3,2]
    x = fmax(x, tdk::dpp_move<0x4E>(x));    // quad_perm [2,3,0,1]
    x = fmax(x, tdk::dpp_move<0x141>(x));   // row_half_mirror
    x = fmax(x, tdk::dpp_move<0x140>(x));   // row_mirror: every lane holds its row's maximum
    return fmax(fmax(lane_value(x, 0), lane_value(x, 16)), fmax(lane_value(x, 32), lane_value(x, 48)));
}

// 1 / a to the last bit or so: v_rcp_f64 and two Newton steps -- a third of the dependent
// operations of the IEEE division, and it sits on the critical path of every column
__device__ __forceinline__ double fast_rcp(double a) {
    double x = __builtin_amdgcn_rcp(a);
    x = fma(fma(-a, x, 1.0), x, x);
    return fma(fma(-a, x, 1.0), x, x);
}

// what the back-substitution needs, kept in registers by every wave: lane l knows the pivot row and
// the reciprocal pivot of columns l and l + 64, and the column its rows l, l + 64 were pivots of
struct RcsPivots {
    int prow0, prow1, ord0, ord1;
    double rdiag0, rdiag1;
};

// [S | e] into the padded LDS matrix M[R][P] (padding zero), the diagonal of S into diag0
template <int NB>
__device__ __forceinline__ void rcs_load(double *M, double *diag0, const double *__restrict__ S,
                                         const double *__restrict__ evec, const double *__restrict__ U,
                                         const double *__restrict__ ea, double mu, int dim) {
    constexpr int R = 16 * NB, P = R + 1;
    const int t = threadIdx.x;
    for (int idx = t; idx < R * P; idx += kBlock) M[idx] = 0.0;
    __syncthreads();
    // S first (independent loads, several in flight), then U* on the diagonal blocks
#pragma unroll 4
    for (int idx = t; idx < dim * dim; idx += kBlock) {
        const int r = idx / dim, c = idx - r * dim;
        M[r * P + c] = c / 6 >= r / 6 ? S[(size_t)r * dim + c] : S[(size_t)c * dim + r];
    }
    __syncthreads();
    for (int idx = t; idx < dim * 6; idx += kBlock) {
        const int r = idx / 6, b = idx - 6 * r, j = r / 6, a = r - 6 * j;
        const int lo = a < b ? a : b, hi = a < b ? b : a;
        const double v = M[r * P + 6 * j + b] + U[21 * j + lo * 6 - lo * (lo - 1) / 2 + (hi - lo)] + (a == b ? mu : 0.0);
        M[r * P + 6 * j + b] = v;
        if (a == b) diag0[r] = v;
    }
    for (int r = t; r < dim; r += kBlock) M[r * P + dim] = evec[r] + ea[r];
    __syncthreads();
}

// Elimination without pivoting, for the positive definite system this is in exact arithmetic
// (stable without pivoting then; the same numbers as an L D L^T factorisation): pivot row and
// pivot column are known in advance, so a column step is one LDS round trip, a reciprocal and one
// barrier.  The columns of one 16-block CB at a time: the row and column blocks that still take
// part (>= CB) are then compile-time, and the loop body has no branches.  Returns false (in every
// thread) at the first pivot that is not safely positive relative to the original diagonal; the
// matrix is then reloaded and eliminated with pivoting.
template <int NB, int CB>
__device__ __forceinline__ bool rcs_spd_columns(double *M, const double *diag0, int dim, RcsPivots &pv) {
    constexpr int R = 16 * NB, P = R + 1, NL = NB - CB;          // NL live blocks each way
    const int t = threadIdx.x, lane = t & 63, ty = t >> 4, tx = t & 15;
    double *Mown = M + (16 * CB + ty) * P + 16 * CB + tx;         // the thread's element of block (CB, CB)
    const int c_end = 16 * CB + 16 < dim ? 16 * CB + 16 : dim;
    // the thread's elements stay in registers for the 16 columns of the block; per column only the
    // next pivot row and pivot column go back to LDS, everything else when the block is done
    double v[NL][NL];
#pragma unroll
    for (int i = 0; i < NL; i++)
#pragma unroll
        for (int j = 0; j < NL; j++) v[i][j] = Mown[16 * i * P + 16 * j];
    for (int c = 16 * CB; c < c_end; c++) {
        const double piv = M[c * P + c], d0 = diag0[c];
        const double *Mc = M + c * P + 16 * CB + tx;
        const double *Mf = M + (16 * CB + ty) * P + c;
        double f[NL], pr[NL];
#pragma unroll
        for (int i = 0; i < NL; i++) f[i] = Mf[16 * i * P];
#pragma unroll
        for (int j = 0; j < NL; j++) pr[j] = Mc[16 * j];
        if (!(piv > 1e-12 * fabs(d0))) return false;               // the same value in every thread
        const double rpiv = fast_rcp(piv);
        if (CB < 4) { if (lane == c) pv.rdiag0 = rpiv; }
        else if (lane == c - 64) pv.rdiag1 = rpiv;
        const bool below = 16 * CB + ty > c, right = 16 * CB + tx > c;   // only block CB straddles the pivot
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const double fi = f[i] * rpiv;
#pragma unroll
            for (int j = 0; j < NL; j++) {
                const double u = v[i][j] - fi * pr[j];
                v[i][j] = (i > 0 || below) && (j > 0 || right) ? u : v[i][j];
            }
        }
        if (16 * CB + ty == c + 1) {                 // the next pivot row, right of its diagonal and the diagonal itself
#pragma unroll
            for (int j = 0; j < NL; j++)
                if (j > 0 || 16 * CB + tx > c) Mown[16 * j] = v[0][j];
        }
        if (16 * CB + tx == c + 1) {                 // the next pivot column, below its diagonal
#pragma unroll
            for (int i = 0; i < NL; i++)
                if (i > 0 || 16 * CB + ty > c + 1) Mown[16 * i * P] = v[i][0];
        }
        __syncthreads();
    }
    const int c = c_end - 1;
    const bool below = 16 * CB + ty > c, right = 16 * CB + tx > c;
#pragma unroll
    for (int i = 0; i < NL; i++)
#pragma unroll
        for (int j = 0; j < NL; j++)
            if ((i > 0 || below) && (j > 0 || right)) Mown[16 * i * P + 16 * j] = v[i][j];
    __syncthreads();
    return true;
}

template <int NB>
__device__ __forceinline__ bool rcs_eliminate_spd(double *M, const double *diag0, int dim, RcsPivots &pv) {
    const int lane = threadIdx.x & 63;
    bool ok = rcs_spd_columns<NB, 0>(M, diag0, dim, pv);
    if constexpr (NB > 1) ok = ok && rcs_spd_columns<NB, 1>(M, diag0, dim, pv);
    if constexpr (NB > 2) ok = ok && rcs_spd_columns<NB, 2>(M, diag0, dim, pv);
    if constexpr (NB > 3) ok = ok && rcs_spd_columns<NB, 3>(M, diag0, dim, pv);
    if constexpr (NB > 4) ok = ok && rcs_spd_columns<NB, 4>(M, diag0, dim, pv);
    if constexpr (NB > 5) ok = ok && rcs_spd_columns<NB, 5>(M, diag0, dim, pv);
    if constexpr (NB > 6) ok = ok && rcs_spd_columns<NB, 6>(M, diag0, dim, pv);
    if constexpr (NB > 7) ok = ok && rcs_spd_columns<NB, 7>(M, diag0, dim, pv);
    pv.prow0 = lane; pv.prow1 = lane + 64;
    pv.ord0 = lane < dim ? lane : -1;
    pv.ord1 = lane + 64 < dim ? lane + 64 : -1;
    return ok;
}

// Gaussian elimination with partial pivoting (largest magnitude, lowest row among equals).  Rows
// are never swapped: a row that has served as pivot is final and drops out (`done` masks), so
// within one column step every element is read and written by its owner thread only, the pivot
// row is read-only, and one barrier per column is enough.  Every wave finds the pivot by itself
// (DPP maximum + ballot).  Column blocks left of block CB are finished; every row block may
// still be live.  Returns false (in every thread) when a column has no non-zero pivot.
template <int NB, int CB>
__device__ __forceinline__ bool rcs_pivoted_columns(double *M, int dim, RcsPivots &pv, unsigned long long &done0,
                                                    unsigned long long &done1) {
    constexpr int R = 16 * NB, P = R + 1, NL = NB - CB;
    const int t = threadIdx.x, lane = t & 63, ty = t >> 4, tx = t & 15;
    double *Mown = M + ty * P + 16 * CB + tx;        // the thread's element of block (0, CB)
    const int c_end = 16 * CB + 16 < dim ? 16 * CB + 16 : dim;
    for (int c = 16 * CB; c < c_end; c++) {
        // loads that do not depend on the pivot go out first: the column, the thread's own elements
        const bool live0 = lane < dim && !((done0 >> lane) & 1);
        const double a0 = lane < R ? M[lane * P + c] : 0.0;
        bool live1 = false;
        double a1 = 0.0;
        if (NB > 4) {
            live1 = lane + 64 < dim && !((done1 >> lane) & 1);
            a1 = lane + 64 < R ? M[(lane + 64) * P + c] : 0.0;
        }
        double f[NB], v[NB][NL];
#pragma unroll
        for (int i = 0; i < NB; i++) f[i] = M[(ty + 16 * i) * P + c];
#pragma unroll
        for (int i = 0; i < NB; i++)
#pragma unroll
            for (int j = 0; j < NL; j++) v[i][j] = Mown[16 * i * P + 16 * j];
        // every lane inverts its own candidate while the maximum is being found
        const double v0 = live0 ? fabs(a0) : -1.0, v1 = live1 ? fabs(a1) : -1.0;
        const double rc0 = 1.0 / a0, rc1 = NB > 4 ? 1.0 / a1 : 0.0;
        const double best = wave_max(NB > 4 ? fmax(v0, v1) : v0);
        if (!(best > 0.0)) return false;             // the same in every wave
        int p;
        double rpiv;
        const unsigned long long m0 = __ballot(v0 == best);
        if (NB <= 4 || m0) {
            const int q = __ffsll(m0) - 1;
            done0 |= 1ull << q;
            if (lane == q) pv.ord0 = c;
            rpiv = lane_value(rc0, q);
            p = q;
        } else {
            const int q = __ffsll((unsigned long long)__ballot(v1 == best)) - 1;
            done1 |= 1ull << q;
            if (lane == q) pv.ord1 = c;
            rpiv = lane_value(rc1, q);
            p = q + 64;
        }
        if (CB < 4) { if (lane == c) { pv.prow0 = p; pv.rdiag0 = rpiv; } }
        else if (lane == c - 64) { pv.prow1 = p; pv.rdiag1 = rpiv; }
        const double *Mp = M + p * P + 16 * CB + tx;
        const bool right = 16 * CB + tx > c;         // left of and in column c nothing is touched (others read column c)
#pragma unroll
        for (int i = 0; i < NB; i++) {
            const int r = ty + 16 * i;
            const bool live = !(((r < 64 ? done0 : done1) >> (r & 63)) & 1);   // finished rows (the pivot included) stay
            const double fi = f[i] * rpiv;
#pragma unroll
            for (int j = 0; j < NL; j++)
                if (live && (j > 0 || right)) Mown[16 * i * P + 16 * j] = v[i][j] - fi * Mp[16 * j];
        }
        __syncthreads();
    }
    return true;
}

template <int NB>
__device__ __forceinline__ bool rcs_eliminate_pivoted(double *M, int dim, RcsPivots &pv) {
    unsigned long long done0 = 0, done1 = 0;         // rows 0..63 / 64..127 that have been pivots (the same in every wave)
    pv.ord0 = pv.ord1 = -1;
    bool ok = rcs_pivoted_columns<NB, 0>(M, dim, pv, done0, done1);
    if constexpr (NB > 1) ok = ok && rcs_pivoted_columns<NB, 1>(M, dim, pv, done0, done1);
    if constexpr (NB > 2) ok = ok && rcs_pivoted_columns<NB, 2>(M, dim, pv, done0, done1);
    if constexpr (NB > 3) ok = ok && rcs_pivoted_columns<NB, 3>(M, dim, pv, done0, done1);
    if constexpr (NB > 4) ok = ok && rcs_pivoted_columns<NB, 4>(M, dim, pv, done0, done1);
    if constexpr (NB > 5) ok = ok && rcs_pivoted_columns<NB, 5>(M, dim, pv, done0, done1);
    if constexpr (NB > 6) ok = ok && rcs_pivoted_columns<NB, 6>(M, dim, pv, done0, done1);
    if constexpr (NB > 7) ok = ok && rcs_pivoted_columns<NB, 7>(M, dim, pv, done0, done1);
    return ok;
}

template <int NB>   // 16 NB >= dim + 1: the matrix is padded to [16 NB][16 NB + 1], 16 x 16 threads own NB x NB elements each
__global__ __launch_bounds__(kBlock) void k_ba_rcs_solve(const double *__restrict__ S, const double *__restrict__ evec,
                                                         const double *__restrict__ U, const double *__restrict__ ea,
                                                         double mu, int dim, int pivoted_only,
                                                         const double *__restrict__ poses, double *__restrict__ da,
                                                         double *__restrict__ cposes, double *__restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int R = 16 * NB, P = R + 1;            // odd pitch: a column walk touches every bank pair
    double *M = reinterpret_cast<double *>(smem);    // [R][P]; column `dim` is the right-hand side, the padding is zero
    double *diag0 = M + R * P;                       // [R] diagonal of S as loaded
    const int t = threadIdx.x, lane = t & 63;
    rcs_load<NB>(M, diag0, S, evec, U, ea, mu, dim);
    RcsPivots pv = {0, 0, -1, -1, 0.0, 0.0};
    double state = 0.0;                              // 0: positive definite path, 2: pivoted, 1: singular
    if (pivoted_only || !rcs_eliminate_spd<NB>(M, diag0, dim, pv)) {
        __syncthreads();
        rcs_load<NB>(M, diag0, S, evec, U, ea, mu, dim);
        state = rcs_eliminate_pivoted<NB>(M, dim, pv) ? 2.0 : 1.0;
    }
    if (t == 0) status[0] = state == 1.0 ? 1.0 : 0.0;
    if (t >= 64 || state == 1.0) return;
    // back-substitution, column oriented, in one wave: lane l keeps the right-hand sides of rows l and
    // l + 64; the matrix entries of four steps are fetched ahead of the dependent chain
    double b0 = lane < dim ? M[lane * P + dim] : 0.0;
    double b1 = NB > 4 && lane + 64 < dim ? M[(lane + 64) * P + dim] : 0.0;
    double x0 = 0.0, x1 = 0.0;
    const double *Ml0 = M + (lane < R ? lane : 0) * P, *Ml1 = M + (NB > 4 && lane + 64 < R ? lane + 64 : 0) * P;
    auto step = [&](int i, double m0, double m1) {
        int p;
        double rd, bi;
        if (NB > 4 && i >= 64) {
            p = __builtin_amdgcn_readlane(pv.prow1, i - 64);
            rd = lane_value(pv.rdiag1, i - 64);
        } else {
            p = __builtin_amdgcn_readlane(pv.prow0, i);
            rd = lane_value(pv.rdiag0, i);
        }
        if (NB > 4 && p >= 64) bi = lane_value(b1, p - 64);
        else bi = lane_value(b0, p);
        const double xi = bi * rd;
        if (NB > 4 && i >= 64) x1 = lane == i - 64 ? xi : x1;
        else x0 = lane == i ? xi : x0;
        b0 = pv.ord0 < i ? b0 - m0 * xi : b0;        // rows that are not pivots (ord -1, padding) carry zeros
        if (NB > 4) b1 = pv.ord1 < i ? b1 - m1 * xi : b1;
    };
    int i = dim - 1;
    for (; i >= 3; i -= 4) {
        double m0[4], m1[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            m0[q] = Ml0[i - q];
            m1[q] = NB > 4 ? Ml1[i - q] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) step(i - q, m0[q], m1[q]);
    }
    for (; i >= 0; i--) step(i, Ml0[i], NB > 4 ? Ml1[i] : 0.0);
    if (lane < dim) { da[lane] = x0; cposes[lane] = poses[lane] + x0; }
    if (NB > 4 && lane + 64 < dim) { da[lane + 64] = x1; cposes[lane + 64] = poses[lane + 64] + x1; }
}

typedef void (*rcs_solve_fn)(const double *, const double *, const double *, const double *, double, int, int,
                             const double *, double *, double *, double *);
rcs_solve_fn rcs_solve_kernel(int dim) {
    switch ((dim + 1 + 15) / 16) {
        case 1: return k_ba_rcs_solve<1>;
        case 2: return k_ba_rcs_solve<2>;
        case 3: return k_ba_rcs_solve<3>;
        case 4: return k_ba_rcs_solve<4>;
        case 5: return k_ba_rcs_solve<5>;
        case 6: return k_ba_rcs_solve<6>;
        case 7: return k_ba_rcs_solve<7>;
        default: return k_ba_rcs_solve<8>;
    }
}

size_t rcs_solve_lds(int dim) { const size_t R = 16 * (size_t)((dim + 16) / 16); return R * (R + 2) * 8; }

// out = a + b (candidate poses when the reduced system was solved on the host)
__global__ void k_ba_add(const double *__restrict__ a, const double *__restrict__ b, int64_t n,
                         double *__restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = a[i] + b[i];
}

// Levenberg-Marquardt update for damping mu from the sums in h->cur (the last
// ba_reduce_launch(..., REDUCE_STEP, h->cur)) at the parameters h->d_poses /
// h->d_points: V*^-1, Schur complement, dense solve of the reduced camera system
// (on the device for local windows, else on the host), back-substitution.
// Leaves da in h->d_da, db in h->d_db and the candidate parameters in
// h->d_cposes / h->d_cpoints; with the device solve nothing is waited for and
// the status word behind h->d_err says whether the system was singular.
tdk_status ba_update_launch(tdk_ba *h, double mu) {
    const int dim = (int)(6 * h->n_poses);
    const int gp = grid_for(h->n_points);
    const bool mfma = h->d_obs_at != nullptr && h->d_mpart != nullptr;
    const bool recomp = ba_recompute_w(h);
    k_ba_invert_V<<<gp, kBlock, 0, tdk::stream()>>>(h->cur.V, mu, h->n_points, h->d_Vinv, h->d_poses, (int)h->n_poses,
                                                    recomp ? h->d_rod : nullptr);
    TDK_LAUNCH_CHECK();
    if (!mfma) {   // the other kernels accumulate into S and e; the MFMA finish writes every entry
        TDK_HIP(hipMemsetAsync(h->d_S, 0, (size_t)dim * dim * 8, tdk::stream()));
        TDK_HIP(hipMemsetAsync(h->d_e, 0, (size_t)dim * 8, tdk::stream()));
    }
    {
    BaTimer schur_timer(h, BA_K_SCHUR);   // closes (second event) at the end of this scope, on every path
    if (mfma) {
        // dense contraction on the FP64 matrix cores (windows of up to 8 poses)
        const int RB = (int)((6 * h->n_poses + 15) / 16);
        const int n_chunks = (int)((h->n_points + kSchurPC - 1) / kSchurPC);
        const int nb = n_chunks < h->mfma_blocks ? n_chunks : h->mfma_blocks;
#define BA_SCHUR_MFMA(RBV)                                                                                         \
    do {                                                                                                           \
        if (recomp)                                                                                               \
            k_ba_schur_mfma<RBV, true><<<nb, kBlock, 0, tdk::stream()>>>(h->d_obs_at, h->cur.W, h->d_Vinv, h->cur.eb, \
                h->n, h->n_points, (int)h->n_poses, n_chunks, h->d_mpart, h->d_poses, h->d_rod, h->d_points);      \
        else                                                                                                      \
            k_ba_schur_mfma<RBV, false><<<nb, kBlock, 0, tdk::stream()>>>(h->d_obs_at, h->cur.W, h->d_Vinv, h->cur.eb, \
                h->n, h->n_points, (int)h->n_poses, n_chunks, h->d_mpart, nullptr, nullptr, nullptr);              \
        constexpr int per = RBV * (RBV + 1) / 2 * 256 + 64;                                                        \
        k_ba_schur_mfma_finish<RBV><<<(per + 31) / 32, kBlock, 0, tdk::stream()>>>(h->d_mpart, nb, dim,           \
                                                                                              h->d_S, h->d_e);    \
    } while (0)
        if (RB == 1) BA_SCHUR_MFMA(1);
        else if (RB == 2) BA_SCHUR_MFMA(2);
        else BA_SCHUR_MFMA(3);
#undef BA_SCHUR_MFMA
    } else if (h->d_obs_at != nullptr) {
        const int pairs = (int)(h->n_poses * (h->n_poses + 1) / 2);
        dim3 grid((unsigned)h->npchunks, (unsigned)pairs);
        k_ba_schur_pairs<<<grid, kBlock, 0, tdk::stream()>>>(h->d_obs_at, h->cur.W, h->d_Vinv, h->cur.eb, h->n,
                                                             h->n_points, (int)h->n_poses, h->pchunk, h->d_spart);
        TDK_LAUNCH_CHECK();
        k_ba_schur_finish<<<pairs, 64, 0, tdk::stream()>>>(h->d_spart, (int)h->npchunks, (int)h->n_poses, dim, h->d_S,
                                                           h->d_e);
    } else {
        int gs = gp > 1024 ? 1024 : gp;
        if (h->n_poses <= kMaxLdsPoses) {
            size_t lds = ((size_t)dim * dim + dim) * 8;
            k_ba_schur<true><<<gs, kBlock, lds, tdk::stream()>>>(h->d_row_ptr, h->d_obs, h->d_vp, h->cur.W, h->d_Vinv,
                                                                 h->cur.eb, h->n, h->n_points, dim, h->d_S, h->d_e);
        } else {
            k_ba_schur<false><<<gs, kBlock, 0, tdk::stream()>>>(h->d_row_ptr, h->d_obs, h->d_vp, h->cur.W, h->d_Vinv,
                                                                h->cur.eb, h->n, h->n_points, dim, h->d_S, h->d_e);
        }
    }
    TDK_LAUNCH_CHECK();
    }
    if (h->dev_solve) {
        BaTimer t(h, BA_K_SOLVE);
        const int pivoted_only = (h->options & TDK_BA_OPT_SOLVE_PIVOTED) ? 1 : 0;   // skip the attempt without pivoting
        rcs_solve_kernel(dim)<<<1, kBlock, rcs_solve_lds(dim), tdk::stream()>>>(
            h->d_S, h->d_e, h->cur.U, h->cur.ea, mu, dim, pivoted_only, h->d_poses, h->d_da, h->d_cposes,
            h->d_err + h->n_poses);
        TDK_LAUNCH_CHECK();
    } else {
        // larger windows: the reduced camera system goes through the host
        std::vector<double> S((size_t)dim * dim), e((size_t)dim), U((size_t)h->n_poses * 21), ea((size_t)dim);
        TDK_HIP(hipMemcpyAsync(S.data(), h->d_S, S.size() * 8, hipMemcpyDeviceToHost, tdk::stream()));
        TDK_HIP(hipMemcpyAsync(e.data(), h->d_e, e.size() * 8, hipMemcpyDeviceToHost, tdk::stream()));
        TDK_HIP(hipMemcpyAsync(U.data(), h->cur.U, U.size() * 8, hipMemcpyDeviceToHost, tdk::stream()));
        TDK_HIP(hipMemcpyAsync(ea.data(), h->cur.ea, ea.size() * 8, hipMemcpyDeviceToHost, tdk::stream()));
        TDK_HIP(hipStreamSynchronize(tdk::stream()));
        // S = blockdiag(U + mu I) - sum Y W^T (upper block triangle from the device), mirrored
        for (int64_t j = 0; j < h->n_poses; j++) {
            int m = 0;
            for (int a = 0; a < 6; a++)
                for (int b = a; b < 6; b++) {
                    double u = U[(size_t)j * 21 + m++] + (a == b ? mu : 0.0);
                    S[(size_t)(6 * j + a) * dim + 6 * j + b] += u;
                    if (a != b) S[(size_t)(6 * j + b) * dim + 6 * j + a] += u;
                }
        }
        for (int r = 0; r < dim; r++)
            for (int c = 0; c < dim; c++)
                if (c / 6 > r / 6) S[(size_t)c * dim + r] = S[(size_t)r * dim + c];
        for (int i = 0; i < dim; i++) e[(size_t)i] += ea[(size_t)i];
        if (dense_solve(S, e, dim) != 0) {
            tdk::set_error("reduced camera system is singular (mu = %g)", mu);
            return TDK_ERR_SINGULAR;
        }
        TDK_HIP(hipMemcpyAsync(h->d_da, e.data(), (size_t)dim * 8, hipMemcpyHostToDevice, tdk::stream()));
        k_ba_add<<<grid_for(dim), kBlock, 0, tdk::stream()>>>(h->d_poses, h->d_da, dim, h->d_cposes);
        TDK_LAUNCH_CHECK();
        TDK_HIP(hipStreamSynchronize(tdk::stream()));   // `e` goes out of scope
    }
    {
        BaTimer t(h, BA_K_BACKSUB);
        if (recomp)
            k_ba_backsub<true><<<gp, kBlock, 0, tdk::stream()>>>(h->d_row_ptr, h->d_obs, h->d_vp, h->cur.W, h->d_Vinv,
                                                                 h->cur.eb, h->d_da, h->n, h->n_points, h->d_points,
                                                                 h->d_db, h->d_cpoints, h->d_poses, h->d_rod);
        else
            k_ba_backsub<false><<<gp, kBlock, 0, tdk::stream()>>>(h->d_row_ptr, h->d_obs, h->d_vp, h->cur.W, h->d_Vinv,
                                                                  h->cur.eb, h->d_da, h->n, h->n_points, h->d_points,
                                                                  h->d_db, h->d_cpoints, nullptr, nullptr);
    }
    TDK_LAUNCH_CHECK();
    return TDK_OK;
}

// second set of block sums for the damping trials of tdk_ba_solve
tdk_status ba_ensure_alt(tdk_ba *h) {
    if (h->alt.U) return TDK_OK;
    TDK_HIP(hipMalloc(&h->alt.U, (size_t)h->n_poses * 21 * 8));
    TDK_HIP(hipMalloc(&h->alt.ea, (size_t)h->n_poses * 6 * 8));
    TDK_HIP(hipMalloc(&h->alt.V, (size_t)h->n_points * 48));
    TDK_HIP(hipMalloc(&h->alt.eb, (size_t)h->n_points * 24));
    if (!ba_recompute_w(h)) TDK_HIP(hipMalloc(&h->alt.W, (size_t)h->n * 18 * 8));
    return TDK_OK;
}

}  // namespace

static tdk_status ba_allocate(tdk_ba *h, int64_t n_poses, int64_t n_points, const int64_t *vp, const int64_t *pt,
                              const double *x_true, int64_t n, int sorted);

extern "C" {

tdk_status tdk_ba_create(int64_t n_poses, int64_t n_points, const int64_t *vp, const int64_t *pt,
                         const double *x_true, int64_t n, tdk_ba **out) {
    TDK_API_GUARD;
    return tdk_ba_create_ex(n_poses, n_points, vp, pt, x_true, n, 0u, out);
}

tdk_status tdk_ba_create_ex(int64_t n_poses, int64_t n_points, const int64_t *vp, const int64_t *pt,
                            const double *x_true, int64_t n, unsigned int options, tdk_ba **out) {
    TDK_API_GUARD;
    TDK_REQUIRE(out && vp && pt && x_true, "null pointer");
    TDK_REQUIRE(options < 16u, "unknown option bits");
    TDK_REQUIRE(n >= 1 && n_poses >= 1 && n_points >= 1 && n_poses <= 2048, "bad sizes");
    TDK_REQUIRE(n < (1ll << 31) && n_points < (1ll << 31), "more than 2^31 observations or points");
    int sorted = 0;
    TDK_TRY(check_indices(vp, pt, n, n_poses, n_points, &sorted));
    TDK_TRY(tdk::ensure_device());
    tdk_ba *h = new tdk_ba();   // value-initialised: every pointer is null until allocated
    h->options = options;
    const tdk_status st = ba_allocate(h, n_poses, n_points, vp, pt, x_true, n, sorted);
    if (st != TDK_OK) {
        tdk_ba_destroy(h);      // nothing leaks when an allocation in the middle fails
        return st;
    }
    *out = h;
    return TDK_OK;
}

static tdk_status ba_allocate(tdk_ba *h, int64_t n_poses, int64_t n_points, const int64_t *vp, const int64_t *pt,
                              const double *x_true, int64_t n, int sorted) {
    h->n_poses = n_poses; h->n_points = n_points; h->n = n; h->sorted = sorted;
    h->profiling = false; h->ev_used = 0;
    for (int k = 0; k < BA_K_COUNT; k++) { h->prof_ms[k] = 0.0; h->prof_launches[k] = 0; }
    BaPlan plan;
    ba_make_plan(vp, pt, n, n_poses, sorted != 0, &plan);
    h->n_segs = (int)plan.segs.size();
    // observation lists per point (CSR), in increasing observation index
    std::vector<int64_t> row_ptr((size_t)n_points + 1, 0), obs((size_t)n);
    for (int64_t k = 0; k < n; k++) row_ptr[(size_t)pt[k] + 1]++;
    for (int64_t i = 0; i < n_points; i++) row_ptr[(size_t)i + 1] += row_ptr[(size_t)i];
    {
        std::vector<int64_t> cursor(row_ptr.begin(), row_ptr.end() - 1);
        for (int64_t k = 0; k < n; k++) obs[(size_t)cursor[(size_t)pt[k]]++] = k;
    }
    const int dim = (int)(6 * n_poses);
    TDK_HIP(hipMalloc(&h->d_poses, (size_t)n_poses * 48));
    TDK_HIP(hipMalloc(&h->d_points, (size_t)n_points * 24));
    TDK_HIP(hipMalloc(&h->d_xt, (size_t)n * 16));
    TDK_HIP(hipMalloc(&h->d_vp, (size_t)n * 8));
    TDK_HIP(hipMalloc(&h->d_pt, (size_t)n * 8));
    TDK_HIP(hipMalloc(&h->d_row_ptr, ((size_t)n_points + 1) * 8));
    TDK_HIP(hipMalloc(&h->d_obs, (size_t)n * 8));
    TDK_HIP(hipMalloc(&h->cur.U, (size_t)n_poses * 21 * 8));
    TDK_HIP(hipMalloc(&h->cur.ea, (size_t)n_poses * 6 * 8));
    TDK_HIP(hipMalloc(&h->cur.V, (size_t)n_points * 48));
    TDK_HIP(hipMalloc(&h->cur.eb, (size_t)n_points * 24));
    TDK_HIP(hipMalloc(&h->d_part, (size_t)h->n_segs * kPoseAccPad * 8));
    TDK_HIP(hipMalloc(&h->d_pt32, (size_t)n * sizeof(int)));
    TDK_HIP(hipMalloc(&h->d_segs, plan.segs.size() * sizeof(BaSeg)));
    TDK_HIP(hipMalloc(&h->d_seg_ptr, plan.seg_ptr.size() * sizeof(int)));
    TDK_HIP(hipMalloc(&h->d_ticket, (size_t)n_poses * sizeof(int)));
    h->d_obs_sorted = nullptr;
    if (!plan.obs_sorted.empty()) {
        TDK_HIP(hipMalloc(&h->d_obs_sorted, (size_t)n * sizeof(int)));
        TDK_HIP(hipMemcpy(h->d_obs_sorted, plan.obs_sorted.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice));
    }
    TDK_HIP(hipMemcpy(h->d_pt32, plan.pt32.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice));
    TDK_HIP(hipMemcpy(h->d_segs, plan.segs.data(), plan.segs.size() * sizeof(BaSeg), hipMemcpyHostToDevice));
    TDK_HIP(hipMemcpy(h->d_seg_ptr, plan.seg_ptr.data(), plan.seg_ptr.size() * sizeof(int), hipMemcpyHostToDevice));
    TDK_HIP(hipMemset(h->d_ticket, 0, (size_t)n_poses * sizeof(int)));
    TDK_HIP(hipMalloc(&h->d_err, ((size_t)n_poses + 1) * 8));   // per-pose squared errors | status word of k_ba_rcs_solve
    TDK_HIP(hipMemset(h->d_err, 0, ((size_t)n_poses + 1) * 8));
    h->dev_solve = false;
    if (dim <= kDevSolveMaxDim) {
        const size_t lds = rcs_solve_lds(dim);
        const bool host_solve = (h->options & TDK_BA_OPT_SOLVE_HOST) != 0;
        h->dev_solve = !host_solve &&
                       (lds <= 64 * 1024 ||
                        hipFuncSetAttribute(reinterpret_cast<const void *>(rcs_solve_kernel(dim)),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess);
    }
    TDK_HIP(hipMalloc(&h->d_Vinv, (size_t)n_points * 48));
    TDK_HIP(hipMalloc(&h->d_S, (size_t)dim * dim * 8));
    TDK_HIP(hipMalloc(&h->d_e, (size_t)dim * 8));
    TDK_HIP(hipMalloc(&h->d_da, (size_t)dim * 8));
    TDK_HIP(hipMalloc(&h->d_db, (size_t)n_points * 24));
    TDK_HIP(hipMalloc(&h->d_Be, (size_t)n * 8 * 8));
    TDK_HIP(hipMalloc(&h->d_cposes, (size_t)n_poses * 48));
    TDK_HIP(hipMalloc(&h->d_cpoints, (size_t)n_points * 24));
    // dense (pose, point) -> observation table for the pair-wise Schur kernel:
    // only when it is small (local BA windows) and no observation is repeated
    h->d_obs_at = nullptr; h->d_spart = nullptr; h->d_mpart = nullptr; h->mfma_blocks = 0;
    TDK_HIP(hipMalloc(&h->d_rod, (size_t)n_poses * 8 * 8));
    h->pchunk = 2048;
    h->npchunks = (n_points + h->pchunk - 1) / h->pchunk;
    if (!(h->options & TDK_BA_OPT_SCHUR_GENERAL) && n_poses * n_points <= (1ll << 26) && n < (1ll << 31) &&
        n_poses <= 256) {
        std::vector<int> table((size_t)(n_poses * n_points), -1);
        bool unique = true;
        for (int64_t k = 0; k < n && unique; k++) {
            int &slot = table[(size_t)(vp[k] * n_points + pt[k])];
            if (slot >= 0) unique = false;
            slot = (int)k;
        }
        if (unique) {
            const int64_t pairs = n_poses * (n_poses + 1) / 2;
            TDK_HIP(hipMalloc(&h->d_obs_at, table.size() * sizeof(int)));
            TDK_HIP(hipMalloc(&h->d_spart, (size_t)(pairs * h->npchunks) * kSchurAccPad * 8));
            if (n_poses <= 8 && !(h->options & TDK_BA_OPT_SCHUR_PAIRS)) {   // 6 P <= 48 rows = three 16-row blocks of the FP64 MFMA tile
                static_assert(8 <= kBlock, "k_ba_invert_V fills rod8 from the first n_poses threads of block 0");
                h->mfma_blocks = 768;   // 3 resident blocks per CU (50 KB of LDS each), ~2 chunks of 32 points per block at 50 000 points
                TDK_HIP(hipMalloc(&h->d_mpart, (size_t)h->mfma_blocks * (6 * 256 + 64) * 8));
            }
            TDK_HIP(hipMemcpy(h->d_obs_at, table.data(), table.size() * sizeof(int), hipMemcpyHostToDevice));
        }
    }
    // W_ij per observation (18 doubles each: 58 MB for the 8 x 50 000 window) only where a kernel reads it
    if (!ba_recompute_w(h)) TDK_HIP(hipMalloc(&h->cur.W, (size_t)n * 18 * 8));
    TDK_HIP(hipMemcpyAsync(h->d_xt, x_true, (size_t)n * 16, hipMemcpyHostToDevice, tdk::stream()));
    TDK_HIP(hipMemcpyAsync(h->d_vp, vp, (size_t)n * 8, hipMemcpyHostToDevice, tdk::stream()));
    TDK_HIP(hipMemcpyAsync(h->d_pt, pt, (size_t)n * 8, hipMemcpyHostToDevice, tdk::stream()));
    TDK_HIP(hipMemcpyAsync(h->d_row_ptr, row_ptr.data(), row_ptr.size() * 8, hipMemcpyHostToDevice, tdk::stream()));
    TDK_HIP(hipMemcpyAsync(h->d_obs, obs.data(), obs.size() * 8, hipMemcpyHostToDevice, tdk::stream()));
    TDK_HIP(hipStreamSynchronize(tdk::stream()));   // the host vectors go out of scope
    return TDK_OK;
}

tdk_status tdk_ba_destroy(tdk_ba *h) {
    TDK_API_GUARD;
    if (!h) return TDK_OK;
    (void)hipStreamSynchronize(tdk::stream());
    void *ptrs[] = {h->d_poses, h->d_points, h->d_xt, h->d_vp, h->d_pt, h->d_row_ptr, h->d_obs, h->cur.U, h->cur.ea,
                    h->cur.V, h->cur.eb, h->d_part, h->d_err, h->cur.W, h->d_Vinv, h->d_S, h->d_e, h->d_da, h->d_db,
                    h->d_Be, h->d_obs_at, h->d_spart, h->d_mpart, h->d_cposes, h->d_cpoints, h->d_obs_sorted, h->d_pt32,
                    h->d_segs, h->d_seg_ptr, h->d_ticket, h->alt.U, h->alt.ea, h->alt.V, h->alt.eb, h->alt.W, h->d_rod};
    for (void *p : ptrs) (void)hipFree(p);
    for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
    delete h;
    return TDK_OK;
}

tdk_status tdk_ba_error(tdk_ba *h, const double *poses, const double *points, double *sum_sq) {
    TDK_API_GUARD;
    TDK_REQUIRE(h && poses && points && sum_sq, "null pointer");
    return ba_reduce(h, poses, points, REDUCE_ERROR, sum_sq);
}

tdk_status tdk_ba_block_sums(tdk_ba *h, const double *poses, const double *points, double *U, double *ea,
                             double *V, double *eb, double *sum_sq) {
    TDK_API_GUARD;
    TDK_REQUIRE(h && poses && points, "null pointer");
    double err = 0.0;
    TDK_TRY(ba_reduce(h, poses, points, REDUCE_SUMS, &err));
    if (sum_sq) *sum_sq = err;
    if (U) TDK_HIP(hipMemcpyAsync(U, h->cur.U, (size_t)h->n_poses * 21 * 8, hipMemcpyDeviceToHost, tdk::stream()));
    if (ea) TDK_HIP(hipMemcpyAsync(ea, h->cur.ea, (size_t)h->n_poses * 6 * 8, hipMemcpyDeviceToHost, tdk::stream()));
    // the per-point sums live as structure of arrays on the device ([6][Q], [3][Q])
    if (V || eb) {
        const size_t Q = (size_t)h->n_points;
        std::vector<double> soa(Q * 9);
        TDK_HIP(hipMemcpyAsync(soa.data(), h->cur.V, Q * 48, hipMemcpyDeviceToHost, tdk::stream()));
        TDK_HIP(hipMemcpyAsync(soa.data() + 6 * Q, h->cur.eb, Q * 24, hipMemcpyDeviceToHost, tdk::stream()));
        TDK_HIP(hipStreamSynchronize(tdk::stream()));
        for (size_t i = 0; i < Q; i++) {
            if (V) for (int m = 0; m < 6; m++) V[6 * i + m] = soa[(size_t)m * Q + i];
            if (eb) for (int a = 0; a < 3; a++) eb[3 * i + a] = soa[(6 + (size_t)a) * Q + i];
        }
    }
    TDK_HIP(hipStreamSynchronize(tdk::stream()));
    return TDK_OK;
}

tdk_status tdk_ba_set_profiling(tdk_ba *h, int enabled) {
    TDK_API_GUARD;
    TDK_REQUIRE(h != nullptr, "handle is NULL");
    h->profiling = enabled != 0;
    h->ev_used = 0;
    for (int k = 0; k < BA_K_COUNT; k++) { h->prof_ms[k] = 0.0; h->prof_launches[k] = 0; }
    return TDK_OK;
}

tdk_status tdk_ba_get_profile(tdk_ba *h, int64_t *launches, double *total_ms) {
    TDK_API_GUARD;
    TDK_REQUIRE(h && launches && total_ms, "null pointer");
    for (int k = 0; k < BA_K_COUNT; k++) { launches[k] = h->prof_launches[k]; total_ms[k] = h->prof_ms[k]; }
    return TDK_OK;
}

tdk_status tdk_ba_step(tdk_ba *h, const double *poses, const double *points, double mu, double *dposes,
                       double *dpoints, double *sum_sq) {
    TDK_API_GUARD;
    TDK_REQUIRE(h && poses && points && dposes && dpoints && sum_sq, "null pointer");
    TDK_REQUIRE(mu >= 0.0, "mu must be non-negative");
    TDK_TRY(ba_upload(h, poses, points));
    TDK_TRY(ba_reduce_launch(h, h->d_poses, h->d_points, REDUCE_STEP, h->cur));
    TDK_TRY(ba_update_launch(h, mu));
    TDK_HIP(hipMemcpyAsync(dposes, h->d_da, (size_t)h->n_poses * 48, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipMemcpyAsync(dpoints, h->d_db, (size_t)h->n_points * 24, hipMemcpyDeviceToHost, tdk::stream()));
    bool singular = false;
    TDK_TRY(ba_wait(h, sum_sq, &singular));
    if (singular) {
        tdk::set_error("reduced camera system is singular (mu = %g)", mu);
        return TDK_ERR_SINGULAR;
    }
    return TDK_OK;
}

// LocalBundleAdjustment.compute (tadataka/local_ba.py:91-134) with the
// parameters resident on the device.  Per damping trial one chain of kernels --
// V*^-1, Schur complement, reduced camera system, back-substitution, and the
// block sums AT THE CANDIDATE into the alternate buffer set -- and one wait for
// the candidate's error.  An accepted candidate swaps the buffer sets, so its
// sums are never computed twice; a rejected one leaves the current sums in place
// for the next damping (the reference recomputes projection and Jacobians per
// trial).  For local windows nothing but the per-pose errors crosses the bus.
tdk_status tdk_ba_solve(tdk_ba *h, double *poses, double *points, int max_iter, double initial_mu, double nu,
                        double absolute_error_threshold, double relative_error_threshold,
                        double *error_history, int *n_iter) {
    TDK_API_GUARD;
    TDK_REQUIRE(h && poses && points, "null pointer");
    TDK_REQUIRE(max_iter >= 0 && initial_mu > 0.0 && nu > 1.0, "bad Levenberg-Marquardt parameters");
    const size_t np6 = (size_t)h->n_poses * 6, nq3 = (size_t)h->n_points * 3;
    const double inv_n = 1.0 / (double)h->n;                    // calc_error is the MEAN squared error (:51-56)
    TDK_TRY(ba_ensure_alt(h));
    double sum_sq = 0.0;
    TDK_TRY(ba_reduce(h, poses, points, REDUCE_STEP, &sum_sq)); // uploads the parameters as well
    double current_error = sum_sq * inv_n, mu = initial_mu;
    if (error_history) error_history[0] = current_error;
    int it = 0;
    for (; it < max_iter; it++) {
        const double error0 = current_error;                    // lm_update's calc_error(poses, points)
        double new_error = 0.0, new_mu = mu;
        // trial dampings of lm_update (:96-113): mu / nu, mu, then mu nu, mu nu^2, ... until no worse
        for (int trial = 0;; trial++) {
            new_mu = trial == 0 ? mu / nu : (trial == 1 ? mu : new_mu * nu);
            TDK_TRY(ba_update_launch(h, new_mu));
            TDK_TRY(ba_reduce_launch(h, h->d_cposes, h->d_cpoints, REDUCE_STEP, h->alt));
            bool singular = false;
            TDK_TRY(ba_wait(h, &sum_sq, &singular));
            if (singular) {
                tdk::set_error("reduced camera system is singular (mu = %g)", new_mu);
                return TDK_ERR_SINGULAR;
            }
            new_error = sum_sq * inv_n;
            if (trial < 2 ? new_error < error0 : !(new_error > error0)) break;
            if (trial > 400) {                                   // mu overflowed to inf long ago
                tdk::set_error("Levenberg-Marquardt damping search does not terminate");
                return TDK_ERR_SINGULAR;
            }
        }
        // accept: the candidate becomes the current parameter set, its sums the current sums
        std::swap(h->d_poses, h->d_cposes);
        std::swap(h->d_points, h->d_cpoints);
        std::swap(h->cur, h->alt);
        mu = new_mu;
        if (error_history) error_history[it + 1] = new_error;
        const double relative_error = fabs((current_error - new_error) / new_error);
        if (new_error < absolute_error_threshold || relative_error < relative_error_threshold) {
            it++;
            break;
        }
        current_error = new_error;
    }
    if (n_iter) *n_iter = it;
    TDK_HIP(hipMemcpyAsync(poses, h->d_poses, np6 * 8, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipMemcpyAsync(points, h->d_points, nq3 * 8, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipStreamSynchronize(tdk::stream()));
    if (h->profiling) ba_collect_profile(h);
    return TDK_OK;
}

}  // extern "C"
