// dvo.hip -- the fused, device-resident DVO path for batches of frame pairs.
//
// One "evaluation" of a pose does, in ONE pass over the source pixels, what the
// reference does in two separate passes per Gauss-Newton iteration
// (tadataka/vo/dvo/__init__.py:93-110):
//   * calc_pose_update (:46-70): warp, mask (in range & z > 0), bilinear
//     samples of the image gradient, Jacobian row (jacobian.py:8-24), weights,
//     and the reduction sum w J^T J / sum w J^T r that solve_linear_equation
//     (tadataka/math.py:32-45) solves by lstsq;
//   * photometric_error (tadataka/metric.py:13-27): warp, mask (in range),
//     bilinear sample of I1, mean squared difference.
// The candidate pose whose error is tested is also the pose the next update
// is linearised at, so error(pose) and the normal equations at the same pose
// share one read of (D0, I0, I1[, W0]).
//
// HBM layout: per pyramid level, struct-of-arrays I0 | D0 | I1 | W0, each
// [n_pairs][stride] float64 with stride = N rounded up to even so that every
// pair starts 16-byte aligned, plus 4 kBlock doubles of padding per array (the
// evaluation's stream loads overrun a block's range).  The image gradient
// (np.gradient of I1) is NOT materialised: its 4+4 bilinear taps are rebuilt
// from 12 neighbouring I1 texels, which removes 16 of the 40 B/px the unfused
// update would read.
//
// Kernels (gfx950, wave64):
//   k_norm_tables     per (pair, level): (x - ox) / fx and (y - oy) / fy, once per
//                     camera upload.
//   k_dvo_eval        1-D XCD-major grid, 256 threads; each thread walks its
//                     block's contiguous pixel range one pixel per step through a
//                     three-stage software pipeline (accumulate n | gathers of n+1
//                     in flight | warp n+2; the warp through per-row terms of the
//                     block's pose tabulated in LDS, see row_terms), keeps 28 f64
//                     accumulators and two
//                     scalar mask counters, transposed wave reduction
//                     (v_permlane swaps + DPP), LDS across the 4 waves, one
//                     30-double partial per block.
//   k_dvo_reduce      grid n_pairs x 256: fixed-order sum of the partials
//                     (bit-reproducible), then -- in loop mode -- lane 0 performs the
//                     monotone accept/reject, the 6x6 solve and the SE(3) update;
//                     the last block publishes the running-pair count to mapped
//                     host memory.
//   k_robust_*,       Student-t / Tukey need global statistics of the masked
//   k_select_*        residuals (tadataka/robust/weights.py:4-35): masked-residual
//                     map, 10 fixed-point variance steps, medians by MSD radix
//                     select with a candidate short cut -- all per pair, on the device.
#include "tdk_math.h"
#include "tdk_runtime.h"
#include "tdk_wave.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

namespace {

using tdk::Cam;

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr int kAcc = 30;       // 21 H + 6 b + sum_sq + n_update + n_error
constexpr int kAccPad = 32;
constexpr int kMaxLevels = 16;
constexpr double kHuberK = 1.345;     // tadataka/robust/weights.py:38
constexpr double kTukeyBeta = 4.6851; // :21
constexpr double kTukeyC = 1.4826;
constexpr double kStudentNu = 5.0;    // :4

struct LevelPtrs {
    const double *I0, *D0, *I1, *W0;
    const double *tab;  // [n_pairs][W + H] normalised pixel coordinates of camera 0 (k_norm_tables)
    int64_t stride;  // elements between consecutive pairs
    int H, W;
    int64_t N;
};

struct PairParams {  // per pair, device memory
    double cam0[4];
    double cam1[4];
};

enum { ST_RUNNING = 0, ST_DONE = 1 };

struct Accum {
    double v[kAcc];
};

// q / z for two numerators sharing one denominator, correctly rounded: the
// v_rcp_f64 seed refined by two Newton steps, then one Markstein correction per
// quotient -- the same fma sequence the compiler emits for an IEEE f64 `/`
// (minus the range scaling, irrelevant for depths), with the reciprocal shared.
// `rcp` returns the refined reciprocal (used for the Jacobian's 1/z).
__device__ __forceinline__ void div2_shared(double nx, double ny, double z, double &qx, double &qy,
                                            double &rcp) {
    double y = __builtin_amdgcn_rcp(z);
    double e = __builtin_fma(-z, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-z, y, 1.0);
    y = __builtin_fma(y, e, y);
    double a = nx * y, b = ny * y;
    double ra = __builtin_fma(-z, a, nx), rb = __builtin_fma(-z, b, ny);
    qx = __builtin_fma(ra, y, a);
    qy = __builtin_fma(rb, y, b);
    rcp = y;
}

// Load base[byte_off / 8] with a 32-bit unsigned byte offset from a
// block-uniform base (global_load_dwordx2 v, voffset, s[base]).
// ... for data that is read once (the D0 / I0 / W0 streams): non-temporal, so that it does not push the I1 rows the
// neighbouring pixels are about to gather out of the caches (the probe's HBM traffic was 28.3 B/px for 24 algorithmic:
// half of the second touches of an I1 row missed; 350 -> 327 us per full-resolution probe, full evaluation 569 -> 564)
__device__ __forceinline__ double ldo_stream(const double *__restrict__ base, uint32_t byte_off) {
    return __builtin_nontemporal_load(reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + byte_off));
}

__device__ __forceinline__ double ldo(const double *__restrict__ base, uint32_t byte_off) {
    return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + byte_off);
}

// What the gradient / bilinear formulas need of the warped coordinate.
struct Warped {
    double w00, w01, w10, w11;   // bilinear weights
    int c0, r0;                  // lower texel
};

struct Taps {   // the 12 I1 texels around (c0, r0): rows r0-1 .. r0+2
    double t0, t1, a0, a1, a2, a3, b0, b1, b2, b3, u0, u1;
};

// Two adjacent texels with one 16-byte load (8-byte aligned only: the address
// follows the warped coordinate): six gather instructions per pixel instead of
// twelve, and 40 % fewer L1 tag look-ups.
typedef double double2_u __attribute__((ext_vector_type(2), aligned(8)));

__device__ __forceinline__ double2_u ldo2(const double *__restrict__ base, uint32_t byte_off) {
    return *reinterpret_cast<const double2_u *>(reinterpret_cast<const char *>(base) + byte_off);
}

// TWICE the np.gradient of I1 sampled bilinearly at the warped coordinate
// (vo/dvo/jacobian.py:27-29 + interpolation): central differences inside.  The
// factor 0.5 is an exact scaling, so it is folded into the focal length the
// gradient is multiplied with (half_f below) instead of costing two multiplies.
__device__ __forceinline__ void gradient2_inside(const Taps &t, const Warped &p, double &gx, double &gy) {
    gx = (t.a2 - t.a0) * p.w00 + (t.a3 - t.a1) * p.w01 + (t.b2 - t.b0) * p.w10 + (t.b3 - t.b1) * p.w11;
    gy = (t.b1 - t.t0) * p.w00 + (t.b2 - t.t1) * p.w01 + (t.u0 - t.a1) * p.w10 + (t.u1 - t.a2) * p.w11;
}

// ... and one-sided differences on the border rows / columns.
__device__ __forceinline__ void gradient2_clamped(const Taps &t, const Warped &p, int H, int W, double &gx,
                                                  double &gy) {
    int c0 = p.c0, r0 = p.r0, c1 = min(c0 + 1, W - 1), r1 = min(r0 + 1, H - 1);
    // one-sided difference: weight 1, central: 1/2 -- times two, as in gradient2_inside
    double sx0 = (c0 == 0 || c0 == W - 1) ? 2.0 : 1.0;
    double sx1 = (c1 == 0 || c1 == W - 1) ? 2.0 : 1.0;
    double sy0 = (r0 == 0 || r0 == H - 1) ? 2.0 : 1.0;
    double sy1 = (r1 == 0 || r1 == H - 1) ? 2.0 : 1.0;
    gx = (t.a2 - t.a0) * sx0 * p.w00 + (t.a3 - t.a1) * sx1 * p.w01 + (t.b2 - t.b0) * sx0 * p.w10 +
         (t.b3 - t.b1) * sx1 * p.w11;
    gy = (t.b1 - t.t0) * sy0 * p.w00 + (t.b2 - t.t1) * sy0 * p.w01 + (t.u0 - t.a1) * sy1 * p.w10 +
         (t.u1 - t.a2) * sy1 * p.w11;
}

// Robust weight of one residual; `ws` is the pair's scale statistic
// (Student-t: variance; Tukey: sigma_mad).  What solve_linear_equation ends up
// applying is sqrt(w)^2 of whatever compute_weights returned, i.e. the value
// below (compute_weights_student_t itself already returns a square root).
// Student-t and Tukey get the RECIPROCAL of the statistic (one division per block instead of one or two IEEE
// division sequences of 11 instructions per pixel; the quotients differ from the reference's in the last bit,
// the weights are held to 1e-9 like every sum they enter).
template <int WMODE>
__device__ __forceinline__ double robust_weight(double r, double w0, double ws) {
    if (WMODE == TDK_W_STUDENT_T) {
        const double t = __builtin_fma(r * r, ws, kStudentNu);          // nu + r^2 / variance
        double y = __builtin_amdgcn_rcp(t);
        y = __builtin_fma(__builtin_fma(-t, y, 1.0), y, y);
        y = __builtin_fma(__builtin_fma(-t, y, 1.0), y, y);
        // the square root of q = (nu + 1) / t in (0, 1.2]: v_rsq_f64 + one Goldschmidt step + one correction (8
        // operations; the compiler's sqrt also rescales denormal arguments and tests for 0 / Inf / NaN: 20)
        const double q = (kStudentNu + 1.0) * y;
        const double y0 = __builtin_amdgcn_rsq(q);
        double g = q * y0, h = 0.5 * y0;
        const double e = __builtin_fma(-h, g, 0.5);
        g = __builtin_fma(g, e, g);
        h = __builtin_fma(h, e, h);
        g = __builtin_fma(__builtin_fma(-g, g, q), h, g);
        // r = +-Inf, or r * r * ws beyond the double range: t = Inf, rcp = 0 and the Newton steps make NaN of 0 * Inf;
        // the reference's (nu + 1) / (nu + Inf) = 0 there, weight 0.  A NaN residual stays NaN.
        return t == INFINITY ? 0.0 : (q > 0.0 ? g : q);
    }
    if (WMODE == TDK_W_TUKEY) {
        const double x = r * ws, q = x * (1.0 / kTukeyBeta), u = 1.0 - q * q;
        return fabs(x) <= kTukeyBeta ? u * u : 0.0;
    }
    if (WMODE == TDK_W_HUBER) {
        // |r| <= 1 < k on [0, 1] images (F4): the division sits behind a branch
        // that is skipped unless some lane of the wave has an outlier
        double ar = fabs(r), w = 1.0;
        if (__builtin_amdgcn_ballot_w64(ar > kHuberK) != 0) w = ar > kHuberK ? kHuberK / ar : 1.0;
        return w;
    }
    if (WMODE == TDK_W_MAP) return w0;
    return 1.0;
}

// Block-level tail of the evaluation kernel: wave64 reduction, LDS across the
// waves, one partial per block.
__device__ __forceinline__ void store_partials(const Accum &acc, double (*red)[kAccPad], int pair, int blk,
                                               int nblk, double *__restrict__ partials) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double s = tdk::wave_sum_transposed(acc.v);
    if ((lane & 1) == 0) red[wave][lane >> 1] = s;
    __syncthreads();
    if (threadIdx.x < kAcc) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < kWaves; w++) s += red[w][threadIdx.x];
        partials[((int64_t)pair * nblk + blk) * kAccPad + threadIdx.x] = s;
    }
}

// rows of the row-term table of an evaluation block: its pixel range, the pipeline's overrun, a partial first row
__host__ __device__ __forceinline__ int eval_rows(int64_t chunk, int W) {
    return (int)((chunk + 3 * kBlock + W - 1) / W) + 1;
}

struct BlockSetup {   // block-uniform pose and (level-scaled) cameras
    double P[12], c[4], fx0, fy0, ox0, oy0;
};

__device__ __forceinline__ void load_setup(BlockSetup &b, const PairParams *__restrict__ params,
                                           const double *__restrict__ poses, int pair, double scale) {
#pragma unroll
    for (int i = 0; i < 12; i++) b.P[i] = poses[12 * pair + i];
    const PairParams pp = params[pair];
    // tadataka.camera.resize: focal length and offset both scale (camera/model.py:69-74)
    b.c[0] = pp.cam1[0] * scale; b.c[1] = pp.cam1[1] * scale;
    b.c[2] = pp.cam1[2] * scale; b.c[3] = pp.cam1[3] * scale;
    b.fx0 = pp.cam0[0] * scale; b.fy0 = pp.cam0[1] * scale;
    b.ox0 = pp.cam0[2] * scale; b.oy0 = pp.cam0[3] * scale;
}

// tab[pair][0 .. W) = (x - ox) / fx, tab[pair][W .. W + H) = (y - oy) / fy of the
// level-scaled camera 0 (rust_bindings.camera.normalize_in_place, src/camera.rs):
// they depend on the pair and the level only, so they are built when the
// cameras are uploaded and the evaluation blocks just copy them into LDS.
__global__ void k_norm_tables(const PairParams *__restrict__ params, double scale, int W, int H,
                              double *__restrict__ tab) {
    const int pair = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W + H) return;
    const PairParams pp = params[pair];
    // un-fused, as NumPy computes them: with contraction the compiler turns x - offset * scale into one fma (the
    // product unrounded), the table entries of a level > 0 differ from the reference's in the last bit for cameras
    // whose scaled offset is not exact, and at the identity prior border pixels change sides of the inclusive mask
    // (found by tests/test_gpu_fuzz.py: 12 of 1975 pixels of a 32 x 141 level)
    {
#pragma clang fp contract(off)
        const double fx0 = pp.cam0[0] * scale, fy0 = pp.cam0[1] * scale;
        const double ox0 = pp.cam0[2] * scale, oy0 = pp.cam0[3] * scale;
        tab[(size_t)pair * (W + H) + i] = i < W ? ((double)i - ox0) / fx0 : ((double)(i - W) - oy0) / fy0;
    }
}

// ---------------------------------------------------------------------------
// k_dvo_eval: one evaluation F(pose) -> (error, H, b, mask sizes) for every pair.
//
// Software pipelined over three pixels of a thread.  While pixel n is
// accumulated, the 12 texels of pixel n+1 are in flight and pixel n+2 is warped:
//
//     accumulate(n)        <- taps(n), issued one step earlier
//     taps(n+1) issued     <- warp(n+1), computed one step earlier
//     warp(n+2)            <- depth(n+2), loaded at the top of the step
//
// so a gather has a whole warp computation (plus the other waves' issue slots)
// to land before it is needed, instead of being waited for right after issue
// (measured: 60 % -> 34 % of wave-cycles in s_waitcnt, VALU issue 58 % -> 80 %).
// One tap buffer and two ping-pong pixel records; one pixel per thread per step.
// Every load is unconditional with a clamped address -- a branch around a load
// makes the compiler fall back to s_waitcnt vmcnt(0), which serialises the
// stages again -- and scheduling barriers keep the stages from being interleaved
// (that would double the live tap and pixel registers and drop to 1-2 waves/SIMD).
// ---------------------------------------------------------------------------
struct Pixel {       // what survives from the warp of one pixel until it is accumulated
    double sx, sy, rz;   // P1x / z', P1y / z', 1 / z' with z' = P1z + 1e-16
    double wx1, wy1;     // fractional parts of the warped coordinate
    int c0, r0;          // its lower texel, clamped into the image (valid or not)
    int mask;            // 0: outside the image; 1: error mask only; 2: error and update mask (P1z > 0)
    int inside;          // the 4x4 texel neighbourhood of (c0, r0) lies strictly inside the image
};

// clamp(v, 0, hi) in one instruction
__device__ __forceinline__ int clamp0(int v, int hi) {
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(v), "s"(hi));
    return r;
}

struct Samples;

// pixel coordinate of the warped point, the reference's un-fused x * f + o
__device__ __forceinline__ void sp_coordinate(const Pixel &p, const double *c, double &u, double &v) {
#pragma clang fp contract(off)
    u = p.sx * c[0] + c[2];
    v = p.sy * c[1] + c[3];
}

// The warp of one pixel, q = R (d (xn, yn, 1)) + t, as q_k = d (R_k0 xn + (R_k1 yn + R_k2)) + t_k: the bracket depends
// on the pixel's ROW and the (block-uniform) pose only, so the evaluation blocks tabulate it per row (row_terms) and a
// pixel costs six fmas instead of the eleven operations of p = d (xn, yn, 1), q = R p + t.  At the identity prior --
// where the whole border lies on the inclusive mask edge and the last bit of q decides -- the three steps
// are exact except the one rounding of d * xn, which is the reference's own; elsewhere the two orders differ by
// roundings (1e-16 relative), as the fma chain did.
__device__ __forceinline__ void row_terms(const double *P, double yn, double &r0, double &r1, double &r2) {
    r0 = __builtin_fma(P[1], yn, P[2]);
    r1 = __builtin_fma(P[4], yn, P[5]);
    r2 = __builtin_fma(P[7], yn, P[8]);
}

__device__ __forceinline__ void sp_warp_rows(Pixel &p, bool live, double xn, double ry0, double ry1, double ry2,
                                             double d0, int H, int W, const double *P, const double *c) {
    double qx = __builtin_fma(d0, __builtin_fma(P[0], xn, ry0), P[9]);
    double qy = __builtin_fma(d0, __builtin_fma(P[3], xn, ry1), P[10]);
    double qz = __builtin_fma(d0, __builtin_fma(P[6], xn, ry2), P[11]);
    {
#pragma clang fp contract(off)
        double z = qz + tdk::kEps16;
        div2_shared(qx, qy, z, p.sx, p.sy, p.rz);
    }
    double u, v;
    sp_coordinate(p, c, u, v);
    // an int, not two bools: loop-carried bools end up as byte arithmetic in VGPRs
    const bool valid = live && u >= 0.0 && u <= (double)(W - 1) && v >= 0.0 && v <= (double)(H - 1);
    p.mask = valid ? (qz > 0.0 ? 2 : 1) : 0;
    // u - floor(u) as one v_fract_f64 and the texel by truncation: the same doubles / integers for every u >= 0, i.e.
    // for every pixel inside the mask.  v_cvt_i32_f64 saturates and maps NaN to 0: masked-out pixels get a clamped,
    // loadable texel
    p.wx1 = __builtin_amdgcn_fract(u);
    p.wy1 = __builtin_amdgcn_fract(v);
    p.c0 = clamp0((int)u, W - 1);
    p.r0 = clamp0((int)v, H - 1);
    p.inside = p.c0 >= 1 && p.c0 <= W - 3 && p.r0 >= 1 && p.r0 <= H - 3;
}

// ... for callers without a row table (the statistics passes, the samplers): the same operations
__device__ __forceinline__ void sp_warp(Pixel &p, bool live, double xn, double yn, double d0, int H, int W,
                                        const double *P, const double *c) {
    double r0, r1, r2;
    row_terms(P, yn, r0, r1, r2);
    sp_warp_rows(p, live, xn, r0, r1, r2, d0, H, W, P, c);
}

// The 12 texels of one pixel as six 16-byte loads whose addresses are clamped
// into the image, so that they can be issued unconditionally (no branch around a
// load: the compiler keeps exact vmcnt bookkeeping across the pipeline stages).
// Interior pixels load (c0, c0+1) on rows r0-1 and r0+2 and (c0-1, c0), (c0+1, c0+2)
// on rows r0 and r0+1; on the border the pairs are shifted inwards and
// sp_fix_border() rebuilds the replicated texels.
struct TapPairs {
    double2_u tm, a01, a23, b01, b23, u01;
};

__device__ __forceinline__ void sp_issue_taps(TapPairs &q, const double *__restrict__ I1, int H, int W, int c0,
                                              int r0, int inside) {
    const uint32_t rowb = (uint32_t)W * 8u;
    uint32_t om, oal, oar, obl, obr, ou;   // byte offsets of the six pairs
    if (__builtin_amdgcn_ballot_w64(!inside) == 0) {   // wave-uniform; address arithmetic only
        const uint32_t o0 = (__umul24((uint32_t)r0, (uint32_t)W) + (uint32_t)c0) * 8u;   // full-rate 24-bit multiply
        om = o0 - rowb; oal = o0 - 8u; oar = o0 + 8u;
        obl = oal + rowb; obr = oar + rowb; ou = o0 + 2u * rowb;
    } else {
        const uint32_t row0 = __umul24((uint32_t)r0, rowb);
        const uint32_t rowm = r0 > 0 ? row0 - rowb : row0;
        const uint32_t row1 = r0 < H - 1 ? row0 + rowb : row0;
        const uint32_t row2 = r0 < H - 2 ? row1 + rowb : row1;
        const uint32_t cl = (uint32_t)max(c0 - 1, 0) * 8u;       // pair (c0-1, c0)
        const uint32_t cm = (uint32_t)min(c0, W - 2) * 8u;       // pair (c0, c0+1)
        const uint32_t cr = (uint32_t)min(c0 + 1, W - 2) * 8u;   // pair (c0+1, c0+2)
        om = rowm + cm; oal = row0 + cl; oar = row0 + cr;
        obl = row1 + cl; obr = row1 + cr; ou = row2 + cm;
    }
    q.tm = ldo2(I1, om);
    q.a01 = ldo2(I1, oal);
    q.a23 = ldo2(I1, oar);
    q.b01 = ldo2(I1, obl);
    q.b23 = ldo2(I1, obr);
    q.u01 = ldo2(I1, ou);
}

__device__ __forceinline__ Taps sp_unpack(const TapPairs &q) {
    Taps t;
    t.t0 = q.tm.x; t.t1 = q.tm.y;
    t.a0 = q.a01.x; t.a1 = q.a01.y; t.a2 = q.a23.x; t.a3 = q.a23.y;
    t.b0 = q.b01.x; t.b1 = q.b01.y; t.b2 = q.b23.x; t.b3 = q.b23.y;
    t.u0 = q.u01.x; t.u1 = q.u01.y;
    return t;
}

// texel (r, clamp(c)) for the columns the shifted pairs could not reach
__device__ __forceinline__ void sp_fix_border(Taps &t, int c0, int W) {
    if (c0 == 0) { t.a1 = t.a0; t.b1 = t.b0; }                    // pair loaded at (0, 1), wanted (0, 0)
    if (c0 >= W - 2) { t.a2 = t.a3; t.b2 = t.b3; }                // loaded at (W-2, W-1), wanted (W-1, W-1)
    if (c0 == W - 1) { t.t0 = t.t1; t.u0 = t.u1; }                // loaded at (W-2, W-1), wanted (W-1, W-1)
}

struct Samples {     // everything that is loaded for the pixel being accumulated next
    TapPairs q;
    double i0, i1, w0;
};

// Error-only evaluation (the "probe" of a candidate pose, see reduce_pair): only the four texels
// of the bilinear sample of I1 are needed -- two 16-byte loads, (c, c + 1) on rows r0 and r0 + 1
// with c = min(c0, W - 2) and the row clamped, unconditional like every load of the pipeline.
__device__ __forceinline__ void sp_issue_taps_probe(TapPairs &q, const double *__restrict__ I1, int H, int W, int c0,
                                                    int r0) {
    const uint32_t rowb = (uint32_t)W * 8u;
    const uint32_t row0 = __umul24((uint32_t)r0, rowb);
    const uint32_t row1 = r0 < H - 1 ? row0 + rowb : row0;
    const uint32_t cm = (uint32_t)min(c0, W - 2) * 8u;
    q.a01 = ldo2(I1, row0 + cm);
    q.b01 = ldo2(I1, row1 + cm);
}

// I1 at the warped coordinate and the squared photometric error term (metric.py:24-27).  ONE
// function for the full and the probe evaluation: the same expression, contracted the same way,
// so both give the same bits for the same pose.
__device__ __forceinline__ double error_term(double a1, double a2, double b1, double b2, double w00, double w01,
                                             double w10, double w11, double i0) {
    const double i1w = a1 * w00 + a2 * w01 + b1 * w10 + b2 * w11;
    const double e = i0 - i1w;
    return e * e;
}

template <int WMODE, bool PROBE>
__device__ __forceinline__ void sp_issue(Samples &s, const Pixel &p, uint32_t off, const double *__restrict__ I0,
                                         const double *__restrict__ I1, const double *__restrict__ W0, int H,
                                         int W, const double *c) {
    s.i0 = ldo_stream(I0, off);
    if (PROBE) {
        sp_issue_taps_probe(s.q, I1, H, W, p.c0, p.r0);
        return;
    }
    s.i1 = ldo(I1, off);
    if (WMODE == TDK_W_MAP) s.w0 = ldo_stream(W0, off);
    sp_issue_taps(s.q, I1, H, W, p.c0, p.r0, p.inside);
}

template <int WMODE, bool PROBE>
__device__ __forceinline__ void sp_accumulate(Accum &a, int &n_error, int &n_update, const Samples &s,
                                              const Pixel &p, double ws, int H, int W, const double *c) {
    // mask sizes: one scalar popcount per wave instead of two f64 adds per lane
    n_error += __builtin_popcountll(__builtin_amdgcn_ballot_w64(p.mask != 0));
    n_update += __builtin_popcountll(__builtin_amdgcn_ballot_w64(p.mask == 2));
    if (p.mask == 0) return;
    if (PROBE) {
        // texels (r0, c0), (r0, c0 + 1), (r0 + 1, c0), (r0 + 1, c0 + 1), replicated at the right edge
        const bool last_col = p.c0 == W - 1;
        const double a1 = last_col ? s.q.a01.y : s.q.a01.x, a2 = s.q.a01.y;
        const double b1 = last_col ? s.q.b01.y : s.q.b01.x, b2 = s.q.b01.y;
        const double wx1 = p.wx1, wy1 = p.wy1;
        const double wx0 = 1.0 - wx1, wy0 = 1.0 - wy1;
        a.v[27] += error_term(a1, a2, b1, b2, wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1, s.i0);
        return;
    }
    Warped w;
    w.c0 = p.c0; w.r0 = p.r0;
    Taps t = sp_unpack(s.q);
    const bool border = __builtin_amdgcn_ballot_w64(!p.inside) != 0;   // wave-uniform
    if (border) sp_fix_border(t, w.c0, W);
    // (lx + 1) - u and 1 - (u - lx) are the same double: u - lx is exact and a
    // multiple of ulp(u), so 1 - (u - lx) is representable
    const double wx1 = p.wx1, wy1 = p.wy1;
    const double wx0 = 1.0 - wx1, wy0 = 1.0 - wy1;
    w.w00 = wx0 * wy0; w.w01 = wx1 * wy0; w.w10 = wx0 * wy1; w.w11 = wx1 * wy1;
    double gx, gy;
    if (border) gradient2_clamped(t, w, H, W, gx, gy);
    else gradient2_inside(t, w, gx, gy);
    // photometric error term (metric.py:24-27): no z test here
    a.v[27] += error_term(t.a1, t.a2, t.b1, t.b2, w.w00, w.w01, w.w10, w.w11, s.i0);
    if (p.mask != 2) return;  // update mask adds P1z > 0 (vo/dvo/__init__.py:49)
    // Jacobian row (vo/dvo/jacobian.py:8-24) with X = x/z, Y = y/z factored out:
    //   [fgx/z, fgy/z, -(fgx X + fgy Y)/z, -fgx XY - fgy (1 + Y^2), fgx (1 + X^2) + fgy XY, fgy X - fgx Y]
    const double X = p.sx, Y = p.sy;
    const double fgx = (0.5 * c[0]) * gx, fgy = (0.5 * c[1]) * gy;   // gx, gy are twice the gradient
    double J[6];
    // with s = fgx X + fgy Y the rows -fgx XY - fgy (1 + Y^2) and fgx (1 + X^2) + fgy XY are -(fgy + Y s) and fgx + X s:
    // 11 instead of 17 operations (round 5: 556.6 -> 552.6 us per full-resolution launch)
    const double sxy = __builtin_fma(fgy, Y, fgx * X);
    J[0] = fgx * p.rz;
    J[1] = fgy * p.rz;
    J[2] = -sxy * p.rz;
    J[3] = __builtin_fma(-Y, sxy, -fgy);
    J[4] = __builtin_fma(X, sxy, fgx);
    J[5] = fgy * X - fgx * Y;
    double r = s.i0 - s.i1;  // un-warped residual (vo/dvo/__init__.py:90)
    if (WMODE == TDK_W_HUBER) {
        // sum w J J^T = sum J J^T + sum (w - 1) J J^T.  Huber's weight is exactly 1 unless |r| > k, which never
        // happens on [0, 1] images (F4): the first sum costs what weights=None costs, the second runs only in waves
        // that hold an outlier (the six w * J products were 2.4 % of the kernel).  The rare path is inline asm on
        // purpose: written as C the compiler merges the two paths' accumulators through copies -- 212 VGPRs, two waves
        // per SIMD (the round-1 attempt, 19 % slower); tied "+v" operands keep 158.
        int k = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) {
#pragma unroll
            for (int q = i; q < 6; q++) a.v[k++] += J[i] * J[q];
            a.v[21 + i] += J[i] * r;
        }
        const double ar = fabs(r);
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(ar > kHuberK) != 0, 0)) {
            const double c = ar > kHuberK ? kHuberK / ar - 1.0 : 0.0;
            k = 0;
#define TDK_FMAC(ACC, X, Y) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(ACC) : "v"(X), "v"(Y))
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const double cj = c * J[i];
#pragma unroll
                for (int q = i; q < 6; q++) { TDK_FMAC(a.v[k], cj, J[q]); k++; }
                TDK_FMAC(a.v[21 + i], cj, r);
            }
#undef TDK_FMAC
        }
        return;
    }
    double wgt = robust_weight<WMODE>(r, s.w0, ws);
    const bool unit_w = (WMODE == TDK_W_NONE);
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        double wj = unit_w ? J[i] : wgt * J[i];
#pragma unroll
        for (int q = i; q < 6; q++) a.v[k++] += wj * J[q];
        a.v[21 + i] += wj * r;
    }
}

template <int WMODE, bool PROBE>
__device__ __forceinline__ void eval_body(const LevelPtrs &L, const PairParams *__restrict__ params,
                                          const double *__restrict__ poses, const double *__restrict__ wscale,
                                          double scale, int64_t chunk, int pair, int blk, int nblk,
                                          double *__restrict__ partials) {
    BlockSetup b;
    load_setup(b, params, poses, pair, scale);
    const double ws = (WMODE == TDK_W_STUDENT_T || WMODE == TDK_W_TUKEY) ? 1.0 / wscale[pair] : 1.0;   // see robust_weight

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double(*red)[kAccPad] = reinterpret_cast<double(*)[kAccPad]>(smem);
    double *xn_tab = reinterpret_cast<double *>(smem + sizeof(double) * kWaves * kAccPad);
    double *ry_tab = xn_tab + L.W;      // [eval_rows(chunk, W)][3]: the row terms of the warp at this block's pose
    const int start = (int)(blk * chunk);
    const int row0 = start / L.W;       // the first image row of this block's pixel range: ry_tab is relative to it
    {
        const double *__restrict__ tab = L.tab + (size_t)pair * (L.W + L.H);
        for (int i = threadIdx.x; i < L.W; i += kBlock) xn_tab[i] = tab[i];
        // only the rows the block's range (and the pipeline's overrun of < 3 kBlock pixels) can touch; rows beyond
        // the image belong to masked-out pixels and stay whatever the LDS held
        const int rows = eval_rows(chunk, L.W);
        for (int i = threadIdx.x; i < rows; i += kBlock) {
            if (row0 + i >= L.H) continue;
            double r0, r1, r2;
            row_terms(b.P, tab[L.W + row0 + i], r0, r1, r2);
            ry_tab[3 * i] = r0; ry_tab[3 * i + 1] = r1; ry_tab[3 * i + 2] = r2;
        }
    }
    __syncthreads();

    const int64_t base = (int64_t)pair * L.stride;
    const double *__restrict__ I0 = L.I0 + base;
    const double *__restrict__ D0 = L.D0 + base;
    const double *__restrict__ I1 = L.I1 + base;
    const double *__restrict__ W0 = (WMODE == TDK_W_MAP) ? L.W0 + base : nullptr;
    const int W = L.W, H = L.H;
    const int N = (int)L.N;

    Accum acc;
#pragma unroll
    for (int i = 0; i < kAcc; i++) acc.v[i] = 0.0;

    const int end = (int)min((int64_t)N, (int64_t)start + chunk);
    // iw: the next pixel to warp, (x, y) its coordinates (y relative to row0), advanced by kBlock per step
    int iw = start + (int)threadIdx.x;
    int y = iw / W, x = iw - y * W;
    y -= row0;
    const int step_y = kBlock / W, step_x = kBlock - step_y * W;
#define TDK_ADVANCE()                      \
    do {                                   \
        iw += kBlock;                      \
        x += step_x;                       \
        y += step_y;                       \
        if (x >= W) { x -= W; y += 1; }    \
    } while (0)
    // stream loads run up to 3 kBlock pixels past `end` (masked out by `live`): into the
    // next chunk or, for the last pair, into the padding every level array is allocated with
#define TDK_OFF(i) ((uint32_t)(i) * 8u)
#define TDK_DEPTH() ldo_stream(D0, TDK_OFF(iw))

    Pixel pa, pb;
    Samples s;
    s.w0 = 1.0;
    int n_error = 0, n_update = 0;   // wave-uniform (SGPRs)
    // prologue: warp pixels 0 and 1, issue the loads of pixel 0
    double d = TDK_DEPTH();
    sp_warp_rows(pa, iw < end, xn_tab[x], ry_tab[3 * y], ry_tab[3 * y + 1], ry_tab[3 * y + 2], d, H, W, b.P, b.c);
    TDK_ADVANCE();
    d = TDK_DEPTH();
    sp_warp_rows(pb, iw < end, xn_tab[x], ry_tab[3 * y], ry_tab[3 * y + 1], ry_tab[3 * y + 2], d, H, W, b.P, b.c);
    TDK_ADVANCE();
    sp_issue<WMODE, PROBE>(s, pa, TDK_OFF(iw - 2 * kBlock), I0, I1, W0, H, W, b.c);
    // steady state: iw - 2 kBlock is the pixel being accumulated, iw the one being warped.
    // The scheduling barriers keep the three stages apart: interleaving them
    // would keep two tap sets and two half-warped pixels alive at once.
#define TDK_STAGE_FENCE() __builtin_amdgcn_sched_barrier(0)
    while (iw - 2 * kBlock < end) {
        d = TDK_DEPTH();
        sp_accumulate<WMODE, PROBE>(acc, n_error, n_update, s, pa, ws, H, W, b.c);
        TDK_STAGE_FENCE();
        sp_issue<WMODE, PROBE>(s, pb, TDK_OFF(iw - kBlock), I0, I1, W0, H, W, b.c);
        TDK_STAGE_FENCE();
        sp_warp_rows(pa, iw < end, xn_tab[x], ry_tab[3 * y], ry_tab[3 * y + 1], ry_tab[3 * y + 2], d, H, W, b.P, b.c);
        TDK_ADVANCE();
        TDK_STAGE_FENCE();

        d = TDK_DEPTH();
        sp_accumulate<WMODE, PROBE>(acc, n_error, n_update, s, pb, ws, H, W, b.c);
        TDK_STAGE_FENCE();
        sp_issue<WMODE, PROBE>(s, pa, TDK_OFF(iw - kBlock), I0, I1, W0, H, W, b.c);
        TDK_STAGE_FENCE();
        sp_warp_rows(pb, iw < end, xn_tab[x], ry_tab[3 * y], ry_tab[3 * y + 1], ry_tab[3 * y + 2], d, H, W, b.P, b.c);
        TDK_ADVANCE();
        TDK_STAGE_FENCE();
    }
#undef TDK_STAGE_FENCE
#undef TDK_ADVANCE
#undef TDK_DEPTH
#undef TDK_OFF
    if ((threadIdx.x & 63) == 0) {   // the wave's mask sizes ride in lane 0's accumulators
        acc.v[28] = (double)n_update;
        acc.v[29] = (double)n_error;
    }
    store_partials(acc, red, pair, blk, nblk, partials);
}

enum { MODE_FULL0 = 0, MODE_PROBE = 1, MODE_FULLK = 2, MODE_FUSED = 3 };   // what a pair's next evaluation is (reduce_pair)

// Occupancy is not what limits this kernel (round 5, both measured on the bench batch and not kept):
//   * forcing four waves per SIMD (amdgpu_waves_per_eu(4, 4): 128 VGPRs, 116 bytes of scratch per lane in the
//     pipeline) makes a full-resolution launch 3.30 ms instead of 0.553;
//   * 14 of the 21 H accumulators in LDS instead of registers ([14][256] doubles per block, ds_add_f64 of the
//     rounded product -- the LDS pipe is otherwise idle): 127 VGPRs without scratch, 38 KB of LDS, four waves per
//     SIMD -- 0.560 ms against 0.553.  The fourth wave finds no free issue slots: FP64 issue at the power cap.
template <int WMODE>
__global__ __launch_bounds__(kBlock) void k_dvo_eval(LevelPtrs L, const PairParams *__restrict__ params,
                                                        const double *__restrict__ poses,
                                                        const int *__restrict__ state,
                                                        const int *__restrict__ mode,
                                                        const double *__restrict__ wscale, double scale,
                                                        int64_t chunk, int n_pairs, int nblk,
                                                        double *__restrict__ partials) {
    // 1-D grid.  Workgroups are dealt to the 8 XCDs round-robin, each XCD has its own L2: XCD k
    // takes pairs k, k + 8, ... one after the other, the blocks of a pair (whose tap halos and
    // stream lines overlap) consecutively -- so they meet in one L2, close in time.
    // (n_pairs < 0: fewer than 8 pairs -- XCD-major order would leave 8 - n XCDs idle; the blocks of a pair
    // are dealt to all XCDs instead, pair = blockIdx / nblk)
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int pair = n_pairs < 0 ? (int)blockIdx.x / nblk : (q / nblk) * 8 + xcd;
    const int blk = n_pairs < 0 ? (int)blockIdx.x - pair * nblk : q - (q / nblk) * nblk;
    if (pair >= (n_pairs < 0 ? -n_pairs : n_pairs)) return;
    if (state != nullptr && state[pair] != ST_RUNNING) return;
    // block-uniform: a candidate pose is first PROBED -- error only, 50 of the 118 FP64 operations
    // and 4 of the 12 texels per pixel -- and evaluated in full only once it has been accepted
    if (mode != nullptr && mode[pair] == MODE_PROBE)
        eval_body<WMODE, true>(L, params, poses, wscale, scale, chunk, pair, blk, nblk, partials);
    else
        eval_body<WMODE, false>(L, params, poses, wscale, scale, chunk, pair, blk, nblk, partials);
}

// The probe body on its own, for launches in which every running pair tests a candidate (the round after a level's
// first evaluation, typically; tdk_dvo_photometric_error): inside k_dvo_eval it shares the full body's 156 VGPRs and
// runs at 3 waves per SIMD; alone it needs 62 and runs at 8 -- 403 -> 352 us per full-resolution launch, 5.4 TB/s
// (0.67 of the HBM peak) on the 24 B/px it reads.  The error does not depend on the weights (metric.py:13-39).
__global__ __launch_bounds__(kBlock) void k_dvo_probe(LevelPtrs L, const PairParams *__restrict__ params,
                                                      const double *__restrict__ poses,
                                                      const int *__restrict__ state, double scale, int64_t chunk,
                                                      int n_pairs, int nblk, double *__restrict__ partials) {
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int pair = n_pairs < 0 ? (int)blockIdx.x / nblk : (q / nblk) * 8 + xcd;
    const int blk = n_pairs < 0 ? (int)blockIdx.x - pair * nblk : q - (q / nblk) * nblk;
    if (pair >= (n_pairs < 0 ? -n_pairs : n_pairs)) return;
    if (state != nullptr && state[pair] != ST_RUNNING) return;
    eval_body<TDK_W_NONE, true>(L, params, poses, nullptr, scale, chunk, pair, blk, nblk, partials);
}

// 8-bit frames -> float64 in [0, 1] as skimage.img_as_float converts them: x * (1 / 255) -- the product with the rounded
// reciprocal (skimage/util/dtype.py: np.multiply(image, 1. / imax_in)), which differs from x / 255 in the last bit for 24
// of the 256 values.  One launch for a range of pairs.
// Two pixels per thread: a 2-byte load and one 16-byte store per lane, both fully coalesced.
__global__ __launch_bounds__(kBlock) void k_u8_to_f64(const uint8_t *__restrict__ src, double *__restrict__ dst,
                                                      int64_t N, int64_t stride) {
    const uint8_t *s = src + (int64_t)blockIdx.y * N;
    double *d = dst + (int64_t)blockIdx.y * stride;   // stride is even: every pair starts 16-byte aligned
    const bool aligned = ((uintptr_t)s & 1) == 0;     // odd N: odd pairs start on an odd byte
    for (int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * 2; i < N; i += (int64_t)gridDim.x * kBlock * 2) {
        if (i + 1 < N) {
            uint8_t a, b;
            if (aligned) {
                const uint16_t v = *reinterpret_cast<const uint16_t *>(s + i);
                a = (uint8_t)(v & 0xff); b = (uint8_t)(v >> 8);
            } else {
                a = s[i]; b = s[i + 1];
            }
            double2 o;
            o.x = (double)a * (1.0 / 255.0);
            o.y = (double)b * (1.0 / 255.0);
            *reinterpret_cast<double2 *>(d + i) = o;
        } else {
            d[i] = (double)s[i] * (1.0 / 255.0);
        }
    }
}

// ---------------------------------------------------------------------------
// Gauss-Newton bookkeeping
// ---------------------------------------------------------------------------
struct LoopState {   // device arrays, one entry per pair
    double *pose;      // [n][12] last accepted pose
    double *cand;      // [n][12] pose being evaluated
    double *prev_err;  // [n]
    int *state;        // [n]
    int *n_evals;      // [n]
    int *active;       // [1] number of pairs still running
    int *ticket;       // [1] blocks of the current k_dvo_reduce launch that are through
    unsigned long long *evals;   // [2] at this level, summed over the pairs: [0] photometric errors evaluated
                                 //     (PhotometricError calls), [1] pose updates solved (calc_pose_update calls)
    int *warn;         // [n] an evaluation of the current estimate found an EMPTY update mask ("pose change is too large")
    int *mode;         // [n] what the pair's NEXT evaluation is: MODE_FULL0 / MODE_PROBE / MODE_FULLK
    int *tested;       // [n] candidates tested at this level (the reference's loop counter k)
    int *stat_state;   // [n] ST_RUNNING where the next evaluation needs robust statistics (running and not a probe)
    int *round;        // [2] pairs whose evaluation in the launch just reduced was full / a probe
    int *next;         // [2] pairs still running whose NEXT evaluation is full / a probe
    int fuse_first;    // the first candidate of this level is evaluated in full straight away (see reduce_pair)
    int *host_flag;    // mapped host memory: [0] `active` as left by the last launch, [2..3] `evals[0]` (64 bit),
                       // [4] / [5] `round`, [6..7] `evals[1]` (64 bit), [8] / [9] `next`
};

// Fixed-order reduction of the per-block partials of one pair; in loop mode the
// bookkeeping of _PoseChangeEstimator.__call__ (:92-111) follows.
// NG groups of 32 lanes: group g adds the partials g, g + NG, ... of each accumulator, then the groups are added
// in order.  8 groups for batches (a pair has ~32 partials there); 32 groups (1024 threads) for the few pairs
// of a drop-in call, whose 60 - 300 partials per pair come from every XCD: ten dependent rounds of loads
// instead of forty.
template <int NG>
__device__ __forceinline__ void reduce_pair(const double *__restrict__ partials, int nblk,
                                            double *__restrict__ results, LoopState ls, int loop_mode, int max_iter) {
    const int pair = blockIdx.x;
    __shared__ double red[NG][kAccPad];
    __shared__ double solve_ws[108];   // workspace of the rank-deficient 6x6 solve (lane 0)
    const int k = threadIdx.x & 31, g = threadIdx.x >> 5;
    double s = 0.0;
    if (k < kAcc)
        for (int b = g; b < nblk; b += NG) s += partials[((int64_t)pair * nblk + b) * kAccPad + k];
    red[g][k] = s;
    __syncthreads();
    if (threadIdx.x < kAccPad) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < NG; i++) t += red[i][threadIdx.x];
        red[0][threadIdx.x] = t;
        results[(int64_t)pair * kAccPad + threadIdx.x] = (threadIdx.x < kAcc) ? t : 0.0;
    }
    __syncthreads();
    if (!loop_mode || threadIdx.x != 0) return;

    // _PoseChangeEstimator.__call__ (:92-111) runs, per level, one PhotometricError at the prior pose
    // and then up to max_iter rounds of calc_pose_update + PhotometricError(candidate), stopping at the
    // first candidate whose error is larger: n updates, n + 1 errors -- the normal equations at the
    // rejected candidate are never formed.  Same here: the first evaluation of a level is FULL (error
    // + normal equations), a candidate is PROBED (error only, ~40 % of the arithmetic and a third of
    // the texels), and only an accepted candidate gets its normal equations (a second, full pass at
    // the same pose).  With the typical 0-1 accepted steps per level that is one full + one probe
    // instead of two full evaluations.  One exception: the FIRST candidate of the coarsest level
    // (fuse_first) starts from the caller's prior and is accepted almost always, so it is evaluated
    // in full straight away -- error and normal equations in one pass, as every evaluation was
    // before -- instead of probe + full.
    const double *R = red[0];
    const double err = R[27] / R[29];  // mean over the error mask; 0/0 = NaN like np.mean([])
    double *pose = ls.pose + 12 * pair, *cand = ls.cand + 12 * pair;
    const int mode = ls.mode[pair];    // what has just been evaluated
    atomicAdd(&ls.round[mode == MODE_PROBE ? 1 : 0], 1);
    bool finished = false, solve = false;
    if (mode == MODE_FULL0) {
        ls.n_evals[pair] += 1;
        atomicAdd(ls.evals, 1ull);
        ls.prev_err[pair] = err;
        if (max_iter == 0) finished = true;
        else solve = true;
    } else if (mode == MODE_PROBE || mode == MODE_FUSED) {
        ls.n_evals[pair] += 1;
        atomicAdd(ls.evals, 1ull);
        const int k = ls.tested[pair] + 1;
        ls.tested[pair] = k;
        if (err > ls.prev_err[pair]) {
            finished = true;  // candidate rejected: keep the last accepted pose (:105-106)
        } else {
            for (int i = 0; i < 12; i++) pose[i] = cand[i];  // accepted (:107-110)
            ls.prev_err[pair] = err;
            if (k == max_iter) finished = true;
            else if (mode == MODE_FUSED) solve = true;       // its normal equations are in R already
            else ls.mode[pair] = MODE_FULLK;                 // now its normal equations are needed
        }
    } else {   // MODE_FULLK: the normal equations at the accepted pose (cand == pose)
        solve = true;
    }
    if (solve) {
        if (R[28] == 0.0) {
            finished = true;  // empty update mask: "pose change is too large" (:98-100)
            ls.warn[pair] = 1;
        } else {
            double xi[6];
            atomicAdd(ls.evals + 1, 1ull);
            tdk::solve6<true>(R, R + 21, xi, R[28], solve_ws);   // R[28]: rows of J (update mask count)
            double next[12];
            tdk::compose_update(xi, pose, next);
            for (int i = 0; i < 12; i++) cand[i] = next[i];
            ls.mode[pair] = (mode == MODE_FULL0 && ls.fuse_first) ? MODE_FUSED : MODE_PROBE;
        }
    }
    ls.stat_state[pair] = (!finished && ls.mode[pair] != MODE_PROBE) ? ST_RUNNING : ST_DONE;
    if (!finished) atomicAdd(&ls.next[ls.mode[pair] == MODE_PROBE ? 1 : 0], 1);
    if (finished) {
        ls.state[pair] = ST_DONE;
        atomicSub(ls.active, 1);
    }
}

constexpr int kFlagGate = 12;          // host_flag[12]: the level being estimated, -1 when the call is complete
constexpr int kFlagLevelEvals = 16;    // host_flag[16 + 4 l ...]: evaluations / updates (64 bit each) of level l

// The speculative level chain of small batches (see k_chain_init below): what the last block of a reduce
// launch needs to close its level and open the next one.  state_all == nullptr: not a chain launch.
struct ChainArgs {
    int *state_all;   // [n_levels][n] per-level state arrays
    int *gate;        // [1] the level being estimated
    int from, n;      // the level this launch belongs to; pairs
    double *poses_out;
    int *warn_out;    // where the host reads the result after level 0
};

// closes level `from` and opens from - 1 (one thread; the batches this runs for have a handful of pairs);
// after level 0 the poses and warnings go where the host reads them
__device__ __forceinline__ void chain_next_level(const LoopState &ls, const ChainArgs &c) {
    const int from = c.from, n = c.n;
    for (int i = 0; i < n; i++) {
        if (from > 0) {
            for (int k = 0; k < 12; k++) ls.cand[12 * i + k] = ls.pose[12 * i + k];
            ls.prev_err[i] = 0.0;
            c.state_all[(from - 1) * n + i] = ST_RUNNING;
            ls.stat_state[i] = ST_RUNNING;
            ls.mode[i] = MODE_FULL0;
            ls.tested[i] = 0;
            ls.n_evals[i] = 0;
        } else {
            for (int k = 0; k < 12; k++) c.poses_out[12 * i + k] = ls.pose[12 * i + k];
            c.warn_out[i] = ls.warn[i];
        }
    }
    *reinterpret_cast<volatile unsigned long long *>(ls.host_flag + kFlagLevelEvals + 4 * from) = atomicAdd(ls.evals, 0ull);
    *reinterpret_cast<volatile unsigned long long *>(ls.host_flag + kFlagLevelEvals + 4 * from + 2) = atomicAdd(ls.evals + 1, 0ull);
    ls.evals[0] = 0ull; ls.evals[1] = 0ull;
    if (from > 0) *ls.active = n;
    *c.gate = from - 1;
    ls.host_flag[kFlagGate] = from - 1;
}

template <int NG>
__global__ __launch_bounds__(32 * NG) void k_dvo_reduce(const double *__restrict__ partials, int nblk,
                                                        double *__restrict__ results, LoopState ls,
                                                        int loop_mode, int max_iter, ChainArgs chain) {
    const int pair = blockIdx.x;
    if (!loop_mode || ls.state[pair] == ST_RUNNING) reduce_pair<NG>(partials, nblk, results, ls, loop_mode, max_iter);
    if (!loop_mode || threadIdx.x != 0) return;
    // the last block through publishes the number of running pairs to the host:
    // no copy kernel between the iterations, the host just waits for the stream
    __threadfence();
    if (atomicAdd(ls.ticket, 1) == (int)gridDim.x - 1) {
        *ls.ticket = 0;
        *reinterpret_cast<volatile unsigned long long *>(ls.host_flag + 2) = atomicAdd(ls.evals, 0ull);
        *reinterpret_cast<volatile unsigned long long *>(ls.host_flag + 6) = atomicAdd(ls.evals + 1, 0ull);
        ls.host_flag[4] = atomicExch(&ls.round[0], 0);
        ls.host_flag[5] = atomicExch(&ls.round[1], 0);
        ls.host_flag[8] = atomicExch(&ls.next[0], 0);
        ls.host_flag[9] = atomicExch(&ls.next[1], 0);
        const int running = atomicAdd(ls.active, 0);
        *ls.host_flag = running;
        // chain launches: the level is done and it is the current one -> on to the next (the evaluation
        // kernels queued behind this launch find their level's state array open)
        if (chain.state_all != nullptr && running == 0 && *chain.gate == chain.from) chain_next_level(ls, chain);
        __threadfence_system();
    }
}

__global__ void k_loop_init(LoopState ls, const double *poses_in, int n) {   // poses_in may be ls.pose itself
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int k = 0; k < 12; k++) {
        double v = poses_in[12 * i + k];
        ls.pose[12 * i + k] = v;
        ls.cand[12 * i + k] = v;
    }
    ls.prev_err[i] = 0.0;
    ls.state[i] = ST_RUNNING;
    ls.stat_state[i] = ST_RUNNING;
    ls.mode[i] = MODE_FULL0;
    ls.tested[i] = 0;
    ls.n_evals[i] = 0;
    if (i == 0) {
        *ls.active = n;
        *ls.ticket = 0;
        ls.evals[0] = 0ull;
        ls.evals[1] = 0ull;
        ls.round[0] = 0;
        ls.round[1] = 0;
        ls.next[0] = 0;
        ls.next[1] = 0;
    }
}

// final poses and warnings of a call, written where the host reads them (tdk_dvo::h_io)
__global__ void k_publish(LoopState ls, double *__restrict__ poses_out, int *__restrict__ warn_out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int k = 0; k < 12; k++) poses_out[12 * i + k] = ls.pose[12 * i + k];
    warn_out[i] = ls.warn[i];
}

// ---------------------------------------------------------------------------
// The speculative level chain of small batches (one pair: the drop-in PoseChangeEstimator call).
//
// A single pair's estimation is a chain of 10 - 30 us kernels; waiting on the host between the rounds
// of a level and between levels cost four ~19 us round trips of a 265 us call.  Here the host queues
// the whole coarse-to-fine chain at once -- per level the evaluations a level typically takes (3 on
// the coarsest, 2 on the others), then a transition kernel -- and waits once.  What keeps a kernel
// from running out of turn is the state array it is given: every level has its OWN state array,
// RUNNING only while that level is being estimated, so k_dvo_eval / k_dvo_reduce launches of a level
// whose turn has not come (the level above needed more rounds than were queued) return at once, as
// they already do for pairs that have finished.  The last block of a k_dvo_reduce launch moves on
// (chain_next_level) -- re-arming the loop state as k_loop_init does and opening the next level's state
// array -- only if its level is the current one (gate) and has no pair left running; otherwise everything
// stays as it is, the host reads the gate and queues more rounds for that level.  Evaluation kernels are
// untouched.
// ---------------------------------------------------------------------------
__global__ void k_chain_init(LoopState ls, int *state_all, int *gate, int n, int n_levels, const double *poses_in) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        for (int k = 0; k < 12; k++) {
            const double v = poses_in[12 * i + k];
            ls.pose[12 * i + k] = v;
            ls.cand[12 * i + k] = v;
        }
        ls.prev_err[i] = 0.0;
        for (int l = 0; l < n_levels; l++) state_all[l * n + i] = l == n_levels - 1 ? ST_RUNNING : ST_DONE;
        ls.stat_state[i] = ST_RUNNING;
        ls.mode[i] = MODE_FULL0;
        ls.tested[i] = 0;
        ls.n_evals[i] = 0;
        ls.warn[i] = 0;
    }
    if (threadIdx.x == 0) {
        *ls.active = n;
        *ls.ticket = 0;
        ls.evals[0] = 0ull; ls.evals[1] = 0ull;
        ls.round[0] = 0; ls.round[1] = 0;
        ls.next[0] = 0; ls.next[1] = 0;
        *gate = n_levels - 1;
        ls.host_flag[kFlagGate] = n_levels - 1;
    }
}

// ---------------------------------------------------------------------------
// Robust scale statistics per pair (Student-t, Tukey)
// ---------------------------------------------------------------------------
// 1 / a to the last bit or so: v_rcp_f64 and two Newton steps instead of the IEEE division sequence
__device__ __forceinline__ double fast_rcp(double a) {
    double x = __builtin_amdgcn_rcp(a);
    x = __builtin_fma(__builtin_fma(-a, x, 1.0), x, x);
    return __builtin_fma(__builtin_fma(-a, x, 1.0), x, x);
}

// masked residual map: rm[i] = I0 - I1 where the pixel is in the UPDATE mask of
// the pose (in range & z > 0), NaN elsewhere; count[pair] = mask size.
// STUDENT: the first step of the Student-t fixed point (variance 1: s (nu + 1) / (nu + s)) rides along -- its
// block partials go where k_robust_student_step leaves them -- so the residual map is read nine times, not ten.
constexpr int kTukeySample = 2048;      // sorted sample per pair (one block, LDS)
constexpr int kTukeyThreads = 512;     // 8 waves: a 16-wave block waited up to 0.6 ms for a CU with that much room beside the pyramid kernel
constexpr double kTukeySigmas = 5.0;
constexpr int kBandBuf = 1024;          // residuals a block of the collecting passes gathers in LDS before one global append

struct TukeyBracket {
    double lo, hi;                  // the median lies in [lo, hi]
    double med;                     // ... the exact median, once it is known
    double dlo, dhi;                // the MAD lies in [dlo, dhi]
    unsigned int below, n_med;      // residuals < lo; residuals collected from [lo, hi]
    unsigned int small, n_dev;      // deviations < dlo; deviations collected from [dlo, dhi]
    unsigned int overflow, n_sample;   // n_sample: top bit = the sample holds every pixel
};

// A block gathers the values it collects in LDS (`buf`, `buf_n`: wave-aggregated LDS atomics) and appends them
// to the pair's band with ONE global atomic (band_flush) -- a global atomic per wave and step put thousands of
// them on one address per pair and tripled the time of the pass.  Values beyond the buffer go out directly.
__device__ __forceinline__ void band_append(bool take, double v, double *buf, unsigned int *buf_n, unsigned int *g_count,
                                            double *band, unsigned int cap, unsigned int *overflow) {
    const uint64_t m = __builtin_amdgcn_ballot_w64(take);
    if (m == 0) return;
    const int lane = threadIdx.x & 63, leader = __builtin_ctzll(m);
    unsigned int base = 0;
    if (lane == leader) base = atomicAdd(buf_n, (unsigned int)__builtin_popcountll(m));
    base = (unsigned int)__builtin_amdgcn_readlane((int)base, leader);
    if (take) {
        const unsigned int slot = base + (unsigned int)__builtin_popcountll(m & ((1ull << lane) - 1ull));
        if (slot < (unsigned int)kBandBuf) buf[slot] = v;
        else {
            const unsigned int g = atomicAdd(g_count, 1u);
            if (g < cap) band[g] = v;
            else *overflow = 1u;
        }
    }
}

__device__ __forceinline__ void band_flush(const double *buf, const unsigned int *buf_n, unsigned int *shared_base,
                                           unsigned int *g_count, double *band, unsigned int cap, unsigned int *overflow) {
    __syncthreads();
    const unsigned int n = *buf_n < (unsigned int)kBandBuf ? *buf_n : (unsigned int)kBandBuf;
    if (threadIdx.x == 0) *shared_base = n ? atomicAdd(g_count, n) : 0u;
    __syncthreads();
    const unsigned int base = *shared_base;
    for (unsigned int i = threadIdx.x; i < n; i += blockDim.x) {
        if (base + i < cap) band[base + i] = buf[i];
        else *overflow = 1u;
    }
}

// Student-t, Taylor scheme (see k_student_taylor below): nine expansion points per pair, three sums per point
constexpr int kStudentPts = 9, kStudentSums = 3 * kStudentPts;
constexpr int kStudentRow = kStudentPts + 1;   // expansion points of a pair + its "one more pass" flag
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Pass A of the scheme in packed FP32: its only product is the next set of expansion points, for which 1e-5 is
// plenty.  The two residuals of a 16-byte load ride in the two halves of v_pk_add/mul/fma_f32 (two FP32 lanes per
// FP64 issue slot) and v_rcp_f32 needs no Newton steps.  s' is capped at 1e4 so that the product of nine factors
// stays inside FP32 (a = s' / (c' + s') is 0.9999 there; the few pixels beyond it shift the predicted points by
// less than 1e-4 of their share of the sum).  Rides in the mask pass (k_robust_mask<.., TAYLOR>): the residuals are
// in registers there and the pass waits for HBM.
struct StudentF32 {
    float c[kStudentPts];            // p_k / p_9 (block-uniform)
    f32x2 acc[kStudentSums];
    double inv_ref;
    __device__ __forceinline__ void init(const double *__restrict__ pts, int pair) {
        const double p_ref = pts[(size_t)pair * kStudentRow + kStudentPts - 1];
        inv_ref = 1.0 / (kStudentNu * p_ref);
        const double inv_p = 1.0 / p_ref;
#pragma unroll
        for (int k = 0; k < kStudentPts; k++) c[k] = (float)(pts[(size_t)pair * kStudentRow + k] * inv_p);
#pragma unroll
        for (int j = 0; j < kStudentSums; j++) acc[j] = f32x2{0.0f, 0.0f};
    }
    // s' in FP64 (r^2 may leave the FP32 range), then capped; NaN (outside the mask): adds nothing
    __device__ __forceinline__ float scaled_square(double x) const {
        const double sd = (x * x) * inv_ref;
        return x == x ? (float)(sd < 1e4 ? sd : 1e4) : 0.0f;
    }
    __device__ __forceinline__ void add(double x0, double x1) {
        const f32x2 s_ = {scaled_square(x0), scaled_square(x1)};
        f32x2 u[kStudentPts], q[kStudentPts];
#pragma unroll
        for (int k = 0; k < kStudentPts; k++) u[k] = s_ + c[k];
        q[0] = u[0];
#pragma unroll
        for (int k = 1; k < kStudentPts; k++) q[k] = q[k - 1] * u[k];
        f32x2 inv = {__builtin_amdgcn_rcpf(q[kStudentPts - 1].x), __builtin_amdgcn_rcpf(q[kStudentPts - 1].y)};
#pragma unroll
        for (int k = kStudentPts - 1; k >= 0; k--) {
            const f32x2 t = k > 0 ? inv * q[k > 0 ? k - 1 : 0] : inv;
            if (k > 0) inv *= u[k];
            const f32x2 a = s_ * t, at = a * t;
            acc[3 * k + 0] += a;
            acc[3 * k + 1] = __builtin_elementwise_fma(a, a, acc[3 * k + 1]);
            acc[3 * k + 2] = __builtin_elementwise_fma(a, at, acc[3 * k + 2]);
        }
    }
    // block partials, [kStudentSums] doubles at `out` (red: [kWaves][kStudentSums] in LDS)
    __device__ __forceinline__ void store(double (*red)[kStudentSums], double *__restrict__ out) {
#pragma unroll
        for (int j = 0; j < kStudentSums; j++) {
            double v = (double)acc[j].x + (double)acc[j].y;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][j] = v;
        }
        __syncthreads();
        if (threadIdx.x < kStudentSums)
            out[threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    }
};

struct TukeyArgs {            // TUKEY: brackets of every pair and the band the pass collects into (see k_tukey_sample)
    TukeyBracket *brackets;
    double *med_bands;
    unsigned int cap;         // doubles per pair in a band
};

// TAYLOR (with STUDENT): pass A of the Taylor scheme around `pts` rides along (StudentF32); its block partials go
// to partial_a[pair][block][kStudentSums].
template <bool STUDENT, bool FAST, bool TUKEY, bool TAYLOR>
__global__ __launch_bounds__(kBlock) void k_robust_mask(LevelPtrs L, const PairParams *__restrict__ params,
                                                        const double *__restrict__ poses,
                                                        const int *__restrict__ state, double scale,
                                                        double *__restrict__ rm, int *__restrict__ count,
                                                        double *__restrict__ partial, TukeyArgs tka,
                                                        const double *__restrict__ pts, double *__restrict__ partial_a) {
    const int pair = blockIdx.y;
    if (state != nullptr && state[pair] != ST_RUNNING) return;
    BlockSetup b;
    load_setup(b, params, poses, pair, scale);
    const int64_t base = (int64_t)pair * L.stride;
    const int W = L.W, H = L.H, N = (int)L.N;
    const double *__restrict__ tab = L.tab + (size_t)pair * (W + H);   // (x - ox) / fx | (y - oy) / fy
    const double *__restrict__ D0 = L.D0 + base, *__restrict__ I0 = L.I0 + base, *__restrict__ I1 = L.I1 + base;
    double *__restrict__ out = rm + base;
    int local = 0;
    double acc = 0.0;
    __shared__ double band_buf[TUKEY ? kBandBuf : 1];
    __shared__ unsigned int band_n, band_base;
    TukeyBracket tb;
    unsigned int tk_below = 0;
    double *med_band = nullptr;
    TukeyBracket *tg = nullptr;
    StudentF32 ta;
    if (TAYLOR) ta.init(pts, pair);
    if (TUKEY) {
        tb = tka.brackets[pair];
        tg = tka.brackets + pair;
        med_band = tka.med_bands + (size_t)pair * tka.cap;
        if (threadIdx.x == 0) band_n = 0;
        __syncthreads();
    }
    auto term = [&](double r, bool in) {
        if (TUKEY) {        // residuals below the median's bracket are counted, those inside collected
            const bool lt = in && r < tb.lo;
            tk_below += lt ? 1u : 0u;
            band_append(in && !lt && r <= tb.hi, r, band_buf, &band_n, &tg->n_med, med_band, tka.cap, &tg->overflow);
        }
        if (STUDENT && in) {
            const double sq = r * r;
            if (FAST) acc += sq * ((kStudentNu + 1.0) * fast_rcp(kStudentNu + sq));
            else acc += sq * ((kStudentNu + 1.0) / (kStudentNu + sq / 1.0));
        }
    };
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    // two consecutive pixels per lane and step: 16-byte loads and stores.  (x, y) of the first one
    // advanced incrementally: one division per thread, not per pixel
    const int step = 2 * gridDim.x * kBlock;
    int i = 2 * (blockIdx.x * kBlock + threadIdx.x);
    int y = i / W, x = i - y * W;
    const int step_y = step / W, step_x = step - step_y * W;
    for (; i + 1 < N; i += step) {
        // pure streams, each byte used once: non-temporal (1.51 -> 1.45 ms per full-resolution Tukey evaluation)
        const double2_u d = __builtin_nontemporal_load(reinterpret_cast<const double2_u *>(D0 + i));
        const double2_u a = __builtin_nontemporal_load(reinterpret_cast<const double2_u *>(I0 + i));
        const double2_u c = __builtin_nontemporal_load(reinterpret_cast<const double2_u *>(I1 + i));
        const bool wrap = x + 1 == W;                   // the second pixel starts the next row
        Pixel p, q;
        sp_warp(p, true, tab[x], tab[W + y], d.x, H, W, b.P, b.c);
        sp_warp(q, true, tab[wrap ? 0 : x + 1], tab[W + (wrap ? y + 1 : y)], d.y, H, W, b.P, b.c);
        const bool in0 = p.mask == 2, in1 = q.mask == 2;
        double2_u r;
        r.x = in0 ? a.x - c.x : nan;
        r.y = in1 ? a.y - c.y : nan;
        __builtin_nontemporal_store(r, reinterpret_cast<double2_u *>(out + i));     // read next by another kernel, from HBM
        term(r.x, in0);
        term(r.y, in1);
        if (TAYLOR) ta.add(r.x, r.y);
        local += (in0 ? 1 : 0) + (in1 ? 1 : 0);
        x += step_x;
        y += step_y;
        if (x >= W) { x -= W; y += 1; }
    }
    {                                                   // odd N: the last pixel (one lane of the grid has i == N - 1)
        const bool tail = i < N;                        // the wave votes together (the Tukey bands are appended by ballot)
        bool in = false;
        double r = 0.0;
        if (tail) {
            Pixel p;
            sp_warp(p, true, tab[x], tab[W + y], D0[i], H, W, b.P, b.c);
            in = p.mask == 2;
            r = I0[i] - I1[i];
            out[i] = in ? r : nan;
            local += in ? 1 : 0;
            if (TAYLOR) ta.add(in ? r : nan, nan);
        }
        if (TUKEY || tail) term(r, in);
    }
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(&count[pair], local);
    if (TUKEY) {
        for (int off = 32; off > 0; off >>= 1) tk_below += __shfl_down(tk_below, off, 64);
        if ((threadIdx.x & 63) == 0 && tk_below) atomicAdd(&tg->below, tk_below);
        band_flush(band_buf, &band_n, &band_base, &tg->n_med, med_band, tka.cap, &tg->overflow);
    }
    if (STUDENT) {
        __shared__ double red[kWaves];
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) partial[(int64_t)pair * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
    if (TAYLOR) {
        __shared__ double red_a[kWaves][kStudentSums];
        ta.store(red_a, partial_a + ((int64_t)pair * gridDim.x + blockIdx.x) * kStudentSums);
    }
}

// one fixed-point step of compute_weights_student_t (weights.py:13-16): the sum of
// s (nu + 1) / (nu + s / variance) over the masked residuals, s = r^2.  Two residuals per lane
// and load (16 bytes).  FAST: s / variance as a product with the pair's 1 / variance and the
// quotient through fast_rcp -- 9 instead of ~35 FP64 operations per residual; the terms differ
// from the IEEE quotients in the last bit, the variance by ~1e-16 relative (the pose by less
// than 1e-15).  TDK_STUDENT_EXACT=1 selects the IEEE divisions of the CPU restatement.
template <bool FAST>
__global__ __launch_bounds__(kBlock) void k_robust_student_step(const double *__restrict__ rm, int64_t stride,
                                                                int N, const int *__restrict__ state,
                                                                const double *__restrict__ variance,
                                                                double *__restrict__ partial) {
    const int pair = blockIdx.y;
    if (state != nullptr && state[pair] != ST_RUNNING) return;
    const double var = variance[pair];
    const double rvar = 1.0 / var;
    const double *r = rm + (int64_t)pair * stride;
    double acc = 0.0;
    auto term = [&](double v) {
        if (v == v) {
            double s = v * v;
            if (FAST) acc += s * ((kStudentNu + 1.0) * fast_rcp(__builtin_fma(s, rvar, kStudentNu)));
            else acc += s * ((kStudentNu + 1.0) / (kStudentNu + s / var));
        }
    };
    const int N2 = N >> 1;
#pragma unroll 4
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < N2; i += gridDim.x * kBlock) {
        const double2_u v = __builtin_nontemporal_load(reinterpret_cast<const double2_u *>(r + 2 * (int64_t)i));
        term(v.x);
        term(v.y);
    }
    if ((N & 1) && blockIdx.x == 0 && threadIdx.x == 0) term(r[N - 1]);
    __shared__ double red[kWaves];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(int64_t)pair * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// variance <- (sum of the block partials) / count: one wave per pair, lane b holds
// partial b, fixed shuffle tree (bit-reproducible); `first`: variance 1 (weights.py:10)
__global__ __launch_bounds__(64) void k_robust_student_update(const double *__restrict__ partial, int nblk,
                                                              const int *__restrict__ count,
                                                              const int *__restrict__ state,
                                                              double *__restrict__ variance, int n_pairs,
                                                              int first) {
    const int pair = blockIdx.x;
    if (state != nullptr && state[pair] != ST_RUNNING) return;
    if (first) {
        if (threadIdx.x == 0) variance[pair] = 1.0;
        return;
    }
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64) s += partial[(int64_t)pair * nblk + b];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (threadIdx.x == 0) variance[pair] = s / (double)count[pair];
}

// MSD radix select on the order-preserving 64-bit image of a double, per pair
__device__ __forceinline__ uint64_t ordered_key(double v) {
    uint64_t b = (uint64_t)__double_as_longlong(v);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}

__device__ __forceinline__ double key_to_double(uint64_t k) {
    uint64_t b = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

// The 64-bit key is consumed in five digits, most significant first: four of 13
// bits and a last one of 12 (8192-bin histograms: 32 KiB of LDS per block).
constexpr int kSelectPasses = 5, kSelectBins = 8192;
__host__ __device__ constexpr int select_shift(int pass) { return pass < 4 ? 51 - 13 * pass : 0; }
__host__ __device__ constexpr int select_bits(int pass) { return pass < 4 ? 13 : 12; }

// Where a pair's values come from: the residual map (the default), or -- Tukey's brackets -- the few per cent of
// them a pass has collected around the wanted order statistic, whose rank then counts from `rank_offset`.
struct SelectSrc {
    const double *data;
    int n, mode;                   // mode 1: |value - center|
    double center;
    unsigned int rank_offset, pad;
    double lo, hi;                 // a collected band: the bracket its values lie in (k_band_median bins by value);
                                   // lo > hi: unknown (the whole residual map)
};

struct SelectState {
    uint64_t prefix, rank;
    uint64_t next_key;             // successor pass: smallest key above the selected one
    unsigned int count_le, even;   // elements <= the selected one; the mask size is even
    // short cut after two digits (26 bits): the keys that still match are few, they are
    // collected and the selection finishes on them (k_select_collect / k_select_finish)
    uint64_t min_above;            // smallest key whose 26-bit prefix is larger than the selected one
    unsigned int group;            // number of keys that share the 26-bit prefix
    unsigned int n_cand;           // keys collected so far
    unsigned int done, rank0;      // the median of this pair is final; rank of the lower middle order statistic in the source
};
constexpr int kSelectCap = 2048;   // candidates kept per pair; a larger group takes all five digit passes

// rank of the LOWER middle order statistic; np.median averages it with its
// successor when the count is even
__global__ void k_select_init(SelectState *st, const int *__restrict__ count, const SelectSrc *__restrict__ src,
                              int n_pairs) {
    int pair = blockIdx.x * blockDim.x + threadIdx.x;
    if (pair >= n_pairs) return;
    int m = count[pair];
    st[pair].prefix = 0;
    const unsigned int off = src ? src[pair].rank_offset : 0u;
    st[pair].rank = (uint64_t)(m > 0 ? (unsigned int)((m - 1) / 2) - off : 0u);
    st[pair].rank0 = (unsigned int)st[pair].rank;
    st[pair].next_key = ~0ull;
    st[pair].count_le = 0;
    st[pair].even = (m > 0 && (m & 1) == 0) ? 1u : 0u;
    st[pair].min_above = ~0ull;
    st[pair].group = 0;
    st[pair].n_cand = 0;
    st[pair].done = 0;
}

__device__ __forceinline__ void select_source(const SelectSrc *__restrict__ src, int pair, const double *__restrict__ rm,
                                              int64_t stride, const double *__restrict__ center, const double *&r, int &N,
                                              int &mode, double &cen) {
    if (src) {
        r = src[pair].data; N = src[pair].n; mode = src[pair].mode; cen = src[pair].center;
    } else {
        r = rm + (int64_t)pair * stride;
        cen = mode ? center[pair] : 0.0;
    }
}

// values: rm (mode 0) or |rm - center[pair]| (mode 1, for the MAD)
__device__ __forceinline__ bool select_key(const double *r, int i, int mode, double cen, uint64_t &k) {
    double v = r[i];
    if (v != v) return false;
    k = ordered_key(mode ? fabs(v - cen) : v);
    return true;
}

// One digit: histogram of the keys that match the prefix found so far, in LDS
// per block, then merged into the pair's global histogram.  A wave whose active
// lanes all hit one bin (exact ties: a noise-free scene has thousands of equal
// residuals) adds its count once instead of serialising 64 atomics on one address.
__global__ __launch_bounds__(kBlock) void k_select_hist(const double *__restrict__ rm, int64_t stride, int N,
                                                        const int *__restrict__ state, int mode,
                                                        const double *__restrict__ center,
                                                        const SelectSrc *__restrict__ src,
                                                        const SelectState *__restrict__ st, int pass,
                                                        unsigned int *__restrict__ hist) {
    const int pair = blockIdx.y;
    if (state != nullptr && state[pair] != ST_RUNNING) return;
    if (st[pair].done) return;
    __shared__ unsigned int h[kSelectBins];
    for (int i = threadIdx.x; i < kSelectBins; i += kBlock) h[i] = 0;
    __syncthreads();
    const int shift = select_shift(pass);
    const unsigned int digit_mask = (1u << select_bits(pass)) - 1u;
    const uint64_t prefix = st[pair].prefix;
    const uint64_t mask = pass == 0 ? 0ull : (~0ull << select_shift(pass - 1));
    const double *r;
    double cen;
    select_source(src, pair, rm, stride, center, r, N, mode, cen);
    const int lane = threadIdx.x & 63;
    for (int i0 = blockIdx.x * kBlock; i0 < N; i0 += gridDim.x * kBlock) {
        const int i = i0 + (int)threadIdx.x;
        uint64_t k = 0;
        const bool hit = i < N && select_key(r, i, mode, cen, k) && (k & mask) == prefix;
        const unsigned int bin = (unsigned int)(k >> shift) & digit_mask;
        const uint64_t hits = __builtin_amdgcn_ballot_w64(hit);
        if (hits == 0) continue;
        const int leader = __builtin_ctzll(hits);
        const unsigned int lb = (unsigned int)__builtin_amdgcn_readlane((int)bin, leader);
        if (__builtin_amdgcn_ballot_w64(hit && bin == lb) == hits) {
            if (lane == leader) atomicAdd(&h[lb], (unsigned int)__builtin_popcountll(hits));
        } else if (hit) {
            atomicAdd(&h[bin], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kSelectBins; i += kBlock)
        if (h[i]) atomicAdd(&hist[(size_t)pair * kSelectBins + i], h[i]);
}

// one block per pair: find the bin that holds the rank, extend the prefix, clear the histogram
__global__ __launch_bounds__(kBlock) void k_select_pick(unsigned int *__restrict__ hist,
                                                        SelectState *__restrict__ st,
                                                        const int *__restrict__ state, int pass) {
    const int pair = blockIdx.x;
    if (state != nullptr && state[pair] != ST_RUNNING) return;
    if (st[pair].done) return;
    unsigned int *h = hist + (size_t)pair * kSelectBins;
    constexpr int kPer = kSelectBins / kBlock;   // consecutive bins per thread
    __shared__ unsigned int sums[kBlock];
    __shared__ unsigned long long rank_in_owner;
    __shared__ int owner;
    unsigned int local[kPer], total = 0;
#pragma unroll
    for (int j = 0; j < kPer; j++) {
        local[j] = h[threadIdx.x * kPer + j];
        h[threadIdx.x * kPer + j] = 0;
        total += local[j];
    }
    sums[threadIdx.x] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t rank = st[pair].rank, cum = 0;
        int t = 0;
        for (; t < kBlock - 1; t++) {
            if (rank < cum + sums[t]) break;
            cum += sums[t];
        }
        owner = t;
        rank_in_owner = rank - cum;
    }
    __syncthreads();
    if ((int)threadIdx.x != owner) return;
    uint64_t rank = rank_in_owner, cum = 0;
    int bin = kPer - 1;
    for (int j = 0; j < kPer; j++) {
        if (rank < cum + local[j]) { bin = j; break; }
        cum += local[j];
    }
    st[pair].rank = rank - cum;
    st[pair].prefix |= ((uint64_t)(threadIdx.x * kPer + bin)) << select_shift(pass);
    st[pair].group = local[bin];
}

// After two digits: gather the keys that share the 26-bit prefix (at most
// kSelectCap of them) and the smallest key above the group.
__global__ __launch_bounds__(kBlock) void k_select_collect(const double *__restrict__ rm, int64_t stride, int N,
                                                           const int *__restrict__ state, int mode,
                                                           const double *__restrict__ center,
                                                           const SelectSrc *__restrict__ src,
                                                           SelectState *__restrict__ st,
                                                           uint64_t *__restrict__ cand, int pass) {
    const int pair = blockIdx.y;
    if (state != nullptr && state[pair] != ST_RUNNING) return;
    if (st[pair].done || st[pair].group > (unsigned int)kSelectCap) return;    // the digit passes go on instead
    const int shift = select_shift(pass);
    const uint64_t group_prefix = st[pair].prefix >> shift;
    const double *r;
    double cen;
    select_source(src, pair, rm, stride, center, r, N, mode, cen);
    uint64_t *mine = cand + (size_t)pair * kSelectCap;
    uint64_t above = ~0ull;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += gridDim.x * kBlock) {
        uint64_t k;
        if (!select_key(r, i, mode, cen, k)) continue;
        const uint64_t top = k >> shift;
        if (top == group_prefix) {
            const unsigned int slot = atomicAdd(&st[pair].n_cand, 1u);
            if (slot < (unsigned int)kSelectCap) mine[slot] = k;
        } else if (top > group_prefix && k < above) {
            above = k;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const uint64_t o = __shfl_down(above, off, 64);
        above = o < above ? o : above;
    }
    if ((threadIdx.x & 63) == 0 && above != ~0ull)
        atomicMin((unsigned long long *)&st[pair].min_above, (unsigned long long)above);
}

// One block per pair: rank the collected keys (position = keys smaller, ties by
// slot) and read off the lower middle order statistic and, for an even count,
// its successor -- inside the group, or the smallest key above it.
__global__ __launch_bounds__(kBlock) void k_select_finish(SelectState *__restrict__ st,
                                                          const uint64_t *__restrict__ cand,
                                                          const int *__restrict__ state, double factor,
                                                          double *__restrict__ out) {
    const int pair = blockIdx.x;
    if (state != nullptr && state[pair] != ST_RUNNING) return;
    const unsigned int c = st[pair].group;
    if (st[pair].done || c == 0 || c > (unsigned int)kSelectCap) return;
    __shared__ uint64_t keys[kSelectCap];
    __shared__ uint64_t lo_s, hi_s;
    const uint64_t *mine = cand + (size_t)pair * kSelectCap;
    for (unsigned int i = threadIdx.x; i < c; i += kBlock) keys[i] = mine[i];
    if (threadIdx.x == 0) { lo_s = ~0ull; hi_s = ~0ull; }
    __syncthreads();
    const unsigned int rank = (unsigned int)st[pair].rank;
    for (unsigned int i = threadIdx.x; i < c; i += kBlock) {
        const uint64_t k = keys[i];
        unsigned int pos = 0;
        for (unsigned int j = 0; j < c; j++) {
            const uint64_t o = keys[j];
            pos += (o < k || (o == k && j < i)) ? 1u : 0u;
        }
        if (pos == rank) lo_s = k;
        if (pos == rank + 1) hi_s = k;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const double lo = key_to_double(lo_s);
    double hi = lo;
    if (st[pair].even) hi = key_to_double(rank + 1 < c ? hi_s : st[pair].min_above);
    out[pair] = factor * ((lo + hi) / 2.0);
    st[pair].done = 1;
}

// successor of the selected key: number of keys <= it and the smallest key above it
__global__ __launch_bounds__(kBlock) void k_select_successor(const double *__restrict__ rm, int64_t stride, int N,
                                                             const int *__restrict__ state, int mode,
                                                             const double *__restrict__ center,
                                                             const SelectSrc *__restrict__ src,
                                                             SelectState *__restrict__ st) {
    const int pair = blockIdx.y;
    if (state != nullptr && state[pair] != ST_RUNNING) return;
    if (st[pair].done || !st[pair].even) return;     // odd count: the median is the selected key itself
    const uint64_t sel = st[pair].prefix;
    const double *r;
    double cen;
    select_source(src, pair, rm, stride, center, r, N, mode, cen);
    unsigned int le = 0;
    uint64_t next = ~0ull;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += gridDim.x * kBlock) {
        uint64_t k;
        if (!select_key(r, i, mode, cen, k)) continue;
        if (k <= sel) le++;
        else if (k < next) next = k;
    }
    for (int off = 32; off > 0; off >>= 1) {
        le += __shfl_down(le, off, 64);
        const uint64_t o = __shfl_down(next, off, 64);
        next = o < next ? o : next;
    }
    if ((threadIdx.x & 63) == 0) {
        if (le) atomicAdd(&st[pair].count_le, le);
        if (next != ~0ull) atomicMin((unsigned long long *)&st[pair].next_key, (unsigned long long)next);
    }
}

// np.median: the selected (lower middle) order statistic, averaged with its
// successor in sorted order when the count is even -- the successor is the same
// value if more than rank + 1 elements are <= it.  Then the optional scaling.
__global__ void k_median_combine(const SelectState *__restrict__ st, const int *__restrict__ count,
                                 const int *__restrict__ state, int n_pairs, double factor, double *out) {
    int pair = blockIdx.x * blockDim.x + threadIdx.x;
    if (pair >= n_pairs) return;
    if (state != nullptr && state[pair] != ST_RUNNING) return;
    if (st[pair].done) return;
    const double lo = key_to_double(st[pair].prefix);
    double hi = lo;
    if (st[pair].even) {
        const unsigned int lower_rank = st[pair].rank0;     // in the source the keys were counted in
        if (st[pair].count_le < lower_rank + 2) hi = key_to_double(st[pair].next_key);
    }
    out[pair] = factor * ((lo + hi) / 2.0);
}

// ---------------------------------------------------------------------------
// Tukey's two medians (median of the masked residuals, median of their absolute deviations from it,
// tadataka/robust/weights.py:21-35): the digit-by-digit selection above, run on a few per cent of the data.
//
// On the whole residual map the radix select costs 2 x (2-3 histogram passes + a collect pass) per evaluation.
// A sorted SAMPLE of the pair's masked residuals (k_tukey_sample: 2048 pixels spread over the frame, evaluated
// and sorted by one block) brackets the median: [lo, hi] holds it with the rank of +-5 standard deviations of a
// sample quantile to spare, ~11 % of the residuals.  The mask pass, which produces every residual anyway,
// counts those below lo and collects those in [lo, hi] (k_robust_mask<.., TUKEY>); the selection kernels then
// run on the collected band with the rank counted from `below` (SelectSrc) -- the same kernels, the same
// tie / even-count handling, a tenth of the data.  The MAD repeats this with the sample's deviations from the
// now exact median (k_tukey_dev_bracket) and one 8-byte-per-pixel pass over the residual map
// (k_tukey_deviations).  A pair whose order statistic turns out to lie outside its bracket (a 1e-6 event by
// construction) or whose band overflows (huge tie groups) gets the whole residual map as its source instead
// (k_tukey_plan): slower, same kernels, exact.  Either way the result is the very double the plain radix path
// returns, so the weights are bit-identical.
// (Two variants were built first and lost: a single pass that brackets the MAD for ANY median in [lo, hi] --
// the median's uncertainty widens the MAD band to ~23 % of the residuals, every band overflowed; and finishing
// by one block per pair that sorts 16384-element bands in LDS -- 200-300 us per kernel, slower than the radix
// path it was to replace.)
// ---------------------------------------------------------------------------

// bitonic sort of keys[0 .. n) (n a power of two) in LDS by the whole block, ascending
__device__ __forceinline__ void block_bitonic_sort(uint64_t *keys, int n) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (n >> 1); t += blockDim.x) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), p = i | j;
                const uint64_t a = keys[i], b = keys[p];
                const bool up = (i & k) == 0;
                if ((a > b) == up) { keys[i] = b; keys[p] = a; }
            }
            __syncthreads();
        }
    }
}

__device__ __forceinline__ int next_pow2(int v) {
    int n = 1;
    while (n < v) n <<= 1;
    return n;
}

// Sample position j of m in a map of N pixels: stratified (one per stretch of N / m pixels) and jittered inside its
// stretch by a hash of (j, pair).  On a regular grid -- every 29.5th pixel of a 213x284 level -- the sample aliased
// with periodic texture: for two pairs of the bench the MAD bracket of the sample missed the level's MAD by far more
// than its 5-sigma rank margin, every evaluation, and they took the exact fallback (0.09 ms each).  m == N: all pixels.
__device__ __forceinline__ int sample_position(int j, int m, int N, int pair) {
    if (m >= N) return j;
    unsigned int x = (unsigned int)j * 2654435761u + (unsigned int)pair * 40503u + 0x9e3779b9u;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    const double u = (double)x * (1.0 / 4294967296.0);
    const int i = (int)(((double)j + u) * ((double)N / (double)m));
    return i < N ? i : N - 1;
}

// masked residual of pixel i at the pair's pose (the arithmetic of k_robust_mask)
__device__ __forceinline__ bool masked_residual(const LevelPtrs &L, const BlockSetup &b, const double *__restrict__ tab,
                                                int64_t base, int i, double &r) {
    const int W = L.W, H = L.H;
    const int y = i / W, x = i - y * W;
    Pixel p;
    sp_warp(p, true, tab[x], tab[W + y], L.D0[base + i], H, W, b.P, b.c);
    r = L.I0[base + i] - L.I1[base + i];
    return p.mask == 2;
}

// ranks r_lo < r_hi of a sample of `ms` values that bracket its lower / upper middle order statistic by
// kTukeySigmas standard deviations of a sample quantile's rank (sqrt(ms) / 2), one more for good measure
__device__ __forceinline__ void bracket_ranks(int ms, bool exhaustive, int &r_lo, int &r_hi) {
    const double g = exhaustive ? 0.0 : kTukeySigmas * 0.5 * sqrt((double)ms) + 1.0;
    r_lo = (int)floor((double)((ms - 1) / 2) - g);
    r_hi = (int)ceil((double)(ms / 2) + g);
}

__global__ __launch_bounds__(kTukeyThreads) void k_tukey_sample(LevelPtrs L, const PairParams *__restrict__ params,
                                                                 const double *__restrict__ poses,
                                                                 const int *__restrict__ state, double scale,
                                                                 TukeyBracket *__restrict__ out,
                                                                 double *__restrict__ samples) {
    __shared__ uint64_t keys[kTukeySample];
    __shared__ int s_valid;
    const int pair = blockIdx.x;
    if (state != nullptr && state[pair] != ST_RUNNING) return;
    BlockSetup b;
    load_setup(b, params, poses, pair, scale);
    const int64_t base = (int64_t)pair * L.stride;
    const int N = (int)L.N;
    const double *__restrict__ tab = L.tab + (size_t)pair * (L.W + L.H);
    const int m = N < kTukeySample ? N : kTukeySample;        // samples (all pixels of a small level)
    const int n_sort = next_pow2(m);
    if (threadIdx.x == 0) s_valid = 0;
    __syncthreads();
    int valid = 0;
    for (int j = threadIdx.x; j < n_sort; j += blockDim.x) {
        uint64_t key = ~0ull;                                  // outside the mask / padding: sorts last
        if (j < m) {
            const int i = sample_position(j, m, N, pair);      // spread over the whole frame
            double r;
            if (masked_residual(L, b, tab, base, i, r)) { key = ordered_key(r); valid++; }
        }
        keys[j] = key;
    }
    for (int off = 32; off > 0; off >>= 1) valid += __shfl_down(valid, off, 64);
    if ((threadIdx.x & 63) == 0 && valid) atomicAdd(&s_valid, valid);
    __syncthreads();
    const int ms = s_valid;
    block_bitonic_sort(keys, n_sort);
    double *__restrict__ mine = samples + (size_t)pair * kTukeySample;
    for (int j = threadIdx.x; j < ms; j += blockDim.x) mine[j] = key_to_double(keys[j]);   // for k_tukey_dev_bracket
    if (threadIdx.x != 0) return;
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    int r_lo, r_hi;
    bracket_ranks(ms, m == N, r_lo, r_hi);
    TukeyBracket t;
    t.lo = (ms > 0 && r_lo >= 0) ? key_to_double(keys[r_lo]) : -inf;
    t.hi = (ms > 0 && r_hi < ms) ? key_to_double(keys[r_hi]) : inf;
    t.med = 0.0; t.dlo = 0.0; t.dhi = inf;
    t.below = 0; t.n_med = 0; t.small = 0; t.n_dev = 0; t.overflow = 0;
    t.n_sample = (unsigned int)ms | (m == N ? 0x80000000u : 0u);
    out[pair] = t;
}

// Source of the selection that follows a collecting pass: the collected band if the wanted order statistics lie
// inside it, the whole residual map otherwise.  which = 0: median (band of residuals, ranks from `below`);
// 1: MAD (band of deviations, ranks from `small`; the map itself is read as |r - median|).
__global__ void k_tukey_plan(const TukeyBracket *__restrict__ tk, const int *__restrict__ count,
                             const int *__restrict__ state, int which, const double *__restrict__ rm, int64_t stride,
                             int N, const double *__restrict__ bands, unsigned int cap, const double *__restrict__ median,
                             int force_fallback, SelectSrc *__restrict__ src, unsigned int *__restrict__ n_fallback,
                             int n_pairs) {
    const int pair = blockIdx.x * blockDim.x + threadIdx.x;
    if (pair >= n_pairs) return;
    if (state != nullptr && state[pair] != ST_RUNNING) return;
    const TukeyBracket t = tk[pair];
    const int n = count[pair];
    const uint64_t k1 = n > 0 ? (uint64_t)(n - 1) / 2 : 0, k2 = n > 0 ? (uint64_t)n / 2 : 0;
    const unsigned int off = which ? t.small : t.below, c = which ? t.n_dev : t.n_med;
    const bool ok = !force_fallback && n > 0 && !t.overflow && c <= cap && k1 >= off && k2 < (uint64_t)off + c;
    SelectSrc s;
    if (ok) {
        s.data = bands + (size_t)pair * cap; s.n = (int)c; s.mode = 0; s.center = 0.0; s.rank_offset = off;
        s.lo = which ? t.dlo : t.lo; s.hi = which ? t.dhi : t.hi;
    } else {
        s.lo = 1.0; s.hi = 0.0;
        s.data = rm + (int64_t)pair * stride; s.n = N; s.mode = which; s.center = which ? median[pair] : 0.0;
        s.rank_offset = 0;
        if (n > 0 && n_fallback) atomicAdd(n_fallback, 1u);
    }
    s.pad = 0;
    src[pair] = s;
}

// k-th smallest (0-based) value of a pair's source: 8-bit MSD radix select by ONE block, histogram in LDS.
// The source is a collected band (a few ten thousand values: ~10 us) or, for a pair whose bracket failed, the
// whole residual map (slower, rare).  Values crowd into one or two bins of the leading digits: a wave whose
// lanes all hit one bin adds its count once instead of serialising 64 LDS atomics on one address.
__device__ uint64_t block_radix_select(const double *__restrict__ r, int N, int mode, double cen, uint64_t rank,
                                       unsigned int *hist, uint64_t *sh) {
    __shared__ unsigned int wave_tot[4];
    uint64_t prefix = 0;
    const int lane = threadIdx.x & 63;
    for (int pass = 0; pass < 8; pass++) {
        const int shift = 56 - 8 * pass;
        for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
        __syncthreads();
        // four independent loads per thread and step: the band comes from L2, one load at a time is all latency
        for (int i0 = 0; i0 < N; i0 += 4 * blockDim.x) {
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = i0 + u * (int)blockDim.x + (int)threadIdx.x;
                v[u] = i < N ? r[i] : __longlong_as_double(0x7ff8000000000000ll);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const bool valid = v[u] == v[u];
                const uint64_t k = ordered_key(mode ? fabs(v[u] - cen) : v[u]);
                const bool hit = valid && (pass == 0 || (k >> (shift + 8)) == (prefix >> (shift + 8)));
                const unsigned int bin = (unsigned int)(k >> shift) & 255u;
                const uint64_t hits = __builtin_amdgcn_ballot_w64(hit);
                if (hits == 0) continue;
                const int leader = __builtin_ctzll(hits);
                const unsigned int lb = (unsigned int)__builtin_amdgcn_readlane((int)bin, leader);
                if (__builtin_amdgcn_ballot_w64(hit && bin == lb) == hits) {
                    if (lane == leader) atomicAdd(&hist[lb], (unsigned int)__builtin_popcountll(hits));
                } else if (hit) {
                    atomicAdd(&hist[bin], 1u);
                }
            }
        }
        __syncthreads();
        // the bin that holds the rank: inclusive scan of the 256 counts by the first four waves (shuffles inside a
        // wave, the three wave totals through LDS) -- a serial walk by one thread cost 16 000 cycles per pass
        unsigned int inc = 0, mine = 0;
        if (threadIdx.x < 256) {
            mine = hist[threadIdx.x];
            inc = mine;
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned int o = __shfl_up(inc, off, 64);
                if (lane >= off) inc += o;
            }
            if (lane == 63) wave_tot[threadIdx.x >> 6] = inc;
        }
        __syncthreads();
        if (threadIdx.x < 256) {
            unsigned int before = 0;
            for (int w = 0; w < (int)(threadIdx.x >> 6); w++) before += wave_tot[w];
            inc += before;
            const uint64_t excl = (uint64_t)(inc - mine);
            const bool last = threadIdx.x == 255;
            if ((rank >= excl && rank < (uint64_t)inc) || (last && rank >= (uint64_t)inc)) {   // exactly one thread
                sh[0] = prefix | ((uint64_t)threadIdx.x << shift);
                sh[1] = rank - excl;
            }
        }
        __syncthreads();
        prefix = sh[0];
        rank = sh[1];
        __syncthreads();
    }
    return prefix;
}

// np.median of a pair's source: the mean of the two middle order statistics (the same one twice for an odd
// count), ranks counted from the source's offset; times `factor`
// The order statistics k1 <= k2 <= k1 + 1 of a band whose values lie in [lo, hi], in two sweeps: values binned
// LINEARLY between the bounds (monotone, so order is kept across bins; a band is a narrow slice of the residuals,
// whose 64-bit images share their leading digits -- the radix passes below spend five of their eight sweeps telling
// nothing apart), the bins of the two ranks found by a scan, their few members collected and ranked by counting.
// Returns false (nothing written) when the band does not qualify: bounds not finite, or more members in the
// selected bins than kBinCand (exact ties) -- the caller then takes the radix path.  Same doubles either way.
constexpr int kSelBins = 2048, kBinCand = 512;   // 12 KB of LDS: a block still fits beside four pyramid tiles (140 of 160 KB) -- with 24 KB it waited for them, 0.6 ms
__device__ bool block_bin_select(const double *__restrict__ r, int N, double lo, double hi, uint64_t k1, uint64_t k2,
                                 unsigned int *hist /* kSelBins */, double *cand /* kBinCand */, double &v1, double &v2) {
    __shared__ unsigned int wave_tot[kTukeyThreads / 64];
    __shared__ unsigned int sel_bin[2], sel_before[2], n_cand;
    __shared__ double res[2];
    const double width = hi - lo;
    if (!(width > 0.0) || !(width < 1e300) || !(fabs(lo) < 1e300)) return false;     // block-uniform
    const double scale = (double)kSelBins / width;
    auto bin_of = [&](double x) {
        const double b = (x - lo) * scale;
        return b >= (double)(kSelBins - 1) ? kSelBins - 1 : (b > 0.0 ? (int)b : 0);
    };
    for (int i = threadIdx.x; i < kSelBins; i += blockDim.x) hist[i] = 0;
    if (threadIdx.x == 0) n_cand = 0;
    __syncthreads();
    for (int i0 = 0; i0 < N; i0 += 4 * blockDim.x) {        // four independent loads per thread and step (L2 latency)
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + u * (int)blockDim.x + (int)threadIdx.x;
            v[u] = i < N ? r[i] : __longlong_as_double(0x7ff8000000000000ll);
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (v[u] == v[u]) atomicAdd(&hist[bin_of(v[u])], 1u);
    }
    __syncthreads();
    // inclusive scan of the bin counts, kSelBins / blockDim.x bins per thread
    constexpr int kPer = kSelBins / kTukeyThreads;
    unsigned int c[kPer], mine = 0;
#pragma unroll
    for (int j = 0; j < kPer; j++) { c[j] = hist[threadIdx.x * kPer + j]; mine += c[j]; }
    unsigned int inc = mine;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned int o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    unsigned int before = inc - mine;
    for (int w = 0; w < wave; w++) before += wave_tot[w];
#pragma unroll
    for (int j = 0; j < kPer; j++) {
        const uint64_t a = before, b = (uint64_t)before + c[j];
        if (k1 >= a && k1 < b) { sel_bin[0] = threadIdx.x * kPer + j; sel_before[0] = before; }
        if (k2 >= a && k2 < b) { sel_bin[1] = threadIdx.x * kPer + j; sel_before[1] = before; }
        before += c[j];
    }
    __syncthreads();
    const unsigned int b1 = sel_bin[0], b2 = sel_bin[1];
    const unsigned int members = hist[b1] + (b2 != b1 ? hist[b2] : 0u);
    if (members > (unsigned int)kBinCand) return false;                              // block-uniform
    for (int i0 = 0; i0 < N; i0 += 4 * blockDim.x) {
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + u * (int)blockDim.x + (int)threadIdx.x;
            v[u] = i < N ? r[i] : __longlong_as_double(0x7ff8000000000000ll);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (v[u] != v[u]) continue;
            const unsigned int b = (unsigned int)bin_of(v[u]);
            if (b == b1 || b == b2) cand[atomicAdd(&n_cand, 1u)] = v[u];
        }
    }
    __syncthreads();
    // rank of every member among the members (ties broken by position: a permutation of 0 .. members - 1); the
    // members of bin b1 come first in value, so rank k - sel_before[0] addresses both statistics
    const unsigned int t1 = (unsigned int)(k1 - sel_before[0]), t2 = (unsigned int)(k2 - sel_before[0]);
    for (unsigned int j = threadIdx.x; j < members; j += blockDim.x) {
        const double x = cand[j];
        unsigned int rank = 0;
        for (unsigned int i = 0; i < members; i++) {
            const double y = cand[i];
            rank += (y < x || (y == x && i < j)) ? 1u : 0u;
        }
        if (rank == t1) res[0] = x;
        if (rank == t2) res[1] = x;
    }
    __syncthreads();
    v1 = res[0];
    v2 = res[1];
    return true;
}

__global__ __launch_bounds__(kTukeyThreads) void k_band_median(const SelectSrc *__restrict__ src,
                                                                const int *__restrict__ count,
                                                                const int *__restrict__ state, double factor,
                                                                double *__restrict__ out, int use_bins) {
    __shared__ unsigned int hist[kSelBins];
    __shared__ double cand[kBinCand];
    __shared__ uint64_t sh[2];
    const int pair = blockIdx.x;
    if (state != nullptr && state[pair] != ST_RUNNING) return;
    const int n = count[pair];
    if (n <= 0) {                                      // empty update mask: the evaluation flags it, any value will do
        if (threadIdx.x == 0) out[pair] = factor;
        return;
    }
    const SelectSrc s = src[pair];
    const uint64_t k1 = (uint64_t)(n - 1) / 2 - s.rank_offset, k2 = (uint64_t)n / 2 - s.rank_offset;
    if (use_bins && s.mode == 0 && s.lo <= s.hi) {
        double v1, v2;
        if (block_bin_select(s.data, s.n, s.lo, s.hi, k1, k2, hist, cand, v1, v2)) {
            if (threadIdx.x == 0) out[pair] = factor * ((v1 + v2) / 2.0);
            return;
        }
        __syncthreads();
    }
    const uint64_t sel = block_radix_select(s.data, s.n, s.mode, s.center, k1, hist, sh);
    const double lo = key_to_double(sel);
    double hi = lo;
    if (k2 != k1) {
        // even count: the next order statistic is the selected key again if more than k1 + 1 values are <= it,
        // else the smallest key above it -- one more sweep instead of a second selection
        __shared__ unsigned int s_le;
        __shared__ unsigned long long s_next;
        if (threadIdx.x == 0) { s_le = 0; s_next = ~0ull; }
        __syncthreads();
        unsigned int le = 0;
        uint64_t next = ~0ull;
        for (int i0 = 0; i0 < s.n; i0 += 4 * blockDim.x) {
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = i0 + u * (int)blockDim.x + (int)threadIdx.x;
                v[u] = i < s.n ? s.data[i] : __longlong_as_double(0x7ff8000000000000ll);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (v[u] != v[u]) continue;
                const uint64_t k = ordered_key(s.mode ? fabs(v[u] - s.center) : v[u]);
                if (k <= sel) le++;
                else if (k < next) next = k;
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            le += __shfl_down(le, off, 64);
            const uint64_t o = __shfl_down(next, off, 64);
            next = o < next ? o : next;
        }
        if ((threadIdx.x & 63) == 0) {
            if (le) atomicAdd(&s_le, le);
            if (next != ~0ull) atomicMin(&s_next, (unsigned long long)next);
        }
        __syncthreads();
        if ((uint64_t)s_le < k1 + 2) hi = key_to_double((uint64_t)s_next);
    }
    if (threadIdx.x == 0) out[pair] = factor * ((lo + hi) / 2.0);
}

// the MAD's bracket: the sample's deviations from the exact median, sorted
__global__ __launch_bounds__(kTukeyThreads) void k_tukey_dev_bracket(const int *__restrict__ state,
                                                                      TukeyBracket *__restrict__ tk,
                                                                      const double *__restrict__ samples,
                                                                      const double *__restrict__ median) {
    __shared__ uint64_t keys[kTukeySample];
    const int pair = blockIdx.x;
    if (state != nullptr && state[pair] != ST_RUNNING) return;
    const unsigned int ns = tk[pair].n_sample;
    const int ms = (int)(ns & 0x7fffffffu);
    const bool exhaustive = (ns & 0x80000000u) != 0;
    const double med = median[pair];
    const int nsort = next_pow2(ms > 0 ? ms : 1);
    const double *__restrict__ mine = samples + (size_t)pair * kTukeySample;
    for (int j = threadIdx.x; j < nsort; j += blockDim.x) keys[j] = j < ms ? ordered_key(fabs(mine[j] - med)) : ~0ull;
    __syncthreads();
    block_bitonic_sort(keys, nsort);
    if (threadIdx.x != 0) return;
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    int r_lo, r_hi;
    bracket_ranks(ms, exhaustive, r_lo, r_hi);
    TukeyBracket *g = tk + pair;
    g->med = med;
    g->dlo = (ms > 0 && r_lo >= 0) ? key_to_double(keys[r_lo]) : 0.0;
    g->dhi = (ms > 0 && r_hi < ms) ? key_to_double(keys[r_hi]) : inf;
    g->overflow = 0;
}

// one 8-byte-per-pixel pass over the residual map: deviations below the bracket are counted, those inside collected
__global__ __launch_bounds__(kBlock) void k_tukey_deviations(const double *__restrict__ rm, int64_t stride, int N,
                                                             const int *__restrict__ state,
                                                             TukeyBracket *__restrict__ tk,
                                                             double *__restrict__ dev_bands, unsigned int cap) {
    __shared__ double band_buf[kBandBuf];
    __shared__ unsigned int band_n, band_base;
    const int pair = blockIdx.y;
    if (state != nullptr && state[pair] != ST_RUNNING) return;
    const TukeyBracket t = tk[pair];
    const double *__restrict__ r = rm + (int64_t)pair * stride;
    double *__restrict__ band = dev_bands + (size_t)pair * cap;
    TukeyBracket *g = tk + pair;
    if (threadIdx.x == 0) band_n = 0;
    __syncthreads();
    unsigned int small = 0;
    auto one = [&](double v) {
        const bool in = v == v;
        const double d = fabs(v - t.med);
        const bool sm = in && d < t.dlo;
        small += sm ? 1u : 0u;
        band_append(in && !sm && d <= t.dhi, d, band_buf, &band_n, &g->n_dev, band, cap, &g->overflow);
    };
    const int N2 = N >> 1;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    for (int i0 = blockIdx.x * kBlock; i0 < N2; i0 += gridDim.x * kBlock) {
        const int i = i0 + (int)threadIdx.x;
        double2_u v;
        v.x = v.y = nan;
        if (i < N2) v = __builtin_nontemporal_load(reinterpret_cast<const double2_u *>(r + 2 * (int64_t)i));
        one(v.x);
        one(v.y);
    }
    if ((N & 1) && blockIdx.x == 0) one(threadIdx.x == 0 ? r[N - 1] : nan);
    for (int off = 32; off > 0; off >>= 1) small += __shfl_down(small, off, 64);
    if ((threadIdx.x & 63) == 0 && small) atomicAdd(&g->small, small);
    band_flush(band_buf, &band_n, &band_base, &g->n_dev, band, cap, &g->overflow);
}

// ---------------------------------------------------------------------------
// Student-t: the ten fixed-point steps of compute_weights_student_t (weights.py:4-16) in two passes over
// the residuals instead of nine.
//
// v_{k+1} = F(v_k), F(v) = mean_i g(s_i; v), g(s; v) = s (nu + 1) / (nu + s / v), s = r^2, v_0 = 1: every step needs
// F at a point that only the previous step knows, hence nine sequential 8-byte-per-pixel passes after the
// mask pass (which gives v_1 = F(1)); they run at 5.2 - 5.8 TB/s, and a working set small enough for the
// 256 MB memory-side cache is no faster (tools/ktrace_levels.py, 32 / 64 / 256 pairs).  But F is analytic in v,
// and with u = nu v + s, t = 1 / u, a = s t:
//     g = (nu+1) v a,   dg/dv = (nu+1) a^2,   d2g/dv2 = -2 nu (nu+1) a^2 t
// so ONE pass can accumulate, for nine expansion points p_1..p_9 at once, the three sums that give F and its
// first two derivatives there (k_student_taylor), and the chain
//     v_{k+1} = F(p_k) + F1(p_k) d + F2(p_k) d^2 / 2,   d = v_k - p_k,   remainder <= (nu + 1) v (d / v)^3
// is then scalar work (k_student_chain).  Pass A expands around the fixed-point sequence of a 2048-residual
// SAMPLE of the pair (k_student_predict: a few per cent off) and yields the iterates to ~1e-5 or better; pass B
// expands around those: remainder far below rounding.  A pair whose pass B still moved a point by more than
// kStudentRedo takes a pass C -- tdk_dvo_get_student_redos counts them.
// v_10 agrees with the sequential passes to < 1e-13 relative (tests/test_gpu_round3.py holds 1e-12).
// tdk_dvo_set_student_passes / TDK_STUDENT=sequential keep the nine passes, TDK_STUDENT_EXACT=1 the nine passes
// with IEEE divisions.
// ---------------------------------------------------------------------------
// (kStudentPts, kStudentSums, kStudentRow: above k_robust_mask, whose Student-t form carries pass A)
// pass B moved a step's expansion point by more than this (relative): its remainder, bounded by
// (nu + 1) (d / v)^3 = 6 * (1e-4)^3 = 6e-12 relative at the threshold (far below the 1e-6 bar on the pose, and
// what the tests hold the variance to is 1e-12 at the d ~ 1e-5 that pass A actually leaves), may no longer be
// negligible -> the pair takes a third pass around the iterates of pass B.  Pass A lands within the threshold
// unless the sample estimate was more than ~8 % off.
constexpr double kStudentRedo = 1e-4;

// the fixed-point sequence v_1 .. v_9 of a sample of the pair's masked residuals: kStudentSample entries of the
// residual map, spread over the frame (all of a small level)
// DIRECT: the sample's residuals are computed here (masked_residual: the mask pass has not run yet -- pass A rides
// in it and needs these points); otherwise read from the residual map.
constexpr int kStudentSample = 2048;
template <bool DIRECT>
__global__ __launch_bounds__(kBlock) void k_student_predict(const double *__restrict__ rm, int64_t stride, int N,
                                                            const int *__restrict__ state, double *__restrict__ pts,
                                                            LevelPtrs L, const PairParams *__restrict__ params,
                                                            const double *__restrict__ poses, double scale) {
    __shared__ double red[kWaves];
    __shared__ int red_n[kWaves];
    const int pair = blockIdx.x;
    if (state != nullptr && state[pair] != ST_RUNNING) return;
    const double *__restrict__ r = DIRECT ? nullptr : rm + (int64_t)pair * stride;
    BlockSetup b;
    const double *__restrict__ tab = nullptr;
    if (DIRECT) {
        load_setup(b, params, poses, pair, scale);
        tab = L.tab + (size_t)pair * (L.W + L.H);
    }
    const int m = N < kStudentSample ? N : kStudentSample;
    constexpr int kPer = kStudentSample / kBlock;
    double sq[kPer];
    int valid = 0;
    if (DIRECT) {
        // the five loads of every sample first (addresses depend on the sample's index only), then the warps: one
        // round of latency instead of kPer (44 -> ~25 us for 256 pairs)
        const int W = L.W, H = L.H;
        const int64_t base = (int64_t)pair * L.stride;
        double tx[kPer], ty[kPer], dd[kPer], ra[kPer], rc[kPer];
#pragma unroll
        for (int j = 0; j < kPer; j++) {
            const int k = threadIdx.x + j * kBlock;
            const int i = sample_position(k < m ? k : 0, m, N, pair);
            const int y = i / W, x = i - y * W;
            tx[j] = tab[x]; ty[j] = tab[W + y];
            dd[j] = L.D0[base + i]; ra[j] = L.I0[base + i]; rc[j] = L.I1[base + i];
        }
#pragma unroll
        for (int j = 0; j < kPer; j++) {
            const int k = threadIdx.x + j * kBlock;
            Pixel p;
            sp_warp(p, true, tx[j], ty[j], dd[j], H, W, b.P, b.c);      // (masked_residual's arithmetic)
            double x = ra[j] - rc[j];
            if (k < m && p.mask == 2 && x == x) valid++; else x = 0.0;
            sq[j] = x * x;
        }
    } else {
#pragma unroll
        for (int j = 0; j < kPer; j++) {
            const int k = threadIdx.x + j * kBlock;
            double x = 0.0;
            if (k < m) {
                x = r[sample_position(k, m, N, pair)];
                if (x == x) valid++; else x = 0.0;
            }
            sq[j] = x * x;                              // outside the mask or the sample: adds nothing to the sum
        }
    }
    for (int off = 32; off > 0; off >>= 1) valid += __shfl_down(valid, off, 64);
    if ((threadIdx.x & 63) == 0) red_n[threadIdx.x >> 6] = valid;
    __syncthreads();
    const int ms = (red_n[0] + red_n[1]) + (red_n[2] + red_n[3]);
    double v = 1.0;
    for (int k = 0; k < kStudentPts; k++) {
        double acc = 0.0;
        const double rv = 1.0 / v;
#pragma unroll
        for (int j = 0; j < kPer; j++)
            acc += sq[j] * ((kStudentNu + 1.0) * fast_rcp(__builtin_fma(sq[j], rv, kStudentNu)));
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        v = ms > 0 ? ((red[0] + red[1]) + (red[2] + red[3])) / (double)ms : 1.0;
        if (!(v > 0.0)) v = 1.0;                        // an all-zero sample: any positive expansion point will do
        if (threadIdx.x == 0) pts[(size_t)pair * kStudentRow + k] = v;
    }
    if (threadIdx.x == 0) pts[(size_t)pair * kStudentRow + kStudentPts] = 1.0;   // passes A and B always run
}

// per block and pair: sum a, sum a^2, sum a^2 t' at each of the nine expansion points, in units of c_ref = nu p_9: s' = s / c_ref, u'_k = p_k / p_9 + s', t'_k = 1 / u'_k (a = s' t'_k = s t_k).
// v_rcp_f64 issues at about a ninth of the FMA rate here; nine of them and their Newton steps were 2/3 of this
// kernel (430 us per pass at 256 x 640x480).  The nine reciprocals of a residual therefore come from ONE:
// prefix products of the u'_k, the reciprocal of the last, and a backward sweep (24 multiplications; the scaling
// keeps the product of nine O(1 + s') factors far from over- and underflow) -- 230 us per pass.
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_student_taylor(const double *__restrict__ rm, int64_t stride, int N,
                                                           const int *__restrict__ state,
                                                           const double *__restrict__ pts,
                                                           double *__restrict__ partial) {
    const int pair = blockIdx.y;
    if (state != nullptr && state[pair] != ST_RUNNING) return;
    if (pts[(size_t)pair * kStudentRow + kStudentPts] == 0.0) return;
    const double p_ref = pts[(size_t)pair * kStudentRow + kStudentPts - 1];
    const double inv_ref = 1.0 / (kStudentNu * p_ref), inv_p = 1.0 / p_ref;
    double c[kStudentPts];                              // p_k / p_9 (block-uniform)
#pragma unroll
    for (int k = 0; k < kStudentPts; k++) c[k] = pts[(size_t)pair * kStudentRow + k] * inv_p;
    double acc[kStudentSums];
#pragma unroll
    for (int j = 0; j < kStudentSums; j++) acc[j] = 0.0;
    const double *r = rm + (int64_t)pair * stride;
#define TDK_STUDENT_TERM(X)                                                                  \
    {                                                                                        \
        const double x_ = (X);                                                               \
        const double s_ = x_ == x_ ? (x_ * x_) * inv_ref : 0.0; /* outside the mask: zeros */\
        double u[kStudentPts], q[kStudentPts];                                               \
        _Pragma("unroll") for (int k = 0; k < kStudentPts; k++) u[k] = c[k] + s_;            \
        q[0] = u[0];                                                                         \
        _Pragma("unroll") for (int k = 1; k < kStudentPts; k++) q[k] = q[k - 1] * u[k];      \
        double inv = fast_rcp(q[kStudentPts - 1]);                                           \
        _Pragma("unroll") for (int k = kStudentPts - 1; k >= 0; k--) {                       \
            const double t = k > 0 ? inv * q[k > 0 ? k - 1 : 0] : inv;                       \
            if (k > 0) inv *= u[k];                                                          \
            const double a = s_ * t, at = a * t;                                             \
            acc[3 * k + 0] += a;                                                             \
            acc[3 * k + 1] = __builtin_fma(a, a, acc[3 * k + 1]);                            \
            acc[3 * k + 2] = __builtin_fma(a, at, acc[3 * k + 2]);                           \
        }                                                                                    \
    }
    const int N2 = N >> 1;
    // ~170 FP64 instructions per 16-byte load: the next load is issued before the arithmetic of this one
    const int step = gridDim.x * kBlock;
    int i = blockIdx.x * kBlock + threadIdx.x;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    double2_u v = i < N2 ? __builtin_nontemporal_load(reinterpret_cast<const double2_u *>(r + 2 * (int64_t)i)) : double2_u{nan, nan};
#pragma unroll 1
    for (; i < N2; i += step) {
        const int nxt = i + step;
        const double2_u w = nxt < N2 ? __builtin_nontemporal_load(reinterpret_cast<const double2_u *>(r + 2 * (int64_t)nxt)) : double2_u{nan, nan};
        TDK_STUDENT_TERM(v.x)
        TDK_STUDENT_TERM(v.y)
        v = w;
    }
    if ((N & 1) && blockIdx.x == 0 && threadIdx.x == 0) TDK_STUDENT_TERM(r[N - 1])
#undef TDK_STUDENT_TERM
    __shared__ double red[kWaves][kStudentSums];
#pragma unroll
    for (int j = 0; j < kStudentSums; j++) {
        double v = acc[j];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][j] = v;
    }
    __syncthreads();
    if (threadIdx.x < kStudentSums)
        partial[((int64_t)pair * gridDim.x + blockIdx.x) * kStudentSums + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// Pass A on its own (TDK_STUDENT_FUSED=0; by default it rides in the mask pass): StudentF32 over the residual map,
// 165 us instead of 320 in FP64 at 256 x 640x480.
__global__ __launch_bounds__(kBlock) void k_student_taylor_f32(const double *__restrict__ rm, int64_t stride, int N,
                                                               const int *__restrict__ state,
                                                               const double *__restrict__ pts,
                                                               double *__restrict__ partial) {
    const int pair = blockIdx.y;
    if (state != nullptr && state[pair] != ST_RUNNING) return;
    if (pts[(size_t)pair * kStudentRow + kStudentPts] == 0.0) return;
    StudentF32 ta;
    ta.init(pts, pair);
    const double *r = rm + (int64_t)pair * stride;
    const int N2 = N >> 1;
    const int step = gridDim.x * kBlock;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    // kDeep 16-byte loads in flight per lane: with one, 4 waves per SIMD had 4 MB on their way chip-wide -- 2 TB/s at
    // the latency of HBM under load, and the arithmetic (packed FP32) needs less time than that
    constexpr int kDeep = 4;
    auto fetch = [&](int j) {
        return j < N2 ? __builtin_nontemporal_load(reinterpret_cast<const double2_u *>(r + 2 * (int64_t)j)) : double2_u{nan, nan};
    };
    int i = blockIdx.x * kBlock + threadIdx.x;
    double2_u ring[kDeep];
#pragma unroll
    for (int d = 0; d < kDeep; d++) ring[d] = fetch(i + d * step);
#pragma unroll 1
    for (; i < N2; i += kDeep * step) {
#pragma unroll
        for (int d = 0; d < kDeep; d++) {
            const double2_u v = ring[d];
            ring[d] = fetch(i + (kDeep + d) * step);
            ta.add(v.x, v.y);
        }
    }
    if ((N & 1) && blockIdx.x == 0 && threadIdx.x == 0) ta.add(r[N - 1], nan);
    __shared__ double red[kWaves][kStudentSums];
    ta.store(red, partial + ((int64_t)pair * gridDim.x + blockIdx.x) * kStudentSums);
}

// one wave per pair: block partials -> sums (fixed order), then the chain v_1 -> v_10 through the nine Taylor
// parabolas.  pass 0 (A): the iterates become the expansion points of pass B.  pass 1 (B): v_10 is the variance,
// unless a point moved by more than kStudentRedo -- then the pair is flagged for pass 2 (C), which is final.
__global__ __launch_bounds__(64) void k_student_chain(const double *__restrict__ partial, int nblk,
                                                      const int *__restrict__ count, const int *__restrict__ state,
                                                      double *__restrict__ pts, double *__restrict__ variance, int pass,
                                                      unsigned int *__restrict__ n_redo) {
    __shared__ double sums[kStudentSums];
    const int pair = blockIdx.x;
    if (state != nullptr && state[pair] != ST_RUNNING) return;
    double *p = pts + (size_t)pair * kStudentRow;
    if (p[kStudentPts] == 0.0) return;
    if (threadIdx.x < kStudentSums) {
        double s = 0.0;
        for (int b = 0; b < nblk; b++) s += partial[((int64_t)pair * nblk + b) * kStudentSums + threadIdx.x];
        sums[threadIdx.x] = s;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const int n = count[pair];
    const double v1 = variance[pair];                   // v_1 = F(1), exact, from the mask pass
    if (pass == 0 && (n <= 0 || !(v1 > 0.0))) {
        // empty mask: 0 / 0 as in the reference; all residuals zero: the second step divides 0 by 0 (weights.py:13)
        variance[pair] = __longlong_as_double(0x7ff8000000000000ll);
        p[kStudentPts] = 0.0;
        return;
    }
    const double inv_n = 1.0 / (double)n, nu = kStudentNu, q = nu + 1.0;
    const double inv_ref = 1.0 / (nu * p[kStudentPts - 1]);   // the sums of a^2 t' are in units of c_ref (k_student_taylor)
    double v = pass == 0 ? v1 : p[0];                   // after pass A, p[0] holds v_1 (variance[] the latest v_10)
    double moved = 0.0;
    double it[kStudentPts];
    for (int k = 0; k < kStudentPts; k++) {
        const double pk = p[k], d = v - pk;
        const double F0 = q * pk * sums[3 * k] * inv_n, F1 = q * sums[3 * k + 1] * inv_n;
        const double F2 = -2.0 * nu * q * (sums[3 * k + 2] * inv_ref) * inv_n;
        it[k] = v;
        moved = fmax(moved, fabs(d) / pk);
        v = F0 + d * (F1 + d * (0.5 * F2));
        if (!(v > 0.0)) v = pk;                         // a polynomial far outside its range: stay put, pass C repairs
    }
    variance[pair] = v;
    const bool again = pass == 0 || (pass == 1 && moved > kStudentRedo);
    if (again) {
        for (int k = 0; k < kStudentPts; k++) p[k] = it[k];
        if (pass == 1) atomicAdd(n_redo, 1u);
    }
    p[kStudentPts] = again ? 1.0 : 0.0;
    // pass C itself moved a point by more than the threshold: the sample's sequence was tens of per cent off (a small
    // mask with gross outliers -- found by tests/test_gpu_fuzz.py: 137 residuals, seven of them 70 sigma out, variance
    // 1.2e-4 off after pass C) and the parabolas have not contracted yet -> the pair takes the nine steps one after
    // the other (k_student_sequential), from v_1 = p[0]
    if (pass == 2 && moved > kStudentRedo) {
        p[kStudentPts] = 2.0;
        atomicAdd(n_redo + 1, 1u);
    }
}

// The nine remaining fixed-point steps of a pair that the Taylor passes gave up on, one block per pair, one step after
// the other over the pair's residual map with the arithmetic of k_robust_student_step<true>.  Every other pair's block
// returns after one load.  Rare by construction (see k_student_chain); a VGA pair costs it ~1 ms.
__global__ __launch_bounds__(kBlock) void k_student_sequential(const double *__restrict__ rm, int64_t stride, int N,
                                                               const int *__restrict__ count,
                                                               const int *__restrict__ state, double *__restrict__ pts,
                                                               double *__restrict__ variance) {
    __shared__ double red[kWaves];
    const int pair = blockIdx.x;
    if (state != nullptr && state[pair] != ST_RUNNING) return;
    double *p = pts + (size_t)pair * kStudentRow;
    if (p[kStudentPts] != 2.0) return;
    const double *r = rm + (int64_t)pair * stride;
    const double n = (double)count[pair];
    double v = p[0];                                    // v_1 = F(1), exact, from the mask pass
    for (int it = 1; it < 10; it++) {
        const double rvar = 1.0 / v;
        double acc = 0.0;
        for (int i = threadIdx.x; i < N; i += kBlock) {
            const double x = r[i];
            if (x == x) {
                const double s = x * x;
                acc += s * ((kStudentNu + 1.0) * fast_rcp(__builtin_fma(s, rvar, kStudentNu)));
            }
        }
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        v = ((red[0] + red[1]) + (red[2] + red[3])) / n;
    }
    if (threadIdx.x == 0) {
        variance[pair] = v;
        p[kStudentPts] = 0.0;
    }
}

// ---- synthetic scene on the device (tadataka_amd/synthetic.py) ------------
__device__ __forceinline__ double tex(double x, double y) {
    return 0.5 + 0.25 * sin(x / 7.0) * cos(y / 5.0) + 0.2 * sin((x + y) / 11.0);
}

__device__ __forceinline__ double unit_noise(uint64_t seed, uint64_t idx) {
    uint64_t z = seed * 0x9E3779B97F4A7C15ull + idx * 0xBF58476D1CE4E5B9ull + 0x94D049BB133111EBull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;  // U(-1, 1)
}

__global__ __launch_bounds__(kBlock) void k_fill_synthetic(double *I0, double *D0, double *I1, double *W0,
                                                           int64_t stride, int H, int W, Cam cam,
                                                           const double *__restrict__ poses,
                                                           uint64_t seed0, double noise) {
    const int pair = blockIdx.y;
    const int64_t N = (int64_t)H * W;
    const double *P = poses + 12 * pair;
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < N; i += (int64_t)gridDim.x * kBlock) {
        int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
        double d = 2.0 + 0.3 * sin((double)x / 40.0) + 0.2 * cos((double)y / 30.0);
        double xn = ((double)x - cam.ox) / cam.fx, yn = ((double)y - cam.oy) / cam.fy;
        double px = xn * d, py = yn * d, pz = d;
        double qx = P[0] * px + P[1] * py + P[2] * pz + P[9];
        double qy = P[3] * px + P[4] * py + P[5] * pz + P[10];
        double qz = P[6] * px + P[7] * py + P[8] * pz + P[11];
        double u = qx / qz * cam.fx + cam.ox, v = qy / qz * cam.fy + cam.oy;
        uint64_t s = seed0 + (uint64_t)pair;
        int64_t o = (int64_t)pair * stride + i;
        D0[o] = d;
        I1[o] = tex((double)x, (double)y) + noise * unit_noise(2 * s, (uint64_t)i);
        I0[o] = tex(u, v) + noise * unit_noise(2 * s + 1, (uint64_t)i);
        if (W0) W0[o] = 1.0;
    }
}

}  // namespace

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
struct tdk_dvo {
    hipStream_t stream;   // every launch and copy of this batch is queued here; batches do not share streams,
                          // so the pyramid of one batch overlaps the (FP64-bound) estimation of another
    int n_pairs, H, W, n_levels;
    double ratio;
    bool with_w;
    struct Level {
        int H, W;
        int64_t N, stride;
        double scale;
        double *I0, *D0, *I1, *W0;
        double *tab;   // [n_pairs][W + H], see k_norm_tables
    } lv[kMaxLevels];
    PairParams *d_params;
    double *d_poses_in;  // [n][12] host-provided poses for evaluate()
    double *d_partials, *d_results;
    LoopState ls;
    int max_blocks;
    // robust statistics (allocated on first use)
    double *d_rm;         // [n][stride0] masked residual map
    double *d_wscale;     // [n] variance (student-t) / sigma_mad (tukey)
    double *d_stat;       // [n][4]: lo, hi, median, spare
    double *d_spartial;   // [n][kStatBlocks][kStudentSums] block partials of the statistics passes | [n][kStatBlocks] of the first step (fused pass A)
    double *d_st_pts;     // [n][kStudentRow] Student-t: expansion points of the Taylor passes, redo flag
    unsigned int *d_st_redo;   // [0] pairs that took a third Taylor pass, [1] pairs that fell back to the nine sequential steps (diagnostics)
    int *d_count;         // [n]
    void *d_select;       // SelectState[n]
    unsigned int *d_hist; // [n][kSelectBins]
    uint64_t *d_cand;     // [n][kSelectCap] keys that share 26 bits with the median
    TukeyBracket *d_tk;   // [n] Tukey: brackets + counters (k_tukey_sample)
    double *d_tk_med, *d_tk_dev;   // [n][tk_cap] residuals / deviations collected from the median / MAD brackets
    unsigned int tk_cap;           // doubles per pair in a band: a quarter of the frame
    void *d_tk_src;                // SelectSrc[n]
    double *d_tk_sample;           // [n][kTukeySample] sorted sample of the masked residuals
    unsigned int *d_tk_fallback;   // [1] pairs that took the exact radix path of k_tukey_finish (diagnostic)
    // profiling of the finest-level evaluation kernel (bench.py roofline leg)
    int profiling;               // 0 off, 1 the finest level's launches, 2 every level's (tdk_dvo_set_profiling)
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used;
    std::vector<int> ev_round;   // per event: [2 i] pairs evaluated in full, [2 i + 1] pairs probed by launch i
    double prof_ms[kMaxLevels][3];   // per level; buckets: full / probe / mixed launches (collect_profile)
    int64_t prof_launches[kMaxLevels][3], prof_pixels[kMaxLevels][3];
    std::vector<int> ev_level;   // level of each event pair
    std::vector<double> cams;   // cameras currently on the device: [cam0 (n x 4) | cam1 (n x 4)]
    hipStream_t copy_stream;    // tdk_dvo_upload_async: the library-wide copy stream (not owned)
    uint8_t *d_u8;              // staging for 8-bit frames (tdk_dvo_upload_async_u8), [n_pairs][N]
    size_t u8_bytes;
    hipEvent_t ev_copy, ev_xs;  // copy stream <-> batch stream; library stream <-> batch stream
    // tdk_dvo_upload_async*: recorded on the copy stream after the last queued upload (and its conversion).  The
    // batch's stream waits for it when it next touches the arrays (after_uploads), not when the upload is queued:
    // a wait queued at once sits in the stream's hardware queue for the whole transfer, and HIP maps all streams
    // onto a few hardware queues (4 unless GPU_MAX_HW_QUEUES says otherwise) -- other batches' kernels stood
    // behind that barrier (the 8-bit stream of bench.py ran 3.4 ms per step instead of 2.5).
    hipEvent_t ev_uploaded;
    bool uploads_pending;
    int *d_mode_probe;          // [n] MODE_PROBE (tdk_dvo_photometric_error), allocated on first use
    int64_t count_error_px, count_update_px;   // tdk_dvo_get_counts: source pixels of the last estimate call
    bool anti_aliasing;         // levels without a plan get skimage's Gaussian prefilter (tdk_dvo_set_anti_aliasing)
    // skimage.transform.rescale to the bit (pyramid.hip): per-level plans from the host (tdk_dvo_set_level_plan),
    // the identity-scale level 0 the reference also sends through rescale, clip=True (tdk_dvo_set_rescale_options)
    struct Plan {
        bool set;
        double map[4];
        std::vector<double> wr, wc;
    } plan[kMaxLevels];
    double *raw[4];             // the uploaded frames of the arrays whose level 0 is a rescale of its own (else null:
                                // lv[0] IS the upload)
    unsigned level0_mask;       // bit k: array k (I0, D0, I1, W0) gets rescale(., 1.0) as its level 0
    bool clip;                  // clip=True: outputs clipped to the extremes of the filtered image
    int clip_clean;             // how many clip slots the last build left reset (few slots: k_clip_small); 0: none
    void *d_clip;               // ClipSlot[n_pairs * 4 * n_levels]
    bool weights_dirty;         // the device copy of the kernels is stale (a plan changed)
    int n_cu, device;           // compute units and index of the batch's device
    int student_mode;           // Student-t variance: 0 two Taylor passes, 1 nine sequential passes, 2 ... with IEEE divisions
    int opt_chain, opt_tukey;   // tdk_dvo_set_option
    double *d_aa_weights;       // the pyramid's 1-D kernels, per level and axis (allocated on first use)
    int *h_flag;   // "pairs still running", written by k_dvo_reduce (mapped pinned host memory)
    int *d_chain_state, *d_gate;   // the speculative level chain: [n_levels][n_pairs] state arrays, the current level
    // Prior poses in, final poses and warnings out of tdk_dvo_estimate*: mapped pinned host memory that the first /
    // last kernel of a call reads / writes directly.  As hipMemcpyAsync these few kilobytes queued behind whatever
    // bulk upload was in flight on the copy engine -- 1.4 ms per 78 MB of frames -- and with them the estimation.
    double *h_io, *d_io;   // [n x 12 in][n x 12 out][n ints]
    std::vector<int> host_warn;   // ls.warn of the last estimate call
};

namespace {

constexpr int kStatBlocks = 64;
constexpr int kSelectBlocks = 16;

int level_dim(int full, double scale) {
    // skimage.transform.rescale: output shape = round(shape * scale) (np.round)
    int v = (int)nearbyint((double)full * scale);
    return v < 1 ? 1 : v;
}

void plan_blocks(const tdk_dvo *h, const tdk_dvo::Level &L, int *nblk, int64_t *chunk) {
    // >= 16 pixels per thread (the pipeline of k_dvo_eval takes ~3 pixel-steps to
    // fill and drain), but no more than ~8192 blocks in the whole grid; measured
    // flat (+-0.5 %) from 12 to 40 pixels per thread on the bench workload
    // -- unless the batch is so small that this would leave CUs without a block
    // (a single pair): then down to 4 pixels per thread, aiming at >= 512 blocks
    constexpr int64_t target_blocks = 512;
    int64_t px_per_thread = L.N * h->n_pairs / (target_blocks * (int64_t)kBlock);
    constexpr int64_t min_px = 4;
    px_per_thread = px_per_thread < min_px ? min_px : (px_per_thread > 16 ? 16 : px_per_thread);
    int64_t per_block = (int64_t)kBlock * px_per_thread;
    int64_t nb = (L.N + per_block - 1) / per_block;
    int64_t cap = 8192 / h->n_pairs;
    if (cap < 1) cap = 1;
    if (nb > cap) nb = cap;
    if (nb > h->max_blocks) nb = h->max_blocks;
    if (nb < 1) nb = 1;
    // Batches that fill the chip several times over: longer blocks whose count is a whole number of resident rounds.
    // A block's fixed cost (tables, three pipeline steps to fill and drain, 30 wave reductions) is ~5 pixel-steps: at
    // 16 pixels per thread a quarter of a coarse level's time, and a last round that is a third full costs as much
    // again.  Among 16 .. 64 pixels per thread take the count with the best (work / rounded-up rounds) x (1 - fixed
    // share).  256 VGA pairs: level 2 six blocks of 40 pixels per thread per pair (two full rounds; 0.131 -> 0.123 ms),
    // level 1 nine of 60 (three rounds; 0.279 -> 0.261 ms), level 0 twenty-one of 57 (seven rounds; 0.545 -> 0.534 ms):
    // bench step 3.09 -> 3.04 ms.  The probe shares the partition (a pose's error is the same double in both
    // modes); at eight blocks per CU its rounds no longer come out whole, which costs it ~1 % at level 0.
    if (px_per_thread >= 16 && h->n_cu > 0) {
        const double resident = 3.0 * h->n_cu;                   // three blocks per CU at 154 VGPRs
        const int64_t nb_lo = std::max<int64_t>(1, (L.N + 64 * kBlock - 1) / (64 * kBlock));
        const int64_t nb_hi = std::min<int64_t>(nb, std::max<int64_t>(1, L.N / (16 * kBlock)));
        double best = -1.0;
        int64_t best_nb = nb;
        for (int64_t cand = nb_lo; cand <= nb_hi; cand++) {
            const double rounds = (double)cand * h->n_pairs / resident;
            const double steps = (double)L.N / ((double)cand * kBlock);
            const double score = rounds / ceil(rounds) * (steps / (steps + 5.0));
            if (score > best * 1.0001) { best = score; best_nb = cand; }
        }
        nb = best_nb;
    }
    int64_t c = (L.N + nb - 1) / nb;
    c = (c + 2 * kBlock - 1) / (2 * kBlock) * (2 * kBlock);  // even, whole block sweeps
    nb = (L.N + c - 1) / c;
    *nblk = (int)nb;
    *chunk = c;
}

LevelPtrs ptrs_of(const tdk_dvo::Level &L) {
    LevelPtrs p;
    p.I0 = L.I0; p.D0 = L.D0; p.I1 = L.I1; p.W0 = L.W0; p.tab = L.tab;
    p.stride = L.stride; p.H = L.H; p.W = L.W; p.N = L.N;
    return p;
}

// the batch's stream waits for the asynchronous uploads queued so far (no-op when there are none)
tdk_status after_uploads(tdk_dvo *h) {
    if (!h->uploads_pending) return TDK_OK;
    TDK_HIP(hipStreamWaitEvent(h->stream, h->ev_uploaded, 0));
    h->uploads_pending = false;
    return TDK_OK;
}

tdk_status upload_params(tdk_dvo *h, const double *cam0, const double *cam1) {
    // every evaluation / estimation entry comes through here: frames uploaded asynchronously are waited for first
    TDK_TRY(after_uploads(h));
    // same cameras as last time (the usual case for a sequence): nothing to do
    const size_t n4 = (size_t)4 * h->n_pairs;
    if (h->cams.size() == 2 * n4 && !memcmp(h->cams.data(), cam0, sizeof(double) * n4) &&
        !memcmp(h->cams.data() + n4, cam1, sizeof(double) * n4))
        return TDK_OK;
    h->cams.assign(cam0, cam0 + n4);
    h->cams.insert(h->cams.end(), cam1, cam1 + n4);
    void *stage;
    TDK_TRY(tdk::pinned(0, sizeof(PairParams) * h->n_pairs, &stage));
    PairParams *pp = (PairParams *)stage;
    for (int i = 0; i < h->n_pairs; i++) {
        for (int k = 0; k < 4; k++) {
            pp[i].cam0[k] = cam0[4 * i + k];
            pp[i].cam1[k] = cam1[4 * i + k];
        }
    }
    TDK_HIP(hipMemcpyAsync(h->d_params, pp, sizeof(PairParams) * h->n_pairs, hipMemcpyHostToDevice,
                           h->stream));
    for (int l = 0; l < h->n_levels; l++) {
        const tdk_dvo::Level &L = h->lv[l];
        dim3 grid((L.W + L.H + 255) / 256, h->n_pairs);
        k_norm_tables<<<grid, 256, 0, h->stream>>>(h->d_params, L.scale, L.W, L.H, L.tab);
        TDK_LAUNCH_CHECK();
    }
    // the staging buffer is reused by the next call: wait for the copy
    TDK_HIP(hipStreamSynchronize(h->stream));
    return TDK_OK;
}

tdk_status ensure_robust_buffers(tdk_dvo *h) {
    if (h->d_hist) return TDK_OK;   // the last buffer allocated below: all or nothing
    const size_t n = (size_t)h->n_pairs;
    TDK_HIP(hipMalloc(&h->d_rm, sizeof(double) * (size_t)h->lv[0].stride * n));
    TDK_HIP(hipMalloc(&h->d_wscale, sizeof(double) * n));
    TDK_HIP(hipMalloc(&h->d_stat, sizeof(double) * 4 * n));
    TDK_HIP(hipMalloc(&h->d_spartial, sizeof(double) * kStatBlocks * (kStudentSums + 1) * n));   // + the first step's partials
    TDK_HIP(hipMalloc(&h->d_st_pts, sizeof(double) * kStudentRow * n));
    TDK_HIP(hipMalloc(&h->d_st_redo, 2 * sizeof(unsigned int)));
    TDK_HIP(hipMemsetAsync(h->d_st_redo, 0, 2 * sizeof(unsigned int), h->stream));
    TDK_HIP(hipMalloc(&h->d_count, sizeof(int) * n));
    TDK_HIP(hipMalloc(&h->d_select, sizeof(SelectState) * n));
    TDK_HIP(hipMalloc(&h->d_cand, sizeof(uint64_t) * kSelectCap * n));
    TDK_HIP(hipMalloc(&h->d_tk, sizeof(TukeyBracket) * n));
    h->tk_cap = (unsigned int)(h->lv[0].N / 4 > 8192 ? h->lv[0].N / 4 : 8192);
    TDK_HIP(hipMalloc(&h->d_tk_med, sizeof(double) * (size_t)h->tk_cap * n));
    TDK_HIP(hipMalloc(&h->d_tk_dev, sizeof(double) * (size_t)h->tk_cap * n));
    TDK_HIP(hipMalloc(&h->d_tk_src, sizeof(SelectSrc) * n));
    TDK_HIP(hipMalloc(&h->d_tk_sample, sizeof(double) * kTukeySample * n));
    TDK_HIP(hipMalloc(&h->d_tk_fallback, sizeof(unsigned int)));
    TDK_HIP(hipMemsetAsync(h->d_tk_fallback, 0, sizeof(unsigned int), h->stream));
    TDK_HIP(hipMalloc(&h->d_hist, sizeof(unsigned int) * kSelectBins * n));
    TDK_HIP(hipMemsetAsync(h->d_hist, 0, sizeof(unsigned int) * kSelectBins * n, h->stream));
    return TDK_OK;
}

// median over each running pair's masked residuals (mode 0) or absolute
// deviations from `center` (mode 1), times `factor` -> out[pair]
tdk_status device_median(tdk_dvo *h, int level, const int *d_state, int mode, const double *center,
                         double factor, double *out, const SelectSrc *src = nullptr) {
    const tdk_dvo::Level &L = h->lv[level];
    const int n = h->n_pairs, tpb = 256, gp = (n + tpb - 1) / tpb;
    dim3 grid(kStatBlocks, n);
    SelectState *st = (SelectState *)h->d_select;
    k_select_init<<<gp, tpb, 0, h->stream>>>(st, h->d_count, src, n);
    TDK_LAUNCH_CHECK();
    // few, long blocks for the histogram passes: zeroing and merging 8192 bins is a
    // fixed cost per block
    dim3 hgrid(kSelectBlocks, n);
    for (int pass = 0; pass < kSelectPasses; pass++) {
        k_select_hist<<<hgrid, kBlock, 0, h->stream>>>(h->d_rm, L.stride, (int)L.N, d_state, mode, center, src, st, pass,
                                                       h->d_hist);
        TDK_LAUNCH_CHECK();
        k_select_pick<<<n, kBlock, 0, h->stream>>>(h->d_hist, st, d_state, pass);
        TDK_LAUNCH_CHECK();
        if (pass <= 1) {
            // usually a handful of keys share 26 bits with the median -- and when the median is close to
            // zero (the residuals themselves, as opposed to their absolute deviations) already its first
            // 13 bits (sign, exponent, one mantissa bit) single out a small group: finish on those.  Pairs
            // whose group is larger than kSelectCap (exact ties) are left for the remaining
            // passes, which return at once for every pair that is done.
            k_select_collect<<<grid, kBlock, 0, h->stream>>>(h->d_rm, L.stride, (int)L.N, d_state, mode, center, src, st,
                                                             h->d_cand, pass);
            TDK_LAUNCH_CHECK();
            k_select_finish<<<n, kBlock, 0, h->stream>>>(st, h->d_cand, d_state, factor, out);
            TDK_LAUNCH_CHECK();
        }
    }
    k_select_successor<<<grid, kBlock, 0, h->stream>>>(h->d_rm, L.stride, (int)L.N, d_state, mode, center, src, st);
    TDK_LAUNCH_CHECK();
    k_median_combine<<<gp, tpb, 0, h->stream>>>(st, h->d_count, d_state, n, factor, out);
    TDK_LAUNCH_CHECK();
    return TDK_OK;
}

// Student-t variance / Tukey sigma_mad of the masked residuals at `d_poses`
tdk_status prepare_robust(tdk_dvo *h, int level, const double *d_poses, const int *d_state, int weight_mode) {
    if (weight_mode != TDK_W_STUDENT_T && weight_mode != TDK_W_TUKEY) return TDK_OK;
    TDK_TRY(ensure_robust_buffers(h));
    const tdk_dvo::Level &L = h->lv[level];
    const int n = h->n_pairs;
    // blocks per pair of the passes over the residuals: 64 at full resolution, fewer and longer ones on the coarse
    // levels -- a block's set-up (pose, tables, bracket) is a chain of dependent loads that two pixels per thread
    // cannot amortise (the mask pass of a 213x284 level took 216 us with 64 blocks per pair, 104 us with 8)
    const int nb = (int)(L.N / 8192 < 8 ? 8 : (L.N / 8192 > kStatBlocks ? kStatBlocks : L.N / 8192));
    dim3 grid(nb, n);
    TDK_HIP(hipMemsetAsync(h->d_count, 0, sizeof(int) * n, h->stream));
    const bool exact = h->student_mode == 2;
    // tdk_dvo_set_option(TDK_DVO_OPT_TUKEY): 1 = the two medians by radix select (2 x 3 passes over the residual
    // map) instead of the sampled brackets + one pass; 2 = brackets, but every pair takes the exact slow path of
    // k_tukey_finish (tests).  All three give the same doubles.
    const int tukey_mode = h->opt_tukey;
    const bool taylor = weight_mode == TDK_W_STUDENT_T && h->student_mode == 0;
    const bool brackets = weight_mode == TDK_W_TUKEY && tukey_mode != 1;
    TukeyArgs tka{h->d_tk, h->d_tk_med, h->tk_cap};
    if (brackets) {
        k_tukey_sample<<<n, kTukeyThreads, 0, h->stream>>>(ptrs_of(L), h->d_params, d_poses, d_state, L.scale, h->d_tk,
                                                           h->d_tk_sample);
        TDK_LAUNCH_CHECK();
    }
    // Student-t, Taylor scheme: pass A rides in the mask pass, so the sample's fixed-point sequence comes first,
    // from residuals computed on the spot
    const bool fused_a = taylor;
    double *partial_v1 = fused_a ? h->d_spartial + (size_t)kStatBlocks * kStudentSums * n : h->d_spartial;
    if (fused_a) {
        k_student_predict<true><<<n, kBlock, 0, h->stream>>>(nullptr, L.stride, (int)L.N, d_state, h->d_st_pts, ptrs_of(L),
                                                             h->d_params, d_poses, L.scale);
        TDK_LAUNCH_CHECK();
    }
#define TDK_MASK(ST, FA, TK, TA)                                                                                      \
    k_robust_mask<ST, FA, TK, TA><<<grid, kBlock, 0, h->stream>>>(ptrs_of(L), h->d_params, d_poses, d_state, L.scale, \
                                                                  h->d_rm, h->d_count, partial_v1, tka, h->d_st_pts,  \
                                                                  h->d_spartial)
    if (brackets) TDK_MASK(false, false, true, false);
    else if (weight_mode != TDK_W_STUDENT_T) TDK_MASK(false, false, false, false);
    else if (exact) TDK_MASK(true, false, false, false);
    else if (fused_a) TDK_MASK(true, true, false, true);
    else TDK_MASK(true, true, false, false);
#undef TDK_MASK
    TDK_LAUNCH_CHECK();
    if (brackets) {
        const int force = tukey_mode == 2 ? 1 : 0, gp = (n + 255) / 256;
        double *median = h->d_stat + 2 * (size_t)n;
        SelectSrc *src = (SelectSrc *)h->d_tk_src;
        k_tukey_plan<<<gp, 256, 0, h->stream>>>(h->d_tk, h->d_count, d_state, 0, h->d_rm, L.stride, (int)L.N, h->d_tk_med,
                                                h->tk_cap, nullptr, force, src, h->d_tk_fallback, n);
        TDK_LAUNCH_CHECK();
        const int use_bins = 1;
        k_band_median<<<n, kTukeyThreads, 0, h->stream>>>(src, h->d_count, d_state, 1.0, median, use_bins);
        TDK_LAUNCH_CHECK();
        k_tukey_dev_bracket<<<n, kTukeyThreads, 0, h->stream>>>(d_state, h->d_tk, h->d_tk_sample, median);
        TDK_LAUNCH_CHECK();
        k_tukey_deviations<<<grid, kBlock, 0, h->stream>>>(h->d_rm, L.stride, (int)L.N, d_state, h->d_tk, h->d_tk_dev,
                                                           h->tk_cap);
        TDK_LAUNCH_CHECK();
        k_tukey_plan<<<gp, 256, 0, h->stream>>>(h->d_tk, h->d_count, d_state, 1, h->d_rm, L.stride, (int)L.N, h->d_tk_dev,
                                                h->tk_cap, median, force, src, h->d_tk_fallback, n);
        TDK_LAUNCH_CHECK();
        k_band_median<<<n, kTukeyThreads, 0, h->stream>>>(src, h->d_count, d_state, kTukeyC, h->d_wscale, use_bins);   // c * MAD (:34)
        TDK_LAUNCH_CHECK();
        return TDK_OK;
    }
    if (weight_mode == TDK_W_STUDENT_T) {
        // the mask pass has left the partial sums of the first step (variance 1, weights.py:10-13)
        k_robust_student_update<<<n, 64, 0, h->stream>>>(partial_v1, nb, h->d_count, d_state,
                                                         h->d_wscale, n, 0);
        TDK_LAUNCH_CHECK();
        if (taylor) {   // steps 2 .. 10 from two passes (see k_student_taylor); a third one for flagged pairs only
            if (!fused_a) {
                k_student_predict<false><<<n, kBlock, 0, h->stream>>>(h->d_rm, L.stride, (int)L.N, d_state, h->d_st_pts,
                                                                      ptrs_of(L), h->d_params, d_poses, L.scale);
                TDK_LAUNCH_CHECK();
            }
            // one round of resident blocks (4 per CU at 128 VGPRs): a block ends with 36 wave reductions -- the price of
            // ~10 loop iterations -- so it should run many (150 at 256 x 640x480), and a second, partial round would idle CUs
            const int nbt = std::max(1, std::min(nb, 4 * h->n_cu / n));
            dim3 tgrid(nbt, n);
            for (int pass = 0; pass < 3; pass++) {
                if (pass == 0 && fused_a) {     // the sums of pass A are where the mask pass left them, nb blocks per pair
                    k_student_chain<<<n, 64, 0, h->stream>>>(h->d_spartial, nb, h->d_count, d_state, h->d_st_pts,
                                                             h->d_wscale, pass, h->d_st_redo);
                    TDK_LAUNCH_CHECK();
                    continue;
                }
                if (pass == 0)
                    k_student_taylor_f32<<<tgrid, kBlock, 0, h->stream>>>(h->d_rm, L.stride, (int)L.N, d_state, h->d_st_pts,
                                                                          h->d_spartial);
                else
                    k_student_taylor<<<tgrid, kBlock, 0, h->stream>>>(h->d_rm, L.stride, (int)L.N, d_state, h->d_st_pts,
                                                                      h->d_spartial);
                TDK_LAUNCH_CHECK();
                k_student_chain<<<n, 64, 0, h->stream>>>(h->d_spartial, nbt, h->d_count, d_state, h->d_st_pts,
                                                         h->d_wscale, pass, h->d_st_redo);
                TDK_LAUNCH_CHECK();
            }
            k_student_sequential<<<n, kBlock, 0, h->stream>>>(h->d_rm, L.stride, (int)L.N, h->d_count, d_state,
                                                              h->d_st_pts, h->d_wscale);
            TDK_LAUNCH_CHECK();
            return TDK_OK;
        }
        for (int it = 1; it < 10; it++) {   // n_iter = 10 (weights.py:4)
            if (exact)
                k_robust_student_step<false><<<grid, kBlock, 0, h->stream>>>(h->d_rm, L.stride, (int)L.N, d_state,
                                                                              h->d_wscale, h->d_spartial);
            else
                k_robust_student_step<true><<<grid, kBlock, 0, h->stream>>>(h->d_rm, L.stride, (int)L.N, d_state,
                                                                             h->d_wscale, h->d_spartial);
            TDK_LAUNCH_CHECK();
            k_robust_student_update<<<n, 64, 0, h->stream>>>(h->d_spartial, nb, h->d_count, d_state,
                                                             h->d_wscale, n, 0);
            TDK_LAUNCH_CHECK();
        }
    } else {
        double *median = h->d_stat + 2 * (size_t)n;
        TDK_TRY(device_median(h, level, d_state, 0, nullptr, 1.0, median));
        TDK_TRY(device_median(h, level, d_state, 1, median, kTukeyC, h->d_wscale));  // c * MAD (:34)
    }
    return TDK_OK;
}

// d_mode: per-pair evaluation mode of the device loop (NULL: full); d_stat_state: the pairs whose
// evaluation needs the robust statistics (running and not a probe; NULL: all)
// probe_only: the caller knows that every running pair is in probe mode -- the statistics (for normal equations
// only) are skipped altogether and the probe body runs as its own kernel (k_dvo_probe)
tdk_status launch_eval(tdk_dvo *h, int level, const double *d_poses, const int *d_state, const int *d_mode,
                       const int *d_stat_state, int weight_mode, bool probe_only = false) {
    const tdk_dvo::Level &L = h->lv[level];
    if (!probe_only) TDK_TRY(prepare_robust(h, level, d_poses, d_stat_state, weight_mode));
    int nblk;
    int64_t chunk;
    plan_blocks(h, L, &nblk, &chunk);
    // see k_dvo_eval: XCD-major order for batches, a pair's blocks over all XCDs for fewer than 8 pairs
    const bool spread = h->n_pairs < 8;
    const int n_pairs_arg = spread ? -h->n_pairs : h->n_pairs;
    dim3 grid(spread ? (unsigned)h->n_pairs * (unsigned)nblk : 8u * (unsigned)((h->n_pairs + 7) / 8) * (unsigned)nblk);
    LevelPtrs P = ptrs_of(L);
    // (Capping the evaluation at 2 blocks per CU by padding this request, so that the other batch's pyramid
    // blocks -- 36 KB of LDS, 121 VGPRs -- co-reside on the same SIMDs: full evaluation 0.55 -> 0.65 ms, bench
    // step 2.73 -> 2.92 ms.  The two kernels do not fill each other's issue gaps; measured in round 4, not kept.)
    const size_t lds = sizeof(double) * (kWaves * kAccPad + (size_t)L.W + 3 * (size_t)eval_rows(chunk, L.W));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->profiling && (level == 0 || h->profiling == 2)) {
        while (h->ev_pool.size() < h->ev_used + 2) {
            hipEvent_t e;
            TDK_HIP(hipEventCreate(&e));
            h->ev_pool.push_back(e);
            h->ev_round.push_back(0);
            h->ev_level.push_back(0);
        }
        h->ev_level[h->ev_used] = level;
        // until the device loop says otherwise: every pair, in full (tdk_dvo_evaluate)
        h->ev_round[h->ev_used] = h->n_pairs;
        h->ev_round[h->ev_used + 1] = 0;
        e0 = h->ev_pool[h->ev_used++];
        e1 = h->ev_pool[h->ev_used++];
        TDK_HIP(hipEventRecord(e0, h->stream));
    }
    if (lds > 160 * 1024) {
        tdk::set_error("frames of %d x %d need %zu bytes of LDS for the coordinate tables (limit 160 KiB)", L.W, L.H,
                       lds);
        return TDK_ERR_INVALID_ARGUMENT;
    }
#define TDK_EVAL(WM)                                                                                    \
    if (lds > 64 * 1024)   /* beyond the default dynamic-LDS limit of a launch */                      \
        TDK_HIP(hipFuncSetAttribute((const void *)k_dvo_eval<WM>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    (int)lds));                                                         \
    k_dvo_eval<WM><<<grid, kBlock, lds, h->stream>>>(P, h->d_params, d_poses, d_state, d_mode, h->d_wscale, \
                                                         L.scale, chunk, n_pairs_arg, nblk, h->d_partials)
    if (probe_only) {
        if (lds > 64 * 1024)
            TDK_HIP(hipFuncSetAttribute((const void *)k_dvo_probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k_dvo_probe<<<grid, kBlock, lds, h->stream>>>(P, h->d_params, d_poses, d_state, L.scale, chunk, n_pairs_arg, nblk,
                                                      h->d_partials);
    } else
    switch (weight_mode) {
        case TDK_W_NONE: TDK_EVAL(TDK_W_NONE); break;
        case TDK_W_HUBER: TDK_EVAL(TDK_W_HUBER); break;
        case TDK_W_STUDENT_T: TDK_EVAL(TDK_W_STUDENT_T); break;
        case TDK_W_TUKEY: TDK_EVAL(TDK_W_TUKEY); break;
        case TDK_W_MAP: TDK_EVAL(TDK_W_MAP); break;
        default:
            tdk::set_error("unknown weight mode %d", weight_mode);
            return TDK_ERR_INVALID_ARGUMENT;
    }
#undef TDK_EVAL
    TDK_LAUNCH_CHECK();
    if (e1) TDK_HIP(hipEventRecord(e1, h->stream));
    return TDK_OK;
}

tdk_status launch_reduce(tdk_dvo *h, int level, int loop_mode, int max_iter, const LoopState *ls = nullptr,
                         const ChainArgs *chain = nullptr) {
    int nblk;
    int64_t chunk;
    plan_blocks(h, h->lv[level], &nblk, &chunk);
    const ChainArgs ca = chain ? *chain : ChainArgs{nullptr, nullptr, 0, 0, nullptr, nullptr};
    if (h->n_pairs < 8 && nblk > 32)
        k_dvo_reduce<32><<<h->n_pairs, 1024, 0, h->stream>>>(h->d_partials, nblk, h->d_results, ls ? *ls : h->ls,
                                                               loop_mode, max_iter, ca);
    else
        k_dvo_reduce<kBlock / 32><<<h->n_pairs, kBlock, 0, h->stream>>>(h->d_partials, nblk, h->d_results,
                                                                         ls ? *ls : h->ls, loop_mode, max_iter, ca);
    TDK_LAUNCH_CHECK();
    return TDK_OK;
}

tdk_status check_level(const tdk_dvo *h, int level) {
    TDK_REQUIRE(h != nullptr, "handle is NULL");
    TDK_REQUIRE(level >= 0 && level < h->n_levels, "level out of range");
    return TDK_OK;
}

tdk_status check_weight_mode(const tdk_dvo *h, int weight_mode) {
    if (weight_mode == TDK_W_MAP && !h->with_w) {
        tdk::set_error("weight map requested but the batch was created without one");
        return TDK_ERR_INVALID_ARGUMENT;
    }
    if (weight_mode < TDK_W_NONE || weight_mode > TDK_W_MAP) {
        tdk::set_error("unknown weight mode %d", weight_mode);
        return TDK_ERR_INVALID_ARGUMENT;
    }
    return TDK_OK;
}

// Event pairs of the full-resolution launches since the last collection, sorted into three buckets
// by what the launch evaluated (ev_round, one entry per pair of events): 0 only full evaluations,
// 1 only probes, 2 both.
tdk_status collect_profile(tdk_dvo *h) {
    for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
        float ms = 0.f;
        TDK_HIP(hipEventElapsedTime(&ms, h->ev_pool[i], h->ev_pool[i + 1]));
        const int n_full = h->ev_round[i], n_probe = h->ev_round[i + 1];
        const int b = n_probe == 0 ? 0 : (n_full == 0 ? 1 : 2);
        const int l = h->ev_level[i];
        h->prof_ms[l][b] += ms;
        h->prof_launches[l][b] += 1;
        h->prof_pixels[l][b] += h->lv[l].N * (int64_t)(n_full + n_probe);
    }
    h->ev_used = 0;
    return TDK_OK;
}

// One pyramid level for the whole batch; poses live in h->ls.pose on entry and exit.
// The host only needs to know when every pair has finished: one 4-byte read per
// round trip.  Large batches take one iteration per round trip (the wait is ~1 % of
// an iteration).  Small ones (a single pair: the drop-in PoseChangeEstimator) are
// latency-bound, so two iterations are queued per round trip -- the second one
// returns at once for pairs that finished in the first (state != RUNNING) -- which
// halves the host waits of the typical 2-3 evaluation level.  Evaluations are
// counted on the device (LoopState::evals), so the pixel bookkeeping stays exact.
// The loop in two halves: queue_round puts the next evaluation(s) of a level on the batch's stream, after_round --
// once that stream has drained -- reads what k_dvo_reduce left in h_flag and says whether the level goes on.
// (Several batches driven in lockstep through these halves, so that one batch's kernels fill the other's round
// trips, was built and measured on the 8-bit stream of bench.py: 2.79 ms per batch against 2.58 ms one after the
// other -- the pyramid and the conversion of the next batches already fill the gaps.  Not kept.)
struct LevelRun {
    int level, round, max_rounds, burst;
    bool probe_only;    // every pair still running tests a candidate next (h_flag[8..9] of the last round)
};

static LevelRun begin_level(tdk_dvo *h, int level, int max_iter) {
    const bool small = (int64_t)h->n_pairs * h->lv[level].N <= (1ll << 22) && !h->profiling;
    constexpr int small_burst = 2;
    // a pair goes through at most 2 max_iter + 1 evaluations: the first, then per tested candidate a
    // probe and -- if it was accepted and is not the last -- the full evaluation at the accepted pose
    return LevelRun{level, 0, 2 * max_iter + 1, small ? small_burst : 1, false};   // a level starts with full evaluations
}

static tdk_status queue_round(tdk_dvo *h, LevelRun &r, int weight_mode, int max_iter) {
    const int nb = r.max_rounds - r.round < r.burst ? r.max_rounds - r.round : r.burst;
    for (int b = 0; b < nb; b++) {
        // (the modes of a burst's second round are not known here: the combined kernel)
        TDK_TRY(launch_eval(h, r.level, h->ls.cand, h->ls.state, h->ls.mode, h->ls.stat_state, weight_mode,
                            b == 0 && r.probe_only));
        TDK_TRY(launch_reduce(h, r.level, 1, max_iter));
    }
    r.round += nb;
    return TDK_OK;
}

// the batch's stream has been waited for; true: another round is needed
static bool after_round(tdk_dvo *h, LevelRun &r) {
    r.probe_only = ((volatile int *)h->h_flag)[8] == 0 && ((volatile int *)h->h_flag)[9] > 0;
    if (h->profiling && (r.level == 0 || h->profiling == 2) && h->ev_used >= 2) {   // what the launch just timed evaluated
        h->ev_round[h->ev_used - 2] = ((volatile int *)h->h_flag)[4];
        h->ev_round[h->ev_used - 1] = ((volatile int *)h->h_flag)[5];
    }
    return *(volatile int *)h->h_flag > 0 && r.round < r.max_rounds;
}

static void end_level(tdk_dvo *h, const LevelRun &r, int64_t *pixel_evals) {
    // evaluations in the reference's sense: one per PhotometricError call (the full evaluation of an
    // accepted candidate is the second half of the evaluation its probe began)
    const int64_t evals = (int64_t)*(volatile unsigned long long *)(h->h_flag + 2);
    const int64_t updates = (int64_t)*(volatile unsigned long long *)(h->h_flag + 6);
    if (pixel_evals) *pixel_evals += h->lv[r.level].N * evals;
    h->count_error_px += h->lv[r.level].N * evals;
    h->count_update_px += h->lv[r.level].N * updates;
}

tdk_status run_level(tdk_dvo *h, int level, int weight_mode, int max_iter, int64_t *pixel_evals) {
    LevelRun r = begin_level(h, level, max_iter);
    do {
        TDK_TRY(queue_round(h, r, weight_mode, max_iter));
        TDK_HIP(hipStreamSynchronize(h->stream));   // k_dvo_reduce left the counts in h_flag
    } while (after_round(h, r));
    end_level(h, r, pixel_evals);
    return TDK_OK;
}

// The whole coarse-to-fine estimation of a small batch with one host wait in the typical case (see
// k_chain_init / chain_next_level).  Weight modes without robust statistics only (their extra kernels per
// full evaluation are decided on the host).
static bool chain_applies(const tdk_dvo *h, int weight_mode) {
    return h->opt_chain && !h->profiling && (int64_t)h->n_pairs * h->lv[0].N <= (1ll << 22) &&
           weight_mode != TDK_W_STUDENT_T && weight_mode != TDK_W_TUKEY;
}

static tdk_status chain_estimate(tdk_dvo *h, int weight_mode, int max_iter, double *poses12, int64_t *pixel_evals) {
    const int n = h->n_pairs, L = h->n_levels;
    volatile int *flag = (volatile int *)h->h_flag;
    LoopState top = h->ls;
    top.state = h->d_chain_state + (size_t)(L - 1) * n;
    k_chain_init<<<1, 256, 0, h->stream>>>(top, h->d_chain_state, h->d_gate, n, L, h->d_io);
    TDK_LAUNCH_CHECK();
    int level = L - 1;
    bool fresh = true;
    // every host round makes progress (at least one more evaluation of the unfinished level), and a level takes
    // at most 2 max_iter + 1 evaluations
    for (int guard = 0; guard < (2 * max_iter + 2) * L + 2; guard++) {
        for (int l = level; l >= 0; l--) {
            LoopState ls = h->ls;
            ls.state = h->d_chain_state + (size_t)l * n;
            ls.fuse_first = l == L - 1;
            const int burst = (l == L - 1 && (fresh || l != level)) ? 3 : 2;
            const ChainArgs ca{h->d_chain_state, h->d_gate, l, n, h->d_io + 12 * (size_t)n, (int *)(h->d_io + 24 * (size_t)n)};
            for (int b = 0; b < burst; b++) {
                TDK_TRY(launch_eval(h, l, h->ls.cand, ls.state, h->ls.mode, nullptr, weight_mode));
                TDK_TRY(launch_reduce(h, l, 1, max_iter, &ls, &ca));
            }
        }
        TDK_HIP(hipStreamSynchronize(h->stream));
        const int g = flag[kFlagGate];
        if (g < 0) {
            for (int l = 0; l < L; l++) {
                const int64_t evals = (int64_t)*(volatile unsigned long long *)(h->h_flag + kFlagLevelEvals + 4 * l);
                const int64_t updates = (int64_t)*(volatile unsigned long long *)(h->h_flag + kFlagLevelEvals + 4 * l + 2);
                if (pixel_evals) *pixel_evals += h->lv[l].N * evals;
                h->count_error_px += h->lv[l].N * evals;
                h->count_update_px += h->lv[l].N * updates;
            }
            memcpy(poses12, h->h_io + 12 * (size_t)n, sizeof(double) * 12 * n);
            memcpy(h->host_warn.data(), h->h_io + 24 * (size_t)n, sizeof(int) * n);
            return TDK_OK;
        }
        level = g;
        fresh = false;
    }
    tdk::set_error("level chain did not terminate");
    return TDK_ERR_HIP;
}

}  // namespace

// where the frames of array k (0 I0, 1 D0, 2 I1, 3 W0) arrive: level 0 itself, unless that array's level 0 is a
// rescale of its own (tdk_dvo_set_rescale_options)
static double *upload_ptr(const tdk_dvo *h, int k) {
    if (h->raw[k]) return h->raw[k];
    const tdk_dvo::Level &L = h->lv[0];
    return k == 0 ? L.I0 : k == 1 ? L.D0 : k == 2 ? L.I1 : L.W0;
}

namespace tdk {

tdk_status dvo_level0(tdk_dvo *h, DvoLevel0 *out) {
    TDK_REQUIRE(h != nullptr && out != nullptr, "null pointer");
    const tdk_dvo::Level &L = h->lv[0];
    out->I0 = upload_ptr(h, 0); out->D0 = upload_ptr(h, 1); out->I1 = upload_ptr(h, 2); out->W0 = upload_ptr(h, 3);
    out->stride = L.stride; out->H = L.H; out->W = L.W; out->n_pairs = h->n_pairs;
    out->stream = h->stream;
    out->poses = h->ls.pose;
    return TDK_OK;
}

}  // namespace tdk

static tdk_status dvo_allocate(tdk_dvo *h, int n_pairs, int height, int width, int n_levels, double ratio,
                               int with_weight_map);

extern "C" {

tdk_status tdk_dvo_create(int n_pairs, int height, int width, int n_levels, double ratio,
                          int with_weight_map, tdk_dvo **out) {
    TDK_API_GUARD;
    TDK_REQUIRE(out != nullptr, "out is NULL");
    TDK_REQUIRE(n_pairs >= 1 && n_pairs <= 65535, "n_pairs must be in [1, 65535]");
    TDK_REQUIRE(height >= 2 && width >= 2, "frames must be at least 2x2");
    TDK_REQUIRE((int64_t)height * width < (1ll << 28), "frame too large");
    TDK_REQUIRE((size_t)(height + width + kWaves * kAccPad) * sizeof(double) <= 160 * 1024,
                "height + width must stay below 20 000 (per-block coordinate tables live in LDS)");
    TDK_REQUIRE(n_levels >= 1 && n_levels <= kMaxLevels, "n_levels must be in [1, 16]");
    TDK_REQUIRE(ratio > 1.0 || n_levels == 1, "layer_size_ratio must be > 1");
    TDK_TRY(tdk::ensure_device());
    tdk_dvo *h = new tdk_dvo();   // value-initialised: every pointer is null until allocated
    const tdk_status st = dvo_allocate(h, n_pairs, height, width, n_levels, ratio, with_weight_map);
    if (st != TDK_OK) {
        tdk_dvo_destroy(h);       // a failed hipMalloc of a large batch must not leak what came before it
        return st;
    }
    *out = h;
    return TDK_OK;
}

static tdk_status dvo_allocate(tdk_dvo *h, int n_pairs, int height, int width, int n_levels, double ratio,
                               int with_weight_map) {
    TDK_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->n_pairs = n_pairs; h->H = height; h->W = width; h->n_levels = n_levels;
    h->ratio = ratio; h->with_w = with_weight_map != 0;
    h->anti_aliasing = true;   // the reference-equivalent pyramid (skimage.transform.rescale's default)
    for (int l = 0; l < kMaxLevels; l++) h->plan[l].set = false;
    for (int k = 0; k < 4; k++) h->raw[k] = nullptr;
    h->level0_mask = 0u;
    h->clip = false;
    h->clip_clean = 0;
    h->d_clip = nullptr;
    h->weights_dirty = true;
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
        h->n_cu = cus > 0 ? cus : 256;
        h->device = dev;
    }
    h->student_mode = 0;
    h->opt_chain = 1;
    h->opt_tukey = 0;
    h->max_blocks = 1024;
    h->d_rm = nullptr; h->d_wscale = nullptr; h->d_stat = nullptr; h->d_spartial = nullptr; h->d_st_pts = nullptr; h->d_st_redo = nullptr;
    h->d_count = nullptr; h->d_select = nullptr; h->d_hist = nullptr;
    h->profiling = 0; h->ev_used = 0;
    memset(h->prof_ms, 0, sizeof(h->prof_ms)); memset(h->prof_launches, 0, sizeof(h->prof_launches));
    memset(h->prof_pixels, 0, sizeof(h->prof_pixels));
    for (int l = 0; l < n_levels; l++) {
        tdk_dvo::Level &L = h->lv[l];
        L.scale = 1.0 / pow(ratio, (double)l);  // level_to_scale, vo/dvo/__init__.py:42-43
        L.H = l == 0 ? height : level_dim(height, L.scale);
        L.W = l == 0 ? width : level_dim(width, L.scale);
        if (L.H < 2 || L.W < 2) {
            tdk::set_error("pyramid level %d would be %dx%d", l, L.H, L.W);
            return TDK_ERR_INVALID_ARGUMENT;
        }
        L.N = (int64_t)L.H * L.W;
        L.stride = (L.N + 1) & ~1ll;
        // + padding: k_dvo_eval's unconditional stream loads overrun a block's range by < 4 kBlock pixels
        size_t bytes = ((size_t)L.stride * n_pairs + 4 * kBlock) * sizeof(double);
        L.W0 = nullptr;
        TDK_HIP(hipMalloc(&L.I0, bytes));
        TDK_HIP(hipMalloc(&L.D0, bytes));
        TDK_HIP(hipMalloc(&L.I1, bytes));
        if (h->with_w) TDK_HIP(hipMalloc(&L.W0, bytes));
        TDK_HIP(hipMalloc(&L.tab, sizeof(double) * (size_t)(L.W + L.H) * n_pairs));
    }
    TDK_HIP(hipMalloc(&h->d_params, sizeof(PairParams) * n_pairs));
    TDK_HIP(hipMalloc(&h->d_poses_in, sizeof(double) * 12 * n_pairs));
    TDK_HIP(hipMalloc(&h->d_partials, sizeof(double) * kAccPad * (size_t)h->max_blocks * n_pairs));
    TDK_HIP(hipMalloc(&h->d_results, sizeof(double) * kAccPad * n_pairs));
    TDK_HIP(hipMalloc(&h->ls.pose, sizeof(double) * 12 * n_pairs));
    TDK_HIP(hipMalloc(&h->ls.cand, sizeof(double) * 12 * n_pairs));
    TDK_HIP(hipMalloc(&h->ls.prev_err, sizeof(double) * n_pairs));
    TDK_HIP(hipMalloc(&h->ls.state, sizeof(int) * n_pairs));
    TDK_HIP(hipMalloc(&h->ls.n_evals, sizeof(int) * n_pairs));
    TDK_HIP(hipMalloc(&h->ls.active, sizeof(int)));
    TDK_HIP(hipMalloc(&h->ls.ticket, sizeof(int)));
    TDK_HIP(hipMalloc(&h->ls.evals, 2 * sizeof(unsigned long long)));
    TDK_HIP(hipMalloc(&h->ls.warn, sizeof(int) * n_pairs));
    TDK_HIP(hipMalloc(&h->ls.mode, sizeof(int) * n_pairs));
    TDK_HIP(hipMalloc(&h->ls.tested, sizeof(int) * n_pairs));
    TDK_HIP(hipMalloc(&h->ls.stat_state, sizeof(int) * n_pairs));
    TDK_HIP(hipMalloc(&h->ls.next, 2 * sizeof(int)));
    TDK_HIP(hipMalloc(&h->ls.round, sizeof(int) * 2));
    h->host_warn.assign((size_t)n_pairs, 0);
    TDK_HIP(hipHostMalloc((void **)&h->h_io, sizeof(double) * 24 * (size_t)n_pairs + sizeof(int) * (size_t)n_pairs, hipHostMallocMapped));
    TDK_HIP(hipHostGetDevicePointer((void **)&h->d_io, h->h_io, 0));
    TDK_HIP(hipHostMalloc(&h->h_flag, (16 + 4 * kMaxLevels) * sizeof(int), hipHostMallocMapped));
    TDK_HIP(hipMalloc(&h->d_chain_state, sizeof(int) * (size_t)n_pairs * n_levels));
    TDK_HIP(hipMalloc(&h->d_gate, sizeof(int)));
    TDK_HIP(hipHostGetDevicePointer((void **)&h->ls.host_flag, h->h_flag, 0));
    return TDK_OK;
}

tdk_status tdk_dvo_destroy(tdk_dvo *h) {
    TDK_API_GUARD;
    if (!h) return TDK_OK;
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (int l = 0; l < h->n_levels; l++) {   // hipFree(nullptr) is a no-op: a partially built handle is fine
        (void)hipFree(h->lv[l].I0); (void)hipFree(h->lv[l].D0); (void)hipFree(h->lv[l].I1);
        if (h->lv[l].W0) (void)hipFree(h->lv[l].W0);
        (void)hipFree(h->lv[l].tab);
    }
    (void)hipFree(h->d_params); (void)hipFree(h->d_poses_in); (void)hipFree(h->d_partials);
    (void)hipFree(h->d_results); (void)hipFree(h->ls.pose); (void)hipFree(h->ls.cand);
    (void)hipFree(h->ls.prev_err); (void)hipFree(h->ls.state); (void)hipFree(h->ls.n_evals);
    (void)hipFree(h->ls.active); (void)hipFree(h->ls.ticket); (void)hipFree(h->ls.evals); (void)hipFree(h->ls.warn);
    (void)hipFree(h->ls.mode); (void)hipFree(h->ls.tested); (void)hipFree(h->ls.stat_state); (void)hipFree(h->ls.round); (void)hipFree(h->ls.next);
    (void)hipFree(h->d_rm); (void)hipFree(h->d_wscale); (void)hipFree(h->d_stat);
    (void)hipFree(h->d_spartial); (void)hipFree(h->d_st_pts); (void)hipFree(h->d_st_redo); (void)hipFree(h->d_count); (void)hipFree(h->d_select);
    (void)hipFree(h->d_hist); (void)hipFree(h->d_cand); (void)hipFree(h->d_mode_probe);
    (void)hipFree(h->d_tk); (void)hipFree(h->d_tk_med); (void)hipFree(h->d_tk_dev); (void)hipFree(h->d_tk_sample); (void)hipFree(h->d_tk_fallback); (void)hipFree(h->d_tk_src);
    for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
    if (h->h_flag) (void)hipHostFree(h->h_flag);
    (void)hipFree(h->d_chain_state); (void)hipFree(h->d_gate);
    if (h->h_io) (void)hipHostFree(h->h_io);
    if (h->d_aa_weights) (void)hipFree(h->d_aa_weights);
    for (int k = 0; k < 4; k++) (void)hipFree(h->raw[k]);
    (void)hipFree(h->d_clip);
    if (h->d_u8) (void)hipFree(h->d_u8);
    if (h->uploads_pending && h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
    if (h->ev_copy) (void)hipEventDestroy(h->ev_copy);
    if (h->ev_uploaded) (void)hipEventDestroy(h->ev_uploaded);
    if (h->ev_xs) (void)hipEventDestroy(h->ev_xs);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return TDK_OK;
}

tdk_status tdk_dvo_upload(tdk_dvo *h, int pair, const double *I0, const double *D0, const double *I1,
                          const double *weight_map) {
    TDK_API_GUARD;
    TDK_REQUIRE(h && I0 && D0 && I1, "null pointer");
    TDK_REQUIRE(pair >= 0 && pair < h->n_pairs, "pair out of range");
    TDK_REQUIRE(weight_map == nullptr || h->with_w, "batch was created without a weight map");
    const tdk_dvo::Level &L = h->lv[0];
    size_t bytes = (size_t)L.N * sizeof(double);
    int64_t off = (int64_t)pair * L.stride;
    TDK_TRY(after_uploads(h));
    TDK_HIP(hipMemcpyAsync(upload_ptr(h, 0) + off, I0, bytes, hipMemcpyHostToDevice, h->stream));
    TDK_HIP(hipMemcpyAsync(upload_ptr(h, 1) + off, D0, bytes, hipMemcpyHostToDevice, h->stream));
    TDK_HIP(hipMemcpyAsync(upload_ptr(h, 2) + off, I1, bytes, hipMemcpyHostToDevice, h->stream));
    if (weight_map)
        TDK_HIP(hipMemcpyAsync(upload_ptr(h, 3) + off, weight_map, bytes, hipMemcpyHostToDevice, h->stream));
    TDK_HIP(hipStreamSynchronize(h->stream));
    return TDK_OK;
}

tdk_status tdk_dvo_upload_mixed(tdk_dvo *h, int pair, const double *const *host4, const double *const *device4) {
    TDK_API_GUARD;
    TDK_REQUIRE(h && host4 && device4, "null pointer");
    TDK_REQUIRE(pair >= 0 && pair < h->n_pairs, "pair out of range");
    TDK_REQUIRE((host4[3] == nullptr && device4[3] == nullptr) || h->with_w, "batch was created without a weight map");
    const tdk_dvo::Level &L = h->lv[0];
    const size_t bytes = (size_t)L.N * sizeof(double);
    const int64_t off = (int64_t)pair * L.stride;
    double *dst[4] = {upload_ptr(h, 0) + off, upload_ptr(h, 1) + off, upload_ptr(h, 2) + off,
                      L.W0 ? upload_ptr(h, 3) + off : nullptr};
    TDK_TRY(after_uploads(h));
    bool any_device = false, any_host = false;
    for (int k = 0; k < 4; k++) any_device |= device4[k] != nullptr;
    if (any_device) {
        // device sources are produced on the library stream (tdk_map / tdk_frame): order the copies
        // behind what is queued there, and what the library stream does next behind the copies
        if (!h->ev_xs) TDK_HIP(hipEventCreateWithFlags(&h->ev_xs, hipEventDisableTiming));
        TDK_HIP(hipEventRecord(h->ev_xs, tdk::stream()));
        TDK_HIP(hipStreamWaitEvent(h->stream, h->ev_xs, 0));
    }
    for (int k = 0; k < 4; k++) {
        if (device4[k])
            TDK_HIP(hipMemcpyAsync(dst[k], device4[k], bytes, hipMemcpyDeviceToDevice, h->stream));
        else if (host4[k]) {
            TDK_HIP(hipMemcpyAsync(dst[k], host4[k], bytes, hipMemcpyHostToDevice, h->stream));
            any_host = true;
        }
    }
    if (any_device) {
        TDK_HIP(hipEventRecord(h->ev_xs, h->stream));
        TDK_HIP(hipStreamWaitEvent(tdk::stream(), h->ev_xs, 0));
    }
    if (any_host) TDK_HIP(hipStreamSynchronize(h->stream));   // the caller's arrays may go away
    return TDK_OK;
}

// One copy stream for the whole library (every batch's uploads take turns on it -- and on one DMA engine:
// with a stream per batch the 8-bit upload of three rotating batches ran at 22 GB/s instead of 50), one event
// per batch to order it with the batch's own stream.
static tdk_status ensure_copy_stream(tdk_dvo *h) {
    // one copy stream per device for all batches (a stream belongs to the device it was created on)
    static hipStream_t g_copy_stream[16] = {};
    if (!h->copy_stream) {
        const int dev = h->device;
        TDK_REQUIRE(dev >= 0 && dev < 16, "device index beyond the copy-stream table");
        if (!g_copy_stream[dev]) TDK_HIP(hipStreamCreateWithFlags(&g_copy_stream[dev], hipStreamNonBlocking));
        h->copy_stream = g_copy_stream[dev];
    }
    if (!h->ev_copy) TDK_HIP(hipEventCreateWithFlags(&h->ev_copy, hipEventDisableTiming));
    if (!h->ev_uploaded) TDK_HIP(hipEventCreateWithFlags(&h->ev_uploaded, hipEventDisableTiming));
    return TDK_OK;
}

tdk_status tdk_dvo_upload_async(tdk_dvo *h, int which, int first_pair, int n_pairs, const double *pinned_host) {
    TDK_API_GUARD;
    TDK_REQUIRE(h && pinned_host, "null pointer");
    TDK_REQUIRE(which >= 0 && which <= 3 && (which != 3 || h->with_w), "no such array");
    TDK_REQUIRE(first_pair >= 0 && n_pairs >= 1 && first_pair + n_pairs <= h->n_pairs, "pair range out of bounds");
    const tdk_dvo::Level &L = h->lv[0];
    double *base = upload_ptr(h, which);
    TDK_TRY(ensure_copy_stream(h));
    // after what the batch's own stream still does with the old frames ...
    TDK_HIP(hipEventRecord(h->ev_copy, h->stream));
    TDK_HIP(hipStreamWaitEvent(h->copy_stream, h->ev_copy, 0));
    const size_t row = (size_t)L.N * sizeof(double);
    if ((int64_t)L.N == L.stride)
        TDK_HIP(hipMemcpyAsync(base + (int64_t)first_pair * L.stride, pinned_host, row * (size_t)n_pairs,
                               hipMemcpyHostToDevice, h->copy_stream));
    else
        TDK_HIP(hipMemcpy2DAsync(base + (int64_t)first_pair * L.stride, (size_t)L.stride * sizeof(double), pinned_host,
                                 row, row, (size_t)n_pairs, hipMemcpyHostToDevice, h->copy_stream));
    // ... and before what it does next (see tdk_dvo::ev_uploaded)
    TDK_HIP(hipEventRecord(h->ev_uploaded, h->copy_stream));
    h->uploads_pending = true;
    return TDK_OK;
}

tdk_status tdk_dvo_upload_async_u8(tdk_dvo *h, int which, int first_pair, int n_pairs, const uint8_t *pinned_host) {
    TDK_API_GUARD;
    TDK_REQUIRE(h && pinned_host, "null pointer");
    TDK_REQUIRE(which >= 0 && which <= 2, "8-bit frames: I0, D0 (rarely) or I1");
    TDK_REQUIRE(first_pair >= 0 && n_pairs >= 1 && first_pair + n_pairs <= h->n_pairs, "pair range out of bounds");
    const tdk_dvo::Level &L = h->lv[0];
    double *base = upload_ptr(h, which);
    TDK_TRY(ensure_copy_stream(h));
    const size_t need = (size_t)h->n_pairs * (size_t)L.N;
    if (h->u8_bytes < need) {
        if (h->d_u8) { TDK_HIP(hipStreamSynchronize(h->stream)); TDK_HIP(hipStreamSynchronize(h->copy_stream)); (void)hipFree(h->d_u8); h->d_u8 = nullptr; }
        TDK_HIP(hipMalloc(&h->d_u8, need));
        h->u8_bytes = need;
    }
    // the conversion overwrites frames the batch's own stream may still be reading
    TDK_HIP(hipEventRecord(h->ev_copy, h->stream));
    TDK_HIP(hipStreamWaitEvent(h->copy_stream, h->ev_copy, 0));
    uint8_t *stage = h->d_u8 + (size_t)first_pair * (size_t)L.N;
    TDK_HIP(hipMemcpyAsync(stage, pinned_host, (size_t)n_pairs * (size_t)L.N, hipMemcpyHostToDevice, h->copy_stream));
    // converted on the copy stream, in order behind the transfer (and before the next one reuses the staging bytes)
    dim3 grid((unsigned)((L.N + kBlock * 4 - 1) / (kBlock * 4)), (unsigned)n_pairs);   // two sweeps of two pixels
    k_u8_to_f64<<<grid, kBlock, 0, h->copy_stream>>>(stage, base + (int64_t)first_pair * L.stride, L.N, L.stride);
    TDK_LAUNCH_CHECK();
    TDK_HIP(hipEventRecord(h->ev_uploaded, h->copy_stream));
    h->uploads_pending = true;
    return TDK_OK;
}

tdk_status tdk_dvo_fill_synthetic(tdk_dvo *h, const double *camera, const double *poses12, uint64_t seed0,
                                  double noise) {
    TDK_API_GUARD;
    TDK_REQUIRE(h && camera && poses12, "null pointer");
    TDK_TRY(after_uploads(h));
    TDK_HIP(hipMemcpyAsync(h->d_poses_in, poses12, sizeof(double) * 12 * h->n_pairs, hipMemcpyHostToDevice,
                           h->stream));
    const tdk_dvo::Level &L = h->lv[0];
    int gx = (int)((L.N + kBlock * 4 - 1) / (kBlock * 4));
    dim3 grid(gx < 1 ? 1 : gx, h->n_pairs);
    Cam cam{camera[0], camera[1], camera[2], camera[3]};
    k_fill_synthetic<<<grid, kBlock, 0, h->stream>>>(upload_ptr(h, 0), upload_ptr(h, 1), upload_ptr(h, 2),
                                                         L.W0 ? upload_ptr(h, 3) : nullptr, L.stride, L.H, L.W, cam,
                                                         h->d_poses_in, seed0, noise);
    TDK_LAUNCH_CHECK();
    TDK_HIP(hipStreamSynchronize(h->stream));
    return TDK_OK;
}

// arrays: bit 0 I0, bit 1 D0, bit 2 I1, bit 3 W0 -- the arrays whose levels are (re)built: levels 1 .. n_levels - 1
// and, for the arrays of level0_mask, level 0 (rescale(., 1.0), tadataka/vo/dvo/__init__.py:144-148).  Every level
// is resampled from the full-resolution frame, exactly as _estimate_at rescales the ORIGINAL I0/D0/I1/W0.
static tdk_status build_pyramid_of(tdk_dvo *h, unsigned arrays) {
    const tdk_dvo::Level &S = h->lv[0];
    if (!h->with_w) arrays &= 7u;
    TDK_TRY(after_uploads(h));
    if (arrays == 0u) return TDK_OK;
    auto dst_of = [&](const tdk_dvo::Level &L, int k) -> double * { return k == 0 ? L.I0 : k == 1 ? L.D0 : k == 2 ? L.I1 : L.W0; };
    std::vector<double> storage((size_t)kMaxLevels * 2 * (2 * tdk::pyramid_max_radius() + 1));
    // a level's description for a subset `idx` of the arrays
    auto describe = [&](int l, const int *idx, int n, tdk::PyramidLevelDesc *d) {
        const tdk_dvo::Level &L = h->lv[l];
        for (int k = 0; k < 4; k++) d->dst[k] = k < n ? dst_of(L, idx[k]) : nullptr;
        d->stride = L.stride; d->H = L.H; d->W = L.W;
        const tdk_dvo::Plan &P = h->plan[l];
        if (P.set) {
            d->mx = tdk::affine_axis(P.map[0], P.map[1]);
            d->my = tdk::affine_axis(P.map[2], P.map[3]);
            d->wr = P.wr.empty() ? nullptr : P.wr.data(); d->Rr = (int)(P.wr.size() / 2);
            d->wc = P.wc.empty() ? nullptr : P.wc.data(); d->Rc = (int)(P.wc.size() / 2);
        } else {
            tdk::ideal_level_plan(d, S.H, S.W, h->anti_aliasing && l > 0,
                                  storage.data() + (size_t)l * 2 * (2 * tdk::pyramid_max_radius() + 1));
        }
    };
    if (h->d_aa_weights == nullptr) {
        TDK_HIP(hipMalloc(&h->d_aa_weights, tdk::pyramid_weight_doubles(h->n_levels) * sizeof(double)));
        h->weights_dirty = true;
    }
    if (h->clip && h->d_clip == nullptr) {
        const size_t bytes = tdk::pyramid_clip_bytes((int64_t)h->n_pairs * 4, h->n_levels);
        TDK_HIP(hipMalloc(&h->d_clip, bytes));
        TDK_HIP(hipMemsetAsync(h->d_clip, 0, bytes, h->stream));   // (every build resets what it uses; never raw memory)
        h->clip_clean = 0;
    }
    const int use_stream = tdk::option(TDK_OPT_PYRAMID_STREAM);   // 0: never, 1 (default): large batches, 2: always
    // two groups of arrays: those with a level 0 of their own (sources: the uploads, levels 0 .. n - 1) and the rest
    // (sources: level 0 = the upload, levels 1 .. n - 1); the device kernels are stored per level: slot l of the
    // weight buffer is level l in both groups
    // (the slots are shared by the two groups: "left clean" is only tracked when a build is one launch group)
    int n_groups_used = 0;
    for (int group = 0; group < 2; group++) {
        int n = 0;
        for (int k = 0; k < 4; k++)
            if (((arrays >> k) & 1u) && (((h->level0_mask >> k) & 1u) != 0) == (group == 0)) n++;
        if (n > 0 && h->n_levels - (group == 0 ? 0 : 1) > 0) n_groups_used++;
    }
    if (n_groups_used != 1) h->clip_clean = 0;
    for (int group = 0; group < 2; group++) {
        int sel[4], n_sel = 0;
        for (int k = 0; k < 4; k++) {
            if (!((arrays >> k) & 1u)) continue;
            const bool own0 = ((h->level0_mask >> k) & 1u) != 0;
            if (own0 == (group == 0)) sel[n_sel++] = k;
        }
        if (n_sel == 0) continue;
        const int first = group == 0 ? 0 : 1;
        const int n_out = h->n_levels - first;
        if (n_out <= 0) continue;
        const double *srcs[4];
        for (int k = 0; k < 4; k++) srcs[k] = k < n_sel ? upload_ptr(h, sel[k]) : nullptr;
        tdk::PyramidLevelDesc lv[kMaxLevels];
        for (int l = first; l < h->n_levels; l++) describe(l, sel, n_sel, &lv[l - first]);
        double *weights = h->d_aa_weights + tdk::pyramid_weight_doubles(first);
        TDK_TRY(tdk::launch_pyramid(srcs, n_sel, S.H, S.W, S.stride, n_out, lv, h->n_pairs, weights, h->weights_dirty,
                                    h->clip ? h->d_clip : nullptr, use_stream, h->stream,
                                    n_groups_used == 1 ? &h->clip_clean : nullptr));
    }
    h->weights_dirty = false;
    return TDK_OK;
}

tdk_status tdk_dvo_build_pyramid(tdk_dvo *h) {
    TDK_API_GUARD;
    TDK_REQUIRE(h != nullptr, "handle is NULL");
    return build_pyramid_of(h, 15u);
}

tdk_status tdk_dvo_build_pyramid_arrays(tdk_dvo *h, unsigned int arrays) {
    TDK_API_GUARD;
    TDK_REQUIRE(h != nullptr, "handle is NULL");
    TDK_REQUIRE(arrays != 0u && arrays <= 15u, "arrays: bit 0 I0, bit 1 D0, bit 2 I1, bit 3 W0");
    TDK_REQUIRE(h->with_w || !(arrays & 8u), "no weight map in this batch");
    return build_pyramid_of(h, arrays);
}

tdk_status tdk_dvo_level_shape(tdk_dvo *h, int level, int *height, int *width) {
    TDK_API_GUARD;
    TDK_TRY(check_level(h, level));
    if (height) *height = h->lv[level].H;
    if (width) *width = h->lv[level].W;
    return TDK_OK;
}

tdk_status tdk_dvo_download(tdk_dvo *h, int pair, int level, int which, double *out) {
    TDK_API_GUARD;
    TDK_TRY(check_level(h, level));
    TDK_REQUIRE(out && pair >= 0 && pair < h->n_pairs && which >= 0 && which <= 3, "bad argument");
    const tdk_dvo::Level &L = h->lv[level];
    const double *src = which == 0 ? L.I0 : which == 1 ? L.D0 : which == 2 ? L.I1 : L.W0;
    TDK_REQUIRE(src != nullptr, "no weight map in this batch");
    TDK_TRY(after_uploads(h));
    TDK_HIP(hipMemcpyAsync(out, src + (int64_t)pair * L.stride, (size_t)L.N * sizeof(double),
                           hipMemcpyDeviceToHost, h->stream));
    TDK_HIP(hipStreamSynchronize(h->stream));
    return TDK_OK;
}

tdk_status tdk_dvo_evaluate(tdk_dvo *h, int level, const double *camera0, const double *camera1,
                            const double *poses12, int weight_mode, double *Hout, double *bout,
                            int64_t *n_update, double *sum_sq, int64_t *n_error) {
    TDK_API_GUARD;
    TDK_TRY(check_level(h, level));
    TDK_REQUIRE(camera0 && camera1 && poses12, "null pointer");
    TDK_TRY(check_weight_mode(h, weight_mode));
    TDK_TRY(upload_params(h, camera0, camera1));
    TDK_HIP(hipMemcpyAsync(h->d_poses_in, poses12, sizeof(double) * 12 * h->n_pairs, hipMemcpyHostToDevice,
                           h->stream));
    TDK_TRY(launch_eval(h, level, h->d_poses_in, nullptr, nullptr, nullptr, weight_mode));
    TDK_TRY(launch_reduce(h, level, 0, 0));
    void *stage;
    TDK_TRY(tdk::pinned(1, sizeof(double) * kAccPad * h->n_pairs, &stage));
    TDK_HIP(hipMemcpyAsync(stage, h->d_results, sizeof(double) * kAccPad * h->n_pairs, hipMemcpyDeviceToHost,
                           h->stream));
    TDK_HIP(hipStreamSynchronize(h->stream));
    const double *r = (const double *)stage;
    for (int i = 0; i < h->n_pairs; i++) {
        const double *ri = r + (size_t)kAccPad * i;
        if (Hout) memcpy(Hout + 21 * i, ri, sizeof(double) * 21);
        if (bout) memcpy(bout + 6 * i, ri + 21, sizeof(double) * 6);
        if (sum_sq) sum_sq[i] = ri[27];
        if (n_update) n_update[i] = (int64_t)ri[28];
        if (n_error) n_error[i] = (int64_t)ri[29];
    }
    if (h->profiling) TDK_TRY(collect_profile(h));
    return TDK_OK;
}

tdk_status tdk_dvo_photometric_error(tdk_dvo *h, int level, const double *camera0, const double *camera1,
                                     const double *poses12, double *sum_sq, int64_t *n_error) {
    TDK_API_GUARD;
    TDK_TRY(check_level(h, level));
    TDK_REQUIRE(camera0 && camera1 && poses12, "null pointer");
    TDK_TRY(upload_params(h, camera0, camera1));
    const int n = h->n_pairs;
    if (!h->d_mode_probe) {   // every pair: error only
        std::vector<int> m((size_t)n, (int)MODE_PROBE);
        TDK_HIP(hipMalloc(&h->d_mode_probe, sizeof(int) * n));
        TDK_HIP(hipMemcpy(h->d_mode_probe, m.data(), sizeof(int) * n, hipMemcpyHostToDevice));
    }
    TDK_HIP(hipMemcpyAsync(h->d_poses_in, poses12, sizeof(double) * 12 * n, hipMemcpyHostToDevice, h->stream));
    // the error does not depend on the weights (metric.py:13-39): the unweighted kernel
    TDK_TRY(launch_eval(h, level, h->d_poses_in, nullptr, h->d_mode_probe, nullptr, TDK_W_NONE, true));
    if (h->profiling && (level == 0 || h->profiling == 2) && h->ev_used >= 2) {   // what the launch just timed evaluated: probes
        h->ev_round[h->ev_used - 2] = 0;
        h->ev_round[h->ev_used - 1] = n;
    }
    TDK_TRY(launch_reduce(h, level, 0, 0));
    void *stage;
    TDK_TRY(tdk::pinned(1, sizeof(double) * kAccPad * n, &stage));
    TDK_HIP(hipMemcpyAsync(stage, h->d_results, sizeof(double) * kAccPad * n, hipMemcpyDeviceToHost, h->stream));
    TDK_HIP(hipStreamSynchronize(h->stream));
    const double *r = (const double *)stage;
    for (int i = 0; i < n; i++) {
        if (sum_sq) sum_sq[i] = r[(size_t)kAccPad * i + 27];
        if (n_error) n_error[i] = (int64_t)r[(size_t)kAccPad * i + 29];
    }
    if (h->profiling) TDK_TRY(collect_profile(h));
    return TDK_OK;
}

// the final poses and warnings of a call -> h_io, wait, -> the caller's array
static tdk_status publish_and_wait(tdk_dvo *h, double *poses12) {
    const int n = h->n_pairs;
    k_publish<<<(n + 255) / 256, 256, 0, h->stream>>>(h->ls, h->d_io + 12 * (size_t)n, (int *)(h->d_io + 24 * (size_t)n), n);
    TDK_LAUNCH_CHECK();
    TDK_HIP(hipStreamSynchronize(h->stream));
    memcpy(poses12, h->h_io + 12 * (size_t)n, sizeof(double) * 12 * n);
    memcpy(h->host_warn.data(), h->h_io + 24 * (size_t)n, sizeof(int) * n);
    return TDK_OK;
}

tdk_status tdk_dvo_estimate_level(tdk_dvo *h, int level, const double *camera0, const double *camera1,
                                  double *poses12, int weight_mode, int max_iter, int *n_evals) {
    TDK_API_GUARD;
    TDK_TRY(check_level(h, level));
    TDK_REQUIRE(camera0 && camera1 && poses12 && max_iter >= 0, "bad argument");
    TDK_TRY(check_weight_mode(h, weight_mode));
    TDK_TRY(upload_params(h, camera0, camera1));
    TDK_HIP(hipMemcpyAsync(h->d_poses_in, poses12, sizeof(double) * 12 * h->n_pairs, hipMemcpyHostToDevice,
                           h->stream));
    int n = h->n_pairs;
    TDK_HIP(hipMemsetAsync(h->ls.warn, 0, sizeof(int) * n, h->stream));
    h->count_error_px = h->count_update_px = 0;
    h->ls.fuse_first = 1;   // a level on its own starts from the caller's prior
    k_loop_init<<<(n + 255) / 256, 256, 0, h->stream>>>(h->ls, h->d_poses_in, n);
    TDK_LAUNCH_CHECK();
    TDK_TRY(run_level(h, level, weight_mode, max_iter, nullptr));
    TDK_HIP(hipMemcpyAsync(poses12, h->ls.pose, sizeof(double) * 12 * n, hipMemcpyDeviceToHost, h->stream));
    TDK_HIP(hipMemcpyAsync(h->host_warn.data(), h->ls.warn, sizeof(int) * n, hipMemcpyDeviceToHost, h->stream));
    if (n_evals)
        TDK_HIP(hipMemcpyAsync(n_evals, h->ls.n_evals, sizeof(int) * n, hipMemcpyDeviceToHost, h->stream));
    TDK_HIP(hipStreamSynchronize(h->stream));
    if (h->profiling) TDK_TRY(collect_profile(h));
    return TDK_OK;
}

tdk_status tdk_dvo_estimate(tdk_dvo *h, const double *camera0, const double *camera1, double *poses12,
                            int weight_mode, int max_iter, int64_t *pixel_evals) {
    TDK_API_GUARD;
    TDK_REQUIRE(h && camera0 && camera1 && poses12 && max_iter >= 0, "bad argument");
    TDK_TRY(check_weight_mode(h, weight_mode));
    TDK_TRY(upload_params(h, camera0, camera1));
    int n = h->n_pairs;
    memcpy(h->h_io, poses12, sizeof(double) * 12 * n);     // read by k_loop_init over the bus (see tdk_dvo::h_io)
    if (pixel_evals) *pixel_evals = 0;
    h->count_error_px = h->count_update_px = 0;
    if (chain_applies(h, weight_mode)) return chain_estimate(h, weight_mode, max_iter, poses12, pixel_evals);
    TDK_HIP(hipMemsetAsync(h->ls.warn, 0, sizeof(int) * n, h->stream));
    for (int level = h->n_levels - 1; level >= 0; level--) {
        // the prior of a level is the result of the coarser one (:131-134), already in ls.pose
        h->ls.fuse_first = level == h->n_levels - 1;
        k_loop_init<<<(n + 255) / 256, 256, 0, h->stream>>>(h->ls, level == h->n_levels - 1 ? h->d_io : h->ls.pose, n);
        TDK_LAUNCH_CHECK();
        TDK_TRY(run_level(h, level, weight_mode, max_iter, pixel_evals));
    }
    TDK_TRY(publish_and_wait(h, poses12));
    if (h->profiling) TDK_TRY(collect_profile(h));
    return TDK_OK;
}

tdk_status tdk_dvo_get_tukey_fallbacks(tdk_dvo *h, int64_t *pairs) {
    TDK_API_GUARD;
    TDK_REQUIRE(h && pairs, "null pointer");
    *pairs = 0;
    if (!h->d_tk_fallback) return TDK_OK;
    unsigned int v = 0;
    TDK_HIP(hipMemcpyAsync(&v, h->d_tk_fallback, sizeof(v), hipMemcpyDeviceToHost, h->stream));
    TDK_HIP(hipStreamSynchronize(h->stream));
    *pairs = (int64_t)v;
    return TDK_OK;
}

tdk_status tdk_dvo_set_student_passes(tdk_dvo *h, int mode) {
    TDK_API_GUARD;
    TDK_REQUIRE(h != nullptr, "handle is NULL");
    TDK_REQUIRE(mode >= 0 && mode <= 2, "mode must be 0 (Taylor passes), 1 (sequential) or 2 (sequential, IEEE divisions)");
    h->student_mode = mode;
    return TDK_OK;
}

tdk_status tdk_dvo_get_robust_scale(tdk_dvo *h, double *scale) {
    TDK_API_GUARD;
    TDK_REQUIRE(h && scale, "null pointer");
    TDK_REQUIRE(h->d_wscale != nullptr, "no robust evaluation has run on this batch");
    TDK_HIP(hipMemcpyAsync(scale, h->d_wscale, sizeof(double) * h->n_pairs, hipMemcpyDeviceToHost, h->stream));
    TDK_HIP(hipStreamSynchronize(h->stream));
    return TDK_OK;
}

tdk_status tdk_dvo_get_student_fallbacks(tdk_dvo *h, int64_t *pairs) {
    TDK_API_GUARD;
    TDK_REQUIRE(h && pairs, "null pointer");
    *pairs = 0;
    if (!h->d_st_redo) return TDK_OK;
    unsigned int v = 0;
    TDK_HIP(hipMemcpyAsync(&v, h->d_st_redo + 1, sizeof(v), hipMemcpyDeviceToHost, h->stream));
    TDK_HIP(hipStreamSynchronize(h->stream));
    *pairs = (int64_t)v;
    return TDK_OK;
}

tdk_status tdk_dvo_get_student_redos(tdk_dvo *h, int64_t *pairs) {
    TDK_API_GUARD;
    TDK_REQUIRE(h && pairs, "null pointer");
    *pairs = 0;
    if (!h->d_st_redo) return TDK_OK;
    unsigned int v = 0;
    TDK_HIP(hipMemcpyAsync(&v, h->d_st_redo, sizeof(v), hipMemcpyDeviceToHost, h->stream));
    TDK_HIP(hipStreamSynchronize(h->stream));
    *pairs = (int64_t)v;
    return TDK_OK;
}

tdk_status tdk_dvo_get_counts(tdk_dvo *h, int64_t *error_pixels, int64_t *update_pixels) {
    TDK_API_GUARD;
    TDK_REQUIRE(h != nullptr, "handle is NULL");
    if (error_pixels) *error_pixels = h->count_error_px;
    if (update_pixels) *update_pixels = h->count_update_px;
    return TDK_OK;
}

tdk_status tdk_dvo_get_warnings(tdk_dvo *h, int *too_large) {
    TDK_API_GUARD;
    TDK_REQUIRE(h && too_large, "null pointer");
    memcpy(too_large, h->host_warn.data(), sizeof(int) * (size_t)h->n_pairs);
    return TDK_OK;
}

tdk_status tdk_dvo_set_anti_aliasing(tdk_dvo *h, int enabled) {
    TDK_API_GUARD;
    TDK_REQUIRE(h != nullptr, "handle is NULL");
    TDK_REQUIRE(enabled == 0 || enabled == 1, "enabled must be 0 or 1");
    h->anti_aliasing = enabled != 0;
    h->weights_dirty = true;
    return TDK_OK;
}

tdk_status tdk_dvo_set_option(tdk_dvo *h, int option, int value) {
    TDK_API_GUARD;
    TDK_REQUIRE(h != nullptr, "handle is NULL");
    switch (option) {
        case TDK_DVO_OPT_CHAIN:
            TDK_REQUIRE(value == 0 || value == 1, "TDK_DVO_OPT_CHAIN: 0 or 1");
            h->opt_chain = value;
            return TDK_OK;
        case TDK_DVO_OPT_TUKEY:
            TDK_REQUIRE(value >= 0 && value <= 2, "TDK_DVO_OPT_TUKEY: 0, 1 or 2");
            h->opt_tukey = value;
            return TDK_OK;
        default:
            tdk::set_error("invalid argument: unknown option %d", option);
            return TDK_ERR_INVALID_ARGUMENT;
    }
}

tdk_status tdk_dvo_set_level_plan(tdk_dvo *h, int level, const double *map, const double *w_rows, int radius_rows,
                                  const double *w_cols, int radius_cols) {
    TDK_API_GUARD;
    TDK_TRY(check_level(h, level));
    tdk_dvo::Plan &P = h->plan[level];
    h->weights_dirty = true;
    h->clip_clean = 0;
    if (map == nullptr) {
        P.set = false;
        P.wr.clear(); P.wc.clear();
        return TDK_OK;
    }
    TDK_REQUIRE(radius_rows >= 0 && radius_cols >= 0 && radius_rows <= tdk::pyramid_max_radius() &&
                radius_cols <= tdk::pyramid_max_radius(), "kernel radius out of range");
    TDK_REQUIRE((radius_rows == 0 || w_rows) && (radius_cols == 0 || w_cols), "kernel is NULL");
    // (a one-pixel output axis: skimage's estimate of the degenerate corner set gives scale 0 and only the offset is used)
    TDK_REQUIRE((map[0] > 0.0 || (h->lv[level].W == 1 && map[0] == 0.0)) &&
                    (map[2] > 0.0 || (h->lv[level].H == 1 && map[2] == 0.0)),
                "the map's scales must be positive");
    for (int k = 0; k < 4; k++) P.map[k] = map[k];
    P.wr.assign(w_rows, w_rows + (radius_rows > 0 ? 2 * radius_rows + 1 : 0));
    P.wc.assign(w_cols, w_cols + (radius_cols > 0 ? 2 * radius_cols + 1 : 0));
    P.set = true;
    return TDK_OK;
}

tdk_status tdk_dvo_set_rescale_options(tdk_dvo *h, unsigned int level0_arrays, int clip) {
    TDK_API_GUARD;
    TDK_REQUIRE(h != nullptr, "handle is NULL");
    TDK_REQUIRE(level0_arrays <= 15u, "level0_arrays: bit 0 I0, bit 1 D0, bit 2 I1, bit 3 W0");
    if (!h->with_w) level0_arrays &= 7u;
    TDK_TRY(after_uploads(h));
    const tdk_dvo::Level &L = h->lv[0];
    const size_t bytes = ((size_t)L.stride * h->n_pairs + 4 * kBlock) * sizeof(double);
    for (int k = 0; k < 4; k++) {
        double *lvl0 = k == 0 ? L.I0 : k == 1 ? L.D0 : k == 2 ? L.I1 : L.W0;
        const bool want = ((level0_arrays >> k) & 1u) != 0;
        if (want && !h->raw[k]) {
            // frames uploaded so far move to the new upload array
            TDK_HIP(hipMalloc(&h->raw[k], bytes));
            TDK_HIP(hipMemcpyAsync(h->raw[k], lvl0, bytes, hipMemcpyDeviceToDevice, h->stream));
        } else if (!want && h->raw[k]) {
            TDK_HIP(hipMemcpyAsync(lvl0, h->raw[k], bytes, hipMemcpyDeviceToDevice, h->stream));
            TDK_HIP(hipStreamSynchronize(h->stream));
            (void)hipFree(h->raw[k]);
            h->raw[k] = nullptr;
        }
    }
    h->level0_mask = level0_arrays;
    h->clip = clip != 0;
    h->clip_clean = 0;                      // (the next build's slot layout may differ: it resets what it uses)
    return TDK_OK;
}

tdk_status tdk_dvo_get_stream(tdk_dvo *h, void **stream_out) {
    TDK_API_GUARD;
    TDK_REQUIRE(h != nullptr && stream_out != nullptr, "null pointer");
    *stream_out = (void *)h->stream;
    return TDK_OK;
}

tdk_status tdk_dvo_set_profiling(tdk_dvo *h, int enabled) {
    TDK_API_GUARD;
    TDK_REQUIRE(h != nullptr, "handle is NULL");
    h->profiling = enabled == 2 ? 2 : (enabled != 0 ? 1 : 0);
    h->ev_used = 0;
    memset(h->prof_ms, 0, sizeof(h->prof_ms)); memset(h->prof_launches, 0, sizeof(h->prof_launches));
    memset(h->prof_pixels, 0, sizeof(h->prof_pixels));
    return TDK_OK;
}

tdk_status tdk_dvo_get_profile(tdk_dvo *h, int64_t *launches, double *total_ms, int64_t *pixels) {
    TDK_API_GUARD;
    TDK_REQUIRE(h != nullptr, "handle is NULL");
    if (launches) *launches = h->prof_launches[0][0];
    if (total_ms) *total_ms = h->prof_ms[0][0];
    if (pixels) *pixels = h->prof_pixels[0][0];
    return TDK_OK;
}

tdk_status tdk_dvo_get_profile_kind(tdk_dvo *h, int kind, int64_t *launches, double *total_ms, int64_t *pixels) {
    TDK_API_GUARD;
    TDK_REQUIRE(h != nullptr && kind >= 0 && kind <= 2, "bad argument");
    if (launches) *launches = h->prof_launches[0][kind];
    if (total_ms) *total_ms = h->prof_ms[0][kind];
    if (pixels) *pixels = h->prof_pixels[0][kind];
    return TDK_OK;
}

tdk_status tdk_dvo_get_profile_level(tdk_dvo *h, int level, int kind, int64_t *launches, double *total_ms,
                                     int64_t *pixels) {
    TDK_API_GUARD;
    TDK_REQUIRE(h != nullptr && kind >= 0 && kind <= 2, "bad argument");
    TDK_TRY(check_level(h, level));
    if (launches) *launches = h->prof_launches[level][kind];
    if (total_ms) *total_ms = h->prof_ms[level][kind];
    if (pixels) *pixels = h->prof_pixels[level][kind];
    return TDK_OK;
}

}  // extern "C"
