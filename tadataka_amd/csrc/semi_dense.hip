// semi_dense.hip -- rust_bindings.semi_dense on the MI355X: increment_age,
// propagate, update_depth / estimate_debug_, the Sobel maps they use, the
// post-steps regularize / fusion -- as 1:1 host-pointer entries and
// as a device-resident session (tdk_sd) that runs the mapping step of
// examples/semi_dense_vo.py:182-199 for a batch of independent tracks per launch.
//
// Compiled with -ffp-contract=off: the warped target pixel is an *index* and the
// per-pixel result a discrete flag, so every + - * / sqrt is kept as one IEEE
// rounding in the reference's operation order (bit-exact against the oracle).
//
// The two forward-warp scatters are order dependent in the reference (a raster
// loop): increment_age keeps the LAST raster writer (src/semi_dense/age.rs:18-29)
// and propagate folds colliding sources SEQUENTIALLY in raster order with a
// non-associative rule (src/semi_dense/propagation.rs:21-46,59-82).  Both warp
// the same source pixel with the same transform to the same target, so ONE
// scatter serves both: sources are threaded onto a per-target linked list with
// atomicExch; one thread per target then walks its list in increasing source
// index -- the largest index is the age winner, the fold in that order is the
// propagated hypothesis.  The warp of a source is recomputed in the fold from
// depth0 (a 16 B gather) instead of being written out and read back (32 B).
//
// update_depth is split so that the heavy per-pixel search only runs on full
// waves: k_ud_classify streams age / prior maps, settles NotProcessed and
// check_args pixels and compacts the rest (wave ballot + block prefix, one
// atomic per block and track); k_ud_estimate runs `estimate` with one lane per
// live pixel.  The key frame's Sobel value is computed in place from its 3x3
// neighbourhood (ImageGradient::get is only ever asked at the integer key
// pixel, src/semi_dense/variance.rs:30-52) -- no Sobel maps in HBM.
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

// ---- Sobel (src/gradient.rs:4-26, src/convolution.rs:29-52) -----------------
__device__ __forceinline__ void sobel_at(const double *__restrict__ img, int H, int W, int x, int y, double &sx,
                                         double &sy) {
    const double kx[9] = {1., 0., -1., 2., 0., -2., 1., 0., -1.};
    const double ky[9] = {1., 2., 1., 0., 0., 0., -1., -2., -1.};
    sx = 0.0;
    sy = 0.0;
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
}

__global__ __launch_bounds__(kBlock) void k_sobel(const double *__restrict__ img, int H, int W,
                                                  double *__restrict__ gx, double *__restrict__ gy) {
    int64_t N = (int64_t)H * W;
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < N; i += (int64_t)gridDim.x * kBlock) {
        int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
        double sx, sy;
        sobel_at(img, H, W, x, y, sx, sy);
        gx[i] = sx;
        gy[i] = sy;
    }
}

// ---- increment_age (src/semi_dense/age.rs:6-32) + propagate (propagation.rs) ----
struct TrackWarp {   // per track and step
    double T10[16];
    double cam0[4], cam1[4];
    uint64_t age_cap;   // increment_age saturates here (session ring, tdk_sd_set_age_policy); ~0 = never
};

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

// 1-D grids of 8 ceil(n_tracks / 8) nb blocks: workgroups are dealt to the 8 XCDs round-robin, so XCD k
// takes tracks k, k + 8, ... one after the other and all blocks of a track share one L2 (the
// list walk of the fold and the forward warp of the scatter touch neighbouring lines).
__device__ __forceinline__ bool xcd_major_track(int nb, int n_tracks, int &track, int &blk,
                                                unsigned b = 0xffffffffu) {
    if (b == 0xffffffffu) b = blockIdx.x;
    const int xcd = b & 7, q = b >> 3;
    track = (q / nb) * 8 + xcd;
    blk = q - (q / nb) * nb;
    return track < n_tracks;
}

// Per-target bookkeeping of the forward warp, one set per track (ints, `stride` apart per track):
//   cnt[t]          number of in-range sources that landed on target t          (zeroed per step)
//   slots[t][0..3]  the first kSlots of them, in arrival order                    (never initialised)
//   ovf[t], next[i] sources beyond kSlots: a chain threaded through next[] whose newest entry is
//                   ovf[t]; it has cnt - kSlots entries, so neither array needs initialising
// Two sources per target cover > 99 % of a frame and four practically all of it; the chain keeps
// the operators exact for any warp (a zoom-out that folds a dozen sources onto every target).
constexpr int kSlots = 4;
struct WarpLists {
    int *cnt, *slots, *ovf, *next;
    int *rng;   // [kRngStride n_tracks] displacement boxes of the gather path (k_sd_targets), a cache line per track
};
constexpr size_t kWarpListInts = 1 + kSlots + 1 + 1;   // ints per pixel and track behind a WarpLists
constexpr int kRngStride = 32;                         // ints per track in WarpLists::rng (128 bytes: atomics of different tracks do not share a line)
// one helper sizes what carve_lists carves: seven regions of m ints (m = stride * n_tracks rounded up to 4,
// which keeps the slot rows 16-byte aligned), the boxes, 16 ints of padding
__host__ __device__ inline int64_t warp_list_region(int64_t stride, int n_tracks) { return (stride * n_tracks + 3) & ~(int64_t)3; }
inline size_t warp_list_bytes(int64_t stride, int n_tracks) {
    return sizeof(int) * (kWarpListInts * (size_t)warp_list_region(stride, n_tracks) + kRngStride * (size_t)n_tracks + 16 + 32);
}

__host__ __device__ inline WarpLists carve_lists(int *buf, int64_t stride, int n_tracks) {
    WarpLists L;
    const int64_t m = warp_list_region(stride, n_tracks);
    L.cnt = buf;
    L.slots = buf + m;
    L.ovf = buf + m * (1 + kSlots);
    L.next = buf + m * (2 + kSlots);
    L.rng = buf + ((m * (3 + kSlots) + 31) & ~(int64_t)31);
    return L;
}

// (the gather path, below; rng[kRngStride track + 4] holds the window limit of the gather kernel that ran)
__device__ __forceinline__ bool slot_path_wanted(const int *__restrict__ rng, int track);
__device__ __forceinline__ bool any_slot_track(const int *__restrict__ rng, int n_tracks);

// Forward warp of every source pixel of every track; an in-range source claims the next slot of its
// target (one returning atomic), writes its index there, or joins the overflow chain.
__global__ __launch_bounds__(kBlock) void k_sd_scatter(int H, int W, const TrackWarp *__restrict__ tw,
                                                       const double *__restrict__ depth0, int64_t stride,
                                                       WarpLists lists, int nb, int n_tracks,
                                                       const int *__restrict__ rng, unsigned vgrid) {
    if (!any_slot_track(rng, n_tracks)) return;
    for (unsigned vb = blockIdx.x; vb < vgrid; vb += gridDim.x) {
    int track, blk;
    if (!xcd_major_track(nb, n_tracks, track, blk, vb)) continue;
    if (!slot_path_wanted(rng, track)) continue;
    const TrackWarp &t = tw[track];
    const Cam c0{t.cam0[0], t.cam0[1], t.cam0[2], t.cam0[3]}, c1{t.cam1[0], t.cam1[1], t.cam1[2], t.cam1[3]};
    const int N = H * W;
    const double *__restrict__ d0 = depth0 + (int64_t)track * stride;
    int *__restrict__ cn = lists.cnt + (int64_t)track * stride;
    int *__restrict__ sl = lists.slots + (int64_t)track * stride * kSlots;
    int *__restrict__ oh = lists.ovf + (int64_t)track * stride;
    int *__restrict__ nx = lists.next + (int64_t)track * stride;
    for (int i = blk * kBlock + threadIdx.x; i < N; i += nb * kBlock) {
        int y0 = i / W, x0 = i - y0 * W;
        double ux, uy, d1;
        tdk::perspective_warp(t.T10, c0, c1, (double)x0, (double)y0, d0[i], ux, uy, d1);
        if (!tdk::in_range(ux, uy, H, W)) continue;
        const int tg = (int)uy * W + (int)ux;  // `as usize`: truncation
        const int k = atomicAdd(&cn[tg], 1);
        if (k < kSlots) sl[(int64_t)tg * kSlots + k] = i;
        else nx[i] = atomicExch(&oh[tg], i);
    }
    }
}

__device__ __forceinline__ void sort2(int &a, int &b) {
    const int lo = min(a, b), hi = max(a, b);
    a = lo; b = hi;
}

// One thread per target pixel: its sources in increasing source index (= raster order of the reference's loop).
//   AGE : age1 = age0[last raster writer] + 1, untouched = 0            (age.rs:18-31)
//   PROP: sequential fold of (depth1, variance1) with handle_collision,
//         misses get the defaults                                         (propagation.rs:59-89)
// The count and the four slots of a target are one 4-byte and one 16-byte coalesced load; up to four
// sources are ordered by a five-exchange network in registers; the rare longer lists go through the
// selection loop over slots + chain.
template <bool AGE, bool PROP>
__global__ __launch_bounds__(kBlock) void k_sd_fold(int H, int W, const TrackWarp *__restrict__ tw, WarpLists lists,
                                                    const uint64_t *__restrict__ age0,
                                                    const double *__restrict__ depth0,
                                                    const double *__restrict__ var0, int64_t stride,
                                                    double default_depth, double default_variance, double bias,
                                                    uint64_t *__restrict__ age1, double *__restrict__ depth1,
                                                    double *__restrict__ var1, int nb, int n_tracks,
                                                    const int *__restrict__ rng, unsigned vgrid) {
    if (!any_slot_track(rng, n_tracks)) return;
    for (unsigned vb = blockIdx.x; vb < vgrid; vb += gridDim.x) {
    int track, blk;
    if (!xcd_major_track(nb, n_tracks, track, blk, vb)) continue;
    if (!slot_path_wanted(rng, track)) continue;
    const TrackWarp &t = tw[track];
    const Cam c0{t.cam0[0], t.cam0[1], t.cam0[2], t.cam0[3]}, c1{t.cam1[0], t.cam1[1], t.cam1[2], t.cam1[3]};
    const int N = H * W;
    const int64_t base = (int64_t)track * stride;
    const int *__restrict__ cn = lists.cnt + base;
    const int4 *__restrict__ sl = reinterpret_cast<const int4 *>(lists.slots + base * kSlots);
    const int *__restrict__ oh = lists.ovf + base;
    const int *__restrict__ nx = lists.next + base;
    for (int tg = blk * kBlock + threadIdx.x; tg < N; tg += nb * kBlock) {
        const int k = cn[tg];
        double d = default_depth, v = default_variance;
        int last = -1;
        bool have = false;
        auto take = [&](int src) {      // the next source in raster order
            if (PROP) {
                int y0 = src / W, x0 = src - y0 * W;
                double sd0 = depth0[base + src], ux, uy, sd1;
                tdk::perspective_warp(t.T10, c0, c1, (double)x0, (double)y0, sd0, ux, uy, sd1);
                double sv1 = propagate_variance(sd0, sd1, var0[base + src], bias);
                if (!have) { d = sd1; v = sv1; have = true; }
                else {
                    double nd, nv;
                    handle_collision(sd1, d, sv1, v, nd, nv);
                    d = nd; v = nv;
                }
            }
            last = src;
        };
        if (k > 0) {
            const int4 q = sl[tg];
            int s0 = q.x, s1 = k > 1 ? q.y : 0x7fffffff, s2 = k > 2 ? q.z : 0x7fffffff, s3 = k > 3 ? q.w : 0x7fffffff;
            if (k <= kSlots) {
                if (k > 1) {      // most targets have one source: skip the network
                    sort2(s0, s1); sort2(s2, s3); sort2(s0, s2); sort2(s1, s3); sort2(s1, s2);
                }
                take(s0);
                if (k > 1) take(s1);
                if (k > 2) take(s2);
                if (k > 3) take(s3);
            } else {
                // slots + chain (k - kSlots entries from ovf[tg]): repeatedly the smallest index above `last`
                for (int n = 0; n < k; n++) {
                    int best = 0x7fffffff;
                    if (s0 > last && s0 < best) best = s0;
                    if (s1 > last && s1 < best) best = s1;
                    if (s2 > last && s2 < best) best = s2;
                    if (s3 > last && s3 < best) best = s3;
                    int j = oh[tg];
                    for (int m = kSlots; m < k; m++) {
                        if (j > last && j < best) best = j;
                        j = nx[j];
                    }
                    take(best);
                }
            }
        }
        if (AGE) {
            uint64_t a = 0;
            if (last >= 0) {
                a = age0[base + last] + 1;
                a = a > t.age_cap ? t.age_cap : a;
            }
            age1[base + tg] = a;
        }
        if (PROP) { depth1[base + tg] = d; var1[base + tg] = v; }
    }
    }
}

// ---------------------------------------------------------------------------
// The forward warp as a GATHER (round 4).  The scatter above is bound by its returning atomics
// (19.7 M for 64 VGA tracks) and the fold by scattered slot reads.  But the displacement field of a
// frame-to-frame warp is smooth: if every in-range source s lands on t(s) = s + d(s) with the integer
// displacement d(s) inside a small box [dmin, dmax] (per track), then the sources of a target t are all
// inside t - box, and scanning that box in raster order IS the reference's fold order -- no slots, no
// chains, no atomics per pixel:
//   k_sd_targets  one thread per source: warp, write the target index (4 B/px, coalesced; -1 = out of
//                 range), and the track's displacement box by four atomic maxima per block;
//   k_sd_gather2  a block takes 64 x 8 targets; the (64 + rx) x (8 + ry) sources that can reach them vote into
//                 per-target slots in LDS (LDS atomics; the slot algorithm of the scatter, but on chip), every
//                 target orders its <= 4 sources and folds them exactly as k_sd_fold does (same order -> same
//                 bits); the hypothesis (depth1, variance1a) of a source comes from k_sd_targets, where every
//                 lane has one to compute -- in the fold the busiest target of a wave sets the pace.
// A track whose box is too large (kGatherMaxRx x kGatherMaxRy, or too many candidates per target for the
// rare plain scan) is left to the slot path: k_sd_gather2 returns at once for it, the scatter /
// fold launches queued behind return at once for all the others -- decided on the device, no host wait;
// the tracks that fell back are counted (tdk_sd_get_warp_fallbacks).
// ---------------------------------------------------------------------------
constexpr int kGatherTW = 64, kGatherTH = 4;              // targets per block tile (one per thread)
constexpr int kGatherMaxRx = 64, kGatherMaxRy = 12;       // largest displacement spread the window holds
constexpr int kGatherMaxCand = 320;                       // (rx + 1)(ry + 1): candidates scanned per target

// rng[kRngStride track ..]: max(-dx), max(-dy), max(dx), max(dy) over the in-range sources (memset to 0x80808080 before)
__device__ __forceinline__ bool gather_applies(const int *__restrict__ rng, int track, int &dxmin, int &dymin,
                                               int &rx, int &ry) {
    const int nx = rng[kRngStride * track], ny = rng[kRngStride * track + 1], mx = rng[kRngStride * track + 2],
              my = rng[kRngStride * track + 3];
    if (mx < -0x40000000) { dxmin = 0; dymin = 0; rx = -1; ry = -1; return true; }   // no source in range at all
    dxmin = -nx; dymin = -ny;
    rx = mx + nx; ry = my + ny;
    return rx <= kGatherMaxRx && ry <= kGatherMaxRy && (rx + 1) * (ry + 1) <= kGatherMaxCand;
}

// PX = pixels per lane: 2 where W is even (16-byte loads of depth0 / var0 and a 8-byte store of the two target
// indices; a wave covers 128 columns), 1 otherwise.  (8-byte accesses run at 0.54 - 0.70 of the 16-byte rate.)
// STORE = false: only the track's box (the fused gather below warps its window itself); max_win is left in
// rng[.. + 4] for gather_applies of the launches that follow.
template <bool PROP, int PX, bool STORE>
__global__ __launch_bounds__(kBlock) void k_sd_targets(int H, int W, const TrackWarp *__restrict__ tw,
                                                       const double *__restrict__ depth0,
                                                       const double *__restrict__ var0, double bias, int64_t stride,
                                                       int *__restrict__ tgt, double2 *__restrict__ warped,
                                                       int *__restrict__ rng, int nb, int n_tracks, int max_win) {
    int track, blk;
    if (!xcd_major_track(nb, n_tracks, track, blk)) return;
    const TrackWarp &t = tw[track];
    const Cam c0{t.cam0[0], t.cam0[1], t.cam0[2], t.cam0[3]}, c1{t.cam1[0], t.cam1[1], t.cam1[2], t.cam1[3]};
    const double *__restrict__ d0 = depth0 + (int64_t)track * stride;
    int *__restrict__ tg = tgt + (int64_t)track * stride;
    constexpr int TW = kGatherTW * PX;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + kGatherTH - 1) / kGatherTH;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    int ndx = -0x7fffffff, ndy = -0x7fffffff, mdx = -0x7fffffff, mdy = -0x7fffffff;
    const double *__restrict__ v0 = var0 + (int64_t)track * stride;
    const int n_tiles = tiles_x * tiles_y;
    // four tiles per round: their eight loads are issued before the first warp is computed
    for (int tile0 = blk; tile0 < n_tiles; tile0 += 4 * nb) {
        int xs[4], ys[4];
        double dd[4][PX], vv[4][PX];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int tile = tile0 + k * nb;
            const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
            xs[k] = txi * TW + lx * PX; ys[k] = tyi * kGatherTH + ly;
            const bool ok = tile < n_tiles && xs[k] < W && ys[k] < H;      // (PX = 2: W even, both or neither)
            if (!ok) xs[k] = -1;
            const int i = ok ? ys[k] * W + xs[k] : 0;
            if (PX == 2) {
                const double2 a = *reinterpret_cast<const double2 *>(d0 + i);
                dd[k][0] = a.x; dd[k][PX - 1] = a.y;
                if (PROP) {
                    const double2 b = *reinterpret_cast<const double2 *>(v0 + i);
                    vv[k][0] = b.x; vv[k][PX - 1] = b.y;
                }
            } else {
                dd[k][0] = d0[i];
                if (PROP) vv[k][0] = v0[i];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (xs[k] < 0) continue;
            int out[PX];
#pragma unroll
            for (int e = 0; e < PX; e++) {
                const int x0 = xs[k] + e, y0 = ys[k], i = y0 * W + x0;
                double ux, uy, d1;
                tdk::perspective_warp(t.T10, c0, c1, (double)x0, (double)y0, dd[k][e], ux, uy, d1);
                out[e] = -1;
                if (tdk::in_range(ux, uy, H, W)) {
                    const int tx = (int)ux, ty = (int)uy;   // `as usize`: truncation
                    out[e] = ty * W + tx;
                    ndx = max(ndx, x0 - tx); mdx = max(mdx, tx - x0);
                    ndy = max(ndy, y0 - ty); mdy = max(mdy, ty - y0);
                    // propagate: the source's hypothesis in the new frame, computed HERE, where every lane has
                    // one -- in the gather the targets of a wave have 1 - 4 sources each and the warp arithmetic
                    // would run for the busiest lane's count with most lanes idle
                    if (PROP && STORE)
                        warped[(int64_t)track * stride + i] =
                            make_double2(d1, propagate_variance(dd[k][e], d1, PROP ? vv[k][e] : 0.0, bias));
                }
            }
            const int i0 = ys[k] * W + xs[k];
            if (!STORE) continue;
            if (PX == 2) *reinterpret_cast<int2 *>(tg + i0) = make_int2(out[0], out[PX - 1]);
            else tg[i0] = out[0];
        }
    }
    // the track's box: wave maxima, one atomic per wave and bound
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ndx = max(ndx, __shfl_xor(ndx, o)); ndy = max(ndy, __shfl_xor(ndy, o));
        mdx = max(mdx, __shfl_xor(mdx, o)); mdy = max(mdy, __shfl_xor(mdy, o));
    }
    // ... block maxima through LDS, and an atomic only for a bound this block actually extends (after the
    // first blocks of a track almost none does: 1.2 M atomics on the tracks' lines cost 4 ms, these cost nothing)
    __shared__ int blk_rng[kBlock / 64][4];
    if (lx == 0) { blk_rng[ly][0] = ndx; blk_rng[ly][1] = ndy; blk_rng[ly][2] = mdx; blk_rng[ly][3] = mdy; }
    __syncthreads();
    if (threadIdx.x < 4) {
        int m = blk_rng[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < kBlock / 64; w++) m = max(m, blk_rng[w][threadIdx.x]);
        int *dst = &rng[kRngStride * track + threadIdx.x];
        if (m > -0x40000000 && m > __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dst, m);
    }
    if (blk == 0 && threadIdx.x == 4) rng[kRngStride * track + 4] = max_win;
}

// The gather.  Its first form (a wave per window row, one 64 x 4 tile at a time: up to four dependent rounds of
// target-index loads per tile, then the hypothesis, then the age -- six latencies per 256 targets at full
// occupancy) took 0.47 ms for 64 VGA tracks; with the latencies of a tile taken together, 0.35 ms:
//   * a tile is 64 x 8 targets, two per thread (rows ly and ly + 4);
//   * the window is walked FLAT (index -> row, column by a multiply-shift), all of a thread's target-index
//     loads are issued before the first vote;
//   * after the votes a thread knows, for both its targets, the first source (whose hypothesis starts the
//     fold) and the last one (whose age is the writer's): those four loads are issued together.  Targets with
//     two to four sources load the rest in the fold; more than four: the plain scan, as above.
// Same sources in the same order as k_sd_fold -> same bits.
constexpr int kG2Batch = 4;                                 // target-index loads in flight per thread

template <bool AGE, bool PROP, int kG2Rows>
__global__ __launch_bounds__(kBlock) void k_sd_gather2(int H, int W, const TrackWarp *__restrict__ tw,
                                                       const int *__restrict__ tgt, const double2 *__restrict__ warped,
                                                       const int *__restrict__ rng,
                                                       const uint64_t *__restrict__ age0, int64_t stride,
                                                       double default_depth, double default_variance, double bias,
                                                       uint64_t *__restrict__ age1, double *__restrict__ depth1,
                                                       double *__restrict__ var1, int nb, int n_tracks,
                                                       unsigned int *__restrict__ fallbacks) {
    constexpr int kG2TH = kGatherTH * kG2Rows;                  // tile height
    __shared__ int cnt[kBlock * kG2Rows];
    __shared__ __attribute__((aligned(16))) int slot[kBlock * kG2Rows * kSlots];
    int track, blk;
    if (!xcd_major_track(nb, n_tracks, track, blk)) return;
    int dxmin, dymin, rx, ry;
    if (!gather_applies(rng, track, dxmin, dymin, rx, ry)) {
        if (blk == 0 && threadIdx.x == 0 && fallbacks) atomicAdd(fallbacks, 1u);
        return;
    }
    const int dxmax = dxmin + rx, dymax = dymin + ry;
    const TrackWarp &t = tw[track];
    const int64_t base = (int64_t)track * stride;
    const int *__restrict__ tg_all = tgt + base;
    const int tiles_x = (W + kGatherTW - 1) / kGatherTW, tiles_y = (H + kG2TH - 1) / kG2TH;
    const int lx = threadIdx.x & 63;
    const int ly = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ww = kGatherTW + max(rx, 0), wh = kG2TH + max(ry, 0);
    const int n_win = rx >= 0 ? ww * wh : 0;                        // <= 128 x 20
    const unsigned m_ww = ((1u << 24) + ww - 1) / ww;               // idx / ww = idx m_ww >> 24 (idx < 4096, ww <= 128)
    const unsigned m_W = (unsigned)std::min<uint64_t>(0x100000000ull / (unsigned)W, 0xffffffffull);  // tg / W: low by at most one
#pragma unroll
    for (int e = 0; e < kG2Rows; e++) cnt[threadIdx.x + e * kBlock] = 0;
    for (int tile = blk; tile < tiles_x * tiles_y; tile += nb) {
        const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
        const int tx0 = txi * kGatherTW, ty0 = tyi * kG2TH;
        const int wx0 = tx0 - dxmax, wy0 = ty0 - dymax;      // window origin in source coordinates
        __syncthreads();                                      // the previous tile's folds are through, counts zero
        for (int i0 = 0; i0 < n_win; i0 += kG2Batch * kBlock) {
            int src[kG2Batch], tg[kG2Batch];
#pragma unroll
            for (int j = 0; j < kG2Batch; j++) {
                const int idx = i0 + j * kBlock + (int)threadIdx.x;
                const int r = (int)(((unsigned)idx * m_ww) >> 24), c = idx - r * ww;
                const int sy = wy0 + r, sx = wx0 + c;
                const bool ok = idx < n_win && (unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W;
                src[j] = ok ? sy * W + sx : -1;
                tg[j] = -1;
                if (i0 + j * kBlock < n_win) tg[j] = tg_all[max(src[j], 0)];      // (uniform test: a whole round or none)
            }
#pragma unroll
            for (int j = 0; j < kG2Batch; j++) {
                if (src[j] < 0 || tg[j] < 0) continue;
                unsigned ty = __umulhi((unsigned)tg[j], m_W);
                unsigned tx = (unsigned)tg[j] - ty * (unsigned)W;
                if (tx >= (unsigned)W) { ty++; tx -= (unsigned)W; }
                const unsigned ux = tx - (unsigned)tx0, uy = ty - (unsigned)ty0;
                if (ux >= (unsigned)kGatherTW || uy >= (unsigned)kG2TH) continue;
                const int tl = (int)uy * kGatherTW + (int)ux;
                const int k = atomicAdd(&cnt[tl], 1);
                if (k < kSlots) slot[tl * kSlots + k] = src[j];
            }
        }
        __syncthreads();
        const int x = tx0 + lx;
        int kk[kG2Rows], me[kG2Rows], s[kG2Rows][kSlots], last[kG2Rows];
        double2 w0[kG2Rows], w1[kG2Rows];
        uint64_t a0[kG2Rows];
#pragma unroll
        for (int e = 0; e < kG2Rows; e++) {
            const int tl = (ly + e * kGatherTH) * kGatherTW + lx, y = ty0 + ly + e * kGatherTH;
            me[e] = (x < W && y < H) ? y * W + x : -1;
            kk[e] = cnt[tl];
            cnt[tl] = 0;                                      // for the next tile (this thread is the only reader)
            const int4 q = *reinterpret_cast<const int4 *>(slot + tl * kSlots);
            const int k = kk[e];
            s[e][0] = q.x; s[e][1] = k > 1 ? q.y : 0x7fffffff; s[e][2] = k > 2 ? q.z : 0x7fffffff;
            s[e][3] = k > 3 ? q.w : 0x7fffffff;
            if (k > 1 && k <= kSlots) {   // the votes arrive in any order: raster order by a five-exchange network
                sort2(s[e][0], s[e][1]); sort2(s[e][2], s[e][3]); sort2(s[e][0], s[e][2]); sort2(s[e][1], s[e][3]);
                sort2(s[e][1], s[e][2]);
            }
            const bool fast = me[e] >= 0 && k > 0 && k <= kSlots;
            last[e] = fast ? (k > 3 ? s[e][3] : k > 2 ? s[e][2] : k > 1 ? s[e][1] : s[e][0]) : -1;
            const int f = fast ? s[e][0] : max(me[e], 0);
            if (PROP) {
                w0[e] = warped[base + f];
                // the second source's hypothesis with the first's, unconditionally (a target without one re-reads the
                // first): most waves hold some target with two sources, and loading it inside the fold was a fourth
                // dependent memory phase per tile
                w1[e] = warped[base + ((fast && k > 1) ? s[e][1] : f)];
            }
            if (AGE) a0[e] = age0[base + (fast ? last[e] : max(me[e], 0))];
        }
#pragma unroll
        for (int e = 0; e < kG2Rows; e++) {
            if (me[e] < 0) continue;
            const int k = kk[e];
            double d = default_depth, v = default_variance;
            int lastw = -1;
            bool have = false;
            auto fold = [&](const double2 w) {
                if (!have) { d = w.x; v = w.y; have = true; }
                else {
                    double nd, nv;
                    handle_collision(w.x, d, w.y, v, nd, nv);
                    d = nd; v = nv;
                }
            };
            uint64_t a = 0;
            if (k > 0 && k <= kSlots) {
                if (PROP) {
                    fold(w0[e]);
                    if (k > 1) fold(w1[e]);
                    if (k > 2) fold(warped[base + s[e][2]]);
                    if (k > 3) fold(warped[base + s[e][3]]);
                }
                lastw = last[e];
                if (AGE) a = a0[e];
            } else if (k > kSlots) {
                // more than four sources on this target (a zoom-out inside the window): the plain scan of its
                // candidates s = t - d, d in the box, in raster order
                const int y = ty0 + ly + e * kGatherTH;
                for (int cj = 0; cj <= ry; cj++) {
                    const int sy = y - dymax + cj;
                    if ((unsigned)sy >= (unsigned)H) continue;
                    for (int ci = 0; ci <= rx; ci++) {
                        const int sx = x - dxmax + ci;
                        if ((unsigned)sx >= (unsigned)W) continue;
                        if (tg_all[sy * W + sx] == me[e]) {
                            if (PROP) fold(warped[base + sy * W + sx]);
                            lastw = sy * W + sx;
                        }
                    }
                }
                if (AGE && lastw >= 0) a = age0[base + lastw];
            }
            if (AGE) {
                if (lastw >= 0) {
                    a = a + 1;
                    a = a > t.age_cap ? t.age_cap : a;
                } else a = 0;
                age1[base + me[e]] = a;
            }
            if (PROP) { depth1[base + me[e]] = d; var1[base + me[e]] = v; }
        }
    }
}

// the slot path only where the gather did not apply (see above): the three launches of launch_warp_step
// that follow k_sd_gather2 test this first
__device__ __forceinline__ bool slot_path_wanted(const int *__restrict__ rng, int track) {
    if (rng == nullptr) return true;
    int a, b, c, d;
    return !gather_applies(rng, track, a, b, c, d);
}

// Does ANY track of the step need the slot path?  Asked once per block of the three slot-path launches, which
// run on a capped grid (kSlotGrid blocks, each walking its share of the virtual grid): when the gather took
// every track -- the normal case -- they cost a few microseconds each instead of the 26 us that 77 000
// blocks needed just to start and return (measured, 64 VGA tracks).
constexpr unsigned kSlotGrid = 2048;
__device__ __forceinline__ bool any_slot_track(const int *__restrict__ rng, int n_tracks) {
    if (rng == nullptr) return true;
    int want = 0;
    for (int t = threadIdx.x; t < n_tracks; t += kBlock) want |= slot_path_wanted(rng, t) ? 1 : 0;
    return __syncthreads_or(want) != 0;
}

__global__ __launch_bounds__(kBlock) void k_sd_zero_counts(int *__restrict__ cnt, int64_t stride, int N,
                                                           const int *__restrict__ rng, int nb, int n_tracks,
                                                           unsigned vgrid) {
    if (!any_slot_track(rng, n_tracks)) return;
    for (unsigned vb = blockIdx.x; vb < vgrid; vb += gridDim.x) {
    int track, blk;
    if (!xcd_major_track(nb, n_tracks, track, blk, vb)) continue;
    if (!slot_path_wanted(rng, track)) continue;
    int *c = cnt + (int64_t)track * stride;
    for (int i = blk * kBlock + threadIdx.x; i < N; i += nb * kBlock) c[i] = 0;
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
// What the search needs of the geometry computed before the last early exit.
struct SearchState {
    double xk, yk;              // normalised key coordinate
    double key_step, key_gradient;
    double xmin_x, xmin_y;      // near end of the epipolar segment in the reference frame
    double dirx, diry;          // its direction
    double rdx, rdy;            // far end - near end
    double key_I[5];            // the five key intensities
    int n;                      // search positions
};

// First half of estimate (:91-158): everything up to the last early exit.  Returns the Flag, 0 = search on.
__device__ int estimate_gate(double ukx, double uky, double prior_id, double prior_var, const Cam &kc,
                             const double *__restrict__ key_image, const RefConst &rf, int H, int W,
                             const EstParams &pr, SearchState &st) {
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
    st.xk = xk; st.yk = yk; st.key_step = key_step; st.key_gradient = key_gradient;
    st.xmin_x = xmin_x; st.xmin_y = xmin_y; st.dirx = dirx; st.diry = diry; st.rdx = rdx; st.rdy = rdy;
#pragma unroll
    for (int i = 0; i < 5; i++) st.key_I[i] = key_I[i];
    st.n = n;
    return 0;
}

// n / d for several n and one d, bit for bit what the compiler's FP64 division gives.  Its sequence is
//   d' = v_div_scale(d), n' = v_div_scale(n); r = v_rcp(d'); two Newton steps on r; q = n' r; e = fma(-d', q, n');
//   v_div_fmas(e, r, q); v_div_fixup
// and v_div_scale / v_div_fmas / v_div_fixup leave their operands alone unless an exponent is extreme (a numerator
// below 2^-969, a denormal divisor or quotient, exponents ~768 apart) or an operand is 0, Inf or NaN.  Here the
// numerators are the five samples of a search window and the divisor the root of the sum of their squares: if
// every sample is "plain" -- zero, or of magnitude in [2^-401, 2^400) -- the divisor (non-zero: checked by the
// caller) lies in [2^-401, 2^402), every quotient in [2^-803, 1], no scaling or fix-up applies and the result IS
// fma(e, r, q) with an r that depends on the divisor alone: refined once, three operations per quotient instead of
// eleven (-8 % of the instructions of k_ud_estimate, which is FP64-issue bound: profiles/r04_update_depth.txt).
// A zero numerator gives +-0 either way (the sign of a zero cannot reach the error: (+-0 - k)^2).  Inf / NaN
// samples count as plain too (v_frexp_exp returns 0 for them): they make the window's error NaN on both paths.
// Anything else -- denormals, |w| outside the range -- takes the division.  tests/test_gpu_round4.py feeds such
// frames through both and compares with the oracle's IEEE divisions.
__device__ __forceinline__ bool plain_numerator(double w) {
    return (unsigned)(__builtin_amdgcn_frexp_exp(w) + 400) <= 800u;
}

__device__ __forceinline__ double refined_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(r, e, r);
}

__device__ __forceinline__ double shared_quotient(double n, double d, double r) {
    const double q = n * r;
    return __builtin_fma(__builtin_fma(-d, q, n), r, q);
}

// Second half: the search along the epipolar line, depth and variance of the best match, final range check.
__device__ int estimate_search(const SearchState &st, double ukx, double uky, const Cam &kc,
                               const double *__restrict__ key_image, const RefConst &rf, int H, int W,
                               const EstParams &pr, double &out_id, double &out_var) {
    const double *T = rf.T_rk;
    const Cam rc{rf.cam[0], rf.cam[1], rf.cam[2], rf.cam[3]};
    const double xk = st.xk, yk = st.yk, key_step = st.key_step, key_gradient = st.key_gradient;
    const double xmin_x = st.xmin_x, xmin_y = st.xmin_y, dirx = st.dirx, diry = st.diry, rdx = st.rdx, rdy = st.rdy;
    const int n = st.n;
    double key_I[5], dtmp;
#pragma unroll
    for (int i = 0; i < 5; i++) key_I[i] = st.key_I[i];

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
    int plain_run = 5;          // samples up to and including w4 that are plain numerators (the initial zeros are)
    for (int i = 0; i < n; i++) {
        double s = (double)i * pr.ref_step, ux, uy;
        tdk::unnormalize(rc, xmin_x + s * dirx, xmin_y + s * diry, ux, uy);
        w0 = w1; w1 = w2; w2 = w3; w3 = w4;
        w4 = sample(rf.image, H, W, ux, uy);
        plain_run = plain_numerator(w4) ? plain_run + 1 : 0;
        if (i < 4) continue;
        double q = ((((w0 * w0 + w1 * w1) + w2 * w2) + w3 * w3) + w4 * w4);
        double sn = sqrt(q);
        double a0 = w0, a1 = w1, a2 = w2, a3 = w3, a4 = w4;
        if (sn != 0.) {
            if (plain_run >= 5) {       // the five IEEE quotients from one refinement of 1 / sn (shared_quotient)
                const double r = refined_rcp(sn);
                a0 = shared_quotient(w0, sn, r); a1 = shared_quotient(w1, sn, r); a2 = shared_quotient(w2, sn, r);
                a3 = shared_quotient(w3, sn, r); a4 = shared_quotient(w4, sn, r);
            } else { a0 = w0 / sn; a1 = w1 / sn; a2 = w2 / sn; a3 = w3 / sn; a4 = w4 / sn; }
        }
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
    // the key coordinate is an integer pixel: the bilinear sample of the Sobel map there is the
    // map value itself (interpolation.rs:9-43 short-circuits integer coordinates), computed in place
    double igx, igy;
    sobel_at(key_image, H, W, (int)ukx, (int)uky, igx, igy);
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

// estimate (:91-158) in one piece (estimate_debug_)
__device__ int estimate(double ukx, double uky, double prior_id, double prior_var, const Cam &kc,
                        const double *__restrict__ key_image, const RefConst &rf, int H, int W,
                        const EstParams &pr, double &out_id, double &out_var) {
    SearchState st;
    const int f = estimate_gate(ukx, uky, prior_id, prior_var, kc, key_image, rf, H, W, pr, st);
    if (f) return f;
    return estimate_search(st, ukx, uky, kc, key_image, rf, H, W, pr, out_id, out_var);
}

// ---- update_depth (:160-234), split into classify + estimate ---------------------
struct TrackKey {   // per track and step: the key frame of update_depth
    double cam[4];
    const double *image;
    int n_ref, pad;
};

enum { SD_ERR_AGE = 1 };

// Streams age / prior maps.  NotProcessed (age == 0, :196-200) and check_args
// failures (:208-214) are final here; everything else is appended to the track's
// list of live pixels.  kClassifySweeps pixels per thread; the block reserves its
// slice of the list with ONE atomic on the track's counter -- counters of different
// tracks sit kCountStride ints apart (own cache lines: atomics on one line
// serialise in its L2 channel).
constexpr int kClassifySweeps = 4;
constexpr int kCountStride = 64;
__global__ __launch_bounds__(kBlock) void k_ud_classify(int N, const TrackKey *__restrict__ keys,
                                                        const uint64_t *__restrict__ age,
                                                        const double *__restrict__ prior_depth,
                                                        const double *__restrict__ prior_var, int64_t stride,
                                                        double vmin, double vmax, double *__restrict__ out_depth,
                                                        double *__restrict__ out_var,
                                                        int64_t *__restrict__ out_flag, int *__restrict__ list,
                                                        int *__restrict__ count, int *__restrict__ err) {
    const int track = blockIdx.y;
    const int64_t base = (int64_t)track * stride;
    const int n_ref = keys[track].n_ref;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ int wave_total[kClassifySweeps][kBlock / 64];
    __shared__ int block_base;
    bool live[kClassifySweeps];
    int before[kClassifySweeps];
#pragma unroll
    for (int s = 0; s < kClassifySweeps; s++) {
        const int i = (blockIdx.x * kClassifySweeps + s) * kBlock + (int)threadIdx.x;
        live[s] = false;
        if (i < N) {
            const uint64_t a = age[base + i];
            const double d = prior_depth[base + i], v = prior_var[base + i];
            int f = -9;  // NotProcessed
            if (a != 0) {
                if (a > (uint64_t)n_ref) atomicOr(err, SD_ERR_AGE);  // the reference exits here (:202-205)
                else f = check_args(tdk::safe_inv(d), v, vmin, vmax);
            }
            live[s] = f == 0;
            // Every pixel is written here, the live ones with what `estimate` returns when it keeps the prior
            // (Err(flag) => (prior, flag): 1 / inv_depth of the prior, its variance) -- full cache lines instead of
            // the 70 % of a line that is final here (257 -> 152 us for 64 VGA tracks); k_ud_estimate then writes a
            // live pixel's flag, and depth / variance only where the search succeeded.
            out_depth[base + i] = live[s] ? tdk::safe_inv(tdk::safe_inv(d)) : d;
            out_var[base + i] = v;
            out_flag[base + i] = f;
        }
        const uint64_t m = __builtin_amdgcn_ballot_w64(live[s]);
        before[s] = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
        if (lane == 0) wave_total[s][wave] = __builtin_popcountll(m);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
#pragma unroll
        for (int s = 0; s < kClassifySweeps; s++)
#pragma unroll
            for (int w = 0; w < kBlock / 64; w++) { int c = wave_total[s][w]; wave_total[s][w] = tot; tot += c; }
        block_base = tot ? atomicAdd(&count[track * kCountStride], tot) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kClassifySweeps; s++) {
        const int i = (blockIdx.x * kClassifySweeps + s) * kBlock + (int)threadIdx.x;
        if (live[s]) list[base + block_base + wave_total[s][wave] + before[s]] = i;
    }
}

// One lane per live pixel, in two phases.  A third of the live pixels leaves `estimate` at one
// of its early exits (texture-less key window, epipolar segment out of range, ...), after about a
// quarter of the arithmetic; the search that follows is FP64-issue bound, and lanes that left
// would idle through it.  So the block runs the gate for its 256 pixels, compacts the survivors'
// state through LDS (15 doubles + 3 ints per lane, structure of arrays) and only as many waves as
// there are survivors run the search -- same arithmetic per pixel, full waves.
constexpr int kStateDoubles = 15;
__global__ __launch_bounds__(kBlock) void k_ud_estimate(int H, int W, const TrackKey *__restrict__ keys,
                                                        const RefConst *__restrict__ refs, int refs_per_track,
                                                        const uint64_t *__restrict__ age,
                                                        const double *__restrict__ prior_depth,
                                                        const double *__restrict__ prior_var, int64_t stride,
                                                        EstParams pr, const int *__restrict__ list,
                                                        const int *__restrict__ count,
                                                        double *__restrict__ out_depth,
                                                        double *__restrict__ out_var,
                                                        int64_t *__restrict__ out_flag, int nb, int n_tracks) {
    // XCD-major (xcd_major_track): all blocks of a track on ONE XCD, one after the other.  A live pixel gathers ~60
    // texels of the track's key and reference frames; as a (blocks, tracks) grid every XCD worked on an eighth of
    // every track and each of the eight L2s fetched both frames of all 64 tracks: 133 B/px of HBM-side traffic for
    // 57.6 algorithmic (rocprofv3 FETCH_SIZE / WRITE_SIZE, profiles/r06_update_depth.txt).
    int track, bx;
    if (!xcd_major_track(nb, n_tracks, track, bx)) return;
    const int n_live = count[track * kCountStride];
    if (bx * kBlock >= n_live) return;                           // block-uniform
    const int k = bx * kBlock + (int)threadIdx.x;
    const int64_t base = (int64_t)track * stride;
    const TrackKey &key = keys[track];
    const Cam kc{key.cam[0], key.cam[1], key.cam[2], key.cam[3]};
    __shared__ double sd[kStateDoubles][kBlock];
    __shared__ int si[3][kBlock];
    __shared__ int wave_total[kBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    // ---- phase 1: the gate ----
    bool go = false;
    SearchState st;
    int i = 0, a = 0;
    if (k < n_live) {
        i = list[base + k];
        a = (int)age[base + i];
        const double d = prior_depth[base + i], v = prior_var[base + i];
        const double pid = tdk::safe_inv(d);
        const int y = i / W, x = i - y * W;
        // refframes[len - age] (:207): refs[track][age - 1] is the frame `age` steps before the key frame
        const RefConst &rf = refs[(int64_t)track * refs_per_track + (a - 1)];
        const int f = estimate_gate((double)x, (double)y, pid, v, kc, key.image, rf, H, W, pr, st);
        go = f == 0;
        if (!go) out_flag[base + i] = f;                         // Err(flag) => (prior, flag): the prior is in place (k_ud_classify)
    }
    const uint64_t m = __builtin_amdgcn_ballot_w64(go);
    const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
    if (lane == 0) wave_total[wave] = __builtin_popcountll(m);
    __syncthreads();
    int slot = before, survivors = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; w++) {
        if (w < wave) slot += wave_total[w];
        survivors += wave_total[w];
    }
    if (go) {
        sd[0][slot] = st.xk; sd[1][slot] = st.yk; sd[2][slot] = st.key_step; sd[3][slot] = st.key_gradient;
        sd[4][slot] = st.xmin_x; sd[5][slot] = st.xmin_y; sd[6][slot] = st.dirx; sd[7][slot] = st.diry;
        sd[8][slot] = st.rdx; sd[9][slot] = st.rdy;
#pragma unroll
        for (int q = 0; q < 5; q++) sd[10 + q][slot] = st.key_I[q];
        si[0][slot] = i; si[1][slot] = a; si[2][slot] = st.n;
    }
    __syncthreads();

    // ---- phase 2: the search, survivors packed into the first waves ----
    if ((int)threadIdx.x >= survivors) return;
    const int t = threadIdx.x;
    st.xk = sd[0][t]; st.yk = sd[1][t]; st.key_step = sd[2][t]; st.key_gradient = sd[3][t];
    st.xmin_x = sd[4][t]; st.xmin_y = sd[5][t]; st.dirx = sd[6][t]; st.diry = sd[7][t];
    st.rdx = sd[8][t]; st.rdy = sd[9][t];
#pragma unroll
    for (int q = 0; q < 5; q++) st.key_I[q] = sd[10 + q][t];
    i = si[0][t]; a = si[1][t]; st.n = si[2][t];
    const int y = i / W, x = i - y * W;
    const RefConst &rf = refs[(int64_t)track * refs_per_track + (a - 1)];
    double id = 0.0, var = 0.0;
    const int f = estimate_search(st, (double)x, (double)y, kc, key.image, rf, H, W, pr, id, var);
    if (f) {                                                     // the final range check failed: the prior stays
        out_flag[base + i] = f;
        return;
    }
    out_depth[base + i] = tdk::safe_inv(id);
    out_var[base + i] = var;                                     // (the flag is already 0)
}

__global__ void k_estimate_one(Cam kc, const double *key_image, const RefConst *refs, double ukx, double uky,
                               double pid, double pvar, int H, int W, EstParams pr,
                               double *out /*[id, var, flag]*/) {
    double id = 0, var = 0;
    int f = estimate(ukx, uky, pid, pvar, kc, key_image, refs[0], H, W, pr, id, var);
    out[0] = id;
    out[1] = var;
    out[2] = (double)f;
}

// flag histogram per track: bins 0, -1, ..., -9
__global__ __launch_bounds__(kBlock) void k_flag_hist(int N, const int64_t *__restrict__ flag, int64_t stride,
                                                      unsigned long long *__restrict__ hist) {
    const int track = blockIdx.y;
    __shared__ unsigned int h[10];
    if (threadIdx.x < 10) h[threadIdx.x] = 0;
    __syncthreads();
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += gridDim.x * kBlock) {
        const int f = (int)flag[(int64_t)track * stride + i];
#pragma unroll
        for (int b = 0; b < 10; b++) {
            const uint64_t m = __builtin_amdgcn_ballot_w64(f == -b);
            if (m && (threadIdx.x & 63) == 0) atomicAdd(&h[b], (unsigned int)__builtin_popcountll(m));
        }
    }
    __syncthreads();
    if (threadIdx.x < 10 && h[threadIdx.x]) atomicAdd(&hist[track * 10 + threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

// ---- post-steps ---------------------------------------------------------
// regularize (src/semi_dense/regularization.rs:5-64): zero padding contributes
// nothing (flag NotProcessed), so out-of-image neighbours are simply skipped; the
// accumulation order is the raster order of the 3x3 patch.
__global__ __launch_bounds__(kBlock) void k_regularize(const double *__restrict__ depth,
                                                       const double *__restrict__ variance,
                                                       const int64_t *__restrict__ flag, int H, int W,
                                                       double *__restrict__ out) {
    const int N = H * W;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += gridDim.x * kBlock) {
        const int y = i / W, x = i - y * W;
        double numerator = 0.0, denominator = 0.0;
#pragma unroll
        for (int dy = -1; dy <= 1; dy++)
#pragma unroll
            for (int dx = -1; dx <= 1; dx++) {
                const int yy = y + dy, xx = x + dx;
                if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                const int j = yy * W + xx;
                if (flag[j] != 0) continue;   // Flag::Success
                const double id = tdk::safe_inv(depth[j]), iv = tdk::safe_inv(variance[j]);
                numerator = numerator + id * iv;
                denominator = denominator + iv;
            }
        out[i] = denominator == 0.0 ? depth[i] : tdk::safe_inv(numerator / denominator);
    }
}

// fusion_arrays (src/semi_dense/fusion.rs:3-42)
__global__ __launch_bounds__(kBlock) void k_fusion(const double *__restrict__ mu1, const double *__restrict__ mu2,
                                                   const double *__restrict__ var1,
                                                   const double *__restrict__ var2, int64_t n,
                                                   double *__restrict__ mu, double *__restrict__ var) {
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const double m1 = mu1[i], m2 = mu2[i], v1 = var1[i], v2 = var2[i];
        const double v = v1 + v2;
        mu[i] = (m1 * v2 + m2 * v1) / v;
        var[i] = (v1 * v2) / v;
    }
}

// tdk_sd_export_dvo: I0 = previous frame, D0 = depth map, I1 = newest frame, W0 = safe_invert(variance)
// (examples/semi_dense_vo.py:44-53; the Python safe_invert with epsilon 1e-16, not the Rust one)
__global__ __launch_bounds__(kBlock) void k_export_dvo(int N, const double *const *__restrict__ prev_image,
                                                       const double *const *__restrict__ new_image,
                                                       const double *__restrict__ depth,
                                                       const double *__restrict__ variance, int64_t stride,
                                                       double *__restrict__ I0, double *__restrict__ D0,
                                                       double *__restrict__ I1, double *__restrict__ W0,
                                                       int64_t dvo_stride) {
    const int track = blockIdx.y;
    const double *__restrict__ p = prev_image[track], *__restrict__ q = new_image[track];
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += gridDim.x * kBlock) {
        const int64_t o = (int64_t)track * dvo_stride + i, m = (int64_t)track * stride + i;
        I0[o] = p[i];
        I1[o] = q[i];
        D0[o] = depth[m];
        if (W0) W0[o] = 1.0 / (variance[m] + tdk::kEps16);   // tadataka.numeric.safe_invert (numeric.py:1-2), as the example calls it
    }
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

tdk_status h2d(int slot, const void *host, size_t bytes, void **dev) {
    TDK_TRY(tdk::scratch(slot, bytes, dev));
    if (bytes) TDK_HIP(hipMemcpyAsync(*dev, host, bytes, hipMemcpyHostToDevice, tdk::stream()));
    return TDK_OK;
}

tdk_status check_image_dims(int H, int W) {
    TDK_REQUIRE(H >= 1 && W >= 1 && (int64_t)H * W < (1ll << 30), "bad image size");
    return TDK_OK;
}

void fill_track_warp(TrackWarp *tw, const double *T10, const double *cam0, const double *cam1,
                     uint64_t age_cap = ~0ull) {
    tw->age_cap = age_cap;
    memcpy(tw->T10, T10, sizeof(double) * 16);
    memcpy(tw->cam0, cam0, sizeof(double) * 4);
    memcpy(tw->cam1, cam1, sizeof(double) * 4);
}

// forward warp (increment_age and / or propagate) for n_tracks maps laid out [track][stride]: the gather path
// where a track's displacement box allows it, scatter + fold where not (both queued; the device decides)
template <bool AGE, bool PROP>
tdk_status launch_warp_step(int n_tracks, int H, int W, const TrackWarp *d_tw, const uint64_t *age0,
                            const double *depth0, const double *var0, int64_t stride, double default_depth,
                            double default_variance, double bias, int *list_buf, uint64_t *age1,
                            double *depth1, double *var1, hipStream_t stream, unsigned int *d_fallbacks = nullptr) {
    // list_buf: kWarpListInts * stride * n_tracks ints (carve_lists); stride must be even (16-byte slot rows)
    const int N = H * W;
    const WarpLists lists = carve_lists(list_buf, stride, n_tracks);
    const int nb = grid_for(N);
    const unsigned grid = 8u * (unsigned)((n_tracks + 7) / 8) * (unsigned)nb;
    const int use_gather = tdk::option(TDK_OPT_SD_WARP_GATHER);   // 0: the slot path for every track (tests)
    // the target indices alias `next`, which the slot path only writes AFTER k_sd_gather2 has run (and only for
    // the tracks that fell back); the boxes have their own 4 n_tracks ints behind the lists
    int *rng = nullptr;
    if (use_gather) {
        rng = lists.rng;
        TDK_HIP(hipMemsetAsync(rng, 0x80, sizeof(int) * kRngStride * (size_t)n_tracks, stream));
        // (depth1, variance1a) of every source: 16 bytes per pixel in the `slots` region, like `next` free until
        // the slot path runs
        double2 *warped = reinterpret_cast<double2 *>(lists.slots);
        // pass 1: four tiles of 64 (or 128: two pixels per lane where W is even) x 4 sources per block and round
        // (measured on 64 VGA tracks: 1 / 2 / 4 / 8 / 19 / 38 tiles per block 0.761 / 0.706 / 0.684 / 0.686 / 0.733 /
        // 0.764 ms for the whole step)
        const int n_tiles = ((W + kGatherTW - 1) / kGatherTW) * ((H + kGatherTH - 1) / kGatherTH);
        const int gnb = std::max(1, (n_tiles + 3) / 4);
        const bool px2 = W % 2 == 0;
        const int n_tiles2 = ((W + 2 * kGatherTW - 1) / (2 * kGatherTW)) * ((H + kGatherTH - 1) / kGatherTH);
        const int tnb = px2 ? std::max(1, (n_tiles2 + 3) / 4) : gnb;
        const unsigned tgrid = 8u * (unsigned)((n_tracks + 7) / 8) * (unsigned)tnb;
        if (px2)
            k_sd_targets<PROP, 2, true><<<tgrid, kBlock, 0, stream>>>(H, W, d_tw, depth0, var0, bias, stride, lists.next,
                                                                       warped, rng, tnb, n_tracks, 0);
        else
            k_sd_targets<PROP, 1, true><<<tgrid, kBlock, 0, stream>>>(H, W, d_tw, depth0, var0, bias, stride, lists.next,
                                                                       warped, rng, tnb, n_tracks, 0);
        TDK_LAUNCH_CHECK();
        {
            const int n_tiles8 = ((W + kGatherTW - 1) / kGatherTW) * ((H + 2 * kGatherTH - 1) / (2 * kGatherTH));
            int g2nb = std::max(1, (n_tiles8 + 1) / 2);
            const unsigned g2grid = 8u * (unsigned)((n_tracks + 7) / 8) * (unsigned)g2nb;
            k_sd_gather2<AGE, PROP, 2><<<g2grid, kBlock, 0, stream>>>(H, W, d_tw, lists.next, warped, rng, age0, stride,
                                                                       default_depth, default_variance, bias, age1,
                                                                       depth1, var1, g2nb, n_tracks, d_fallbacks);
        }
        TDK_LAUNCH_CHECK();
    }
    // with the gather queued the slot path is the exception: a capped grid whose blocks ask first whether any
    // track needs them (any_slot_track); without it, the full grid
    const unsigned sgrid = rng ? std::min(grid, kSlotGrid) : grid;
    k_sd_zero_counts<<<sgrid, kBlock, 0, stream>>>(lists.cnt, stride, N, rng, nb, n_tracks, grid);
    TDK_LAUNCH_CHECK();
    k_sd_scatter<<<sgrid, kBlock, 0, stream>>>(H, W, d_tw, depth0, stride, lists, nb, n_tracks, rng, grid);
    TDK_LAUNCH_CHECK();
    k_sd_fold<AGE, PROP><<<sgrid, kBlock, 0, stream>>>(H, W, d_tw, lists, age0, depth0, var0, stride,
                                                       default_depth, default_variance, bias, age1, depth1, var1,
                                                       nb, n_tracks, rng, grid);
    TDK_LAUNCH_CHECK();
    return TDK_OK;
}

tdk_status launch_update_depth(int n_tracks, int H, int W, const TrackKey *d_keys, const RefConst *d_refs,
                               int refs_per_track, const uint64_t *age, const double *prior_depth,
                               const double *prior_var, int64_t stride, const EstParams &pr, int *list,
                               int *count, int *err, double *out_depth, double *out_var, int64_t *out_flag,
                               hipStream_t stream) {
    const int N = H * W;
    TDK_HIP(hipMemsetAsync(count, 0, sizeof(int) * kCountStride * n_tracks, stream));
    dim3 cgrid((N + kBlock * kClassifySweeps - 1) / (kBlock * kClassifySweeps), n_tracks);
    k_ud_classify<<<cgrid, kBlock, 0, stream>>>(N, d_keys, age, prior_depth, prior_var, stride, pr.vmin, pr.vmax,
                                                out_depth, out_var, out_flag, list, count, err);
    TDK_LAUNCH_CHECK();
    const int nb = (N + kBlock - 1) / kBlock;
    const int64_t eblocks = (int64_t)8 * ((n_tracks + 7) / 8) * nb;
    if (eblocks >= (1ll << 31)) { tdk::set_error("update_depth: grid too large"); return TDK_ERR_INVALID_ARGUMENT; }
    k_ud_estimate<<<(unsigned)eblocks, kBlock, 0, stream>>>(H, W, d_keys, d_refs, refs_per_track, age, prior_depth,
                                                            prior_var, stride, pr, list, count, out_depth, out_var,
                                                            out_flag, nb, n_tracks);
    TDK_LAUNCH_CHECK();
    return TDK_OK;
}

}  // namespace

extern "C" {

tdk_status tdk_sobel(const double *image, int H, int W, double *gx, double *gy) {
    TDK_API_GUARD;
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
    TDK_API_GUARD;
    TDK_REQUIRE(age0 && camera0 && camera1 && T10 && depth0 && age1, "null pointer");
    TDK_TRY(check_image_dims(H, W));
    const int N = H * W;
    void *d_age0, *d_depth, *d_lists, *d_age1, *d_tw;
    TDK_TRY(h2d(0, age0, (size_t)N * 8, &d_age0));
    TDK_TRY(h2d(1, depth0, (size_t)N * 8, &d_depth));
    TDK_TRY(tdk::scratch(2, warp_list_bytes(N, 1), &d_lists));
    TDK_TRY(tdk::scratch(4, (size_t)N * 8, &d_age1));
    TrackWarp tw;
    fill_track_warp(&tw, T10, camera0, camera1);
    TDK_TRY(h2d(10, &tw, sizeof(tw), &d_tw));
    TDK_TRY((launch_warp_step<true, false>(1, H, W, (const TrackWarp *)d_tw, (const uint64_t *)d_age0,
                                           (const double *)d_depth, nullptr, N, 0., 0., 0., (int *)d_lists,
                                           (uint64_t *)d_age1, nullptr, nullptr, tdk::stream())));
    TDK_HIP(hipMemcpyAsync(age1, d_age1, (size_t)N * 8, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipStreamSynchronize(tdk::stream()));  // tw must outlive the H2D copy
    return TDK_OK;
}

tdk_status tdk_propagate(const double *T10, const double *camera0, const double *camera1, const double *depth0,
                         const double *variance0, int H, int W, double default_depth, double default_variance,
                         double uncertaintity_bias, double *depth1, double *variance1) {
    TDK_API_GUARD;
    TDK_REQUIRE(T10 && camera0 && camera1 && depth0 && variance0 && depth1 && variance1, "null pointer");
    TDK_TRY(check_image_dims(H, W));
    const int N = H * W;
    size_t b8 = (size_t)N * 8;
    void *d_d0, *d_v0, *d_lists, *d_d1, *d_v1, *d_tw;
    TDK_TRY(h2d(0, depth0, b8, &d_d0));
    TDK_TRY(h2d(1, variance0, b8, &d_v0));
    TDK_TRY(tdk::scratch(2, warp_list_bytes(N, 1), &d_lists));
    TDK_TRY(tdk::scratch(4, b8, &d_d1));
    TDK_TRY(tdk::scratch(5, b8, &d_v1));
    TrackWarp tw;
    fill_track_warp(&tw, T10, camera0, camera1);
    TDK_TRY(h2d(10, &tw, sizeof(tw), &d_tw));
    TDK_TRY((launch_warp_step<false, true>(1, H, W, (const TrackWarp *)d_tw, nullptr, (const double *)d_d0,
                                           (const double *)d_v0, N, default_depth, default_variance,
                                           uncertaintity_bias, (int *)d_lists, nullptr,
                                           (double *)d_d1, (double *)d_v1, tdk::stream())));
    TDK_HIP(hipMemcpyAsync(depth1, d_d1, b8, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipMemcpyAsync(variance1, d_v1, b8, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipStreamSynchronize(tdk::stream()));
    return TDK_OK;
}

}  // extern "C"

struct tdk_frame {          // a device-resident image (rust_bindings.semi_dense.Frame keeps one per frame)
    double *image;
    int H, W;
};

namespace {

// update_depth with the images already on the device: d_key_image and d_ref_images[r] (host array
// of n_ref device pointers, in the caller's refframes order).  Maps in and out are host arrays.
tdk_status update_depth_resident(const double *key_camera, const double *d_key_image, const double *key_T, int n_ref,
                                 const double *ref_cameras, const double *const *d_ref_images, const double *ref_Ts,
                                 const uint64_t *age, const double *prior_depth, const double *prior_variance,
                                 int H, int W, const tdk_semi_dense_params *params, double *depth,
                                 double *variance, int64_t *flag) {
    const int N = H * W;
    size_t b8 = (size_t)N * 8;
    void *d_age, *d_pd, *d_pv, *d_od, *d_ov, *d_of, *d_rc, *d_keys, *d_list, *d_cnt;
    TDK_TRY(h2d(4, age, b8, &d_age));
    TDK_TRY(h2d(5, prior_depth, b8, &d_pd));
    TDK_TRY(h2d(6, prior_variance, b8, &d_pv));
    TDK_TRY(tdk::scratch(7, b8, &d_od));
    TDK_TRY(tdk::scratch(8, b8, &d_ov));
    TDK_TRY(tdk::scratch(9, b8, &d_of));
    TDK_TRY(tdk::scratch(1, (size_t)N * 4, &d_list));
    TDK_TRY(tdk::scratch(2, (kCountStride + 1) * sizeof(int), &d_cnt));
    // indexed by age - 1: refframes[n_ref - age] (:207)
    std::vector<RefConst> rcs((size_t)(n_ref > 0 ? n_ref : 1));
    for (int a = 1; a <= n_ref; a++) {
        const int r = n_ref - a;
        TDK_TRY(make_ref_const(key_T, ref_Ts + 16 * r, ref_cameras + 4 * r, d_ref_images[r], &rcs[a - 1]));
    }
    TDK_TRY(h2d(10, rcs.data(), sizeof(RefConst) * rcs.size(), &d_rc));
    TrackKey key;
    memcpy(key.cam, key_camera, sizeof(double) * 4);
    key.image = d_key_image;
    key.n_ref = n_ref;
    key.pad = 0;
    TDK_TRY(h2d(11, &key, sizeof(key), &d_keys));
    int *d_err = (int *)d_cnt + kCountStride;
    TDK_HIP(hipMemsetAsync(d_err, 0, sizeof(int), tdk::stream()));
    TDK_TRY(launch_update_depth(1, H, W, (const TrackKey *)d_keys, (const RefConst *)d_rc, n_ref > 0 ? n_ref : 1,
                                (const uint64_t *)d_age, (const double *)d_pd, (const double *)d_pv, N,
                                est_params(params), (int *)d_list, (int *)d_cnt, d_err, (double *)d_od,
                                (double *)d_ov, (int64_t *)d_of, tdk::stream()));
    int err = 0;
    TDK_HIP(hipMemcpyAsync(&err, d_err, sizeof(int), hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipStreamSynchronize(tdk::stream()));  // rcs / key must outlive the H2D copies
    if (err & SD_ERR_AGE) {
        // the reference exits the process if some age exceeds len(refframes) (:202-205)
        tdk::set_error("Age exceeds the refframe size");
        return TDK_ERR_AGE_EXCEEDS_REFFRAMES;
    }
    TDK_HIP(hipMemcpyAsync(depth, d_od, b8, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipMemcpyAsync(variance, d_ov, b8, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipMemcpyAsync(flag, d_of, b8, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipStreamSynchronize(tdk::stream()));
    return TDK_OK;
}

}  // namespace

extern "C" {

tdk_status tdk_update_depth(const double *key_camera, const double *key_image, const double *key_T, int n_ref,
                            const double *ref_cameras, const double *ref_images, const double *ref_Ts,
                            const uint64_t *age, const double *prior_depth, const double *prior_variance, int H,
                            int W, const tdk_semi_dense_params *params, double *depth, double *variance,
                            int64_t *flag) {
    TDK_API_GUARD;
    TDK_REQUIRE(key_camera && key_image && key_T && age && prior_depth && prior_variance && params && depth &&
                    variance && flag && n_ref >= 0,
                "bad argument");
    TDK_REQUIRE(n_ref == 0 || (ref_cameras && ref_images && ref_Ts), "null reference frames");
    TDK_TRY(check_image_dims(H, W));
    const size_t b8 = (size_t)H * W * 8;
    void *d_key, *d_refs_img;
    TDK_TRY(h2d(0, key_image, b8, &d_key));
    TDK_TRY(h2d(3, ref_images, b8 * (size_t)n_ref, &d_refs_img));
    std::vector<const double *> ptrs((size_t)n_ref);
    for (int r = 0; r < n_ref; r++) ptrs[(size_t)r] = (const double *)d_refs_img + (size_t)r * H * W;
    return update_depth_resident(key_camera, (const double *)d_key, key_T, n_ref, ref_cameras, ptrs.data(), ref_Ts,
                                 age, prior_depth, prior_variance, H, W, params, depth, variance, flag);
}

tdk_status tdk_frame_create(const double *image, int height, int width, tdk_frame **out) {
    TDK_API_GUARD;
    TDK_REQUIRE(image && out, "null pointer");
    TDK_TRY(check_image_dims(height, width));
    TDK_TRY(tdk::ensure_device());
    tdk_frame *f = new tdk_frame();
    f->H = height; f->W = width;
    const size_t bytes = (size_t)height * width * 8;
    if (hipMalloc(&f->image, bytes) != hipSuccess) {
        delete f;
        tdk::set_error("hipMalloc of a %d x %d frame failed", height, width);
        return TDK_ERR_HIP;
    }
    // on the upload stream: the wait below is for this copy only, the kernels of the previous frame keep running
    hipError_t e = hipMemcpyAsync(f->image, image, bytes, hipMemcpyHostToDevice, tdk::upload_stream());
    if (e == hipSuccess) e = hipStreamSynchronize(tdk::upload_stream());   // the caller's array may go away
    if (e != hipSuccess) {
        (void)hipFree(f->image);
        delete f;
        tdk::set_error("frame upload failed: %s", hipGetErrorString(e));
        return TDK_ERR_HIP;
    }
    *out = f;
    return TDK_OK;
}

tdk_status tdk_frame_destroy(tdk_frame *f) {
    TDK_API_GUARD;
    if (!f) return TDK_OK;
    (void)hipStreamSynchronize(tdk::stream());
    (void)hipFree(f->image);
    delete f;
    return TDK_OK;
}

tdk_status tdk_update_depth_frames(const double *key_camera, const tdk_frame *key_frame, const double *key_T,
                                   int n_ref, const double *ref_cameras, const tdk_frame *const *ref_frames,
                                   const double *ref_Ts, const uint64_t *age, const double *prior_depth,
                                   const double *prior_variance, const tdk_semi_dense_params *params,
                                   double *depth, double *variance, int64_t *flag) {
    TDK_API_GUARD;
    TDK_REQUIRE(key_camera && key_frame && key_T && age && prior_depth && prior_variance && params && depth &&
                    variance && flag && n_ref >= 0,
                "bad argument");
    TDK_REQUIRE(n_ref == 0 || (ref_cameras && ref_frames && ref_Ts), "null reference frames");
    const int H = key_frame->H, W = key_frame->W;
    std::vector<const double *> ptrs((size_t)n_ref);
    for (int r = 0; r < n_ref; r++) {
        TDK_REQUIRE(ref_frames[r] && ref_frames[r]->H == H && ref_frames[r]->W == W,
                    "reference frames must have the key frame's shape");
        ptrs[(size_t)r] = ref_frames[r]->image;
    }
    return update_depth_resident(key_camera, key_frame->image, key_T, n_ref, ref_cameras, ptrs.data(), ref_Ts, age,
                                 prior_depth, prior_variance, H, W, params, depth, variance, flag);
}

// ---------------------------------------------------------------------------------------------
// Device-resident maps (tdk_map): the loop of examples/semi_dense_vo.py:182-199 hands every map a
// call returns straight back into the next call.  With the host-pointer entries above each of those
// hand-overs is a download followed by an upload of the same bytes; with maps the three operators
// take and return handles, run asynchronously on the library stream, and a map crosses PCIe only
// when the caller looks at it (tdk_map_download).
// ---------------------------------------------------------------------------------------------
}  // extern "C"

struct tdk_map {            // H x W elements of 8 bytes (float64 / uint64 / int64) on the device
    void *data;
    int H, W;
    // Read as uint64 (an age map): no element exceeds this; kNoBound = unknown.  Kept on the host so that
    // tdk_update_depth_maps knows without a device round trip that no age exceeds the reference frames it was given.
    uint64_t bound;
};

namespace {

// Freed maps are kept for the next create of the same size: the operators are stream-ordered on the
// library stream, so a buffer can be handed out again while its last reader is still queued, and the
// loop above (three maps in, six out per frame) never reaches hipMalloc / hipFree after its first frame.
struct MapPool {
    std::vector<std::pair<size_t, void *>> free_list;
};
MapPool g_map_pool;
constexpr size_t kMapPoolMax = 48;
constexpr size_t kMapPoolMaxBytes = (size_t)1 << 30;   // ... and at most 1 GiB parked (48 maps of a 4K frame would be 3 GB)
size_t g_map_pool_bytes = 0;

tdk_status map_alloc(size_t bytes, void **out) {
    for (size_t i = g_map_pool.free_list.size(); i-- > 0;)
        if (g_map_pool.free_list[i].first == bytes) {
            *out = g_map_pool.free_list[i].second;
            g_map_pool.free_list.erase(g_map_pool.free_list.begin() + (long)i);
            g_map_pool_bytes -= bytes;
            return TDK_OK;
        }
    TDK_HIP(hipMalloc(out, bytes));
    return TDK_OK;
}

void map_release(size_t bytes, void *p) {
    if (g_map_pool.free_list.size() < kMapPoolMax && g_map_pool_bytes + bytes <= kMapPoolMaxBytes) {
        g_map_pool.free_list.emplace_back(bytes, p);
        g_map_pool_bytes += bytes;
        return;
    }
    (void)hipStreamSynchronize(tdk::stream());
    (void)hipFree(p);
}

// Small host structs (per-call constants) go to the device through a ring of pinned blocks, so the
// caller's stack memory may go away at once and no call has to wait for its own upload.
constexpr int kRingSlots = 16;
constexpr size_t kRingBytes = 64 << 10;
struct StageRing {
    char *base = nullptr;
    hipEvent_t ev[kRingSlots];
    bool pending[kRingSlots] = {};
    int next = 0;
};
StageRing g_ring;

// tdk_set_device leaves the device these belong to: parked maps are freed (a later map_alloc must not hand out a
// buffer of the previous device), the ring's pinned block and events go (they were recorded on that device's stream)
void release_map_state() {
    (void)hipStreamSynchronize(tdk::stream());
    for (auto &e : g_map_pool.free_list) (void)hipFree(e.second);
    g_map_pool.free_list.clear();
    g_map_pool_bytes = 0;
    if (g_ring.base) {
        for (int i = 0; i < kRingSlots; i++) {
            if (g_ring.pending[i]) (void)hipEventSynchronize(g_ring.ev[i]);
            (void)hipEventDestroy(g_ring.ev[i]);
            g_ring.pending[i] = false;
        }
        (void)hipHostFree(g_ring.base);
        g_ring.base = nullptr;
        g_ring.next = 0;
    }
}
const bool g_release_registered = (tdk::on_device_release(release_map_state), true);

tdk_status h2d_small(int slot, const void *host, size_t bytes, void **dev) {
    if (bytes > kRingBytes) {   // e.g. thousands of reference frames: plain copy, then wait
        TDK_TRY(h2d(slot, host, bytes, dev));
        TDK_HIP(hipStreamSynchronize(tdk::stream()));
        return TDK_OK;
    }
    TDK_TRY(tdk::ensure_device());
    if (!g_ring.base) {
        TDK_HIP(hipHostMalloc((void **)&g_ring.base, kRingSlots * kRingBytes, hipHostMallocDefault));
        for (int i = 0; i < kRingSlots; i++) TDK_HIP(hipEventCreateWithFlags(&g_ring.ev[i], hipEventDisableTiming));
    }
    const int i = g_ring.next;
    g_ring.next = (i + 1) % kRingSlots;
    if (g_ring.pending[i]) TDK_HIP(hipEventSynchronize(g_ring.ev[i]));
    char *block = g_ring.base + (size_t)i * kRingBytes;
    memcpy(block, host, bytes);
    TDK_TRY(tdk::scratch(slot, bytes, dev));
    TDK_HIP(hipMemcpyAsync(*dev, block, bytes, hipMemcpyHostToDevice, tdk::stream()));
    TDK_HIP(hipEventRecord(g_ring.ev[i], tdk::stream()));
    g_ring.pending[i] = true;
    return TDK_OK;
}

__global__ __launch_bounds__(kBlock) void k_safe_invert(const double *__restrict__ v, double eps, double *__restrict__ out,
                                                        int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        out[i] = 1.0 / (v[i] + eps);   // tadataka/numeric.py:1-2
}

bool same_shape(const tdk_map *a, const tdk_map *b) { return a->H == b->H && a->W == b->W; }

constexpr uint64_t kNoBound = ~0ull;

uint64_t max_u64(const void *host, size_t n) {
    const uint64_t *p = (const uint64_t *)host;
    uint64_t m0 = 0, m1 = 0, m2 = 0, m3 = 0;
    size_t i = 0;
    for (; i + 4 <= n; i += 4) {
        m0 = p[i] > m0 ? p[i] : m0; m1 = p[i + 1] > m1 ? p[i + 1] : m1;
        m2 = p[i + 2] > m2 ? p[i + 2] : m2; m3 = p[i + 3] > m3 ? p[i + 3] : m3;
    }
    for (; i < n; i++) m0 = p[i] > m0 ? p[i] : m0;
    m0 = m0 > m1 ? m0 : m1; m2 = m2 > m3 ? m2 : m3;
    return m0 > m2 ? m0 : m2;
}

}  // namespace

extern "C" {

tdk_status tdk_map_create(int height, int width, const void *host, tdk_map **out) {
    TDK_API_GUARD;
    TDK_REQUIRE(out != nullptr, "null pointer");
    TDK_TRY(check_image_dims(height, width));
    TDK_TRY(tdk::ensure_device());
    tdk_map *m = new tdk_map();
    m->H = height; m->W = width;
    const size_t bytes = (size_t)height * width * 8;
    tdk_status st = map_alloc(bytes, &m->data);
    if (st != TDK_OK) { delete m; return st; }
    m->bound = kNoBound;
    if (host) {
        // stream-ordered after the last queued reader of a pooled buffer
        hipError_t e = hipMemcpyAsync(m->data, host, bytes, hipMemcpyHostToDevice, tdk::stream());
        m->bound = max_u64(host, (size_t)height * width);              // while the copy runs
        if (e == hipSuccess) e = hipStreamSynchronize(tdk::stream());   // the caller's array may go away
        if (e != hipSuccess) {
            map_release(bytes, m->data);
            delete m;
            tdk::set_error("map upload failed: %s", hipGetErrorString(e));
            return TDK_ERR_HIP;
        }
    }
    *out = m;
    return TDK_OK;
}

tdk_status tdk_map_destroy(tdk_map *m) {
    TDK_API_GUARD;
    if (!m) return TDK_OK;
    map_release((size_t)m->H * m->W * 8, m->data);
    delete m;
    return TDK_OK;
}

tdk_status tdk_map_upload(tdk_map *m, const void *host) {
    TDK_API_GUARD;
    TDK_REQUIRE(m && host, "null pointer");
    TDK_HIP(hipMemcpyAsync(m->data, host, (size_t)m->H * m->W * 8, hipMemcpyHostToDevice, tdk::stream()));
    m->bound = max_u64(host, (size_t)m->H * m->W);
    TDK_HIP(hipStreamSynchronize(tdk::stream()));
    return TDK_OK;
}

tdk_status tdk_map_download(const tdk_map *m, void *host) {
    TDK_API_GUARD;
    TDK_REQUIRE(m && host, "null pointer");
    TDK_HIP(hipMemcpyAsync(host, m->data, (size_t)m->H * m->W * 8, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipStreamSynchronize(tdk::stream()));   // also waits for the kernels that produce the map
    return TDK_OK;
}

tdk_status tdk_map_shape(const tdk_map *m, int *height, int *width) {
    TDK_API_GUARD;
    TDK_REQUIRE(m && height && width, "null pointer");
    *height = m->H; *width = m->W;
    return TDK_OK;
}

tdk_status tdk_map_device_ptr(const tdk_map *m, void **ptr) {
    TDK_API_GUARD;
    TDK_REQUIRE(m && ptr, "null pointer");
    *ptr = m->data;
    return TDK_OK;
}

tdk_status tdk_frame_download(const tdk_frame *f, double *image) {
    TDK_API_GUARD;
    TDK_REQUIRE(f && image, "null pointer");
    TDK_HIP(hipMemcpyAsync(image, f->image, (size_t)f->H * f->W * 8, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipStreamSynchronize(tdk::stream()));
    return TDK_OK;
}

tdk_status tdk_frame_device_ptr(const tdk_frame *f, void **ptr) {
    TDK_API_GUARD;
    TDK_REQUIRE(f && ptr, "null pointer");
    *ptr = f->image;
    return TDK_OK;
}

tdk_status tdk_map_safe_invert(const tdk_map *v, double epsilon, tdk_map *out) {
    TDK_API_GUARD;
    TDK_REQUIRE(v && out && same_shape(v, out), "maps must share one shape");
    const int64_t n = (int64_t)v->H * v->W;
    int g = grid_for(n);
    k_safe_invert<<<g > 4096 ? 4096 : g, kBlock, 0, tdk::stream()>>>((const double *)v->data, epsilon,
                                                                      (double *)out->data, n);
    TDK_LAUNCH_CHECK();
    out->bound = kNoBound;
    return TDK_OK;
}

tdk_status tdk_increment_age_maps(const tdk_map *age0, const double *camera0, const double *camera1,
                                  const double *T10, const tdk_map *depth0, tdk_map *age1) {
    TDK_API_GUARD;
    TDK_REQUIRE(age0 && camera0 && camera1 && T10 && depth0 && age1, "null pointer");
    TDK_REQUIRE(same_shape(age0, depth0) && same_shape(age0, age1), "maps must share one shape");
    TDK_REQUIRE(age1->data != age0->data, "age1 must not alias age0");
    const int H = age0->H, W = age0->W, N = H * W;
    void *d_lists, *d_tw;
    TDK_TRY(tdk::scratch(2, warp_list_bytes(N, 1), &d_lists));
    TrackWarp tw;
    fill_track_warp(&tw, T10, camera0, camera1);
    TDK_TRY(h2d_small(10, &tw, sizeof(tw), &d_tw));
    // age1 = age0 + 1 where a source pixel lands, 0 elsewhere (age.rs:6-32)
    age1->bound = age0->bound >= kNoBound - 1 ? kNoBound : age0->bound + 1;
    return launch_warp_step<true, false>(1, H, W, (const TrackWarp *)d_tw, (const uint64_t *)age0->data,
                                         (const double *)depth0->data, nullptr, N, 0., 0., 0., (int *)d_lists,
                                         (uint64_t *)age1->data, nullptr, nullptr, tdk::stream());
}

tdk_status tdk_propagate_maps(const double *T10, const double *camera0, const double *camera1,
                              const tdk_map *depth0, const tdk_map *variance0, double default_depth,
                              double default_variance, double uncertaintity_bias, tdk_map *depth1,
                              tdk_map *variance1) {
    TDK_API_GUARD;
    TDK_REQUIRE(T10 && camera0 && camera1 && depth0 && variance0 && depth1 && variance1, "null pointer");
    TDK_REQUIRE(same_shape(depth0, variance0) && same_shape(depth0, depth1) && same_shape(depth0, variance1),
                "maps must share one shape");
    TDK_REQUIRE(depth1->data != depth0->data && depth1->data != variance0->data &&
                    variance1->data != depth0->data && variance1->data != variance0->data &&
                    depth1->data != variance1->data,
                "outputs must not alias the inputs");
    const int H = depth0->H, W = depth0->W, N = H * W;
    void *d_lists, *d_tw;
    TDK_TRY(tdk::scratch(2, warp_list_bytes(N, 1), &d_lists));
    TrackWarp tw;
    fill_track_warp(&tw, T10, camera0, camera1);
    TDK_TRY(h2d_small(10, &tw, sizeof(tw), &d_tw));
    depth1->bound = variance1->bound = kNoBound;
    return launch_warp_step<false, true>(1, H, W, (const TrackWarp *)d_tw, nullptr, (const double *)depth0->data,
                                         (const double *)variance0->data, N, default_depth, default_variance,
                                         uncertaintity_bias, (int *)d_lists, nullptr,
                                         (double *)depth1->data, (double *)variance1->data, tdk::stream());
}

tdk_status tdk_update_depth_maps(const double *key_camera, const tdk_frame *key_frame, const double *key_T, int n_ref,
                                 const double *ref_cameras, const tdk_frame *const *ref_frames, const double *ref_Ts,
                                 const tdk_map *age, const tdk_map *prior_depth, const tdk_map *prior_variance,
                                 const tdk_semi_dense_params *params, tdk_map *depth, tdk_map *variance,
                                 tdk_map *flag) {
    TDK_API_GUARD;
    TDK_REQUIRE(key_camera && key_frame && key_T && age && prior_depth && prior_variance && params && depth &&
                    variance && flag && n_ref >= 0,
                "bad argument");
    TDK_REQUIRE(n_ref == 0 || (ref_cameras && ref_frames && ref_Ts), "null reference frames");
    const int H = key_frame->H, W = key_frame->W, N = H * W;
    for (const tdk_map *m : {age, prior_depth, prior_variance, (const tdk_map *)depth, (const tdk_map *)variance,
                             (const tdk_map *)flag})
        TDK_REQUIRE(m->H == H && m->W == W, "maps and keyframe image must share one shape");
    for (const tdk_map *o : {(const tdk_map *)depth, (const tdk_map *)variance, (const tdk_map *)flag})
        TDK_REQUIRE(o->data != age->data && o->data != prior_depth->data && o->data != prior_variance->data,
                    "outputs must not alias the inputs");
    std::vector<RefConst> rcs((size_t)(n_ref > 0 ? n_ref : 1));
    for (int a = 1; a <= n_ref; a++) {   // indexed by age - 1: refframes[n_ref - age] (:207)
        const int r = n_ref - a;
        TDK_REQUIRE(ref_frames[r] && ref_frames[r]->H == H && ref_frames[r]->W == W,
                    "reference frames must have the key frame's shape");
        TDK_TRY(make_ref_const(key_T, ref_Ts + 16 * r, ref_cameras + 4 * r, ref_frames[r]->image, &rcs[a - 1]));
    }
    void *d_rc, *d_keys, *d_list, *d_cnt;
    TDK_TRY(tdk::scratch(1, (size_t)N * 4, &d_list));
    TDK_TRY(tdk::scratch(4, (kCountStride + 1) * sizeof(int), &d_cnt));
    TDK_TRY(h2d_small(11, rcs.data(), sizeof(RefConst) * rcs.size(), &d_rc));
    TrackKey key;
    memcpy(key.cam, key_camera, sizeof(double) * 4);
    key.image = key_frame->image;
    key.n_ref = n_ref;
    key.pad = 0;
    TDK_TRY(h2d_small(12, &key, sizeof(key), &d_keys));
    int *d_err = (int *)d_cnt + kCountStride;
    TDK_HIP(hipMemsetAsync(d_err, 0, sizeof(int), tdk::stream()));
    TDK_TRY(launch_update_depth(1, H, W, (const TrackKey *)d_keys, (const RefConst *)d_rc, n_ref > 0 ? n_ref : 1,
                                (const uint64_t *)age->data, (const double *)prior_depth->data,
                                (const double *)prior_variance->data, N, est_params(params), (int *)d_list,
                                (int *)d_cnt, d_err, (double *)depth->data, (double *)variance->data,
                                (int64_t *)flag->data, tdk::stream()));
    depth->bound = variance->bound = flag->bound = kNoBound;
    // The reference exits the process if some age exceeds len(refframes) (semi_dense.rs:202-205); here that is an
    // error code, and it has to be known before the call returns -- a wait for everything queued so far, unless
    // the host already knows that no age can (an uploaded map's maximum, + 1 per tdk_increment_age_maps): the
    // loop of examples/semi_dense_vo.py, whose refframes grow with the ages, never waits.
    if (age->bound != kNoBound && age->bound <= (uint64_t)n_ref) return TDK_OK;
    void *h_err;
    TDK_TRY(tdk::pinned(2, sizeof(int), &h_err));
    TDK_HIP(hipMemcpyAsync(h_err, d_err, sizeof(int), hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipStreamSynchronize(tdk::stream()));
    if (*(const int *)h_err & SD_ERR_AGE) {
        tdk::set_error("Age exceeds the refframe size");
        return TDK_ERR_AGE_EXCEEDS_REFFRAMES;
    }
    return TDK_OK;
}

tdk_status tdk_estimate_one(const int64_t *u_key, double prior_depth, double prior_variance,
                            const double *key_camera, const double *key_image, const double *key_T,
                            const double *ref_camera, const double *ref_image, const double *ref_T, int H, int W,
                            const tdk_semi_dense_params *params, double *depth, double *variance, int64_t *flag) {
    TDK_API_GUARD;
    TDK_REQUIRE(u_key && key_camera && key_image && key_T && ref_camera && ref_image && ref_T && params &&
                    depth && variance && flag,
                "null pointer");
    TDK_TRY(check_image_dims(H, W));
    TDK_REQUIRE(u_key[0] >= 0 && u_key[0] < W && u_key[1] >= 0 && u_key[1] < H, "u_key outside the image");
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
    void *d_key, *d_ref, *d_rc, *d_out;
    TDK_TRY(h2d(0, key_image, b8, &d_key));
    TDK_TRY(h2d(3, ref_image, b8, &d_ref));
    RefConst rc;
    TDK_TRY(make_ref_const(key_T, ref_T, ref_camera, (const double *)d_ref, &rc));
    TDK_TRY(h2d(10, &rc, sizeof(RefConst), &d_rc));
    TDK_TRY(tdk::scratch(7, 3 * 8, &d_out));
    k_estimate_one<<<1, 1, 0, tdk::stream()>>>(cam_of(key_camera), (const double *)d_key, (const RefConst *)d_rc,
                                               (double)u_key[0], (double)u_key[1], pid, prior_variance, H, W, pr,
                                               (double *)d_out);
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

tdk_status tdk_regularize(const double *depth, const double *variance, const int64_t *flag, int H, int W,
                          double *regularized) {
    TDK_API_GUARD;
    TDK_REQUIRE(depth && variance && flag && regularized, "null pointer");
    TDK_TRY(check_image_dims(H, W));
    size_t b8 = (size_t)H * W * 8;
    void *d_d, *d_v, *d_f, *d_o;
    TDK_TRY(h2d(0, depth, b8, &d_d));
    TDK_TRY(h2d(1, variance, b8, &d_v));
    TDK_TRY(h2d(2, flag, b8, &d_f));
    TDK_TRY(tdk::scratch(3, b8, &d_o));
    k_regularize<<<grid_for((int64_t)H * W), kBlock, 0, tdk::stream()>>>(
        (const double *)d_d, (const double *)d_v, (const int64_t *)d_f, H, W, (double *)d_o);
    TDK_LAUNCH_CHECK();
    TDK_HIP(hipMemcpyAsync(regularized, d_o, b8, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipStreamSynchronize(tdk::stream()));
    return TDK_OK;
}

tdk_status tdk_fusion_arrays(const double *mu1, const double *mu2, const double *var1, const double *var2,
                             int64_t n, double *mu, double *var) {
    TDK_API_GUARD;
    TDK_REQUIRE(n >= 0 && (n == 0 || (mu1 && mu2 && var1 && var2 && mu && var)), "null pointer");
    if (n == 0) return TDK_OK;
    size_t b8 = (size_t)n * 8;
    void *d_m1, *d_m2, *d_v1, *d_v2, *d_m, *d_v;
    TDK_TRY(h2d(0, mu1, b8, &d_m1));
    TDK_TRY(h2d(1, mu2, b8, &d_m2));
    TDK_TRY(h2d(2, var1, b8, &d_v1));
    TDK_TRY(h2d(3, var2, b8, &d_v2));
    TDK_TRY(tdk::scratch(4, b8, &d_m));
    TDK_TRY(tdk::scratch(5, b8, &d_v));
    k_fusion<<<grid_for(n), kBlock, 0, tdk::stream()>>>((const double *)d_m1, (const double *)d_m2,
                                                        (const double *)d_v1, (const double *)d_v2, n,
                                                        (double *)d_m, (double *)d_v);
    TDK_LAUNCH_CHECK();
    TDK_HIP(hipMemcpyAsync(mu, d_m, b8, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipMemcpyAsync(var, d_v, b8, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipStreamSynchronize(tdk::stream()));
    return TDK_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// tdk_sd: device-resident session
// ---------------------------------------------------------------------------
struct tdk_sd {
    hipStream_t stream;
    int n, H, W, R;          // tracks, frame size, max reference frames; the ring holds R + 1 frames
    int N;
    int64_t stride;          // elements between consecutive tracks in every map
    double *images;          // [n][R + 1][stride]
    // maps: cur = state, out = results of the last step, prior = propagate output
    uint64_t *age[2];
    double *depth[2], *var[2];
    double *prior_depth, *prior_var;
    int64_t *flag;
    int cur, result_buf;
    int *warp_lists;         // cnt | slots | ovf | next of every track (carve_lists)
    int *list, *count, *err;
    unsigned long long *hist;
    TrackWarp *d_tw;
    TrackKey *d_keys;
    RefConst *d_refs;
    const double **d_img_ptrs;   // [2][n]: previous / newest frame of every track
    void *stage;                 // pinned staging for the per-step constants
    size_t stage_bytes;
    std::vector<int64_t> n_frames;             // frames pushed per track
    std::vector<double> cams, Twf;             // [n][R + 1][4], [n][R + 1][16] by ring slot
    std::vector<char> has_T;
    tdk_semi_dense_params params;
    double default_depth, default_variance, bias;
    bool params_set, have_result, result_has_flag;
    bool saturate_age;       // tdk_sd_set_age_policy
    unsigned int *d_warp_fallbacks;   // tracks x steps whose forward warp took the slot path (tdk_sd_get_warp_fallbacks)
    hipEvent_t ev[4];
    double ms[3];
};

namespace {

int sd_slot(const tdk_sd *h, int64_t frame_index) { return (int)(frame_index % (h->R + 1)); }

double *sd_image(const tdk_sd *h, int track, int slot) {
    return h->images + ((size_t)track * (h->R + 1) + slot) * (size_t)h->stride;
}

tdk_status sd_check_track(const tdk_sd *h, int track) {
    TDK_REQUIRE(h != nullptr, "handle is NULL");
    TDK_REQUIRE(track >= 0 && track < h->n, "track out of range");
    return TDK_OK;
}

}  // namespace

extern "C" {

tdk_status tdk_sd_destroy(tdk_sd *h) {
    TDK_API_GUARD;
    if (!h) return TDK_OK;
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    (void)hipFree(h->images);
    for (int k = 0; k < 2; k++) { (void)hipFree(h->age[k]); (void)hipFree(h->depth[k]); (void)hipFree(h->var[k]); }
    (void)hipFree(h->prior_depth); (void)hipFree(h->prior_var); (void)hipFree(h->flag);
    (void)hipFree(h->warp_lists); (void)hipFree(h->list); (void)hipFree(h->count);
    (void)hipFree(h->d_warp_fallbacks);
    (void)hipFree(h->hist); (void)hipFree(h->d_tw); (void)hipFree(h->d_keys); (void)hipFree(h->d_refs);
    (void)hipFree((void *)h->d_img_ptrs);
    if (h->stage) (void)hipHostFree(h->stage);
    for (hipEvent_t e : h->ev) if (e) (void)hipEventDestroy(e);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return TDK_OK;
}

tdk_status tdk_sd_create(int n_tracks, int height, int width, int max_refframes, tdk_sd **out) {
    TDK_API_GUARD;
    TDK_REQUIRE(out != nullptr, "out is NULL");
    TDK_REQUIRE(n_tracks >= 1 && n_tracks <= 65535, "n_tracks must be in [1, 65535]");
    TDK_REQUIRE(max_refframes >= 1 && max_refframes <= 1024, "max_refframes must be in [1, 1024]");
    TDK_TRY(check_image_dims(height, width));
    TDK_REQUIRE((int64_t)height * width < (1ll << 30), "frame too large");
    TDK_TRY(tdk::ensure_device());
    tdk_sd *h = new tdk_sd();   // value-initialised: every pointer is null until allocated
    h->n = n_tracks; h->H = height; h->W = width; h->R = max_refframes;
    h->N = height * width;
    h->stride = ((int64_t)h->N + 1) & ~1ll;
    h->cur = 0; h->result_buf = 0;
    h->params_set = false; h->have_result = false; h->result_has_flag = false;
    h->saturate_age = false;  // the reference's rule; saturation is opt-in (tdk_sd_set_age_policy)
    h->n_frames.assign((size_t)n_tracks, 0);
    h->cams.assign((size_t)n_tracks * (h->R + 1) * 4, 0.0);
    h->Twf.assign((size_t)n_tracks * (h->R + 1) * 16, 0.0);
    h->has_T.assign((size_t)n_tracks * (h->R + 1), 0);
    const size_t m = (size_t)h->stride * n_tracks;
    h->stage_bytes = (sizeof(TrackWarp) + sizeof(TrackKey) + sizeof(RefConst) * h->R + 2 * sizeof(double *)) *
                         (size_t)n_tracks + 64;
#define SD_ALLOC(call)                          \
    do {                                        \
        hipError_t e_ = (call);                 \
        if (e_ != hipSuccess) {                 \
            tdk::set_error("%s failed: %s", #call, hipGetErrorString(e_)); \
            tdk_sd_destroy(h);                  \
            return TDK_ERR_HIP;                 \
        }                                       \
    } while (0)
    SD_ALLOC(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    SD_ALLOC(hipMalloc(&h->images, sizeof(double) * m * (h->R + 1)));
    for (int k = 0; k < 2; k++) {
        SD_ALLOC(hipMalloc(&h->age[k], 8 * m));
        SD_ALLOC(hipMalloc(&h->depth[k], 8 * m));
        SD_ALLOC(hipMalloc(&h->var[k], 8 * m));
    }
    SD_ALLOC(hipMalloc(&h->prior_depth, 8 * m));
    SD_ALLOC(hipMalloc(&h->prior_var, 8 * m));
    SD_ALLOC(hipMalloc(&h->flag, 8 * m));
    SD_ALLOC(hipMalloc(&h->warp_lists, warp_list_bytes(h->stride, n_tracks)));
    SD_ALLOC(hipMalloc(&h->d_warp_fallbacks, sizeof(unsigned int)));
    SD_ALLOC(hipMemset(h->d_warp_fallbacks, 0, sizeof(unsigned int)));
    SD_ALLOC(hipMalloc(&h->list, 4 * m));
    SD_ALLOC(hipMalloc(&h->count, sizeof(int) * ((size_t)kCountStride * n_tracks + 1)));
    h->err = h->count + (size_t)kCountStride * n_tracks;
    SD_ALLOC(hipMalloc(&h->hist, sizeof(unsigned long long) * 10 * n_tracks));
    SD_ALLOC(hipMalloc(&h->d_tw, sizeof(TrackWarp) * n_tracks));
    SD_ALLOC(hipMalloc(&h->d_keys, sizeof(TrackKey) * n_tracks));
    SD_ALLOC(hipMalloc(&h->d_refs, sizeof(RefConst) * (size_t)h->R * n_tracks));
    SD_ALLOC(hipMalloc((void **)&h->d_img_ptrs, sizeof(double *) * 2 * n_tracks));
    SD_ALLOC(hipHostMalloc(&h->stage, h->stage_bytes, hipHostMallocDefault));
    for (int k = 0; k < 4; k++) SD_ALLOC(hipEventCreate(&h->ev[k]));
    // fresh maps: age 0 everywhere, depth / variance undefined until tdk_sd_set_maps
    SD_ALLOC(hipMemsetAsync(h->age[0], 0, 8 * m, h->stream));
    SD_ALLOC(hipMemsetAsync(h->depth[0], 0, 8 * m, h->stream));
    SD_ALLOC(hipMemsetAsync(h->var[0], 0, 8 * m, h->stream));
    SD_ALLOC(hipStreamSynchronize(h->stream));
#undef SD_ALLOC
    *out = h;
    return TDK_OK;
}

tdk_status tdk_sd_set_params(tdk_sd *h, const tdk_semi_dense_params *params, double default_depth,
                             double default_variance, double uncertaintity_bias) {
    TDK_API_GUARD;
    TDK_REQUIRE(h && params, "null pointer");
    h->params = *params;
    h->default_depth = default_depth;
    h->default_variance = default_variance;
    h->bias = uncertaintity_bias;
    h->params_set = true;
    return TDK_OK;
}

tdk_status tdk_sd_set_age_policy(tdk_sd *h, int saturate) {
    TDK_API_GUARD;
    TDK_REQUIRE(h != nullptr, "handle is NULL");
    h->saturate_age = saturate != 0;
    return TDK_OK;
}

tdk_status tdk_sd_get_warp_fallbacks(tdk_sd *h, int64_t *tracks) {
    TDK_API_GUARD;
    TDK_REQUIRE(h && tracks, "null pointer");
    unsigned int v = 0;
    TDK_HIP(hipMemcpyAsync(&v, h->d_warp_fallbacks, sizeof(v), hipMemcpyDeviceToHost, h->stream));
    TDK_HIP(hipStreamSynchronize(h->stream));
    *tracks = (int64_t)v;
    return TDK_OK;
}

tdk_status tdk_sd_set_maps(tdk_sd *h, int track, const double *depth, const double *variance,
                           const uint64_t *age) {
    TDK_API_GUARD;
    TDK_TRY(sd_check_track(h, track));
    const size_t off = (size_t)track * h->stride, b8 = (size_t)h->N * 8;
    if (depth) TDK_HIP(hipMemcpyAsync(h->depth[h->cur] + off, depth, b8, hipMemcpyHostToDevice, h->stream));
    if (variance) TDK_HIP(hipMemcpyAsync(h->var[h->cur] + off, variance, b8, hipMemcpyHostToDevice, h->stream));
    if (age) TDK_HIP(hipMemcpyAsync(h->age[h->cur] + off, age, b8, hipMemcpyHostToDevice, h->stream));
    TDK_HIP(hipStreamSynchronize(h->stream));
    return TDK_OK;
}

static tdk_status sd_read(tdk_sd *h, int which, int track, double *depth, double *variance, uint64_t *age,
                          int64_t *flag) {
    const size_t off = (size_t)track * h->stride, b8 = (size_t)h->N * 8;
    if (depth) TDK_HIP(hipMemcpyAsync(depth, h->depth[which] + off, b8, hipMemcpyDeviceToHost, h->stream));
    if (variance) TDK_HIP(hipMemcpyAsync(variance, h->var[which] + off, b8, hipMemcpyDeviceToHost, h->stream));
    if (age) TDK_HIP(hipMemcpyAsync(age, h->age[which] + off, b8, hipMemcpyDeviceToHost, h->stream));
    if (flag) TDK_HIP(hipMemcpyAsync(flag, h->flag + off, b8, hipMemcpyDeviceToHost, h->stream));
    TDK_HIP(hipStreamSynchronize(h->stream));
    return TDK_OK;
}

tdk_status tdk_sd_get_maps(tdk_sd *h, int track, double *depth, double *variance, uint64_t *age,
                           int64_t *flag) {
    TDK_API_GUARD;
    TDK_TRY(sd_check_track(h, track));
    TDK_REQUIRE(flag == nullptr || (h->have_result && h->result_has_flag), "no update_depth has run yet: there is no flag map");
    return sd_read(h, h->cur, track, depth, variance, age, flag);
}

tdk_status tdk_sd_get_results(tdk_sd *h, int track, double *depth, double *variance, uint64_t *age,
                              int64_t *flag) {
    TDK_API_GUARD;
    TDK_TRY(sd_check_track(h, track));
    TDK_REQUIRE(h->have_result, "no step has run yet");
    TDK_REQUIRE(flag == nullptr || h->result_has_flag, "the last call (tdk_sd_propagate) produced no flag map");
    return sd_read(h, h->result_buf, track, depth, variance, age, flag);
}

tdk_status tdk_sd_push_frame(tdk_sd *h, int track, const double *camera, const double *image,
                             const double *transform_wf) {
    TDK_API_GUARD;
    TDK_TRY(sd_check_track(h, track));
    TDK_REQUIRE(camera && image, "null pointer");
    const int64_t idx = h->n_frames[track];
    const int slot = sd_slot(h, idx);
    const size_t s = (size_t)track * (h->R + 1) + slot;
    memcpy(&h->cams[4 * s], camera, sizeof(double) * 4);
    h->has_T[s] = transform_wf != nullptr;
    if (transform_wf) memcpy(&h->Twf[16 * s], transform_wf, sizeof(double) * 16);
    TDK_HIP(hipMemcpyAsync(sd_image(h, track, slot), image, (size_t)h->N * 8, hipMemcpyHostToDevice, h->stream));
    TDK_HIP(hipStreamSynchronize(h->stream));
    h->n_frames[track] = idx + 1;
    return TDK_OK;
}

}  // extern "C"

namespace {

// Per-step constants, assembled in pinned memory [TrackWarp n][TrackKey n][RefConst n R][ptrs 2 n][err]
// and copied to the device on the session's stream.
struct SdStage {
    TrackWarp *tw;
    TrackKey *keys;
    RefConst *refs;
    const double **ptrs;
    int *err;
};

SdStage sd_stage(tdk_sd *h) {
    SdStage st;
    st.tw = (TrackWarp *)h->stage;
    st.keys = (TrackKey *)(st.tw + h->n);
    st.refs = (RefConst *)(st.keys + h->n);
    st.ptrs = (const double **)(st.refs + (size_t)h->n * h->R);
    st.err = (int *)(st.ptrs + 2 * (size_t)h->n);
    return st;
}

tdk_status sd_require_frames(const tdk_sd *h, int64_t need) {
    for (int t = 0; t < h->n; t++)
        if (h->n_frames[t] < need) {
            tdk::set_error("track %d holds %lld frame(s), this call needs %lld", t, (long long)h->n_frames[t],
                           (long long)need);
            return TDK_ERR_INVALID_ARGUMENT;
        }
    return TDK_OK;
}

// warp constants of every track: previous frame -> newest frame
tdk_status sd_upload_warp(tdk_sd *h, const double *transforms10) {
    TDK_TRY(sd_require_frames(h, 2));
    SdStage st = sd_stage(h);
    for (int t = 0; t < h->n; t++) {
        const int64_t nf = h->n_frames[t];
        const size_t kb = (size_t)t * (h->R + 1) + sd_slot(h, nf - 1), pb = (size_t)t * (h->R + 1) + sd_slot(h, nf - 2);
        // the update_depth that follows sees n_ref = min(frames - 1, R) reference frames; with the
        // saturating policy a pixel tracked for longer keeps using the oldest frame of the ring
        const uint64_t cap = h->saturate_age ? (uint64_t)(nf - 1 < h->R ? nf - 1 : h->R) : ~0ull;
        fill_track_warp(&st.tw[t], transforms10 + 16 * (size_t)t, &h->cams[4 * pb], &h->cams[4 * kb], cap);
    }
    TDK_HIP(hipMemcpyAsync(h->d_tw, st.tw, sizeof(TrackWarp) * h->n, hipMemcpyHostToDevice, h->stream));
    return TDK_OK;
}

// key frame + reference frame constants of every track (update_depth)
tdk_status sd_upload_keys(tdk_sd *h, const double *key_transforms_wf) {
    TDK_TRY(sd_require_frames(h, 1));
    SdStage st = sd_stage(h);
    const int n = h->n, R = h->R;
    for (int t = 0; t < n; t++) {
        const int64_t nf = h->n_frames[t];
        const int ks = sd_slot(h, nf - 1);
        const size_t kb = (size_t)t * (R + 1) + ks;
        if (key_transforms_wf) {
            memcpy(&h->Twf[16 * kb], key_transforms_wf + 16 * (size_t)t, sizeof(double) * 16);
            h->has_T[kb] = 1;
        }
        TDK_REQUIRE(h->has_T[kb], "the newest frame has no transform_wf");
        memcpy(st.keys[t].cam, &h->cams[4 * kb], sizeof(double) * 4);
        st.keys[t].image = sd_image(h, t, ks);
        const int n_ref = (int)(nf - 1 < R ? nf - 1 : R);
        st.keys[t].n_ref = n_ref;
        st.keys[t].pad = 0;
        for (int a = 1; a <= R; a++) {
            RefConst *rc = &st.refs[(size_t)t * R + (a - 1)];
            if (a > n_ref) { memset(rc, 0, sizeof(*rc)); continue; }
            const int rs = sd_slot(h, nf - 1 - a);
            const size_t rb = (size_t)t * (R + 1) + rs;
            TDK_REQUIRE(h->has_T[rb], "a reference frame has no transform_wf");
            TDK_TRY(make_ref_const(&h->Twf[16 * kb], &h->Twf[16 * rb], &h->cams[4 * rb], sd_image(h, t, rs), rc));
        }
    }
    TDK_HIP(hipMemcpyAsync(h->d_keys, st.keys, sizeof(TrackKey) * n, hipMemcpyHostToDevice, h->stream));
    TDK_HIP(hipMemcpyAsync(h->d_refs, st.refs, sizeof(RefConst) * (size_t)n * R, hipMemcpyHostToDevice, h->stream));
    return TDK_OK;
}

// closes a session call: optional flag histogram, error word, sync, timing
tdk_status sd_finish(tdk_sd *h, bool with_flags, int64_t *flag_histogram, int result_buf, int commit) {
    hipStream_t s = h->stream;
    SdStage st = sd_stage(h);
    if (flag_histogram && with_flags) {
        TDK_HIP(hipMemsetAsync(h->hist, 0, sizeof(unsigned long long) * 10 * h->n, s));
        dim3 grid(64, h->n);
        k_flag_hist<<<grid, kBlock, 0, s>>>(h->N, h->flag, h->stride, h->hist);
        TDK_LAUNCH_CHECK();
        TDK_HIP(hipMemcpyAsync(flag_histogram, h->hist, sizeof(int64_t) * 10 * h->n, hipMemcpyDeviceToHost, s));
    }
    TDK_HIP(hipMemcpyAsync(st.err, h->err, sizeof(int), hipMemcpyDeviceToHost, s));
    TDK_HIP(hipStreamSynchronize(s));
    float ms01 = 0.f, ms12 = 0.f, ms02 = 0.f;
    TDK_HIP(hipEventElapsedTime(&ms01, h->ev[0], h->ev[1]));
    TDK_HIP(hipEventElapsedTime(&ms12, h->ev[1], h->ev[2]));
    TDK_HIP(hipEventElapsedTime(&ms02, h->ev[0], h->ev[2]));
    h->ms[0] = ms01; h->ms[1] = ms12; h->ms[2] = ms02;
    if (*st.err & SD_ERR_AGE) {
        tdk::set_error("Age exceeds the refframe size");   // the reference exits the process (:202-205)
        return TDK_ERR_AGE_EXCEEDS_REFFRAMES;
    }
    h->result_buf = result_buf;
    h->have_result = true;
    h->result_has_flag = with_flags;
    if (commit) h->cur = result_buf;
    return TDK_OK;
}

}  // namespace

extern "C" {

tdk_status tdk_sd_step(tdk_sd *h, const double *transforms10, const double *key_transforms_wf, int commit,
                       int64_t *flag_histogram) {
    TDK_API_GUARD;
    TDK_REQUIRE(h && transforms10, "null pointer");
    TDK_REQUIRE(h->params_set, "tdk_sd_set_params has not been called");
    TDK_TRY(sd_upload_warp(h, transforms10));
    TDK_TRY(sd_upload_keys(h, key_transforms_wf));
    hipStream_t s = h->stream;
    TDK_HIP(hipMemsetAsync(h->err, 0, sizeof(int), s));
    const int c = h->cur, o = c ^ 1;
    TDK_HIP(hipEventRecord(h->ev[0], s));
    TDK_TRY((launch_warp_step<true, true>(h->n, h->H, h->W, h->d_tw, h->age[c], h->depth[c], h->var[c], h->stride,
                                          h->default_depth, h->default_variance, h->bias, h->warp_lists,
                                          h->age[o], h->prior_depth, h->prior_var, s, h->d_warp_fallbacks)));
    TDK_HIP(hipEventRecord(h->ev[1], s));
    TDK_TRY(launch_update_depth(h->n, h->H, h->W, h->d_keys, h->d_refs, h->R, h->age[o], h->prior_depth,
                                h->prior_var, h->stride, est_params(&h->params), h->list, h->count, h->err,
                                h->depth[o], h->var[o], h->flag, s));
    TDK_HIP(hipEventRecord(h->ev[2], s));
    return sd_finish(h, true, flag_histogram, o, commit);
}

tdk_status tdk_sd_propagate(tdk_sd *h, const double *transforms10, int commit) {
    TDK_API_GUARD;
    TDK_REQUIRE(h && transforms10, "null pointer");
    TDK_REQUIRE(h->params_set, "tdk_sd_set_params has not been called");
    TDK_TRY(sd_upload_warp(h, transforms10));
    hipStream_t s = h->stream;
    TDK_HIP(hipMemsetAsync(h->err, 0, sizeof(int), s));
    const int c = h->cur, o = c ^ 1;
    TDK_HIP(hipEventRecord(h->ev[0], s));
    TDK_TRY((launch_warp_step<true, true>(h->n, h->H, h->W, h->d_tw, h->age[c], h->depth[c], h->var[c], h->stride,
                                          h->default_depth, h->default_variance, h->bias, h->warp_lists,
                                          h->age[o], h->depth[o], h->var[o], s, h->d_warp_fallbacks)));
    TDK_HIP(hipEventRecord(h->ev[1], s));
    TDK_HIP(hipEventRecord(h->ev[2], s));
    return sd_finish(h, false, nullptr, o, commit);
}

tdk_status tdk_sd_update_depth(tdk_sd *h, const double *key_transforms_wf, int commit,
                               int64_t *flag_histogram) {
    TDK_API_GUARD;
    TDK_REQUIRE(h != nullptr, "handle is NULL");
    TDK_REQUIRE(h->params_set, "tdk_sd_set_params has not been called");
    TDK_TRY(sd_upload_keys(h, key_transforms_wf));
    hipStream_t s = h->stream;
    TDK_HIP(hipMemsetAsync(h->err, 0, sizeof(int), s));
    const int c = h->cur, o = c ^ 1;
    TDK_HIP(hipEventRecord(h->ev[0], s));
    TDK_HIP(hipEventRecord(h->ev[1], s));
    TDK_TRY(launch_update_depth(h->n, h->H, h->W, h->d_keys, h->d_refs, h->R, h->age[c], h->depth[c], h->var[c],
                                h->stride, est_params(&h->params), h->list, h->count, h->err, h->depth[o],
                                h->var[o], h->flag, s));
    TDK_HIP(hipEventRecord(h->ev[2], s));
    // the age map is an input only: the result set carries it along (outside the timed region)
    TDK_HIP(hipMemcpyAsync(h->age[o], h->age[c], 8 * (size_t)h->stride * h->n, hipMemcpyDeviceToDevice, s));
    return sd_finish(h, true, flag_histogram, o, commit);
}

tdk_status tdk_sd_export_dvo(tdk_sd *h, tdk_dvo *batch) {
    TDK_API_GUARD;
    TDK_REQUIRE(h && batch, "null pointer");
    tdk::DvoLevel0 L;
    TDK_TRY(tdk::dvo_level0(batch, &L));
    TDK_REQUIRE(L.n_pairs == h->n && L.H == h->H && L.W == h->W,
                "the DVO batch must have one pair per track and the same frame size");
    const int n = h->n;
    const double **ptrs = (const double **)h->stage;
    for (int t = 0; t < n; t++) {
        const int64_t nf = h->n_frames[t];
        TDK_REQUIRE(nf >= 2, "every track needs the previous and the newest frame");
        ptrs[t] = sd_image(h, t, sd_slot(h, nf - 2));
        ptrs[n + t] = sd_image(h, t, sd_slot(h, nf - 1));
    }
    TDK_HIP(hipMemcpyAsync((void *)h->d_img_ptrs, ptrs, sizeof(double *) * 2 * n, hipMemcpyHostToDevice, h->stream));
    dim3 grid(grid_for(h->N), n);
    k_export_dvo<<<grid, kBlock, 0, h->stream>>>(h->N, h->d_img_ptrs, h->d_img_ptrs + n, h->depth[h->cur],
                                                 h->var[h->cur], h->stride, L.I0, L.D0, L.I1, L.W0, L.stride);
    TDK_LAUNCH_CHECK();
    // the batch's own stream continues after the export
    TDK_HIP(hipEventRecord(h->ev[3], h->stream));
    TDK_HIP(hipStreamWaitEvent(L.stream, h->ev[3], 0));
    TDK_HIP(hipStreamSynchronize(h->stream));   // the staging buffer is reused by the next call
    return TDK_OK;
}

tdk_status tdk_sd_get_timing(tdk_sd *h, double *ms3) {
    TDK_API_GUARD;
    TDK_REQUIRE(h && ms3, "null pointer");
    TDK_REQUIRE(h->have_result, "no step has run yet");
    for (int k = 0; k < 3; k++) ms3[k] = h->ms[k];
    return TDK_OK;
}

}  // extern "C"
