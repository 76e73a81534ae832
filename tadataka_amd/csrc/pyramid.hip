// pyramid.hip -- skimage.transform.rescale on the device: the pyramid of the DVO batch and the
// host-pointer tdk_rescale* entries.
//
// What the reference calls (tadataka/vo/dvo/__init__.py:144-148, for EVERY level, level 0 at scale 1.0
// included): skimage.transform.rescale(image, scale) == resize(image, round(shape * scale)):
//     1. scipy.ndimage.gaussian_filter(image, sigma = max(0, (factor - 1) / 2) per axis, mode='mirror'),
//        axis 0 first; an axis with sigma <= 1e-15 is skipped
//     2. warp(): bilinear, sample position of output (oy, ox) = (ay * oy + by, ax * ox + bx) with an affine map
//        that resize() ESTIMATES from three corner correspondences (SVD), taps floor / ceil of the position,
//        boundary 'reflect' (d c b | a b c d | c b a)
//     3. clip=True: numpy.clip to the minimum / maximum of the FILTERED image.
// Pinned against scikit-image 0.18.3 run in the build container (tests/golden/skimage_*.npz; the tests hold a
// CPU restatement of the same pipeline beside it).  The map and the Gaussian kernels are products of the caller's
// NumPy / LAPACK / libm (a few ulp / 1e-13 from their ideal values, differently on every build), so they come
// in from the host: tdk_rescale_skimage, tdk_dvo_set_level_plan (tadataka_amd/rescale_plan.py makes the same
// NumPy calls skimage and scipy make).  Without a plan a level uses the IDEAL map (i + 0.5) * factor - 0.5 and
// kernels from libm's exp -- the reading of rounds 1-4, kept for tdk_rescale / tdk_rescale_anti_aliased.
//
// Compiled with -ffp-contract=off: every product and sum is one IEEE rounding, in scipy's / skimage's order.
//
// Kernels:
//   k_rescale_generic     any radii, any map, taps outside the image reflected: one level of one image per
//                         (blockIdx.x range, y, z); (2 Rr + 1)(2 Rc + 1) loads per tap -- small images, deep
//                         levels, enlargements, level 0 of small batches
//   k_rescale_aa_multi    LDS tiles (vertical Gaussian once per tile, horizontal at the taps), all tiled levels
//                         of a batch in one launch, XCD-major
//   k_pyramid_stream      one pass over the source for the first TWO shrinking levels (radii 1 and 3: ratio 1.5):
//                         full-height strips, one source column per thread, V rows in per-level LDS rings
//   k_level0_rows         the identity-scale level (no filter): a row per wave, 4 cached loads per output
//   k_clip_bounds / k_clip_apply   step 3, see "clip" below
#include "tdk_runtime.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

namespace {

using tdk::AxisMap;
using tdk::PyramidLevelDesc;

constexpr int kMaxGaussRadius = 64;
constexpr int kMaxOut = 16;          // levels per launch (level 0 + 15)

// sample position of output index o along one axis
__host__ __device__ __forceinline__ double axis_pos(const AxisMap &m, int o) {
    return m.ideal ? ((double)o + 0.5) * m.s - 0.5 : m.a * (double)o + m.b;
}

// skimage _shared/interpolation.pxd coord_map, mode 'R'
__host__ __device__ __forceinline__ int skimage_reflect(int coord, int dim) {
    if ((unsigned)coord < (unsigned)dim) return coord;
    if (dim == 1) return 0;
    const int cmax = dim - 1;
    if (coord < 0) {
        const int n = -coord;
        return ((n / cmax) & 1) ? cmax - (n % cmax) : n % cmax;
    }
    return ((coord / cmax) & 1) ? cmax - (coord % cmax) : coord % cmax;
}

// ndimage 'mirror': d c b | a b c d | c b a
__device__ __forceinline__ int mirror_idx(int i, int n) {
    if ((unsigned)i < (unsigned)n) return i;
    if (n == 1) return 0;
    const int p = 2 * (n - 1);
    i %= p;
    if (i < 0) i += p;
    if (i >= n) i = p - i;
    return i;
}

// ---------------------------------------------------------------------------
// clip=True.  The bounds are the extremes of the filtered image, which no kernel here ever forms.  But an
// output is a rounded convex combination of four filtered values (its taps), so it can leave [lo, hi] only
// by rounding, and only if its taps sit within an ulp of an extreme -- a plateau (a constant depth plane,
// a saturated region).  Every kernel therefore tracks, per (image, level), the extremes of its outputs and
// of the taps it evaluated (all of them elements of the filtered image):
//     out_max <= tap_max (<= hi)  and  out_min >= tap_min (>= lo)  and  no NaN output   =>   clip is a no-op.
// Otherwise (rare; decided on the device) k_clip_bounds evaluates the whole filtered image of that
// (image, level) for its true extremes -- numpy's: NaN if any element is NaN -- and k_clip_apply applies
// numpy.clip.  Both launches return at once for every (image, level) that is not flagged.
// ---------------------------------------------------------------------------
struct ClipSlot {
    unsigned long long out_max, out_min, tap_max, tap_min;   // order-preserving keys (enc)
    unsigned long long lo, hi;                               // true bounds, by k_clip_bounds
    unsigned int nan_out, nan_img;
};

__device__ __forceinline__ unsigned long long enc(double x) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(x);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double dec(unsigned long long k) {
    const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m, 64));
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmin(v, __shfl_xor(v, m, 64));
    return v;
}

// v_max_f64 / v_min_f64 as they are: fmax / fmin put a canonicalising v_max_f64 x, x in front of every operand that
// comes from memory (6 of the 16 operations of a tracked output).  A quiet NaN operand is dropped, as with fmax.
__device__ __forceinline__ double raw_max(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double raw_min(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Per thread only the running extremes of the taps live in registers.  An output that could leave the GLOBAL tap
// range must leave the thread's running range first, so only such outputs (rare: a plateau at an extreme, a NaN)
// are recorded -- straight into the slot, behind a plain load that drops what would not move it.
struct ClipTrack {
    double tmax, tmin;
    __device__ __forceinline__ void init() {
        tmax = -INFINITY;
        tmin = INFINITY;
    }
    __device__ __forceinline__ void add(ClipSlot *slot, double out, double f00, double f01, double f10, double f11) {
        tmax = raw_max(tmax, raw_max(raw_max(f00, f01), raw_max(f10, f11)));
        tmin = raw_min(tmin, raw_min(raw_min(f00, f01), raw_min(f10, f11)));
        if (!(out <= tmax && out >= tmin)) record(slot, out);     // also taken by a NaN
    }
    __device__ __noinline__ static void record(ClipSlot *slot, double out) {
        if (out != out) {
            if (!*(volatile unsigned int *)&slot->nan_out) atomicOr(&slot->nan_out, 1u);
            return;
        }
        const unsigned long long k = enc(out);
        if (k > *(volatile unsigned long long *)&slot->out_max) atomicMax(&slot->out_max, k);
        if (k < *(volatile unsigned long long *)&slot->out_min) atomicMin(&slot->out_min, k);
    }
    // every lane of the wave must call this
    __device__ __forceinline__ void flush(ClipSlot *slot) {
        const double c = wave_max(tmax), d = wave_min(tmin);
        if ((threadIdx.x & 63) == 0 && (c != -INFINITY || d != INFINITY)) {
            atomicMax(&slot->tap_max, enc(c));
            atomicMin(&slot->tap_min, enc(d));
        }
    }
};

// The tracker of a level whose taps are the source pixels themselves (the identity-scale level inside the streaming
// kernel): a thread sees every pixel of its column once, so the EXACT extremes of taps and outputs cost two
// operations each per row -- no per-output comparison, no recording path.  A NaN output only sets a flag.
struct ClipTrackExact {
    double tmax, tmin, omax, omin;
    int nan;                                                      // wave-uniform
    __device__ __forceinline__ void init() {
        tmax = omax = -INFINITY;
        tmin = omin = INFINITY;
        nan = 0;
    }
    __device__ __forceinline__ void tap(double x) {
        tmax = raw_max(tmax, x);
        tmin = raw_min(tmin, x);
    }
    __device__ __forceinline__ void out(double v) {
        omax = raw_max(omax, v);
        omin = raw_min(omin, v);
    }
    // called by EVERY lane of the wave (a scalar OR of the comparison's mask): lanes that hold no output of their
    // own carry blends of real pixels, so a NaN among them means a NaN in the image -- which flags the slot anyway
    __device__ __forceinline__ void nan_check(double v, bool live) {   // live: wave-uniform
        const unsigned long long m = __ballot(v != v);
        nan |= (m != 0ull) & live;
    }
    // every lane of the wave must call this
    __device__ __forceinline__ void flush(ClipSlot *slot) {
        const double c = wave_max(tmax), d = wave_min(tmin), e = wave_max(omax), f = wave_min(omin);
        if ((threadIdx.x & 63) == 0) {
            if (c != -INFINITY || d != INFINITY) {
                atomicMax(&slot->tap_max, enc(c));
                atomicMin(&slot->tap_min, enc(d));
            }
            if (e != -INFINITY || f != INFINITY) {
                atomicMax(&slot->out_max, enc(e));
                atomicMin(&slot->out_min, enc(f));
            }
            if (nan) atomicOr(&slot->nan_out, 1u);
        }
    }
};

// (the untouched keys 0 / ~0 decode to NaNs: every comparison with them is false)
__device__ __forceinline__ bool clip_needed(const ClipSlot &s) {
    if (s.nan_out) return true;
    return dec(s.out_max) > dec(s.tap_max) || dec(s.out_min) < dec(s.tap_min);
}

// behind the n slots: an int counter and the list of flagged slots (k_clip_gate)
__device__ __forceinline__ int *clip_list(ClipSlot *slots, int n) { return reinterpret_cast<int *>(slots + n); }

__device__ __forceinline__ void clip_slot_clear(ClipSlot *slot) {
    ClipSlot s;
    s.out_max = 0ull; s.tap_max = 0ull; s.hi = 0ull;
    s.out_min = ~0ull; s.tap_min = ~0ull; s.lo = ~0ull;
    s.nan_out = 0u; s.nan_img = 0u;
    *slot = s;
}

__global__ void k_clip_reset(ClipSlot *slots, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) clip_list(slots, n)[0] = 0;
    if (i >= n) return;
    clip_slot_clear(slots + i);
}

struct AaLevel {
    const double *wr, *wc;   // device: 2 R + 1 weights each (a one-tap kernel {1} for an unfiltered axis)
    int Rr, Rc;
};

// correlate1d with a symmetric kernel along a column of the source image: centre tap first, then the pairs
// from the outermost inwards (scipy ni_filters.c, the `symmetric > 0` branch)
__device__ __forceinline__ double column_tap(const double *__restrict__ s, int H, int W, int y, int x,
                                             const double *__restrict__ w, int R) {
    if (R == 0) return s[(int64_t)y * W + x];                      // gaussian_filter skips the axis
    double tmp = s[(int64_t)y * W + x] * w[R];
    for (int j = -R; j < 0; j++)
        tmp += (s[(int64_t)mirror_idx(y + j, H) * W + x] + s[(int64_t)mirror_idx(y - j, H) * W + x]) * w[R + j];
    return tmp;
}

// ... and along a row of the vertically filtered image
__device__ __forceinline__ double filtered_tap(const double *__restrict__ s, int H, int W, int y, int x,
                                               const AaLevel &a) {
    if (a.Rc == 0) return column_tap(s, H, W, y, x, a.wr, a.Rr);
    double tmp = column_tap(s, H, W, y, x, a.wr, a.Rr) * a.wc[a.Rc];
    for (int j = -a.Rc; j < 0; j++)
        tmp += (column_tap(s, H, W, y, mirror_idx(x + j, W), a.wr, a.Rr) +
                column_tap(s, H, W, y, mirror_idx(x - j, W), a.wr, a.Rr)) * a.wc[a.Rc + j];
    return tmp;
}

struct DevLevel {
    double *dst[4];
    int64_t stride;
    int Ho, Wo;
    AxisMap mx, my;
    AaLevel aa;
};

struct GenericArgs {
    const double *src[4];
    int64_t src_stride;
    int H, W, n;
    int lvl[kMaxOut];        // index of the level (slot numbering)
    int blk_end[kMaxOut];    // cumulative block count per entry
    DevLevel lv[kMaxOut];
    ClipSlot *slots;         // [image][n_out] or null
    int n_arrays, n_out;
};

// One output row per wave, lanes along the row: the general form of skimage's warp (ceil taps, reflected
// coordinates), every tap evaluated from global memory.
__global__ __launch_bounds__(256) void k_rescale_generic(GenericArgs a) {
    int e = 0;
    while (e + 1 < a.n && (int)blockIdx.x >= a.blk_end[e]) e++;
    const DevLevel &L = a.lv[e];
    const int arr = blockIdx.y, pair = blockIdx.z;
    const int oy = (((int)blockIdx.x - (e ? a.blk_end[e - 1] : 0)) << 2) + (int)(threadIdx.x >> 6);
    const int H = a.H, W = a.W;
    const double *s = a.src[arr] + (int64_t)pair * a.src_stride;
    ClipSlot *slot = a.slots ? a.slots + ((int64_t)pair * a.n_arrays + arr) * a.n_out + a.lvl[e] : nullptr;
    ClipTrack tr;
    tr.init();
    if (oy < L.Ho) {
        const double r = axis_pos(L.my, oy);
        const double fr = floor(r);
        const double dr = r - fr;
        const int y0 = skimage_reflect((int)fr, H), y1 = skimage_reflect((int)ceil(r), H);
        double *d = L.dst[arr] + (int64_t)pair * L.stride + (int64_t)oy * L.Wo;
        for (int ox = threadIdx.x & 63; ox < L.Wo; ox += 64) {
            const double c = axis_pos(L.mx, ox);
            const double fc = floor(c);
            const double dc = c - fc;
            const int x0 = skimage_reflect((int)fc, W), x1 = skimage_reflect((int)ceil(c), W);
            const double f00 = filtered_tap(s, H, W, y0, x0, L.aa), f01 = filtered_tap(s, H, W, y0, x1, L.aa);
            const double f10 = filtered_tap(s, H, W, y1, x0, L.aa), f11 = filtered_tap(s, H, W, y1, x1, L.aa);
            const double top = (1.0 - dc) * f00 + dc * f01;
            const double bot = (1.0 - dc) * f10 + dc * f11;
            const double v = (1.0 - dr) * top + dr * bot;
            d[ox] = v;
            if (slot) tr.add(slot, v, f00, f01, f10, f11);
        }
    }
    if (slot) tr.flush(slot);
}

// The identity-scale level (rescale(image, 1.0): no filter, a map a few ulp from the identity): the same
// arithmetic as k_rescale_generic with Rr = Rc = 0, organised for bandwidth -- a wave takes kL0Rows
// consecutive output rows of a 64-column group, the row terms are wave-uniform, the column terms per lane,
// the four taps are plain loads (three of them cache hits).
constexpr int kL0Rows = 8;
struct Level0Args {
    const double *src[4];
    double *dst[4];
    int64_t src_stride, dst_stride;
    int H, W, n_arrays, batch, lvl, n_out;
    AxisMap mx, my;
    ClipSlot *slots;
};

__global__ __launch_bounds__(256) void k_level0_rows(Level0Args a) {
    const int H = a.H, W = a.W;
    const int gx = (W + 63) >> 6, gy = (H + 4 * kL0Rows - 1) / (4 * kL0Rows);
    const int per_image = gx * gy;
    // XCD-major: XCD k takes images k, k + 8, ...
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int images = a.n_arrays * a.batch;
    const bool few = images < 8;
    const int image = few ? (int)blockIdx.x / per_image : (q / per_image) * 8 + xcd;
    const int part = few ? (int)blockIdx.x - image * per_image : q - (q / per_image) * per_image;
    if (image >= images) return;
    const int pair = image / a.n_arrays, arr = image - pair * a.n_arrays;
    const int by = part / gx, bx = part - by * gx;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ox = bx * 64 + lane;
    const int oy0 = (by * 4 + wave) * kL0Rows;
    const double *s = a.src[arr] + (int64_t)pair * a.src_stride;
    double *d = a.dst[arr] + (int64_t)pair * a.dst_stride;
    ClipSlot *slot = a.slots ? a.slots + ((int64_t)pair * a.n_arrays + arr) * a.n_out + a.lvl : nullptr;
    ClipTrack tr;
    tr.init();
    if (ox < W) {
        const double c = axis_pos(a.mx, ox);
        const double fc = floor(c);
        const double dc = c - fc;
        const int x0 = skimage_reflect((int)fc, W), x1 = skimage_reflect((int)ceil(c), W);
#pragma unroll
        for (int i = 0; i < kL0Rows; i++) {
            const int oy = oy0 + i;
            if (oy >= H) break;
            const double r = axis_pos(a.my, oy);
            const double fr = floor(r);
            const double dr = r - fr;
            const int y0 = skimage_reflect((int)fr, H), y1 = skimage_reflect((int)ceil(r), H);
            const double *r0 = s + (int64_t)y0 * W, *r1 = s + (int64_t)y1 * W;
            const double f00 = r0[x0], f01 = r0[x1], f10 = r1[x0], f11 = r1[x1];
            const double top = (1.0 - dc) * f00 + dc * f01;
            const double bot = (1.0 - dc) * f10 + dc * f11;
            const double v = (1.0 - dr) * top + dr * bot;
            d[(int64_t)oy * W + ox] = v;
            if (slot) tr.add(slot, v, f00, f01, f10, f11);
        }
    }
    if (slot) tr.flush(slot);
}

// ---------------------------------------------------------------------------
// LDS tiles.  A block produces tile_rows x kAaCols output pixels of a SHRINKING level (all four taps inside
// the image, x1 = x0 + 1, y1 = y0 + 1 -- checked on the host with the level's own map):
//   1. the vertical Gaussian is evaluated once per (row, column) its taps and their horizontal support touch
//      (mirror boundary), from global memory into an LDS tile V -- ndimage filters axis 0 first, so V is
//      rounded exactly like its intermediate image,
//   2. every thread evaluates the horizontal Gaussian of V at its four taps and blends.
// Same operations in the same order as filtered_tap(): bit-identical with k_rescale_generic.
// ---------------------------------------------------------------------------
constexpr int kAaCols = 64;

struct AaTileArgs {
    const double *src[4];
    double *dst[4];
    int64_t src_stride, dst_stride;
    int H, W, Ho, Wo;
    AxisMap mx, my;
    AaLevel aa;
    int lvl;
    int tile_rows;              // output rows per block
    int max_v_rows, max_cols;   // LDS tile bounds, computed on the host from the map
};

// Any radii (from the arguments): the levels aa_tile_fixed<R> has no instantiation for.
__device__ __forceinline__ void aa_tile_generic(const AaTileArgs &a, int tile, int arr, int pair,
                                                unsigned char *aa_smem, ClipTrack &tr, ClipSlot *slot) {
    const int Rr = a.aa.Rr, Rc = a.aa.Rc;
    const int SC = a.max_cols;
    double *V = reinterpret_cast<double *>(aa_smem);           // [max_v_rows][SC] vertically filtered
    double *wr = V + (size_t)a.max_v_rows * SC;                 // [2 Rr + 1] kernel weights, LDS copies
    double *wc = wr + 2 * Rr + 1;                               // [2 Rc + 1]
    for (int k = threadIdx.x; k < 2 * Rr + 1; k += 256) wr[k] = a.aa.wr[k];
    for (int k = threadIdx.x; k < 2 * Rc + 1; k += 256) wc[k] = a.aa.wc[k];
    const int tiles_x = (a.Wo + kAaCols - 1) / kAaCols;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int H = a.H, W = a.W;
    const double *s = a.src[arr] + (int64_t)pair * a.src_stride;
    const int oy0 = ty * a.tile_rows, oy1 = min(oy0 + a.tile_rows, a.Ho);
    const int ox0 = tx * kAaCols, ox1 = min(ox0 + kAaCols, a.Wo);
    const int yv0 = (int)floor(axis_pos(a.my, oy0));
    const int yv1 = min((int)floor(axis_pos(a.my, oy1 - 1)) + 1, H - 1);
    const int xv0 = (int)floor(axis_pos(a.mx, ox0));
    const int xv1 = min((int)floor(axis_pos(a.mx, ox1 - 1)) + 1, W - 1);
    const int nv = yv1 - yv0 + 1;                                // V rows
    const int nc = xv1 - xv0 + 1 + 2 * Rc;                       // columns incl. the horizontal support
    const int xs0 = xv0 - Rc;
    __syncthreads();                                             // weights in place
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = wave; r < nv; r += 4) {
        const int y = yv0 + r;
        for (int c = lane; c < nc; c += 64) {
            const double *col = s + mirror_idx(xs0 + c, W);
            double tmp;
            if (Rr == 0) tmp = col[(int64_t)y * W];
            else {
                tmp = col[(int64_t)y * W] * wr[Rr];
                for (int j = -Rr; j < 0; j++)
                    tmp += (col[(int64_t)mirror_idx(y + j, H) * W] + col[(int64_t)mirror_idx(y - j, H) * W]) * wr[Rr + j];
            }
            V[r * SC + c] = tmp;
        }
    }
    __syncthreads();
    const int ox = ox0 + (int)(threadIdx.x & 63);
    if (ox >= ox1) return;
    const double cx = axis_pos(a.mx, ox);
    const double fx0 = floor(cx);
    const double wx = cx - fx0;
    const int x0 = (int)fx0;
    for (int oy = oy0 + (int)(threadIdx.x >> 6); oy < oy1; oy += 4) {
        const double cy = axis_pos(a.my, oy);
        const double fy0 = floor(cy);
        const double wy = cy - fy0;
        const int y0 = (int)fy0;
        double f[2][2];
#pragma unroll
        for (int ry = 0; ry < 2; ry++) {
#pragma unroll
            for (int rx = 0; rx < 2; rx++) {
                const double *row = V + (y0 + ry - yv0) * SC + (x0 + rx - xs0);
                double tmp;
                if (Rc == 0) tmp = row[0];
                else {
                    tmp = row[0] * wc[Rc];
                    for (int j = -Rc; j < 0; j++) tmp += (row[j] + row[-j]) * wc[Rc + j];
                }
                f[ry][rx] = tmp;
            }
        }
        const double top = (1.0 - wx) * f[0][0] + wx * f[0][1];
        const double bot = (1.0 - wx) * f[1][0] + wx * f[1][1];
        const double v = (1.0 - wy) * top + wy * bot;
        a.dst[arr][(int64_t)pair * a.dst_stride + (int64_t)oy * a.Wo + ox] = v;
        if (slot) tr.add(slot, v, f[0][0], f[0][1], f[1][0], f[1][1]);
    }
}

// The tile for a compile-time radius R (both axes): the pyramid levels of the bench (R = 1, 3, 5 at ratio
// 1.5).  Same products and sums in the same order as aa_tile_generic / filtered_tap(), so bit-identical; what
// differs is the bookkeeping around them -- on this kernel 70 % of the issued VALU work was integer address
// arithmetic and predication, not the FP64 filter:
//   * the wave index is made scalar (readfirstlane), so row numbers, boundary reflection of rows and row base
//     pointers live in SGPRs: a source load is `global_load v, voff, s[row]` with one per-lane column offset;
//   * a wave loads the CH + 2 R source rows of its column walk unconditionally (rows beyond the wave's share
//     are clamped to the last one and unused): no per-load exec masking;
//   * x1 = x0 + 1 and y1 = y0 + 1, so the two horizontal filters of a V row share 2 R of their 2 R + 1 LDS reads;
//   * the per-row terms (wy and the V offset of the upper tap) are computed once per tile into LDS.
template <int R>
__device__ __forceinline__ void aa_tile_fixed(const AaTileArgs &a, int tile, int arr, int pair,
                                              unsigned char *aa_smem, ClipTrack &tr, ClipSlot *slot) {
    static_assert(R > 0, "compile-time radius");
    constexpr int CH = 8;                                         // V rows per wave held in registers
    const int SC = a.max_cols;
    double *V = reinterpret_cast<double *>(aa_smem);              // [max_v_rows][SC] vertically filtered
    double *row_wy = V + (size_t)a.max_v_rows * SC;               // [tile_rows] weight of the lower row tap
    int *row_off = reinterpret_cast<int *>(row_wy + a.tile_rows); // [tile_rows] V offset of the upper row tap
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tiles_x = (a.Wo + kAaCols - 1) / kAaCols;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int H = a.H, W = a.W;
    const double *s = a.src[arr] + (int64_t)pair * a.src_stride;
    const int oy0 = ty * a.tile_rows, oy1 = min(oy0 + a.tile_rows, a.Ho);
    const int ox0 = tx * kAaCols, ox1 = min(ox0 + kAaCols, a.Wo);
    const int yv0 = (int)floor(axis_pos(a.my, oy0));
    const int yv1 = min((int)floor(axis_pos(a.my, oy1 - 1)) + 1, H - 1);
    const int xv0 = (int)floor(axis_pos(a.mx, ox0));
    const int xv1 = min((int)floor(axis_pos(a.mx, ox1 - 1)) + 1, W - 1);
    const int nv = yv1 - yv0 + 1;                                // V rows
    const int nc = xv1 - xv0 + 1 + 2 * R;                        // columns incl. the horizontal support
    const int xs0 = xv0 - R;
    double wk[R + 1], wck[R + 1];                                // kernel halves (uniform: scalar loads)
#pragma unroll
    for (int k = 0; k <= R; k++) { wk[k] = a.aa.wr[k]; wck[k] = a.aa.wc[k]; }
    if ((int)threadIdx.x < oy1 - oy0) {                          // per-row terms of the blend
        const double cy = axis_pos(a.my, oy0 + (int)threadIdx.x);
        const double fy0 = floor(cy);
        row_wy[threadIdx.x] = cy - fy0;
        row_off[threadIdx.x] = ((int)fy0 - yv0) * SC;
    }
    {
        const int chunk = (nv + 3) >> 2;
        const int r0 = wave * chunk, r1 = min(r0 + chunk, nv);
        const int rows = r1 - r0;                                // wave-uniform
        const int ytop = yv0 + r0 - R;
        const bool inside = ytop >= 0 && yv0 + r1 - 1 + R <= H - 1;   // no reflection needed
        if (rows > 0 && rows <= CH) {
            const double *rowp[CH + 2 * R];                      // uniform row base pointers
#pragma unroll
            for (int k = 0; k < CH + 2 * R; k++) {
                const int y = ytop + (k < rows + 2 * R ? k : rows + 2 * R - 1);
                rowp[k] = s + (int64_t)(inside ? y : mirror_idx(y, H)) * W;
            }
            for (int c = lane; c < nc; c += 64) {
                const unsigned xs = (unsigned)mirror_idx(xs0 + c, W);
                double v[CH + 2 * R];
#pragma unroll
                for (int k = 0; k < CH + 2 * R; k++) v[k] = rowp[k][xs];
#pragma unroll
                for (int i = 0; i < CH; i++) {
                    if (i < rows) {                              // uniform
                        double tmp = v[i + R] * wk[R];
#pragma unroll
                        for (int j = -R; j < 0; j++) tmp += (v[i + R + j] + v[i + R - j]) * wk[R + j];
                        V[(r0 + i) * SC + c] = tmp;
                    }
                }
            }
        } else if (rows > 0) {                                   // taller tiles: sliding window
            for (int c = lane; c < nc; c += 64) {
                const double *col = s + mirror_idx(xs0 + c, W);
                double win[2 * R + 1];
#pragma unroll
                for (int k = 0; k < 2 * R; k++) {
                    const int y = ytop + k;
                    win[k + 1] = col[(int64_t)(inside ? y : mirror_idx(y, H)) * W];
                }
                for (int r = r0; r < r1; r++) {
#pragma unroll
                    for (int k = 0; k < 2 * R; k++) win[k] = win[k + 1];
                    const int y = yv0 + r + R;
                    win[2 * R] = col[(int64_t)(inside ? y : mirror_idx(y, H)) * W];
                    double tmp = win[R] * wk[R];
#pragma unroll
                    for (int j = -R; j < 0; j++) tmp += (win[R + j] + win[R - j]) * wk[R + j];
                    V[r * SC + c] = tmp;
                }
            }
        }
    }
    __syncthreads();
    const int ox = ox0 + lane;
    if (ox >= ox1) return;
    const double cx = axis_pos(a.mx, ox);
    const double fx0 = floor(cx);
    const double wx = cx - fx0;
    const double *Vx = V + ((int)fx0 - xs0);                     // the lane's left tap in V row 0
    double *dst = a.dst[arr] + (int64_t)pair * a.dst_stride;
    for (int i = wave; i < oy1 - oy0; i += 4) {                  // i is wave-uniform
        const double wy = row_wy[i];
        const double *p = Vx + row_off[i];
        double u[2][2 * R + 2];                                  // V rows y0, y0 + 1, columns x0 - R .. x0 + 1 + R
#pragma unroll
        for (int ry = 0; ry < 2; ry++)
#pragma unroll
            for (int q = 0; q < 2 * R + 2; q++) u[ry][q] = p[ry * SC + q - R];
        double f[2][2];
#pragma unroll
        for (int ry = 0; ry < 2; ry++) {
#pragma unroll
            for (int rx = 0; rx < 2; rx++) {
                double tmp = u[ry][R + rx] * wck[R];
#pragma unroll
                for (int j = -R; j < 0; j++) tmp += (u[ry][R + rx + j] + u[ry][R + rx - j]) * wck[R + j];
                f[ry][rx] = tmp;
            }
        }
        const double top = (1.0 - wx) * f[0][0] + wx * f[0][1];
        const double bot = (1.0 - wx) * f[1][0] + wx * f[1][1];
        const double v = (1.0 - wy) * top + wy * bot;
        dst[(int64_t)(oy0 + i) * a.Wo + ox] = v;
        if (slot) tr.add(slot, v, f[0][0], f[0][1], f[1][0], f[1][1]);
    }
}

// Every tiled level of the pyramid in ONE launch: the tiles of level 1, then level 2, ... of one (array, pair)
// are consecutive work items of ONE XCD, so the coarser levels of an image are produced right after the finer
// ones and find the full-resolution source in that L2.
constexpr int kAaMaxFused = 4;
struct AaMultiArgs {
    int n, n_arrays, batch, n_out;
    int tile_end[kAaMaxFused];
    AaTileArgs lv[kAaMaxFused];
    ClipSlot *slots;
};

__global__ __launch_bounds__(256) void k_rescale_aa_multi(AaMultiArgs m) {
    extern __shared__ __attribute__((aligned(16))) unsigned char aa_smem[];
    // 1-D grid.  Workgroups are dealt to the 8 XCDs round-robin and every XCD has its own L2: XCD k takes
    // images k, k + 8, ... one after the other, all tiles of all levels of an image consecutively
    // (fewer than 8 images -- a single pair's three arrays: their tiles go to all XCDs instead)
    const int tiles_total = m.tile_end[m.n - 1];
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const bool few = m.n_arrays * m.batch < 8;
    const int image = few ? (int)blockIdx.x / tiles_total : (q / tiles_total) * 8 + xcd;
    const int t = few ? (int)blockIdx.x - image * tiles_total : q - (q / tiles_total) * tiles_total;
    if (image >= m.n_arrays * m.batch) return;
    const int pair = image / m.n_arrays, arr = image - pair * m.n_arrays;
    int l = 0;
    while (l + 1 < m.n && t >= m.tile_end[l]) l++;
    const int tile = t - (l ? m.tile_end[l - 1] : 0);
    const AaTileArgs &a = m.lv[l];
    const int R = a.aa.Rr == a.aa.Rc ? a.aa.Rr : 0;
    ClipTrack tr;
    tr.init();
    ClipSlot *slot = m.slots ? m.slots + ((int64_t)pair * m.n_arrays + arr) * m.n_out + a.lvl : nullptr;
    switch (R) {      // block-uniform
        case 1: aa_tile_fixed<1>(a, tile, arr, pair, aa_smem, tr, slot); break;   // ratio 1.5, level 1
        case 3: aa_tile_fixed<3>(a, tile, arr, pair, aa_smem, tr, slot); break;   // level 2
        case 5: aa_tile_fixed<5>(a, tile, arr, pair, aa_smem, tr, slot); break;   // level 3
        default: aa_tile_generic(a, tile, arr, pair, aa_smem, tr, slot); break;
    }
    // (lanes that returned early from the tile functions rejoin here: the returns are inlined branches)
    if (slot) tr.flush(slot);
}

// first output index o in [0, n_out] whose lower tap max(floor(pos(o)), 0) is >= s0 (pos is increasing)
__host__ __device__ __forceinline__ int first_owned(int s0, const AxisMap &m, int n_out) {
    const double guess = m.ideal ? ((double)s0 + 0.5) / m.s - 0.5 : ((double)s0 - m.b) / m.a;
    int o = (int)ceil(guess);
    o = o < 0 ? 0 : (o > n_out ? n_out : o);
    while (o > 0 && ((int)floor(axis_pos(m, o - 1)) > 0 ? (int)floor(axis_pos(m, o - 1)) : 0) >= s0) o--;
    while (o < n_out && ((int)floor(axis_pos(m, o)) > 0 ? (int)floor(axis_pos(m, o)) : 0) < s0) o++;
    return o;
}

// ---------------------------------------------------------------------------
// Row-streaming form of the anti-aliased pyramid: one pass over the source for up to TWO levels.
//
// A block owns a full-height strip of source columns, one column per thread, and walks DOWN it:
//   * every source texel is loaded exactly once (rows of the strip are contiguous: coalesced 8-byte loads),
//     the loads of chunk c + 1 are issued before chunk c is computed (software prefetch in registers);
//   * the thread keeps the last 2 RM rows of its column in registers, so the vertical Gaussians of BOTH
//     levels (radius RA and RB) come from the same registers: K new V rows per level per chunk, written to
//     per-level LDS rings;
//   * after a barrier the waves take (output row, 64-column group) units of both levels whose two V rows are
//     now complete: horizontal Gaussian at the four taps from LDS, blend, store -- aa_tile_fixed's arithmetic,
//     operation by operation, so the output is bit-identical.
// V rows live in rings of K + 2 rows per level (two barriers per chunk): 35.8 KB of LDS for a VGA strip,
// 4 blocks per CU.  Measured on 256 VGA pairs x 3 arrays (profiles/r04_pyramid.txt): rings of 2 K + 1 rows
// with one barrier (2 blocks per CU) 1.44 ms, K + 2 rows 0.99, ring pitch = the strip's columns instead of
// 256 0.91; K = 4 / 6 / 8 / 9 / 12: 1.31 / 1.09 / 0.91 / 0.90 / 0.95; the tiles above 1.035.
// ---------------------------------------------------------------------------
constexpr int kStreamK = 8;                       // source rows per chunk
constexpr int kStreamRing = kStreamK + 2;         // V rows per level ring (>= K + 2; < 2 K + 1: a second barrier per chunk)
constexpr int kStreamGroupsA = 3;                 // 64-column groups of outputs per strip: radius 1 (factor >= 1.25; checked on
constexpr int kStreamGroupsB = 2;                 // the host), radius 3 (factor >= 2.25: at most 110 outputs per 248 columns)

struct StreamLevel {
    double *dst[4];
    int64_t dst_stride;
    int Ho, Wo;
    AxisMap mx, my;
    const double *wr, *wc;                        // device: kernel halves incl. centre (R + 1 used)
    int lvl;
};

constexpr int kStreamMaxStrips = 16, kStreamMaxSegs = 16;
struct StreamRange { int a0, a1, b0, b1; };      // first / end output index of the two levels

struct StreamArgs {
    const double *src[4];
    int64_t src_stride;
    int H, W, n_arrays, batch, n_out;
    int n_strips, strip_w;                        // owned source columns per strip
    int pitch;                                    // ring row pitch: strip_w + 2 RM + 1 columns, rounded up (<= 256 threads)
    int wave_cols;                                // columns a wave advances by: 64, or 62 when the identity-scale level rides
                                                  // along (lanes 0 / 63 repeat the neighbour waves' border columns)
    int n_segs, seg_rows;                         // row segments: a block emits the outputs whose upper tap lies in its segment
    StreamLevel lv[2];
    // the identity-scale level (rescale(., 1.0)), written from the same pass: bit `arr` of l0_mask set = this array has one
    double *l0_dst[4];
    int64_t l0_stride;
    AxisMap l0_mx, l0_my;
    int l0_lvl;
    unsigned l0_mask;
    ClipSlot *slots;
    // the output columns of a strip / output rows of a segment (first_owned of their bounds), tabulated by the host:
    // the block reads its two entries instead of searching for them (eight searches: a tenth of a 15-chunk block's time)
    StreamRange strip_tab[kStreamMaxStrips], seg_tab[kStreamMaxSegs];
};

// lane i <- lane i - 1 / lane i + 1 of the wave (DPP wave_shr:1 / wave_shl:1); the lane without a source gets 0
// (bound_ctrl with nothing to preserve: one v_mov_b32_dpp per half, no copy of x in front of it) -- the callers
// never use that lane's value
__device__ __forceinline__ double from_left_lane(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x138, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double from_right_lane(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x130, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// Vector memory operations the compiler does not count (k_pyramid_stream).  gfx950 has ONE counter for loads and stores
// (vmcnt) and they retire in issue order, so "the rows prefetched at the top of the chunk have arrived" can only be
// asked for as "at most N younger operations are in flight" -- and the compiler, which cannot know how many stores the
// data-dependent emission loops issued behind the prefetch, asks with N = 0: every wave drained all its stores at the
// end of every chunk (a quarter of its resident time in s_waitcnt, SQ_WAIT_INST_ANY, profiles/r06_pyramid.txt).  The
// kernel counts for itself: the prefetch and the identity level's stores (the last stores of a chunk, a wave-uniform
// number of them) go through these, and stream_rows_arrived() waits with that number.  Between stream_load() and
// stream_rows_arrived() nothing may read or copy the destination registers: the wait names them all as operands, which
// is the only use the compiler sees (checked in the ISA: tools/check_pyramid_isa.py).
// (the scalar-base form `global_load_dwordx2 v, voff, s[..]` saves a 64-bit vector add per access and costs 44 more
// spilled SGPRs: the same 1.24 ms on the same box)
__device__ __forceinline__ double stream_load(const double *row, unsigned byte_off) {
    double v;
    const double *p = reinterpret_cast<const double *>(reinterpret_cast<const char *>(row) + byte_off);
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p));
    return v;
}
__device__ __forceinline__ void stream_store(double *row, unsigned byte_off, double v) {
    double *p = reinterpret_cast<double *>(reinterpret_cast<char *>(row) + byte_off);
    asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(p), "v"(v) : "memory");
}
// `younger8`: wave-uniform, the last 8 vector memory operations issued are stores that may stay in flight.  ONE asm
// statement for both cases: with two, the register allocator met them with a phi and copied the (not yet arrived)
// registers in front of one of the waits.
__device__ __forceinline__ void stream_rows_arrived(double (&r)[8], int younger8) {
    asm volatile("s_cmp_lg_u32 %8, 0\n\t"
                 "s_cbranch_scc0 1f\n\t"
                 "s_waitcnt vmcnt(8)\n\t"
                 "s_branch 2f\n"
                 "1:\n\t"
                 "s_waitcnt vmcnt(0)\n"
                 "2:"
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                 : "s"(younger8) : "scc");
}

// ---------------------------------------------------------------------------
// The identity-scale level as a streaming kernel of its own (batches whose other levels do not go through
// k_pyramid_stream: one-level pyramids such as BASELINE configs[3]).  A wave owns 62 columns (lanes 1 .. 62; lanes
// 0 and 63 carry the neighbour columns, reflected at the image border) and walks down a segment of 64 rows with the
// rows y - 1, y, y + 1 of its columns in registers: every pixel is loaded once (8 rows ahead), the neighbour column
// comes by DPP, no LDS, no barrier.  Same operations as k_level0_rows / k_rescale_generic: bit-identical.
// ---------------------------------------------------------------------------
constexpr int kL0Cols = 62, kL0Seg = 64, kL0Ahead = 8;

__global__ __launch_bounds__(256) void k_level0_stream(Level0Args a) {
    const int H = a.H, W = a.W;
    const int strips = (W + kL0Cols - 1) / kL0Cols, sblocks = (strips + 3) >> 2, segs = (H + kL0Seg - 1) / kL0Seg;
    const int per_image = sblocks * segs;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int images = a.n_arrays * a.batch;
    const int image = (q / per_image) * 8 + xcd, part = q - (q / per_image) * per_image;
    if (image >= images) return;
    const int pair = image / a.n_arrays, arr = image - pair * a.n_arrays;
    const int seg = part / sblocks, sb = part - seg * sblocks;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int strip = sb * 4 + wave;
    if (strip >= strips) return;                                  // wave-uniform
    const int x = strip * kL0Cols - 1 + lane;                     // this lane's column (before reflection)
    const bool own = lane >= 1 && lane <= kL0Cols && x < W;
    const unsigned xcol = (unsigned)skimage_reflect(min(x, W + 1), W);
    const double *s = a.src[arr] + (int64_t)pair * a.src_stride;
    double *d = a.dst[arr] + (int64_t)pair * a.dst_stride;
    ClipSlot *slot = a.slots ? a.slots + ((int64_t)pair * a.n_arrays + arr) * a.n_out + a.lvl : nullptr;
    ClipTrackExact tr;
    tr.init();
    // column terms: h = wa * own + wn * nb (see k_pyramid_stream: the blend of a SOURCE row, shared by the two output
    // rows that tap it)
    const double c = axis_pos(a.mx, x);
    const double fc = floor(c);
    const double dc = c - fc;
    const bool left = (int)fc < x, right = (int)ceil(c) > x;      // taps (x - 1, x) / (x, x + 1) / x itself
    const double wa = left ? dc : 1.0 - dc, wn = left ? 1.0 - dc : dc;
    // row terms of the segment: lane j computes those of row y0 + j
    const int y0 = seg * kL0Seg, y1 = min(y0 + kL0Seg, H);
    const double my_r = axis_pos(a.my, y0 + lane);
    const double my_fr = floor(my_r);
    const double my_dr = my_r - my_fr;
    const unsigned long long up_bits = __ballot((int)my_fr < y0 + lane);
    const unsigned long long down_bits = __ballot((int)ceil(my_r) > y0 + lane);
    // window: w[i] = row y - 1 + i for the group of kL0Ahead rows that starts at y
    double w[kL0Ahead + 2], nxt[kL0Ahead];
#pragma unroll
    for (int i = 0; i < kL0Ahead + 2; i++) w[i] = s[(int64_t)skimage_reflect(y0 - 1 + i, H) * W + xcol];
    auto hblend = [&](double v) {
        const double lv = from_left_lane(v), rv = from_right_lane(v);
        double nb = right ? rv : lv;
        if (!(left || right)) nb = v;
        return wa * v + wn * nb;
    };
    for (int y = y0; y < y1; y += kL0Ahead) {
        if (y + kL0Ahead < y1) {                                  // the next group's rows y + 9 .. y + 16
#pragma unroll
            for (int i = 0; i < kL0Ahead; i++)
                nxt[i] = s[(int64_t)skimage_reflect(y + kL0Ahead + 1 + i, H) * W + xcol];
        }
        double h_prev = hblend(w[0]), h_cur = hblend(w[1]);
#pragma unroll
        for (int j = 0; j < kL0Ahead; j++) {                      // (straight-line: rows beyond the segment are computed, not stored)
            const int oy = y + j;
            const int k = min(oy - y0, 63);
            const double h_next = hblend(w[j + 2]);
            const double dr = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(my_dr), k),
                                               __builtin_amdgcn_readlane(__double2loint(my_dr), k));
            const bool up = (up_bits >> k) & 1ull, down = (down_bits >> k) & 1ull;   // uniform
            const double top = up ? h_prev : h_cur;
            const double bot = down ? h_next : h_cur;
            const double v = (1.0 - dr) * top + dr * bot;
            if (own && oy < y1) {
                d[(int64_t)oy * W + x] = v;
                tr.tap(w[j + 1]); tr.out(v);
            }
            tr.nan_check(v, oy < y1);
            h_prev = h_cur; h_cur = h_next;
        }
#pragma unroll
        for (int i = 0; i < 2; i++) w[i] = w[kL0Ahead + i];
#pragma unroll
        for (int i = 0; i < kL0Ahead; i++) w[2 + i] = nxt[i];
    }
    if (slot) tr.flush(slot);
}

// The identity-scale level of the streaming kernel for one chunk: K output rows y .. y + K - 1 of this thread's column.
// The estimated map of rescale(., 1.0) is a o + b with a = 1 and |b| ~ 1e-13, so along an axis the sample position is
// o + b while b survives the rounding (taps: the pixel and the NEXT one for b > 0), then o itself (one tap), then -- from
// where a o rounds away from o -- the previous pixel and o: three runs per axis, and all but a few waves (columns) and
// chunks (rows) lie inside one.  CM / RMode name the run, wave-uniform: 1 = the pixel and the next one (right / down),
// 2 = the previous one and the pixel (left / up), 3 = the pixel itself, 0 = mixed (selected per lane / per row: the general
// form).  Whatever the mode, the operations on the taps are those of k_level0_rows: h = wa * own + wn * nb per SOURCE row
// (shared by the two output rows that tap it), v = (1 - dr) * top + dr * bot.
template <int CM, int RMode, int RM, int K>
__device__ __forceinline__ void stream_level0_chunk(const double (&w)[K + 2 * RM], const double wa, const double wn,
                                                    const bool c_left, const bool c_right, const double my_dr,
                                                    const unsigned up_bits, const unsigned down_bits, const int y,
                                                    const int oy_end, const bool own, double *__restrict__ dst, const int W,
                                                    const unsigned boff, ClipTrackExact &tr) {
    auto hblend = [&](int i) {                                    // i: row y - 1 + i of the window
        const double px = w[RM - 1 + i];
        double nb;
        if (CM == 1) nb = from_right_lane(px);
        else if (CM == 2) nb = from_left_lane(px);
        else if (CM == 3) nb = px;
        else {
            const double lv = from_left_lane(px), rv = from_right_lane(px);
            nb = c_right ? rv : lv;
            if (!(c_left || c_right)) nb = px;
        }
        return wa * px + wn * nb;
    };
    double h_prev = 0.0, h_cur = 0.0;
    if (RMode == 0 || RMode == 2) h_prev = hblend(0);
    if (RMode == 0 || RMode == 1) h_cur = hblend(1);
#pragma unroll
    for (int j = 0; j < K; j++) {                                 // (straight-line: rows beyond the segment are computed, not stored)
        const int oy = y + j;
        const double dr = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(my_dr), j),
                                           __builtin_amdgcn_readlane(__double2loint(my_dr), j));
        double top, bot;
        if (RMode == 1) {                                         // rows (y, y + 1)
            const double h_next = hblend(j + 2);
            top = h_cur; bot = h_next;
            h_cur = h_next;
        } else if (RMode == 2) {                                  // rows (y - 1, y)
            const double h = hblend(j + 1);
            top = h_prev; bot = h;
            h_prev = h;
        } else if (RMode == 3) {                                  // row y
            top = bot = hblend(j + 1);
        } else {
            const double h_next = hblend(j + 2);
            const bool up = (up_bits >> j) & 1u, down = (down_bits >> j) & 1u;   // uniform
            top = up ? h_prev : h_cur;
            bot = down ? h_next : h_cur;
            h_prev = h_cur; h_cur = h_next;
        }
        const double v = (1.0 - dr) * top + dr * bot;
        if (own && oy < oy_end) {
            // (a store the compiler does not count, see stream_load: issued iff the wave has a lane with an output in
            // this row -- the caller knows how many rows that is)
            stream_store(dst + (int64_t)oy * W, boff, v);
            tr.tap(w[RM + j]); tr.out(v);                         // (tracked whether or not there is a slot: 4 operations)
        }
        tr.nan_check(v, oy < oy_end);
        __builtin_amdgcn_sched_barrier(0);                        // one row at a time
    }
}

template <int R>
__device__ __forceinline__ double stream_vtap(const double *w, int c, const double (&wk)[R + 1]) {
    double tmp = w[c] * wk[R];
#pragma unroll
    for (int j = -R; j < 0; j++) tmp += (w[c + j] + w[c - j]) * wk[R + j];
    return tmp;
}

// one output per lane: horizontal Gaussian at the four taps from two ring rows, blend.  aa_tile_fixed's arithmetic.
template <int R>
__device__ __forceinline__ double stream_unit(const double *p0, const double *p1, double wx, const double wy,
                                              const double (&wck)[R + 1], double (&f)[2][2]) {
#pragma unroll
    for (int ry = 0; ry < 2; ry++) {
        const double *pr = ry ? p1 : p0;
        double u[2 * R + 2];                                      // one V row, columns x0 - R .. x0 + 1 + R
#pragma unroll
        for (int q = 0; q < 2 * R + 2; q++) u[q] = pr[q - R];
#pragma unroll
        for (int rx = 0; rx < 2; rx++) {
            double tmp = u[R + rx] * wck[R];
#pragma unroll
            for (int j = -R; j < 0; j++) tmp += (u[R + rx + j] + u[R + rx - j]) * wck[R + j];
            f[ry][rx] = tmp;
        }
        if (R >= 3) __builtin_amdgcn_sched_barrier(0);            // the two rows one after the other: 16 VGPRs less at R = 3
    }
    asm volatile("" : "+v"(wx));                                  // 1 - wx is computed here, not held per group across the chunk
    const double top = (1.0 - wx) * f[0][0] + wx * f[0][1];
    const double bot = (1.0 - wx) * f[1][0] + wx * f[1][1];
    return (1.0 - wy) * top + wy * bot;
}

// How the last 64-column group of a strip is emitted.  It holds `tail` = ncols - 64 (n_groups - 1) columns -- 15 of 143 at
// level 1, 31 of 95 at level 2 of a VGA strip -- so a unit of its own per output row runs 23 % / 48 % full.  With pack > 1 a
// unit takes the tail columns of `pack` consecutive output rows instead: lane = (row in the pack, tail column), the row terms
// per lane instead of per wave (~25 vector instructions more per unit, for 4 / 2 units less).
struct StreamTail {
    int pack, tail;     // rows per packed unit (1: not packed), columns of the tail group
    int rs;             // this lane's row within a pack (rs >= pack: idle); its tail column is lane - rs * tail
    __device__ __forceinline__ int ct(int lane) const { return lane - rs * tail; }
};
__device__ __forceinline__ StreamTail stream_tail(int ncols, int n_groups, int lane) {
    StreamTail t;
    t.tail = ncols - 64 * (n_groups - 1);
    t.pack = (n_groups >= 1 && t.tail >= 1 && t.tail <= 32) ? min(64 / t.tail, 4) : 1;
    const int tl = max(t.tail, 1);
    t.rs = lane / tl;
    return t;
}

// the horizontal pass + blend of one level for the output rows [oy_lo, oy_hi) of this chunk.  xoff / wxs: the lane's
// left tap (ring column) and blend weight in the FULL groups g < n_groups - 1; xo_t / wx_t: in the last group (its tail
// column when that group is packed) -- separate scalars, not a G-th array element: an element picked by the run-time
// n_groups - 1 sends the arrays to scratch.
template <int R, int G>
__device__ __forceinline__ int stream_emit(const AxisMap Lmy, const int LWo, const double *__restrict__ ring, int SW, double *dst,
                                           int oy_lo, int oy_end, int ynew, int ya, int n_groups,
                                           int ox_first,
                                           const int (&xoff)[G - 1], const double (&wxs)[G - 1], const int xo_t,
                                           const double wx_t, const double (&wck)[R + 1], const StreamTail tl,
                                           int wave, int lane, int &unit, ClipTrack &tr, ClipSlot *slot) {
    // the row terms of the next rows: lane i computes those of row oy_lo + i, the rows read them by lane;
    // the rows to emit now are those whose lower tap y0 + 1 is in the ring (a prefix: y0 is monotone)
    const double my_cy = axis_pos(Lmy, oy_lo + lane);
    const double my_fy = floor(my_cy);
    const double my_wy = my_cy - my_fy;
    const int my_y0 = (int)my_fy;
    const int n_rows = __builtin_popcountll(__ballot(oy_lo + lane < oy_end && my_y0 + 1 <= ynew));
    const int gt = n_groups - 1;                                  // the last group: `tail` columns
    // the ring slot of the upper tap, per lane with the row terms (one multiply-shift for 64 rows) instead of a scalar
    // modulo per row (s_mul_hi + 5, twice per row: a sixth of the kernel's scalar instructions)
    static_assert(kStreamRing == 10, "the multiply-shift below divides by 10");
    const unsigned my_n0 = (unsigned)max(my_y0 - ya, 0) & 0xffffu;     // (rows that are emitted have 0 <= y0 - ya < 2^16)
    const int my_s0 = (int)(my_n0 - ((my_n0 * 0xCCCDu) >> 19) * (unsigned)kStreamRing);
    for (int i = 0; i < n_rows; i++) {                            // wave-uniform
        const double wy = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(my_wy), i),
                                           __builtin_amdgcn_readlane(__double2loint(my_wy), i));
        const int slot0 = __builtin_amdgcn_readlane(my_s0, i);
        const int slot1 = slot0 + 1 == kStreamRing ? 0 : slot0 + 1;
        double *dst_row = dst + (int64_t)(oy_lo + i) * LWo + ox_first;   // uniform base, the lane is the offset
#pragma unroll
        for (int g = 0; g < G - 1; g++) {                         // full groups
            if (g >= gt) break;
            const bool mine = ((unit++) & 3) == wave;             // units dealt round-robin to the waves
            if (!mine) continue;
            double f[2][2];
            const double v = stream_unit<R>(ring + slot0 * SW + xoff[g], ring + slot1 * SW + xoff[g], wxs[g], wy, wck, f);
            (dst_row + g * 64)[lane] = v;
            if (slot) tr.add(slot, v, f[0][0], f[0][1], f[1][0], f[1][1]);
        }
        if (tl.pack == 1 && gt >= 0) {                            // the last group, a unit per row
            const bool mine = ((unit++) & 3) == wave;
            if (mine && lane < tl.tail) {
                double f[2][2];
                const double v = stream_unit<R>(ring + slot0 * SW + xo_t, ring + slot1 * SW + xo_t, wx_t, wy, wck, f);
                (dst_row + gt * 64)[lane] = v;
                if (slot) tr.add(slot, v, f[0][0], f[0][1], f[1][0], f[1][1]);
            }
        }
    }
    if (tl.pack > 1) {                                            // the last group, `pack` rows per unit
        for (int i0 = 0; i0 < n_rows; i0 += tl.pack) {            // wave-uniform
            const bool mine = ((unit++) & 3) == wave;
            if (!mine) continue;
            const int r = i0 + tl.rs;
            const bool active = tl.rs < tl.pack && r < n_rows;
            const int rc = min(r, n_rows - 1);                    // (idle lanes compute a valid row and do not store)
            // the row's terms per lane: the same operations on the same operands as my_cy above
            const double cy = axis_pos(Lmy, oy_lo + rc);
            const double fy = floor(cy);
            const double wy = cy - fy;
            const unsigned n0 = (unsigned)((int)fy - ya);         // < 2^16 (a segment's rows): n0 / 10 by multiplication
            const unsigned s0 = n0 - ((n0 * 0xCCCDu) >> 19) * (unsigned)kStreamRing;
            const unsigned s1 = s0 + 1u == (unsigned)kStreamRing ? 0u : s0 + 1u;
            double f[2][2];
            const double v = stream_unit<R>(ring + s0 * (unsigned)SW + xo_t, ring + s1 * (unsigned)SW + xo_t, wx_t, wy, wck, f);
            if (active) {
                dst[(int64_t)(oy_lo + rc) * LWo + ox_first + gt * 64 + tl.ct(lane)] = v;
                if (slot) tr.add(slot, v, f[0][0], f[0][1], f[1][0], f[1][1]);
            }
        }
    }
    return oy_lo + n_rows;
}

template <int RA, int RB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void k_pyramid_stream(StreamArgs a) {
    constexpr int RM = RA > RB ? RA : RB;
    constexpr int NL = RB > 0 ? 2 : 1;
    constexpr int K = kStreamK;
    static_assert(K == 8, "the level-0 row terms are computed by lane & 7");
    extern __shared__ __attribute__((aligned(16))) unsigned char aa_smem[];
    double *ringA = reinterpret_cast<double *>(aa_smem);                    // [Ring][SW]
    const int SW = a.pitch;
    double *ringB = ringA + (NL > 1 ? kStreamRing * SW : 0);

    // 1-D grid, XCD-major like k_rescale_aa_multi: XCD k takes images k, k + 8, ...; the strips of an
    // image are neighbours in dispatch order (their halo columns meet in one L2)
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int per_image = a.n_strips * a.n_segs;
    const int image = (q / per_image) * 8 + xcd, part = q - (q / per_image) * per_image;
    const int seg = part / a.n_strips, strip = part - seg * a.n_strips;
    if (image >= a.n_arrays * a.batch) return;
    const int pair = image / a.n_arrays, arr = image - pair * a.n_arrays;
    const int H = a.H, W = a.W;
    const double *s = a.src[arr] + (int64_t)pair * a.src_stride;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int xa = strip * a.strip_w, xb = min(xa + a.strip_w, W);
    const int ya = seg * a.seg_rows, yb = min(ya + a.seg_rows, H);   // V rows ya .. min(yb, H - 1) are needed
    // this thread's source column, as an index into the strip's support (owned columns + RM on the left, RM + 1 on the
    // right); threads beyond the support repeat its last column.  With the identity-scale level on board a wave advances
    // by 62 columns only: its lanes 0 and 63 hold the neighbour waves' border columns, so that a lane's left / right
    // neighbour column is always a lane of its own wave (DPP) -- no exchange through LDS, six columns of the vertical
    // pass computed twice (they write the same doubles to the same ring cells).
    const int ci = wave * a.wave_cols + lane;
    const unsigned xcol = (unsigned)mirror_idx(xa - RM + min(ci, xb - xa + 2 * RM), W);

    // per level: the strip's output columns (those whose left tap lies in [xa, xb)), per 64-column
    // group the lane's left tap as a ring column and its blend weight
    double wkA[RA + 1], wckA[RA + 1];
    double wkB[RB + 1], wckB[RB + 1];
    int oxA0, ncolsA, ngA, xoffA[kStreamGroupsA - 1], xoA_t;
    double wxA[kStreamGroupsA - 1], wxA_t;
    int oxB0 = 0, ncolsB = 0, ngB = 0, xoffB[kStreamGroupsB - 1], xoB_t = 0;
    double wxB[kStreamGroupsB - 1], wxB_t = 0.0;
    StreamTail tlA = stream_tail(64, 1, lane), tlB = stream_tail(64, 1, lane);
    {
        const StreamLevel &L = a.lv[0];
#pragma unroll
        for (int k = 0; k <= RA; k++) { wkA[k] = L.wr[k]; wckA[k] = L.wc[k]; }
        oxA0 = a.strip_tab[strip].a0;
        ncolsA = a.strip_tab[strip].a1 - oxA0;
        ngA = (ncolsA + 63) >> 6;
        tlA = stream_tail(ncolsA, ngA, lane);
#pragma unroll
        for (int g = 0; g < kStreamGroupsA - 1; g++) {
            const double cx = axis_pos(L.mx, oxA0 + g * 64 + lane);
            const double fx0 = floor(cx);
            wxA[g] = cx - fx0;
            xoffA[g] = min(max((int)fx0 - (xa - RM), RA), SW - RA - 2);   // clamp: lanes beyond ncols
        }
        {
            const double cx = axis_pos(L.mx, oxA0 + max(ngA - 1, 0) * 64 + (tlA.pack > 1 ? tlA.ct(lane) : lane));
            const double fx0 = floor(cx);
            wxA_t = cx - fx0;
            xoA_t = min(max((int)fx0 - (xa - RM), RA), SW - RA - 2);
        }
    }
    if constexpr (NL > 1) {
        const StreamLevel &L = a.lv[1];
#pragma unroll
        for (int k = 0; k <= RB; k++) { wkB[k] = L.wr[k]; wckB[k] = L.wc[k]; }
        oxB0 = a.strip_tab[strip].b0;
        ncolsB = a.strip_tab[strip].b1 - oxB0;
        ngB = (ncolsB + 63) >> 6;
        tlB = stream_tail(ncolsB, ngB, lane);
#pragma unroll
        for (int g = 0; g < kStreamGroupsB - 1; g++) {
            const double cx = axis_pos(L.mx, oxB0 + g * 64 + lane);
            const double fx0 = floor(cx);
            wxB[g] = cx - fx0;
            xoffB[g] = min(max((int)fx0 - (xa - RM), RB), SW - RB - 2);
        }
        {
            const double cx = axis_pos(L.mx, oxB0 + max(ngB - 1, 0) * 64 + (tlB.pack > 1 ? tlB.ct(lane) : lane));
            const double fx0 = floor(cx);
            wxB_t = cx - fx0;
            xoB_t = min(max((int)fx0 - (xa - RM), RB), SW - RB - 2);
        }
    }
    double *dstA = a.lv[0].dst[arr] + (int64_t)pair * a.lv[0].dst_stride;
    double *dstB = NL > 1 ? a.lv[1].dst[arr] + (int64_t)pair * a.lv[1].dst_stride : nullptr;
    ClipTrack trA, trB;
    trA.init();
    trB.init();
    ClipSlot *slotA = a.slots ? a.slots + ((int64_t)pair * a.n_arrays + arr) * a.n_out + a.lv[0].lvl : nullptr;
    ClipSlot *slotB = a.slots && NL > 1 ? a.slots + ((int64_t)pair * a.n_arrays + arr) * a.n_out + a.lv[1].lvl : nullptr;
    // level 0: thread = output column.  Its taps are its own column and ONE neighbour column (the map is within a
    // pixel of the identity -- checked on the host), rows y - 1 .. y + 1 of the register window.
    const bool do_l0 = ((a.l0_mask >> arr) & 1u) != 0;            // block-uniform
    double *dst0 = nullptr;
    ClipSlot *slot0 = nullptr;
    ClipTrackExact tr0;
    tr0.init();
    if (do_l0) {
        dst0 = a.l0_dst[arr] + (int64_t)pair * a.l0_stride;
        slot0 = a.slots ? a.slots + ((int64_t)pair * a.n_arrays + arr) * a.n_out + a.l0_lvl : nullptr;
    }

    // level 0's column terms (held across the chunks: five registers).  The horizontal blend of a SOURCE row is
    // h = wa * own + wn * nb with (wa, wn, nb) = (1 - dc, dc, right) / (dc, 1 - dc, left) / (1 - dc, dc, own) for taps (x, x + 1) /
    // (x - 1, x) / x alone; cm: which of the three runs ALL of this wave's own columns lie in (0: mixed), see stream_level0_chunk;
    // left and right exclude each other (floor(c) < x < ceil(c) has no integer x)
    double l0_wa = 0.0, l0_wn = 0.0;
    int l0_flags = 0, cm = 0;                                     // bit 0 left, 1 right, 2 this lane owns an output column
    unsigned l0_boff = 0u;
    bool l0_any_own = false;
    if (do_l0) {
        const int l0_x = xa - RM + ci;                            // lanes 0 / 63 and the halo threads hold neighbour / reflected columns
        const bool own = l0_x >= xa && l0_x < xb && lane >= 1 && lane <= 62;
        const double l0_c = axis_pos(a.l0_mx, l0_x);
        const double l0_fc = floor(l0_c);
        const double l0_dc = l0_c - l0_fc;
        const bool left = (int)l0_fc < l0_x;                      // taps (x - 1, x)
        const bool right = (int)ceil(l0_c) > l0_x;                // taps (x, x + 1); neither: the position is the pixel itself
        l0_wa = left ? l0_dc : 1.0 - l0_dc;
        l0_wn = left ? 1.0 - l0_dc : l0_dc;
        l0_flags = (left ? 1 : 0) | (right ? 2 : 0) | (own ? 4 : 0);
        const unsigned long long own_m = __ballot(own), right_m = __ballot(right), left_m = __ballot(left);
        cm = (own_m & ~right_m) == 0ull ? 1 : (own_m & ~left_m) == 0ull ? 2 : (own_m & (left_m | right_m)) == 0ull ? 3 : 0;
        l0_any_own = own_m != 0ull;
        l0_boff = (unsigned)max(l0_x, 0) * 8u;
    }

    // the column's window: w[i] = source row (y - RM + i) for the chunk that starts at V row y
    double w[K + 2 * RM], nxt[K];
#pragma unroll
    for (int i = 0; i < 2 * RM; i++) w[i] = s[(int64_t)mirror_idx(ya + i - RM, H) * W + xcol];
#pragma unroll
    for (int i = 0; i < K; i++) w[2 * RM + i] = s[(int64_t)mirror_idx(ya + RM + i, H) * W + xcol];
    int nextA = a.seg_tab[seg].a0, nextB = NL > 1 ? a.seg_tab[seg].b0 : 0, unit = 0;
    const int endA = a.seg_tab[seg].a1;
    const int endB = NL > 1 ? a.seg_tab[seg].b1 : 0;
    const int y_last = min(yb, H - 1);                            // last V row this block needs
    const int n_chunks = (y_last - ya + K) / K;
    // Arguments used once per chunk (the levels' row maps, the identity level's maps) are read from the kernel-argument
    // segment when they are needed -- scalar loads -- instead of living in SGPRs across the loop: the kernel had ~50
    // SGPRs spilled into VGPR lanes and ~100 v_readlane per chunk to get them back (12 % of its vector instructions).
    // The pointer is laundered per chunk so that the loads stay inside the loop.
    typedef const __attribute__((address_space(4))) StreamArgs *KArgs;
    KArgs ka = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    // The window's first rows must have arrived before the loop is entered: left pending, the compiler guards their first
    // use INSIDE the loop with s_waitcnt vmcnt(7 .. 0) behind the prefetch of the next chunk -- counts that are right for
    // the first pass and in every later one wait for the loads that were issued a moment ago (the prefetch distance gone).
    __builtin_amdgcn_s_waitcnt(0x0F70);                          // vmcnt(0)
    int slot_c = 0;                                               // (c K) % ring, carried instead of divided
    static_assert(kStreamK < kStreamRing, "");
    for (int c = 0; c < n_chunks; c++) {
        asm volatile("" : "+s"(ka));
        const int y = ya + c * K;                                 // first V row of this chunk
        if (c + 1 < n_chunks) {                                   // prefetch the next chunk's K rows
            const int r0 = y + K + RM;
            if (r0 + K - 1 <= H - 1) {                            // inside the image: uniform row bases, the column is the offset
#pragma unroll
                for (int i = 0; i < K; i++) nxt[i] = stream_load(s + (int64_t)(r0 + i) * W, xcol * 8u);
            } else {
#pragma unroll
                for (int i = 0; i < K; i++) nxt[i] = stream_load(s + (int64_t)mirror_idx(r0 + i, H) * W, xcol * 8u);
            }
        }
        int l0_stores_8 = 0;                                      // the chunk's last 8 vector memory operations are level-0 stores
        // vertical Gaussians of both levels at V rows y .. y + K - 1 (rows >= H: computed, never read)
#pragma unroll
        for (int j = 0; j < K; j++) {
            const int slot = slot_c + j >= kStreamRing ? slot_c + j - kStreamRing : slot_c + j;   // (c K + j) % ring
            const double va = stream_vtap<RA>(w, j + RM, wkA);
            double vb = 0.0;
            if constexpr (NL > 1) vb = stream_vtap<RB>(w, j + RM, wkB);
            if (ci < SW) {                                        // threads beyond the pitch hold a repeated column
                ringA[slot * SW + ci] = va;
                if constexpr (NL > 1) ringB[slot * SW + ci] = vb;
            }
        }
        __syncthreads();
        // outputs whose lower row tap y0 + 1 is now in the ring: y0 + 1 <= y + K - 1
        const int ynew = y + K - 1 >= y_last ? (1 << 30) : y + K - 1;   // the last chunk emits whatever is left
        {
            AxisMap my;
            my.a = ka->lv[0].my.a; my.b = ka->lv[0].my.b; my.s = ka->lv[0].my.s; my.ideal = ka->lv[0].my.ideal;
            nextA = stream_emit<RA, kStreamGroupsA>(my, ka->lv[0].Wo, ringA, SW, dstA, nextA, endA, ynew, ya, ngA, oxA0,
                                                    xoffA, wxA, xoA_t, wxA_t, wckA, tlA, wave, lane, unit, trA, slotA);
        }
        if constexpr (NL > 1) {
            AxisMap my;
            my.a = ka->lv[1].my.a; my.b = ka->lv[1].my.b; my.s = ka->lv[1].my.s; my.ideal = ka->lv[1].my.ideal;
            nextB = stream_emit<RB, kStreamGroupsB>(my, ka->lv[1].Wo, ringB, SW, dstB, nextB, endB, ynew, ya, ngB, oxB0,
                                                    xoffB, wxB, xoB_t, wxB_t, wckB, tlB, wave, lane, unit, trB, slotB);
        }
        if (do_l0) {
            const int oy_end0 = min(yb, H);                       // this block's level-0 rows: those of its segment
            AxisMap l0my;
            l0my.a = ka->l0_my.a; l0my.b = ka->l0_my.b; l0my.s = ka->l0_my.s; l0my.ideal = ka->l0_my.ideal;
            // the row terms: lane j computes those of row y + j, the rows read them by lane (-0.04 ms of 1.73)
            const double my_r = axis_pos(l0my, y + (lane & 7));
            const double my_fr = floor(my_r);
            const double my_dr = my_r - my_fr;
            const unsigned up_bits = (unsigned)__ballot((int)my_fr < y + (lane & 7)) & 0xffu;
            const unsigned down_bits = (unsigned)__ballot((int)ceil(my_r) > y + (lane & 7)) & 0xffu;
            // The horizontal blend of a SOURCE row, h(i) = (1 - dc) f0 + dc f1 with (f0, f1) = (neighbour, own) /
            // (own, neighbour) / (own, own), is the same double for both output rows that tap the row, so it is
            // computed once per source row and an output row is one vertical blend of two of them.  The products
            // commute, so one form serves all three cases: h = wa * own + wn * nb with the weights and nb picked per column.
            const bool l0_left = (l0_flags & 1) != 0, l0_right = (l0_flags & 2) != 0, l0_own = (l0_flags & 4) != 0;
            const int rm = down_bits == 0xffu ? 1 : up_bits == 0xffu ? 2 : (up_bits | down_bits) == 0u ? 3 : 0;
            l0_stores_8 = l0_any_own && y + K <= oy_end0 ? 1 : 0;   // a store per row, each issued (some lane owns an output)
#define TDK_L0_CASE(CM_, RM_)                                                                                              \
    case CM_ * 4 + RM_:                                                                                                    \
        stream_level0_chunk<CM_, RM_, RM, K>(w, l0_wa, l0_wn, l0_left, l0_right, my_dr, up_bits, down_bits, y, oy_end0,   \
                                             l0_own, dst0, W, l0_boff, tr0);                                               \
        break;
            switch (cm && rm ? cm * 4 + rm : 0) {                 // wave-uniform
                TDK_L0_CASE(1, 1) TDK_L0_CASE(1, 2) TDK_L0_CASE(1, 3)
                TDK_L0_CASE(2, 1) TDK_L0_CASE(2, 2) TDK_L0_CASE(2, 3)
                TDK_L0_CASE(3, 1) TDK_L0_CASE(3, 2) TDK_L0_CASE(3, 3)
                default:
                    stream_level0_chunk<0, 0, RM, K>(w, l0_wa, l0_wn, l0_left, l0_right, my_dr, up_bits, down_bits, y, oy_end0,
                                                     l0_own, dst0, W, l0_boff, tr0);
            }
#undef TDK_L0_CASE
        }
        if (kStreamRing < 2 * K + 1) __syncthreads();             // the next chunk's V rows overwrite rows read above
        slot_c = slot_c + K >= kStreamRing ? slot_c + K - kStreamRing : slot_c + K;
        if (c + 1 < n_chunks) {                                   // the prefetched rows (stream_load)
            stream_rows_arrived(nxt, __builtin_amdgcn_readfirstlane(l0_stores_8));
        }
#pragma unroll
        for (int i = 0; i < 2 * RM; i++) w[i] = w[K + i];
#pragma unroll
        for (int i = 0; i < K; i++) w[2 * RM + i] = nxt[i];
    }
    if (slot0) tr0.flush(slot0);
    if (slotA) trA.flush(slotA);
    if constexpr (NL > 1)
        if (slotB) trB.flush(slotB);
}

// ---------------------------------------------------------------------------
// clip: true bounds of a flagged (image, level) and numpy.clip
// ---------------------------------------------------------------------------
struct ClipArgs {
    const double *src[4];
    int64_t src_stride;
    int H, W, n_arrays, batch, n_out;
    DevLevel lv[kMaxOut];
    ClipSlot *slots;
};

constexpr int kClipTileRows = 16, kClipTileCols = 64;

// one thread per slot: the flagged ones go on a list (almost always empty)
__global__ void k_clip_gate(ClipSlot *slots, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !clip_needed(slots[i])) return;
    int *list = clip_list(slots, n);
    list[1 + atomicAdd(&list[0], 1)] = i;
}

// A fixed grid walks (flagged slot) x (tile of the SOURCE image); nothing flagged: every block returns after one load.
// A tile's filtered values: vertical pass into LDS (columns + 2 Rc of halo), horizontal pass from LDS -- scipy's
// order, as everywhere in this file.
__global__ __launch_bounds__(256) void k_clip_bounds(ClipArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char aa_smem[];
    const int n_slots = a.n_arrays * a.batch * a.n_out;
    const int *list = clip_list(a.slots, n_slots);
    const int n_flagged = list[0];
    if (n_flagged == 0) return;
    const int H = a.H, W = a.W;
    const int tiles_x = (W + kClipTileCols - 1) / kClipTileCols, tiles_y = (H + kClipTileRows - 1) / kClipTileRows;
    const int per_slot = tiles_x * tiles_y;
    double *V = reinterpret_cast<double *>(aa_smem);
    for (int64_t item = blockIdx.x; item < (int64_t)n_flagged * per_slot; item += gridDim.x) {
        const int k = (int)(item / per_slot), t = (int)(item - (int64_t)k * per_slot);
        const int si = list[1 + k];
        const int image = si / a.n_out, l = si - image * a.n_out;
        ClipSlot *slot = a.slots + si;
        const int pair = image / a.n_arrays, arr = image - pair * a.n_arrays;
        const DevLevel &L = a.lv[l];
        const int Rr = L.aa.Rr, Rc = L.aa.Rc;
        const double *s = a.src[arr] + (int64_t)pair * a.src_stride;
        const int SC = kClipTileCols + 2 * Rc;
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        const int y0 = ty * kClipTileRows, x0 = tx * kClipTileCols;
        const int nr = min(kClipTileRows, H - y0), nc = min(kClipTileCols, W - x0);
        double vmax = -INFINITY, vmin = INFINITY;
        bool nan = false;
        __syncthreads();
        for (int i = threadIdx.x; i < nr * (nc + 2 * Rc); i += 256) {
            const int r = i / (nc + 2 * Rc), c = i - r * (nc + 2 * Rc);
            V[r * SC + c] = column_tap(s, H, W, y0 + r, mirror_idx(x0 - Rc + c, W), L.aa.wr, Rr);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nr * nc; i += 256) {
            const int r = i / nc, c = i - r * nc;
            const double *row = V + r * SC + c + Rc;
            double tmp;
            if (Rc == 0) tmp = row[0];
            else {
                tmp = row[0] * L.aa.wc[Rc];
                for (int j = -Rc; j < 0; j++) tmp += (row[j] + row[-j]) * L.aa.wc[Rc + j];
            }
            vmax = fmax(vmax, tmp);
            vmin = fmin(vmin, tmp);
            nan |= tmp != tmp;
        }
        vmax = wave_max(vmax);
        vmin = wave_min(vmin);
        const bool any_nan = __ballot(nan) != 0ull;
        if ((threadIdx.x & 63) == 0) {
            if (vmax != -INFINITY || vmin != INFINITY) {
                atomicMax(&slot->hi, enc(vmax));
                atomicMin(&slot->lo, enc(vmin));
            }
            if (any_nan) atomicOr(&slot->nan_img, 1u);
        }
    }
}

constexpr int kClipApplyChunk = 4096;

__global__ __launch_bounds__(256) void k_clip_apply(ClipArgs a) {
    const int n_slots = a.n_arrays * a.batch * a.n_out;
    const int *list = clip_list(a.slots, n_slots);
    const int n_flagged = list[0];
    if (n_flagged == 0) return;
    int64_t n_max = 0;
    for (int l = 0; l < a.n_out; l++) n_max = max(n_max, (int64_t)a.lv[l].Ho * a.lv[l].Wo);
    const int per_slot = (int)((n_max + kClipApplyChunk - 1) / kClipApplyChunk);
    for (int64_t item = blockIdx.x; item < (int64_t)n_flagged * per_slot; item += gridDim.x) {
        const int k = (int)(item / per_slot), chunk = (int)(item - (int64_t)k * per_slot);
        const int si = list[1 + k];
        const int image = si / a.n_out, l = si - image * a.n_out;
        const ClipSlot slot = a.slots[si];
        const int pair = image / a.n_arrays, arr = image - pair * a.n_arrays;
        const DevLevel &L = a.lv[l];
        double *d = L.dst[arr] + (int64_t)pair * L.stride;
        const int64_t n = (int64_t)L.Ho * L.Wo;
        // ndarray.min() / .max() of an image with a NaN are NaN, and numpy.clip with a NaN bound returns NaN
        const double lo = slot.nan_img ? NAN : dec(slot.lo), hi = slot.nan_img ? NAN : dec(slot.hi);
        const int64_t i0 = (int64_t)chunk * kClipApplyChunk, i1 = min(i0 + kClipApplyChunk, n);
        for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
            const double x = d[i];
            const double m = (x != x || lo != lo) ? NAN : (x > lo ? x : lo);
            const double v = (m != m || hi != hi) ? NAN : (m < hi ? m : hi);
            if (__double_as_longlong(v) != __double_as_longlong(x)) d[i] = v;
        }
    }
}

// A level whose taps skip elements of the filtered image (a shrinking map with no prefilter, or with a kernel
// narrower than the gaps between its taps -- never skimage's own sigma rule, but the plan is the caller's): a NaN
// that no tap sees still makes numpy.clip's bounds NaN, i.e. every output.  One pass over the source flags the
// slots of those levels (found by tests/test_gpu_fuzz.py: rescale(anti_aliasing=False) of an image with one NaN).
__global__ __launch_bounds__(256) void k_clip_nan_scan(ClipArgs a, unsigned level_mask, int per_image) {
    const int image = blockIdx.x / per_image, part = blockIdx.x - image * per_image;   // 1-D grid: no limit on the batch
    const int pair = image / a.n_arrays, arr = image - pair * a.n_arrays;
    const double *s = a.src[arr] + (int64_t)pair * a.src_stride;
    const int64_t n = (int64_t)a.H * a.W;
    bool nan = false;
    for (int64_t i = (int64_t)part * 256 + threadIdx.x; i < n; i += (int64_t)per_image * 256) nan |= s[i] != s[i];
    if (__ballot(nan) == 0ull || (threadIdx.x & 63) != 0) return;
    for (int l = 0; l < a.n_out; l++)
        if ((level_mask >> l) & 1u) atomicOr(&a.slots[((int64_t)pair * a.n_arrays + arr) * a.n_out + l].nan_out, 1u);
}

// Few slots (a single pair: 3 images x 3 levels): gate, bounds and clip in ONE launch, a block per slot -- a flagged
// slot's block walks the whole source image by itself (rare, and then a few hundred microseconds); the three
// launches above cost a single-pair call 10 us more than this one.
__global__ __launch_bounds__(256) void k_clip_small(ClipArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char aa_smem[];
    __shared__ double red_max[4], red_min[4];
    __shared__ int red_nan[4];
    const int si = blockIdx.x;
    ClipSlot *slot = a.slots + si;
    if (!clip_needed(*slot)) {                                     // block-uniform
        if (threadIdx.x == 0) clip_slot_clear(slot);               // left clean for the next build (no reset launch)
        return;
    }
    const int image = si / a.n_out, l = si - image * a.n_out;
    const int pair = image / a.n_arrays, arr = image - pair * a.n_arrays;
    const DevLevel &L = a.lv[l];
    const int H = a.H, W = a.W, Rr = L.aa.Rr, Rc = L.aa.Rc;
    const double *s = a.src[arr] + (int64_t)pair * a.src_stride;
    const int tiles_x = (W + kClipTileCols - 1) / kClipTileCols, tiles_y = (H + kClipTileRows - 1) / kClipTileRows;
    const int SC = kClipTileCols + 2 * Rc;
    double *V = reinterpret_cast<double *>(aa_smem);
    double vmax = -INFINITY, vmin = INFINITY;
    bool nan = false;
    for (int t = 0; t < tiles_x * tiles_y; t++) {
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        const int y0 = ty * kClipTileRows, x0 = tx * kClipTileCols;
        const int nr = min(kClipTileRows, H - y0), nc = min(kClipTileCols, W - x0);
        __syncthreads();
        for (int i = threadIdx.x; i < nr * (nc + 2 * Rc); i += 256) {
            const int r = i / (nc + 2 * Rc), c = i - r * (nc + 2 * Rc);
            V[r * SC + c] = column_tap(s, H, W, y0 + r, mirror_idx(x0 - Rc + c, W), L.aa.wr, Rr);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nr * nc; i += 256) {
            const int r = i / nc, c = i - r * nc;
            const double *row = V + r * SC + c + Rc;
            double tmp;
            if (Rc == 0) tmp = row[0];
            else {
                tmp = row[0] * L.aa.wc[Rc];
                for (int j = -Rc; j < 0; j++) tmp += (row[j] + row[-j]) * L.aa.wc[Rc + j];
            }
            vmax = fmax(vmax, tmp);
            vmin = fmin(vmin, tmp);
            nan |= tmp != tmp;
        }
    }
    vmax = wave_max(vmax);
    vmin = wave_min(vmin);
    const bool any_nan = __ballot(nan) != 0ull;
    if ((threadIdx.x & 63) == 0) {
        red_max[threadIdx.x >> 6] = vmax; red_min[threadIdx.x >> 6] = vmin; red_nan[threadIdx.x >> 6] = any_nan;
    }
    __syncthreads();
    const bool img_nan = red_nan[0] | red_nan[1] | red_nan[2] | red_nan[3];
    // ndarray.min() / .max() of an image with a NaN are NaN, and numpy.clip with a NaN bound returns NaN
    const double hi = img_nan ? NAN : fmax(fmax(red_max[0], red_max[1]), fmax(red_max[2], red_max[3]));
    const double lo = img_nan ? NAN : fmin(fmin(red_min[0], red_min[1]), fmin(red_min[2], red_min[3]));
    double *d = L.dst[arr] + (int64_t)pair * L.stride;
    const int64_t n = (int64_t)L.Ho * L.Wo;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const double x = d[i];
        const double m = (x != x || lo != lo) ? NAN : (x > lo ? x : lo);
        const double v = (m != m || hi != hi) ? NAN : (m < hi ? m : hi);
        if (__double_as_longlong(v) != __double_as_longlong(x)) d[i] = v;
    }
    if (threadIdx.x == 0) clip_slot_clear(slot);
}

// scipy.ndimage._filters._gaussian_kernel1d (order 0) with libm's exp and a sequential sum: the kernels of a
// level without a plan (the ideal reading; a plan brings the caller's NumPy kernels)
void gaussian_weights(double sigma, int radius, double *w) {
    const double sigma2 = sigma * sigma;
    double sum = 0.0;
    for (int i = -radius; i <= radius; i++) {
        w[i + radius] = exp(-0.5 / sigma2 * (double)(i * i));
        sum += w[i + radius];
    }
    for (int i = 0; i <= 2 * radius; i++) w[i] = w[i] / sum;
}

// all four taps of every output inside the image, so that x1 = x0 + 1 / y1 = y0 + 1 hold (tiles, stream)
bool taps_inside(const AxisMap &m, int n_in, int n_out) {
    if (n_out >= n_in) return false;
    const double p0 = axis_pos(m, 0), p1 = axis_pos(m, n_out - 1);
    if (!(p0 >= 0.0) || !(p1 > p0)) return false;
    if ((int)floor(p1) + 1 > n_in - 1) return false;
    // skimage's second tap is ceil(p): at a sample position that is an integer it is the FIRST tap again, and the
    // pixel next to it is never read.  The tiles and the streaming kernel read it with weight 0 -- the same double for
    // finite images, NaN for an Inf / NaN neighbour (found by tests/test_gpu_fuzz.py on an 11 x 71 image whose
    // estimated row map 11/7 o + 2/7 hits 5.0).  A level with such a position goes to the general kernel, which taps
    // floor / ceil.  The ideal reading is the same warp at other positions (oracle: warp_bilinear), and hits integers
    // whenever in / out reduces to odd / odd (119 -> 79: (o + 0.5) 119 / 79 - 0.5 = 59 at o = 39).
    for (int o = 0; o < n_out; o++) {
        const double p = axis_pos(m, o);
        if (p == floor(p)) return false;
    }
    return true;
}

}  // namespace

namespace tdk {

AxisMap ideal_axis(int n_in, int n_out) {
    AxisMap m;
    m.a = 0.0; m.b = 0.0; m.s = (double)n_in / (double)n_out; m.ideal = 1;
    return m;
}

AxisMap affine_axis(double a, double b) {
    AxisMap m;
    m.a = a; m.b = b; m.s = 0.0; m.ideal = 0;
    return m;
}

void ideal_level_plan(PyramidLevelDesc *lv, int H, int W, bool anti_aliasing, double *storage) {
    lv->mx = ideal_axis(W, lv->W);
    lv->my = ideal_axis(H, lv->H);
    lv->wr = lv->wc = nullptr;
    lv->Rr = lv->Rc = 0;
    if (!anti_aliasing) return;
    // sigma = max(0, (factor - 1) / 2) per axis; an axis with sigma <= 1e-15 is not filtered
    const double sg[2] = {((double)H / (double)lv->H - 1.0) / 2.0, ((double)W / (double)lv->W - 1.0) / 2.0};
    for (int ax = 0; ax < 2; ax++) {
        if (!(sg[ax] > 1e-15)) continue;
        const int R = (int)(4.0 * sg[ax] + 0.5);
        if (R > kMaxGaussRadius) {      // reported by launch_pyramid
            (ax ? lv->Rc : lv->Rr) = R;
            continue;
        }
        double *w = storage + (size_t)ax * (2 * kMaxGaussRadius + 1);
        gaussian_weights(sg[ax], R, w);
        if (ax == 0) { lv->wr = w; lv->Rr = R; }
        else { lv->wc = w; lv->Rc = R; }
    }
}

size_t pyramid_weight_doubles(int n_out) { return (size_t)n_out * 2 * (2 * kMaxGaussRadius + 1); }
size_t pyramid_clip_bytes(int64_t n_images, int n_out) {
    const size_t n = (size_t)n_images * (size_t)n_out;
    return sizeof(ClipSlot) * n + sizeof(int) * (n + 1);      // slots, counter, list of flagged slots
}
int pyramid_max_radius() { return kMaxGaussRadius; }

// Every level in `levels` (n_out of them; a level-0 entry has H x W = the source's) of `n_arrays` arrays of
// `batch` images.  `weights` is a device buffer of pyramid_weight_doubles(n_out) doubles owned by the caller;
// the kernels of every level are copied into it when upload_weights is set (they depend on shapes and plans
// only).  `clip_slots`: a device buffer of pyramid_clip_bytes(batch * n_arrays, n_out) bytes, or null for
// clip=False.  stream_mode: 0 never the streaming kernel, 1 for batches that fill the chip, 2 always.
tdk_status launch_pyramid(const double *const *srcs, int n_arrays, int H, int W, int64_t src_stride, int n_out,
                          const PyramidLevelDesc *levels, int batch, double *weights, bool upload_weights,
                          void *clip_slots, int stream_mode, hipStream_t stream, int *slots_clean) {
    if (n_out <= 0) return TDK_OK;
    if (n_out > kMaxOut || n_arrays > 4) {
        set_error("pyramid too deep");
        return TDK_ERR_INVALID_ARGUMENT;
    }
    constexpr int kSlot = 2 * kMaxGaussRadius + 1;
    std::vector<double> host((size_t)n_out * 2 * kSlot, 0.0);
    DevLevel dv[kMaxOut];
    bool done[kMaxOut] = {};
    for (int l = 0; l < n_out; l++) {
        const PyramidLevelDesc &P = levels[l];
        if (P.Rr > kMaxGaussRadius || P.Rc > kMaxGaussRadius || P.Rr < 0 || P.Rc < 0) {
            set_error("anti-aliasing kernel radius %d exceeds %d", std::max(P.Rr, P.Rc), kMaxGaussRadius);
            return TDK_ERR_INVALID_ARGUMENT;
        }
        for (int i = 0; i < 4; i++) dv[l].dst[i] = i < n_arrays ? P.dst[i] : nullptr;
        dv[l].stride = P.stride; dv[l].Ho = P.H; dv[l].Wo = P.W;
        dv[l].mx = P.mx; dv[l].my = P.my;
        double *wr = host.data() + ((size_t)l * 2 + 0) * kSlot, *wc = host.data() + ((size_t)l * 2 + 1) * kSlot;
        if (P.wr && P.Rr > 0) memcpy(wr, P.wr, sizeof(double) * (2 * P.Rr + 1)); else wr[0] = 1.0;
        if (P.wc && P.Rc > 0) memcpy(wc, P.wc, sizeof(double) * (2 * P.Rc + 1)); else wc[0] = 1.0;
        dv[l].aa.wr = weights + ((size_t)l * 2 + 0) * kSlot;
        dv[l].aa.wc = weights + ((size_t)l * 2 + 1) * kSlot;
        dv[l].aa.Rr = (P.wr && P.Rr > 0) ? P.Rr : 0;
        dv[l].aa.Rc = (P.wc && P.Rc > 0) ? P.Rc : 0;
    }
    if (upload_weights) {
        TDK_HIP(hipMemcpyAsync(weights, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice, stream));
        TDK_HIP(hipStreamSynchronize(stream));   // `host` goes out of scope
    }
    ClipSlot *slots = static_cast<ClipSlot *>(clip_slots);
    const int64_t images = (int64_t)n_arrays * batch;
    // (a build with few slots leaves them clean -- k_clip_small -- so the next one of the same owner needs no reset,
    // PROVIDED it uses the same number of slots: *slots_clean is that number.  A build of more arrays or levels than
    // the last one would otherwise read slots nobody ever initialised)
    if (slots && !(slots_clean && *slots_clean == (int)(images * n_out))) {
        const int n = (int)(images * n_out);
        k_clip_reset<<<(n + 255) / 256, 256, 0, stream>>>(slots, n);
        TDK_LAUNCH_CHECK();
    }
    if (slots_clean) *slots_clean = 0;
    auto shrinks = [&](int l) {
        return taps_inside(dv[l].mx, W, dv[l].Wo) && taps_inside(dv[l].my, H, dv[l].Ho);
    };
    // --- the first level with radius 1 (and the one with radius 3 if there is one) of a batch large enough to
    // fill the chip with full-height strips: one streaming pass over the source
    {
        int lA = -1, lB = -1;
        for (int l = 0; l < n_out && lA < 0; l++)
            if (!done[l] && dv[l].aa.Rr == 1 && dv[l].aa.Rc == 1 && shrinks(l)) lA = l;
        for (int l = 0; l < n_out && lA >= 0 && lB < 0; l++)
            if (!done[l] && dv[l].aa.Rr == 3 && dv[l].aa.Rc == 3 && shrinks(l)) lB = l;
        // the identity-scale level rides along if its map stays within a pixel of the identity (every output's
        // taps are its own column and one neighbour, rows y - 1 .. y + 1)
        int l0 = -1;
        for (int l = 0; l < n_out && l0 < 0; l++) {
            if (done[l] || dv[l].Ho != H || dv[l].Wo != W || dv[l].aa.Rr || dv[l].aa.Rc) continue;
            auto near_identity = [](const AxisMap &m, int n) {
                return fabs(axis_pos(m, 0)) < 0.5 && fabs(axis_pos(m, n - 1) - (double)(n - 1)) < 0.5;
            };
            if (near_identity(dv[l].mx, W) && near_identity(dv[l].my, H)) l0 = l;
        }
        if (stream_mode == 3) l0 = -1;                            // (experiment: level 0 by k_level0_stream)
        // a strip's support (owned columns + 2 RM + 1) must fit the columns the block's four waves hold
        const int wave_cols = l0 >= 0 ? 62 : 64;
        const int max_strip = 3 * wave_cols + 64 - 7 - 1;        // 248 / 242
        const int n_strips = (W + max_strip - 1) / max_strip, strip_w = (W + n_strips - 1) / n_strips;
        if (stream_mode && lA >= 0 && (images * n_strips >= 256 || stream_mode == 2)) {
            const int nl = lB >= 0 ? 2 : 1;
            StreamArgs sa;
            for (int i = 0; i < 4; i++) sa.src[i] = i < n_arrays ? srcs[i] : nullptr;
            sa.src_stride = src_stride; sa.H = H; sa.W = W; sa.n_arrays = n_arrays; sa.batch = batch; sa.n_out = n_out;
            sa.n_strips = n_strips; sa.strip_w = strip_w; sa.wave_cols = wave_cols;
            sa.pitch = std::min(256, (strip_w + 2 * 3 + 1 + 7) & ~7);
            sa.slots = slots;
            sa.l0_mask = 0u; sa.l0_lvl = 0; sa.l0_stride = 0;
            sa.l0_mx = sa.l0_my = ideal_axis(1, 1);
            for (int i = 0; i < 4; i++) sa.l0_dst[i] = nullptr;
            if (l0 >= 0) {
                for (int i = 0; i < 4; i++) sa.l0_dst[i] = dv[l0].dst[i];
                sa.l0_stride = dv[l0].stride; sa.l0_mx = dv[l0].mx; sa.l0_my = dv[l0].my; sa.l0_lvl = l0;
                sa.l0_mask = (1u << n_arrays) - 1u;
            }
            // row segments: enough blocks for several full rounds of the 1024 resident ones (a segment pays
            // 2 RM warm-up rows and a prologue: 256 VGA pairs x 3 arrays, 2 / 3 / 4 / 8 segments: 1.21 / 1.21 / 1.23 / 1.27 ms)
            int n_segs = 1;
            while (images * n_strips * n_segs < 4096 && H / (n_segs * 2) >= 48 && n_segs * 2 <= kStreamMaxSegs) n_segs *= 2;
            sa.seg_rows = (H + n_segs - 1) / n_segs;
            sa.n_segs = (H + sa.seg_rows - 1) / sa.seg_rows;
            size_t lds = 0;
            for (int k = 0; k < 2; k++) {
                const int l = k == 0 ? lA : (nl == 2 ? lB : lA);
                for (int i = 0; i < 4; i++) sa.lv[k].dst[i] = dv[l].dst[i];
                sa.lv[k].dst_stride = dv[l].stride; sa.lv[k].Ho = dv[l].Ho; sa.lv[k].Wo = dv[l].Wo;
                sa.lv[k].mx = dv[l].mx; sa.lv[k].my = dv[l].my;
                sa.lv[k].wr = dv[l].aa.wr; sa.lv[k].wc = dv[l].aa.wc; sa.lv[k].lvl = l;
                if (k < nl) lds += sizeof(double) * kStreamRing * sa.pitch;
            }
            // outputs per strip must fit the 64-column groups the kernel holds terms for
            bool fits = true;
            for (int k = 0; k < nl; k++) {
                const AxisMap &m = sa.lv[k].mx;
                const double step = (axis_pos(m, sa.lv[k].Wo - 1) - axis_pos(m, 0)) / std::max(1, sa.lv[k].Wo - 1);
                if (!(step > 0.0) || (double)strip_w / step + 2.0 > 64.0 * (k == 0 ? kStreamGroupsA : kStreamGroupsB)) fits = false;
                const AxisMap &my = sa.lv[k].my;
                const double stepy = (axis_pos(my, sa.lv[k].Ho - 1) - axis_pos(my, 0)) / std::max(1, sa.lv[k].Ho - 1);
                if (!(stepy > 0.0) || (double)kStreamK / stepy + 2.0 > 64.0) fits = false;
            }
            if (n_strips > kStreamMaxStrips || sa.n_segs > kStreamMaxSegs) fits = false;
            if (sa.seg_rows + 2 * kStreamK >= 60000) fits = false;    // (ring slots by a 16-bit multiply-shift: stream_emit)
            for (int t = 0; fits && t < n_strips; t++) {
                const int xa = t * strip_w, xb = std::min(xa + strip_w, W);
                StreamRange &r = sa.strip_tab[t];
                r.a0 = first_owned(xa, sa.lv[0].mx, sa.lv[0].Wo);
                r.a1 = xb >= W ? sa.lv[0].Wo : first_owned(xb, sa.lv[0].mx, sa.lv[0].Wo);
                r.b0 = first_owned(xa, sa.lv[1].mx, sa.lv[1].Wo);
                r.b1 = xb >= W ? sa.lv[1].Wo : first_owned(xb, sa.lv[1].mx, sa.lv[1].Wo);
            }
            for (int t = 0; fits && t < sa.n_segs; t++) {
                const int ya = t * sa.seg_rows, yb = std::min(ya + sa.seg_rows, H);
                StreamRange &r = sa.seg_tab[t];
                r.a0 = first_owned(ya, sa.lv[0].my, sa.lv[0].Ho);
                r.a1 = yb >= H ? sa.lv[0].Ho : first_owned(yb, sa.lv[0].my, sa.lv[0].Ho);
                r.b0 = first_owned(ya, sa.lv[1].my, sa.lv[1].Ho);
                r.b1 = yb >= H ? sa.lv[1].Ho : first_owned(yb, sa.lv[1].my, sa.lv[1].Ho);
            }
            const int64_t blocks = 8 * ((images + 7) / 8) * n_strips * sa.n_segs;
            if (fits && blocks < (1ll << 31) && lds <= 160 * 1024) {
                if (lds > 64 * 1024) {   // more than 64 KiB of dynamic LDS has to be asked for -- per device (the attribute
                                         // belongs to the function ON the current device), so not remembered in a static
                    TDK_HIP(hipFuncSetAttribute(nl == 2 ? (const void *)k_pyramid_stream<1, 3> : (const void *)k_pyramid_stream<1, 0>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                }
                if (nl == 2) k_pyramid_stream<1, 3><<<(unsigned)blocks, 256, lds, stream>>>(sa);
                else k_pyramid_stream<1, 0><<<(unsigned)blocks, 256, lds, stream>>>(sa);
                TDK_LAUNCH_CHECK();
                done[lA] = true;
                if (nl == 2) done[lB] = true;
                if (l0 >= 0) done[l0] = true;
            }
        }
    }
    // --- the identity-scale level (no filter, same shape) where the streaming kernel did not take it
    for (int l = 0; l < n_out; l++) {
        if (done[l] || dv[l].Ho != H || dv[l].Wo != W || dv[l].aa.Rr || dv[l].aa.Rc) continue;
        Level0Args a;
        for (int i = 0; i < 4; i++) { a.src[i] = i < n_arrays ? srcs[i] : nullptr; a.dst[i] = dv[l].dst[i]; }
        a.src_stride = src_stride; a.dst_stride = dv[l].stride; a.H = H; a.W = W;
        a.n_arrays = n_arrays; a.batch = batch; a.lvl = l; a.n_out = n_out;
        a.mx = dv[l].mx; a.my = dv[l].my; a.slots = slots;
        auto near_identity = [](const AxisMap &m, int n) {
            return fabs(axis_pos(m, 0)) < 0.5 && fabs(axis_pos(m, n - 1) - (double)(n - 1)) < 0.5;
        };
        const int64_t per_image_s = (int64_t)(((W + kL0Cols - 1) / kL0Cols + 3) / 4) * ((H + kL0Seg - 1) / kL0Seg);
        if (images * per_image_s >= 1024 && W >= 3 && H >= 3 && near_identity(a.mx, W) && near_identity(a.my, H)) {
            const int64_t blocks = 8 * ((images + 7) / 8) * per_image_s;
            if (blocks >= (1ll << 31)) { set_error("level 0: grid too large"); return TDK_ERR_INVALID_ARGUMENT; }
            k_level0_stream<<<(unsigned)blocks, 256, 0, stream>>>(a);
        } else {
            const int64_t per_image = (int64_t)((W + 63) / 64) * ((H + 4 * kL0Rows - 1) / (4 * kL0Rows));
            const int64_t blocks = (images < 8 ? images : 8 * ((images + 7) / 8)) * per_image;
            if (blocks >= (1ll << 31)) { set_error("level 0: grid too large"); return TDK_ERR_INVALID_ARGUMENT; }
            k_level0_rows<<<(unsigned)blocks, 256, 0, stream>>>(a);
        }
        TDK_LAUNCH_CHECK();
        done[l] = true;
    }
    // --- levels that shrink both axes and whose tiles fit in LDS take the tiled kernel, all of them in one
    // launch; whatever is left (an enlarged axis, very deep levels, level 0 with a filter) the general one
    AaMultiArgs m;
    m.n = 0;
    size_t lds_max = 0;
    int tiles_total = 0;
    for (int l = 0; l < n_out; l++) {
        if (done[l] || m.n == kAaMaxFused || !shrinks(l)) continue;
        AaTileArgs t;
        const int Rk = dv[l].aa.Rr == dv[l].aa.Rc ? dv[l].aa.Rr : 0;
        // output rows per block: 20 for the 3-tap level, 12 from R = 3 on (measured on the VGA bench batch)
        t.tile_rows = Rk == 1 ? 20 : 12;
        t.mx = dv[l].mx; t.my = dv[l].my;
        // LDS tile bounds from the level's own map
        int max_v = 0, max_c = 0;
        for (int oy0 = 0; oy0 < dv[l].Ho; oy0 += t.tile_rows) {
            const int oy1 = std::min(oy0 + t.tile_rows, dv[l].Ho);
            const int yv0 = (int)floor(axis_pos(t.my, oy0)), yv1 = std::min((int)floor(axis_pos(t.my, oy1 - 1)) + 1, H - 1);
            max_v = std::max(max_v, yv1 - yv0 + 1);
        }
        for (int ox0 = 0; ox0 < dv[l].Wo; ox0 += kAaCols) {
            const int ox1 = std::min(ox0 + kAaCols, dv[l].Wo);
            const int xv0 = (int)floor(axis_pos(t.mx, ox0)), xv1 = std::min((int)floor(axis_pos(t.mx, ox1 - 1)) + 1, W - 1);
            max_c = std::max(max_c, xv1 - xv0 + 1 + 2 * dv[l].aa.Rc);
        }
        t.max_v_rows = max_v;
        t.max_cols = max_c;
        const size_t lds = sizeof(double) * ((size_t)t.max_v_rows * t.max_cols + 2 * dv[l].aa.Rr +
                                             2 * dv[l].aa.Rc + 2) + (size_t)t.tile_rows * 12 + 8;
        if (lds > 64 * 1024) continue;
        for (int i = 0; i < 4; i++) { t.src[i] = i < n_arrays ? srcs[i] : nullptr; t.dst[i] = dv[l].dst[i]; }
        t.src_stride = src_stride; t.dst_stride = dv[l].stride;
        t.H = H; t.W = W; t.Ho = dv[l].Ho; t.Wo = dv[l].Wo;
        t.aa = dv[l].aa;
        t.lvl = l;
        tiles_total += ((dv[l].Ho + t.tile_rows - 1) / t.tile_rows) * ((dv[l].Wo + kAaCols - 1) / kAaCols);
        m.lv[m.n] = t;
        m.tile_end[m.n] = tiles_total;
        m.n++;
        done[l] = true;
        if (lds > lds_max) lds_max = lds;
    }
    if (m.n > 0) {
        m.n_arrays = n_arrays;
        m.batch = batch;
        m.n_out = n_out;
        m.slots = slots;
        const int64_t blocks = (images < 8 ? images : 8 * ((images + 7) / 8)) * tiles_total;
        if (blocks >= (1ll << 31)) {
            set_error("anti-aliased pyramid: %lld blocks exceed the grid limit", (long long)blocks);
            return TDK_ERR_INVALID_ARGUMENT;
        }
        k_rescale_aa_multi<<<(unsigned)blocks, 256, lds_max, stream>>>(m);
        TDK_LAUNCH_CHECK();
    }
    {
        GenericArgs g;
        g.n = 0;
        int total = 0;
        for (int l = 0; l < n_out; l++) {
            if (done[l]) continue;
            g.lv[g.n] = dv[l];
            g.lvl[g.n] = l;
            total += (dv[l].Ho + 3) / 4;   // four output rows per block
            g.blk_end[g.n] = total;
            g.n++;
        }
        if (g.n > 0) {
            for (int i = 0; i < 4; i++) g.src[i] = i < n_arrays ? srcs[i] : nullptr;
            g.src_stride = src_stride; g.H = H; g.W = W; g.slots = slots; g.n_arrays = n_arrays; g.n_out = n_out;
            dim3 grid(total, n_arrays, batch);
            k_rescale_generic<<<grid, 256, 0, stream>>>(g);
            TDK_LAUNCH_CHECK();
        }
    }
    if (slots) {
        ClipArgs c;
        for (int i = 0; i < 4; i++) c.src[i] = i < n_arrays ? srcs[i] : nullptr;
        c.src_stride = src_stride; c.H = H; c.W = W; c.n_arrays = n_arrays; c.batch = batch; c.n_out = n_out;
        for (int l = 0; l < n_out; l++) c.lv[l] = dv[l];
        c.slots = slots;
        int rc_max = 0;
        for (int l = 0; l < n_out; l++) rc_max = std::max(rc_max, dv[l].aa.Rc);
        const size_t lds = sizeof(double) * (size_t)kClipTileRows * (kClipTileCols + 2 * rc_max);
        const int n = (int)(images * n_out);
        unsigned sparse = 0u;                       // levels whose taps can miss a NaN of the filtered image
        for (int l = 0; l < n_out; l++) {
            auto skips = [](const AxisMap &m, int n_in, int n_o, int R) {
                const double p0 = axis_pos(m, 0), p1 = axis_pos(m, n_o - 1);
                const double step = n_o > 1 ? (p1 - p0) / (double)(n_o - 1) : 0.0;
                if (!(step == step) || !(p0 == p0)) return true;
                const int gap = step > 1.0 ? (int)ceil(step) - 1 : 0;
                const int lo = std::max(0, (int)floor(std::min(p0, p1))), hi = std::min(n_in - 1, (int)ceil(std::max(p0, p1)));
                return gap >= 2 * R + 1 || lo > R || (n_in - 1 - hi) > R;
            };
            if (skips(dv[l].mx, W, dv[l].Wo, dv[l].aa.Rc) || skips(dv[l].my, H, dv[l].Ho, dv[l].aa.Rr)) sparse |= 1u << l;
        }
        if (sparse) {
            const int64_t per = std::min<int64_t>(std::max<int64_t>(((int64_t)H * W + 256 * 16 - 1) / (256 * 16), 1), 256);
            if (per * images >= (1ll << 31)) { set_error("clip: grid too large"); return TDK_ERR_INVALID_ARGUMENT; }
            k_clip_nan_scan<<<(unsigned)(per * images), 256, 0, stream>>>(c, sparse, (int)per);
            TDK_LAUNCH_CHECK();
        }
        if (n <= 64) {
            k_clip_small<<<n, 256, lds, stream>>>(c);
            TDK_LAUNCH_CHECK();
            if (slots_clean) *slots_clean = n;
        } else {
            k_clip_gate<<<(n + 255) / 256, 256, 0, stream>>>(slots, n);
            TDK_LAUNCH_CHECK();
            k_clip_bounds<<<1024, 256, lds, stream>>>(c);
            TDK_LAUNCH_CHECK();
            k_clip_apply<<<1024, 256, 0, stream>>>(c);
            TDK_LAUNCH_CHECK();
        }
    }
    return TDK_OK;
}

}  // namespace tdk

namespace {

// one image, one level, host pointers in and out
tdk_status rescale_host(const double *image, int H, int W, double *out, int Ho, int Wo, tdk::PyramidLevelDesc &lv,
                        bool clip) {
    void *d_img, *d_out, *d_w, *d_clip = nullptr;
    TDK_TRY(tdk::scratch(0, (size_t)H * W * 8, &d_img));
    TDK_HIP(hipMemcpyAsync(d_img, image, (size_t)H * W * 8, hipMemcpyHostToDevice, tdk::stream()));
    TDK_TRY(tdk::scratch(1, (size_t)Ho * Wo * 8, &d_out));
    TDK_TRY(tdk::scratch(2, tdk::pyramid_weight_doubles(1) * 8, &d_w));
    if (clip) TDK_TRY(tdk::scratch(3, tdk::pyramid_clip_bytes(1, 1), &d_clip));
    const double *srcs[1] = {(const double *)d_img};
    lv.dst[0] = (double *)d_out; lv.dst[1] = lv.dst[2] = lv.dst[3] = nullptr;
    lv.stride = 0; lv.H = Ho; lv.W = Wo;
    TDK_TRY(tdk::launch_pyramid(srcs, 1, H, W, 0, 1, &lv, 1, (double *)d_w, true, d_clip,
                                tdk::option(TDK_OPT_PYRAMID_STREAM), tdk::stream(), nullptr));
    TDK_HIP(hipMemcpyAsync(out, d_out, (size_t)Ho * Wo * 8, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipStreamSynchronize(tdk::stream()));
    return TDK_OK;
}

}  // namespace

extern "C" {

tdk_status tdk_rescale(const double *image, int H, int W, double *out, int Ho, int Wo) {
    TDK_API_GUARD;
    TDK_REQUIRE(H > 0 && W > 0 && Ho > 0 && Wo > 0 && image && out, "bad argument");
    tdk::PyramidLevelDesc lv;
    lv.H = Ho; lv.W = Wo;
    std::vector<double> storage(2 * (2 * kMaxGaussRadius + 1));
    tdk::ideal_level_plan(&lv, H, W, false, storage.data());
    return rescale_host(image, H, W, out, Ho, Wo, lv, false);
}

tdk_status tdk_rescale_anti_aliased(const double *image, int H, int W, double *out, int Ho, int Wo) {
    TDK_API_GUARD;
    TDK_REQUIRE(H > 0 && W > 0 && Ho > 0 && Wo > 0 && image && out, "bad argument");
    tdk::PyramidLevelDesc lv;
    lv.H = Ho; lv.W = Wo;
    std::vector<double> storage(2 * (2 * kMaxGaussRadius + 1));
    tdk::ideal_level_plan(&lv, H, W, true, storage.data());
    return rescale_host(image, H, W, out, Ho, Wo, lv, false);
}

tdk_status tdk_rescale_skimage(const double *image, int H, int W, double *out, int Ho, int Wo, const double *map,
                               const double *w_rows, int radius_rows, const double *w_cols, int radius_cols,
                               int clip) {
    TDK_API_GUARD;
    TDK_REQUIRE(H > 0 && W > 0 && Ho > 0 && Wo > 0 && image && out && map, "bad argument");
    TDK_REQUIRE(radius_rows >= 0 && radius_cols >= 0 && radius_rows <= kMaxGaussRadius && radius_cols <= kMaxGaussRadius,
                "kernel radius out of range");
    // (a one-pixel output axis: skimage's estimate of the degenerate corner set gives scale 0 and only the offset is used)
    TDK_REQUIRE((map[0] > 0.0 || (Wo == 1 && map[0] == 0.0)) && (map[2] > 0.0 || (Ho == 1 && map[2] == 0.0)),
                "the map's scales must be positive");
    tdk::PyramidLevelDesc lv;
    lv.mx = tdk::affine_axis(map[0], map[1]);
    lv.my = tdk::affine_axis(map[2], map[3]);
    lv.wr = radius_rows > 0 ? w_rows : nullptr; lv.Rr = radius_rows;
    lv.wc = radius_cols > 0 ? w_cols : nullptr; lv.Rc = radius_cols;
    return rescale_host(image, H, W, out, Ho, Wo, lv, clip != 0);
}

}  // extern "C"
