// granular.hip -- the parity-granular operators: one C-ABI entry per function
// of the reference's native boundary (rust_bindings.{warp,projection,transform,
// interpolation,triangulation}, tadataka.camera._normalizer), host pointers in
// and out.  Compiled with -ffp-contract=off: each + - * / is one IEEE rounding,
// so these agree bit for bit with the reference arithmetic order.
//
// They exist for drop-in compatibility and parity checking; the fast path is
// the fused, device-resident DVO batch in dvo.hip.
#include "tdk_math.h"
#include "tdk_runtime.h"

#include <vector>

namespace {

using tdk::Cam;

constexpr int kBlock = 256;

inline int grid_for(int64_t n) {
    int64_t g = (n + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if (g > 65536) g = 65536;
    return (int)g;
}

__global__ __launch_bounds__(kBlock) void k_normalize(const double2 *__restrict__ kp, int64_t n,
                                                      Cam c, double2 *__restrict__ out, int inverse) {
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        double2 p = kp[i], q;
        if (inverse) tdk::unnormalize(c, p.x, p.y, q.x, q.y);
        else tdk::normalize(c, p.x, p.y, q.x, q.y);
        out[i] = q;
    }
}

__global__ __launch_bounds__(kBlock) void k_project(const double *__restrict__ P, int64_t n,
                                                    double2 *__restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        double2 q;
        tdk::project(P[3 * i], P[3 * i + 1], P[3 * i + 2], q.x, q.y);
        out[i] = q;
    }
}

__global__ __launch_bounds__(kBlock) void k_inv_project(const double2 *__restrict__ xs,
                                                        const double *__restrict__ d, int64_t n,
                                                        double *__restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        double2 x = xs[i];
        double di = d[i];
        out[3 * i] = x.x * di;
        out[3 * i + 1] = x.y * di;
        out[3 * i + 2] = 1.0 * di;
    }
}

struct Mat4 {
    double m[16];
};

__global__ __launch_bounds__(kBlock) void k_transform(Mat4 T, const double *__restrict__ P, int64_t n,
                                                      double *__restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        double qx, qy, qz;
        tdk::transform(T.m, P[3 * i], P[3 * i + 1], P[3 * i + 2], qx, qy, qz);
        out[3 * i] = qx;
        out[3 * i + 1] = qy;
        out[3 * i + 2] = qz;
    }
}

__global__ __launch_bounds__(kBlock) void k_warp(Mat4 T, const double2 *__restrict__ xs,
                                                 const double *__restrict__ d, int64_t n,
                                                 double2 *__restrict__ oxs, double *__restrict__ od) {
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        double2 x = xs[i], q;
        double d1;
        tdk::warp(T.m, x.x, x.y, d[i], q.x, q.y, d1);
        oxs[i] = q;
        od[i] = d1;
    }
}

__global__ __launch_bounds__(kBlock) void k_interpolation(const double *__restrict__ img, int H, int W,
                                                          const double2 *__restrict__ c, int64_t m,
                                                          double *__restrict__ out,
                                                          int *__restrict__ bad) {
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kBlock) {
        double2 p = c[i];
        if (!tdk::in_range(p.x, p.y, H, W)) {
            atomicOr(bad, 1);
            out[i] = 0.0;
            continue;
        }
        out[i] = tdk::bilinear(img, H, W, p.x, p.y);
    }
}

// np.gradient, unit spacing: central differences inside, one-sided at the
// borders (tadataka/vo/dvo/jacobian.py:27-29).  One thread per pixel.
__global__ __launch_bounds__(kBlock) void k_image_gradient(const double *__restrict__ I, int H, int W,
                                                           double *__restrict__ GX,
                                                           double *__restrict__ GY) {
    int64_t N = (int64_t)H * W;
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < N; i += (int64_t)gridDim.x * kBlock) {
        int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
        int xl = max(x - 1, 0), xr = min(x + 1, W - 1);
        int yl = max(y - 1, 0), yr = min(y + 1, H - 1);
        double gx = 0.0, gy = 0.0;
        if (xr > xl) gx = (I[(int64_t)y * W + xr] - I[(int64_t)y * W + xl]) / (double)(xr - xl);
        if (yr > yl) gy = (I[(int64_t)yr * W + x] - I[(int64_t)yl * W + x]) / (double)(yr - yl);
        GX[i] = gx;
        GY[i] = gy;
    }
}

__device__ __forceinline__ int reflect_idx(int64_t i, int n) {
    if (n == 1) return 0;
    int64_t p = 2 * (int64_t)n;
    i %= p;
    if (i < 0) i += p;
    if (i >= n) i = p - 1 - i;
    return (int)i;
}

}  // namespace

namespace {

__global__ __launch_bounds__(256) void k_rescale(const double *__restrict__ src, int H, int W,
                                                 double *__restrict__ dst, int Ho, int Wo,
                                                 int64_t src_stride, int64_t dst_stride) {
    const double *s = src + (int64_t)blockIdx.y * src_stride;
    double *d = dst + (int64_t)blockIdx.y * dst_stride;
    double sy = (double)H / (double)Ho, sx = (double)W / (double)Wo;
    int64_t No = (int64_t)Ho * Wo;
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < No; i += (int64_t)gridDim.x * 256) {
        int oy = (int)(i / Wo), ox = (int)(i - (int64_t)oy * Wo);
        double cy = ((double)oy + 0.5) * sy - 0.5;
        double cx = ((double)ox + 0.5) * sx - 0.5;
        double fy0 = floor(cy), fx0 = floor(cx);
        double wy = cy - fy0, wx = cx - fx0;
        int y0 = reflect_idx((int64_t)fy0, H), y1 = reflect_idx((int64_t)fy0 + 1, H);
        int x0 = reflect_idx((int64_t)fx0, W), x1 = reflect_idx((int64_t)fx0 + 1, W);
        double top = s[(int64_t)y0 * W + x0] * (1.0 - wx) + s[(int64_t)y0 * W + x1] * wx;
        double bot = s[(int64_t)y1 * W + x0] * (1.0 - wx) + s[(int64_t)y1 * W + x1] * wx;
        d[i] = top * (1.0 - wy) + bot * wy;
    }
}

}  // namespace

namespace {

// ---------------------------------------------------------------------------
// Whole pyramid in one pass over the full-resolution frames.
//
// _estimate_at rescales the ORIGINAL I0/D0/I1/W0 for every level
// (tadataka/vo/dvo/__init__.py:144-148), so building the levels one by one reads
// level 0 (n_levels - 1) times.  Here a block stages one source tile (plus a
// 1-texel reflected halo) in LDS once and emits, for every level, exactly the
// output pixels whose lower tap (floor of the sample position) lies in its tile
// -- a partition of each level's pixels that needs no inter-block agreement.
// Per-pixel arithmetic is k_rescale's, so the result is bit-identical.
// ---------------------------------------------------------------------------
constexpr int kPyrTW = 94, kPyrTH = 32;   // 94 source columns -> <= 64 outputs per row at ratio 1.5
constexpr int kPyrLW = kPyrTW + 2, kPyrLH = kPyrTH + 2;

struct PyrLevel {
    double *dst[4];
    int64_t stride;
    int Ho, Wo;
};

struct PyrArgs {
    const double *src[4];
    int64_t src_stride;
    int H, W, n_arrays, n_out;   // n_out = levels to produce (levels 1 .. n_out)
    PyrLevel lv[15];
};

__device__ __forceinline__ int reflect_fast(int i, int n) {
    return ((unsigned)i < (unsigned)n) ? i : reflect_idx((int64_t)i, n);
}

// first output index o in [0, n_out] whose lower tap max(floor((o + 0.5) s - 0.5), 0) is >= s0
__device__ __forceinline__ int first_owned(int s0, double s, int n_out) {
    int o = (int)ceil(((double)s0 + 0.5) / s - 0.5);
    o = max(0, min(o, n_out));
    while (o > 0 && max((int)floor(((double)(o - 1) + 0.5) * s - 0.5), 0) >= s0) o--;
    while (o < n_out && max((int)floor(((double)o + 0.5) * s - 0.5), 0) < s0) o++;
    return o;
}

__global__ __launch_bounds__(256) void k_pyramid(PyrArgs a) {
    __shared__ double tile[kPyrLH][kPyrLW];
    const int arr = blockIdx.z % a.n_arrays, pair = blockIdx.z / a.n_arrays;
    const double *src = a.src[arr] + (int64_t)pair * a.src_stride;
    const int sx0 = blockIdx.x * kPyrTW, sy0 = blockIdx.y * kPyrTH;
    const int sx1 = min(sx0 + kPyrTW, a.W), sy1 = min(sy0 + kPyrTH, a.H);
    const int lw = sx1 - sx0 + 2, lh = sy1 - sy0 + 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // LDS position (r, c) holds source texel (sy0 - 1 + r, sx0 - 1 + c), reflected
    for (int r = wave; r < lh; r += 4) {
        const double *row = src + (int64_t)reflect_fast(sy0 - 1 + r, a.H) * a.W;
        for (int c = lane; c < lw; c += 64) tile[r][c] = row[reflect_fast(sx0 - 1 + c, a.W)];
    }
    __syncthreads();
    for (int l = 0; l < a.n_out; l++) {
        const PyrLevel &L = a.lv[l];
        const double sy = (double)a.H / (double)L.Ho, sx = (double)a.W / (double)L.Wo;
        // this tile owns the outputs whose (clamped) lower tap lies inside it;
        // the last tile also takes whatever maps beyond the image edge
        const int ox_lo = first_owned(sx0, sx, L.Wo), oy_lo = first_owned(sy0, sy, L.Ho);
        const int ox_hi = sx1 >= a.W ? L.Wo : first_owned(sx1, sx, L.Wo);
        const int oy_hi = sy1 >= a.H ? L.Ho : first_owned(sy1, sy, L.Ho);
        double *dst = L.dst[arr] + (int64_t)pair * L.stride;
        for (int oy = oy_lo + wave; oy < oy_hi; oy += 4) {
            double cy = ((double)oy + 0.5) * sy - 0.5;
            double fy0 = floor(cy);
            const double wy = cy - fy0;
            const int r = (int)fy0 - (sy0 - 1);
            for (int ox = ox_lo + lane; ox < ox_hi; ox += 64) {
                double cx = ((double)ox + 0.5) * sx - 0.5;
                double fx0 = floor(cx);
                const double wx = cx - fx0;
                const int c = (int)fx0 - (sx0 - 1);
                double top = tile[r][c] * (1.0 - wx) + tile[r][c + 1] * wx;
                double bot = tile[r + 1][c] * (1.0 - wx) + tile[r + 1][c + 1] * wx;
                dst[(int64_t)oy * L.Wo + ox] = top * (1.0 - wy) + bot * wy;
            }
        }
    }
}

// Every level of every array in one launch (k_rescale's arithmetic, bit-identical),
// one output row per wave.  blockIdx.x runs over the row groups of level 1, then
// level 2, ... of ONE (pair, array); y = array, z = pair.  Blocks
// are dispatched x-fastest, so all the levels of an image are resampled within
// microseconds of each other and only the first pass over its level-0 texels
// comes from HBM -- the later ones hit the 256 MiB Infinity Cache.
struct RescaleArgs {
    const double *src[4];
    int64_t src_stride;
    int H, W, n_out;
    int blk_end[15];   // cumulative block count per level
    PyrLevel lv[15];
};

__global__ __launch_bounds__(256) void k_rescale_levels(RescaleArgs a) {
    int l = 0;
    while (l + 1 < a.n_out && (int)blockIdx.x >= a.blk_end[l]) l++;
    const PyrLevel &L = a.lv[l];
    const int arr = blockIdx.y, pair = blockIdx.z;
    // one output row per wave (4 rows per block), lanes stride along the row: no
    // division per pixel, the row terms are wave-uniform
    const int oy = (((int)blockIdx.x - (l ? a.blk_end[l - 1] : 0)) << 2) + (int)(threadIdx.x >> 6);
    if (oy >= L.Ho) return;
    const int H = a.H, W = a.W;
    const double sy = (double)H / (double)L.Ho, sx = (double)W / (double)L.Wo;
    double cy = ((double)oy + 0.5) * sy - 0.5;
    double fy0 = floor(cy);
    double wy = cy - fy0;
    const int iy = (int)fy0;
    const double *row0 = a.src[arr] + (int64_t)pair * a.src_stride + (int64_t)reflect_fast(iy, H) * W;
    const double *row1 = a.src[arr] + (int64_t)pair * a.src_stride + (int64_t)reflect_fast(iy + 1, H) * W;
    double *d = L.dst[arr] + (int64_t)pair * L.stride + (int64_t)oy * L.Wo;
    for (int ox = threadIdx.x & 63; ox < L.Wo; ox += 64) {
        double cx = ((double)ox + 0.5) * sx - 0.5;
        double fx0 = floor(cx);
        double wx = cx - fx0;
        const int ix = (int)fx0;
        const int x0 = reflect_fast(ix, W), x1 = reflect_fast(ix + 1, W);
        double top = row0[x0] * (1.0 - wx) + row0[x1] * wx;
        double bot = row1[x0] * (1.0 - wx) + row1[x1] * wx;
        d[ox] = top * (1.0 - wy) + bot * wy;
    }
}

}  // namespace

namespace {

// ---------------------------------------------------------------------------
// Anti-aliased rescale: skimage.transform.rescale's default when it shrinks an
// image (scikit-image 0.15+; the reference pins 0.16.2 and calls
// rescale(image, scale), vo/dvo/__init__.py:144-148): a Gaussian prefilter
// scipy.ndimage.gaussian_filter(image, sigma = (factor - 1) / 2, mode='mirror')
// and then the bilinear warp of k_rescale.  The arithmetic follows ndimage's
// correlate1d operation by operation (the CPU restatement the tests compare with
// is pinned against scipy.ndimage itself, bit for bit).  The filtered image
// is never stored: every output pixel evaluates the separable filter at its
// four taps (vertical pass first, as ndimage does axis 0 first).  Simple rather
// than fast -- (2 Rr + 1)(2 Rc + 1) loads per tap -- it serves the drop-in API,
// the headline bench uses the plain bilinear pyramid of SURVEY cfg2.
// ---------------------------------------------------------------------------
constexpr int kMaxGaussRadius = 64;

struct AaLevel {
    const double *wr, *wc;   // device: 2 R + 1 weights each
    int Rr, Rc;
};

struct RescaleAaArgs {
    RescaleArgs r;
    AaLevel aa[15];
};

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

// correlate1d with a symmetric kernel along a column of the source image:
// centre tap first, then the pairs from the outermost inwards
__device__ __forceinline__ double column_tap(const double *__restrict__ s, int H, int W, int y, int x,
                                             const double *__restrict__ w, int R) {
    double tmp = s[y * W + x] * w[R];
    for (int j = -R; j < 0; j++) tmp += (s[mirror_idx(y + j, H) * W + x] + s[mirror_idx(y - j, H) * W + x]) * w[R + j];
    return tmp;
}

// ... and along a row of the vertically filtered image
__device__ __forceinline__ double filtered_tap(const double *__restrict__ s, int H, int W, int y, int x,
                                               const AaLevel &a) {
    double tmp = column_tap(s, H, W, y, x, a.wr, a.Rr) * a.wc[a.Rc];
    for (int j = -a.Rc; j < 0; j++)
        tmp += (column_tap(s, H, W, y, mirror_idx(x + j, W), a.wr, a.Rr) +
                column_tap(s, H, W, y, mirror_idx(x - j, W), a.wr, a.Rr)) * a.wc[a.Rc + j];
    return tmp;
}

__global__ __launch_bounds__(256) void k_rescale_levels_aa(RescaleAaArgs args) {
    const RescaleArgs &a = args.r;
    int l = 0;
    while (l + 1 < a.n_out && (int)blockIdx.x >= a.blk_end[l]) l++;
    const PyrLevel &L = a.lv[l];
    const AaLevel &aa = args.aa[l];
    const int arr = blockIdx.y, pair = blockIdx.z;
    // one thread per output pixel: each one is hundreds of dependent-address loads,
    // so a single frame wants all the parallelism it can get
    const int i = ((int)blockIdx.x - (l ? a.blk_end[l - 1] : 0)) * 256 + (int)threadIdx.x;
    if (i >= L.Ho * L.Wo) return;
    const int oy = i / L.Wo, ox = i - oy * L.Wo;
    const int H = a.H, W = a.W;
    const double *s = a.src[arr] + (int64_t)pair * a.src_stride;
    const double sy = (double)H / (double)L.Ho, sx = (double)W / (double)L.Wo;
    double cy = ((double)oy + 0.5) * sy - 0.5;
    double cx = ((double)ox + 0.5) * sx - 0.5;
    double fy0 = floor(cy), fx0 = floor(cx);
    double wy = cy - fy0, wx = cx - fx0;
    const int iy = (int)fy0, ix = (int)fx0;
    const int y0 = reflect_fast(iy, H), y1 = reflect_fast(iy + 1, H);
    const int x0 = reflect_fast(ix, W), x1 = reflect_fast(ix + 1, W);
    double top = filtered_tap(s, H, W, y0, x0, aa) * (1.0 - wx) + filtered_tap(s, H, W, y0, x1, aa) * wx;
    double bot = filtered_tap(s, H, W, y1, x0, aa) * (1.0 - wx) + filtered_tap(s, H, W, y1, x1, aa) * wx;
    L.dst[arr][(int64_t)pair * L.stride + i] = top * (1.0 - wy) + bot * wy;
}

// Tiled form of the same arithmetic for one level, used whenever the level shrinks both
// axes and its tiles fit in LDS: a block produces tile_rows x kAaCols output pixels.
//   1. the vertical Gaussian is evaluated once per (row, column) its taps and their
//      horizontal support touch (mirror boundary), from global memory into an LDS tile
//      V -- ndimage filters axis 0 first, so V is rounded exactly like its intermediate image,
//   2. every thread evaluates the horizontal Gaussian of V at its four taps and blends.
// Same operations in the same order as filtered_tap(), so the results are bit-identical
// to k_rescale_levels_aa; ~70 LDS reads per output instead of ~200 global loads.
// tile_rows x 64 outputs per block (several per thread): with 4 x 64 a VGA batch was
// 430 000 blocks of two barriers and ~5 us of dependent latency each -- block-turnover
// bound at 1.6 TB/s; taller tiles also cut the vertical halo ((rows f + 2 R + 1) / (rows f)).
constexpr int kAaCols = 64;

struct AaTileArgs {
    const double *src[4];
    double *dst[4];
    int64_t src_stride, dst_stride;
    int H, W, Ho, Wo;
    AaLevel aa;
    int tile_rows;              // output rows per block (kAaRows, or the per-radius choice of the host)
    int max_v_rows, max_cols;   // LDS tile bounds (host: ceil(rows * factor) + 2, ceil(cols * factor) + 2 + 2 Rc)
};

// Any radii (from the arguments): the levels aa_tile_fixed<R> has no instantiation for.
__device__ __forceinline__ void aa_tile_generic(const AaTileArgs &a, int tile, int arr, int pair,
                                                unsigned char *aa_smem) {
    const int Rr = a.aa.Rr, Rc = a.aa.Rc;
    const int SC = a.max_cols;
    double *V = reinterpret_cast<double *>(aa_smem);           // [max_v_rows][SC] vertically filtered
    double *wr = V + (size_t)a.max_v_rows * SC;                 // [2 Rr + 1] kernel weights, LDS copies:
    double *wc = wr + 2 * Rr + 1;                               // [2 Rc + 1] broadcast reads in the inner loops
    for (int k = threadIdx.x; k < 2 * Rr + 1; k += 256) wr[k] = a.aa.wr[k];
    for (int k = threadIdx.x; k < 2 * Rc + 1; k += 256) wc[k] = a.aa.wc[k];
    const int tiles_x = (a.Wo + kAaCols - 1) / kAaCols;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int H = a.H, W = a.W;
    const double *s = a.src[arr] + (int64_t)pair * a.src_stride;
    const double sy = (double)H / (double)a.Ho, sx = (double)W / (double)a.Wo;
    const int oy0 = ty * a.tile_rows, oy1 = min(oy0 + a.tile_rows, a.Ho);
    const int ox0 = tx * kAaCols, ox1 = min(ox0 + kAaCols, a.Wo);
    // first / last lower tap of the tile (a shrinking level: all taps lie inside the image)
    const int yv0 = (int)floor(((double)oy0 + 0.5) * sy - 0.5);
    const int yv1 = min((int)floor(((double)(oy1 - 1) + 0.5) * sy - 0.5) + 1, H - 1);
    const int xv0 = (int)floor(((double)ox0 + 0.5) * sx - 0.5);
    const int xv1 = min((int)floor(((double)(ox1 - 1) + 0.5) * sx - 0.5) + 1, W - 1);
    const int nv = yv1 - yv0 + 1;                                // V rows
    const int nc = xv1 - xv0 + 1 + 2 * Rc;                       // columns incl. the horizontal support
    const int xs0 = xv0 - Rc;
    __syncthreads();                                             // weights in place
    // vertical Gaussian straight from global memory (a source texel is re-read 2 Rr + 1
    // times by the block: L1 hits); a wave per V row, lanes along it: coalesced, no division
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = wave; r < nv; r += 4) {
        const int y = yv0 + r;
        for (int c = lane; c < nc; c += 64) {
            const double *col = s + mirror_idx(xs0 + c, W);
            double tmp = col[(int64_t)y * W] * wr[Rr];
            for (int j = -Rr; j < 0; j++)
                tmp += (col[(int64_t)mirror_idx(y + j, H) * W] + col[(int64_t)mirror_idx(y - j, H) * W]) * wr[Rr + j];
            V[r * SC + c] = tmp;
        }
    }
    __syncthreads();
    const int ox = ox0 + (int)(threadIdx.x & 63);
    if (ox >= ox1) return;
    const double cx = ((double)ox + 0.5) * sx - 0.5;
    const double fx0 = floor(cx);
    const double wx = cx - fx0;
    const int ix = (int)fx0;
    const int x0 = reflect_fast(ix, W), x1 = reflect_fast(ix + 1, W);
    for (int oy = oy0 + (int)(threadIdx.x >> 6); oy < oy1; oy += 4) {
        const double cy = ((double)oy + 0.5) * sy - 0.5;
        const double fy0 = floor(cy);
        const double wy = cy - fy0;
        const int iy = (int)fy0;
        const int y0 = reflect_fast(iy, H), y1 = reflect_fast(iy + 1, H);
        double f[2][2];
#pragma unroll
        for (int ry = 0; ry < 2; ry++) {
#pragma unroll
            for (int rx = 0; rx < 2; rx++) {
                const double *row = V + ((ry ? y1 : y0) - yv0) * SC + ((rx ? x1 : x0) - xs0);
                double tmp = row[0] * wc[Rc];
                for (int j = -Rc; j < 0; j++) tmp += (row[j] + row[-j]) * wc[Rc + j];
                f[ry][rx] = tmp;
            }
        }
        double top = f[0][0] * (1.0 - wx) + f[0][1] * wx;
        double bot = f[1][0] * (1.0 - wx) + f[1][1] * wx;
        a.dst[arr][(int64_t)pair * a.dst_stride + (int64_t)oy * a.Wo + ox] = top * (1.0 - wy) + bot * wy;
    }
}

// The tile for a compile-time radius R (both axes): the pyramid levels of the bench (R = 1, 3, 5
// at ratio 1.5).  Same products and sums in the same order as aa_tile_generic / filtered_tap(), so
// bit-identical; what differs is the bookkeeping around them -- on this kernel 70 % of the
// issued VALU work was integer address arithmetic and predication, not the FP64 filter:
//   * the wave index is made scalar (readfirstlane), so row numbers, boundary reflection of
//     rows and row base pointers live in SGPRs: a source load is `global_load v, voff, s[row]`
//     with one per-lane column offset for the whole walk -- no 64-bit multiply per load;
//   * a wave loads the CH + 2 R source rows of its column walk unconditionally (rows beyond
//     the wave's share are clamped to the last one and unused): no per-load exec masking;
//   * a shrinking level has all four taps inside the image, x1 = x0 + 1 and y1 = y0 + 1, so the
//     two horizontal filters of a V row share 2 R of their 2 R + 1 LDS reads;
//   * the per-row terms (wy and the V offset of the upper tap) are computed once per tile into
//     LDS instead of once per wave and output row in FP64 on the vector ALU.
template <int R>
__device__ __forceinline__ void aa_tile_fixed(const AaTileArgs &a, int tile, int arr, int pair,
                                              unsigned char *aa_smem) {
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
    const double sy = (double)H / (double)a.Ho, sx = (double)W / (double)a.Wo;
    const int oy0 = ty * a.tile_rows, oy1 = min(oy0 + a.tile_rows, a.Ho);
    const int ox0 = tx * kAaCols, ox1 = min(ox0 + kAaCols, a.Wo);
    // first / last lower tap of the tile (a shrinking level: all taps lie inside the image)
    const int yv0 = (int)floor(((double)oy0 + 0.5) * sy - 0.5);
    const int yv1 = min((int)floor(((double)(oy1 - 1) + 0.5) * sy - 0.5) + 1, H - 1);
    const int xv0 = (int)floor(((double)ox0 + 0.5) * sx - 0.5);
    const int xv1 = min((int)floor(((double)(ox1 - 1) + 0.5) * sx - 0.5) + 1, W - 1);
    const int nv = yv1 - yv0 + 1;                                // V rows
    const int nc = xv1 - xv0 + 1 + 2 * R;                        // columns incl. the horizontal support
    const int xs0 = xv0 - R;
    double wk[R + 1], wck[R + 1];                                // kernel halves (uniform: scalar loads)
#pragma unroll
    for (int k = 0; k <= R; k++) { wk[k] = a.aa.wr[k]; wck[k] = a.aa.wc[k]; }
    if ((int)threadIdx.x < oy1 - oy0) {                          // per-row terms of the blend
        const double cy = ((double)(oy0 + (int)threadIdx.x) + 0.5) * sy - 0.5;
        const double fy0 = floor(cy);
        row_wy[threadIdx.x] = cy - fy0;
        row_off[threadIdx.x] = ((int)fy0 - yv0) * SC;
    }
    // vertical Gaussian straight from global memory: every wave takes a quarter of the V rows and
    // walks down its columns with the source texels of the whole walk in registers, all loads
    // issued before the first one is used
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
        } else if (rows > 0) {                                   // taller tiles (tuning knobs): sliding window
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
    const double cx = ((double)ox + 0.5) * sx - 0.5;
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
        const double top = f[0][0] * (1.0 - wx) + f[0][1] * wx;
        const double bot = f[1][0] * (1.0 - wx) + f[1][1] * wx;
        dst[(int64_t)(oy0 + i) * a.Wo + ox] = top * (1.0 - wy) + bot * wy;
    }
}

// Every tiled level of the pyramid in ONE launch: the tiles of level 1, then level 2, ... of one
// (array, pair) are consecutive work items of ONE XCD (see the kernel), so the coarser levels of an
// image are produced right after the finer ones and find the full-resolution source in that L2.
constexpr int kAaMaxFused = 4;
struct AaMultiArgs {
    int n, n_arrays, batch;
    int tile_end[kAaMaxFused];
    AaTileArgs lv[kAaMaxFused];
};

__global__ __launch_bounds__(256) void k_rescale_aa_multi(AaMultiArgs m) {
    extern __shared__ __attribute__((aligned(16))) unsigned char aa_smem[];
    // 1-D grid.  Workgroups are dealt to the 8 XCDs round-robin and every XCD has its own L2: XCD k
    // takes images k, k + 8, ... one after the other, all tiles of all levels of an image consecutively
    // -- so the halo rows that neighbouring tiles share and the second level's pass over the same
    // source (2.4 MB per VGA array, the L2 holds 4 MB) are L2 hits instead of fabric traffic.
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
    switch (R) {      // block-uniform
        case 1: aa_tile_fixed<1>(a, tile, arr, pair, aa_smem); break;   // ratio 1.5, level 1
        case 3: aa_tile_fixed<3>(a, tile, arr, pair, aa_smem); break;   // level 2
        case 5: aa_tile_fixed<5>(a, tile, arr, pair, aa_smem); break;   // level 3
        default: aa_tile_generic(a, tile, arr, pair, aa_smem); break;
    }
}

// ---------------------------------------------------------------------------
// Row-streaming form of the anti-aliased pyramid (round 4): one pass over the source for
// up to TWO levels.
//
// The tiles above re-load a tile's source rows once per level and once per 8 V rows
// (14 loads per 8 rows at R = 3), a block lives for two short phases that do not overlap,
// and level 2 costs as much as level 1 although it moves no HBM bytes.  Here a block owns
// a full-height strip of source columns, one column per thread, and walks DOWN it:
//   * every source texel is loaded exactly once (rows of the strip are contiguous:
//     coalesced 8-byte loads), the loads of chunk c + 1 are issued before chunk c is
//     computed (software prefetch in registers);
//   * the thread keeps the last 2 RM rows of its column in registers, so the vertical
//     Gaussians of BOTH levels (radius RA and RB) come from the same registers: K new V rows
//     per level per chunk, written to per-level LDS rings;
//   * after a barrier the waves take (output row, 64-column group) units of both levels
//     whose two V rows are now complete: horizontal Gaussian at the four taps from LDS,
//     blend, store -- aa_tile_fixed's arithmetic, operation by operation (centre tap, pairs
//     from the outermost inwards; top / bottom / rows), so the output is bit-identical.
// V rows live in rings of K + 2 rows per level (two barriers per chunk): 35.8 KB of LDS for a VGA
// strip, 4 blocks per CU.  Measured on 256 VGA pairs x 3 arrays (profiles/r04_pyramid.txt): rings of
// 2 K + 1 rows with one barrier (2 blocks per CU) 1.44 ms, K + 2 rows 0.99, ring pitch = the strip's
// columns instead of 256 0.91; K = 4 / 6 / 8 / 9 / 12: 1.31 / 1.09 / 0.91 / 0.90 / 0.95 (K >= 10 loses a
// block per CU); the tiles above 1.035.  Timing-only ablations (-DTDK_STREAM_ABL_*): no emission 0.355
// (the source streams at 5.3 TB/s), emission without LDS reads and stores 0.56, + LDS reads 0.68,
// + stores 0.73, both 0.91 -- the phases add up instead of overlapping; conflict-free (wrong) tap
// addresses 0.89, 16-byte stores by lane pairs 0.93, the second barrier nothing.
// ---------------------------------------------------------------------------
#ifndef TDK_STREAM_K
#define TDK_STREAM_K 8
#endif
#ifndef TDK_STREAM_RING
#define TDK_STREAM_RING (TDK_STREAM_K + 2)
#endif
constexpr int kStreamK = TDK_STREAM_K;            // source rows per chunk
constexpr int kStreamRing = TDK_STREAM_RING;      // V rows per level ring (>= K + 2; < 2 K + 1: a second barrier per chunk)
static_assert(kStreamRing >= kStreamK + 2 && kStreamK <= 32, "ring too small");
constexpr int kStreamMaxGroups = 4;               // 64-column groups of outputs per strip and level

struct StreamLevel {
    double *dst[4];
    int64_t dst_stride;
    int Ho, Wo;
    const double *wr, *wc;                        // device: kernel halves incl. centre (R + 1 used)
};

struct StreamArgs {
    const double *src[4];
    int64_t src_stride;
    int H, W, n_arrays, batch;
    int n_strips, strip_w;                        // owned source columns per strip
    int pitch;                                    // ring row pitch: strip_w + 2 RM + 1 columns, rounded up (<= 256 threads)
    int n_segs, seg_rows;                         // row segments: a block emits the outputs whose upper tap lies in its segment
    StreamLevel lv[2];
};

__device__ __forceinline__ double wx_fake(double a, int q) { return a + (double)q; }   // ablation builds only

template <int R>
__device__ __forceinline__ double stream_vtap(const double *w, int c, const double (&wk)[R + 1]) {
    double tmp = w[c] * wk[R];
#pragma unroll
    for (int j = -R; j < 0; j++) tmp += (w[c + j] + w[c - j]) * wk[R + j];
    return tmp;
}

// the horizontal pass + blend of one level for the output rows [oy_lo, oy_hi) of this chunk
template <int R>
__device__ __forceinline__ int stream_emit(const StreamLevel &L, const double *__restrict__ ring, int SW, double *dst,
                                           double sy, int oy_lo, int oy_end, int ynew, int ya, int n_groups,
                                           int ncols, int ox_first,
                                           const int (&xoff)[kStreamMaxGroups],
                                           const double (&wxs)[kStreamMaxGroups], const double (&wck)[R + 1],
                                           int wave, int lane, int &unit) {
    // the row terms of the next rows: lane i computes those of row oy_lo + i, the rows read them by lane;
    // the rows to emit now are those whose lower tap y0 + 1 is in the ring (a prefix: y0 is monotone)
    const double my_cy = ((double)(oy_lo + lane) + 0.5) * sy - 0.5;
    const double my_fy = floor(my_cy);
    const double my_wy = my_cy - my_fy;
    const int my_y0 = (int)my_fy;
    const int n_rows = __builtin_popcountll(__ballot(oy_lo + lane < oy_end && my_y0 + 1 <= ynew));
    for (int i = 0; i < n_rows; i++) {                            // wave-uniform
        const int y0 = __builtin_amdgcn_readlane(my_y0, i);
        const double wy = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(my_wy), i),
                                           __builtin_amdgcn_readlane(__double2loint(my_wy), i));
        const int slot0 = (y0 - ya) % kStreamRing, slot1 = (y0 + 1 - ya) % kStreamRing;
        double *dst_row = dst + (int64_t)(oy_lo + i) * L.Wo + ox_first;   // uniform base, the lane is the offset
#pragma unroll
        for (int g = 0; g < kStreamMaxGroups; g++) {
            if (g >= n_groups) break;
            const bool mine = ((unit++) & 3) == wave;             // units dealt round-robin to the waves
            if (!mine) continue;
            if (g * 64 + lane >= ncols) continue;
            const double *p0 = ring + slot0 * SW + xoff[g];
            const double *p1 = ring + slot1 * SW + xoff[g];
            double u[2][2 * R + 2];                               // V rows y0, y0 + 1, columns x0 - R .. x0 + 1 + R
#pragma unroll
#ifdef TDK_STREAM_ABL_NOLDSREAD
            for (int q = 0; q < 2 * R + 2; q++) { u[0][q] = wx_fake(wxs[g], q); u[1][q] = wx_fake(wy, q); }
#else
            for (int q = 0; q < 2 * R + 2; q++) { u[0][q] = p0[q - R]; u[1][q] = p1[q - R]; }
#endif
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
            const double wx = wxs[g];
            const double top = f[0][0] * (1.0 - wx) + f[0][1] * wx;
            const double bot = f[1][0] * (1.0 - wx) + f[1][1] * wx;
#ifdef TDK_STREAM_ABL_NOSTORE
            if (wy == 12345.0)
#endif
            (dst_row + g * 64)[lane] = top * (1.0 - wy) + bot * wy;
        }
    }
    return oy_lo + n_rows;
}

template <int RA, int RB>
__global__ __launch_bounds__(256) void k_pyramid_stream(StreamArgs a) {
    constexpr int RM = RA > RB ? RA : RB;
    constexpr int NL = RB > 0 ? 2 : 1;
    constexpr int K = kStreamK;
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
    // this thread's source column; threads beyond the strip's support (owned columns + RM on the left,
    // RM + 1 on the right) repeat its last column
    const unsigned xcol = (unsigned)mirror_idx(xa - RM + min((int)threadIdx.x, xb - xa + 2 * RM), W);

    // per level: the strip's output columns (those whose left tap lies in [xa, xb)), per 64-column
    // group the lane's left tap as a ring column and its blend weight; per-row terms into LDS
    double wkA[RA + 1], wckA[RA + 1];
    double wkB[RB + 1], wckB[RB + 1];
    int oxA0, ncolsA, ngA, xoffA[kStreamMaxGroups];
    double wxA[kStreamMaxGroups];
    int oxB0 = 0, ncolsB = 0, ngB = 0, xoffB[kStreamMaxGroups];
    double wxB[kStreamMaxGroups];
    double syA, syB = 1.0;
    {
        const StreamLevel &L = a.lv[0];
#pragma unroll
        for (int k = 0; k <= RA; k++) { wkA[k] = L.wr[k]; wckA[k] = L.wc[k]; }
        const double sx = (double)W / (double)L.Wo;
        syA = (double)H / (double)L.Ho;
        oxA0 = first_owned(xa, sx, L.Wo);
        ncolsA = (xb >= W ? L.Wo : first_owned(xb, sx, L.Wo)) - oxA0;
        ngA = (ncolsA + 63) >> 6;
#pragma unroll
        for (int g = 0; g < kStreamMaxGroups; g++) {
            const double cx = ((double)(oxA0 + g * 64 + lane) + 0.5) * sx - 0.5;
            const double fx0 = floor(cx);
            wxA[g] = cx - fx0;
            xoffA[g] = min(max((int)fx0 - (xa - RM), RA), SW - RA - 2);   // clamp: lanes beyond ncols
        }
    }
    if constexpr (NL > 1) {
        const StreamLevel &L = a.lv[1];
#pragma unroll
        for (int k = 0; k <= RB; k++) { wkB[k] = L.wr[k]; wckB[k] = L.wc[k]; }
        const double sx = (double)W / (double)L.Wo;
        syB = (double)H / (double)L.Ho;
        oxB0 = first_owned(xa, sx, L.Wo);
        ncolsB = (xb >= W ? L.Wo : first_owned(xb, sx, L.Wo)) - oxB0;
        ngB = (ncolsB + 63) >> 6;
#pragma unroll
        for (int g = 0; g < kStreamMaxGroups; g++) {
            const double cx = ((double)(oxB0 + g * 64 + lane) + 0.5) * sx - 0.5;
            const double fx0 = floor(cx);
            wxB[g] = cx - fx0;
            xoffB[g] = min(max((int)fx0 - (xa - RM), RB), SW - RB - 2);
        }
    }
    double *dstA = a.lv[0].dst[arr] + (int64_t)pair * a.lv[0].dst_stride;
    double *dstB = NL > 1 ? a.lv[1].dst[arr] + (int64_t)pair * a.lv[1].dst_stride : nullptr;

    // the column's window: w[i] = source row (y - RM + i) for the chunk that starts at V row y
    double w[K + 2 * RM], nxt[K];
#pragma unroll
    for (int i = 0; i < 2 * RM; i++) w[i] = s[(int64_t)mirror_idx(ya + i - RM, H) * W + xcol];
#pragma unroll
    for (int i = 0; i < K; i++) w[2 * RM + i] = s[(int64_t)mirror_idx(ya + RM + i, H) * W + xcol];
    const bool last_seg = yb >= H;
    int nextA = first_owned(ya, syA, a.lv[0].Ho), nextB = NL > 1 ? first_owned(ya, syB, a.lv[1].Ho) : 0, unit = 0;
    const int endA = last_seg ? a.lv[0].Ho : first_owned(yb, syA, a.lv[0].Ho);
    const int endB = NL > 1 ? (last_seg ? a.lv[1].Ho : first_owned(yb, syB, a.lv[1].Ho)) : 0;
    const int y_last = min(yb, H - 1);                            // last V row this block needs
    const int n_chunks = (y_last - ya + K) / K;
    for (int c = 0; c < n_chunks; c++) {
        const int y = ya + c * K;                                 // first V row of this chunk
        if (c + 1 < n_chunks) {                                   // prefetch the next chunk's K rows
            const int r0 = y + K + RM;
            if (r0 + K - 1 <= H - 1) {                            // inside the image: uniform row bases, the column is the offset
#pragma unroll
                for (int i = 0; i < K; i++) nxt[i] = (s + (int64_t)(r0 + i) * W)[xcol];
            } else {
#pragma unroll
                for (int i = 0; i < K; i++) nxt[i] = s[(int64_t)mirror_idx(r0 + i, H) * W + xcol];
            }
        }
        // vertical Gaussians of both levels at V rows y .. y + K - 1 (rows >= H: computed, never read)
#pragma unroll
        for (int j = 0; j < K; j++) {
            const int slot = (c * K + j) % kStreamRing;
#ifdef TDK_STREAM_ABL_NOV
            const double va = w[j + RM];
            double vb = w[j + RM];
#else
            const double va = stream_vtap<RA>(w, j + RM, wkA);
            double vb = 0.0;
            if constexpr (NL > 1) vb = stream_vtap<RB>(w, j + RM, wkB);
#endif
            if ((int)threadIdx.x < SW) {                          // threads beyond the pitch hold a repeated column
                ringA[slot * SW + threadIdx.x] = va;
                if constexpr (NL > 1) ringB[slot * SW + threadIdx.x] = vb;
            }
        }
        __syncthreads();
        // outputs whose lower row tap y0 + 1 is now in the ring: y0 + 1 <= y + K - 1
        const int ynew = y + K - 1 >= y_last ? (1 << 30) : y + K - 1;   // the last chunk emits whatever is left
#ifndef TDK_STREAM_ABL_NOEMIT
        nextA = stream_emit<RA>(a.lv[0], ringA, SW, dstA, syA, nextA, endA, ynew, ya, ngA, ncolsA, oxA0, xoffA, wxA, wckA,
                                wave, lane, unit);
        if constexpr (NL > 1)
            nextB = stream_emit<RB>(a.lv[1], ringB, SW, dstB, syB, nextB, endB, ynew, ya, ngB, ncolsB, oxB0, xoffB, wxB,
                                    wckB, wave, lane, unit);
#endif
#ifndef TDK_STREAM_ABL_NOBAR2
        if (kStreamRing < 2 * K + 1) __syncthreads();             // the next chunk's V rows overwrite rows read above
#endif
#pragma unroll
        for (int i = 0; i < 2 * RM; i++) w[i] = w[K + i];
#pragma unroll
        for (int i = 0; i < K; i++) w[2 * RM + i] = nxt[i];
    }
}

// scipy.ndimage._filters._gaussian_kernel1d (order 0), radius int(4 sigma + 0.5)
void gaussian_weights(double sigma, int radius, double *w) {
    const double sigma2 = sigma * sigma;
    double sum = 0.0;
    for (int i = -radius; i <= radius; i++) {
        w[i + radius] = exp(-0.5 / sigma2 * (double)(i * i));
        sum += w[i + radius];
    }
    for (int i = 0; i <= 2 * radius; i++) w[i] = w[i] / sum;
}

}  // namespace

namespace tdk {

// Anti-aliased variant of launch_pyramid (mode 0 geometry).  `weights` is a device
// buffer of n_out * 2 * (2 kMaxGaussRadius + 1) doubles owned by the caller; the
// kernels of every level are computed here and copied into it.
tdk_status launch_pyramid_aa(const double *const *srcs, int n_arrays, int H, int W, int64_t src_stride, int n_out,
                             const PyramidLevelDesc *levels, int batch, double *weights, bool upload_weights,
                             hipStream_t stream, unsigned skip_mask) {
    if (n_out <= 0) return TDK_OK;
    if (n_out > 15 || n_arrays > 4) {
        set_error("pyramid too deep");
        return TDK_ERR_INVALID_ARGUMENT;
    }
    RescaleAaArgs args;
    RescaleArgs &r = args.r;
    for (int i = 0; i < 4; i++) r.src[i] = i < n_arrays ? srcs[i] : nullptr;
    r.src_stride = src_stride; r.H = H; r.W = W; r.n_out = n_out;
    constexpr int kSlot = 2 * kMaxGaussRadius + 1;
    std::vector<double> host((size_t)n_out * 2 * kSlot, 0.0);
    int blocks = 0;
    for (int l = 0; l < n_out; l++) {
        for (int i = 0; i < 4; i++) r.lv[l].dst[i] = i < n_arrays ? levels[l].dst[i] : nullptr;
        r.lv[l].stride = levels[l].stride; r.lv[l].Ho = levels[l].H; r.lv[l].Wo = levels[l].W;
        blocks += (int)(((int64_t)levels[l].H * levels[l].W + 255) / 256);   // one thread per output pixel
        r.blk_end[l] = blocks;
        // sigma = max(0, (factor - 1) / 2) per axis; sigma 0 = the one-tap kernel {1} (x * 1.0 is exact)
        double sg[2] = {((double)H / (double)levels[l].H - 1.0) / 2.0, ((double)W / (double)levels[l].W - 1.0) / 2.0};
        int R[2];
        for (int ax = 0; ax < 2; ax++) {
            double *w = host.data() + ((size_t)l * 2 + ax) * kSlot;
            if (!(sg[ax] > 1e-15)) {
                R[ax] = 0;
                w[0] = 1.0;
                continue;
            }
            R[ax] = (int)(4.0 * sg[ax] + 0.5);
            if (R[ax] > kMaxGaussRadius) {
                set_error("anti-aliasing kernel radius %d exceeds %d", R[ax], kMaxGaussRadius);
                return TDK_ERR_INVALID_ARGUMENT;
            }
            gaussian_weights(sg[ax], R[ax], w);
        }
        args.aa[l].wr = weights + ((size_t)l * 2 + 0) * kSlot;
        args.aa[l].wc = weights + ((size_t)l * 2 + 1) * kSlot;
        args.aa[l].Rr = R[0];
        args.aa[l].Rc = R[1];
    }
    if (upload_weights) {   // they depend on the shapes only: a batch uploads them once
        TDK_HIP(hipMemcpyAsync(weights, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice, stream));
        TDK_HIP(hipStreamSynchronize(stream));   // `host` goes out of scope
    }
    // The first level (R = 1) or the first two (R = 1, 3: ratio 1.5) of a batch large enough to fill the
    // chip with full-height strips: one streaming pass over the source (k_pyramid_stream).
    // TDK_PYRAMID_STREAM=0 keeps them on the tiles below (bit-identical either way: tested).
    {
        const char *env = getenv("TDK_PYRAMID_STREAM");       // 0: never, 1 (default): large batches, 2: always
        const int use_stream = env ? atoi(env) : 1;
        auto fits = [&](int l, int R) {
            return l < n_out && !((skip_mask >> l) & 1u) && args.aa[l].Rr == R && args.aa[l].Rc == R &&
                   r.lv[l].Ho < H && r.lv[l].Wo < W;
        };
        const int n_strips = (W + 247) / 248, strip_w = (W + n_strips - 1) / n_strips;
        const int64_t images = (int64_t)n_arrays * batch;
        if (use_stream && fits(0, 1) && (images * n_strips >= 256 || use_stream == 2)) {
            const int nl = fits(1, 3) ? 2 : 1;
            StreamArgs sa;
            for (int i = 0; i < 4; i++) sa.src[i] = r.src[i];
            sa.src_stride = src_stride; sa.H = H; sa.W = W; sa.n_arrays = n_arrays; sa.batch = batch;
            sa.n_strips = n_strips; sa.strip_w = strip_w;
            sa.pitch = std::min(256, (strip_w + 2 * 3 + 1 + 7) & ~7);
            // row segments: enough blocks for several full rounds of the 1024 resident ones (a segment pays
            // 2 RM warm-up rows); TDK_STREAM_SEGS overrides
            int n_segs = 1;
            while (images * n_strips * n_segs < 8192 && H / (n_segs * 2) >= 48) n_segs *= 2;
            if (const char *v = getenv("TDK_STREAM_SEGS")) n_segs = std::max(1, atoi(v));
            sa.n_segs = n_segs; sa.seg_rows = (H + n_segs - 1) / n_segs;
            sa.n_segs = (H + sa.seg_rows - 1) / sa.seg_rows;
            size_t lds = 0;
            for (int l = 0; l < 2; l++) {
                const int k = l < nl ? l : 0;
                for (int i = 0; i < 4; i++) sa.lv[l].dst[i] = r.lv[k].dst[i];
                sa.lv[l].dst_stride = r.lv[k].stride; sa.lv[l].Ho = r.lv[k].Ho; sa.lv[l].Wo = r.lv[k].Wo;
                sa.lv[l].wr = args.aa[k].wr; sa.lv[l].wc = args.aa[k].wc;
                if (l < nl) lds += sizeof(double) * kStreamRing * sa.pitch;
            }
            const int64_t blocks = 8 * ((images + 7) / 8) * n_strips * sa.n_segs;
            if (blocks < (1ll << 31) && lds <= 160 * 1024) {
                static bool attr_set = false;
                if (!attr_set) {   // > 64 KiB of dynamic LDS has to be asked for
                    TDK_HIP(hipFuncSetAttribute((const void *)k_pyramid_stream<1, 3>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    TDK_HIP(hipFuncSetAttribute((const void *)k_pyramid_stream<1, 0>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    attr_set = true;
                }
                if (nl == 2) k_pyramid_stream<1, 3><<<(unsigned)blocks, 256, lds, stream>>>(sa);
                else k_pyramid_stream<1, 0><<<(unsigned)blocks, 256, lds, stream>>>(sa);
                TDK_LAUNCH_CHECK();
                skip_mask |= nl == 2 ? 3u : 1u;
            }
        }
    }
    // levels that shrink both axes and whose tiles fit in LDS take the tiled kernel -- all of
    // them in one launch (k_rescale_aa_multi); whatever is left (an enlarged axis, very deep
    // levels) the general one
    bool general = false;
    bool is_tiled[15] = {};
    AaMultiArgs m;
    m.n = 0;
    size_t lds_max = 0;
    int tiles_total = 0;
    for (int l = 0; l < n_out; l++) {
        if ((skip_mask >> l) & 1u) continue;
        const PyrLevel &L = r.lv[l];
        const double fy = (double)H / (double)L.Ho, fx = (double)W / (double)L.Wo;
        AaTileArgs t;
        const int Rk = args.aa[l].Rr == args.aa[l].Rc ? args.aa[l].Rr : 0;
        // output rows per block: 20 for the 3-tap level, 12 from R = 3 on -- the tallest tiles whose
        // V rows still fit the 8-rows-per-wave register walk of aa_tile_fixed (measured on the VGA bench
        // batch, pyramid time with the current kernel: 10/8 rows 1.33 ms, 14/8 1.20, 20/8 1.13, 20/12 1.09;
        // 16 rows per wave and taller tiles are slower -- 30/12 1.09 but level 1 alone 0.65 against
        // 0.57, 20/16 1.35: the LDS footprint leaves fewer blocks per CU)
        t.tile_rows = Rk == 1 ? 20 : 12;
        {   // tuning knobs (experiments): TDK_AA_ROWS_R1 / _R3 / _R5 / _R0
            char name[32];
            snprintf(name, sizeof(name), "TDK_AA_ROWS_R%d", Rk);
            const char *v = getenv(name);
            if (v && atoi(v) > 0) t.tile_rows = atoi(v);
        }
        t.max_v_rows = (int)ceil(t.tile_rows * fy) + 2;
        t.max_cols = (int)ceil(kAaCols * fx) + 2 + 2 * args.aa[l].Rc;
        // V tile, then either the two kernels (generic radius) or the per-row blend terms (fixed radius)
        const size_t lds = sizeof(double) * ((size_t)t.max_v_rows * t.max_cols + 2 * args.aa[l].Rr +
                                             2 * args.aa[l].Rc + 2) + (size_t)t.tile_rows * 12 + 8;
        if (L.Ho > H || L.Wo > W || lds > 64 * 1024 || m.n == kAaMaxFused) {
            general = true;
            continue;
        }
        for (int i = 0; i < 4; i++) { t.src[i] = r.src[i]; t.dst[i] = L.dst[i]; }
        t.src_stride = src_stride; t.dst_stride = L.stride;
        t.H = H; t.W = W; t.Ho = L.Ho; t.Wo = L.Wo;
        t.aa = args.aa[l];
        tiles_total += ((L.Ho + t.tile_rows - 1) / t.tile_rows) * ((L.Wo + kAaCols - 1) / kAaCols);
        m.lv[m.n] = t;
        m.tile_end[m.n] = tiles_total;
        m.n++;
        is_tiled[l] = true;
        if (lds > lds_max) lds_max = lds;
    }
    if (m.n > 0) {
        m.n_arrays = n_arrays;
        m.batch = batch;
        const int64_t images = (int64_t)n_arrays * batch;
        const int64_t blocks = (images < 8 ? images : 8 * ((images + 7) / 8)) * tiles_total;
        if (blocks >= (1ll << 31)) {
            set_error("anti-aliased pyramid: %lld blocks exceed the grid limit", (long long)blocks);
            return TDK_ERR_INVALID_ARGUMENT;
        }
        k_rescale_aa_multi<<<(unsigned)blocks, 256, lds_max, stream>>>(m);
        TDK_LAUNCH_CHECK();
    }
    if (general) {
        // recompute the cumulative block counts over the levels that are left
        int total = 0;
        for (int l = 0; l < n_out; l++) {
            const PyrLevel &L = r.lv[l];
            if (!is_tiled[l] && !((skip_mask >> l) & 1u)) total += (int)(((int64_t)L.Ho * L.Wo + 255) / 256);
            r.blk_end[l] = total;
        }
        dim3 grid(total, n_arrays, batch);
        k_rescale_levels_aa<<<grid, 256, 0, stream>>>(args);
        TDK_LAUNCH_CHECK();
    }
    return TDK_OK;
}

size_t pyramid_aa_weight_doubles(int n_out) { return (size_t)n_out * 2 * (2 * kMaxGaussRadius + 1); }

// Used by dvo.hip to build pyramid levels of device-resident batches; lives in
// this translation unit so that the pyramid arithmetic is contraction-free.
tdk_status launch_rescale(const double *src, int H, int W, double *dst, int Ho, int Wo, int batch,
                          int64_t src_stride, int64_t dst_stride, hipStream_t stream) {
    dim3 grid(grid_for((int64_t)Ho * Wo), batch);
    k_rescale<<<grid, 256, 0, stream>>>(src, H, W, dst, Ho, Wo, src_stride, dst_stride);
    TDK_LAUNCH_CHECK();
    return TDK_OK;
}

// All pyramid levels of all arrays of a batch in one launch.  srcs/dsts hold
// n_arrays device pointers per level (level-major for dsts).
tdk_status launch_pyramid(const double *const *srcs, int n_arrays, int H, int W, int64_t src_stride,
                          int n_out, const PyramidLevelDesc *levels, int batch, int mode, hipStream_t stream) {
    if (n_out <= 0) return TDK_OK;
    if (n_out > 15 || n_arrays > 4) {
        set_error("pyramid too deep");
        return TDK_ERR_INVALID_ARGUMENT;
    }
    if (mode == 0) {
        RescaleArgs r;
        for (int i = 0; i < 4; i++) r.src[i] = i < n_arrays ? srcs[i] : nullptr;
        r.src_stride = src_stride; r.H = H; r.W = W; r.n_out = n_out;
        int blocks = 0;
        for (int l = 0; l < n_out; l++) {
            for (int i = 0; i < 4; i++) r.lv[l].dst[i] = i < n_arrays ? levels[l].dst[i] : nullptr;
            r.lv[l].stride = levels[l].stride; r.lv[l].Ho = levels[l].H; r.lv[l].Wo = levels[l].W;
            blocks += (levels[l].H + 3) / 4;   // four output rows per block
            r.blk_end[l] = blocks;
        }
        dim3 grid(blocks, n_arrays, batch);
        k_rescale_levels<<<grid, 256, 0, stream>>>(r);
        TDK_LAUNCH_CHECK();
        return TDK_OK;
    }
    PyrArgs a;
    for (int i = 0; i < 4; i++) a.src[i] = i < n_arrays ? srcs[i] : nullptr;
    a.src_stride = src_stride; a.H = H; a.W = W; a.n_arrays = n_arrays; a.n_out = n_out;
    for (int l = 0; l < n_out; l++) {
        for (int i = 0; i < 4; i++) a.lv[l].dst[i] = i < n_arrays ? levels[l].dst[i] : nullptr;
        a.lv[l].stride = levels[l].stride; a.lv[l].Ho = levels[l].H; a.lv[l].Wo = levels[l].W;
        // the tile partition assumes a downscale (sample step >= 1 texel)
        if (levels[l].H > H || levels[l].W > W) {
            set_error("pyramid levels must not be larger than level 0");
            return TDK_ERR_INVALID_ARGUMENT;
        }
    }
    dim3 grid((W + kPyrTW - 1) / kPyrTW, (H + kPyrTH - 1) / kPyrTH, batch * n_arrays);
    k_pyramid<<<grid, 256, 0, stream>>>(a);
    TDK_LAUNCH_CHECK();
    return TDK_OK;
}

}  // namespace tdk

namespace {

// H2D of `bytes` into scratch slot, returns the device pointer.
tdk_status to_device(int slot, const void *host, size_t bytes, void **dev) {
    TDK_TRY(tdk::scratch(slot, bytes, dev));
    if (bytes) TDK_HIP(hipMemcpyAsync(*dev, host, bytes, hipMemcpyHostToDevice, tdk::stream()));
    return TDK_OK;
}

tdk_status to_host(void *host, const void *dev, size_t bytes) {
    if (bytes) TDK_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, tdk::stream()));
    TDK_HIP(hipStreamSynchronize(tdk::stream()));
    return TDK_OK;
}

Cam cam_of(const double *c) { return Cam{c[0], c[1], c[2], c[3]}; }

tdk_status normalize_impl(const double *kp, int64_t n, const double *camera, double *out, int inverse) {
    TDK_REQUIRE(n >= 0 && camera && (n == 0 || (kp && out)), "null pointer");
    if (n == 0) return tdk::ensure_device();
    void *d_in, *d_out;
    TDK_TRY(to_device(0, kp, (size_t)n * 16, &d_in));
    TDK_TRY(tdk::scratch(1, (size_t)n * 16, &d_out));
    k_normalize<<<grid_for(n), kBlock, 0, tdk::stream()>>>((const double2 *)d_in, n, cam_of(camera),
                                                           (double2 *)d_out, inverse);
    TDK_LAUNCH_CHECK();
    return to_host(out, d_out, (size_t)n * 16);
}

// skimage.color.rgb2gray as the examples use it (examples/dvo_pose_change.py:22-31):
// luma of the first three interleaved channels; uint8 input is scaled by 1/255 first
// (img_as_float).  -ffp-contract=off keeps the two products and two sums as written.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_rgb2gray(const T *__restrict__ rgb, int64_t n, int channels,
                                                     double *__restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const T *p = rgb + i * channels;
        double r, g, b;
        if (sizeof(T) == 1) { r = (double)p[0] / 255.0; g = (double)p[1] / 255.0; b = (double)p[2] / 255.0; }
        else { r = (double)p[0]; g = (double)p[1]; b = (double)p[2]; }
        out[i] = (0.2125 * r + 0.7154 * g) + 0.0721 * b;
    }
}

}  // namespace

extern "C" {

tdk_status tdk_normalize(const double *kp, int64_t n, const double *camera, double *out) {
    return normalize_impl(kp, n, camera, out, 0);
}

tdk_status tdk_unnormalize(const double *kp, int64_t n, const double *camera, double *out) {
    return normalize_impl(kp, n, camera, out, 1);
}

tdk_status tdk_project_vecs(const double *points, int64_t n, double *out) {
    TDK_REQUIRE(n >= 0 && (n == 0 || (points && out)), "null pointer");
    if (n == 0) return tdk::ensure_device();
    void *d_in, *d_out;
    TDK_TRY(to_device(0, points, (size_t)n * 24, &d_in));
    TDK_TRY(tdk::scratch(1, (size_t)n * 16, &d_out));
    k_project<<<grid_for(n), kBlock, 0, tdk::stream()>>>((const double *)d_in, n, (double2 *)d_out);
    TDK_LAUNCH_CHECK();
    return to_host(out, d_out, (size_t)n * 16);
}

tdk_status tdk_inv_project_vecs(const double *xs, const double *depths, int64_t n, double *out) {
    TDK_REQUIRE(n >= 0 && (n == 0 || (xs && depths && out)), "null pointer");
    if (n == 0) return tdk::ensure_device();
    void *d_xs, *d_d, *d_out;
    TDK_TRY(to_device(0, xs, (size_t)n * 16, &d_xs));
    TDK_TRY(to_device(1, depths, (size_t)n * 8, &d_d));
    TDK_TRY(tdk::scratch(2, (size_t)n * 24, &d_out));
    k_inv_project<<<grid_for(n), kBlock, 0, tdk::stream()>>>((const double2 *)d_xs, (const double *)d_d,
                                                             n, (double *)d_out);
    TDK_LAUNCH_CHECK();
    return to_host(out, d_out, (size_t)n * 24);
}

tdk_status tdk_transform(const double *T, const double *points, int64_t n, double *out) {
    TDK_REQUIRE(n >= 0 && T && (n == 0 || (points && out)), "null pointer");
    if (n == 0) return tdk::ensure_device();
    Mat4 M;
    for (int i = 0; i < 16; i++) M.m[i] = T[i];
    void *d_in, *d_out;
    TDK_TRY(to_device(0, points, (size_t)n * 24, &d_in));
    TDK_TRY(tdk::scratch(1, (size_t)n * 24, &d_out));
    k_transform<<<grid_for(n), kBlock, 0, tdk::stream()>>>(M, (const double *)d_in, n, (double *)d_out);
    TDK_LAUNCH_CHECK();
    return to_host(out, d_out, (size_t)n * 24);
}

tdk_status tdk_warp_vecs(const double *T10, const double *xs, const double *depths, int64_t n,
                         double *out_xs, double *out_depths) {
    TDK_REQUIRE(n >= 0 && T10 && (n == 0 || (xs && depths && out_xs && out_depths)), "null pointer");
    if (n == 0) return tdk::ensure_device();
    Mat4 M;
    for (int i = 0; i < 16; i++) M.m[i] = T10[i];
    void *d_xs, *d_d, *d_oxs, *d_od;
    TDK_TRY(to_device(0, xs, (size_t)n * 16, &d_xs));
    TDK_TRY(to_device(1, depths, (size_t)n * 8, &d_d));
    TDK_TRY(tdk::scratch(2, (size_t)n * 16, &d_oxs));
    TDK_TRY(tdk::scratch(3, (size_t)n * 8, &d_od));
    k_warp<<<grid_for(n), kBlock, 0, tdk::stream()>>>(M, (const double2 *)d_xs, (const double *)d_d, n,
                                                      (double2 *)d_oxs, (double *)d_od);
    TDK_LAUNCH_CHECK();
    TDK_HIP(hipMemcpyAsync(out_xs, d_oxs, (size_t)n * 16, hipMemcpyDeviceToHost, tdk::stream()));
    return to_host(out_depths, d_od, (size_t)n * 8);
}

tdk_status tdk_interpolation(const double *image, int H, int W, const double *coordinates, int64_t m,
                             double *out) {
    TDK_REQUIRE(H > 0 && W > 0 && m >= 0 && image && (m == 0 || (coordinates && out)), "bad argument");
    if (m == 0) return tdk::ensure_device();
    void *d_img, *d_c, *d_out, *d_bad;
    TDK_TRY(to_device(0, image, (size_t)H * W * 8, &d_img));
    TDK_TRY(to_device(1, coordinates, (size_t)m * 16, &d_c));
    TDK_TRY(tdk::scratch(2, (size_t)m * 8, &d_out));
    TDK_TRY(tdk::scratch(3, sizeof(int), &d_bad));
    TDK_HIP(hipMemsetAsync(d_bad, 0, sizeof(int), tdk::stream()));
    k_interpolation<<<grid_for(m), kBlock, 0, tdk::stream()>>>((const double *)d_img, H, W,
                                                               (const double2 *)d_c, m, (double *)d_out,
                                                               (int *)d_bad);
    TDK_LAUNCH_CHECK();
    int bad = 0;
    TDK_HIP(hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, tdk::stream()));
    TDK_TRY(to_host(out, d_out, (size_t)m * 8));
    if (bad) {
        tdk::set_error("coordinates out of image range");
        return TDK_ERR_OUT_OF_RANGE;
    }
    return TDK_OK;
}

tdk_status tdk_calc_depth0(const double *T10, const double *x0, const double *x1, double *depth) {
    TDK_REQUIRE(T10 && x0 && x1 && depth, "null pointer");
    *depth = tdk::calc_depth0(T10, x0[0], x0[1], x1[0], x1[1]);
    return TDK_OK;
}

tdk_status tdk_image_gradient(const double *image, int H, int W, double *gx, double *gy) {
    TDK_REQUIRE(H > 0 && W > 0 && image && gx && gy, "bad argument");
    size_t bytes = (size_t)H * W * 8;
    void *d_img, *d_gx, *d_gy;
    TDK_TRY(to_device(0, image, bytes, &d_img));
    TDK_TRY(tdk::scratch(1, bytes, &d_gx));
    TDK_TRY(tdk::scratch(2, bytes, &d_gy));
    k_image_gradient<<<grid_for((int64_t)H * W), kBlock, 0, tdk::stream()>>>((const double *)d_img, H, W,
                                                                             (double *)d_gx, (double *)d_gy);
    TDK_LAUNCH_CHECK();
    TDK_HIP(hipMemcpyAsync(gx, d_gx, bytes, hipMemcpyDeviceToHost, tdk::stream()));
    return to_host(gy, d_gy, bytes);
}

tdk_status tdk_rescale(const double *image, int H, int W, double *out, int Ho, int Wo) {
    TDK_REQUIRE(H > 0 && W > 0 && Ho > 0 && Wo > 0 && image && out, "bad argument");
    void *d_img, *d_out;
    TDK_TRY(to_device(0, image, (size_t)H * W * 8, &d_img));
    TDK_TRY(tdk::scratch(1, (size_t)Ho * Wo * 8, &d_out));
    TDK_TRY(tdk::launch_rescale((const double *)d_img, H, W, (double *)d_out, Ho, Wo, 1, 0, 0, tdk::stream()));
    return to_host(out, d_out, (size_t)Ho * Wo * 8);
}

tdk_status tdk_rescale_anti_aliased(const double *image, int H, int W, double *out, int Ho, int Wo) {
    TDK_REQUIRE(H > 0 && W > 0 && Ho > 0 && Wo > 0 && image && out, "bad argument");
    void *d_img, *d_out, *d_w;
    TDK_TRY(to_device(0, image, (size_t)H * W * 8, &d_img));
    TDK_TRY(tdk::scratch(1, (size_t)Ho * Wo * 8, &d_out));
    TDK_TRY(tdk::scratch(2, tdk::pyramid_aa_weight_doubles(1) * 8, &d_w));
    const double *srcs[1] = {(const double *)d_img};
    tdk::PyramidLevelDesc lv;
    lv.dst[0] = (double *)d_out; lv.dst[1] = lv.dst[2] = lv.dst[3] = nullptr;
    lv.stride = 0; lv.H = Ho; lv.W = Wo;
    TDK_TRY(tdk::launch_pyramid_aa(srcs, 1, H, W, 0, 1, &lv, 1, (double *)d_w, true, tdk::stream()));
    return to_host(out, d_out, (size_t)Ho * Wo * 8);
}

tdk_status tdk_rgb2gray(const double *rgb, int H, int W, int channels, double *gray) {
    TDK_REQUIRE(H > 0 && W > 0 && channels >= 3 && channels <= 4 && rgb && gray, "bad argument");
    const int64_t n = (int64_t)H * W;
    void *d_rgb, *d_out;
    TDK_TRY(to_device(0, rgb, (size_t)n * channels * 8, &d_rgb));
    TDK_TRY(tdk::scratch(1, (size_t)n * 8, &d_out));
    k_rgb2gray<double><<<grid_for(n), kBlock, 0, tdk::stream()>>>((const double *)d_rgb, n, channels, (double *)d_out);
    TDK_LAUNCH_CHECK();
    return to_host(gray, d_out, (size_t)n * 8);
}

tdk_status tdk_rgb2gray_u8(const uint8_t *rgb, int H, int W, int channels, double *gray) {
    TDK_REQUIRE(H > 0 && W > 0 && channels >= 3 && channels <= 4 && rgb && gray, "bad argument");
    const int64_t n = (int64_t)H * W;
    void *d_rgb, *d_out;
    TDK_TRY(to_device(0, rgb, (size_t)n * channels, &d_rgb));
    TDK_TRY(tdk::scratch(1, (size_t)n * 8, &d_out));
    k_rgb2gray<uint8_t><<<grid_for(n), kBlock, 0, tdk::stream()>>>((const uint8_t *)d_rgb, n, channels, (double *)d_out);
    TDK_LAUNCH_CHECK();
    return to_host(gray, d_out, (size_t)n * 8);
}

}  // extern "C"
