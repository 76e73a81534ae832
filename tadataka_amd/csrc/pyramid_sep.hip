// pyramid_sep.hip -- the anti-aliased pyramid levels as ONE separable resampling filter per axis.
//
// skimage.transform.rescale(image, scale) (tadataka/vo/dvo/__init__.py:144-148; anti-aliasing is its
// default when it shrinks) is a Gaussian prefilter, sigma = (factor - 1) / 2 per axis, followed by a
// bilinear warp.  granular.hip evaluates it in scipy.ndimage's operation order (vertical correlate1d,
// horizontal correlate1d, bilinear blend): bit-identical with the CPU restatement, but ~(2R + 1) taps
// per Gaussian per bilinear tap, 120 M FP64 instructions and 92 M LDS bank conflicts per 256-pair batch.
//
// Both steps are linear and separable, so an output pixel is
//
//     out[oy][ox] = sum_r cv[oy][r] * ( sum_c ch[ox][c] * src[r][c] )
//
// with ONE list of at most T = 2R + 2 (row, weight) pairs per output row and one per output column: the
// two bilinear taps of an axis are neighbours, their Gaussians overlap in all but one position, mirror /
// reflect boundaries only fold indices together.  The lists depend on the shapes alone and are built
// once per batch on the host (build_axis), in double; the kernel is two short FMA chains:
//
//   1. vertical pass at the OUTPUT rows only:  V[j][c] = sum_k cv[j][k] * src[row_k][c]       (T FMAs),
//      the T source rows as independent coalesced loads (L1 / L2 serve the re-reads of neighbouring
//      output rows), V to LDS,
//   2. horizontal pass: out[j][i] = sum_k ch[i][k] * V[j][col_k]                              (T FMAs).
// (Two other formulations were built and measured on the 256-pair VGA batch, two arrays, both slower:
// staging the source tile in LDS first -- 1.38 ms, every block a chain of load -> barrier -> pass ->
// barrier -> pass latencies; and blocks that walk down a strip of columns with the source rows in an
// LDS ring, prefetched one step ahead -- 1.06 ms, 4-6 us per 8-row step, still latency-bound.)
//
// Per output pixel that is factor * T + T FMAs (10 at ratio 1.5 level 1, 26 at level 2) instead of
// ~80 / ~200 operations, and no arithmetic at source resolution at all.  The result differs from the
// ndimage operation order in the last bits only (different association of the same products; measured
// <= 2e-15 absolute on [0, 3] data).
//
// STATUS: opt-in (tdk_dvo_set_anti_aliasing(h, 3)), NOT the default -- it lost.  Measured on the bench
// batch (256 pairs 640x480, levels 1 + 2, MI355X; tools/kbench_pyr.py, rocprofv3 kernel trace):
//     ndimage-order tiles (granular.hip), 3 arrays             1.10 ms   (0.37 ms per array)
//     this kernel, I0 + I1, + ndimage order for D0             1.39 ms   (0.52 ms per array + 0.35)
//   ablation of this kernel (2 arrays): no loads 0.40 ms, no stores 0.92, neither 0.35 -- a third of its
//   time is per-block fixed cost (118 000 blocks of argument / tap-table fetches, one barrier each), two
//   thirds the vertical pass's loads: every source row is fetched by T / factor = 2.7 - 3.6 output rows
//   and L1 does not hold the working set of eight resident blocks, so they come from L2 again.
//   Issuing more of them at once made it worse (48 loads in flight per wave: 2.43 ms).
//   Two restructurings reached the same 1.05 ms for two arrays: the source tile staged in LDS first
//   (load -> barrier -> pass -> barrier -> pass: latency chains, 4 blocks per CU), and blocks walking down a
//   strip with the source rows in an LDS ring prefetched a step ahead (4 - 6 us per 8-row step).
//   The ndimage-order kernel wins because its vertical pass keeps a column's source texels in registers
//   for a whole tile (each texel fetched once per tile) -- which needs the window offsets at compile time,
//   i.e. V at SOURCE rows; folding the bilinear blend into the vertical taps gives that up.
// And it cannot take the DEPTH map in any case: at the identity prior the right / bottom border of a
// level projects exactly onto the inclusive mask boundary, the last bit of D0 decides on which side a
// border pixel falls (a few hundred pixels; 2e-5 in the pose of a 120x160 pair) -- so D0 has to be
// bit-identical with the CPU restatement for poses to be reproducible, and mode 3 keeps D0 on the
// ndimage-order kernels.
//
// Compiled with the default -ffp-contract (FMA).
#include "tdk_runtime.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <vector>

namespace {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / 64;
constexpr int kMaxSepLevels = 4;        // levels fused into one launch
constexpr int kTileCols = 64;           // output columns of a tile (one per lane in the horizontal pass)
constexpr int kColChunks = 3;           // a tile's source columns: at most 64 * kColChunks
constexpr int kTileRows = 16;           // output rows of a tile (default; TDK_SEP_ROWS)

struct SepLevel {
    const double *src[4];
    double *dst[4];
    int64_t src_stride, dst_stride;
    int H, W, Ho, Wo;
    int Tv, Th;                  // taps per output row / column
    int TR;                      // output rows per tile
    int tiles_c;                 // tiles per row of tiles
    int pitch;                   // LDS pitch (doubles) of V: max source columns of a tile, odd
    const int *vidx;             // [Ho][Tv] source row of every vertical tap (reflected / mirrored already)
    const double *vw;            // [Ho][Tv] its weight
    const int *hidx;             // [Wo][Th] source column of every horizontal tap
    const double *hw;            // [Wo][Th] its weight
    const int *hrange;           // [tiles_c][2] first / last source column of a tile
};

struct SepArgs {
    int n, n_arrays, batch;
    int tile_end[kMaxSepLevels];
    SepLevel lv[kMaxSepLevels];
};

// One block produces TR x kTileCols outputs of one level of one image in two passes:
//
//     vertical    V[j][c]  = sum_k vw[j][k] * src[row_k][c]      for the tile's OUTPUT rows j and the source
//                            columns c its outputs touch: every lane owns up to three columns, the T rows
//                            of a tap list are T independent, coalesced 512-byte loads straight from global
//                            memory (neighbouring output rows re-read them from L1 / L2; nothing is staged,
//                            so all loads of a thread are in flight together), result to LDS
//     barrier
//     horizontal  out[j][i] = sum_k hw[i][k] * V[j][col_k]       one output column per lane, its taps in
//                            registers (fetched before the vertical pass, their latency hides under it)
//
// TV / TH > 0: compile-time tap counts (unrolled); 0: from the arguments.
template <int TV, int TH>
__device__ __forceinline__ void sep_tile(const SepLevel &a, int tile, int arr, int pair, double *V) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tr = tile / a.tiles_c, tc = tile - tr * a.tiles_c;
    const int j0 = tr * a.TR, i0 = tc * kTileCols;
    const int nj = min(a.TR, a.Ho - j0), ni = min(kTileCols, a.Wo - i0);
    const int cmin = a.hrange[2 * tc], cmax = a.hrange[2 * tc + 1];
    const int ncols = cmax - cmin + 1;
    const int pitch = a.pitch;
    const int Tv = TV > 0 ? TV : a.Tv, Th = TH > 0 ? TH : a.Th;
    const double *__restrict__ src = a.src[arr] + (int64_t)pair * a.src_stride + cmin;
    double *__restrict__ dst = a.dst[arr] + (int64_t)pair * a.dst_stride;

    // horizontal taps of this lane's output column
    const bool col_live = lane < ni;
    const int hb = (i0 + (col_live ? lane : 0)) * Th;
    double hwr[TH > 0 ? TH : 1];
    int hoff[TH > 0 ? TH : 1];
    if (TH > 0) {
#pragma unroll
        for (int k = 0; k < TH; k++) { hwr[k] = a.hw[hb + k]; hoff[k] = a.hidx[hb + k] - cmin; }
    }

    int cc[kColChunks];          // clamped: loads are never predicated, the LDS writes are
#pragma unroll
    for (int q = 0; q < kColChunks; q++) cc[q] = min(lane + 64 * q, ncols - 1);
    const int nq = (ncols + 63) >> 6;   // wave-uniform

    for (int j = wave; j < nj; j += kWavesPerBlock) {
        const int tb = (j0 + j) * Tv;    // wave-uniform: rows and weights are scalar loads
        if (TV > 0) {
            const double *rows[TV > 0 ? TV : 1];
            double w[TV > 0 ? TV : 1];
#pragma unroll
            for (int k = 0; k < TV; k++) { rows[k] = src + (int64_t)a.vidx[tb + k] * a.W; w[k] = a.vw[tb + k]; }
#pragma unroll
            for (int q = 0; q < kColChunks; q++) {
                if (q < nq) {
                    double t[TV > 0 ? TV : 1];
#pragma unroll
                    for (int k = 0; k < TV; k++) t[k] = rows[k][cc[q]];
                    double acc = 0.0;
#pragma unroll
                    for (int k = 0; k < TV; k++) acc = __builtin_fma(w[k], t[k], acc);
                    if (lane + 64 * q < ncols) V[j * pitch + lane + 64 * q] = acc;
                }
            }
        } else {
            for (int q = 0; q < nq; q++) {
                double acc = 0.0;
                for (int k = 0; k < Tv; k++) acc = __builtin_fma(a.vw[tb + k], src[(int64_t)a.vidx[tb + k] * a.W + cc[q < kColChunks ? q : 0]], acc);
                if (lane + 64 * q < ncols) V[j * pitch + lane + 64 * q] = acc;
            }
        }
    }
    __syncthreads();
    if (!col_live) return;
    for (int j = wave; j < nj; j += kWavesPerBlock) {
        const double *__restrict__ v = V + j * pitch;
        double acc = 0.0;
        if (TH > 0) {
#pragma unroll
            for (int k = 0; k < TH; k++) acc = __builtin_fma(hwr[k], v[hoff[k]], acc);
        } else {
            for (int k = 0; k < Th; k++) acc = __builtin_fma(a.hw[hb + k], v[a.hidx[hb + k] - cmin], acc);
        }
        dst[(int64_t)(j0 + j) * a.Wo + (i0 + lane)] = acc;
    }
}

__global__ __launch_bounds__(kBlock) void k_pyramid_sep(SepArgs m) {
    extern __shared__ __attribute__((aligned(16))) double sep_smem[];
    // 1-D grid, XCD-major (as k_dvo_eval / k_rescale_aa_multi): XCD k takes images k, k + 8, ... one after
    // the other and all tiles of all levels of an image consecutively, so the rows neighbouring tiles
    // share and the coarser level's pass over the same source are hits in that XCD's L2.
    const int tiles_total = m.tile_end[m.n - 1];
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int image = (q / tiles_total) * 8 + xcd, t = q - (q / tiles_total) * tiles_total;
    if (image >= m.n_arrays * m.batch) return;
    const int pair = image / m.n_arrays, arr = image - pair * m.n_arrays;
    int l = 0;
    while (l + 1 < m.n && t >= m.tile_end[l]) l++;
    const int tile = t - (l ? m.tile_end[l - 1] : 0);
    const SepLevel &a = m.lv[l];
    if (a.Tv == 4 && a.Th == 4) sep_tile<4, 4>(a, tile, arr, pair, sep_smem);        // ratio 1.5, level 1 (R = 1)
    else if (a.Tv == 8 && a.Th == 8) sep_tile<8, 8>(a, tile, arr, pair, sep_smem);   // level 2 (R = 3)
    else sep_tile<0, 0>(a, tile, arr, pair, sep_smem);
}

// ---- host: the per-axis tap lists ------------------------------------------------------------------

// numpy / skimage warp boundary 'reflect' (d c b a | a b c d | d c b a) for the bilinear taps
int reflect_idx_h(int64_t i, int n) {
    if (n == 1) return 0;
    int64_t p = 2 * (int64_t)n;
    i %= p;
    if (i < 0) i += p;
    if (i >= n) i = p - 1 - i;
    return (int)i;
}

// ndimage 'mirror' (d c b | a b c d | c b a) for the Gaussian taps
int mirror_idx_h(int i, int n) {
    if ((unsigned)i < (unsigned)n) return i;
    if (n == 1) return 0;
    const int p = 2 * (n - 1);
    i %= p;
    if (i < 0) i += p;
    if (i >= n) i = p - i;
    return i;
}

struct Axis {
    int T = 0;
    std::vector<int> idx;      // [n_out][T]
    std::vector<double> w;     // [n_out][T]
};

// scipy.ndimage._filters._gaussian_kernel1d (order 0), radius int(4 sigma + 0.5); sigma 0 -> {1}
std::vector<double> gaussian(double sigma, int *radius) {
    if (!(sigma > 1e-15)) { *radius = 0; return {1.0}; }
    const int R = (int)(4.0 * sigma + 0.5);
    std::vector<double> w((size_t)2 * R + 1);
    double sum = 0.0;
    for (int i = -R; i <= R; i++) { w[(size_t)(i + R)] = exp(-0.5 / (sigma * sigma) * (double)(i * i)); sum += w[(size_t)(i + R)]; }
    for (double &v : w) v = v / sum;
    *radius = R;
    return w;
}

Axis build_axis(int n_in, int n_out) {
    const double factor = (double)n_in / (double)n_out;
    int R;
    const std::vector<double> g = gaussian((factor - 1.0) / 2.0, &R);
    Axis ax;
    ax.T = 2 * R + 2;
    ax.idx.assign((size_t)n_out * ax.T, 0);
    ax.w.assign((size_t)n_out * ax.T, 0.0);
    for (int o = 0; o < n_out; o++) {
        const double c = ((double)o + 0.5) * factor - 0.5;
        const double f0 = floor(c), wt = c - f0;
        const int i0 = (int)f0;
        const int a[2] = {reflect_idx_h(i0, n_in), reflect_idx_h((int64_t)i0 + 1, n_in)};
        const double bw[2] = {1.0 - wt, wt};
        std::map<int, double> taps;
        for (int s = 0; s < 2; s++)
            for (int j = -R; j <= R; j++) taps[mirror_idx_h(a[s] + j, n_in)] += bw[s] * g[(size_t)(j + R)];
        // at most 2R + 2 distinct indices: the two bilinear taps are neighbours (or coincide at a border)
        int k = 0;
        for (const auto &kv : taps) {
            if (k >= ax.T) return Axis();     // cannot happen; an empty axis makes the caller fall back
            ax.idx[(size_t)o * ax.T + k] = kv.first;
            ax.w[(size_t)o * ax.T + k] = kv.second;
            k++;
        }
        for (; k < ax.T; k++) ax.idx[(size_t)o * ax.T + k] = taps.begin()->first;   // weight 0, loadable index
    }
    return ax;
}

}  // namespace

namespace tdk {

struct PyramidSepPlan {
    int H, W, n_out;
    std::vector<int> Ho, Wo;
    unsigned handled_mask;          // levels this plan builds (bit l = level index l of the `levels` array)
    void *d_tables;                 // one device allocation holding every table
    SepArgs args;                   // pointers into d_tables; src / dst filled per launch
    int level_of_slot[kMaxSepLevels];
    size_t lds_bytes;
};

tdk_status pyramid_sep_destroy(PyramidSepPlan *p) {
    if (!p) return TDK_OK;
    if (p->d_tables) (void)hipFree(p->d_tables);
    delete p;
    return TDK_OK;
}

// Plans the separable kernel for every level that shrinks both axes and whose tiles fit in LDS (the first
// kMaxSepLevels of them); *handled_mask tells the caller which levels are left for the ndimage-order kernels.
tdk_status pyramid_sep_create(int H, int W, int n_out, const int *Ho, const int *Wo, hipStream_t stream,
                              PyramidSepPlan **out, unsigned *handled_mask) {
    PyramidSepPlan *p = new PyramidSepPlan();
    p->H = H; p->W = W; p->n_out = n_out;
    p->Ho.assign(Ho, Ho + n_out); p->Wo.assign(Wo, Wo + n_out);
    p->handled_mask = 0; p->d_tables = nullptr; p->lds_bytes = 0;
    memset(&p->args, 0, sizeof(p->args));
    std::vector<unsigned char> blob;
    struct Off { size_t vidx, hidx, vw, hw, vrange, hrange; } offs[kMaxSepLevels];
    auto append = [&blob](const void *data, size_t bytes) {
        size_t at = (blob.size() + 15) & ~(size_t)15;
        blob.resize(at + bytes);
        memcpy(blob.data() + at, data, bytes);
        return at;
    };
    int tiles_total = 0;
    SepArgs &m = p->args;
    // tuning knob (experiments): TDK_SEP_ROWS = output rows per tile
    static const int rows_env = [] { const char *v = getenv("TDK_SEP_ROWS"); return v ? atoi(v) : 0; }();
    for (int l = 0; l < n_out && m.n < kMaxSepLevels; l++) {
        if (Ho[l] > H || Wo[l] > W || Ho[l] < 1 || Wo[l] < 1) continue;
        Axis av = build_axis(H, Ho[l]), ah = build_axis(W, Wo[l]);
        if (av.T == 0 || ah.T == 0) continue;
        const int tiles_c = (Wo[l] + kTileCols - 1) / kTileCols;
        std::vector<int> hr((size_t)tiles_c * 2);
        int max_cols = 0;
        for (int t = 0; t < tiles_c; t++) {
            int lo = W, hi = -1;
            for (int o = t * kTileCols; o < std::min(Wo[l], (t + 1) * kTileCols); o++)
                for (int k = 0; k < ah.T; k++) { lo = std::min(lo, ah.idx[(size_t)o * ah.T + k]); hi = std::max(hi, ah.idx[(size_t)o * ah.T + k]); }
            hr[(size_t)2 * t] = lo; hr[(size_t)2 * t + 1] = hi;
            max_cols = std::max(max_cols, hi - lo + 1);
        }
        if (max_cols > 64 * kColChunks) continue;     // deep levels: left to the ndimage-order kernels
        const int TR = rows_env > 0 ? rows_env : kTileRows;
        SepLevel &L = m.lv[m.n];
        L.H = H; L.W = W; L.Ho = Ho[l]; L.Wo = Wo[l];
        L.Tv = av.T; L.Th = ah.T; L.TR = TR; L.tiles_c = tiles_c;
        L.pitch = max_cols | 1;
        Off &o = offs[m.n];
        o.vidx = append(av.idx.data(), av.idx.size() * sizeof(int));
        o.hidx = append(ah.idx.data(), ah.idx.size() * sizeof(int));
        o.vw = append(av.w.data(), av.w.size() * sizeof(double));
        o.hw = append(ah.w.data(), ah.w.size() * sizeof(double));
        o.hrange = append(hr.data(), hr.size() * sizeof(int));
        o.vrange = 0;
        tiles_total += ((Ho[l] + TR - 1) / TR) * tiles_c;
        m.tile_end[m.n] = tiles_total;
        p->level_of_slot[m.n] = l;
        p->handled_mask |= 1u << l;
        p->lds_bytes = std::max(p->lds_bytes, sizeof(double) * (size_t)L.pitch * (size_t)TR);
        m.n++;
    }
    if (m.n > 0) {
        if (hipMalloc(&p->d_tables, blob.size()) != hipSuccess) {
            delete p;
            set_error("hipMalloc of the pyramid tap tables failed");
            return TDK_ERR_HIP;
        }
        hipError_t e = hipMemcpyAsync(p->d_tables, blob.data(), blob.size(), hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);   // `blob` goes out of scope
        if (e != hipSuccess) {
            (void)hipFree(p->d_tables);
            delete p;
            set_error("upload of the pyramid tap tables failed: %s", hipGetErrorString(e));
            return TDK_ERR_HIP;
        }
        const unsigned char *base = (const unsigned char *)p->d_tables;
        for (int s = 0; s < m.n; s++) {
            SepLevel &L = m.lv[s];
            L.vidx = (const int *)(base + offs[s].vidx); L.hidx = (const int *)(base + offs[s].hidx);
            L.vw = (const double *)(base + offs[s].vw); L.hw = (const double *)(base + offs[s].hw);
            L.hrange = (const int *)(base + offs[s].hrange);
        }
    }
    *out = p;
    if (handled_mask) *handled_mask = p->handled_mask;
    return TDK_OK;
}

bool pyramid_sep_matches(const PyramidSepPlan *p, int H, int W, int n_out, const PyramidLevelDesc *levels) {
    if (!p || p->H != H || p->W != W || p->n_out != n_out) return false;
    for (int l = 0; l < n_out; l++)
        if (p->Ho[(size_t)l] != levels[l].H || p->Wo[(size_t)l] != levels[l].W) return false;
    return true;
}

unsigned pyramid_sep_mask(const PyramidSepPlan *p) { return p ? p->handled_mask : 0u; }

tdk_status launch_pyramid_sep(PyramidSepPlan *p, const double *const *srcs, int n_arrays, int64_t src_stride,
                              const PyramidLevelDesc *levels, int batch, hipStream_t stream) {
    SepArgs &m = p->args;
    if (m.n == 0) return TDK_OK;
    m.n_arrays = n_arrays;
    m.batch = batch;
    for (int s = 0; s < m.n; s++) {
        const PyramidLevelDesc &d = levels[p->level_of_slot[s]];
        for (int i = 0; i < 4; i++) {
            m.lv[s].src[i] = i < n_arrays ? srcs[i] : nullptr;
            m.lv[s].dst[i] = i < n_arrays ? d.dst[i] : nullptr;
        }
        m.lv[s].src_stride = src_stride;
        m.lv[s].dst_stride = d.stride;
    }
    const int64_t images = (int64_t)n_arrays * batch;
    const int64_t blocks = 8 * ((images + 7) / 8) * m.tile_end[m.n - 1];
    if (blocks >= (1ll << 31)) {
        set_error("pyramid: %lld blocks exceed the grid limit", (long long)blocks);
        return TDK_ERR_INVALID_ARGUMENT;
    }
    k_pyramid_sep<<<(unsigned)blocks, kBlock, p->lds_bytes, stream>>>(m);
    TDK_LAUNCH_CHECK();
    return TDK_OK;
}

}  // namespace tdk
