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
        out[i] = tdk::bilinear_exact(img, H, W, p.x, p.y);
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

}  // namespace

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
// luma of the first three interleaved channels; uint8 input is multiplied by 1/255 first (img_as_float)
// (img_as_float).  -ffp-contract=off keeps the two products and two sums as written.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_rgb2gray(const T *__restrict__ rgb, int64_t n, int channels,
                                                     double *__restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const T *p = rgb + i * channels;
        double r, g, b;
        if (sizeof(T) == 1) {   // img_as_float multiplies by the reciprocal (skimage/util/dtype.py: np.multiply(image, 1. / imax_in)): 24 of the 256 values differ from x / 255 in the last bit
            const double s = 1.0 / 255.0;
            r = (double)p[0] * s; g = (double)p[1] * s; b = (double)p[2] * s;
        }
        else { r = (double)p[0]; g = (double)p[1]; b = (double)p[2]; }
        out[i] = (0.2125 * r + 0.7154 * g) + 0.0721 * b;
    }
}

}  // namespace

extern "C" {

tdk_status tdk_normalize(const double *kp, int64_t n, const double *camera, double *out) {
    TDK_API_GUARD;
    return normalize_impl(kp, n, camera, out, 0);
}

tdk_status tdk_unnormalize(const double *kp, int64_t n, const double *camera, double *out) {
    TDK_API_GUARD;
    return normalize_impl(kp, n, camera, out, 1);
}

tdk_status tdk_project_vecs(const double *points, int64_t n, double *out) {
    TDK_API_GUARD;
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
    TDK_API_GUARD;
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
    TDK_API_GUARD;
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
    TDK_API_GUARD;
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
    TDK_API_GUARD;
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
    TDK_API_GUARD;
    TDK_REQUIRE(T10 && x0 && x1 && depth, "null pointer");
    *depth = tdk::calc_depth0(T10, x0[0], x0[1], x1[0], x1[1]);
    return TDK_OK;
}

tdk_status tdk_image_gradient(const double *image, int H, int W, double *gx, double *gy) {
    TDK_API_GUARD;
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

tdk_status tdk_rgb2gray(const double *rgb, int H, int W, int channels, double *gray) {
    TDK_API_GUARD;
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
    TDK_API_GUARD;
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
