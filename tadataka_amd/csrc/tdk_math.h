// tdk_math.h -- per-point arithmetic shared by the HIP kernels and the host
// side of libtadataka_hip.so.  Written for gfx950; every function names the
// reference lines whose arithmetic it reproduces (paths relative to the
// reference checkout).
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#define TDK_HD __host__ __device__ __forceinline__

namespace tdk {

constexpr double kEps16 = 1e-16;                  // src/projection.rs:4
constexpr double kEpsM = 2.220446049250313e-16;   // f64::EPSILON, src/numeric.rs:1

struct Cam {
    double fx, fy, ox, oy;
};

// src/camera.rs:36-41, tadataka/camera/_normalizer.cpp:16-17
TDK_HD void normalize(const Cam &c, double ux, double uy, double &x, double &y) {
    x = (ux - c.ox) / c.fx;
    y = (uy - c.oy) / c.fy;
}

// src/camera.rs:43-48, tadataka/camera/_normalizer.cpp:25-26
TDK_HD void unnormalize(const Cam &c, double x, double y, double &ux, double &uy) {
    ux = x * c.fx + c.ox;
    uy = y * c.fy + c.oy;
}

// src/projection.rs:11-14
TDK_HD void project(double px, double py, double pz, double &x, double &y) {
    double z = pz + kEps16;
    x = px / z;
    y = py / z;
}

// src/transform.rs:17-23: (T [p;1])[0:3]
TDK_HD void transform(const double *T, double px, double py, double pz, double &qx, double &qy,
                      double &qz) {
    qx = ((T[0] * px + T[1] * py) + T[2] * pz) + T[3] * 1.0;
    qy = ((T[4] * px + T[5] * py) + T[6] * pz) + T[7] * 1.0;
    qz = ((T[8] * px + T[9] * py) + T[10] * pz) + T[11] * 1.0;
}

// src/warp.rs:11-29: inv_project -> transform -> project, depth = z
TDK_HD void warp(const double *T10, double x0, double y0, double d0, double &x1, double &y1,
                 double &d1) {
    double qx, qy, qz;
    transform(T10, x0 * d0, y0 * d0, 1.0 * d0, qx, qy, qz);
    project(qx, qy, qz, x1, y1);
    d1 = qz;
}

// src/warp.rs:70-86 PerspectiveWarp
TDK_HD void perspective_warp(const double *T10, const Cam &c0, const Cam &c1, double u0x,
                             double u0y, double d0, double &u1x, double &u1y, double &d1) {
    double x0, y0, x1, y1;
    normalize(c0, u0x, u0y, x0, y0);
    warp(T10, x0, y0, d0, x1, y1, d1);
    unnormalize(c1, x1, y1, u1x, u1y);
}

// src/image_range.rs:11-17, tadataka/utils.py:35-44 (inclusive, on floats)
TDK_HD bool in_range(double x, double y, int H, int W) {
    return 0. <= x && x <= (double)W - 1. && 0. <= y && y <= (double)H - 1.;
}

// src/interpolation.rs:9-43.  The reference short-circuits exact-integer
// coordinates so that column W / row H is never read; clamping the upper
// index and keeping all four terms is value-identical for finite images
// (the extra terms are exact zeros) and branch-free.
TDK_HD double bilinear(const double *img, int H, int W, double cx, double cy) {
    double lx = floor(cx), ly = floor(cy);
    int lxi = (int)lx, lyi = (int)ly;
    int uxi = min(lxi + 1, W - 1), uyi = min(lyi + 1, H - 1);
    double ux = lx + 1.0, uy = ly + 1.0;
    const double *r0 = img + (int64_t)lyi * W;
    const double *r1 = img + (int64_t)uyi * W;
    return r0[lxi] * (ux - cx) * (uy - cy) + r0[uxi] * (cx - lx) * (uy - cy) +
           r1[lxi] * (ux - cx) * (cy - ly) + r1[uxi] * (cx - lx) * (cy - ly);
}

// The same function with the reference's three short cuts taken literally (a term the reference does not form is
// replaced by -0.0, the neutral element of the addition, instead of texel * 0): identical to the above for finite
// images, and for an image with Inf / NaN texels it returns what the reference returns when a coordinate is an
// integer -- the texel itself, not NaN from 0 * Inf of a neighbour the reference never reads.  The parity-granular
// operator uses this one; the search loop of update_depth keeps the shorter form (frames are finite).
TDK_HD double bilinear_exact(const double *img, int H, int W, double cx, double cy) {
    double lx = floor(cx), ly = floor(cy);
    int lxi = (int)lx, lyi = (int)ly;
    int uxi = min(lxi + 1, W - 1), uyi = min(lyi + 1, H - 1);
    double ux = lx + 1.0, uy = ly + 1.0;
    const double *r0 = img + (int64_t)lyi * W;
    const double *r1 = img + (int64_t)uyi * W;
    const bool ix = lx == cx, iy = ly == cy;
    const double t00 = r0[lxi] * (ux - cx) * (uy - cy);
    const double t01 = ix ? -0.0 : r0[uxi] * (cx - lx) * (uy - cy);
    const double t10 = iy ? -0.0 : r1[lxi] * (ux - cx) * (cy - ly);
    const double t11 = (ix || iy) ? -0.0 : r1[uxi] * (cx - lx) * (cy - ly);
    return t00 + t01 + t10 + t11;
}

// src/numeric.rs:3-5
TDK_HD double safe_inv(double v) { return 1. / (v + kEpsM); }

// src/triangulation.rs:8-39
TDK_HD double calc_depth0(const double *T10, double x0x, double x0y, double x1x, double x1y) {
    int i = fabs(T10[3]) > fabs(T10[7]) ? 0 : 1;
    const double *ri = &T10[4 * i], *rz = &T10[8];
    double ti = T10[4 * i + 3], tz = T10[11];
    double x1i = i == 0 ? x1x : x1y;
    double n = ti - tz * x1i;
    double rzy = (rz[0] * x0x + rz[1] * x0y) + rz[2] * 1.0;
    double riy = (ri[0] * x0x + ri[1] * x0y) + ri[2] * 1.0;
    double d = rzy * x1i - riy;
    return n / (d + kEps16);
}

// ---- SE(3) on 12-double poses {R row-major, t} ------------------------------

// Rodrigues, R = exp([w]x) (what scipy Rotation.from_rotvec(w).as_matrix()
// returns; tadataka/pose.py:44-46)
TDK_HD void exp_so3(const double *w, double *R) {
    double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double th = sqrt(t2);
    double A, B;
    if (th < 1e-8) {
        A = 1.0 - t2 / 6.0;
        B = 0.5 - t2 / 24.0;
    } else {
        A = sin(th) / th;
        B = (1.0 - cos(th)) / t2;
    }
    double x = w[0], y = w[1], z = w[2];
    R[0] = 1.0 - B * (y * y + z * z); R[1] = -A * z + B * x * y;         R[2] = A * y + B * x * z;
    R[3] = A * z + B * x * y;         R[4] = 1.0 - B * (x * x + z * z); R[5] = -A * x + B * y * z;
    R[6] = -A * y + B * x * z;        R[7] = A * x + B * y * z;         R[8] = 1.0 - B * (x * x + y * y);
}

// exp_se3_t_ (tadataka/se3.py:15-29): t = V(w) v with K built from the
// *normalised* rotation vector; Taylor branch when theta < 1e-16.
TDK_HD void exp_se3_t(const double *xi, double *t) {
    const double *v = xi, *w = xi + 3;
    double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    double k0 = 0, k1 = 0, k2 = 0;
    if (th != 0) { k0 = w[0] / th; k1 = w[1] / th; k2 = w[2] / th; }
    double c1, c2;
    if (th < kEps16) { c1 = th / 2; c2 = th * th / 6; }
    else { c1 = (1 - cos(th)) / th; c2 = (th - sin(th)) / th; }
    // K v = k x v ; K K v = k x (k x v)
    double a0 = k1 * v[2] - k2 * v[1], a1 = k2 * v[0] - k0 * v[2], a2 = k0 * v[1] - k1 * v[0];
    double b0 = k1 * a2 - k2 * a1, b1 = k2 * a0 - k0 * a2, b2 = k0 * a1 - k1 * a0;
    t[0] = v[0] + c1 * a0 + c2 * b0;
    t[1] = v[1] + c1 * a1 + c2 * b1;
    t[2] = v[2] + c1 * a2 + c2 * b2;
}

// candidate = Pose.from_se3(xi) * pose  (tadataka/pose.py:44-46,51-53):
// R <- dR R,  t <- dR t + dt
TDK_HD void compose_update(const double *xi, const double *pose, double *out) {
    double dR[9], dt[3];
    exp_so3(xi + 3, dR);
    exp_se3_t(xi, dt);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++)
            out[3 * i + j] = dR[3 * i] * pose[j] + dR[3 * i + 1] * pose[3 + j] + dR[3 * i + 2] * pose[6 + j];
        out[9 + i] = dR[3 * i] * pose[9] + dR[3 * i + 1] * pose[10] + dR[3 * i + 2] * pose[11] + dt[i];
    }
}

// Cholesky solve of the (diagonally pre-scaled) 6x6 system; returns false when a
// pivot is not safely positive, i.e. the system is numerically rank deficient.
TDK_HD bool solve6_cholesky(const double *H21, const double *b, double *x) {
    double A[6][6], s[6], y[6];
    int k = 0;
    for (int i = 0; i < 6; i++)
        for (int j = i; j < 6; j++) { A[i][j] = H21[k]; A[j][i] = H21[k]; k++; }
    for (int i = 0; i < 6; i++) {
        if (!(A[i][i] > 0.0)) return false;
        s[i] = 1.0 / sqrt(A[i][i]);
    }
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) A[i][j] *= s[i] * s[j];   // unit diagonal
    for (int j = 0; j < 6; j++) {
        double d = A[j][j];
        for (int m = 0; m < j; m++) d -= A[j][m] * A[j][m];
        if (!(d > 1e-10)) return false;   // scaled pivot: cond(J) beyond ~1e5 goes to the eigen path
        d = sqrt(d);
        A[j][j] = d;
        for (int i = j + 1; i < 6; i++) {
            double v = A[i][j];
            for (int m = 0; m < j; m++) v -= A[i][m] * A[j][m];
            A[i][j] = v / d;
        }
    }
    for (int i = 0; i < 6; i++) {
        double v = s[i] * b[i];
        for (int m = 0; m < i; m++) v -= A[i][m] * y[m];
        y[i] = v / A[i][i];
    }
    for (int i = 5; i >= 0; i--) {
        double v = y[i];
        for (int m = i + 1; m < 6; m++) v -= A[m][i] * x[m];
        x[i] = v / A[i][i];
    }
    for (int i = 0; i < 6; i++) x[i] *= s[i];
    return true;
}

// Solves the 6x6 SPD system given as upper triangle H21 (row-major) and b:
// the normal-equation form of solve_linear_equation (tadataka/math.py:32-45).
// Well-conditioned systems take the Cholesky path; otherwise a cyclic Jacobi
// eigen-decomposition with a relative eigenvalue cut-off gives the minimum-norm
// least-squares solution, as lstsq does when J is rank deficient.  Returns the
// number of eigen-directions used.
// The rank-deficient path works on three 6x6 matrices with run-time indexing.  EXTERNAL: they live in a
// caller-provided workspace of 108 doubles (LDS for lane 0 of k_dvo_reduce: as private arrays they went to
// scratch and took the reduce kernel from 16 to 24 us, for a path that almost never runs); otherwise on the stack.
template <bool EXTERNAL>
TDK_HD int solve6_eigen(const double *H21, const double *b, double *x, const bool *zero_col, double *ws);

template <bool EXTERNAL = false>
TDK_HD int solve6(const double *H21, const double *b, double *x, double n_rows = 0.0, double *ws = nullptr) {
    // gelsd drops singular values below rcond * sigma_max with rcond = eps * max(n, 6)
    // (numpy's default).  A column of J whose norm is below that bound is such a direction
    // whatever the other columns are; column scaling would turn its rounding noise into a
    // unit column, so it is zeroed before anything else (plane scenes with a 1-D texture:
    // the y-gradient of a resampled pyramid level is 1e-17, not 0).
    double cut = 2.220446049250313e-16 * fmax(n_rows, 6.0);
    cut *= cut;
    double dmax = 0.0;
    {
        int k = 0;
        for (int i = 0; i < 6; i++) { dmax = fmax(dmax, H21[k]); k += 6 - i; }
    }
    bool zero_col[6], any_zero = false;
    {
        int k = 0;
        for (int i = 0; i < 6; i++) { zero_col[i] = !(H21[k] > cut * dmax); any_zero |= zero_col[i]; k += 6 - i; }
    }
    if (!any_zero && solve6_cholesky(H21, b, x)) return 6;
    return solve6_eigen<EXTERNAL>(H21, b, x, zero_col, ws);
}

template <bool EXTERNAL>
TDK_HD int solve6_eigen(const double *H21, const double *b, double *x, const bool *zero_col, double *ws) {
    double local[EXTERNAL ? 1 : 108];
    double *w = EXTERNAL ? ws : local;
    double(*A)[6] = reinterpret_cast<double(*)[6]>(w);
    double(*V)[6] = reinterpret_cast<double(*)[6]>(w + 36);
    double(*N)[6] = reinterpret_cast<double(*)[6]>(w + 72);
    int k = 0;
    for (int i = 0; i < 6; i++)
        for (int j = i; j < 6; j++) {
            double v = (zero_col[i] || zero_col[j]) ? 0.0 : H21[k];
            A[i][j] = v; A[j][i] = v; k++;
        }
    // Jacobi scaling keeps the rotation/translation columns comparable
    double s[6];
    for (int i = 0; i < 6; i++) s[i] = A[i][i] > 0 ? 1.0 / sqrt(A[i][i]) : 1.0;
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) { A[i][j] *= s[i] * s[j]; V[i][j] = (i == j) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = 0.0;
        for (int i = 0; i < 6; i++)
            for (int j = i + 1; j < 6; j++) off += A[i][j] * A[i][j];
        if (off < 1e-30) break;
        for (int p = 0; p < 5; p++) {
            for (int q = p + 1; q < 6; q++) {
                double apq = A[p][q];
                if (fabs(apq) < 1e-300) continue;
                double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                for (int r = 0; r < 6; r++) {
                    double arp = A[r][p], arq = A[r][q];
                    A[r][p] = c * arp - sn * arq;
                    A[r][q] = sn * arp + c * arq;
                }
                for (int r = 0; r < 6; r++) {
                    double apr = A[p][r], aqr = A[q][r];
                    A[p][r] = c * apr - sn * aqr;
                    A[q][r] = sn * apr + c * aqr;
                }
                for (int r = 0; r < 6; r++) {
                    double vrp = V[r][p], vrq = V[r][q];
                    V[r][p] = c * vrp - sn * vrq;
                    V[r][q] = sn * vrp + c * vrq;
                }
            }
        }
    }
    double lmax = 0.0;
    for (int i = 0; i < 6; i++) lmax = fmax(lmax, A[i][i]);
    int kept = 0;
    double y[6] = {0, 0, 0, 0, 0, 0};
    bool dropped[6];
    for (int i = 0; i < 6; i++) {
        double lam = A[i][i];
        dropped[i] = !(lam > 1e-13 * lmax);
        if (dropped[i]) continue;
        kept++;
        double proj = 0.0;
        for (int r = 0; r < 6; r++) proj += V[r][i] * (s[r] * b[r]);
        proj /= lam;
        for (int r = 0; r < 6; r++) y[r] += proj * V[r][i];
    }
    for (int r = 0; r < 6; r++) x[r] = s[r] * y[r];
    if (kept < 6) {
        // x is the minimum-norm solution in the column-scaled coordinates.  lstsq
        // (gelsd on the unscaled J, tadataka/math.py:17-19) returns the minimum-norm one
        // in the original coordinates: remove from x its component in the null space of J,
        // which is spanned by n_i = S v_i for the dropped eigenvectors v_i (modified
        // Gram-Schmidt; at most 6 vectors of length 6).
        int m = 0;
        for (int i = 0; i < 6; i++) {
            if (!dropped[i]) continue;
            double n[6], nn = 0.0;
            for (int r = 0; r < 6; r++) n[r] = s[r] * V[r][i];
            for (int q = 0; q < m; q++) {
                double d = 0.0;
                for (int r = 0; r < 6; r++) d += N[q][r] * n[r];
                for (int r = 0; r < 6; r++) n[r] -= d * N[q][r];
            }
            for (int r = 0; r < 6; r++) nn += n[r] * n[r];
            if (!(nn > 0.0)) continue;
            nn = 1.0 / sqrt(nn);
            for (int r = 0; r < 6; r++) N[m][r] = n[r] * nn;
            m++;
        }
        for (int q = 0; q < m; q++) {
            double d = 0.0;
            for (int r = 0; r < 6; r++) d += N[q][r] * x[r];
            for (int r = 0; r < 6; r++) x[r] -= d * N[q][r];
        }
    }
    return kept;
}

}  // namespace tdk
