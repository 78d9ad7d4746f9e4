// device_fit.h -- per-contour geometry on the device: cv::contourArea, cv::fitEllipseDirect (with its
// general-fit fallback), cv::RotatedRect::points, rm::lightblob and rm::armour construction.
//
// Reference call sites: /root/reference/src/objdetect.cpp:64-84 (area gate, ellipse fit, ratio and
// tilt tests), src/core.cpp:9-19 + 265-283 (lightblob), src/core.cpp:21-49 + 285-404 (armour).
//
// Bit-exactness contract: double/float IEEE arithmetic in a FIXED operation order -- sequential
// accumulation over the contour points in contour order, no FMA contraction (-ffp-contract=off),
// correctly rounded div/sqrt, transcendentals from pinned_math.h.  Integer-valued sums (area) could be
// reduced in any order; the scaled moment sums cannot: each is a sequential chain in contour order (one lane per chain, wave_detect.h).
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>

#include "../../include/rmcv_abi.h"
#include "pinned_math.h"
#include "device_eig3.h"

namespace rmcv {

#define RMCV_PI 3.1415926535897932384626433832795

__device__ __forceinline__ double dabs(double x) { return __builtin_fabs(x); }
__device__ __forceinline__ double dsqrt(double x) { return ::sqrt(x); }

// ---- lane-resident small arrays -----------------------------------------------------------------
// The scalar tails below (3x3 eigen-solver, 5x5 Jacobi) index small matrices with run-time subscripts.  As
// thread-private arrays those would live in scratch memory (hundreds of cycles per access).  Every lane of the
// wavefront executes the tail redundantly with identical values, so the arrays are kept "across the lanes" of ONE
// register instead: element i lives in lane i; a read is a v_readlane pair (the subscript is wave-uniform), a
// write is a select.  The proxy types keep the algorithm text identical to the CPU restatement.
__device__ __forceinline__ double lane_get(double reg, int idx)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(reg);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, idx);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), idx);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
struct LaneRef {
    double* reg;
    int idx, lane;
    __device__ __forceinline__ operator double() const { return lane_get(*reg, idx); }
    __device__ __forceinline__ LaneRef& operator=(double v) { *reg = (lane == idx) ? v : *reg; return *this; }
    __device__ __forceinline__ LaneRef& operator=(const LaneRef& o) { return *this = (double)o; }
    __device__ __forceinline__ LaneRef& operator+=(double v) { return *this = (double)*this + v; }
    __device__ __forceinline__ LaneRef& operator-=(double v) { return *this = (double)*this - v; }
};
struct LaneVec { // up to 64 doubles
    double reg;
    int lane;
    __device__ __forceinline__ explicit LaneVec(int l) : reg(0.0), lane(l) {}
    __device__ __forceinline__ LaneRef operator[](int i) { return LaneRef{&reg, i, lane}; }
    __device__ __forceinline__ double get(int i) const { return lane_get(reg, i); }
};
struct LaneRow {
    double* reg;
    int base, lane;
    __device__ __forceinline__ LaneRef operator[](int j) { return LaneRef{reg, base + j, lane}; }
};
struct LaneMat3 { // 3 x 3, row-major in lanes 0..8
    double reg;
    int lane;
    __device__ __forceinline__ explicit LaneMat3(int l) : reg(0.0), lane(l) {}
    __device__ __forceinline__ LaneRow operator[](int i) { return LaneRow{&reg, 3 * i, lane}; }
};

// ---- 3x3 real non-symmetric eigen-solver: JAMA orthes + hqr2 (cv::eigenNonSymmetric) -------------
__device__ inline void cdiv_(double xr, double xi, double yr, double yi, double* cr, double* ci)
{
    double r, d;
    if (dabs(yr) > dabs(yi)) {
        r = yi / yr;
        d = yr + r * yi;
        *cr = (xr + r * xi) / d;
        *ci = (xi - r * xr) / d;
    } else {
        r = yr / yi;
        d = yi + r * yr;
        *cr = (r * xr + xi) / d;
        *ci = (r * xi - xr) / d;
    }
}

__device__ inline void eig_orthes(LaneMat3& H, LaneMat3& V, int lane)
{
    LaneVec ort(lane);
    const int low = 0, high = 2;
    for (int m = low + 1; m <= high - 1; m++) {
        double scale = 0.0;
        for (int i = m; i <= high; i++) scale = scale + dabs(H[i][m - 1]);
        if (scale != 0.0) {
            double h = 0.0;
            for (int i = high; i >= m; i--) {
                ort[i] = H[i][m - 1] / scale;
                h += ort[i] * ort[i];
            }
            double g = dsqrt(h);
            if (ort[m] > 0) g = -g;
            h = h - ort[m] * g;
            ort[m] = ort[m] - g;
            for (int j = m; j < 3; j++) {
                double f = 0.0;
                for (int i = high; i >= m; i--) f += ort[i] * H[i][j];
                f = f / h;
                for (int i = m; i <= high; i++) H[i][j] -= f * ort[i];
            }
            for (int i = 0; i <= high; i++) {
                double f = 0.0;
                for (int j = high; j >= m; j--) f += ort[j] * H[i][j];
                f = f / h;
                for (int j = m; j <= high; j++) H[i][j] -= f * ort[j];
            }
            ort[m] = scale * ort[m];
            H[m][m - 1] = scale * g;
        }
    }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) V[i][j] = (i == j ? 1.0 : 0.0);
    for (int m = high - 1; m >= low + 1; m--) {
        if (H[m][m - 1] != 0.0) {
            for (int i = m + 1; i <= high; i++) ort[i] = H[i][m - 1];
            for (int j = m; j <= high; j++) {
                double g = 0.0;
                for (int i = m; i <= high; i++) g += ort[i] * V[i][j];
                g = (g / ort[m]) / H[m][m - 1];
                for (int i = m; i <= high; i++) V[i][j] += g * ort[i];
            }
        }
    }
}

__device__ inline void eig_hqr2(LaneMat3& H, LaneMat3& V, LaneVec& d, LaneVec& e)
{
    const int nn = 3;
    int n = nn - 1;
    const int low = 0, high = nn - 1;
    const double eps = 2.220446049250313e-16;
    double exshift = 0.0;
    double p = 0, q = 0, r = 0, s = 0, z = 0, t, w, x, y;
    double norm = 0.0;
    for (int i = 0; i < nn; i++)
        for (int j = (i - 1 > 0 ? i - 1 : 0); j < nn; j++) norm = norm + dabs(H[i][j]);

    int iter = 0;
    while (n >= low) {
        int l = n;
        while (l > low) {
            s = dabs(H[l - 1][l - 1]) + dabs(H[l][l]);
            if (s == 0.0) s = norm;
            if (dabs(H[l][l - 1]) < eps * s) break;
            l--;
        }
        if (l == n) {
            H[n][n] = H[n][n] + exshift;
            d[n] = H[n][n];
            e[n] = 0.0;
            n--;
            iter = 0;
        } else if (l == n - 1) {
            w = H[n][n - 1] * H[n - 1][n];
            p = (H[n - 1][n - 1] - H[n][n]) / 2.0;
            q = p * p + w;
            z = dsqrt(dabs(q));
            H[n][n] = H[n][n] + exshift;
            H[n - 1][n - 1] = H[n - 1][n - 1] + exshift;
            x = H[n][n];
            if (q >= 0) {
                if (p >= 0) z = p + z; else z = p - z;
                d[n - 1] = x + z;
                d[n] = d[n - 1];
                if (z != 0.0) d[n] = x - w / z;
                e[n - 1] = 0.0;
                e[n] = 0.0;
                x = H[n][n - 1];
                s = dabs(x) + dabs(z);
                p = x / s;
                q = z / s;
                r = dsqrt(p * p + q * q);
                p = p / r;
                q = q / r;
                for (int j = n - 1; j < nn; j++) {
                    z = H[n - 1][j];
                    H[n - 1][j] = q * z + p * H[n][j];
                    H[n][j] = q * H[n][j] - p * z;
                }
                for (int i = 0; i <= n; i++) {
                    z = H[i][n - 1];
                    H[i][n - 1] = q * z + p * H[i][n];
                    H[i][n] = q * H[i][n] - p * z;
                }
                for (int i = low; i <= high; i++) {
                    z = V[i][n - 1];
                    V[i][n - 1] = q * z + p * V[i][n];
                    V[i][n] = q * V[i][n] - p * z;
                }
            } else {
                d[n - 1] = x + p;
                d[n] = x + p;
                e[n - 1] = z;
                e[n] = -z;
            }
            n = n - 2;
            iter = 0;
        } else {
            x = H[n][n];
            y = 0.0;
            w = 0.0;
            if (l < n) {
                y = H[n - 1][n - 1];
                w = H[n][n - 1] * H[n - 1][n];
            }
            if (iter == 10) {
                exshift += x;
                for (int i = low; i <= n; i++) H[i][i] -= x;
                s = dabs(H[n][n - 1]) + dabs(H[n - 1][n - 2]);
                x = y = 0.75 * s;
                w = -0.4375 * s * s;
            }
            if (iter == 30) {
                s = (y - x) / 2.0;
                s = s * s + w;
                if (s > 0) {
                    s = dsqrt(s);
                    if (y < x) s = -s;
                    s = x - w / ((y - x) / 2.0 + s);
                    for (int i = low; i <= n; i++) H[i][i] -= s;
                    exshift += s;
                    x = y = w = 0.964;
                }
            }
            iter = iter + 1;
            if (iter > 300) { // termination guard shared with the oracle
                d[n] = H[n][n] + exshift;
                e[n] = 0.0;
                n--;
                iter = 0;
                continue;
            }
            int m = n - 2;
            while (m >= l) {
                z = H[m][m];
                r = x - z;
                s = y - z;
                p = (r * s - w) / H[m + 1][m] + H[m][m + 1];
                q = H[m + 1][m + 1] - z - r - s;
                r = H[m + 2][m + 1];
                s = dabs(p) + dabs(q) + dabs(r);
                p = p / s;
                q = q / s;
                r = r / s;
                if (m == l) break;
                if (dabs(H[m][m - 1]) * (dabs(q) + dabs(r)) <
                    eps * (dabs(p) * (dabs(H[m - 1][m - 1]) + dabs(z) + dabs(H[m + 1][m + 1]))))
                    break;
                m--;
            }
            for (int i = m + 2; i <= n; i++) {
                H[i][i - 2] = 0.0;
                if (i > m + 2) H[i][i - 3] = 0.0;
            }
            for (int k = m; k <= n - 1; k++) {
                const bool notlast = (k != n - 1);
                if (k != m) {
                    p = H[k][k - 1];
                    q = H[k + 1][k - 1];
                    r = (notlast ? H[k + 2][k - 1] : 0.0);
                    x = dabs(p) + dabs(q) + dabs(r);
                    if (x != 0.0) {
                        p = p / x;
                        q = q / x;
                        r = r / x;
                    }
                }
                if (x == 0.0) break;
                s = dsqrt(p * p + q * q + r * r);
                if (p < 0) s = -s;
                if (s != 0) {
                    if (k != m) H[k][k - 1] = -s * x;
                    else if (l != m) H[k][k - 1] = -H[k][k - 1];
                    p = p + s;
                    x = p / s;
                    y = q / s;
                    z = r / s;
                    q = q / p;
                    r = r / p;
                    for (int j = k; j < nn; j++) {
                        p = H[k][j] + q * H[k + 1][j];
                        if (notlast) {
                            p = p + r * H[k + 2][j];
                            H[k + 2][j] = H[k + 2][j] - p * z;
                        }
                        H[k][j] = H[k][j] - p * x;
                        H[k + 1][j] = H[k + 1][j] - p * y;
                    }
                    const int imax = (n < k + 3 ? n : k + 3);
                    for (int i = 0; i <= imax; i++) {
                        p = x * H[i][k] + y * H[i][k + 1];
                        if (notlast) {
                            p = p + z * H[i][k + 2];
                            H[i][k + 2] = H[i][k + 2] - p * r;
                        }
                        H[i][k] = H[i][k] - p;
                        H[i][k + 1] = H[i][k + 1] - p * q;
                    }
                    for (int i = low; i <= high; i++) {
                        p = x * V[i][k] + y * V[i][k + 1];
                        if (notlast) {
                            p = p + z * V[i][k + 2];
                            V[i][k + 2] = V[i][k + 2] - p * r;
                        }
                        V[i][k] = V[i][k] - p;
                        V[i][k + 1] = V[i][k + 1] - p * q;
                    }
                }
            }
        }
    }

    if (norm == 0.0) return;

    for (n = nn - 1; n >= 0; n--) {
        p = d[n];
        q = e[n];
        if (q == 0) {
            int l = n;
            H[n][n] = 1.0;
            for (int i = n - 1; i >= 0; i--) {
                w = H[i][i] - p;
                r = 0.0;
                for (int j = l; j <= n; j++) r = r + H[i][j] * H[j][n];
                if (e[i] < 0.0) {
                    z = w;
                    s = r;
                } else {
                    l = i;
                    if (e[i] == 0.0) {
                        if (w != 0.0) H[i][n] = -r / w;
                        else H[i][n] = -r / (eps * norm);
                    } else {
                        x = H[i][i + 1];
                        y = H[i + 1][i];
                        q = (d[i] - p) * (d[i] - p) + e[i] * e[i];
                        t = (x * s - z * r) / q;
                        H[i][n] = t;
                        if (dabs(x) > dabs(z)) H[i + 1][n] = (-r - w * t) / x;
                        else H[i + 1][n] = (-s - y * t) / z;
                    }
                    t = dabs(H[i][n]);
                    if ((eps * t) * t > 1)
                        for (int j = i; j <= n; j++) H[j][n] = H[j][n] / t;
                }
            }
        } else if (q < 0) {
            int l = n - 1;
            double cr, ci;
            if (dabs(H[n][n - 1]) > dabs(H[n - 1][n])) {
                H[n - 1][n - 1] = q / H[n][n - 1];
                H[n - 1][n] = -(H[n][n] - p) / H[n][n - 1];
            } else {
                cdiv_(0.0, -H[n - 1][n], H[n - 1][n - 1] - p, q, &cr, &ci);
                H[n - 1][n - 1] = cr;
                H[n - 1][n] = ci;
            }
            H[n][n - 1] = 0.0;
            H[n][n] = 1.0;
            for (int i = n - 2; i >= 0; i--) {
                double ra = 0.0, sa = 0.0, vr, vi;
                for (int j = l; j <= n; j++) {
                    ra = ra + H[i][j] * H[j][n - 1];
                    sa = sa + H[i][j] * H[j][n];
                }
                w = H[i][i] - p;
                if (e[i] < 0.0) {
                    z = w;
                    r = ra;
                    s = sa;
                } else {
                    l = i;
                    if (e[i] == 0) {
                        cdiv_(-ra, -sa, w, q, &cr, &ci);
                        H[i][n - 1] = cr;
                        H[i][n] = ci;
                    } else {
                        x = H[i][i + 1];
                        y = H[i + 1][i];
                        vr = (d[i] - p) * (d[i] - p) + e[i] * e[i] - q * q;
                        vi = (d[i] - p) * 2.0 * q;
                        if (vr == 0.0 && vi == 0.0)
                            vr = eps * norm * (dabs(w) + dabs(q) + dabs(x) + dabs(y) + dabs(z));
                        cdiv_(x * r - z * ra + q * sa, x * s - z * sa - q * ra, vr, vi, &cr, &ci);
                        H[i][n - 1] = cr;
                        H[i][n] = ci;
                        if (dabs(x) > (dabs(z) + dabs(q))) {
                            H[i + 1][n - 1] = (-ra - w * H[i][n - 1] + q * H[i][n]) / x;
                            H[i + 1][n] = (-sa - w * H[i][n] - q * H[i][n - 1]) / x;
                        } else {
                            cdiv_(-r - y * H[i][n - 1], -s - y * H[i][n], z, q, &cr, &ci);
                            H[i + 1][n - 1] = cr;
                            H[i + 1][n] = ci;
                        }
                    }
                    t = dabs(H[i][n - 1]) > dabs(H[i][n]) ? dabs(H[i][n - 1]) : dabs(H[i][n]);
                    if ((eps * t) * t > 1)
                        for (int j = i; j <= n; j++) {
                            H[j][n - 1] = H[j][n - 1] / t;
                            H[j][n] = H[j][n] / t;
                        }
                }
            }
        }
    }
    for (int j = nn - 1; j >= low; j--)
        for (int i = low; i <= high; i++) {
            z = 0.0;
            const int kmax = (j < high ? j : high);
            for (int k = low; k <= kmax; k++) z = z + V[i][k] * H[k][j];
            V[i][j] = z;
        }
}

// cv::eigenNonSymmetric: eigenvalues sorted descending (stable), eigenvectors as rows
__device__ inline void eigen_nonsymmetric3(const double M[3][3], double eval[3], double evec[3][3], int lane)
{
#ifndef RMCV_EIG3_GENERAL
    // the 3 x 3 specialisation with compile-time subscripts (device_eig3.h) on the lane-resident arrays: same operations in the
    // same order, but an element access is a v_readlane with an immediate lane / a select against a constant mask, and there is
    // no loop or subscript bookkeeping on the wave's dependent chain
    Eig3T<LaneMat3, LaneVec> E(lane);
    eig3_solve(M, E);
    LaneVec& d = E.d;
    LaneMat3& V = E.V;
#else
    LaneMat3 H(lane), V(lane);
    LaneVec d(lane), e(lane);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) H[i][j] = M[i][j];
    eig_orthes(H, V, lane);
    eig_hqr2(H, V, d, e);
#endif
    const double d0 = d.get(0), d1 = d.get(1), d2 = d.get(2);
    // insertion sort of (0,1,2) by d, descending, ties keep their order
    int i0 = 0, i1 = 1, i2 = 2;
    double e0 = d0, e1 = d1;
    if (e0 < d1) { i0 = 1; i1 = 0; e0 = d1; e1 = d0; }
    if (e1 < d2) {
        i2 = i1;
        if (e0 < d2) { i1 = i0; i0 = 2; }
        else { i1 = 2; }
    }
    const int idx[3] = {i0, i1, i2};
#pragma unroll
    for (int i = 0; i < 3; i++) {
        eval[i] = d.get(idx[i]);
#pragma unroll
        for (int j = 0; j < 3; j++) evec[i][j] = lane_get(V.reg, 3 * j + idx[i]);
    }
}

// ---- symmetric k x k cyclic Jacobi + normal-equation least squares (general-fit fallback) --------
// K is a compile-time constant and the loops over p, q, r are unrolled: every subscript is a constant lane (v_readlane with an
// immediate, a select against a constant mask), no subscript arithmetic.  Operations and their order are unchanged, and so is the
// time (14-17 us of a general fit's 33 for K = 5, measured before and after): the sweep is a chain of dependent divisions and
// square roots -- ~6 sweeps x 10 rotations x 5 of them --, not of subscripts.  Kept for the two VGPRs it gives back.
// Round 4, -DRMCV_JACOBI_PLAIN (off): the sweeps on PLAIN (wave-uniform, replicated) registers, the lane-resident arrays read once
// and written once: 15 -> 10.5 us for K = 5 (what is left is the divisions and square roots), the fused sparse kernel 0.121 -> 0.105
// ms per batch ALONE -- at 180 instead of 165 VGPRs.  Measured where it would count and not kept: in the 4-wavefront kernel the
// pipelined step is unchanged (alternating pipelines of one process, against an identical pair: 0.993-0.995), in the 8-wavefront
// kernel two wavefronts per SIMD no longer fit beside two of k_binary's (2 x 184 + 2 x 80 > 512) and a batch with one dense frame
// goes from 0.27 to 0.31 ms per step; what remains is 3 % of a lone batch.
template <int K>
__device__ __forceinline__ void jacobi_sym(LaneVec& A_, LaneVec& lam, LaneVec& V_)
{
#ifdef RMCV_JACOBI_PLAIN
    double A[K * K], V[K * K];
#pragma unroll
    for (int i = 0; i < K * K; i++) A[i] = A_.get(i);
#else
    LaneVec& A = A_;
    LaneVec& V = V_;
#endif
#pragma unroll
    for (int i = 0; i < K; i++)
#pragma unroll
        for (int j = 0; j < K; j++) V[i * K + j] = (i == j) ? 1.0 : 0.0;
#pragma unroll 1
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0;
#pragma unroll
        for (int p = 0; p < K; p++)
#pragma unroll
            for (int q = p + 1; q < K; q++) off += dabs(A[p * K + q]);
        if (off == 0.0) break;
#pragma unroll
        for (int p = 0; p < K; p++)
#pragma unroll
            for (int q = p + 1; q < K; q++) {
                double apq = A[p * K + q];
                if (apq == 0.0) continue;
                double app = A[p * K + p], aqq = A[q * K + q];
                if (dabs(apq) < 1e-300 || dabs(apq) <= 1.1102230246251565e-16 * 1e-3 * dsqrt(dabs(app * aqq))) {
                    A[p * K + q] = 0.0;
                    A[q * K + p] = 0.0;
                    continue;
                }
                double theta = (aqq - app) / (2.0 * apq);
                double t = 1.0 / (dabs(theta) + dsqrt(theta * theta + 1.0));
                if (theta < 0) t = -t;
                double c = 1.0 / dsqrt(t * t + 1.0);
                double s = t * c;
                A[p * K + p] = app - t * apq;
                A[q * K + q] = aqq + t * apq;
                A[p * K + q] = 0.0;
                A[q * K + p] = 0.0;
#pragma unroll
                for (int r = 0; r < K; r++) {
                    if (r != p && r != q) {
                        double arp = A[r * K + p], arq = A[r * K + q];
                        double nrp = c * arp - s * arq;
                        double nrq = s * arp + c * arq;
                        A[r * K + p] = nrp;
                        A[p * K + r] = nrp;
                        A[r * K + q] = nrq;
                        A[q * K + r] = nrq;
                    }
                    double vrp = V[r * K + p], vrq = V[r * K + q];
                    V[r * K + p] = c * vrp - s * vrq;
                    V[r * K + q] = s * vrp + c * vrq;
                }
            }
    }
#ifdef RMCV_JACOBI_PLAIN
#pragma unroll
    for (int i = 0; i < K * K; i++) {
        A_[i] = A[i];
        V_[i] = V[i];
    }
#endif
#pragma unroll
    for (int i = 0; i < K; i++) lam[i] = (double)A[i * K + i];
}

// Least squares through the normal equations, device twin of oracle/rmcv_oracle.c normal_factor / normal_apply: the Jacobi
// eigen-decomposition G = V diag(lam) V^T is kept (across lanes) so that the pseudo-inverse can be applied again to the
// right-hand sides of the refinement steps (wave_detect.h normal_refine).
struct NormalFac {
    LaneVec lam, V;
    int k, use; // use: bit c set <=> singular value c is above the SVBackSubst threshold
    __device__ __forceinline__ explicit NormalFac(int lane) : lam(lane), V(lane), k(0), use(0) {}
};

// A: K x K in lanes 0..K*K-1 (destroyed); wmax / wmin: the extreme singular values of the design matrix
template <int K>
__device__ __forceinline__ void normal_factor(LaneVec& A, NormalFac& F, double* wmax, double* wmin)
{
    F.k = K;
    jacobi_sym<K>(A, F.lam, F.V);
    double wsum = 0, mx = 0, mn = 0;
#pragma unroll
    for (int i = 0; i < K; i++) {
        const double li = F.lam[i];
        const double wi = li > 0 ? dsqrt(li) : 0.0;
        wsum += wi;
        if (i == 0 || wi > mx) mx = wi;
        if (i == 0 || wi < mn) mn = wi;
    }
    const double thr = 2.0 * DBL_EPSILON * wsum;
    F.use = 0;
#pragma unroll
    for (int i = 0; i < K; i++) {
        const double li = F.lam[i];
        const double wi = li > 0 ? dsqrt(li) : 0.0;
        F.use |= (wi > thr ? 1 : 0) << i;
    }
    *wmax = mx;
    *wmin = mn;
}

// x (K <= 5 entries, the rest 0) = pinv(G) g;  g in lanes 0..K-1 of g.reg
template <int K>
__device__ __forceinline__ void normal_apply(NormalFac& F, LaneVec& g, double* x, int lane)
{
    LaneVec xs(lane);
#pragma unroll
    for (int i = 0; i < K; i++) xs[i] = 0.0;
#pragma unroll
    for (int c = 0; c < K; c++) {
        if (!((F.use >> c) & 1)) continue;
        double dot = 0.0;
#pragma unroll
        for (int r = 0; r < K; r++) dot += (double)F.V[r * K + c] * (double)g[r];
        dot = dot / (double)F.lam[c];
#pragma unroll
        for (int r = 0; r < K; r++) xs[r] += dot * (double)F.V[r * K + c];
    }
#pragma unroll
    for (int i = 0; i < 5; i++) x[i] = i < K ? xs.get(i) : 0.0;
}

__device__ __forceinline__ void get_ofs(int i, float eps, float* ox, float* oy)
{
    *ox = (float)(((i & 1) * 2 - 1)) * eps;
    *oy = (float)(((i & 2) - 1)) * eps;
}

// ---- scalar tails of the ellipse fits (the point sums are accumulated by the caller, wave-cooperatively,
//      in contour order: see k_detect.hip) -------------------------------------------------------------

// general conic ("LIN") fit, after the 5-parameter solve: centre from the conic gradient
__device__ inline void general_centre(const double gfp[5], double rp[5])
{
    const double a00 = 2 * gfp[0], a01 = gfp[2], a11 = 2 * gfp[1];
    const double det = a00 * a11 - a01 * a01;
    rp[0] = rp[1] = 0.0;
    if (det != 0.0) {
        rp[0] = (gfp[3] * a11 - gfp[4] * a01) / det;
        rp[1] = (a00 * gfp[4] - a01 * gfp[3]) / det;
    }
}

// general fit, after the 3-parameter re-fit: angle, radii, RotatedRect
__device__ inline void general_finish(const double gfp[3], double rp[5], double scale, float cx, float cy, rmcv_rrect* box)
{
    const double min_eps = 1e-8;
    double t;
    rp[4] = -0.5 * pm_atan2(gfp[2], gfp[1] - gfp[0]);
    if (dabs(gfp[2]) > min_eps) t = gfp[2] / pm_sin(-2.0 * rp[4]);
    else t = gfp[1] - gfp[0];
    rp[2] = dabs(gfp[0] + gfp[1] - t);
    if (rp[2] > min_eps) rp[2] = dsqrt(2.0 / rp[2]);
    rp[3] = dabs(gfp[0] + gfp[1] + t);
    if (rp[3] > min_eps) rp[3] = dsqrt(2.0 / rp[3]);
    box->cx = (float)(rp[0] / scale) + cx;
    box->cy = (float)(rp[1] / scale) + cy;
    box->w = (float)(rp[2] * 2 / scale);
    box->h = (float)(rp[3] * 2 / scale);
    box->angle = 0.0f;
    if (box->w > box->h) {
        float tmp = box->w;
        box->w = box->h;
        box->h = tmp;
        box->angle = (float)(90 + rp[4] * 180 / RMCV_PI);
    }
    if (box->angle < -180) box->angle += 360;
    if (box->angle > 360) box->angle -= 360;
}

__device__ __forceinline__ bool is_good_box(const rmcv_rrect* b) { return (b->h <= b->w * 30) && (b->w <= b->h * 30); }

// direct fit: scatter matrix DM (6x6, already divided by n) -> TM, Ts, M; returns det(M)
__device__ inline double direct_reduce(const double DM[6][6], double TM[3][3], double* Ts_out, double M[3][3])
{
#pragma unroll
    for (int c = 0; c < 3; c++) {
        TM[0][c] = DM[c][5] * DM[3][5] * DM[4][4] - DM[c][5] * DM[3][4] * DM[4][5] - DM[c][4] * DM[3][5] * DM[5][4] +
                   DM[c][3] * DM[4][5] * DM[5][4] + DM[c][4] * DM[3][4] * DM[5][5] - DM[c][3] * DM[4][4] * DM[5][5];
        TM[1][c] = DM[c][5] * DM[3][3] * DM[4][5] - DM[c][5] * DM[3][5] * DM[4][3] + DM[c][4] * DM[3][5] * DM[5][3] -
                   DM[c][3] * DM[4][5] * DM[5][3] - DM[c][4] * DM[3][3] * DM[5][5] + DM[c][3] * DM[4][3] * DM[5][5];
        TM[2][c] = DM[c][5] * DM[3][4] * DM[4][3] - DM[c][5] * DM[3][3] * DM[4][4] - DM[c][4] * DM[3][4] * DM[5][3] +
                   DM[c][3] * DM[4][4] * DM[5][3] + DM[c][4] * DM[3][3] * DM[5][4] - DM[c][3] * DM[4][3] * DM[5][4];
    }
    const double Ts = (-(DM[3][5] * DM[4][4] * DM[5][3]) + DM[3][4] * DM[4][5] * DM[5][3] + DM[3][5] * DM[4][3] * DM[5][4] -
                       DM[3][3] * DM[4][5] * DM[5][4] - DM[3][4] * DM[4][3] * DM[5][5] + DM[3][3] * DM[4][4] * DM[5][5]);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        M[0][c] = (DM[2][c] + (DM[2][3] * TM[0][c] + DM[2][4] * TM[1][c] + DM[2][5] * TM[2][c]) / Ts) / 2.;
        M[1][c] = -DM[1][c] - (DM[1][3] * TM[0][c] + DM[1][4] * TM[1][c] + DM[1][5] * TM[2][c]) / Ts;
        M[2][c] = (DM[0][c] + (DM[0][3] * TM[0][c] + DM[0][4] * TM[1][c] + DM[0][5] * TM[2][c]) / Ts) / 2.;
    }
    *Ts_out = Ts;
    return M[0][0] * (M[1][1] * M[2][2] - M[2][1] * M[1][2]) - M[0][1] * (M[1][0] * M[2][2] - M[2][0] * M[1][2]) +
           M[0][2] * (M[1][0] * M[2][1] - M[2][0] * M[1][1]);
}

// direct fit: eigenvector selection, conic -> RotatedRect (width <= height, angle in [0,180))
__device__ inline void direct_finish(const double M[3][3], const double TM[3][3], double Ts, double scale, double cx,
                                     double cy, rmcv_rrect* box, int lane)
{
    double eval[3], ev[3][3], cond[3];
    int i;
    eigen_nonsymmetric3(M, eval, ev, lane);
    cond[0] = (4.0 * ev[0][0] * ev[0][2] - ev[0][1] * ev[0][1]);
    cond[1] = (4.0 * ev[1][0] * ev[1][2] - ev[1][1] * ev[1][1]);
    cond[2] = (4.0 * ev[2][0] * ev[2][2] - ev[2][1] * ev[2][1]);
    if (cond[0] < cond[1]) i = (cond[1] < cond[2]) ? 2 : 1;
    else i = (cond[0] < cond[2]) ? 2 : 0;
    const double e0 = i == 0 ? ev[0][0] : (i == 1 ? ev[1][0] : ev[2][0]);
    const double e1 = i == 0 ? ev[0][1] : (i == 1 ? ev[1][1] : ev[2][1]);
    const double e2 = i == 0 ? ev[0][2] : (i == 1 ? ev[1][2] : ev[2][2]);
    double norm = dsqrt(e0 * e0 + e1 * e1 + e2 * e2);
    if (((e0 < 0.0 ? -1 : 1) * (e1 < 0.0 ? -1 : 1) * (e2 < 0.0 ? -1 : 1)) <= 0.0) norm = -1.0 * norm;
    const double pv0 = e0 / norm, pv1 = e1 / norm, pv2 = e2 / norm;
    const double q0 = (TM[0][0] * pv0 + TM[0][1] * pv1 + TM[0][2] * pv2) / Ts;
    const double q1 = (TM[1][0] * pv0 + TM[1][1] * pv1 + TM[1][2] * pv2) / Ts;
    const double q2 = (TM[2][0] * pv0 + TM[2][1] * pv1 + TM[2][2] * pv2) / Ts;
    const double u1 = pv2 * q0 * q0 - pv1 * q0 * q1 + pv0 * q1 * q1 + pv1 * pv1 * q2;
    const double u2 = pv0 * pv2 * q2;
    const double l1 = dsqrt(pv1 * pv1 + (pv0 - pv2) * (pv0 - pv2));
    const double l2 = pv0 + pv2;
    const double l3 = pv1 * pv1 - 4 * pv0 * pv2;
    const double p1 = 2 * pv2 * q0 - pv1 * q1;
    const double p2 = 2 * pv0 * q1 - pv1 * q0;
    const double x0 = (p1 / l3 / scale) + cx;
    const double y0 = (p2 / l3 / scale) + cy;
    const double a = dsqrt(2.) * dsqrt((u1 - 4.0 * u2) / ((l1 - l2) * l3)) / scale;
    const double b = dsqrt(2.) * dsqrt(-1.0 * ((u1 - 4.0 * u2) / ((l1 + l2) * l3))) / scale;
    double theta;
    if (pv1 == 0) theta = (pv0 < pv2) ? 0 : RMCV_PI / 2.;
    else theta = RMCV_PI / 2. + 0.5 * pm_atan2(pv1, (pv0 - pv2));
    box->cx = (float)x0;
    box->cy = (float)y0;
    box->w = (float)(2.0 * a);
    box->h = (float)(2.0 * b);
    if (box->w > box->h) {
        float tmp = box->w;
        box->w = box->h;
        box->h = tmp;
        box->angle = (float)(pm_fmod180(90 + theta * 180 / RMCV_PI));
    } else {
        box->angle = (float)(pm_fmod180(theta * 180 / RMCV_PI));
    }
}

// ---- cv::RotatedRect::points, rm::lightblob (core.cpp:9-19, 265-283) ----------------------------
__device__ inline void rrect_points(const rmcv_rrect* r, float pt[4][2])
{
    const double ang = r->angle * RMCV_PI / 180.;
    const float b = (float)pm_cos(ang) * 0.5f;
    const float a = (float)pm_sin(ang) * 0.5f;
    pt[0][0] = r->cx - a * r->h - b * r->w;
    pt[0][1] = r->cy + b * r->h - a * r->w;
    pt[1][0] = r->cx + a * r->h - b * r->w;
    pt[1][1] = r->cy - b * r->h - a * r->w;
    pt[2][0] = 2 * r->cx - pt[0][0];
    pt[2][1] = 2 * r->cy - pt[0][1];
    pt[3][0] = 2 * r->cx - pt[1][0];
    pt[3][1] = 2 * r->cy - pt[1][1];
}

__device__ inline void make_lightblob(const rmcv_rrect* box, int camp, rmcv_lightblob* out)
{
    out->angle = box->angle > 90 ? box->angle - 90 : box->angle + 90;
    out->target = camp;
    out->center[0] = box->cx;
    out->center[1] = box->cy;
    float t[4][2];
    rrect_points(box, t);
    // std::sort on 4 elements by y = insertion sort, ties keep their order
    for (int i = 1; i < 4; i++) {
        float kx = t[i][0], ky = t[i][1];
        int j = i - 1;
        while (j >= 0 && ky < t[j][1]) {
            t[j + 1][0] = t[j][0];
            t[j + 1][1] = t[j][1];
            j--;
        }
        t[j + 1][0] = kx;
        t[j + 1][1] = ky;
    }
    const bool swap_up = t[0][0] < t[1][0], swap_down = t[2][0] < t[3][0];
    const int i0 = swap_down ? 2 : 3, i1 = swap_up ? 0 : 1, i2 = swap_up ? 1 : 0, i3 = swap_down ? 3 : 2;
    out->vertices[0][0] = t[i0][0]; out->vertices[0][1] = t[i0][1];
    out->vertices[1][0] = t[i1][0]; out->vertices[1][1] = t[i1][1];
    out->vertices[2][0] = t[i2][0]; out->vertices[2][1] = t[i2][1];
    out->vertices[3][0] = t[i3][0]; out->vertices[3][1] = t[i3][1];
    out->size[0] = box->h < box->w ? box->h : box->w;
    out->size[1] = box->h < box->w ? box->w : box->h;
}

// ---- rm::armour (core.cpp:21-49) and its helpers (core.cpp:285-404) ------------------------------
__device__ __forceinline__ float point_distance(const float a[2], const float b[2])
{
    const double dx = (double)(a[0] - b[0]), dy = (double)(a[1] - b[1]);
    return (float)dsqrt(dx * dx + dy * dy);
}

// the reference's unqualified abs / atan2 / sin / cos on floats -- `ov` bit 0: abs(float) resolves to int abs(int)
// (the argument is truncated towards zero), bit 1: atan2 / sin / cos resolve to the double functions; 0: the float overloads
// (RMCV_OPT_OVERLOADS; oracle: orc_set_overload_mode)
__device__ __forceinline__ float abs_ov(float x, int ov)
{
    if (ov & 1) { const int i = (int)x; return (float)(i < 0 ? -i : i); }
    return __builtin_fabsf(x);
}
// `atan2(y, x) * 180.0f / static_cast<float>(CV_PI)` assigned to a float (objdetect.cpp:137)
__device__ __forceinline__ float atan2_deg_ov(float y, float x, int ov)
{
    if (ov & 2) return (float)(pm_atan2((double)y, (double)x) * (double)180.0f / (double)(float)RMCV_PI);
    return pm_atan2f(y, x) * 180.0f / (float)RMCV_PI;
}

__device__ inline void extend_cord(const float pt1[2], const float pt2[2], float deltaLen, float dst1[2], float dst2[2], int ov)
{
    if (pt1[0] == pt2[0]) {
        dst1[0] = pt1[0];
        dst2[0] = pt1[0];
        if (pt1[1] > pt2[1]) { dst1[1] = pt1[1] + deltaLen; dst2[1] = pt2[1] - deltaLen; }
        else                 { dst1[1] = pt1[1] - deltaLen; dst2[1] = pt2[1] + deltaLen; }
    } else if (pt1[1] == pt2[1]) {
        dst1[1] = pt1[1];
        dst2[1] = pt1[1];
        if (pt1[0] > pt2[0]) { dst1[0] = pt1[0] + deltaLen; dst2[0] = pt2[0] - deltaLen; }
        else                 { dst1[0] = pt1[0] - deltaLen; dst2[0] = pt2[0] + deltaLen; }
    } else {
        const float k = (float)(pt1[1] - pt2[1]) / (float)(pt1[0] - pt2[0]);
        const float ay = abs_ov(pt1[1] - pt2[1], ov), ax = abs_ov(pt1[0] - pt2[0], ov); // :336 (float)abs(...)
        float theta, zoomY, zoomX;
        if (ov & 2) { // the double functions: float theta = atan2(double, double); sin(theta) * deltaLen in double
            theta = (float)pm_atan2((double)ay, (double)ax);
            zoomY = (float)(pm_sin((double)theta) * (double)deltaLen);
            zoomX = (float)(pm_cos((double)theta) * (double)deltaLen);
        } else {
            theta = pm_atan2f(ay, ax);
            zoomY = pm_sinf(theta) * deltaLen;
            zoomX = pm_cosf(theta) * deltaLen;
        }
        if (k > 0) {
            if (pt1[0] > pt2[0]) {
                dst1[0] = pt1[0] + zoomX; dst1[1] = pt1[1] + zoomY;
                dst2[0] = pt2[0] - zoomX; dst2[1] = pt2[1] - zoomY;
            } else {
                dst1[0] = pt1[0] - zoomX; dst1[1] = pt1[1] - zoomY;
                dst2[0] = pt2[0] + zoomX; dst2[1] = pt2[1] + zoomY;
            }
        } else {
            if (pt1[0] < pt2[0]) {
                dst1[0] = pt1[0] - zoomX; dst1[1] = pt1[1] + zoomY;
                dst2[0] = pt2[0] + zoomX; dst2[1] = pt2[1] - zoomY;
            } else {
                dst1[0] = pt1[0] + zoomX; dst1[1] = pt1[1] - zoomY;
                dst2[0] = pt2[0] - zoomX; dst2[1] = pt2[1] + zoomY;
            }
        }
    }
}

__device__ __forceinline__ void line_center(const float a[2], const float b[2], float out[2])
{
    out[0] = a[0] / 2 + b[0] / 2;
    out[1] = a[1] / 2 + b[1] / 2;
}

__device__ inline void make_armour(const rmcv_lightblob* a, const rmcv_lightblob* b, rmcv_armour* out, int ov)
{
    const rmcv_lightblob *L = a, *R = b;
    if (b->center[0] < a->center[0]) { L = b; R = a; }
    float v[4][2];
    v[0][0] = L->vertices[3][0]; v[0][1] = L->vertices[3][1];
    v[1][0] = L->vertices[2][0]; v[1][1] = L->vertices[2][1];
    v[2][0] = R->vertices[1][0]; v[2][1] = R->vertices[1][1];
    v[3][0] = R->vertices[0][0]; v[3][1] = R->vertices[0][1];
    const float distanceL = point_distance(v[0], v[1]);
    const float distanceR = point_distance(v[2], v[3]);
    const float offsetL = __builtin_roundf((distanceL / 0.50f - distanceL) / 2);
    const float offsetR = __builtin_roundf((distanceR / 0.50f - distanceR) / 2);
    float ic[4][2];
    extend_cord(v[0], v[1], offsetL, ic[0], ic[1], ov);
    extend_cord(v[3], v[2], offsetR, ic[3], ic[2], ov);
    float minx = ic[0][0], maxx = minx, miny = ic[0][1], maxy = miny;
    for (int i = 0; i < 4; i++) {
        out->icon[i][0] = ic[i][0];
        out->icon[i][1] = ic[i][1];
        if (i) {
            if (ic[i][0] < minx) minx = ic[i][0];
            if (ic[i][0] > maxx) maxx = ic[i][0];
            if (ic[i][1] < miny) miny = ic[i][1];
            if (ic[i][1] > maxy) maxy = ic[i][1];
        }
    }
    const int ix = (int)__builtin_floorf(minx), iy = (int)__builtin_floorf(miny);
    const int ax = (int)__builtin_floorf(maxx), ay = (int)__builtin_floorf(maxy);
    out->bbox[0] = (float)ix;
    out->bbox[1] = (float)iy;
    out->bbox[2] = (float)(ax - ix + 1);
    out->bbox[3] = (float)(ay - iy + 1);
    const float leftHeight = point_distance(v[0], v[1]);
    const float rightHeight = point_distance(v[2], v[3]);
    const float maxHeight = leftHeight > rightHeight ? leftHeight : rightHeight;
    const float sw = maxHeight * 1.0f, sh = maxHeight;
    float c01[2], c23[2], c[2];
    line_center(v[0], v[1], c01);
    line_center(v[2], v[3], c23);
    line_center(c01, c23, c);
    out->vertices[0][0] = c[0] - sw / 2; out->vertices[0][1] = c[1] - sh / 2;
    out->vertices[1][0] = c[0] - sw / 2; out->vertices[1][1] = c[1] + sh / 2;
    out->vertices[2][0] = c[0] + sw / 2; out->vertices[2][1] = c[1] + sh / 2;
    out->vertices[3][0] = c[0] + sw / 2; out->vertices[3][1] = c[1] - sh / 2;
}

} // namespace rmcv
