// device_eig3.h -- cv::eigenNonSymmetric for a 3 x 3 matrix (JAMA orthes + hqr2, oracle/rmcv_oracle.c eig_orthes / eig_hqr2), with
// every subscript a compile-time constant.
//
// The general routine walks its matrices with run-time subscripts (n, l, m, k shrink and move as eigenvalues deflate).  For n = 3
// the walk has very few shapes: the active block ends at N = 2, 1 or 0; the scan for a small sub-diagonal element stops at l = N
// (one eigenvalue), l = N - 1 (a 2 x 2 block) or -- only for N = 2 -- at l = 0, the one case that takes a double-shift QR step,
// and that step always starts at m = 0 and sweeps k = 0, 1.  Written out per shape there is no loop or subscript bookkeeping on the
// wave's single dependent chain and every element access has a constant position (the run-time-subscript form of device_fit.h
// spent about two thirds of a direct fit, 18-22 us, here).  Operations, their order and every rounding are those of the general
// routine: the values are bit-identical (tests: every fit of the parity suites).
#pragma once

namespace rmcv {

// Mat: H[i][j], Vec: d[i] readable and assignable as doubles.  With plain arrays (double[3][3]) the state is 24 doubles in registers
// -- +45 VGPRs in the fused sparse kernel, which costs its co-residency with the pixel kernels; the kernels therefore instantiate
// it with the lane-resident arrays of device_fit.h (one register per matrix, element e in lane e): with constant subscripts an
// access is then a v_readlane with an immediate lane or a select against a constant lane mask, and the register count stays put.
template <class Mat, class Vec>
struct Eig3T {
    Mat H, V;
    Vec d, e;
    double exshift, p, q, r, s, z, t, w, x, y, norm; // JAMA keeps these across iterations and phases: so do we
    int n, iter;
    template <class... A>
    __device__ __forceinline__ explicit Eig3T(A... a) : H(a...), V(a...), d(a...), e(a...) {}
};

#define E3_DABS(v) __builtin_fabs(v)

__device__ __forceinline__ void eig3_cdiv(double xr, double xi, double yr, double yi, double* cr, double* ci)
{
    double r, d;
    if (E3_DABS(yr) > E3_DABS(yi)) {
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

// orthes with low = 0, high = 2: the only column to reduce is m = 1
template <class EE>
__device__ __forceinline__ void eig3_orthes(EE& E)
{
    auto& H = E.H;
    auto& V = E.V;
    double ort1 = 0.0, ort2 = 0.0;
    double scale = 0.0;
    scale = scale + E3_DABS(H[1][0]);
    scale = scale + E3_DABS(H[2][0]);
    if (scale != 0.0) {
        double h = 0.0;
        ort2 = H[2][0] / scale;
        h += ort2 * ort2;
        ort1 = H[1][0] / scale;
        h += ort1 * ort1;
        double g = ::sqrt(h);
        if (ort1 > 0) g = -g;
        h = h - ort1 * g;
        ort1 = ort1 - g;
#pragma unroll
        for (int j = 1; j < 3; j++) {
            double f = 0.0;
            f += ort2 * H[2][j];
            f += ort1 * H[1][j];
            f = f / h;
            H[1][j] = H[1][j] - f * ort1;
            H[2][j] = H[2][j] - f * ort2;
        }
#pragma unroll
        for (int i = 0; i < 3; i++) {
            double f = 0.0;
            f += ort2 * H[i][2];
            f += ort1 * H[i][1];
            f = f / h;
            H[i][1] = H[i][1] - f * ort1;
            H[i][2] = H[i][2] - f * ort2;
        }
        ort1 = scale * ort1;
        H[1][0] = scale * g;
    }
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) V[i][j] = (i == j ? 1.0 : 0.0);
    if (H[1][0] != 0.0) {
        ort2 = H[2][0];
#pragma unroll
        for (int j = 1; j < 3; j++) {
            double g = 0.0;
            g += ort1 * V[1][j];
            g += ort2 * V[2][j];
            g = (g / ort1) / H[1][0];
            V[1][j] = V[1][j] + g * ort1;
            V[2][j] = V[2][j] + g * ort2;
        }
    }
}

// one pass of hqr2's outer loop with the active block ending at row/column N
template <int N, class EE>
__device__ __forceinline__ void eig3_step(EE& E)
{
    auto& H = E.H;
    auto& V = E.V;
    const double eps = 2.220446049250313e-16;
    // l = N; while (l > 0) { ...; if (small) break; l--; }
    int l = N;
    {
        bool scanning = true;
#pragma unroll
        for (int ll = N; ll >= 1; ll--) {
            if (scanning) {
                E.s = E3_DABS(H[ll - 1][ll - 1]) + E3_DABS(H[ll][ll]);
                if (E.s == 0.0) E.s = E.norm;
                if (E3_DABS(H[ll][ll - 1]) < eps * E.s) scanning = false;
                else l = ll - 1;
            }
        }
    }
    if (l == N) { // one root found
        H[N][N] = H[N][N] + E.exshift;
        E.d[N] = H[N][N];
        E.e[N] = 0.0;
        E.n = N - 1;
        E.iter = 0;
        return;
    }
    if constexpr (N >= 1) {
        if (l == N - 1) { // two roots found
            E.w = H[N][N - 1] * H[N - 1][N];
            E.p = (H[N - 1][N - 1] - H[N][N]) / 2.0;
            E.q = E.p * E.p + E.w;
            E.z = ::sqrt(E3_DABS(E.q));
            H[N][N] = H[N][N] + E.exshift;
            H[N - 1][N - 1] = H[N - 1][N - 1] + E.exshift;
            E.x = H[N][N];
            if (E.q >= 0) { // real pair
                if (E.p >= 0) E.z = E.p + E.z;
                else E.z = E.p - E.z;
                E.d[N - 1] = E.x + E.z;
                E.d[N] = E.d[N - 1];
                if (E.z != 0.0) E.d[N] = E.x - E.w / E.z;
                E.e[N - 1] = 0.0;
                E.e[N] = 0.0;
                E.x = H[N][N - 1];
                E.s = E3_DABS(E.x) + E3_DABS(E.z);
                E.p = E.x / E.s;
                E.q = E.z / E.s;
                E.r = ::sqrt(E.p * E.p + E.q * E.q);
                E.p = E.p / E.r;
                E.q = E.q / E.r;
#pragma unroll
                for (int j = N - 1; j < 3; j++) { // row modification
                    E.z = H[N - 1][j];
                    H[N - 1][j] = E.q * E.z + E.p * H[N][j];
                    H[N][j] = E.q * H[N][j] - E.p * E.z;
                }
#pragma unroll
                for (int i = 0; i <= N; i++) { // column modification
                    E.z = H[i][N - 1];
                    H[i][N - 1] = E.q * E.z + E.p * H[i][N];
                    H[i][N] = E.q * H[i][N] - E.p * E.z;
                }
#pragma unroll
                for (int i = 0; i < 3; i++) { // accumulate transformations
                    E.z = V[i][N - 1];
                    V[i][N - 1] = E.q * E.z + E.p * V[i][N];
                    V[i][N] = E.q * V[i][N] - E.p * E.z;
                }
            } else { // complex pair
                E.d[N - 1] = E.x + E.p;
                E.d[N] = E.x + E.p;
                E.e[N - 1] = E.z;
                E.e[N] = -E.z;
            }
            E.n = N - 2;
            E.iter = 0;
            return;
        }
    }
    if constexpr (N == 2) { // l == 0: no convergence yet, one double-shift QR step on the whole matrix (m = 0, k = 0 then 1)
        E.x = H[2][2];
        E.y = 0.0;
        E.w = 0.0;
        E.y = H[1][1]; // l < n
        E.w = H[2][1] * H[1][2];
        if (E.iter == 10) { // Wilkinson's original ad hoc shift
            E.exshift += E.x;
#pragma unroll
            for (int i = 0; i < 3; i++) H[i][i] = H[i][i] - E.x;
            E.s = E3_DABS(H[2][1]) + E3_DABS(H[1][0]);
            E.x = E.y = 0.75 * E.s;
            E.w = -0.4375 * E.s * E.s;
        }
        if (E.iter == 30) { // MATLAB's new ad hoc shift
            E.s = (E.y - E.x) / 2.0;
            E.s = E.s * E.s + E.w;
            if (E.s > 0) {
                E.s = ::sqrt(E.s);
                if (E.y < E.x) E.s = -E.s;
                E.s = E.x - E.w / ((E.y - E.x) / 2.0 + E.s);
#pragma unroll
                for (int i = 0; i < 3; i++) H[i][i] = H[i][i] - E.s;
                E.exshift += E.s;
                E.x = E.y = E.w = 0.964;
            }
        }
        E.iter = E.iter + 1;
        if (E.iter > 300) { // termination guard shared with the oracle
            E.d[2] = H[2][2] + E.exshift;
            E.e[2] = 0.0;
            E.n = 1;
            E.iter = 0;
            return;
        }
        // m = n - 2 = 0 = l: the scan for two consecutive small sub-diagonal elements computes p, q, r once and stops
        E.z = H[0][0];
        E.r = E.x - E.z;
        E.s = E.y - E.z;
        E.p = (E.r * E.s - E.w) / H[1][0] + H[0][1];
        E.q = H[1][1] - E.z - E.r - E.s;
        E.r = H[2][1];
        E.s = E3_DABS(E.p) + E3_DABS(E.q) + E3_DABS(E.r);
        E.p = E.p / E.s;
        E.q = E.q / E.s;
        E.r = E.r / E.s;
        H[2][0] = 0.0; // i = m + 2
        // ---- k = 0 (= m, not the last)
        if (E.x == 0.0) return; // "break" out of the k loop
        E.s = ::sqrt(E.p * E.p + E.q * E.q + E.r * E.r);
        if (E.p < 0) E.s = -E.s;
        if (E.s != 0) {
            // k == m and l == m: H[k][k-1] is not touched
            E.p = E.p + E.s;
            E.x = E.p / E.s;
            E.y = E.q / E.s;
            E.z = E.r / E.s;
            E.q = E.q / E.p;
            E.r = E.r / E.p;
#pragma unroll
            for (int j = 0; j < 3; j++) { // row modification
                E.p = H[0][j] + E.q * H[1][j];
                E.p = E.p + E.r * H[2][j];
                H[2][j] = H[2][j] - E.p * E.z;
                H[0][j] = H[0][j] - E.p * E.x;
                H[1][j] = H[1][j] - E.p * E.y;
            }
#pragma unroll
            for (int i = 0; i < 3; i++) { // column modification, i <= min(n, k + 3) = 2
                E.p = E.x * H[i][0] + E.y * H[i][1];
                E.p = E.p + E.z * H[i][2];
                H[i][2] = H[i][2] - E.p * E.r;
                H[i][0] = H[i][0] - E.p;
                H[i][1] = H[i][1] - E.p * E.q;
            }
#pragma unroll
            for (int i = 0; i < 3; i++) { // accumulate transformations
                E.p = E.x * V[i][0] + E.y * V[i][1];
                E.p = E.p + E.z * V[i][2];
                V[i][2] = V[i][2] - E.p * E.r;
                V[i][0] = V[i][0] - E.p;
                V[i][1] = V[i][1] - E.p * E.q;
            }
        }
        // ---- k = 1 (the last)
        E.p = H[1][0];
        E.q = H[2][0];
        E.r = 0.0;
        E.x = E3_DABS(E.p) + E3_DABS(E.q) + E3_DABS(E.r);
        if (E.x != 0.0) {
            E.p = E.p / E.x;
            E.q = E.q / E.x;
            E.r = E.r / E.x;
        }
        if (E.x == 0.0) return;
        E.s = ::sqrt(E.p * E.p + E.q * E.q + E.r * E.r);
        if (E.p < 0) E.s = -E.s;
        if (E.s != 0) {
            H[1][0] = -E.s * E.x;
            E.p = E.p + E.s;
            E.x = E.p / E.s;
            E.y = E.q / E.s;
            E.z = E.r / E.s;
            E.q = E.q / E.p;
            E.r = E.r / E.p;
#pragma unroll
            for (int j = 1; j < 3; j++) {
                E.p = H[1][j] + E.q * H[2][j];
                H[1][j] = H[1][j] - E.p * E.x;
                H[2][j] = H[2][j] - E.p * E.y;
            }
#pragma unroll
            for (int i = 0; i < 3; i++) { // i <= min(n, k + 3) = 2
                E.p = E.x * H[i][1] + E.y * H[i][2];
                H[i][1] = H[i][1] - E.p;
                H[i][2] = H[i][2] - E.p * E.q;
            }
#pragma unroll
            for (int i = 0; i < 3; i++) {
                E.p = E.x * V[i][1] + E.y * V[i][2];
                V[i][1] = V[i][1] - E.p;
                V[i][2] = V[i][2] - E.p * E.q;
            }
        }
    }
}

// back-substitution for the eigenvector of eigenvalue N
template <int N, class EE>
__device__ __forceinline__ void eig3_backsub(EE& E)
{
    auto& H = E.H;
    const double eps = 2.220446049250313e-16;
    E.p = E.d[N];
    E.q = E.e[N];
    if (E.q == 0) { // real vector
        int l = N;
        H[N][N] = 1.0;
#pragma unroll
        for (int i = N - 1; i >= 0; i--) {
            E.w = H[i][i] - E.p;
            E.r = 0.0;
#pragma unroll
            for (int j = 0; j <= N; j++)
                if (j >= l) E.r = E.r + H[i][j] * H[j][N];
            if (E.e[i] < 0.0) {
                E.z = E.w;
                E.s = E.r;
            } else {
                l = i;
                if (E.e[i] == 0.0) {
                    if (E.w != 0.0) H[i][N] = -E.r / E.w;
                    else H[i][N] = -E.r / (eps * E.norm);
                } else { // solve real equations
                    E.x = H[i][i + 1];
                    E.y = H[i + 1][i];
                    E.q = (E.d[i] - E.p) * (E.d[i] - E.p) + E.e[i] * E.e[i];
                    E.t = (E.x * E.s - E.z * E.r) / E.q;
                    H[i][N] = E.t;
                    if (E3_DABS(E.x) > E3_DABS(E.z)) H[i + 1][N] = (-E.r - E.w * E.t) / E.x;
                    else H[i + 1][N] = (-E.s - E.y * E.t) / E.z;
                }
                E.t = E3_DABS(H[i][N]); // overflow control
                if ((eps * E.t) * E.t > 1) {
#pragma unroll
                    for (int j = 0; j <= N; j++)
                        if (j >= i) H[j][N] = H[j][N] / E.t;
                }
            }
        }
    } else if (E.q < 0) { // complex vector: only for N >= 1 (the pair (N - 1, N) carries e = +z, -z)
        if constexpr (N >= 1) {
            int l = N - 1;
            double cr, ci;
            if (E3_DABS(H[N][N - 1]) > E3_DABS(H[N - 1][N])) {
                H[N - 1][N - 1] = E.q / H[N][N - 1];
                H[N - 1][N] = -(H[N][N] - E.p) / H[N][N - 1];
            } else {
                eig3_cdiv(0.0, -H[N - 1][N], H[N - 1][N - 1] - E.p, E.q, &cr, &ci);
                H[N - 1][N - 1] = cr;
                H[N - 1][N] = ci;
            }
            H[N][N - 1] = 0.0;
            H[N][N] = 1.0;
#pragma unroll
            for (int i = N - 2; i >= 0; i--) {
                double ra = 0.0, sa = 0.0, vr, vi;
#pragma unroll
                for (int j = 0; j <= N; j++)
                    if (j >= l) {
                        ra = ra + H[i][j] * H[j][N - 1];
                        sa = sa + H[i][j] * H[j][N];
                    }
                E.w = H[i][i] - E.p;
                if (E.e[i] < 0.0) {
                    E.z = E.w;
                    E.r = ra;
                    E.s = sa;
                } else {
                    l = i;
                    if (E.e[i] == 0) {
                        eig3_cdiv(-ra, -sa, E.w, E.q, &cr, &ci);
                        H[i][N - 1] = cr;
                        H[i][N] = ci;
                    } else { // solve complex equations
                        E.x = H[i][i + 1];
                        E.y = H[i + 1][i];
                        vr = (E.d[i] - E.p) * (E.d[i] - E.p) + E.e[i] * E.e[i] - E.q * E.q;
                        vi = (E.d[i] - E.p) * 2.0 * E.q;
                        if (vr == 0.0 && vi == 0.0)
                            vr = eps * E.norm * (E3_DABS(E.w) + E3_DABS(E.q) + E3_DABS(E.x) + E3_DABS(E.y) + E3_DABS(E.z));
                        eig3_cdiv(E.x * E.r - E.z * ra + E.q * sa, E.x * E.s - E.z * sa - E.q * ra, vr, vi, &cr, &ci);
                        H[i][N - 1] = cr;
                        H[i][N] = ci;
                        if (E3_DABS(E.x) > (E3_DABS(E.z) + E3_DABS(E.q))) {
                            H[i + 1][N - 1] = (-ra - E.w * H[i][N - 1] + E.q * H[i][N]) / E.x;
                            H[i + 1][N] = (-sa - E.w * H[i][N] - E.q * H[i][N - 1]) / E.x;
                        } else {
                            eig3_cdiv(-E.r - E.y * H[i][N - 1], -E.s - E.y * H[i][N], E.z, E.q, &cr, &ci);
                            H[i + 1][N - 1] = cr;
                            H[i + 1][N] = ci;
                        }
                    }
                    E.t = E3_DABS(H[i][N - 1]) > E3_DABS(H[i][N]) ? E3_DABS(H[i][N - 1]) : E3_DABS(H[i][N]); // overflow control
                    if ((eps * E.t) * E.t > 1) {
#pragma unroll
                        for (int j = 0; j <= N; j++)
                            if (j >= i) {
                                H[j][N - 1] = H[j][N - 1] / E.t;
                                H[j][N] = H[j][N] / E.t;
                            }
                    }
                }
            }
        }
    }
}

// eigenvalues d[] (+ i e[]) and the eigenvector matrix V (columns) of M, exactly as orthes + hqr2 leave them
template <class EE>
__device__ __forceinline__ void eig3_solve(const double M[3][3], EE& E)
{
    auto& H = E.H;
    auto& V = E.V;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) H[i][j] = M[i][j];
#pragma unroll
    for (int i = 0; i < 3; i++) E.d[i] = E.e[i] = 0.0;
    eig3_orthes(E);
    E.exshift = 0.0;
    E.p = E.q = E.r = E.s = E.z = 0.0;
    E.t = E.w = E.x = E.y = 0.0;
    E.norm = 0.0;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = (i - 1 > 0 ? i - 1 : 0); j < 3; j++) E.norm = E.norm + E3_DABS(H[i][j]);
    E.n = 2;
    E.iter = 0;
    while (E.n >= 0) {
        if (E.n == 2) eig3_step<2>(E);
        else if (E.n == 1) eig3_step<1>(E);
        else eig3_step<0>(E);
    }
    if (E.norm == 0.0) return;
    eig3_backsub<2>(E);
    eig3_backsub<1>(E);
    eig3_backsub<0>(E);
#pragma unroll
    for (int j = 2; j >= 0; j--)
#pragma unroll
        for (int i = 0; i < 3; i++) {
            double z = 0.0;
#pragma unroll
            for (int k = 0; k <= j; k++) z = z + V[i][k] * H[k][j];
            V[i][j] = z;
        }
}

#undef E3_DABS

} // namespace rmcv
