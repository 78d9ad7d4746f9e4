// wave_detect.h -- wave-cooperative device code of rm::filter_lightblobs / rm::filter_armours shared by k_detect.hip (the
// stand-alone kernels of the stage-wise entry points) and k_contours.hip (the fused per-frame kernel): the ellipse fit of one
// contour by one wavefront, the ordered compaction into the positive / negative lists and the pair loop.
#pragma once
#include "device_classify.h"
#include "device_fit.h"
#include "rmcv_internal.h"

namespace rmcv {

__device__ __forceinline__ int lanes_below(uint64_t m, int lane) { return __popcll(m & ((1ull << lane) - 1)); }

// ---- wave-cooperative per-contour work ------------------------------------------------------------------
// One wavefront owns one contour.  What is order dependent (the double-precision sums of scaled coordinates)
// stays a strictly sequential chain in contour order, but the chains are independent of each other, so the 21
// entries of the 6x6 scatter matrix (resp. 20 and 9 sums of the general fit) are accumulated by 21 different
// lanes at once: per 64-point chunk every lane prepares the design-matrix row of ITS point (order independent)
// in wave-private LDS, then lane e walks the 64 rows in order for ITS entry.  Integer-exact sums (shoelace area,
// coordinate sums) are reduced in any order -- every partial sum is an exactly representable integer.
struct WaveLds {
    double rows[64][6];
    double terms[64];
    double dm[32];
};

__device__ __forceinline__ double wave_sum_f64(double v)
{
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
__device__ __forceinline__ long long wave_sum_i64(long long v)
{
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// sequential sum, in point order, of one double per point (the per-point term is computed by the point's lane)
template <typename F>
__device__ __forceinline__ double seq_sum_terms(int n, int lane, WaveLds& L, F term)
{
    double s = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const double t = i < n ? term(i) : 0.0;
        const int m = n - base < 64 ? n - base : 64;
        if (m == 64) {
#pragma unroll
            for (int j = 0; j < 64; j++) s += lane_get(t, j); // in point order: lane j holds point base+j
        } else {
            for (int j = 0; j < m; j++) s += lane_get(t, j);
        }
    }
    return s;
}

// two such sums in one walk (two independent chains: the walk is bound by the chain's latency, not by its additions)
template <typename F0, typename F1>
__device__ __forceinline__ void seq_sum_terms2(int n, int lane, F0 term0, F1 term1, double* s0_, double* s1_)
{
    double s0 = 0, s1 = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const double t0 = i < n ? term0(i) : 0.0, t1 = i < n ? term1(i) : 0.0;
        const int m = n - base < 64 ? n - base : 64;
        if (m == 64) {
#pragma unroll
            for (int j = 0; j < 64; j++) { s0 += lane_get(t0, j); s1 += lane_get(t1, j); }
        } else {
            for (int j = 0; j < m; j++) { s0 += lane_get(t0, j); s1 += lane_get(t1, j); }
        }
    }
    *s0_ = s0;
    *s1_ = s1;
}

// Two sets of NS (<= 32) product sums over the SAME points in one walk: the wavefront's halves take 32 points at a time -- lanes 0..31
// prepare row set 0 of their point, lanes 32..63 row set 1 of the same point -- and lane e of each half walks its half's 32 rows, in
// point order, for its entry: every chain is the sequence of additions seq_sum_products makes, two attempts for the time of one.
// Returns the entry of set (lane >> 5) for lane (lane & 31) < ns.
template <int NR, typename F>
__device__ __forceinline__ double seq_sum_products_two(int n, int lane, WaveLds& L, int ns, int la, int lb, F make_row /* (i, set, r) */)
{
    double acc = 0;
    const int half = lane >> 5, l5 = lane & 31;
    for (int base = 0; base < n; base += 32) {
        const int i = base + l5;
        if (i < n) {
            double r[NR];
            make_row(i, half, r);
#pragma unroll
            for (int k = 0; k < NR; k++) L.rows[lane][k] = r[k];
        }
        __builtin_amdgcn_wave_barrier();
        const int m = n - base < 32 ? n - base : 32;
        if (l5 < ns) {
            const int r0 = half * 32;
#pragma unroll 8
            for (int j = 0; j < m; j++) acc += L.rows[r0 + j][la] * L.rows[r0 + j][lb];
        }
        __builtin_amdgcn_wave_barrier();
    }
    return acc;
}

// NR = row length, NS = number of sums; lane e < NS accumulates rows[j][la] * (lb < 0 ? konst : rows[j][lb])
template <int NR, typename F>
__device__ __forceinline__ double seq_sum_products(int n, int lane, WaveLds& L, int ns, int la, int lb, double konst, F make_row)
{
    double acc = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        if (i < n) {
            double r[NR];
            make_row(i, r);
#pragma unroll
            for (int k = 0; k < NR; k++) L.rows[lane][k] = r[k];
        }
        __builtin_amdgcn_wave_barrier();
        const int m = n - base < 64 ? n - base : 64;
        if (lane < ns) {
            if (lb >= 0) {
#pragma unroll 8
                for (int j = 0; j < m; j++) acc += L.rows[j][la] * L.rows[j][lb];
            } else {
#pragma unroll 8
                for (int j = 0; j < m; j++) acc += L.rows[j][la] * konst;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    return acc;
}

// The sums of the GENERAL fit's normal equations (oracle/rmcv_oracle.c fit_ellipse_general): the upper triangle of row x row
// (K (K + 1) / 2 sums) then row[a] * konst (K sums), written to out[] in that order by every lane (wave-uniform).  Unlike the
// direct fit's scatter matrix, whose summation order restates OpenCV's, the order here is the build's own, and it is the
// wave-shaped one of normal_refine: every lane sums its own points (i = lane, lane + 64, ... in increasing i), then one
// butterfly per sum (strides 32 .. 1).  As sequential chains in point order these two passes cost 10 and 6 us per contour.
template <int K, typename F>
__device__ __forceinline__ void par_sum_products(int n, int lane, double konst, double* out, F make_row)
{
    constexpr int NT = K * (K + 1) / 2, NS = NT + K;
    double acc[NS];
#pragma unroll
    for (int e = 0; e < NS; e++) acc[e] = 0.0;
#pragma unroll 1
    for (int i = lane; i < n; i += 64) {
        double row[K];
        make_row(i, row);
        int e = 0;
#pragma unroll
        for (int a = 0; a < K; a++)
#pragma unroll
            for (int b = a; b < K; b++, e++) acc[e] += row[a] * row[b];
#pragma unroll
        for (int a = 0; a < K; a++) acc[NT + a] += row[a] * konst;
    }
    for (int d = 32; d >= 1; d >>= 1) {
        double o[NS];
#pragma unroll
        for (int e = 0; e < NS; e++) o[e] = __shfl_xor(acc[e], d);
#pragma unroll
        for (int e = 0; e < NS; e++) acc[e] += o[e];
    }
#pragma unroll
    for (int e = 0; e < NS; e++) out[e] = acc[e];
}

// upper-triangle index e -> (a, b), a <= b, row-major, for a k x k symmetric matrix
__device__ __forceinline__ void tri_index(int e, int k, int* a, int* b)
{
    int r = 0, rem = e;
    while (r < k && rem >= k - r) { rem -= k - r; r++; }
    *a = r;
    *b = r + rem;
}

// Two steps of iterative refinement x += pinv(G) A^T (b - A x) of a normal-equation solution (oracle/rmcv_oracle.c
// normal_refine): the n x K design matrix is never stored, every lane makes the rows of ITS points (i = lane, lane + 64, ...,
// in increasing i), forms their residuals against the wave-uniform x and accumulates K partial sums; the partial sums are
// reduced by the butterfly of wave_sum_f64 (strides 32 .. 1) -- the oracle adds in exactly this shape.
template <int K, typename F>
__device__ __forceinline__ void normal_refine(NormalFac& Fac, int n, int lane, double bconst, double* x /* [5] */, F make_row)
{
    LaneVec xv(lane); // x across the lanes of one register: inside the point loop its entries are scalar operands (v_readlane)
#pragma unroll
    for (int a = 0; a < K; a++) xv.reg = lane == a ? x[a] : xv.reg;
    for (int step = 0; step < 2; step++) {
        double xs[K];
#pragma unroll
        for (int a = 0; a < K; a++) xs[a] = xv.get(a);
        double acc[K];
#pragma unroll
        for (int a = 0; a < K; a++) acc[a] = 0.0;
#pragma unroll 1
        for (int i = lane; i < n; i += 64) {
            double row[K];
            make_row(i, row);
            double t = row[0] * xs[0];
#pragma unroll
            for (int b = 1; b < K; b++) t += row[b] * xs[b];
            const double r = bconst - t;
#pragma unroll
            for (int a = 0; a < K; a++) acc[a] += row[a] * r;
        }
        // the K butterflies of wave_sum_f64 stage by stage, so that the K shuffles of a stage are in flight together (the same
        // additions in the same order as K separate reductions, whose dependent round trips would follow one another)
        for (int d = 32; d >= 1; d >>= 1) {
            double o[K];
#pragma unroll
            for (int a = 0; a < K; a++) o[a] = __shfl_xor(acc[a], d);
#pragma unroll
            for (int a = 0; a < K; a++) acc[a] += o[a];
        }
        LaneVec h(lane);
#pragma unroll
        for (int a = 0; a < K; a++) h.reg = lane == a ? acc[a] : h.reg;
        double dx[5];
        normal_apply<K>(Fac, h, dx, lane);
#pragma unroll
        for (int a = 0; a < K; a++) xv.reg = lane == a ? xs[a] + dx[a] : xv.reg;
    }
#pragma unroll
    for (int a = 0; a < K; a++) x[a] = xv.get(a);
}

// cv::fitEllipseDirect (objdetect.cpp:68) for one contour, by one wavefront.  sumx/sumy = integer coordinate sums.
// returns 0 = direct solution, 1 = general fit.  All results are wave-uniform.
#ifdef RMCV_PROFILE_FITS
#define FSTAMP(k) do { if (prof) prof[k] = wall_clock64(); } while (0)
#else
#define FSTAMP(k) do {} while (0)
#endif
__device__ inline int fit_ellipse_wave(const rmcv_point* __restrict__ pts, int n, long long sumx, long long sumy, WaveLds& L, int lane,
                                rmcv_rrect* box, long long* prof = nullptr)
{
    FSTAMP(0);
    bool g_have = false; // the general fit's scale sum, accumulated beside the direct fit's
    double g_s = 0;
    // ------------------------------------------------ direct (Fitzgibbon / Halir-Flusser)
    {
        const double cx = (double)sumx / n, cy = (double)sumy / n;
        // (the general fit's scale sum rides along: a bar -- the contour this path is made for -- fails the direct fit and needs it)
        g_have = sumx < (1ll << 24) && sumy < (1ll << 24);
        const float gcx = (float)sumx / (float)n, gcy = (float)sumy / (float)n;
        double s;
        if (g_have) {
            seq_sum_terms2(n, lane,
                           [&](int i) { return dabs((float)pts[i].x - cx) + dabs((float)pts[i].y - cy); },
                           [&](int i) { const float px = (float)pts[i].x - gcx, py = (float)pts[i].y - gcy; return dabs((double)px) + dabs((double)py); },
                           &s, &g_s);
        } else {
            s = seq_sum_terms(n, lane, L, [&](int i) { return dabs((float)pts[i].x - cx) + dabs((float)pts[i].y - cy); });
        }
        const double scale = 100.0 / (s > FLT_EPSILON ? s : (double)FLT_EPSILON);
        FSTAMP(1);
        int la = 0, lb = 0;
        tri_index((lane & 31) < 21 ? (lane & 31) : 0, 6, &la, &lb);
        double DM[6][6], TM[3][3], M[3][3], Ts = 0;
        // Both attempts' scatter sums in ONE walk (the second attempt's jitter depends on s and n only): lanes 0..20 hold the plain
        // points' entries, lanes 32..52 the jittered points'.  The second set is looked at only if the first determinant is too small.
        const float eps1 = (float)(s / (n * 2) * 1e-2);
        const double acc = seq_sum_products_two<6>(n, lane, L, 21, la, lb, [&](int i, int set, double* r) {
            float ox, oy;
            get_ofs(i, set ? eps1 : 0.0f, &ox, &oy);
            const double px = (((float)pts[i].x + ox) - cx) * scale, py = (((float)pts[i].y + oy) - cy) * scale;
            r[0] = px * px; r[1] = px * py; r[2] = py * py; r[3] = px; r[4] = py; r[5] = 1.0;
        });
        const double inv_n = 1.0 / n;
        if ((lane & 31) < 21) (lane < 32 ? L.dm : L.terms)[lane & 31] = acc * inv_n;
        __builtin_amdgcn_wave_barrier();
        // ... and both reductions at once: the lanes of the lower half reduce the plain attempt's matrix, those of the upper half the
        // jittered one's (the same instruction stream on two sets of values); the first attempt whose determinant passes is broadcast
        int iter;
        {
            const double* src = lane < 32 ? L.dm : L.terms;
            {
                int e = 0;
#pragma unroll
                for (int a = 0; a < 6; a++)
#pragma unroll
                    for (int b = a; b < 6; b++, e++) DM[a][b] = DM[b][a] = src[e];
            }
            const double det = direct_reduce(DM, TM, &Ts, M);
            const double det0 = lane_get(det, 0), det1 = lane_get(det, 32);
            iter = dabs(det0) > 1.0e-10 ? 0 : (dabs(det1) > 1.0e-10 ? 1 : 2);
            const int sel = iter == 1 ? 32 : 0;
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int b = 0; b < 3; b++) {
                    TM[a][b] = lane_get(TM[a][b], sel);
                    M[a][b] = lane_get(M[a][b], sel);
                }
            Ts = lane_get(Ts, sel);
        }
        __builtin_amdgcn_wave_barrier();
        FSTAMP(2);
        if (iter < 2) {
            direct_finish(M, TM, Ts, scale, cx, cy, box, lane);
            FSTAMP(3);
            if (is_good_box(box)) return 0;
        }
        FSTAMP(3);
    }
    // ------------------------------------------------ general conic fit (fallback)
    {
        float cx, cy;
        if (sumx < (1ll << 24) && sumy < (1ll << 24)) { // float accumulation is exact below 2^24
            cx = (float)sumx;
            cy = (float)sumy;
        } else {
            cx = 0;
            cy = 0;
            for (int i = 0; i < n; i++) { cx += (float)pts[i].x; cy += (float)pts[i].y; }
        }
        cx /= (float)n;
        cy /= (float)n;
        const double s = g_have ? g_s : seq_sum_terms(n, lane, L, [&](int i) {
            const float px = (float)pts[i].x - cx, py = (float)pts[i].y - cy;
            return dabs((double)px) + dabs((double)py);
        });
        const double scale = 100.0 / (s > FLT_EPSILON ? s : (double)FLT_EPSILON);
        FSTAMP(8);
        double gfp[5], rp[5] = {0, 0, 0, 0, 0};
        float eps = 0.0f;
        int la = 0, lb = 0;
        if (lane < 15) tri_index(lane, 5, &la, &lb);
        else { la = lane < 20 ? lane - 15 : 0; lb = -1; }
        for (int iter = 0; iter < 2; iter++) {
            auto row5 = [&](int i, double* r) {
                float ox = 0, oy = 0;
                if (iter) get_ofs(i, eps, &ox, &oy);
                const float fx = ((float)pts[i].x + ox) - cx, fy = ((float)pts[i].y + oy) - cy;
                const double px = fx * scale, py = fy * scale;
                r[0] = -px * px; r[1] = -py * py; r[2] = -px * py; r[3] = px; r[4] = py;
            };
            {
                double sums[20];
                par_sum_products<5>(n, lane, 10000.0, sums, row5);
#pragma unroll
                for (int e = 0; e < 20; e++)
                    if (lane == e) L.dm[e] = sums[e];
            }
            __builtin_amdgcn_wave_barrier();
            LaneVec G(lane), g(lane);
            { // lane a*5+b takes G(a,b) = G(b,a) from the upper-triangle sums; lane a takes g(a)
                const int a = lane / 5, b = lane - 5 * a;
                const int lo = a < b ? a : b, hi = a < b ? b : a;
                const int e = lo * 5 - lo * (lo - 1) / 2 + (hi - lo);
                G.reg = lane < 25 ? L.dm[e] : 0.0;
                g.reg = lane < 5 ? L.dm[15 + lane] : 0.0;
            }
            __builtin_amdgcn_wave_barrier();
            FSTAMP(9);
            NormalFac Fac(lane);
            double wmax, wmin;
            normal_factor<5>(G, Fac, &wmax, &wmin);
            FSTAMP(10);
            if (iter == 0 && wmax * FLT_EPSILON > wmin) {
                eps = (float)(s / (n * 2) * 1e-3);
                continue;
            }
            normal_apply<5>(Fac, g, gfp, lane);
            FSTAMP(11);
            normal_refine<5>(Fac, n, lane, 10000.0, gfp, row5);
            break;
        }
        FSTAMP(4);
        general_centre(gfp, rp);
        if (lane < 6) tri_index(lane, 3, &la, &lb);
        else { la = lane < 9 ? lane - 6 : 0; lb = -1; }
        const double r0 = rp[0], r1 = rp[1];
        auto row3 = [&](int i, double* r) {
            float ox = 0, oy = 0;
            if (eps != 0.0f) get_ofs(i, eps, &ox, &oy);
            const float fx = ((float)pts[i].x + ox) - cx, fy = ((float)pts[i].y + oy) - cy;
            const double px = fx * scale, py = fy * scale;
            r[0] = (px - r0) * (px - r0); r[1] = (py - r1) * (py - r1); r[2] = (px - r0) * (py - r1);
        };
        {
            double sums[9];
            par_sum_products<3>(n, lane, 1.0, sums, row3);
#pragma unroll
            for (int e = 0; e < 9; e++)
                if (lane == e) L.dm[e] = sums[e];
        }
        __builtin_amdgcn_wave_barrier();
        LaneVec G(lane), g(lane);
        {
            const int a = lane / 3, b = lane - 3 * a;
            const int lo = a < b ? a : b, hi = a < b ? b : a;
            const int e = lo * 3 - lo * (lo - 1) / 2 + (hi - lo);
            G.reg = lane < 9 ? L.dm[e] : 0.0;
            g.reg = lane < 3 ? L.dm[6 + lane] : 0.0;
        }
        __builtin_amdgcn_wave_barrier();
        FSTAMP(5);
        NormalFac Fac(lane);
        double wmax3, wmin3;
        normal_factor<3>(G, Fac, &wmax3, &wmin3);
        normal_apply<3>(Fac, g, gfp, lane);
        FSTAMP(12);
        normal_refine<3>(Fac, n, lane, 1.0, gfp, row3);
        FSTAMP(13);
        general_finish(gfp, rp, scale, cx, cy, box);
        FSTAMP(6);
    }
    return 1;
}

// the per-contour part of rm::filter_lightblobs (objdetect.cpp:62-80) for the contour with discovery index k of frame f, by one
// wavefront: size/area gate, ellipse fit, ratio/tilt tests -> slot kind (0 skipped, 1 positive, 2 negative) + fitted ellipse
struct FitGates {
    float tilt_max, ratio_lo, ratio_hi;
    double area_lo, area_hi;
    int ov; // SURVEY A.6 (device_fit.h: abs_ov)
};
// (start, len): the contour's place in the frame's point list -- the caller's copy, so that a workgroup that has just produced
// them need not read them back from global memory (a dependent round trip per contour: microseconds beside streaming kernels)
__device__ inline void fit_contour_slot_at(int f, int k, int n, const rmcv_point* __restrict__ pts, int start, int len, int max_contours,
                                           int max_points, const FitGates& G, int32_t* __restrict__ slot_kind,
                                           rmcv_rrect* __restrict__ slot_ell, WaveLds& L, int lane)
{
    const int c = n - 1 - k; // findContours order
    int kind = 0;
    rmcv_rrect ell = {0, 0, 0, 0, 0};
    if (len >= 6 && start + len <= max_points) { // objdetect.cpp:64
        const rmcv_point* cp = pts + start;
        // cv::contourArea + coordinate sums: integer-exact, any reduction order
        double a00 = 0;
        long long sx = 0, sy = 0;
        for (int i = lane; i < len; i += 64) {
            const rmcv_point p = cp[i], q = cp[i == 0 ? len - 1 : i - 1];
            a00 += (double)(float)q.x * (float)p.y - (double)(float)q.y * (float)p.x;
            sx += p.x;
            sy += p.y;
        }
        a00 = wave_sum_f64(a00);
        sx = wave_sum_i64(sx);
        sy = wave_sum_i64(sy);
        const double area = dabs(a00 * 0.5);
        if (area >= G.area_lo && area <= G.area_hi) {
#ifdef RMCV_PROFILE_FITS // (every fit prints: the phase timings of -DRMCV_PROFILE are only meaningful WITHOUT this one)
            long long pr[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            const int path = fit_ellipse_wave(cp, len, sx, sy, L, lane, &ell, pr);
            if (lane == 0 && ((f == 0 && len > 150) || (path == 1 && f < 64)))
                printf("[fit f%d c%d n=%d path=%d] s %.1f moments+reduce %.1f finish %.1f | gen: s %.1f sums20 %.1f jacobi5 %.1f apply %.1f refine5 %.1f | "
                       "centre+sums9 %.1f jacobi3+apply %.1f refine3 %.1f finish %.1f us\n", f, c, len, path, (pr[1] - pr[0]) / 100.0,
                       (pr[2] - pr[1]) / 100.0, (pr[3] - pr[2]) / 100.0, (pr[8] - pr[3]) / 100.0, (pr[9] - pr[8]) / 100.0, (pr[10] - pr[9]) / 100.0,
                       (pr[11] - pr[10]) / 100.0, (pr[4] - pr[11]) / 100.0, (pr[5] - pr[4]) / 100.0, (pr[12] - pr[5]) / 100.0,
                       (pr[13] - pr[12]) / 100.0, (pr[6] - pr[13]) / 100.0);
#else
            fit_ellipse_wave(cp, len, sx, sy, L, lane, &ell); // :68  (:69 minAreaRect is dead code in the reference)
#endif
            bool negative = false;
            const float mx = ell.w > ell.h ? ell.w : ell.h, mn = ell.w < ell.h ? ell.w : ell.h;
            const float ratio = mx / mn; // :71-73
            if (!(ratio >= G.ratio_lo && ratio <= G.ratio_hi)) negative = true;
            const float angle = ell.angle > 90 ? ell.angle - 90 : ell.angle + 90; // :78
            if (abs_ov(angle - 90, G.ov) > G.tilt_max) negative = true;           // :79
            kind = negative ? 2 : 1;
        }
    }
    if (lane == 0) {
        slot_kind[(int64_t)f * max_contours + c] = kind;
        slot_ell[(int64_t)f * max_contours + c] = ell;
    }
}

__device__ inline void fit_contour_slot(int f, int k, int n, const rmcv_point* __restrict__ pts, const int32_t* __restrict__ cs,
                                        const int32_t* __restrict__ cl, int max_contours, int max_points, const FitGates& G,
                                        int32_t* __restrict__ slot_kind, rmcv_rrect* __restrict__ slot_ell, WaveLds& L, int lane)
{
    fit_contour_slot_at(f, k, n, pts, cs[k], cl[k], max_contours, max_points, G, slot_kind, slot_ell, L, lane);
}

struct FitTail { // what k_pairs needs to finish filter_lightblobs and run filter_armours
    rmcv_lightblob* blobs;
    int32_t* blob_src;
    rmcv_rrect* ellipses;
    int32_t* neg_idx;
    int32_t* n_blobs;
    int32_t* n_neg;
    int32_t* status;
    rmcv_armour* armours;
    int32_t* n_armours;
    int max_blobs, max_armours, enemy, do_pairs;
    float angle_diff_max, shear_max, length_ratio_max;
    int ov; // SURVEY A.6 (device_fit.h: abs_ov)
};

// ordered compaction of the per-contour results of frame f into the reference's `positive` / `negative` lists
// (one wavefront); returns the number of positives
__device__ inline int blob_compact_frame(int f, int lane, const int32_t* slot_kind, const rmcv_rrect* slot_ell, int n, int max_contours,
                                  int enemy, rmcv_lightblob* blobs, int32_t* blob_src, rmcv_rrect* ellipses, int32_t* neg_idx,
                                  int32_t* n_blobs, int32_t* n_neg, int32_t* status, int max_blobs)
{
    rmcv_lightblob* ob = blobs + (int64_t)f * max_blobs;
    int32_t* osrc = blob_src + (int64_t)f * max_blobs;
    rmcv_rrect* oell = ellipses + (int64_t)f * max_blobs;
    int32_t* oneg = neg_idx + (int64_t)f * max_contours;
    int np = 0, nn = 0;
    for (int base = 0; base < n; base += 64) {
        const int c = base + lane;
        const int word = c < n ? slot_kind[(int64_t)f * max_contours + c] : 0;
        const int kind = word & 15, camp_code = word >> 4; // the legacy matcher votes a camp per contour (code = camp + 2)
        const uint64_t mp = __ballot(kind == 1), mn_ = __ballot(kind == 2);
        if (kind == 1) {
            const int o = np + lanes_below(mp, lane);
            if (o < max_blobs) {
                const rmcv_rrect ell = slot_ell[(int64_t)f * max_contours + c];
                make_lightblob(&ell, camp_code ? camp_code - 2 : enemy, &ob[o]); // :83 -> core.cpp:9-19
                osrc[o] = c;
                oell[o] = ell;
            }
        } else if (kind == 2) {
            oneg[nn + lanes_below(mn_, lane)] = c; // :82
        }
        np += __popcll(mp);
        nn += __popcll(mn_);
    }
    if (np > max_blobs) {
        if (lane == 0) atomicOr(&status[f], RMCV_FRAME_OVF_BLOBS);
        np = max_blobs;
    }
    if (lane == 0) {
        n_blobs[f] = np;
        n_neg[f] = nn;
    }
    return np;
}

__device__ __forceinline__ bool pair_ok(const rmcv_lightblob& a, const rmcv_lightblob& b, float angle_diff_max,
                                        float shear_max, float length_ratio_max, int ov)
{
    const float angle_difference = abs_ov(a.angle - b.angle, ov); // objdetect.cpp:131
    if (angle_difference > angle_diff_max) return false;
    const float y = abs_ov(a.center[1] - b.center[1], ov);
    const float x = abs_ov(a.center[0] - b.center[0], ov);
    const float rect_angle = atan2_deg_ov(y, x, ov); // :137
    const float shear_i = abs_ov(a.angle > 90 ? abs_ov(a.angle - rect_angle, ov) - 90 : abs_ov(180 - a.angle - rect_angle, ov) - 90, ov);
    const float shear_j = abs_ov(b.angle > 90 ? abs_ov(b.angle - rect_angle, ov) - 90 : abs_ov(180 - b.angle - rect_angle, ov) - 90, ov);
    if (shear_i > shear_max || shear_j > shear_max) return false; // :144
    const float hi = a.size[1], hj = b.size[1];
    const float mn = hi < hj ? hi : hj, mx = hi < hj ? hj : hi;
    if (mn / mx < length_ratio_max) return false;                                                     // :149
    if (abs_ov(a.center[1] - b.center[1], ov) > (a.size[1] + b.size[1]) / 2) return false;       // :153
    if (abs_ov(a.center[0] - b.center[0], ov) > (a.size[1] + b.size[1]) * 2) return false;       // :157
    return true;
}

// rm::filter_armours for frame f (one wavefront, n = number of light blobs).  The reference's double loop visits the pairs
// (i, j), i < j, in lexicographic order; here pair number p (same order) is lane p of a round of 64 pairs, so a frame's ~20 pairs
// are tested -- and the accepted ones built -- in one round instead of one round per i.
__device__ inline void armours_frame(int f, int lane, const rmcv_lightblob* blobs, int n, int max_blobs, float angle_diff_max,
                                     float shear_max, float length_ratio_max, int enemy, rmcv_armour* armours, int32_t* n_armours,
                                     int32_t* status, int max_armours, int ov)
{
    const rmcv_lightblob* lb = blobs + (int64_t)f * max_blobs;
    rmcv_armour* out = armours + (int64_t)f * max_armours;
    int na = 0;
    if (n >= 2) { // :120
        const int total = n * (n - 1) / 2;
        // this lane's pair of the first round, advanced by 64 pairs per round
        int i = 0, rem = lane;
        while (i < n - 1 && rem >= n - 1 - i) { rem -= n - 1 - i; i++; }
        for (int base = 0; base < total; base += 64) {
            const int j = i + 1 + rem;
            bool ok = false;
            rmcv_lightblob a, b;
            if (base + lane < total) {
                a = lb[i];
                b = lb[j];
                ok = a.target == enemy && b.target == enemy && pair_ok(a, b, angle_diff_max, shear_max, length_ratio_max, ov); // :124-157
            }
            const uint64_t m = __ballot(ok);
            if (ok) {
                const int o = na + lanes_below(m, lane);
                if (o < max_armours) {
                    make_armour(&a, &b, &out[o], ov); // :161 -> core.cpp:21-49
                    out[o].blob_i = i;
                    out[o].blob_j = j;
                }
            }
            na += __popcll(m);
            rem += 64;
            while (i < n - 1 && rem >= n - 1 - i) { rem -= n - 1 - i; i++; }
        }
    }
    if (lane == 0) {
        if (na > max_armours) {
            atomicOr(&status[f], RMCV_FRAME_OVF_ARMOURS);
            na = max_armours;
        }
        n_armours[f] = na;
    }
}

// what the fused per-frame kernel (k_contours.hip) needs beyond findContours' own arguments
struct SparseTail {
    int fused; // 0: findContours only
    FitGates G;
    rmcv_rrect* slot_ell;
    FitTail T;
    ClassifyArgs C; // C.enabled: the frame's armours are classified by the same workgroup (RMCV_STAGE_IDENTITY)
};

hipError_t launch_contours_w4(const Geom& g, const Bufs& b, const Limits& lim, const SparseTail& X, int force_literal, const SparseSched& Q, int grid,
                              hipStream_t s);
// the lean build for dense streams (k_contours_lean.hip): every frame on the mid tier, two workgroups per CU; flags: 2 (+ 8: deferred frames only)
hipError_t launch_contours_lean(const Geom& g, const Bufs& b, const Limits& lim, const SparseTail& X, int flags, const SparseSched& Q, int grid, hipStream_t s);

} // namespace rmcv
