// k_detect.hip -- rm::filter_lightblobs (/root/reference/src/objdetect.cpp:55-87) and
// rm::filter_armours (/root/reference/src/objdetect.cpp:114-166) on the device.
//
// k_fit:     one wavefront per contour (findContours order): size/area gate, ellipse fit, ratio/tilt tests.
// k_pairs:   one wavefront per frame appends the positives / negatives IN ORDER (ballot + prefix popcount),
//            builds the rm::lightblob PODs and runs the pair loop of filter_armours.
// k_armours: one wavefront per frame; for each i the lanes test 64 partners j > i at once and append the
//            accepted pairs in (i, j) lexicographic order, again by ballot + prefix popcount.
#include "device_fit.h"
#include "device_hull.h"
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

// upper-triangle index e -> (a, b), a <= b, row-major, for a k x k symmetric matrix
__device__ __forceinline__ void tri_index(int e, int k, int* a, int* b)
{
    int r = 0, rem = e;
    while (r < k && rem >= k - r) { rem -= k - r; r++; }
    *a = r;
    *b = r + rem;
}

// cv::fitEllipseDirect (objdetect.cpp:68) for one contour, by one wavefront.  sumx/sumy = integer coordinate sums.
// returns 0 = direct solution, 1 = general fit.  All results are wave-uniform.
#ifdef RMCV_PROFILE
#define FSTAMP(k) do { if (prof) prof[k] = wall_clock64(); } while (0)
#else
#define FSTAMP(k) do {} while (0)
#endif
__device__ int fit_ellipse_wave(const rmcv_point* __restrict__ pts, int n, long long sumx, long long sumy, WaveLds& L, int lane,
                                rmcv_rrect* box, long long* prof = nullptr)
{
    FSTAMP(0);
    // ------------------------------------------------ direct (Fitzgibbon / Halir-Flusser)
    {
        const double cx = (double)sumx / n, cy = (double)sumy / n;
        const double s = seq_sum_terms(n, lane, L, [&](int i) {
            return dabs((float)pts[i].x - cx) + dabs((float)pts[i].y - cy);
        });
        const double scale = 100.0 / (s > FLT_EPSILON ? s : (double)FLT_EPSILON);
        FSTAMP(1);
        int la = 0, lb = 0;
        tri_index(lane < 21 ? lane : 0, 6, &la, &lb);
        double DM[6][6], TM[3][3], M[3][3], Ts = 0;
        float eps = 0;
        int iter;
        for (iter = 0; iter < 2; iter++) {
            const double acc = seq_sum_products<6>(n, lane, L, 21, la, lb, 0.0, [&](int i, double* r) {
                float ox, oy;
                get_ofs(i, eps, &ox, &oy);
                const double px = (((float)pts[i].x + ox) - cx) * scale, py = (((float)pts[i].y + oy) - cy) * scale;
                r[0] = px * px; r[1] = px * py; r[2] = py * py; r[3] = px; r[4] = py; r[5] = 1.0;
            });
            const double inv_n = 1.0 / n;
            if (lane < 21) L.dm[lane] = acc * inv_n;
            __builtin_amdgcn_wave_barrier();
            {
                int e = 0;
#pragma unroll
                for (int a = 0; a < 6; a++)
#pragma unroll
                    for (int b = a; b < 6; b++, e++) DM[a][b] = DM[b][a] = L.dm[e];
            }
            __builtin_amdgcn_wave_barrier();
            const double det = direct_reduce(DM, TM, &Ts, M);
            if (dabs(det) > 1.0e-10) break;
            eps = (float)(s / (n * 2) * 1e-2);
        }
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
        const double s = seq_sum_terms(n, lane, L, [&](int i) {
            const float px = (float)pts[i].x - cx, py = (float)pts[i].y - cy;
            return dabs((double)px) + dabs((double)py);
        });
        const double scale = 100.0 / (s > FLT_EPSILON ? s : (double)FLT_EPSILON);
        double gfp[5], rp[5] = {0, 0, 0, 0, 0};
        float eps = 0.0f;
        int la = 0, lb = 0;
        if (lane < 15) tri_index(lane, 5, &la, &lb);
        else { la = lane < 20 ? lane - 15 : 0; lb = -1; }
        for (int iter = 0; iter < 2; iter++) {
            const double acc = seq_sum_products<5>(n, lane, L, 20, la, lb, 10000.0, [&](int i, double* r) {
                float ox = 0, oy = 0;
                if (iter) get_ofs(i, eps, &ox, &oy);
                const float fx = ((float)pts[i].x + ox) - cx, fy = ((float)pts[i].y + oy) - cy;
                const double px = fx * scale, py = fy * scale;
                r[0] = -px * px; r[1] = -py * py; r[2] = -px * py; r[3] = px; r[4] = py;
            });
            if (lane < 20) L.dm[lane] = acc;
            __builtin_amdgcn_wave_barrier();
            double wmax, wmin;
            LaneVec G(lane), g(lane);
            { // lane a*5+b takes G(a,b) = G(b,a) from the upper-triangle sums; lane a takes g(a)
                const int a = lane / 5, b = lane - 5 * a;
                const int lo = a < b ? a : b, hi = a < b ? b : a;
                const int e = lo * 5 - lo * (lo - 1) / 2 + (hi - lo);
                G.reg = lane < 25 ? L.dm[e] : 0.0;
                g.reg = lane < 5 ? L.dm[15 + lane] : 0.0;
            }
            __builtin_amdgcn_wave_barrier();
            normal_solve(G, g, 5, gfp, &wmax, &wmin, lane);
            if (iter == 0 && wmax * FLT_EPSILON > wmin) {
                eps = (float)(s / (n * 2) * 1e-3);
                continue;
            }
            break;
        }
        FSTAMP(4);
        general_centre(gfp, rp);
        if (lane < 6) tri_index(lane, 3, &la, &lb);
        else { la = lane < 9 ? lane - 6 : 0; lb = -1; }
        const double r0 = rp[0], r1 = rp[1];
        const double acc = seq_sum_products<3>(n, lane, L, 9, la, lb, 1.0, [&](int i, double* r) {
            float ox = 0, oy = 0;
            if (eps != 0.0f) get_ofs(i, eps, &ox, &oy);
            const float fx = ((float)pts[i].x + ox) - cx, fy = ((float)pts[i].y + oy) - cy;
            const double px = fx * scale, py = fy * scale;
            r[0] = (px - r0) * (px - r0); r[1] = (py - r1) * (py - r1); r[2] = (px - r0) * (py - r1);
        });
        if (lane < 9) L.dm[lane] = acc;
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
        normal_solve(G, g, 3, gfp, 0, 0, lane);
        general_finish(gfp, rp, scale, cx, cy, box);
        FSTAMP(6);
    }
    return 1;
}

__device__ int blob_compact_frame(int f, int lane, const int32_t* slot_kind, const rmcv_rrect* slot_ell, int n, int max_contours,
                                  int enemy, rmcv_lightblob* blobs, int32_t* blob_src, rmcv_rrect* ellipses, int32_t* neg_idx,
                                  int32_t* n_blobs, int32_t* n_neg, int32_t* status, int max_blobs);
__device__ void armours_frame(int f, int lane, const rmcv_lightblob* blobs, int n, int max_blobs, float angle_diff_max,
                              float shear_max, float length_ratio_max, int enemy, rmcv_armour* armours, int32_t* n_armours,
                              int32_t* status, int max_armours);

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
};

// kinds: 0 skipped, 1 positive, 2 negative.  grid (frames, FIT_CHUNKS), 4 wavefronts per block, one contour each.
static constexpr int FIT_CHUNKS = 4; // x 4 wavefronts = 16 contours of a frame at a time (a frame has ~10 with >= 6 points)
__global__ __launch_bounds__(256) void k_fit(const rmcv_point* __restrict__ points, const int32_t* __restrict__ cont_start,
                                            const int32_t* __restrict__ cont_len, const int32_t* __restrict__ n_contours,
                                            int max_contours, int max_points, float tilt_max, float ratio_lo, float ratio_hi,
                                            double area_lo, double area_hi, int32_t* __restrict__ slot_kind,
                                            rmcv_rrect* __restrict__ slot_ell, const int32_t* __restrict__ elig,
                                            const int32_t* __restrict__ n_elig)
{
    __shared__ WaveLds lds[4];
    __builtin_amdgcn_s_setprio(3); // latency-bound: issue ahead of the streaming pixel kernel of the next batch sharing the CU
    const int f = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    WaveLds& L = lds[wave];
    const int n = n_contours[f];
    const rmcv_point* pts = points + (int64_t)f * max_points;
    const int32_t* cs = cont_start + (int64_t)f * max_contours;
    const int32_t* cl = cont_len + (int64_t)f * max_contours;
    // the work list holds the discovery indices of the contours with >= 6 points (k_contours / load_contours wrote it
    // and marked the others "skipped"), so wavefronts are only spent on contours that reach the fit
    const int ne = n_elig[f];
    for (int e = blockIdx.y * 4 + wave; e < ne; e += 4 * gridDim.y) {
        const int k = elig[(int64_t)f * max_contours + e];
        const int c = n - 1 - k; // findContours order
        const int start = cs[k], len = cl[k];
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
            if (area >= area_lo && area <= area_hi) {
#ifdef RMCV_PROFILE
                long long pr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                const int path = fit_ellipse_wave(cp, len, sx, sy, L, lane, &ell, pr);
                if (lane == 0 && f == 0 && len > 150)
                    printf("[fit f%d c%d n=%d path=%d] s %.1f moments+reduce %.1f finish %.1f | gen s+G+solve %.1f refit %.1f solve3+finish %.1f us\n", f, c,
                           len, path, (pr[1] - pr[0]) / 100.0, (pr[2] - pr[1]) / 100.0, (pr[3] - pr[2]) / 100.0,
                           (pr[4] - pr[3]) / 100.0, (pr[5] - pr[4]) / 100.0, (pr[6] - pr[5]) / 100.0);
#else
                fit_ellipse_wave(cp, len, sx, sy, L, lane, &ell); // :68  (:69 minAreaRect is dead code in the reference)
#endif
                bool negative = false;
                const float mx = ell.w > ell.h ? ell.w : ell.h, mn = ell.w < ell.h ? ell.w : ell.h;
                const float ratio = mx / mn; // :71-73
                if (!(ratio >= ratio_lo && ratio <= ratio_hi)) negative = true;
                const float angle = ell.angle > 90 ? ell.angle - 90 : ell.angle + 90; // :78
                if (__builtin_fabsf(angle - 90) > tilt_max) negative = true;          // :79
                kind = negative ? 2 : 1;
            }
        }
        if (lane == 0) {
            slot_kind[(int64_t)f * max_contours + c] = kind;
            slot_ell[(int64_t)f * max_contours + c] = ell;
        }
    }
}


// ---- legacy matcher: rm::MatchLightBlob + the camp vote of rm::FindLightBlobs ----------------------
// One wavefront per contour of the frame's work list, one wavefront per workgroup (the hull tables take 16 B of LDS per
// frame column).  mode 0: the matcher (objdetect.cpp:9-28, 43-51); mode 1: cv::minAreaRect alone, every gate open (the
// stage-wise hook rmcv_min_area_rect).  Results go to the same per-contour slots k_fit uses, so k_pairs does the ordered
// compaction; a matching contour's slot word is 1 | (camp + 2) << 4.
struct MatchArgs {
    float min_ratio, max_ratio, tilt_angle, min_area, max_area;
    int fit_ellipse, mode;
    int wcap, hcap, ccap; // hull table sizes of this launch (device_hull.h)
    int pass;             // 0: all contours, small tables, what does not fit is marked MATCH_DEFERRED; 1: the marked ones, full tables
    const uint8_t* frames; // may be null: no camp vote (the blob gets CAMP_NEUTRAL)
    int64_t frame_pitch;
    int stride;
};

static constexpr int MATCH_CHUNKS = 16;
static constexpr int MATCH_DEFERRED = 0x100; // slot word: kind 0 (skipped by the compaction) + "repeat with full-size tables"
__global__ __launch_bounds__(64) void k_match(const rmcv_point* __restrict__ points, const int32_t* __restrict__ cont_start,
                                             const int32_t* __restrict__ cont_len, const int32_t* __restrict__ n_contours,
                                             int max_contours, int max_points, MatchArgs A, int32_t* __restrict__ slot_kind,
                                             rmcv_rrect* __restrict__ slot_ell, const int32_t* __restrict__ elig,
                                             const int32_t* __restrict__ n_elig, int32_t* __restrict__ status)
{
    extern __shared__ unsigned long long match_smem[];
    __builtin_amdgcn_s_setprio(3);
    const int f = blockIdx.x, lane = threadIdx.x;
    WaveLds& L = *reinterpret_cast<WaveLds*>(match_smem);
    HullLds H;
    hull_lds_carve(reinterpret_cast<unsigned char*>(match_smem) + sizeof(WaveLds), A.wcap, A.hcap, A.ccap, H);
    const bool final_pass = A.pass == 1 || A.mode == 1;
    const int n = n_contours[f];
    const rmcv_point* pts = points + (int64_t)f * max_points;
    const int32_t* cs = cont_start + (int64_t)f * max_contours;
    const int32_t* cl = cont_len + (int64_t)f * max_contours;
    const int ne = A.mode == 1 ? n : n_elig[f]; // the hook takes any contour, the matcher only those with >= 6 points
    for (int e = blockIdx.y; e < ne; e += gridDim.y) {
        const int k = A.mode == 1 ? e : elig[(int64_t)f * max_contours + e];
        const int c = n - 1 - k; // findContours order
        const int start = cs[k], len = cl[k];
        if (A.pass == 1 && slot_kind[(int64_t)f * max_contours + c] != MATCH_DEFERRED) continue;
        int word = 0;
        rmcv_rrect box = {0, 0, 0, 0, 0};
        if (len >= 1 && start + len <= max_points && (A.mode == 1 || len >= 6)) { // objdetect.cpp:12
            const rmcv_point* cp = pts + start;
            double a00 = 0;
            long long sx = 0, sy = 0;
            int minx = INT_MAX, maxx = INT_MIN, miny = INT_MAX, maxy = INT_MIN;
            for (int i = lane; i < len; i += 64) {
                const rmcv_point p = cp[i], q = cp[i == 0 ? len - 1 : i - 1];
                a00 += (double)(float)q.x * (float)p.y - (double)(float)q.y * (float)p.x;
                sx += p.x;
                sy += p.y;
                minx = p.x < minx ? p.x : minx;
                maxx = p.x > maxx ? p.x : maxx;
                miny = p.y < miny ? p.y : miny;
                maxy = p.y > maxy ? p.y : maxy;
            }
            a00 = wave_sum_f64(a00);
            sx = wave_sum_i64(sx);
            sy = wave_sum_i64(sy);
            for (int d = 32; d >= 1; d >>= 1) {
                const int a = __shfl_xor(minx, d), b = __shfl_xor(maxx, d), cc = __shfl_xor(miny, d), dd = __shfl_xor(maxy, d);
                minx = a < minx ? a : minx;
                maxx = b > maxx ? b : maxx;
                miny = cc < miny ? cc : miny;
                maxy = dd > maxy ? dd : maxy;
            }
            const int W = maxx - minx + 1, Hh = maxy - miny + 1; // cv::boundingRect on int points: inclusive box
            const double area = dabs(a00 * 0.5);
            int ovf = 0;
            if (minx < 0 || miny < 0 || W > A.wcap || maxy >= HULL_MAX_DIM || len > (1 << 20)) ovf = 1;
            if (A.mode == 1) {
                if (!ovf) min_area_rect_wave(cp, len, minx, W, H, lane, &box, &ovf);
                word = 1;
                if (ovf && lane == 0) atomicOr(&status[f], RMCV_FRAME_HULL);
            } else if (!(area < A.min_area || area > A.max_area)) { // :12
                rmcv_rrect ellipse;
                fit_ellipse_wave(cp, len, sx, sy, L, lane, &ellipse); // :15
                if (A.fit_ellipse) {
                    box = ellipse;
                    ovf = 0;
                } else if (!ovf) {
                    min_area_rect_wave(cp, len, minx, W, H, lane, &box, &ovf); // :16
                }
                if (ovf) { // the box is missing: repeat with the full-size tables, or give up (never for a findContours border)
                    word = final_pass ? 0 : MATCH_DEFERRED;
                    if (final_pass && lane == 0) atomicOr(&status[f], RMCV_FRAME_HULL);
                } else {
                    const float mx = box.w > box.h ? box.w : box.h, mn = box.w < box.h ? box.w : box.h;
                    const float ratio = mx / mn; // :19
                    bool ok = !(ratio > A.max_ratio || ratio < A.min_ratio);
                    const float angle = ellipse.angle > 90 ? ellipse.angle - 90 : ellipse.angle + 90; // :23
                    if (__builtin_fabsf(angle - 90) > A.tilt_angle) ok = false;                       // :24
                    if (ok) {
                        int camp = RMCV_CAMP_NEUTRAL;
                        if (A.frames) // :43-51
                            camp = camp_from_mean_wave(A.frames + (int64_t)f * A.frame_pitch, A.stride, minx, miny, W, Hh, lane);
                        word = 1 | ((camp + 2) << 4);
                    }
                }
            }
        }
        if (lane == 0) {
            slot_kind[(int64_t)f * max_contours + c] = word;
            slot_ell[(int64_t)f * max_contours + c] = box;
        }
    }
}

// ordered compaction of the per-contour results of frame f into the reference's `positive` / `negative` lists
// (one wavefront); returns the number of positives
__device__ int blob_compact_frame(int f, int lane, const int32_t* slot_kind, const rmcv_rrect* slot_ell, int n, int max_contours,
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
                                        float shear_max, float length_ratio_max)
{
    const float angle_difference = __builtin_fabsf(a.angle - b.angle); // objdetect.cpp:131
    if (angle_difference > angle_diff_max) return false;
    const float y = __builtin_fabsf(a.center[1] - b.center[1]);
    const float x = __builtin_fabsf(a.center[0] - b.center[0]);
    const float rect_angle = pm_atan2f(y, x) * 180.0f / (float)RMCV_PI; // :137
    const float shear_i = __builtin_fabsf(a.angle > 90 ? __builtin_fabsf(a.angle - rect_angle) - 90
                                                       : __builtin_fabsf(180 - a.angle - rect_angle) - 90);
    const float shear_j = __builtin_fabsf(b.angle > 90 ? __builtin_fabsf(b.angle - rect_angle) - 90
                                                       : __builtin_fabsf(180 - b.angle - rect_angle) - 90);
    if (shear_i > shear_max || shear_j > shear_max) return false; // :144
    const float hi = a.size[1], hj = b.size[1];
    const float mn = hi < hj ? hi : hj, mx = hi < hj ? hj : hi;
    if (mn / mx < length_ratio_max) return false;                                                     // :149
    if (__builtin_fabsf(a.center[1] - b.center[1]) > (a.size[1] + b.size[1]) / 2) return false;       // :153
    if (__builtin_fabsf(a.center[0] - b.center[0]) > (a.size[1] + b.size[1]) * 2) return false;       // :157
    return true;
}

// rm::filter_armours for frame f (one wavefront, n = number of light blobs)
__device__ void armours_frame(int f, int lane, const rmcv_lightblob* blobs, int n, int max_blobs, float angle_diff_max,
                              float shear_max, float length_ratio_max, int enemy, rmcv_armour* armours, int32_t* n_armours,
                              int32_t* status, int max_armours)
{
    const rmcv_lightblob* lb = blobs + (int64_t)f * max_blobs;
    rmcv_armour* out = armours + (int64_t)f * max_armours;
    int na = 0;
    if (n >= 2) { // :120
        for (int i = 0; i < n - 1; i++) {
            const rmcv_lightblob a = lb[i];
            if (a.target != enemy) continue; // :124
            for (int jb = i + 1; jb < n; jb += 64) {
                const int j = jb + lane;
                bool ok = false;
                rmcv_lightblob b;
                if (j < n) {
                    b = lb[j];
                    ok = (b.target == enemy) && pair_ok(a, b, angle_diff_max, shear_max, length_ratio_max);
                }
                const uint64_t m = __ballot(ok);
                if (ok) {
                    const int o = na + lanes_below(m, lane);
                    if (o < max_armours) {
                        make_armour(&a, &b, &out[o]); // :161 -> core.cpp:21-49
                        out[o].blob_i = i;
                        out[o].blob_j = j;
                    }
                }
                na += __popcll(m);
            }
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

// stand-alone pairing (the stage-wise rmcv_filter_armours entry point, or RMCV_STAGE_ARMOURS without RMCV_STAGE_BLOBS)
__global__ __launch_bounds__(64) void k_armours(const rmcv_lightblob* __restrict__ blobs, const int32_t* __restrict__ n_blobs,
                                               int max_blobs, float angle_diff_max, float shear_max,
                                               float length_ratio_max, int enemy, rmcv_armour* __restrict__ armours,
                                               int32_t* __restrict__ n_armours, int32_t* __restrict__ status, int max_armours)
{
    armours_frame(blockIdx.x, threadIdx.x, blobs, n_blobs[blockIdx.x], max_blobs, angle_diff_max, shear_max, length_ratio_max, enemy,
                  armours, n_armours, status, max_armours);
}

// the tail of filter_lightblobs (ordered compaction) and filter_armours (pair loop) of one frame in one launch:
// one wavefront per frame.  (Kept out of k_fit on purpose: folded into k_fit's last-arriving workgroup it cost
// 26 VGPRs, one wave of occupancy and +0.13 ms on MI355X.)
__global__ __launch_bounds__(64) void k_pairs(const int32_t* __restrict__ slot_kind, const rmcv_rrect* __restrict__ slot_ell,
                                             const int32_t* __restrict__ n_contours, int max_contours, FitTail T)
{
    const int f = blockIdx.x, lane = threadIdx.x;
    __builtin_amdgcn_s_setprio(3);
    const int np = blob_compact_frame(f, lane, slot_kind, slot_ell, n_contours[f], max_contours, T.enemy, T.blobs, T.blob_src,
                                      T.ellipses, T.neg_idx, T.n_blobs, T.n_neg, T.status, T.max_blobs);
    if (T.do_pairs) {
        __threadfence(); // the pair loop re-reads, across lanes, the blobs this wave just wrote
        armours_frame(f, lane, T.blobs, np, T.max_blobs, T.angle_diff_max, T.shear_max, T.length_ratio_max, T.enemy, T.armours,
                      T.n_armours, T.status, T.max_armours);
    }
}

// frame-major compaction of the per-frame armour slots into one list + offsets (the payload of the
// multi-GPU detection gather).  One workgroup; n_frames is a few hundred.
__global__ __launch_bounds__(256) void k_compact_armours(const rmcv_armour* __restrict__ armours,
                                                        const int32_t* __restrict__ n_armours, int n_frames, int max_armours,
                                                        rmcv_armour* __restrict__ out, int cap, int32_t* __restrict__ frame_offs)
{
    __shared__ int s_part[256];
    __shared__ int s_base;
    const int tid = threadIdx.x;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int f0 = 0; f0 < n_frames; f0 += 256) {
        const int f = f0 + tid;
        const int c = f < n_frames ? n_armours[f] : 0;
        s_part[tid] = c;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) { // inclusive Hillis-Steele scan
            int v = tid >= d ? s_part[tid - d] : 0;
            __syncthreads();
            s_part[tid] += v;
            __syncthreads();
        }
        const int excl = s_base + s_part[tid] - c;
        if (f < n_frames) {
            frame_offs[f] = excl;
            const uint32_t* src = reinterpret_cast<const uint32_t*>(armours + (int64_t)f * max_armours);
            uint32_t* dst = reinterpret_cast<uint32_t*>(out + excl);
            for (int k = 0; k < c; k++)
                if (excl + k < cap)
                    for (int w = 0; w < (int)(sizeof(rmcv_armour) / 4); w++) dst[k * (sizeof(rmcv_armour) / 4) + w] = src[k * (sizeof(rmcv_armour) / 4) + w];
        }
        __syncthreads();
        if (tid == 255) s_base += s_part[255];
        __syncthreads();
    }
    if (tid == 0) frame_offs[n_frames] = s_base;
}

hipError_t launch_compact_armours(const Geom& g, const Bufs& b, const Limits& lim, rmcv_armour* d_out, int cap,
                                  int32_t* d_frame_offs, hipStream_t s)
{
    hipLaunchKernelGGL(k_compact_armours, dim3(1), dim3(256), 0, s, b.armours, b.n_armours, g.n_frames, lim.max_armours, d_out,
                       cap, d_frame_offs);
    return hipGetLastError();
}

static hipError_t launch_pairs_tail(const Geom& g, const Bufs& b, const Limits& lim, const rmcv_params& p, bool pairs, hipStream_t s)
{
    FitTail T;
    T.blobs = b.blobs;
    T.blob_src = b.blob_src;
    T.ellipses = b.ellipses;
    T.neg_idx = b.neg_idx;
    T.n_blobs = b.n_blobs;
    T.n_neg = b.n_neg;
    T.status = b.status;
    T.armours = b.armours;
    T.n_armours = b.n_armours;
    T.max_blobs = lim.max_blobs;
    T.max_armours = lim.max_armours;
    T.enemy = p.camp;
    T.do_pairs = pairs ? 1 : 0;
    T.angle_diff_max = p.angle_diff_max;
    T.shear_max = p.shear_max;
    T.length_ratio_max = p.length_ratio_max;
    hipLaunchKernelGGL(k_pairs, dim3(g.n_frames), dim3(64), 0, s, b.slot_kind, b.slot_ell, b.n_contours, lim.max_contours, T);
    return hipGetLastError();
}

static hipError_t launch_fit(const Geom& g, const Bufs& b, const Limits& lim, const rmcv_params& p, bool pairs, hipStream_t s)
{
    hipLaunchKernelGGL(k_fit, dim3(g.n_frames, FIT_CHUNKS), dim3(256), 0, s, b.points, b.cont_start, b.cont_len, b.n_contours,
                       lim.max_contours, lim.max_points, p.tilt_max, p.ratio_lo, p.ratio_hi, p.area_lo, p.area_hi, b.slot_kind,
                       b.slot_ell, b.elig, b.n_elig);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_pairs_tail(g, b, lim, p, pairs, s);
}

// legacy matcher: k_match fills the per-contour slots, k_pairs compacts them (and pairs the blobs of camp p.camp)
hipError_t launch_match(const Geom& g, const Bufs& b, const Limits& lim, const rmcv_params& p, const rmcv_legacy_params& lp,
                        int mode, bool with_frames, bool pairs, hipStream_t s)
{
    MatchArgs A;
    A.min_ratio = lp.min_ratio;
    A.max_ratio = lp.max_ratio;
    A.tilt_angle = lp.tilt_angle;
    A.min_area = lp.min_area;
    A.max_area = lp.max_area;
    A.fit_ellipse = lp.fit_ellipse;
    A.mode = mode;
    A.frames = with_frames ? b.frames : nullptr;
    A.frame_pitch = g.frame_pitch;
    A.stride = g.stride;
    const int full_w = lim.max_width < HULL_MAX_DIM ? lim.max_width : HULL_MAX_DIM;
    const bool hull = mode == 1 || !lp.fit_ellipse;
    // pass 0 (every contour): small hull tables, ~10 KB of LDS per wavefront; pass 1 repeats the contours pass 0 marked
    // (wider than 256 columns, or a hull with more points than the small tables hold) with the full-size tables.  The hook
    // (mode 1) is a single full-size pass; with fitEllipse there is no hull at all.
    for (int pass = 0; pass < ((hull && mode == 0) ? 2 : 1); pass++) {
        const bool full = mode == 1 || pass == 1;
        A.pass = pass;
        A.wcap = !hull ? 1 : full ? full_w : (full_w < 256 ? full_w : 256);
        A.hcap = !hull ? 1 : full ? HULL_CAP : 128;
        A.ccap = !hull ? 1 : full ? HULL_CHAIN_CAP : 64;
        const size_t lds = sizeof(WaveLds) + hull_lds_bytes(A.wcap, A.hcap, A.ccap);
        static size_t lds_set = 0;
        if (lds > lds_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_match), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            lds_set = lds;
        }
        hipLaunchKernelGGL(k_match, dim3(g.n_frames, pass ? 2 : MATCH_CHUNKS), dim3(64), lds, s, b.points, b.cont_start, b.cont_len,
                           b.n_contours, lim.max_contours, lim.max_points, A, b.slot_kind, b.slot_ell, b.elig, b.n_elig, b.status);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return launch_pairs_tail(g, b, lim, p, pairs, s);
}

hipError_t launch_blobs(const Geom& g, const Bufs& b, const Limits& lim, const rmcv_params& p, hipStream_t s)
{
    return launch_fit(g, b, lim, p, false, s);
}

// filter_lightblobs + filter_armours: k_fit, then one k_pairs launch for both tails
hipError_t launch_blobs_armours(const Geom& g, const Bufs& b, const Limits& lim, const rmcv_params& p, hipStream_t s)
{
    return launch_fit(g, b, lim, p, true, s);
}

hipError_t launch_armours(const Geom& g, const Bufs& b, const Limits& lim, const rmcv_params& p, hipStream_t s)
{
    hipLaunchKernelGGL(k_armours, dim3(g.n_frames), dim3(64), 0, s, b.blobs, b.n_blobs, lim.max_blobs, p.angle_diff_max,
                       p.shear_max, p.length_ratio_max, p.camp, b.armours, b.n_armours, b.status, lim.max_armours);
    return hipGetLastError();
}

} // namespace rmcv
