// device_hull.h -- cv::minAreaRect (convex hull + rotating calipers), cv::boundingRect on int points and the camp vote of
// rm::FindLightBlobs (/root/reference/src/objdetect.cpp:43-51), each for one contour by one wavefront.
// The [OCV] algorithms are restated in oracle/rmcv_oracle_legacy.c; this is the wave64 formulation of the same results.
//
// Hull.  OpenCV sorts all points by (x, y, index) and runs Sklansky's scan over four quarter chains.  A closed 8-connected
// border visits every column of its bounding box, and only a column's extreme points can survive the scan, so the sort is
// replaced by four LDS words per column, filled with packed atomics in one pass over the points:
//     slot 0 (ymin, lowest index)   slot 1 (ymin, highest index)   slot 2 (ymax, lowest index)   slot 3 (ymax, highest index)
// (which duplicate of a point ends up in the hull depends on the scan direction, hence both indices).  Entry k = 4*column +
// slot is "sorted position k"; the scan over these 4W entries returns the same indices in the same order as the scan over
// all sorted points (oracle: orc_convex_hull_pruned, 116 k contours in tests/test_oracle_legacy.py's stress run).
// The four chains are four lanes of the wavefront running the scan side by side; everything after that is short and
// sequential (float rotating calipers over the <= few dozen hull vertices) and is executed wave-uniformly from LDS.
#pragma once
#include "device_fit.h"

namespace rmcv {

// Table sizes are run-time: the matcher first runs with small tables (most light-blob contours are a few dozen columns wide,
// their hulls a few dozen points) so that many wavefronts fit a CU, and repeats the rare contour that does not fit with
// the full-size ones.
static constexpr int HULL_CHAIN_CAP = 512; // full size: stack entries per quarter chain (a strictly convex lattice chain in a 4096 box has < 300)
static constexpr int HULL_CAP = 1024;      // full size: hull vertices (< 3.6 * 4096^(2/3) = 910)
static constexpr int HULL_MAX_DIM = 4096;  // y is packed into 12 bits, the point index into 20

struct HullLds {
    uint32_t* col;  // [4 * wcap]      packed (y << 20 | index code)
    uint16_t* stk;  // [4][ccap]       Sklansky stacks (entry numbers k)
    int32_t* hidx;  // [hcap]          hull as point indices, OpenCV's order
    float* hx;      // [hcap]
    float* hy;      // [hcap]
    float* inv;     // [hcap]          1/|edge|; doubles as the scratch of the cyclic shift
    int wcap, hcap, ccap;
};

__host__ __device__ inline size_t hull_lds_bytes(int wcap, int hcap, int ccap)
{
    return (size_t)4 * wcap * 4 + (size_t)4 * ccap * 2 + (size_t)hcap * 16;
}

__device__ inline void hull_lds_carve(unsigned char* base, int wcap, int hcap, int ccap, HullLds& H)
{
    H.wcap = wcap;
    H.hcap = hcap;
    H.ccap = ccap;
    H.col = reinterpret_cast<uint32_t*>(base);
    base += (size_t)4 * wcap * 4;
    H.hidx = reinterpret_cast<int32_t*>(base);
    H.hx = reinterpret_cast<float*>(base + (size_t)hcap * 4);
    H.hy = reinterpret_cast<float*>(base + (size_t)hcap * 8);
    H.inv = reinterpret_cast<float*>(base + (size_t)hcap * 12);
    H.stk = reinterpret_cast<uint16_t*>(base + (size_t)hcap * 16);
}

// LDS hand-over between the lanes of ONE wavefront (the functions below never synchronise across wavefronts)
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ int hull_entry_y(const uint32_t* col, int k) { return (int)(col[k] >> 20); }
__device__ __forceinline__ int hull_entry_idx(const uint32_t* col, int k)
{
    const uint32_t low = col[k] & 0xFFFFFu;
    const int slot = k & 3;
    return (int)((slot == 0 || slot == 3) ? low : 0xFFFFFu - low);
}

__device__ __forceinline__ int sgn_i64(long long v) { return (v > 0) - (v < 0); }

// [OCV] Sklansky_ over entries start..end of the column table (x of entry k is k >> 2).  One lane.
__device__ inline int hull_sklansky(const uint32_t* col, int start, int end, uint16_t* stack, int cap, int nsign, int sign2, int* ovf)
{
    const int incr = end > start ? 1 : -1;
    int pprev = start, pcur = pprev + incr, pnext = pcur + incr;
    int stacksize = 3;
    if (start == end || ((start >> 2) == (end >> 2) && hull_entry_y(col, start) == hull_entry_y(col, end))) {
        stack[0] = (uint16_t)start;
        return 1;
    }
    stack[0] = (uint16_t)pprev;
    stack[1] = (uint16_t)pcur;
    stack[2] = (uint16_t)pnext;
    end += incr;
    while (pnext != end) {
        const int cury = hull_entry_y(col, pcur), nexty = hull_entry_y(col, pnext);
        const int by = nexty - cury;
        if (((by > 0) - (by < 0)) != nsign) {
            const int ax = (pcur >> 2) - (pprev >> 2);
            const int bx = (pnext >> 2) - (pcur >> 2);
            const int ay = cury - hull_entry_y(col, pprev);
            const long long convexity = (long long)ay * bx - (long long)ax * by;
            if (sgn_i64(convexity) == sign2 && (ax != 0 || ay != 0)) {
                if (stacksize >= cap) { *ovf = 1; break; }
                pprev = pcur;
                pcur = pnext;
                pnext += incr;
                stack[stacksize] = (uint16_t)pnext;
                stacksize++;
            } else {
                if (pprev == start) {
                    pcur = pnext;
                    stack[1] = (uint16_t)pcur;
                    pnext += incr;
                    stack[2] = (uint16_t)pnext;
                } else {
                    stack[stacksize - 2] = (uint16_t)pnext;
                    pcur = pprev;
                    pprev = stack[stacksize - 4];
                    stacksize--;
                }
            }
        } else {
            pnext += incr;
            stack[stacksize - 1] = (uint16_t)pnext;
        }
    }
    return --stacksize;
}

// cv::convexHull(contour, clockwise = false) by one wavefront.  minx / W: the contour's column range.  Returns the number
// of hull points (wave-uniform); H.hidx[0..n) holds their indices into pts in OpenCV's order.  *ovf is set when a
// capacity of the LDS tables was exceeded (1; cannot happen for frames up to 4096 x 4096) or the points leave a column of
// their bounding box empty (2; never the case for a contour of findContours).
__device__ inline int hull_wave(const rmcv_point* __restrict__ pts, int n, int minx, int W, HullLds& H, int lane, int* ovf)
{
    const int total = 4 * W;
    for (int k = lane; k < total; k += 64) H.col[k] = (k & 2) ? 0u : 0xFFFFFFFFu;
    wave_lds_sync();
    for (int i = lane; i < n; i += 64) {
        const rmcv_point p = pts[i];
        uint32_t* c = H.col + 4 * (p.x - minx);
        const uint32_t y = (uint32_t)p.y << 20, lo = (uint32_t)i, hi = 0xFFFFFu - (uint32_t)i;
        atomicMin(&c[0], y | lo);
        atomicMin(&c[1], y | hi);
        atomicMax(&c[2], y | hi);
        atomicMax(&c[3], y | lo);
    }
    wave_lds_sync();
    // first entry with the largest / smallest y (OpenCV updates on strict comparisons while walking the sorted array)
    unsigned long long kmax = 0, kmin = ~0ull;
    bool hole = false;
    for (int k = lane; k < total; k += 64) {
        hole |= H.col[k] == 0xFFFFFFFFu; // slot 0/1 of a column no point fell into (y = 4095 with index 2^20-1 is out of range)
        const unsigned long long y = H.col[k] >> 20;
        const unsigned long long a = (y << 32) | (0xFFFFFFFFull - (unsigned)k), b = (y << 32) | (unsigned)k;
        kmax = a > kmax ? a : kmax;
        kmin = b < kmin ? b : kmin;
    }
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long a = __shfl_xor(kmax, d), b = __shfl_xor(kmin, d);
        kmax = a > kmax ? a : kmax;
        kmin = b < kmin ? b : kmin;
    }
    const int maxy_ind = (int)(0xFFFFFFFFull - (kmax & 0xFFFFFFFFull)), miny_ind = (int)(kmin & 0xFFFFFFFFull);
    if (__ballot(hole)) { // not a closed 8-connected border: the column table does not describe the sorted point set
        *ovf = 2;
        return 0;
    }
    if (W == 1 && hull_entry_y(H.col, 0) == hull_entry_y(H.col, total - 1)) { // all points coincide
        if (lane == 0) H.hidx[0] = hull_entry_idx(H.col, 0);
        wave_lds_sync();
        return 1;
    }
    // the four quarter chains side by side: lane 0 top-left, 1 top-right, 2 bottom-left, 3 bottom-right
    int cnt = 0, my_ovf = 0;
    if (lane < 4) {
        const int start = (lane & 1) ? total - 1 : 0;
        const int end = (lane & 2) ? miny_ind : maxy_ind;
        const int nsign = (lane & 2) ? 1 : -1;
        const int sign2 = (lane == 0 || lane == 3) ? 1 : -1;
        cnt = hull_sklansky(H.col, start, end, H.stk + lane * H.ccap, H.ccap, nsign, sign2, &my_ovf);
    }
    if (__ballot(my_ovf != 0)) {
        *ovf = 1;
        return 0;
    }
    wave_lds_sync();
    const int c_tl = __shfl(cnt, 0), c_tr = __shfl(cnt, 1);
    int c_bl = __shfl(cnt, 2), c_br = __shfl(cnt, 3);
    const uint16_t* TL = H.stk;
    const uint16_t* TR = H.stk + H.ccap;
    const uint16_t* BL = H.stk + 2 * H.ccap;
    const uint16_t* BR = H.stk + 3 * H.ccap;
    // counter-clockwise assembly: top-right chain forward, top-left chain backward, bottom-left forward, bottom-right backward
    const int stop_idx = c_tl > 2 ? TL[1] : c_tr > 2 ? TR[c_tr - 2] : -1;
    if (stop_idx >= 0) {
        const int check_idx = c_bl > 2 ? BL[1] : c_bl + c_br > 2 ? BR[2 - c_bl] : -1;
        if (check_idx == stop_idx || (check_idx >= 0 && (check_idx >> 2) == (stop_idx >> 2) &&
                                      hull_entry_y(H.col, check_idx) == hull_entry_y(H.col, stop_idx))) {
            c_bl = c_bl < 2 ? c_bl : 2; // all points on one line: the bottom part mirrors the top part
            c_br = c_br < 2 ? c_br : 2;
        }
    }
    const int na = c_tr > 1 ? c_tr - 1 : 0, nb = c_tl > 1 ? c_tl - 1 : 0, nc = c_bl > 1 ? c_bl - 1 : 0, nd = c_br > 1 ? c_br - 1 : 0;
    const int nout = na + nb + nc + nd;
    if (nout > H.hcap) {
        *ovf = 1;
        return 0;
    }
    for (int o = lane; o < nout; o += 64) {
        int k;
        if (o < na) k = TR[o];
        else if (o < na + nb) k = TL[c_tl - 1 - (o - na)];
        else if (o < na + nb + nc) k = BL[o - na - nb];
        else k = BR[c_br - 1 - (o - na - nb - nc)];
        H.hidx[o] = hull_entry_idx(H.col, k);
    }
    wave_lds_sync();
    // cyclic shift that makes the index sequence ascending or descending, when one exists (wave-uniform, sequential)
    if (nout >= 3) {
        int min_idx = 0, max_idx = 0, lt = 0;
        int prev = H.hidx[0], vmin = prev, vmax = prev;
        for (int i = 1; i < nout; i++) {
            const int idx = H.hidx[i];
            lt += prev < idx;
            if (lt > 1 && lt <= i - 2) break;
            if (idx < vmin) { vmin = idx; min_idx = i; }
            if (idx > vmax) { vmax = idx; max_idx = i; }
            prev = idx;
        }
        const int mmdist = max_idx > min_idx ? max_idx - min_idx : min_idx - max_idx;
        if ((mmdist == 1 || mmdist == nout - 1) && (lt <= 1 || lt >= nout - 2)) {
            const int ascending = (max_idx + 1) % nout == min_idx;
            const int i0 = ascending ? min_idx : max_idx;
            if (i0 > 0) {
                // every consecutive pair of the rotated sequence must step the same way
                bool bad = false;
                int32_t* tmp = reinterpret_cast<int32_t*>(H.inv);
                for (int i = lane; i < nout; i += 64) {
                    int j = i0 + i;
                    j = j >= nout ? j - nout : j;
                    const int nj = j + 1 < nout ? j + 1 : 0;
                    const int cur = H.hidx[j], nxt = H.hidx[nj];
                    tmp[i] = cur;
                    if (i < nout - 1 && (ascending != (cur < nxt))) bad = true;
                }
                const bool any_bad = __ballot(bad) != 0;
                wave_lds_sync();
                if (!any_bad)
                    for (int i = lane; i < nout; i += 64) H.hidx[i] = tmp[i];
                wave_lds_sync();
            }
        }
    }
    return nout;
}

// [OCV] rotatingCalipers(CALIPERS_MINAREARECT) over the hull points H.hx/hy[0..n), n >= 3; wave-uniform.
__device__ inline void calipers_wave(HullLds& H, int n, int lane, float out[6])
{
    // edge vectors are recomputed from the points (a float subtraction, exactly what OpenCV stores); 1/|edge| needs a
    // double sqrt and divide per edge and is tabulated by all lanes
    for (int i = lane; i < n; i += 64) {
        const int nx = i + 1 < n ? i + 1 : 0;
        const double dx = H.hx[nx] - H.hx[i], dy = H.hy[nx] - H.hy[i];
        H.inv[i] = (float)(1. / dsqrt(dx * dx + dy * dy));
    }
    // extreme points: first index reaching the extreme value (coordinates are small non-negative integers)
    unsigned kl = 0xFFFFFFFFu, kb = 0xFFFFFFFFu, kr = 0, kt = 0;
    for (int i = lane; i < n; i += 64) {
        const unsigned x = (unsigned)(int)H.hx[i], y = (unsigned)(int)H.hy[i];
        const unsigned lo = (unsigned)i, hi = 0xFFFFu - (unsigned)i;
        kl = min(kl, (x << 16) | lo);
        kb = min(kb, (y << 16) | lo);
        kr = max(kr, (x << 16) | hi);
        kt = max(kt, (y << 16) | hi);
    }
    for (int d = 32; d >= 1; d >>= 1) {
        kl = min(kl, (unsigned)__shfl_xor((int)kl, d));
        kb = min(kb, (unsigned)__shfl_xor((int)kb, d));
        kr = max(kr, (unsigned)__shfl_xor((int)kr, d));
        kt = max(kt, (unsigned)__shfl_xor((int)kt, d));
    }
    wave_lds_sync();
    int seq[4] = {(int)(kb & 0xFFFFu), (int)(0xFFFFu - (kr & 0xFFFFu)), (int)(0xFFFFu - (kt & 0xFFFFu)), (int)(kl & 0xFFFFu)};
    auto vx = [&](int i) { const int nx = i + 1 < n ? i + 1 : 0; return H.hx[nx] - H.hx[i]; };
    auto vy = [&](int i) { const int nx = i + 1 < n ? i + 1 : 0; return H.hy[nx] - H.hy[i]; };
    float orientation = 0;
    {
        double ax = vx(n - 1), ay = vy(n - 1);
        for (int i = 0; i < n; i++) {
            const double bx = vx(i), by = vy(i);
            const double convexity = ax * by - ay * bx;
            if (convexity != 0) {
                orientation = (convexity > 0) ? 1.f : (-1.f);
                break;
            }
            ax = bx;
            ay = by;
        }
    }
    float base_a = orientation, base_b = 0;
    float minarea = FLT_MAX;
    int buf_left = 0, buf_bottom = 0;
    float buf_a = 0, buf_b = 0, buf_w = 0, buf_h = 0;
    for (int k = 0; k < n; k++) {
        float ex[4], ey[4], il[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { ex[q] = vx(seq[q]); ey[q] = vy(seq[q]); il[q] = H.inv[seq[q]]; }
        const float dp0 = +base_a * ex[0] + base_b * ey[0];
        const float dp1 = -base_b * ex[1] + base_a * ey[1];
        const float dp2 = -base_a * ex[2] - base_b * ey[2];
        const float dp3 = +base_b * ex[3] - base_a * ey[3];
        float maxcos = dp0 * il[0];
        int main_element = 0;
        float cosalpha = dp1 * il[1];
        if (cosalpha > maxcos) { main_element = 1; maxcos = cosalpha; }
        cosalpha = dp2 * il[2];
        if (cosalpha > maxcos) { main_element = 2; maxcos = cosalpha; }
        cosalpha = dp3 * il[3];
        if (cosalpha > maxcos) { main_element = 3; maxcos = cosalpha; }
        const float mex = main_element == 0 ? ex[0] : main_element == 1 ? ex[1] : main_element == 2 ? ex[2] : ex[3];
        const float mey = main_element == 0 ? ey[0] : main_element == 1 ? ey[1] : main_element == 2 ? ey[2] : ey[3];
        const float mil = main_element == 0 ? il[0] : main_element == 1 ? il[1] : main_element == 2 ? il[2] : il[3];
        const float lead_x = mex * mil, lead_y = mey * mil;
        switch (main_element) {
        case 0: base_a = lead_x; base_b = lead_y; break;
        case 1: base_a = lead_y; base_b = -lead_x; break;
        case 2: base_a = -lead_x; base_b = -lead_y; break;
        default: base_a = -lead_y; base_b = lead_x; break;
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (q == main_element) { seq[q] += 1; seq[q] = (seq[q] == n) ? 0 : seq[q]; }
        float dx = H.hx[seq[1]] - H.hx[seq[3]];
        float dy = H.hy[seq[1]] - H.hy[seq[3]];
        const float width = dx * base_a + dy * base_b;
        dx = H.hx[seq[2]] - H.hx[seq[0]];
        dy = H.hy[seq[2]] - H.hy[seq[0]];
        const float height = -dx * base_b + dy * base_a;
        const float area = width * height;
        if (area <= minarea) {
            minarea = area;
            buf_left = seq[3];
            buf_a = base_a;
            buf_w = width;
            buf_b = base_b;
            buf_h = height;
            buf_bottom = seq[0];
        }
    }
    const float A1 = buf_a, B1 = buf_b, A2 = -buf_b, B2 = buf_a;
    const float C1 = A1 * H.hx[buf_left] + H.hy[buf_left] * B1;
    const float C2 = A2 * H.hx[buf_bottom] + H.hy[buf_bottom] * B2;
    const float idet = 1.f / (A1 * B2 - A2 * B1);
    out[0] = (C1 * B2 - C2 * B1) * idet;
    out[1] = (A1 * C2 - A2 * C1) * idet;
    out[2] = A1 * buf_w;
    out[3] = B1 * buf_w;
    out[4] = A2 * buf_h;
    out[5] = B2 * buf_h;
}

// cv::minAreaRect(contour) by one wavefront (wave-uniform result).  minx / W from the contour's bounding box.
__device__ inline void min_area_rect_wave(const rmcv_point* __restrict__ pts, int n, int minx, int W, HullLds& H, int lane,
                                          rmcv_rrect* box, int* ovf)
{
    box->cx = box->cy = box->w = box->h = box->angle = 0;
    const int nh = hull_wave(pts, n, minx, W, H, lane, ovf);
    for (int i = lane; i < nh; i += 64) {
        const rmcv_point p = pts[H.hidx[i]];
        H.hx[i] = (float)p.x;
        H.hy[i] = (float)p.y;
    }
    wave_lds_sync();
    if (nh > 2) {
        float out[6];
        calipers_wave(H, nh, lane, out);
        box->cx = out[0] + (out[2] + out[4]) * 0.5f;
        box->cy = out[1] + (out[3] + out[5]) * 0.5f;
        box->w = (float)dsqrt((double)out[2] * out[2] + (double)out[3] * out[3]);
        box->h = (float)dsqrt((double)out[4] * out[4] + (double)out[5] * out[5]);
        box->angle = (float)pm_atan2((double)out[3], (double)out[2]);
    } else if (nh == 2) {
        box->cx = (H.hx[0] + H.hx[1]) * 0.5f;
        box->cy = (H.hy[0] + H.hy[1]) * 0.5f;
        const double dx = H.hx[1] - H.hx[0], dy = H.hy[1] - H.hy[0];
        box->w = (float)dsqrt(dx * dx + dy * dy);
        box->h = 0;
        box->angle = (float)pm_atan2(dy, dx);
    } else if (nh == 1) {
        box->cx = H.hx[0];
        box->cy = H.hy[0];
    }
    box->angle = (float)(box->angle * 180 / RMCV_PI);
    wave_lds_sync(); // the tables are reused by the next contour
}

// objdetect.cpp:43-51: camp from the channel means over the contour's bounding rectangle.  cv::mean multiplies the
// integer channel sums by one positive factor (1/N), which preserves their order, so the sums are compared directly.
__device__ inline int camp_from_mean_wave(const uint8_t* __restrict__ frame, int stride, int minx, int miny, int W, int Hh, int lane)
{
    // one flat loop over the W x Hh pixels (rows are short: a per-row loop would wait for one load round trip per row)
    unsigned long long s0 = 0, s1 = 0, s2 = 0;
    const uint8_t* roi = frame + (int64_t)miny * stride + (int64_t)minx * 3;
    const int total = W * Hh, dy = 64 / W, dx = 64 - dy * W;
    int y = lane / W, x = lane - y * W;
#pragma unroll 4
    for (int i = lane; i < total; i += 64) {
        const uint8_t* px = roi + (int64_t)y * stride + 3 * x;
        s0 += px[0];
        s1 += px[1];
        s2 += px[2];
        y += dy;
        x += dx;
        if (x >= W) { x -= W; y++; }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        s0 += __shfl_xor(s0, d);
        s1 += __shfl_xor(s1, d);
        s2 += __shfl_xor(s2, d);
    }
    if (s1 > s0 && s1 > s2) return RMCV_CAMP_GUIDELIGHT;
    return s0 > s2 ? RMCV_CAMP_BLUE : RMCV_CAMP_RED;
}

} // namespace rmcv
