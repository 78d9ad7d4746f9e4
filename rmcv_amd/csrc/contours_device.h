// contours_device.h -- device code of cv::findContours on bit planes shared by the two builds of the per-frame sparse kernel
// (k_contours.hip: 8 wavefronts per frame, k_contours_w4.hip: 4).  See k_contours.hip for the method.
#pragma once
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include "rmcv_internal.h"
#include "wave_detect.h"

namespace rmcv {

__device__ __forceinline__ uint64_t ld_l2(const uint64_t* p)
{ // bypass the (non-coherent) vector L1: labels are written with L2 atomics by another lane
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// 64-bit window of row y starting at pixel xb (xb multiple of 32, may be -32 .. ) of a padded plane
__device__ __forceinline__ uint64_t win_load(const uint32_t* plane32, int prow, int y, int xb)
{
    const uint32_t* p = plane32 + ((int64_t)(y + 1) * prow + 1) * 2 + (xb >> 5);
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

struct Tracer {
    const uint32_t* F32;
    uint64_t* LAB;
    uint64_t* NEG;
    int prow;
    uint64_t r0, r1, r2; // rows y-1, y, y+1 of the window
    int xb;              // window origin (pixel), multiple of 32
    int x, y;

    __device__ __forceinline__ void recentre()
    {
        xb = ((x - 24) >> 5) * 32;
        r0 = win_load(F32, prow, y - 1, xb);
        r1 = win_load(F32, prow, y, xb);
        r2 = win_load(F32, prow, y + 1, xb);
    }
    // neighbour mask: bit s set <=> neighbour in direction s is foreground
    // s: 0=E 1=NE 2=N 3=NW 4=W 5=SW 6=S 7=SE  (y grows downward)
    __device__ __forceinline__ uint32_t nbmask() const
    {
        const int sh = x - xb - 1;
        uint32_t u = (uint32_t)(r0 >> sh) & 7u, m = (uint32_t)(r1 >> sh) & 7u, d = (uint32_t)(r2 >> sh) & 7u;
        return ((m >> 2) & 1u) | (((u >> 2) & 1u) << 1) | (((u >> 1) & 1u) << 2) | ((u & 1u) << 3) | ((m & 1u) << 4) |
               ((d & 1u) << 5) | (((d >> 1) & 1u) << 6) | (((d >> 2) & 1u) << 7);
    }
    __device__ __forceinline__ void move(int s)
    {
        const int dx = (s == 0 || s == 1 || s == 7) ? 1 : ((s >= 3 && s <= 5) ? -1 : 0);
        const int dy = (s >= 1 && s <= 3) ? -1 : ((s >= 5) ? 1 : 0);
        x += dx;
        y += dy;
        const int bx = x - xb;
        if (bx < 1 || bx > 62) {
            recentre();
        } else if (dy > 0) {
            r0 = r1; r1 = r2; r2 = win_load(F32, prow, y + 1, xb);
        } else if (dy < 0) {
            r2 = r1; r1 = r0; r0 = win_load(F32, prow, y - 1, xb);
        }
    }
    __device__ __forceinline__ void label(bool right_exit)
    {
        const int64_t idx = (int64_t)(y + 1) * prow + 1 + (x >> 6);
        const uint64_t bit = 1ull << (x & 63);
        atomicOr((unsigned long long*)(LAB + idx), (unsigned long long)bit);
        if (right_exit) atomicOr((unsigned long long*)(NEG + idx), (unsigned long long)bit);
    }
};

// icvFetchContour (outer border, CHAIN_APPROX_NONE) from start (x0,y0).  Writes at most `room` points
// to out (keeps tracing and labelling beyond that), returns the number of points; *ymax = lowest row.
__device__ int trace_border(Tracer& t, int x0, int y0, rmcv_point* out, int room, int* ymax)
{
    t.x = x0;
    t.y = y0;
    t.recentre();
    int n = 0, ym = y0;
    uint32_t nb = t.nbmask();
    // clockwise search for the first neighbour, starting just past west: s = 3,2,1,0,7,6,5,(4)
    int s = 4;
    do { s = (s - 1) & 7; } while (!((nb >> s) & 1u) && s != 4);
    if (s == 4) { // single pixel (west is background by construction)
        t.label(true);
        if (n < room) { out[n].x = x0; out[n].y = y0; }
        *ymax = y0;
        return 1;
    }
    const int dx1 = (s == 0 || s == 1 || s == 7) ? 1 : ((s >= 3 && s <= 5) ? -1 : 0);
    const int dy1 = (s >= 1 && s <= 3) ? -1 : ((s >= 5) ? 1 : 0);
    const int x1 = x0 + dx1, y1 = y0 + dy1; // i1
    for (;;) {
        const int s_end = s;
        const int k = (s_end + 1) & 7;
        const uint32_t rot = ((nb | (nb << 8)) >> k) & 0xFFu;
        s = (k + (__ffs((int)rot) - 1)) & 7; // counter-clockwise sweep from s_end+1 to the first foreground neighbour
        t.label((unsigned)(s - 1) < (unsigned)s_end);
        if (n < room) { out[n].x = t.x; out[n].y = t.y; }
        n++;
        const int cx = t.x, cy = t.y;
        t.move(s);
        if (t.y > ym) ym = t.y;
        if (t.x == x0 && t.y == y0 && cx == x1 && cy == y1) break; // i4 == i0 && i3 == i1
        nb = t.nbmask();
        s = (s + 4) & 7;
    }
    *ymax = ym;
    return n;
}

// first acceptable border-start candidate of row y at x >= xmin, or -1
__device__ int scan_row(const uint64_t* F, const uint64_t* LAB, const uint64_t* NEG, int prow, int ww, int y, int xmin,
                        uint64_t occ)
{
    const int64_t base = (int64_t)(y + 1) * prow + 1;
    uint64_t carry = 0;
    bool last_pos = false; // no labelled pixel yet -> lnbd is the zero frame column -> accept
    for (int k = 0; k < ww; k++) {
        if (k < 64 && !((occ >> k) & 1ull)) { carry = 0; continue; }
        const uint64_t f = F[base + k];
        if (f == 0) { carry = 0; continue; }
        const uint64_t l = ld_l2(LAB + base + k), ng = ld_l2(NEG + base + k);
        uint64_t cand = f & ~((f << 1) | carry) & ~l;
        if (k * 64 + 63 < xmin) cand = 0;
        else if (k * 64 < xmin) cand &= ~0ull << (xmin - k * 64);
        while (cand) {
            const int b = __ffsll((long long)cand) - 1;
            cand &= cand - 1;
            const uint64_t below = l & ((1ull << b) - 1);
            const bool pos = below ? !((ng >> (63 - __clzll((long long)below))) & 1ull) : last_pos;
            if (!pos) return k * 64 + b;
        }
        if (l) last_pos = !((ng >> (63 - __clzll((long long)l))) & 1ull);
        carry = f >> 63;
    }
    return -1;
}

// The literal scanner for one frame, run by ONE wavefront (lane = 0..63).  Exact for every input; it is the
// fallback of k_contours for frames with nested components.  LAB/NEG must be zero on entry.
__device__ void literal_frame(int lane, const uint64_t* F, uint64_t* LAB, uint64_t* NEG, int h, int ww, int prow, rmcv_point* pts,
                              int32_t* cs, int32_t* cl, int max_contours, int max_points, const uint32_t* rowmask,
                              int* nc_out, int* np_out, int* st_out)
{
    Tracer t;
    t.F32 = reinterpret_cast<const uint32_t*>(F);
    t.LAB = LAB;
    t.NEG = NEG;
    t.prow = prow;
    int nc = 0, np = 0, st = 0; // wave-uniform
    for (int band = 0; band * 64 < h; band++) {
        const int y = band * 64 + lane;
        const uint64_t occ = (rowmask && y < h) ? (uint64_t)rowmask[y] : ~0ull;
        bool done = y >= h || occ == 0, dirty = true;
        int xmin = 0, found = -1;
        for (;;) {
            if (dirty && !done) {
                found = scan_row(F, LAB, NEG, prow, ww, y, xmin, occ);
                dirty = false;
            }
            const uint64_t m = __ballot(!done && found >= 0);
            if (!m) break;
            const int L = __ffsll((long long)m) - 1;
            const int x0 = __shfl(found, L), y0 = band * 64 + L;
            if (lane < L) done = true; // rows above the start are behind the raster scan
            int len = 0, ymax = y0;
            if (lane == 0) {
                const int room = (nc < max_contours && np < max_points) ? (max_points - np) : 0;
                len = trace_border(t, x0, y0, pts + np, room, &ymax);
                if (nc < max_contours) { cs[nc] = np; cl[nc] = len; }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // label atomics performed before the re-scan
            }
            len = __shfl(len, 0);
            ymax = __shfl(ymax, 0);
            if (nc >= max_contours) st |= RMCV_FRAME_OVF_CONTOURS;
            if (np + len > max_points) { st |= RMCV_FRAME_OVF_POINTS; }
            nc++;
            np = (np + len > max_points) ? max_points : np + len;
            if (lane == L) { xmin = x0 + 1; dirty = true; }
            else if (lane > L && y <= ymax) dirty = true;
        }
    }
    *nc_out = nc < max_contours ? nc : max_contours;
    *np_out = np;
    *st_out = st;
}

// ---- wave-cooperative border following ------------------------------------------------------------------
// The 64 lanes of a wavefront hold a 64-row x 64-column window of F (lane i = row wy0+i) in registers; the walk
// itself is wave-uniform scalar work that fetches the three rows it needs with v_readlane -- no memory access per
// step.  The window is re-centred (one load per lane) when the walk leaves it.
struct WWin {
    uint64_t fw;
    int xb, wy0;
};

__device__ __forceinline__ uint64_t rl64(uint64_t v, int l)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ void wwin_load(WWin& W, const uint32_t* F32, int prow, int h, int x, int y, int lane)
{
    W.xb = ((x - 24) >> 5) * 32;
    W.wy0 = y - 12;
    const int r = W.wy0 + lane;
    W.fw = (r >= -1 && r <= h) ? win_load(F32, prow, r, W.xb) : 0ull;
}

__device__ __forceinline__ uint32_t nbmask3(uint64_t r0, uint64_t r1, uint64_t r2, int sh)
{
    const uint32_t u = (uint32_t)(r0 >> sh) & 7u, m = (uint32_t)(r1 >> sh) & 7u, d = (uint32_t)(r2 >> sh) & 7u;
    return ((m >> 2) & 1u) | (((u >> 2) & 1u) << 1) | (((u >> 1) & 1u) << 2) | ((u & 1u) << 3) | ((m & 1u) << 4) |
           ((d & 1u) << 5) | (((d >> 1) & 1u) << 6) | (((d >> 2) & 1u) << 7);
}

// ---- table-driven walk + parallel replay -----------------------------------------------------------------
// One border-following step is a pure function of (direction back to the previous pixel, 3x3 neighbourhood):
// a 4096-entry byte table in LDS, index = s_back<<9 | up<<6 | mid<<3 | down (each 3 bits: x-1, x, x+1), value =
// s_new | right_exit<<3 | (dx+1)<<4 | (dy+1)<<6.  The walk records a 4-bit code per step in lane registers (lane
// n>>5 holds steps 32*(n>>5)..+31), so a kept contour is not walked twice: the codes are replayed by all lanes in
// parallel (prefix sum of the per-lane displacements) to write the points and the labels.
static constexpr int CODE_CAP = 2048; // steps recorded per contour (64 lanes x 128 bits / 4); longer ones are re-walked

__device__ __forceinline__ int dir_dx(int s) { return (s == 0 || s == 1 || s == 7) ? 1 : ((s >= 3 && s <= 5) ? -1 : 0); }
__device__ __forceinline__ int dir_dy(int s) { return (s >= 1 && s <= 3) ? -1 : ((s >= 5) ? 1 : 0); }

__device__ void lut_build(uint8_t* lut, int tid, int nthreads)
{
    for (int idx = tid; idx < 4096; idx += nthreads) {
        const int s_end = idx >> 9;
        const uint32_t u = (idx >> 6) & 7, m = (idx >> 3) & 7, d = idx & 7;
        const uint32_t nb = ((m >> 2) & 1u) | (((u >> 2) & 1u) << 1) | (((u >> 1) & 1u) << 2) | ((u & 1u) << 3) | ((m & 1u) << 4) |
                            ((d & 1u) << 5) | (((d >> 1) & 1u) << 6) | (((d >> 2) & 1u) << 7);
        uint8_t e = 0;
        if (nb) {
            const int k = (s_end + 1) & 7;
            const uint32_t rot = ((nb | (nb << 8)) >> k) & 0xFFu;
            const int sn = (k + (__ffs((int)rot) - 1)) & 7;
            const int rex = ((unsigned)(sn - 1) < (unsigned)s_end) ? 1 : 0;
            e = (uint8_t)(sn | (rex << 3) | ((dir_dx(sn) + 1) << 4) | ((dir_dy(sn) + 1) << 6));
        }
        lut[idx] = e;
    }
}

// Walk the border from (x0,y0) without writing anything to memory.  Returns the number of points; *state:
// 0 = complete, 1 = single pixel, 2 = aborted (a raster-earlier pixel was met: (x0,y0) is not a first pixel).
__device__ int walk_record(WWin& W, const uint32_t* F32, int prow, int h, int x0, int y0, int lane, const uint8_t* lut,
                           int* state, uint64_t* c0_out, uint64_t* c1_out)
{
    const uint32_t key0 = ((uint32_t)y0 << 16) | (uint32_t)x0;
    int x = x0, y = y0, n = 0;
    uint64_t c0 = 0, c1 = 0;
    *state = 0;
    int s_back;
    int x1, y1;
    {
        const int ly = y - W.wy0;
        const uint32_t nb = nbmask3(rl64(W.fw, ly - 1), rl64(W.fw, ly), rl64(W.fw, ly + 1), x - W.xb - 1);
        int s = 4;
        do { s = (s - 1) & 7; } while (!((nb >> s) & 1u) && s != 4);
        if (s == 4) { // single pixel: one step, no move, negative label (icvFetchContour's isolated-pixel case)
            *state = 1;
            *c0_out = (lane == 0) ? 8ull : 0ull;
            *c1_out = 0;
            return 1;
        }
        x1 = x0 + dir_dx(s);
        y1 = y0 + dir_dy(s);
        s_back = s;
    }
    for (;;) {
        const int ly = y - W.wy0, sh = x - W.xb - 1;
        const uint32_t u = (uint32_t)(rl64(W.fw, ly - 1) >> sh) & 7u, m = (uint32_t)(rl64(W.fw, ly) >> sh) & 7u,
                       d = (uint32_t)(rl64(W.fw, ly + 1) >> sh) & 7u;
        const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)lut[(s_back << 9) | (u << 6) | (m << 3) | d]);
        if ((((uint32_t)y << 16) | (uint32_t)x) < key0) { *state = 2; break; }
        // ---- straight vertical runs in one go.  In the steady states "came from the north, go on south" and "came from
        // the south, go on north" every further pixel of the column is judged with the SAME back direction, so whether
        // it repeats this very step is a function of its own 3x3 neighbourhood: all 64 window rows evaluate the table
        // at once (rows above/below via lane shuffles), a ballot gives the length of the run, and its steps are
        // recorded together.  (Bars are mostly vertical edges: this removes most of the sequential steps.)
        int extra = 0;
        const int sdir = (int)(e & 7u);
        if ((sdir == 6 && s_back == 2) || (sdir == 2 && s_back == 6 && x != x0)) {
            const uint64_t up = __shfl_up(W.fw, 1), dn = __shfl_down(W.fw, 1);
            const uint32_t ur = (uint32_t)(up >> sh) & 7u, mr = (uint32_t)(W.fw >> sh) & 7u, dr = (uint32_t)(dn >> sh) & 7u;
            const bool same = lane >= 1 && lane <= 62 && (uint32_t)lut[(s_back << 9) | (ur << 6) | (mr << 3) | dr] == e;
            const uint64_t okm = __ballot(same);
            if (sdir == 6) { // rows below the current one
                const uint64_t t = ly < 63 ? okm >> (ly + 1) : 0ull;
                extra = (~t) ? __ffsll((long long)~t) - 1 : 64;
            } else {         // rows above
                const uint64_t t = ly > 0 ? okm << (64 - ly) : 0ull;
                extra = (~t) ? __clzll((long long)~t) : 64;
                // the raster-smallest pixel of the run is its top end
                if (extra > 0 && ((((uint32_t)(y - extra)) << 16) | (uint32_t)x) < key0) { *state = 2; break; }
            }
        }
        { // record the step(s) in the lanes that own them -- branch-free (selects, no EXEC change in the hot loop)
            const int a = n - 32 * lane, b = n + extra + 1 - 32 * lane; // this lane owns steps [0,32) of [a,b)
            const int lo = a < 0 ? 0 : a, hi = b > 32 ? 32 : b;
            const bool any = lo < hi && n + extra < CODE_CAP;
            const uint64_t pat = 0x1111111111111111ull * (uint64_t)(e & 15u);
            // nibble ranges [lo,hi) split over the two 16-step registers
            const int l0 = lo < 16 ? lo : 16, h0 = hi < 16 ? hi : 16, l1 = lo > 16 ? lo - 16 : 0, h1 = hi > 16 ? hi - 16 : 0;
            const uint64_t m0 = (h0 >= 16 ? ~0ull : ((1ull << (4 * h0)) - 1)) & ~((1ull << (4 * l0)) - 1);
            const uint64_t m1 = (h1 >= 16 ? ~0ull : ((1ull << (4 * h1)) - 1)) & ~((1ull << (4 * l1)) - 1);
            c0 |= (any && l0 < h0) ? (pat & m0) : 0ull;
            c1 |= (any && l1 < h1) ? (pat & m1) : 0ull;
        }
        n += 1 + extra;
        const int cx = x, cy = y + extra * ((int)((e >> 6) & 3u) - 1);
        x += (int)((e >> 4) & 3u) - 1;
        y += (1 + extra) * ((int)((e >> 6) & 3u) - 1);
        if (x == x0 && y == y0 && cx == x1 && cy == y1) break; // i4 == i0 && i3 == i1
        if (n >= (1 << 22)) break;                             // cannot happen on a consistent plane
        const int wx = x - W.xb, wy = y - W.wy0;
        if (wx < 1 || wx > 62 || wy < 1 || wy > 62) wwin_load(W, F32, prow, h, x, y, lane);
        s_back = ((int)(e & 7u) + 4) & 7;
    }
    *c0_out = c0;
    *c1_out = c1;
    return n;
}

__device__ __forceinline__ int wave_excl_scan_i32(int v, int lane)
{
    int inc = v;
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(inc, d);
        if (lane >= d) inc += t;
    }
    return inc - v;
}

// Sparse label store of the fast path: one LDS slot per NON-EMPTY word of F (labels only exist where F is set).
// slot(y, k) = rowbase[y] + popcount(rowmask[y] & ((1 << k) - 1)).
struct LabelStore {
    const uint32_t* rowmask;
    const uint16_t* rowbase;
    unsigned long long* lab;
    unsigned long long* neg;
    __device__ __forceinline__ int slot(int y, int k) const { return rowbase[y] + __popc(rowmask[y] & ((1u << k) - 1u)); }
};

// all lanes replay the recorded steps: points -> out[0..n), labels -> the LDS label store (n <= CODE_CAP).
// The codes sit 32 per lane (lane n>>5); the replay spreads them over all 64 lanes, spl = 1, 2, 4 .. 32 consecutive
// steps per lane (each lane fetches its owner's code registers with a shuffle), so a 150-point contour costs 4
// dependent LDS round trips per lane instead of 32.
__device__ void replay_emit(int n, int x0, int y0, int lane, uint64_t c0, uint64_t c1, rmcv_point* out, const LabelStore& LS)
{
    int spl = 1;
    while (spl * 64 < n) spl <<= 1; // <= 32 because n <= CODE_CAP
    const int first = lane * spl;
    int cnt = n - first;
    cnt = cnt < 0 ? 0 : (cnt > spl ? spl : cnt);
    const int owner = (first >> 5) & 63, sub = first & 31;
    const uint64_t o0 = __shfl(c0, owner), o1 = __shfl(c1, owner);
    // the lane's codes, 4 bits each, starting at bit 0 (sub is a multiple of spl, so the run never straddles c0/c1
    // unless spl == 32, where sub == 0)
    const uint64_t lo = sub < 16 ? (o0 >> (4 * sub)) : (o1 >> (4 * (sub - 16)));
    const uint64_t hi = o1; // only used when spl == 32 (steps 16..31)
    int dx = 0, dy = 0;
    for (int j = 0; j < cnt; j++) {
        const int sdir = (int)((j < 16 ? lo >> (4 * j) : hi >> (4 * (j - 16))) & 7u);
        dx += dir_dx(sdir);
        dy += dir_dy(sdir);
    }
    int x = x0 + wave_excl_scan_i32(dx, lane), y = y0 + wave_excl_scan_i32(dy, lane);
    int pslot = -1, py = -1, pk = -1; // pending label word
    unsigned long long plab = 0, pneg = 0;
    for (int j = 0; j < cnt; j++) {
        const uint32_t code = (uint32_t)(j < 16 ? lo >> (4 * j) : hi >> (4 * (j - 16))) & 15u;
        rmcv_point p;
        p.x = x;
        p.y = y;
        out[first + j] = p;
        if (y != py || (x >> 6) != pk) {
            if (plab) atomicOr(LS.lab + pslot, plab);
            if (pneg) atomicOr(LS.neg + pslot, pneg);
            py = y;
            pk = x >> 6;
            pslot = LS.slot(y, pk);
            plab = 0;
            pneg = 0;
        }
        plab |= 1ull << (x & 63);
        if (code & 8u) pneg |= 1ull << (x & 63);
        x += dir_dx((int)(code & 7u));
        y += dir_dy((int)(code & 7u));
    }
    if (plab) atomicOr(LS.lab + pslot, plab);
    if (pneg) atomicOr(LS.neg + pslot, pneg);
}

// ---- k_contours: one workgroup (8 wavefronts) per frame ---------------------------------------------------
//  T  every thread scans non-empty rows for LOCAL TOPS (run starts whose run touches nothing in the row above):
//     the raster-first pixel of every 8-connected component is one of them
//  S  wavefronts pull tops from a queue and walk them; a walk that meets no raster-earlier pixel started at the
//     first pixel of a component and followed its outer border -> kept, replayed into points + labels.
//     Walks are independent, so all components of a frame are followed concurrently.
//  V  verification of OpenCV's RETR_EXTERNAL bookkeeping on the merged labels: every kept start must have been
//     accepted (nearest labelled pixel to its left negative or absent) and every other unlabelled run start
//     rejected.  True for frames without nested components; then discovery order = raster order of the starts.
//  F  otherwise the frame is redone by the literal scanner (exact for every input) on the global LAB/NEG planes,
//     which are zero between launches (the literal path clears what it set).
// Non-empty rows/words come from the row masks k_binary writes next to the bit plane (bit k of rowmask[y] = word k
// of row y is non-zero; a superset is fine).
// LDS budget: the workgroup shares its CU with the pixel kernels of the next two batches (2 x 2 x 11.5 KB) and with other
// frames' workgroups, so the tables are sized for ~50 KB (3 per CU); measured +4-8 % on the 3-stream bench against 75 KB.
// Frames beyond a capacity take the literal path (tests/test_gpu_parity.py covers each limit).
static constexpr int CAND_CAP = 1024;
static constexpr int KEPT_CAP = 512;
static constexpr int SLOT_CAP = 1024;  // non-empty words of a frame the LDS label store can hold
// Workgroup size is a template parameter: 8 wavefronts walk and fit the bars of a frame concurrently (lowest latency for one
// batch), 4 wavefronts leave room on every CU for the pixel kernels of the next batches AND the sparse kernel of the previous
// one (VGPR budget per SIMD: 2 x 160 for this kernel at 8 wavefronts, 2 x 80 per pixel kernel, 512 in all) -- the better choice
// when several batches are in flight (886 k against 800 k frames/s with three batches; 0.57 against 0.475 ms for a lone batch).
static constexpr int CT_THREADS_MAX = 512;
static constexpr int CT_MAXH = 2048;   // rows covered by the LDS row tables (taller/wider frames take the literal path)

struct ContoursLds {
    unsigned long long lab[SLOT_CAP], neg[SLOT_CAP];
    uint32_t rowmask[CT_MAXH];
    uint32_t cand[CAND_CAP];
    uint32_t kkey[KEPT_CAP];
    int32_t koff[KEPT_CAP], klen[KEPT_CAP];
    uint16_t rows[CT_MAXH], rowbase[CT_MAXH];
    int scan[CT_THREADS_MAX];
    uint8_t lut[4096];
    int ncand, next, nkept, cursor, flags, nrows, nslots, nelig, lit[3];
    int dummy[64]; // per-lane sinks: lanes != 0 add 0 here so that a wave-wide atomic does not serialise on one word
};

} // namespace rmcv
