// k_contours.hip -- cv::findContours(binary, RETR_EXTERNAL, CHAIN_APPROX_NONE), the last step of
// rm::extract_color (/root/reference/src/imgproc.cpp:71-72), on the bit plane K1 leaves in HBM.
//
// OpenCV's Suzuki-Abe scanner is a sequential raster scan that rewrites the image with labels while it
// traces; its RETR_EXTERNAL rule ("skip an outer-border start if the last labelled pixel met on this
// row is positive") makes the result depend on that label state.  Restated on bit planes:
//   F    foreground (closed binary)                       read-only
//   LAB  pixel has been visited by a border trace         (label != 0, 1)
//   NEG  ... and at least one visit passed the east neighbour as zero (label 2|-128 instead of 2)
// A run start (F[x]=1, F[x-1]=0) that is unlabelled is a border-start candidate; it is accepted iff
// the nearest labelled pixel to its left on the row is NEG or there is none.  Only run starts and
// labelled pixels matter, so a row is scanned 64 pixels per operation.
//
// k_contours_literal: one wavefront per frame.  Lane l owns row 64*band + l of the current band and
// keeps the first acceptable candidate of its row; the wave repeatedly takes the raster-first one,
// traces it (border following on a 3-row x 64-bit register window of F), ORs the labels into
// LAB/NEG, and re-scans only the rows the trace touched below the start.  Exact for every input.
//
// Output is kept in DISCOVERY order (cont_start/cont_len); findContours order is its reverse.
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
static constexpr int CT_THREADS = 512; // 8 wavefronts: the bars of a frame are walked concurrently
static constexpr int CT_MAXH = 2048;   // rows covered by the LDS row tables (taller/wider frames take the literal path)

struct ContoursLds {
    unsigned long long lab[SLOT_CAP], neg[SLOT_CAP];
    uint32_t rowmask[CT_MAXH];
    uint32_t cand[CAND_CAP];
    uint32_t kkey[KEPT_CAP];
    int32_t koff[KEPT_CAP], klen[KEPT_CAP];
    uint16_t rows[CT_MAXH], rowbase[CT_MAXH];
    int scan[CT_THREADS];
    uint8_t lut[4096];
    int ncand, next, nkept, cursor, flags, nrows, nslots, nelig, lit[3];
    int dummy[64]; // per-lane sinks: lanes != 0 add 0 here so that a wave-wide atomic does not serialise on one word
};

__global__ __launch_bounds__(CT_THREADS) void k_contours(const uint64_t* __restrict__ bits, const uint32_t* __restrict__ rowmasks,
                                                        int rm_pitch, uint64_t* lab, uint64_t* neg, int w, int h, int ww, int prow,
                                                        int64_t plane_pitch, rmcv_point* points, int32_t* cont_start,
                                                        int32_t* cont_len, int32_t* n_contours, int32_t* n_points,
                                                        int32_t* status, int max_contours, int max_points, int force_literal,
                                                        int32_t* __restrict__ elig, int32_t* __restrict__ n_elig,
                                                        int32_t* __restrict__ slot_kind, SparseTail X)
{
    extern __shared__ unsigned long long smem_raw[];
    __builtin_amdgcn_s_setprio(3); // latency-bound: issue ahead of the streaming pixel kernel of the next batch sharing the CU
    ContoursLds& S = *reinterpret_cast<ContoursLds*>(smem_raw);
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t* F = bits + (int64_t)f * plane_pitch;
    const uint32_t* F32 = reinterpret_cast<const uint32_t*>(F);
    uint64_t* LAB = lab + (int64_t)f * plane_pitch;
    uint64_t* NEG = neg + (int64_t)f * plane_pitch;
    rmcv_point* pts = points + (int64_t)f * max_points;
    int32_t* cs = cont_start + (int64_t)f * max_contours;
    int32_t* cl = cont_len + (int64_t)f * max_contours;
    enum { FL_COMPLEX = 1 };
#ifdef RMCV_PROFILE
    long long t_[8]; int ti_ = 0;
#define STAMP() do { __syncthreads(); t_[ti_++] = wall_clock64(); } while (0)
#else
#define STAMP() do {} while (0)
#endif
    STAMP();
    const bool summarised = (h <= CT_MAXH && ww <= 32);
    if (tid == 0) {
        S.ncand = 0; S.next = 0; S.nkept = 0; S.cursor = 0; S.nrows = 0; S.nslots = 0; S.nelig = 0;
        S.flags = (force_literal || !summarised) ? FL_COMPLEX : 0;
    }
    lut_build(S.lut, tid, CT_THREADS);
    // ---------------- row tables: masks from k_binary, slot bases by a workgroup prefix sum, list of non-empty rows
    if (summarised) {
        const int per = (h + CT_THREADS - 1) / CT_THREADS; // consecutive rows per thread
        int cnt = 0;
        for (int u = 0; u < per; u++) {
            const int y = tid * per + u;
            const uint32_t m = y < h ? rowmasks[(int64_t)f * rm_pitch + y] : 0u;
            if (y < h) S.rowmask[y] = m;
            cnt += __popc(m);
        }
        S.scan[tid] = cnt;
        __syncthreads();
        for (int d = 1; d < CT_THREADS; d <<= 1) { // Hillis-Steele inclusive scan
            const int v = tid >= d ? S.scan[tid - d] : 0;
            __syncthreads();
            S.scan[tid] += v;
            __syncthreads();
        }
        int base = S.scan[tid] - cnt;
        for (int u = 0; u < per; u++) {
            const int y = tid * per + u;
            if (y < h) {
                const uint32_t m = S.rowmask[y];
                S.rowbase[y] = (uint16_t)(base < 65535 ? base : 65535);
                base += __popc(m);
                if (m) S.rows[atomicAdd(&S.nrows, 1)] = (uint16_t)y;
            }
        }
        if (tid == CT_THREADS - 1) {
            S.nslots = S.scan[tid];
            if (S.scan[tid] > SLOT_CAP) S.flags |= FL_COMPLEX;
        }
    }
    __syncthreads();
    const int nrows = S.nrows;
    if (!(S.flags & FL_COMPLEX))
        for (int i = tid; i < S.nslots; i += CT_THREADS) { S.lab[i] = 0; S.neg[i] = 0; }
    LabelStore LS;
    LS.rowmask = S.rowmask;
    LS.rowbase = S.rowbase;
    LS.lab = S.lab;
    LS.neg = S.neg;
    STAMP();

    // ---------------- T: local tops
    if (!(S.flags & FL_COMPLEX))
        for (int r = tid; r < nrows; r += CT_THREADS) {
            const int y = S.rows[r];
            const int64_t base = (int64_t)(y + 1) * prow + 1, up = base - prow;
            const uint32_t occ = S.rowmask[y], occ_up = y > 0 ? S.rowmask[y - 1] : 0u;
            bool in_run = false, touched = false;
            int run_x = 0;
            for (int k = 0; k < ww; k++) {
                const uint64_t fwd = ((occ >> k) & 1u) ? F[base + k] : 0ull;
                if (!fwd && !in_run) continue;
                uint64_t ad = 0;
                if ((occ_up >> (k > 0 ? k - 1 : 0)) & (k > 0 ? 7u : 3u)) { // anything above in words k-1..k+1
                    const uint64_t a = F[up + k];
                    ad = a | (a << 1) | (F[up + k - 1] >> 63) | (a >> 1) | (F[up + k + 1] << 63);
                }
                const uint64_t touch = fwd & ad;
                uint64_t rem = fwd;
                if (in_run) {
                    const int lead = (~fwd) ? __ffsll((long long)~fwd) - 1 : 64;
                    const uint64_t mask = lead == 64 ? ~0ull : ((1ull << lead) - 1);
                    touched |= (touch & mask) != 0;
                    if (lead == 64) continue;
                    if (!touched) {
                        const int i = atomicAdd(&S.ncand, 1);
                        if (i < CAND_CAP) S.cand[i] = ((uint32_t)y << 16) | (uint32_t)run_x;
                    }
                    in_run = false;
                    rem = fwd & ~mask;
                }
                while (rem) {
                    const int st = __ffsll((long long)rem) - 1;
                    const uint64_t t = rem >> st;
                    const int len = (~t) ? __ffsll((long long)~t) - 1 : 64;
                    const uint64_t mask = (len == 64 ? ~0ull : ((1ull << len) - 1)) << st;
                    const bool tch = (touch & mask) != 0;
                    if (st + len == 64) { in_run = true; touched = tch; run_x = k * 64 + st; break; }
                    if (!tch) {
                        const int i = atomicAdd(&S.ncand, 1);
                        if (i < CAND_CAP) S.cand[i] = ((uint32_t)y << 16) | (uint32_t)(k * 64 + st);
                    }
                    rem &= ~mask;
                }
            }
            if (in_run && !touched) {
                const int i = atomicAdd(&S.ncand, 1);
                if (i < CAND_CAP) S.cand[i] = ((uint32_t)y << 16) | (uint32_t)run_x;
            }
        }
    __syncthreads();
    STAMP();
    const int ncand = S.ncand;
    if (ncand > CAND_CAP && tid == 0) S.flags |= FL_COMPLEX;
    __syncthreads();

    // ---------------- S: speculative walks, one wavefront per candidate
    if (!(S.flags & FL_COMPLEX)) {
        WWin W;
        for (;;) {
            // every lane issues the LDS atomic (lanes != 0 add 0), so the loop control stays wave-uniform
            const int i = __builtin_amdgcn_readfirstlane(atomicAdd(lane == 0 ? &S.next : &S.dummy[lane], lane == 0 ? 1 : 0));
            if (i >= ncand) break;
            const uint32_t key0 = S.cand[i];
            const int x0 = (int)(key0 & 0xFFFFu), y0 = (int)(key0 >> 16);
            int state = 0;
            uint64_t c0 = 0, c1 = 0;
#ifdef RMCV_PROFILE
            const long long tp0 = wall_clock64();
#endif
            wwin_load(W, F32, prow, h, x0, y0, lane);
#ifdef RMCV_PROFILE
            const long long tp1 = wall_clock64();
#endif
            const int len = walk_record(W, F32, prow, h, x0, y0, lane, S.lut, &state, &c0, &c1);
#ifdef RMCV_PROFILE
            const long long tp2 = wall_clock64();
            if (f == 0 && lane == 0 && state == 2) printf("[cand f0 w%d i%d] aborted len=%d load %.1f walk %.1f us\n", wave, i, len, (tp1 - tp0) / 100.0, (tp2 - tp1) / 100.0);
#endif
            if (state == 2) continue; // not the first pixel of its component (or a hole border)
            const int off = __builtin_amdgcn_readfirstlane(atomicAdd(lane == 0 ? &S.cursor : &S.dummy[lane], lane == 0 ? len : 0));
            const int slot = __builtin_amdgcn_readfirstlane(atomicAdd(lane == 0 ? &S.nkept : &S.dummy[lane], lane == 0 ? 1 : 0));
            if (slot >= KEPT_CAP || slot >= max_contours || off + len > max_points || len > CODE_CAP) {
                // capacity, or a contour longer than the code registers (> 2048 points): the literal scanner
                // redoes the frame and reports an overflow exactly
                atomicOr(&S.flags, FL_COMPLEX);
                continue;
            }
            replay_emit(len, x0, y0, lane, c0, c1, pts + off, LS);
#ifdef RMCV_PROFILE
            if (f == 0 && lane == 0) printf("[cand f0 w%d i%d] len=%d load %.1f walk %.1f replay %.1f us (t=%.1f)\n", wave, i, len, (tp1 - tp0) / 100.0, (tp2 - tp1) / 100.0, (wall_clock64() - tp2) / 100.0, (tp0 - t_[2]) / 100.0);
#endif
            S.kkey[slot] = key0; // same value from every lane
            S.koff[slot] = off;
            S.klen[slot] = len;
        }
    }
    __syncthreads();
    STAMP();

    // ---------------- V: verification against the merged labels (all in LDS)
    const int nkept = S.nkept;
    if (!(S.flags & FL_COMPLEX)) {
        // V1: every kept start was acceptable: nearest labelled pixel to its left is negative, or there is none
        for (int e = tid; e < nkept; e += CT_THREADS) {
            const uint32_t key = S.kkey[e];
            const int x0 = (int)(key & 0xFFFFu), y0 = (int)(key >> 16);
            const uint32_t occ = S.rowmask[y0];
            int k = x0 >> 6;
            unsigned long long l = S.lab[LS.slot(y0, k)] & ((1ull << (x0 & 63)) - 1);
            uint32_t left = occ & ((1u << k) - 1u);
            while (!l && left) {
                k = 31 - __clz((int)left);
                left &= ~(1u << k);
                l = S.lab[LS.slot(y0, k)];
            }
            if (l) {
                const int top = 63 - __clzll((long long)l);
                if (!((S.neg[LS.slot(y0, k)] >> top) & 1ull)) atomicOr(&S.flags, FL_COMPLEX);
            }
        }
        // V2: every unlabelled run start would have been rejected
        for (int r = tid; r < nrows; r += CT_THREADS) {
            const int y = S.rows[r];
            const int64_t base = (int64_t)(y + 1) * prow + 1;
            const uint32_t occ = S.rowmask[y];
            int slot = S.rowbase[y];
            uint64_t carry = 0;
            bool last_pos = false;
            for (int k = 0; k < ww; k++) {
                if (!((occ >> k) & 1u)) { carry = 0; continue; }
                const uint64_t fwd = F[base + k];
                const unsigned long long l = S.lab[slot], ng = S.neg[slot];
                slot++;
                uint64_t cand = fwd & ~((fwd << 1) | carry) & ~l;
                while (cand) {
                    const int b = __ffsll((long long)cand) - 1;
                    cand &= cand - 1;
                    const uint64_t below = l & ((1ull << b) - 1);
                    const bool pos = below ? !((ng >> (63 - __clzll((long long)below))) & 1ull) : last_pos;
                    if (!pos) atomicOr(&S.flags, FL_COMPLEX);
                }
                if (l) last_pos = !((ng >> (63 - __clzll((long long)l))) & 1ull);
                carry = fwd >> 63;
            }
        }
    }
    __syncthreads();
    STAMP();
    const bool complex = (S.flags & FL_COMPLEX) != 0;

    if (complex) {
        // ---------------- F: literal scanner (one wavefront), exact for nested components
        if (wave == 0) {
            int nc, np, st;
            literal_frame(lane, F, LAB, NEG, h, ww, prow, pts, cs, cl, max_contours, max_points,
                          summarised ? S.rowmask : nullptr, &nc, &np, &st);
            if (lane == 0) { S.lit[0] = nc; S.lit[1] = np; S.lit[2] = st; }
        }
        __threadfence();
        __syncthreads();
        // the literal scanner labels in the global planes: clear what it set (labels only exist where F is set)
        for (int y = tid; y < h; y += CT_THREADS) {
            const int64_t base = (int64_t)(y + 1) * prow + 1;
            const uint32_t occ = summarised ? S.rowmask[y] : ~0u;
            for (int k = 0; k < ww; k++)
                if ((k >= 32 || ((occ >> k) & 1u)) && F[base + k]) { LAB[base + k] = 0; NEG[base + k] = 0; }
        }
        { // work list of the fit stage: contours with at least 6 points (objdetect.cpp:64); the others are "skipped"
            const int nc = S.lit[0];
            int32_t* el = elig + (int64_t)f * max_contours;
            for (int k = tid; k < nc; k += CT_THREADS) {
                if (cl[k] >= 6) el[atomicAdd(&S.nelig, 1)] = k;
                else slot_kind[(int64_t)f * max_contours + (nc - 1 - k)] = 0;
            }
        }
        __syncthreads();
        if (tid == 0) {
            n_contours[f] = S.lit[0];
            n_points[f] = S.lit[1];
            n_elig[f] = S.nelig;
            status[f] = S.lit[2] | RMCV_FRAME_SLOW_PATH;
        }
    } else {
    STAMP();
#ifdef RMCV_PROFILE
    if (tid == 0 && (f == 0 || f == 100))
        printf("[f%d] rows=%d slots=%d cand=%d kept=%d pts=%d | tables %.1f T %.1f S %.1f V %.1f us\n", f, nrows, S.nslots, ncand,
               nkept, S.cursor, (t_[1] - t_[0]) / 100.0, (t_[2] - t_[1]) / 100.0, (t_[3] - t_[2]) / 100.0, (t_[4] - t_[3]) / 100.0);
#endif
    // discovery order = raster order of the starts: rank the kept entries by key
    for (int e = tid; e < nkept; e += CT_THREADS) {
        const uint32_t key = S.kkey[e];
        int rank = 0;
        for (int j = 0; j < nkept; j++) rank += S.kkey[j] < key;
        cs[rank] = S.koff[e];
        cl[rank] = S.klen[e];
        // work list of the fit stage: contours with at least 6 points (objdetect.cpp:64); the others are "skipped"
        if (S.klen[e] >= 6) elig[(int64_t)f * max_contours + atomicAdd(&S.nelig, 1)] = rank;
        else slot_kind[(int64_t)f * max_contours + (nkept - 1 - rank)] = 0;
    }
    __syncthreads();
    if (tid == 0) {
        n_contours[f] = nkept;
        n_points[f] = S.cursor;
        n_elig[f] = S.nelig;
        status[f] = 0;
    }
    }
    if (!X.fused) return;
    // ---------------- fused tail: rm::filter_lightblobs + rm::filter_armours of THIS frame in the same workgroup, so a batch
    // has one sparse launch and a frame's stages follow each other without waiting for the slowest frame of every stage.
    // The 8 wavefronts take the contours of the work list (one contour each at a time), then wavefront 0 compacts and pairs.
    __syncthreads(); // the lists this workgroup wrote to global memory are visible to all of its wavefronts
    const int nc = complex ? S.lit[0] : S.nkept, ne = S.nelig;
    __syncthreads(); // everybody has read the counters: the contour tables make room for the fit's wave-private rows
    {
        WaveLds& L = reinterpret_cast<WaveLds*>(smem_raw)[wave];
        const int32_t* el = elig + (int64_t)f * max_contours;
        // static split: pulling contours from an LDS counter instead measured 1 % slower (a frame has ~10 eligible contours)
        for (int e = wave; e < ne; e += CT_THREADS / 64)
            fit_contour_slot(f, el[e], nc, pts, cs, cl, max_contours, max_points, X.G, slot_kind, X.slot_ell, L, lane);
    }
    __syncthreads();
    if (wave == 0) {
        const FitTail& T = X.T;
        const int np = blob_compact_frame(f, lane, slot_kind, X.slot_ell, nc, max_contours, T.enemy, T.blobs, T.blob_src, T.ellipses,
                                          T.neg_idx, T.n_blobs, T.n_neg, T.status, T.max_blobs);
        if (T.do_pairs) {
            __threadfence(); // the pair loop re-reads, across lanes, the blobs this wave just wrote
            armours_frame(f, lane, T.blobs, np, T.max_blobs, T.angle_diff_max, T.shear_max, T.length_ratio_max, T.enemy, T.armours,
                          T.n_armours, T.status, T.max_armours);
        }
    }
}

static_assert(sizeof(ContoursLds) >= (CT_THREADS / 64) * sizeof(WaveLds), "the fit rows reuse the contour tables");

static hipError_t launch_contours_x(const Geom& g, const Bufs& b, const Limits& lim, const SparseTail& X, hipStream_t s)
{
    static const int force_literal = getenv("RMCV_CONTOURS_LITERAL") ? atoi(getenv("RMCV_CONTOURS_LITERAL")) : 0; // dev/test knob
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_contours), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)sizeof(ContoursLds));
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_contours, dim3(g.n_frames), dim3(CT_THREADS), sizeof(ContoursLds), s, b.bits, b.rowmask, g.h, b.lab, b.neg,
                       g.w, g.h, g.ww, g.prow, g.plane_pitch, b.points, b.cont_start, b.cont_len, b.n_contours, b.n_points,
                       b.status, lim.max_contours, lim.max_points, force_literal, b.elig, b.n_elig, b.slot_kind, X);
    return hipGetLastError();
}

hipError_t launch_contours(const Geom& g, const Bufs& b, const Limits& lim, hipStream_t s)
{
    SparseTail X;
    memset(&X, 0, sizeof(X));
    return launch_contours_x(g, b, lim, X, s);
}

// findContours + filter_lightblobs (+ filter_armours) of every frame in ONE launch
hipError_t launch_sparse(const Geom& g, const Bufs& b, const Limits& lim, const rmcv_params& p, bool pairs, hipStream_t s)
{
    SparseTail X;
    memset(&X, 0, sizeof(X));
    X.fused = 1;
    X.G = {p.tilt_max, p.ratio_lo, p.ratio_hi, p.area_lo, p.area_hi};
    X.slot_ell = b.slot_ell;
    FitTail& T = X.T;
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
    return launch_contours_x(g, b, lim, X, s);
}

// contours of every frame as CSR in findContours order (reverse discovery), for download
__global__ void k_pack_contours(const rmcv_point* __restrict__ points, const int32_t* __restrict__ cont_start,
                                const int32_t* __restrict__ cont_len, const int32_t* __restrict__ n_contours, int max_contours,
                                int max_points, rmcv_point* __restrict__ pts_out, int32_t* __restrict__ offs_out)
{
    const int f = blockIdx.x;
    const int n = n_contours[f];
    const int32_t* cs = cont_start + (int64_t)f * max_contours;
    const int32_t* cl = cont_len + (int64_t)f * max_contours;
    int32_t* offs = offs_out + (int64_t)f * (max_contours + 1);
    __shared__ int s_off;
    if (threadIdx.x == 0) {
        int o = 0;
        for (int i = 0; i < n; i++) {
            offs[i] = o;
            int len = cl[n - 1 - i];
            int room = max_points - cs[n - 1 - i];
            o += len < room ? len : (room > 0 ? room : 0);
        }
        offs[n] = o;
        s_off = o;
    }
    __syncthreads();
    for (int i = 0; i < n; i++) {
        const int k = n - 1 - i;
        const int o = offs[i], len = offs[i + 1] - offs[i];
        const rmcv_point* src = points + (int64_t)f * max_points + cs[k];
        rmcv_point* dst = pts_out + (int64_t)f * max_points + o;
        for (int j = threadIdx.x; j < len; j += blockDim.x) dst[j] = src[j];
    }
}

hipError_t launch_pack_contours(const Geom& g, const Bufs& b, const Limits& lim, rmcv_point* d_pts_out, int32_t* d_offs_out,
                                hipStream_t s)
{
    hipLaunchKernelGGL(k_pack_contours, dim3(g.n_frames), dim3(256), 0, s, b.points, b.cont_start, b.cont_len, b.n_contours,
                       lim.max_contours, lim.max_points, d_pts_out, d_offs_out);
    return hipGetLastError();
}

} // namespace rmcv
