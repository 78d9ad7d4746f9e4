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
__device__ __forceinline__ unsigned long long ld_l2(const unsigned long long* p)
{
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
__device__ __forceinline__ int trace_border(Tracer& t, int x0, int y0, rmcv_point* out, int room, int* ymax)
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
__device__ __forceinline__ int scan_row(const uint64_t* F, const uint64_t* LAB, const uint64_t* NEG, int prow, int ww, int y, int xmin,
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
__device__ __forceinline__ void literal_frame(int lane, const uint64_t* F, uint64_t* LAB, uint64_t* NEG, int h, int ww, int prow, rmcv_point* pts,
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

__device__ __forceinline__ int dir_dx(int s) { return (s == 0 || s == 1 || s == 7) ? 1 : ((s >= 3 && s <= 5) ? -1 : 0); }
__device__ __forceinline__ int dir_dy(int s) { return (s >= 1 && s <= 3) ? -1 : ((s >= 5) ? 1 : 0); }

// Sparse label store of the fast path: one LDS slot per NON-EMPTY word of F (labels only exist where F is set).
// slot(y, k) = rowbase[y] + popcount(rowmask[y] & ((1 << k) - 1)).
struct LabelStore {
    const uint32_t* rowmask;
    const uint16_t* rowbase;
    unsigned long long* lab;
    unsigned long long* neg;
    __device__ __forceinline__ int slot(int y, int k) const { return rowbase[y] + __popc(rowmask[y] & ((1u << k) - 1u)); }
};

// ---- k_contours: one workgroup (8 wavefronts) per frame ---------------------------------------------------
//  C  all borders of the frame at once as cycles of visits (cycles_frame below): nodes from the bit plane, linked, ranked by
//     pointer doubling -> points + labels of every outer border.
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
// (The capacities are per translation unit: k_contours_lean.hip compiles the same kernel body with tiny LDS-tier tables for frames that
// take the mid tier anyway -- everything that depends on them has internal linkage or is device code.)
#ifndef RMCV_KEPT_CAP
#define RMCV_KEPT_CAP 512
#endif
static constexpr int KEPT_CAP = RMCV_KEPT_CAP;
// Non-empty words of a frame the LDS label store can hold.  Round 5: 1664 instead of 1024 at the SAME struct size -- the per-word border
// masks (bmask / e2mask: written in N1, last read in N3) and the label planes (lab / neg: first written in N7) never live at the same
// time and share their storage, 20 bytes per word instead of 36.  A scene of 400 specks and a few lit windows (the synthetic stream's
// dense2: 950-1560 non-empty words, 2 300 border points per frame) now stays in the LDS tier instead of taking the mid tier at four times
// the cost per frame -- and its batches stay "calm" for the pipeline's hot contexts.
#ifndef RMCV_SLOT_CAP
#define RMCV_SLOT_CAP 1664
#endif
static constexpr int SLOT_CAP = RMCV_SLOT_CAP;
// Workgroup size is a build parameter (k_contours_kernel.inc): 8 wavefronts build the borders and fit the bars of a frame (lowest
// latency for one batch), 4 wavefronts leave room on every CU for the pixel kernels of the next batches AND the sparse kernel of the previous
// one (VGPR budget per SIMD: 2 x 160 for this kernel at 8 wavefronts, 2 x 80 per pixel kernel, 512 in all) -- the better choice
// when several batches are in flight (886 k against 800 k frames/s with three batches; 0.57 against 0.475 ms for a lone batch).
#ifndef RMCV_CT_THREADS_MAX
#define RMCV_CT_THREADS_MAX 512
#endif
static constexpr int CT_THREADS_MAX = RMCV_CT_THREADS_MAX; // the largest workgroup a translation unit instantiates the kernel body with
static constexpr int MULTI_CAP = 32;    // pixels of a frame visited 3 or 4 times (junctions of 1-pixel lines) the cycle formulation lists
#ifndef RMCV_NN_CAP
#define RMCV_NN_CAP VISIT_CAP
#endif
static constexpr int NN_CAP = RMCV_NN_CAP;  // border visits (nodes) of a frame the cycle formulation holds in LDS
static_assert(NN_CAP <= VISIT_CAP, "the visits' global scratch (Bufs::visit_xy) is VISIT_CAP entries per frame");
#ifndef RMCV_MID_PAD
#define RMCV_MID_PAD 0
#endif
#ifndef RMCV_CT_MAXH
#define RMCV_CT_MAXH 2048
#endif
static constexpr int CT_MAXH = RMCV_CT_MAXH;   // rows covered by the LDS row tables (taller/wider frames take the literal path)

// The workgroup's LDS: this struct, followed by the ROW TABLES (RowTabs below), which are sized by the frame's height at launch:
// a row costs 8 bytes (mask of its non-empty words, its place in the list of non-empty rows, its slot base), 2048 rows 16 KB but the
// 1200 rows of a 1920x1200 frame 9.6 KB -- and with 16 KB the workgroup did not fit a CU's 160 KB beside the four pixel-kernel
// workgroups (4 x 19.8 KB) of two batches at that width (round 3: C5 +4-6 % once it does, tools/ab_process_r3.sh c5_lds).
// Layout: first the tables of the LDS tier that are dead once the contours are out -- the fused tail overlays its wave-private rows on
// the first 8 x sizeof(WaveLds) = 30 720 bytes of them, and the mid tier runs its pointer doubling on the whole of them (mid_lds_words)
// --, then what every tier shares (scan, ringtab), then what the fused tail needs (the work list's copies, the node tables), then the scalars.
struct ContoursLds {
    unsigned long long lab[SLOT_CAP], neg[SLOT_CAP]; // N1-N3: the words' border masks (bmask / e2mask below); from N7 on: the label planes
    uint16_t nbase[SLOT_CAP], spos[SLOT_CAP]; // cycle formulation: node-id base and (row, word) per slot
    uint32_t kkey[KEPT_CAP];
    int32_t koff[KEPT_CAP], klen[KEPT_CAP];
    uint8_t kacc[KEPT_CAP];    // cycles_frame: outer border e is (still) accepted by the RETR_EXTERNAL rule
    uint32_t multi[MULTI_CAP]; // cycles_frame: pixels the border visits 3 or 4 times: slot:16 | bit:6 << 16 | count << 24
#if RMCV_MID_PAD > 0
    unsigned long long mid_pad[RMCV_MID_PAD / 8]; // (lean build: the mid tier's pointer doubling and staging want this much in front of `scan`)
#endif
    // ---- shared by the tiers
    int scan[CT_THREADS_MAX];
    uint32_t ringtab[256];
    // ---- the fused tail's: starts and lengths of the work list's contours (cycle path), the node tables (work list, sort keys)
    uint32_t w_off[KEPT_CAP], w_cnt[KEPT_CAP];
    uint16_t n_a[NN_CAP], n_b[NN_CAP], n_d[NN_CAP];
    int nnodes, nkept, cursor, flags, nrows, nslots, nelig, lit[3];
    int nmulti;
    int revoked;  // cycles_frame: some acceptance was revoked in this round
    int wnext;    // fused tail: next entry of the sorted work list
    int sink[64]; // per-lane sinks: lanes != 0 add 0 here so that a wave-wide atomic stays convergent and does not serialise on one word
};
static_assert(offsetof(ContoursLds, scan) >= (CT_THREADS_MAX / 64) * 3840, "the fused tail's wave-private rows (sizeof(WaveLds) = 3840 each) overlay dead tables only");
static_assert(offsetof(ContoursLds, scan) % 8 == 0 && offsetof(ContoursLds, w_off) % 8 == 0, "alignment");
// words of LDS the mid tier may use for its pointer doubling: the LDS tier's own tables in front of `scan`
static constexpr int MID_LDS_WORDS = (int)(offsetof(ContoursLds, scan) / 4);

// the row tables behind the struct: `rows_cap` rows (the frame's height rounded up to 64, at most CT_MAXH)
struct RowTabs {
    uint32_t* rowmask; // [row] bit k: word k of the row is non-empty
    uint16_t* rows;    // the non-empty rows
    uint16_t* rowbase; // [row] slot of the row's first non-empty word
};
__host__ __device__ static inline int lds_rows_cap(int h) { const int r = (h + 63) & ~63; return r < CT_MAXH ? r : CT_MAXH; }
__host__ __device__ static inline size_t lds_bytes(int h) { return sizeof(ContoursLds) + (size_t)lds_rows_cap(h) * 8; }
__device__ __forceinline__ RowTabs row_tabs(void* smem, int rows_cap)
{
    RowTabs R;
    R.rowmask = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(smem) + sizeof(ContoursLds));
    R.rows = reinterpret_cast<uint16_t*>(R.rowmask + rows_cap);
    R.rowbase = R.rows + rows_cap;
    return R;
}

// ---- cycle formulation of the border following --------------------------------------------------------------------------
// Every VISIT of Suzuki's follower to a pixel is a node: (pixel, maximal arc of consecutive background neighbours that contains
// a 4-neighbour; the follower arrives with its back to the foreground neighbour at the arc's clockwise end and leaves towards
// the one at its counter-clockwise end).  The follower's step is a bijection on nodes, so a border is a CYCLE of nodes; its
// points in OpenCV's order are the cycle read from its raster-first node, and it is an outer border iff that node's arc
// contains the west neighbour (tools/proto/cycle_contours.py checks this restatement against the oracle).  Instead of walking
// a border step by step (one wavefront, ~0.3 us per step, the longest contour of a frame bounding the stage) all nodes of the
// frame are created at once from the bit plane, linked, and ranked by pointer doubling: log2(nodes) rounds of the whole
// workgroup, independent of how long any single contour is.
//   ringtab[ring]: cnt:3 | back0..3 (3 bits each) | next0..3 (3 bits each) | west(arc 0):1 | neg0..3:4   (arc 0 = the arc with W)
//   pxy[node]    : x:12 | y:12 | back:3 | neg:1 | west:1 | next:3   (global scratch: written once, read twice)
constexpr uint32_t ring_entry(int ring)
{
    uint32_t backs[4] = {0, 0, 0, 0}, nexts[4] = {0, 0, 0, 0}, negs[4] = {0, 0, 0, 0};
    int cnt = 0, west0 = 0;
    if (ring == 0) { // isolated pixel: one visit that stays; OpenCV gives it the negative label
        cnt = 1;
        west0 = 1;
        negs[0] = 1;
    } else {
        int wa = -1;
        for (int b = 0; b < 8; b++) {
            if (!((ring >> b) & 1)) continue;
            int d = (b + 1) & 7, axis = 0, west = 0, len = 0;
            while (!((ring >> d) & 1)) {
                axis |= !(d & 1);
                west |= d == 4;
                d = (d + 1) & 7;
                len++;
            }
            if (!len || !axis) continue; // no background 4-neighbour in the arc: the follower never comes this way
            if (cnt < 4) {
                backs[cnt] = (uint32_t)b;
                nexts[cnt] = (uint32_t)d;
                negs[cnt] = ((unsigned)(d - 1) < (unsigned)b) ? 1u : 0u; // the sweep passed the east neighbour
                if (west) wa = cnt;
            }
            cnt++;
        }
        if (wa > 0 && wa < 4) { // the west arc first
            uint32_t t = backs[0]; backs[0] = backs[wa]; backs[wa] = t;
            t = nexts[0]; nexts[0] = nexts[wa]; nexts[wa] = t;
            t = negs[0]; negs[0] = negs[wa]; negs[wa] = t;
        }
        west0 = wa >= 0;
    }
    uint32_t e = (uint32_t)(cnt > 7 ? 7 : cnt);
    for (int a = 0; a < 4; a++) e |= (backs[a] << (3 + 3 * a)) | (nexts[a] << (15 + 3 * a)) | (negs[a] << (28 + a));
    e |= (uint32_t)west0 << 27;
    return e;
}
struct RingTab {
    uint32_t v[256];
    constexpr RingTab() : v()
    {
        for (int r = 0; r < 256; r++) v[r] = ring_entry(r);
    }
};
static __device__ const RingTab RINGTAB = RingTab();

// 8-bit ring of pixel bit b of word c from the three rows' words (l = word k-1, c = word k, r = word k+1 of each row)
__device__ __forceinline__ uint32_t ring_of(int b, uint64_t ul, uint64_t uc, uint64_t ur, uint64_t ml, uint64_t mc, uint64_t mr,
                                            uint64_t dl, uint64_t dc, uint64_t dr)
{
    // 3 bits (x-1, x, x+1) of a row
    auto three = [&](uint64_t l, uint64_t c, uint64_t r) -> uint32_t {
        if (b == 0) return (uint32_t)(l >> 63) | ((uint32_t)(c & 3ull) << 1);
        if (b == 63) return (uint32_t)(c >> 62) | ((uint32_t)(r & 1ull) << 2);
        return (uint32_t)(c >> (b - 1)) & 7u;
    };
    const uint32_t u = three(ul, uc, ur), m = three(ml, mc, mr), d = three(dl, dc, dr);
    return ((m >> 2) & 1u) | (((u >> 2) & 1u) << 1) | (((u >> 1) & 1u) << 2) | ((u & 1u) << 3) | ((m & 1u) << 4) | ((d & 1u) << 5) |
           (((d >> 1) & 1u) << 6) | (((d >> 2) & 1u) << 7);
}

// The border pixels of a word and how often the follower visits each, 64 pixels per operation.  A pixel is visited once per
// maximal run of background neighbours (counter-clockwise round its 8-ring) that contains a 4-neighbour; such a run is counted at
// its FIRST background 4-neighbour a, i.e. where not both a - 1 and a - 2 are background too:
//   visits = [~E & (SE | S)] + [~N & (NE | E)] + [~W & (NW | N)] + [~S & (SW | W)]        (an isolated pixel: 1)
// -- equal to ringtab's count for every ring of a border pixel (checked exhaustively).  Replaces a loop over
// the word's border pixels with a table lookup each: a wavefront ran as long as the fullest of its 64 words needed.
__device__ __forceinline__ void visit_masks(uint64_t ul, uint64_t uc, uint64_t ur, uint64_t ml, uint64_t mc, uint64_t mr, uint64_t dl,
                                            uint64_t dc, uint64_t dr, uint64_t* B, uint64_t* E2, uint64_t* E3, uint64_t* E4)
{
    const uint64_t W = (mc << 1) | (ml >> 63), E = (mc >> 1) | (mr << 63), N = uc, S = dc;
    const uint64_t NW = (uc << 1) | (ul >> 63), NE = (uc >> 1) | (ur << 63), SW = (dc << 1) | (dl >> 63), SE = (dc >> 1) | (dr << 63);
    const uint64_t b = mc & ~(N & S & W & E);
    const uint64_t cE = ~E & (SE | S), cN = ~N & (NE | E), cW = ~W & (NW | N), cS = ~S & (SW | W);
    const uint64_t t1 = cE ^ cN, c1 = cE & cN, t2 = cW ^ cS, c2 = cW & cS;
    *B = b;
    *E2 = b & (c1 | c2 | (t1 & t2));
    *E3 = b & ((c1 & (cW | cS)) | (c2 & (cE | cN)));
    *E4 = b & c1 & c2;
}

// nodes beyond two of the listed 3-/4-visit pixels of `slot` below bit position `bit` (64: the whole word), and the count of the
// pixel at `bit` itself (0 if it is not listed)
__device__ __forceinline__ int multi_extra(const ContoursLds& S, int slot, int bit, int* own)
{
    int extra = 0;
    *own = 0;
    const int nm = S.nmulti < MULTI_CAP ? S.nmulti : MULTI_CAP;
    for (int m = 0; m < nm; m++) {
        const uint32_t e = S.multi[m];
        if ((int)(e & 0xFFFFu) != slot) continue;
        const int b = (int)((e >> 16) & 63u), c = (int)(e >> 24);
        if (b < bit) extra += c - 2;
        else if (b == bit) *own = c;
    }
    return extra;
}

// findContours of one frame by the whole workgroup (T threads); on return S.kkey/koff/klen/nkept/cursor describe the kept contours
// (in any order; the caller ranks them by key), the points are written and the labels of the accepted borders are in the LDS
// label store.
// FLCAP is OR-ed into S.flags when a capacity of the LDS tables is exceeded (the frame then takes the mid tier), FL when the
// formulation met something it cannot express (the literal scanner settles it).
template <int T>
__device__ __forceinline__ void cycles_frame(ContoursLds& S, const RowTabs& RT, const uint64_t* __restrict__ F, int prow, int h, int ww, int nrows, int tid,
                             rmcv_point* __restrict__ pts, int max_points, int max_contours, int FL, int FLCAP, uint32_t* __restrict__ pxy)
{
    constexpr int NPT = NN_CAP / T; // nodes per thread in the doubling rounds
    const LabelStore LS = {RT.rowmask, RT.rowbase, S.lab, S.neg};
    unsigned long long* const bmask = S.lab;  // border pixels / two-visit pixels per slot: N1-N3 only -- the label planes take their place in N7
    unsigned long long* const e2mask = S.neg;
    uint16_t* const nxt = S.n_a; // successor of every node (N3 .. N5)
#ifdef RMCV_PROFILE
    long long tc_[12]; int tci_ = 0;
#define CSTAMP() do { __syncthreads(); tc_[tci_++] = wall_clock64(); } while (0)
#else
#define CSTAMP() do {} while (0)
#endif
    CSTAMP();
    for (int i = tid; i < 256; i += T) S.ringtab[i] = RINGTAB.v[i];
    if (tid == 0) { S.nnodes = 0; S.nmulti = 0; }
    __syncthreads();
    CSTAMP();
    // slot -> (row, word): y | k << 11
    for (int r = tid; r < nrows; r += T) {
        const int y = RT.rows[r];
        uint32_t occ = RT.rowmask[y];
        int slot = RT.rowbase[y];
        while (occ) {
            const int k = __ffs((int)occ) - 1;
            occ &= occ - 1;
            S.spos[slot++] = (uint16_t)(y | (k << 11));
        }
    }
    __syncthreads();
    CSTAMP();
    // ---- N1: per non-empty word: border pixels, pixels visited twice, node count
    const int nslots = S.nslots;
    for (int slot = tid; slot < nslots; slot += T) {
        const int y = S.spos[slot] & 2047, k = S.spos[slot] >> 11;
        const int64_t base = (int64_t)(y + 1) * prow + 1;
        const uint64_t mc = F[base + k];
        const uint64_t ul = F[base - prow + k - 1], uc = F[base - prow + k], ur = F[base - prow + k + 1];
        const uint64_t ml = F[base + k - 1], mr = F[base + k + 1];
        const uint64_t dl = F[base + prow + k - 1], dc = F[base + prow + k], dr = F[base + prow + k + 1];
        uint64_t B, E2, E3, E4;
        visit_masks(ul, uc, ur, ml, mc, mr, dl, dc, dr, &B, &E2, &E3, &E4);
        int extra = 0;
        uint64_t rem = E3; // junctions of 1-pixel lines (visited 3 or 4 times): listed on the side
        while (rem) {
            const int b = __ffsll((long long)rem) - 1;
            rem &= rem - 1;
            const uint32_t cnt = 3u + (uint32_t)((E4 >> b) & 1ull);
            const int m = atomicAdd(&S.nmulti, 1);
            if (m < MULTI_CAP) S.multi[m] = (uint32_t)slot | ((uint32_t)b << 16) | (cnt << 24);
            else atomicOr(&S.flags, FLCAP);
            extra += (int)cnt - 2;
        }
        bmask[slot] = B;
        e2mask[slot] = E2;
        S.nbase[slot] = (uint16_t)(__popcll(B) + __popcll(E2) + extra);
    }
    __syncthreads();
    { // exclusive prefix of the node counts over the slots (raster order): node ids ascend in raster order
        const int ns = nslots, per = (ns + T - 1) / T;
        int sum = 0;
        for (int u = 0; u < per; u++) {
            const int i = tid * per + u;
            if (i < ns) sum += S.nbase[i];
        }
        S.scan[tid] = sum;
        __syncthreads();
        for (int d = 1; d < T; d <<= 1) {
            const int v = tid >= d ? S.scan[tid - d] : 0;
            __syncthreads();
            S.scan[tid] += v;
            __syncthreads();
        }
        int run = S.scan[tid] - sum;
        for (int u = 0; u < per; u++) {
            const int i = tid * per + u;
            if (i < ns) {
                const int c = S.nbase[i];
                S.nbase[i] = (uint16_t)(run < 65535 ? run : 65535);
                run += c;
            }
        }
        if (tid == T - 1) {
            S.nnodes = S.scan[tid];
            if (S.scan[tid] > NN_CAP) S.flags |= FLCAP;
        }
    }
    __syncthreads();
    if (S.flags) return;
    const int nn = S.nnodes;
    CSTAMP();
    // ---- N2: the nodes.  Round 5: one border PIXEL per lane.  (One non-empty WORD per lane, walking its border pixels one after the
    // other, ran as long as the fullest of a wavefront's 64 words -- the edge of a lit window is 64 border pixels in one word, a speck
    // two: 35 of the 95 us a frame of 400 specks spent on its borders.)  A wavefront takes 64 words at a time: every lane stages its
    // word's three centre words and six neighbouring edge bits in wave-private LDS (the node tables n_a .. n_d are still free), the lanes
    // scatter (word, bit) of their border pixels into a list at the word's exclusive prefix, and then every lane takes list entries:
    // ring -> visits -> node ids exactly as the word-wise walk numbered them (the id of a pixel's first visit = the word's base + the
    // visits of the word's earlier pixels: the formula N3 uses to find a successor).
    {
        constexpr int NWV = T / 64;
        constexpr int STG = (int)(3 * NN_CAP * sizeof(uint16_t)) / NWV; // bytes of staging per wavefront
        constexpr int LCAP = (STG - 2048) / 2;                           // list entries per pass
        static_assert(LCAP >= 256 && offsetof(ContoursLds, n_b) == offsetof(ContoursLds, n_a) + NN_CAP * sizeof(uint16_t) &&
                      offsetof(ContoursLds, n_d) == offsetof(ContoursLds, n_b) + NN_CAP * sizeof(uint16_t), "n_a .. n_d are one block");
        const int lane = tid & 63, wave = tid >> 6;
        unsigned char* const stg = reinterpret_cast<unsigned char*>(S.n_a) + wave * STG;
        unsigned long long* const sw = reinterpret_cast<unsigned long long*>(stg); // [4][64]: uc, mc, dc, edge bits
        uint16_t* const list = reinterpret_cast<uint16_t*>(stg + 2048);
        for (int c0 = wave * 64; c0 < nslots; c0 += T) { // (wave-uniform)
            const int slot = c0 + lane;
            uint64_t B = 0;
            if (slot < nslots) {
                B = bmask[slot];
                const int y = S.spos[slot] & 2047, k = S.spos[slot] >> 11;
                const int64_t base = (int64_t)(y + 1) * prow + 1 + k;
                const uint64_t ul = F[base - prow - 1], uc = F[base - prow], ur = F[base - prow + 1];
                const uint64_t ml = F[base - 1], mc = F[base], mr = F[base + 1];
                const uint64_t dl = F[base + prow - 1], dc = F[base + prow], dr = F[base + prow + 1];
                sw[lane] = uc;
                sw[64 + lane] = mc;
                sw[128 + lane] = dc;
                sw[192 + lane] = (ul >> 63) | ((ur & 1ull) << 1) | ((ml >> 63) << 2) | ((mr & 1ull) << 3) | ((dl >> 63) << 4) | ((dr & 1ull) << 5);
            }
            const int pc = __popcll(B);
            int incl = pc;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int v = __shfl_up(incl, d);
                if (lane >= d) incl += v;
            }
            const int excl = incl - pc, P = __shfl(incl, 63);
            for (int p0 = 0; p0 < P; p0 += LCAP) { // (wave-uniform)
                uint64_t rem = B;
                int r = excl - p0;
                while (rem) {
                    const int b = __ffsll((long long)rem) - 1;
                    rem &= rem - 1;
                    if (r >= 0 && r < LCAP) list[r] = (uint16_t)((lane << 6) | b);
                    r++;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int m = P - p0 < LCAP ? P - p0 : LCAP;
                for (int j = lane; j < m; j += 64) {
                    const int en = list[j], sl = en >> 6, bb = en & 63;
                    const uint64_t uc = sw[sl], mc = sw[64 + sl], dc = sw[128 + sl], eb = sw[192 + sl];
                    const int slot2 = c0 + sl;
                    const uint32_t e = S.ringtab[ring_of(bb, (eb & 1ull) << 63, uc, (eb >> 1) & 1ull, ((eb >> 2) & 1ull) << 63, mc, (eb >> 3) & 1ull,
                                                         ((eb >> 4) & 1ull) << 63, dc, (eb >> 5) & 1ull)];
                    const int cnt = (int)(e & 7u);
                    const int y = S.spos[slot2] & 2047, k = S.spos[slot2] >> 11;
                    const uint32_t xy = (uint32_t)(k * 64 + bb) | ((uint32_t)y << 12);
                    const uint64_t below = (1ull << bb) - 1;
                    int id = S.nbase[slot2] + __popcll(bmask[slot2] & below) + __popcll(e2mask[slot2] & below);
                    if (S.nmulti) {
                        int own;
                        id += multi_extra(S, slot2, bb, &own);
                    }
                    for (int a = 0; a < cnt && a < 4; a++) {
                        const uint32_t back = (e >> (3 + 3 * a)) & 7u, nextd = (e >> (15 + 3 * a)) & 7u, ng = (e >> (28 + a)) & 1u;
                        const uint32_t west = a == 0 ? (e >> 27) & 1u : 0u;
                        if (id < NN_CAP) pxy[id] = xy | (back << 24) | (ng << 27) | (west << 28) | (nextd << 29);
                        id++;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier(); // the list (and, behind the last pass, the staged words) are rewritten
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    }
    __syncthreads();
    CSTAMP();
    // ---- N3: successor and predecessor of every node
    for (int i = tid; i < nn; i += T) {
        const uint32_t p = pxy[i];
        const int x = (int)(p & 0xFFFu), y = (int)((p >> 12) & 0xFFFu), nd = (int)(p >> 29);
        int succ = i;
        const int xs = x + dir_dx(nd), ys = y + dir_dy(nd);
        const int ks = xs >> 6, bs = xs & 63;
        bool ok = xs >= 0 && ys >= 0 && ys < h && ks < ww && ((RT.rowmask[ys] >> ks) & 1u);
        int slot2 = 0;
        if (ok) {
            slot2 = LS.slot(ys, ks);
            ok = (bmask[slot2] >> bs) & 1ull;
        }
        if (ok) {
            const uint64_t below = (1ull << bs) - 1;
            int id0 = S.nbase[slot2] + __popcll(bmask[slot2] & below) + __popcll(e2mask[slot2] & below);
            int cnt2 = 1 + (int)((e2mask[slot2] >> bs) & 1ull);
            if (S.nmulti) {
                int own;
                id0 += multi_extra(S, slot2, bs, &own);
                if (own) cnt2 = own;
            }
            const uint32_t back2 = (uint32_t)((nd + 4) & 7);
            succ = id0; // the visit of the successor pixel whose back direction points here
            for (int a = 1; a < cnt2; a++)
                if (((pxy[id0 + a] >> 24) & 7u) == back2) succ = id0 + a;
        }
        // an isolated pixel has no foreground neighbour at all: its "next" leads to background -> it stays (ok == false, succ == i);
        // for any other node a missing successor would contradict the bijection: flagged
        if (!ok) {
            const int k0 = x >> 6;
            const int64_t base = (int64_t)(y + 1) * prow + 1;
            const uint32_t ring = ring_of(x & 63, F[base - prow + k0 - 1], F[base - prow + k0], F[base - prow + k0 + 1], F[base + k0 - 1],
                                          F[base + k0], F[base + k0 + 1], F[base + prow + k0 - 1], F[base + prow + k0], F[base + prow + k0 + 1]);
            if (ring != 0) atomicOr(&S.flags, FL);
        }
        nxt[i] = (uint16_t)succ;
    }
    __syncthreads();
    if (S.flags) return;
    CSTAMP();
    // ---- N4: smallest node id of every cycle, by pointer doubling.  Round 5: ONE packed word per node (mn : 16 | jp : 16), swept in
    // place -- a word always describes a true segment [i, jp) of the cycle with its minimum, so a sweep may read words other threads
    // have already advanced (after `rounds` sweeps every segment is at least nn long, i.e. covers its whole cycle): one gather and one
    // barrier per node and round instead of two and two (the mid tier's form; 14.4 -> 7 us per phase on a frame of 2 000 visits).
    static_assert(offsetof(ContoursLds, n_d) == offsetof(ContoursLds, n_b) + NN_CAP * sizeof(uint16_t), "n_b + n_d hold one 32-bit word per node");
    uint32_t* const Rw = reinterpret_cast<uint32_t*>(S.n_b);
    int rounds = 0;
    while ((1 << rounds) < nn) rounds++;
    for (int i = tid; i < nn; i += T) Rw[i] = ((uint32_t)i << 16) | nxt[i];
    __syncthreads();
    for (int rd = 0; rd < rounds; rd++) { // (u * T < nn: wave-uniform -- a frame of 400 visits touches 2 of the 16 words a thread could own)
        uint32_t w[NPT], wt[NPT];
#pragma unroll
        for (int u = 0; u < NPT; u++)
            if (u * T < nn) w[u] = tid + u * T < nn ? Rw[tid + u * T] : 0u;
#pragma unroll
        for (int u = 0; u < NPT; u++)
            if (u * T < nn) wt[u] = Rw[w[u] & 0xFFFFu];
#pragma unroll
        for (int u = 0; u < NPT; u++)
            if (tid + u * T < nn) Rw[tid + u * T] = (((w[u] >> 16) < (wt[u] >> 16) ? (w[u] >> 16) : (wt[u] >> 16)) << 16) | (wt[u] & 0xFFFFu);
        __syncthreads();
    }
    CSTAMP();
    // ---- N5: number of steps from every node FORWARD to its cycle's start (the smallest id); a node's position in the contour is
    // the cycle length minus that.  The doubling pointers are the successors with the start made absorbing; the word is (dist : 16 | jp : 16).
    // From here on: mn lives where the successors were (n_a), the distances in n_b, the starts' kept-contour slots in n_d.
    uint16_t* const mn = S.n_a;
    uint16_t* const dist = S.n_b;
    uint16_t* const jp = S.n_d;
    uint16_t succ0[NPT]; // the successor of a start node: the last node of its contour
#pragma unroll
    for (int u = 0; u < NPT; u++) { // (a thread touches only its own nodes' words: no barrier inside)
        const int i = tid + u * T;
        succ0[u] = 0;
        if (i < nn) {
            const uint32_t mnv = Rw[i] >> 16, sc = nxt[i];
            succ0[u] = (uint16_t)sc;
            mn[i] = (uint16_t)mnv; // (overwrites nxt[i])
            Rw[i] = mnv == (uint32_t)i ? (uint32_t)i : ((1u << 16) | sc);
        }
    }
    __syncthreads();
    for (int rd = 0; rd < rounds; rd++) {
        uint32_t w[NPT], wt[NPT];
#pragma unroll
        for (int u = 0; u < NPT; u++)
            if (u * T < nn) w[u] = tid + u * T < nn ? Rw[tid + u * T] : 0u;
#pragma unroll
        for (int u = 0; u < NPT; u++)
            if (u * T < nn) wt[u] = Rw[w[u] & 0xFFFFu];
#pragma unroll
        for (int u = 0; u < NPT; u++)
            if (tid + u * T < nn) Rw[tid + u * T] = (((w[u] >> 16) + (wt[u] >> 16)) << 16) | (wt[u] & 0xFFFFu);
        __syncthreads();
    }
    {   // the distances move into their 16-bit table (half of the words' own storage: everybody reads before anybody writes)
        uint16_t dv[NPT];
#pragma unroll
        for (int u = 0; u < NPT; u++) dv[u] = tid + u * T < nn ? (uint16_t)(Rw[tid + u * T] >> 16) : (uint16_t)0;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NPT; u++)
            if (tid + u * T < nn) dist[tid + u * T] = dv[u];
        __syncthreads();
    }
    CSTAMP();
    // ---- N6: every cycle whose start visit contains the west neighbour is an outer border: a candidate contour
#pragma unroll
    for (int u = 0; u < NPT; u++) {
        const int i = tid + u * T;
        if (i >= nn) continue;
        uint16_t ks = 0xFFFF;
        if (mn[i] == i && ((pxy[i] >> 28) & 1u)) {
            const int len = dist[succ0[u]] + 1;
            const int slot = atomicAdd(&S.nkept, 1);
            if (slot >= KEPT_CAP || slot >= max_contours) {
                atomicOr(&S.flags, FLCAP);
            } else {
                const uint32_t p = pxy[i];
                S.kkey[slot] = ((p >> 12) & 0xFFFu) << 16 | (p & 0xFFFu);
                S.klen[slot] = len;
                S.kacc[slot] = 1;
                ks = (uint16_t)slot;
            }
        }
        if (mn[i] == i) jp[i] = ks;
    }
    __syncthreads();
    if (S.flags) return;
    CSTAMP();
    // ---- N7: RETR_EXTERNAL.  OpenCV's scanner skips an outer-border start when the last labelled pixel it met on the row is
    // positive, and only the borders it traced carry labels: a component inside a hole of a traced one is skipped.  A start's
    // left context consists of raster-earlier borders only, so the scanner's decisions are the fixed point of "label the accepted
    // borders, revoke every accepted start whose nearest labelled pixel to the left is positive": a revocation is always final
    // (such a start lies inside a hole of the labelled border), an acceptance may be revoked once a wrongly accepted neighbour
    // lost its labels.  Frames without nested components -- the usual case -- take one round.
    const int ncand = S.nkept;
    for (int round = 0;; round++) {
        if (tid == 0) S.revoked = 0;
        for (int i = tid; i < S.nslots; i += T) { S.lab[i] = 0; S.neg[i] = 0; }
        __syncthreads();
        for (int i = tid; i < nn; i += T) {
            const int ks = jp[mn[i]];
            if (ks == 0xFFFF || !S.kacc[ks]) continue; // a hole border, or a border that is not (or no longer) accepted
            const uint32_t p = pxy[i];
            const int x = (int)(p & 0xFFFu), y = (int)((p >> 12) & 0xFFFu);
            const int slot = LS.slot(y, x >> 6);
            atomicOr(&S.lab[slot], 1ull << (x & 63));
            if ((p >> 27) & 1u) atomicOr(&S.neg[slot], 1ull << (x & 63));
        }
        __syncthreads();
        for (int e = tid; e < ncand; e += T) {
            if (!S.kacc[e]) continue;
            const uint32_t key = S.kkey[e];
            const int x0 = (int)(key & 0xFFFFu), y0 = (int)(key >> 16);
            const uint32_t occ = RT.rowmask[y0];
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
                if (!((S.neg[LS.slot(y0, k)] >> top) & 1ull)) { // positive: inside a hole of that border
                    S.kacc[e] = 0;
                    S.revoked = 1;
                }
            }
        }
        __syncthreads();
        const int again = S.revoked;
        __syncthreads();
        if (!again) break;
        if (round >= 32) { // a long chain of nested siblings: the literal scanner settles it
            if (tid == 0) S.flags |= FL;
            __syncthreads();
            return;
        }
    }
    CSTAMP();
    // ---- N8: output of the accepted borders
    for (int e = tid; e < ncand; e += T)
        if (S.kacc[e]) {
            const int off = atomicAdd(&S.cursor, S.klen[e]);
            if (off + S.klen[e] > max_points) atomicOr(&S.flags, FLCAP);
            S.koff[e] = off;
        }
    __syncthreads();
    if (S.flags) return;
    for (int i = tid; i < nn; i += T) {
        const int ks = jp[mn[i]];
        if (ks == 0xFFFF || !S.kacc[ks]) continue;
        const uint32_t p = pxy[i];
        rmcv_point q;
        q.x = (int)(p & 0xFFFu);
        q.y = (int)((p >> 12) & 0xFFFu);
        const int len = S.klen[ks], pos = dist[i] ? len - dist[i] : 0;
        if (pos >= 0 && pos < len) pts[S.koff[ks] + pos] = q; // (always, for a consistent plane)
    }
    __syncthreads();
    { // the kept list = the accepted candidates (order is irrelevant, the caller ranks them by key); the node tables are free
        uint32_t* const t_key = reinterpret_cast<uint32_t*>(S.n_a);
        int32_t* const t_off = reinterpret_cast<int32_t*>(S.n_b);
        int32_t* const t_len = reinterpret_cast<int32_t*>(S.n_d);
        if (tid == 0) S.nkept = 0;
        __syncthreads();
        for (int e = tid; e < ncand; e += T)
            if (S.kacc[e]) {
                const int o = atomicAdd(&S.nkept, 1);
                t_key[o] = S.kkey[e];
                t_off[o] = S.koff[e];
                t_len[o] = S.klen[e];
            }
        __syncthreads();
        for (int e = tid; e < S.nkept; e += T) {
            S.kkey[e] = t_key[e];
            S.koff[e] = t_off[e];
            S.klen[e] = t_len[e];
        }
        __syncthreads();
    }
#ifdef RMCV_PROFILE
    CSTAMP();
    if (tid == 0 && (blockIdx.x == 100 || blockIdx.x == 1))
        printf("[cycles b%d nn=%d slots=%d cand=%d rounds=%d] ringtab+spos %.1f N1+scan %.1f N2 %.1f N3 %.1f N4 %.1f N5 %.1f N6 %.1f N7 %.1f N8+kept %.1f us\n", (int)blockIdx.x, nn, nslots, ncand, rounds,
               (tc_[2] - tc_[0]) / 100.0, (tc_[3] - tc_[2]) / 100.0, (tc_[4] - tc_[3]) / 100.0, (tc_[5] - tc_[4]) / 100.0,
               (tc_[6] - tc_[5]) / 100.0, (tc_[7] - tc_[6]) / 100.0, (tc_[8] - tc_[7]) / 100.0, (tc_[9] - tc_[8]) / 100.0, (tc_[10] - tc_[9]) / 100.0);
#endif
}


// ---- mid tier: the same cycle formulation with its tables in GLOBAL memory ---------------------------------------------------
// cv::findContours has no bound (src/imgproc.cpp:71-72).  A frame beyond what the LDS tables hold (VISIT_CAP visits, SLOT_CAP
// non-empty words, KEPT_CAP contours, MULTI_CAP junction pixels) used to go straight to the literal scanner -- one wavefront
// walking borders step by step, ~0.3 us per step, and the rest of its launch waiting for it.  This tier keeps the parallel
// formulation and moves the per-word and per-visit tables into a scratch block of the workgroup's own in HBM (L2-resident for
// frames a little over the LDS tier): up to NN_MID visits, every word of the frame, CAND_MID outer borders, any number of
// junction pixels (the 3- and 4-visit pixels get mask planes of their own instead of a side list).  The row tables (non-empty
// rows, their word masks, slot bases) stay in LDS.  Differences from cycles_frame that the size forces:
//   * node ids are 32-bit; (smallest id of the cycle | doubling pointer) and (distance to the start | doubling pointer) are
//     packed into ONE 64-bit word each, read and written whole: a sweep of the pointer doubling then needs no read-all /
//     write-all phases (a word is a consistent segment description whenever it is read), only a barrier per sweep;
//   * discovery ranks and point offsets come from prefix sums over the candidates in node order (= raster order of the
//     starts) instead of an all-pairs ranking.
// The literal scanner remains the last resort (more visits than NN_MID, wider/taller than the row tables, verification failed).
struct MidTables {
    unsigned long long *lab, *neg, *bmask, *e2, *e3, *e4; // [slot_cap] per non-empty word
    uint32_t *nbase, *spos;                               // [slot_cap]
    unsigned long long *link, *dist;                      // [NN_MID]   mn << 32 | jp    and    dist << 32 | jp
    uint32_t *pxy, *succ;                                 // [NN_MID]
    uint32_t *cand, *klen, *koff, *kacc;                  // [CAND_MID]
    int slot_cap;
};
__device__ __forceinline__ MidTables mid_tables(uint8_t* base, int slot_cap)
{
    MidTables M;
    unsigned long long* q = reinterpret_cast<unsigned long long*>(base);
    M.lab = q; q += slot_cap;
    M.neg = q; q += slot_cap;
    M.bmask = q; q += slot_cap;
    M.e2 = q; q += slot_cap;
    M.e3 = q; q += slot_cap;
    M.e4 = q; q += slot_cap;
    M.link = q; q += NN_MID;
    M.dist = q; q += NN_MID;
    uint32_t* r = reinterpret_cast<uint32_t*>(q);
    M.nbase = r; r += slot_cap;
    M.spos = r; r += slot_cap;
    M.pxy = r; r += NN_MID;
    M.succ = r; r += NN_MID;
    M.cand = r; r += CAND_MID;
    M.klen = r; r += CAND_MID;
    M.koff = r; r += CAND_MID;
    M.kacc = r; r += CAND_MID;
    M.slot_cap = slot_cap;
    return M;
}
__device__ __forceinline__ unsigned long long ld64(const unsigned long long* p)
{ // one 8-byte access, never split or cached in a register across a sweep
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void st64(unsigned long long* p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// exclusive prefix of one value per thread over the workgroup (scan[] holds T ints); *total = the sum
template <int T>
__device__ __forceinline__ int wg_scan_excl(int* scan, int tid, int v, int* total)
{
    scan[tid] = v;
    __syncthreads();
    for (int d = 1; d < T; d <<= 1) {
        const int t = tid >= d ? scan[tid - d] : 0;
        __syncthreads();
        scan[tid] += t;
        __syncthreads();
    }
    const int incl = scan[tid];
    *total = scan[T - 1];
    __syncthreads();
    return incl - v;
}

// findContours of one frame on the mid tier.  On success (no FL in S.flags): points, cs[rank], cl[rank] (discovery order) are
// written, *nc_out / *np_out hold the counts.  FL is OR-ed into S.flags when the frame is beyond this tier too.
template <int T>
__device__ __forceinline__ void cycles_frame_mid(ContoursLds& S, const RowTabs& RT, const MidTables& M, const uint64_t* __restrict__ F, int prow, int h, int ww, int nrows,
                                 int tid, rmcv_point* __restrict__ pts, int32_t* __restrict__ cs, int32_t* __restrict__ cl,
                                 int max_points, int max_contours, int FL, int* nc_out, int* np_out)
{
    const LabelStore LS = {RT.rowmask, RT.rowbase, nullptr, nullptr};
#ifdef RMCV_PROFILE
    long long tm_[12]; int tmi_ = 0;
#define MSTAMP() do { __syncthreads(); if (tmi_ < 12) tm_[tmi_++] = wall_clock64(); } while (0)
#else
#define MSTAMP() do {} while (0)
#endif
    MSTAMP();
    for (int i = tid; i < 256; i += T) S.ringtab[i] = RINGTAB.v[i];
    if (tid == 0) { S.nnodes = 0; S.revoked = 0; }
    const int nslots = S.nslots;
    if (nslots > M.slot_cap || nslots > 65535) { // (rowbase is 16 bits wide)
        if (tid == 0) S.flags |= FL;
        __syncthreads();
        return;
    }
    __syncthreads();
    // slot -> (row, word)
    for (int r = tid; r < nrows; r += T) {
        const int y = RT.rows[r];
        uint32_t occ = RT.rowmask[y];
        int slot = RT.rowbase[y];
        while (occ) {
            const int k = __ffs((int)occ) - 1;
            occ &= occ - 1;
            M.spos[slot++] = (uint32_t)(y | (k << 11));
        }
    }
    __syncthreads();
    // The loops below are LATENCY-bound: the tables live in global memory (L2 hits at best, ~1 us a round trip beside the streaming
    // kernels) and one workgroup has few threads to hide that with.  So every loop handles U items per thread per trip, with all the
    // loads of a dependency level issued before the first is used: a trip costs one round trip per LEVEL instead of one per item
    // (first version, one item per trip: 700 us for a frame of 8 500 visits with 4 wavefronts).
    // ---- M1: per non-empty word: border pixels, pixels visited >= 2, >= 3, 4 times; node count
    for (int s0 = tid; s0 < nslots; s0 += 2 * T) {
        uint32_t sp[2];
        uint64_t wd[2][9];
#pragma unroll
        for (int u = 0; u < 2; u++) sp[u] = s0 + u * T < nslots ? M.spos[s0 + u * T] : 0u;
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int y = sp[u] & 2047, k = sp[u] >> 11;
            const int64_t base = (int64_t)(y + 1) * prow + 1 + k;
            const bool ok = s0 + u * T < nslots;
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int c = 0; c < 3; c++) wd[u][3 * r + c] = ok ? F[base + (r - 1) * prow + (c - 1)] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int slot = s0 + u * T;
            if (slot >= nslots) continue;
            const uint64_t ul = wd[u][0], uc = wd[u][1], ur = wd[u][2], ml = wd[u][3], mc = wd[u][4], mr = wd[u][5], dl = wd[u][6], dc = wd[u][7], dr = wd[u][8];
            uint64_t B, E2, E3, E4;
            visit_masks(ul, uc, ur, ml, mc, mr, dl, dc, dr, &B, &E2, &E3, &E4);
            M.bmask[slot] = B;
            M.e2[slot] = E2;
            M.e3[slot] = E3;
            M.e4[slot] = E4;
            M.nbase[slot] = (uint32_t)(__popcll(B) + __popcll(E2) + __popcll(E3) + __popcll(E4));
        }
    }
    __syncthreads();
    int nn;
    { // exclusive prefix of the node counts over the slots (raster order); a thread's slots are consecutive, read eight at a time
        const int per = (nslots + T - 1) / T, lo = tid * per, hi = lo + per < nslots ? lo + per : nslots;
        int sum = 0;
        for (int i0 = lo; i0 < hi; i0 += 8) {
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = i0 + u < hi ? M.nbase[i0 + u] : 0u;
#pragma unroll
            for (int u = 0; u < 8; u++) sum += (int)v[u];
        }
        int run = wg_scan_excl<T>(S.scan, tid, sum, &nn);
        for (int i0 = lo; i0 < hi; i0 += 8) {
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = i0 + u < hi ? M.nbase[i0 + u] : 0u;
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (i0 + u < hi) {
                    M.nbase[i0 + u] = (uint32_t)run;
                    run += (int)v[u];
                }
        }
    }
    if (nn > NN_MID) {
        if (tid == 0) S.flags |= FL;
        __syncthreads();
        return;
    }
    if (tid == 0) S.nnodes = nn;
    __syncthreads();
    MSTAMP(); // 1: M0 + M1 + prefix
    // ---- M2: the nodes, one border PIXEL per lane (see N2 of cycles_frame): a wavefront stages 64 words -- centre words, edge bits, the four
    // visit masks, node base, position -- in the LDS the tier-0 tables do not need here, scatters (word, bit) of their border pixels into a
    // list and takes list entries.  (Word-wise: 100-130 us of a 380 us frame of 8 500 visits, as long as the fullest word of every 64.)
    {
        constexpr int NWV = T / 64;
        constexpr int WST = 64 * (8 * 8 + 2 * 4);                           // staged bytes per wavefront
        constexpr int LCAP = (int)(3 * NN_CAP * sizeof(uint16_t)) / NWV / 2; // list entries per pass (the node tables n_a .. n_d)
        static_assert(NWV * WST <= (int)offsetof(ContoursLds, scan), "the staged words fit the tier-0 tables");
        const int lane = tid & 63, wave = tid >> 6;
        unsigned char* const stg = reinterpret_cast<unsigned char*>(S.lab) + wave * WST;
        unsigned long long* const sw = reinterpret_cast<unsigned long long*>(stg); // [8][64]: uc, mc, dc, edge bits, B, E2, E3, E4
        uint32_t* const sn = reinterpret_cast<uint32_t*>(stg + 64 * 64);            // [2][64]: nbase, spos
        uint16_t* const list = S.n_a + wave * LCAP;
        for (int c0 = wave * 64; c0 < nslots; c0 += T) { // (wave-uniform)
            const int slot = c0 + lane;
            uint64_t B = 0;
            if (slot < nslots) {
                const uint32_t sp = M.spos[slot];
                const int y = sp & 2047, k = sp >> 11;
                const int64_t base = (int64_t)(y + 1) * prow + 1 + k;
                const uint64_t ul = F[base - prow - 1], uc = F[base - prow], ur = F[base - prow + 1];
                const uint64_t ml = F[base - 1], mc = F[base], mr = F[base + 1];
                const uint64_t dl = F[base + prow - 1], dc = F[base + prow], dr = F[base + prow + 1];
                B = M.bmask[slot];
                sw[lane] = uc;
                sw[64 + lane] = mc;
                sw[128 + lane] = dc;
                sw[192 + lane] = (ul >> 63) | ((ur & 1ull) << 1) | ((ml >> 63) << 2) | ((mr & 1ull) << 3) | ((dl >> 63) << 4) | ((dr & 1ull) << 5);
                sw[256 + lane] = B;
                sw[320 + lane] = M.e2[slot];
                sw[384 + lane] = M.e3[slot];
                sw[448 + lane] = M.e4[slot];
                sn[lane] = M.nbase[slot];
                sn[64 + lane] = sp;
            }
            const int pc = __popcll(B);
            int incl = pc;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int v = __shfl_up(incl, d);
                if (lane >= d) incl += v;
            }
            const int excl = incl - pc, P = __shfl(incl, 63);
            for (int p0 = 0; p0 < P; p0 += LCAP) { // (wave-uniform)
                uint64_t rem = B;
                int r = excl - p0;
                while (rem) {
                    const int b = __ffsll((long long)rem) - 1;
                    rem &= rem - 1;
                    if (r >= 0 && r < LCAP) list[r] = (uint16_t)((lane << 6) | b);
                    r++;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int m = P - p0 < LCAP ? P - p0 : LCAP;
                for (int j = lane; j < m; j += 64) {
                    const int en = list[j], sl = en >> 6, bb = en & 63;
                    const uint64_t uc = sw[sl], mc = sw[64 + sl], dc = sw[128 + sl], eb = sw[192 + sl];
                    const uint32_t e = S.ringtab[ring_of(bb, (eb & 1ull) << 63, uc, (eb >> 1) & 1ull, ((eb >> 2) & 1ull) << 63, mc, (eb >> 3) & 1ull,
                                                         ((eb >> 4) & 1ull) << 63, dc, (eb >> 5) & 1ull)];
                    const int cnt = (int)(e & 7u);
                    const uint32_t sp = sn[64 + sl];
                    const int y = sp & 2047, k = sp >> 11;
                    const uint32_t xy = (uint32_t)(k * 64 + bb) | ((uint32_t)y << 12);
                    const uint64_t below = (1ull << bb) - 1;
                    uint32_t id = sn[sl] + (uint32_t)(__popcll(sw[256 + sl] & below) + __popcll(sw[320 + sl] & below) + __popcll(sw[384 + sl] & below) +
                                                      __popcll(sw[448 + sl] & below));
                    for (int a2 = 0; a2 < cnt && a2 < 4; a2++) {
                        const uint32_t back = (e >> (3 + 3 * a2)) & 7u, nextd = (e >> (15 + 3 * a2)) & 7u, ng = (e >> (28 + a2)) & 1u;
                        const uint32_t west = a2 == 0 ? (e >> 27) & 1u : 0u;
                        M.pxy[id++] = xy | (back << 24) | (ng << 27) | (west << 28) | (nextd << 29);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier(); // the list (and, behind the last pass, the staged words) are rewritten
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    }
    __syncthreads();
    MSTAMP(); // 2: M2
    // ---- M3: successor of every node
    for (int i0 = tid; i0 < nn; i0 += 4 * T) {
        uint32_t p[4], id0[4], succ[4];
        int slot2[4], cnt2[4];
        unsigned long long B2[4], E2[4], E3[4], E4[4];
        uint32_t nb[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; u++) p[u] = i0 + u * T < nn ? M.pxy[i0 + u * T] : 0u;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int x = (int)(p[u] & 0xFFFu), y = (int)((p[u] >> 12) & 0xFFFu), nd = (int)(p[u] >> 29);
            const int xs = x + dir_dx(nd), ys = y + dir_dy(nd);
            const int ks = xs >> 6;
            ok[u] = i0 + u * T < nn && xs >= 0 && ys >= 0 && ys < h && ks < ww && ((RT.rowmask[ys < h && ys >= 0 ? ys : 0] >> ks) & 1u);
            slot2[u] = ok[u] ? LS.slot(ys, ks) : 0;
            B2[u] = M.bmask[slot2[u]];
            E2[u] = M.e2[slot2[u]];
            E3[u] = M.e3[slot2[u]];
            E4[u] = M.e4[slot2[u]];
            nb[u] = M.nbase[slot2[u]];
        }
        uint32_t alt[4][3];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int x = (int)(p[u] & 0xFFFu), nd = (int)(p[u] >> 29);
            const int bs = (x + dir_dx(nd)) & 63;
            ok[u] = ok[u] && ((B2[u] >> bs) & 1ull);
            const uint64_t below = (1ull << bs) - 1;
            id0[u] = nb[u] + (uint32_t)(__popcll(B2[u] & below) + __popcll(E2[u] & below) + __popcll(E3[u] & below) + __popcll(E4[u] & below));
            cnt2[u] = 1 + (int)((E2[u] >> bs) & 1ull) + (int)((E3[u] >> bs) & 1ull) + (int)((E4[u] >> bs) & 1ull);
            // the further visits of the successor pixel (almost never present: a pixel the border passes more than once)
#pragma unroll
            for (int a2 = 1; a2 < 4; a2++) alt[u][a2 - 1] = (ok[u] && a2 < cnt2[u]) ? M.pxy[id0[u] + a2] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + u * T;
            if (i >= nn) continue;
            const int x = (int)(p[u] & 0xFFFu), y = (int)((p[u] >> 12) & 0xFFFu), nd = (int)(p[u] >> 29);
            succ[u] = (uint32_t)i;
            if (ok[u]) {
                const uint32_t back2 = (uint32_t)((nd + 4) & 7);
                succ[u] = id0[u]; // the visit of the successor pixel whose back direction points here
#pragma unroll
                for (int a2 = 1; a2 < 4; a2++)
                    if (a2 < cnt2[u] && ((alt[u][a2 - 1] >> 24) & 7u) == back2) succ[u] = id0[u] + (uint32_t)a2;
            } else { // only an isolated pixel has no successor (it stays where it is); anything else contradicts the bijection
                const int k0 = x >> 6;
                const int64_t base = (int64_t)(y + 1) * prow + 1;
                const uint32_t ring = ring_of(x & 63, F[base - prow + k0 - 1], F[base - prow + k0], F[base - prow + k0 + 1], F[base + k0 - 1],
                                              F[base + k0], F[base + k0 + 1], F[base + prow + k0 - 1], F[base + prow + k0], F[base + prow + k0 + 1]);
                if (ring != 0) atomicOr(&S.flags, FL);
            }
            M.succ[i] = succ[u];
            M.link[i] = ((unsigned long long)(uint32_t)i << 32) | succ[u];
        }
    }
    __syncthreads();
    if (S.flags & FL) return;
    MSTAMP(); // 3: M3
    // ---- M4: smallest node id of every cycle.  A word (mn, jp) always describes a true segment [i, jp) of the cycle with its
    // minimum, so a sweep may read words other threads have already advanced: after `rounds` sweeps every segment is at least
    // nn long, i.e. covers its whole cycle.
    int rounds = 0;
    while ((1 << rounds) < nn) rounds++;
    // The doubling sweeps are rounds x 2 dependent accesses per node.  Up to NN_LDS nodes they run on a packed 32-bit word per node
    // (mn or dist : 16 | jp : 16) in the LDS the tier-0 tables do not need here (lab .. multi, 38 KB) -- an LDS round trip is a tenth
    // of an L2 one; the final values go to the global tables the later phases read.  Larger frames sweep the global tables.
    constexpr int NN_LDS = MID_LDS_WORDS;
    uint32_t* const R = reinterpret_cast<uint32_t*>(S.lab);
    const bool in_lds = nn <= NN_LDS && nn <= 65535;
    if (in_lds) {
        for (int i0 = tid; i0 < nn; i0 += 8 * T) {
            uint32_t sc[8];
#pragma unroll
            for (int u = 0; u < 8; u++) sc[u] = i0 + u * T < nn ? M.succ[i0 + u * T] : 0u;
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (i0 + u * T < nn) R[i0 + u * T] = ((uint32_t)(i0 + u * T) << 16) | sc[u];
        }
        __syncthreads();
        for (int rd = 0; rd < rounds; rd++) {
            for (int i0 = tid; i0 < nn; i0 += 8 * T) {
                uint32_t w[8], wt[8];
#pragma unroll
                for (int u = 0; u < 8; u++) w[u] = i0 + u * T < nn ? R[i0 + u * T] : 0u;
#pragma unroll
                for (int u = 0; u < 8; u++) wt[u] = R[w[u] & 0xFFFFu];
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (i0 + u * T < nn) R[i0 + u * T] = (((w[u] >> 16) < (wt[u] >> 16) ? (w[u] >> 16) : (wt[u] >> 16)) << 16) | (wt[u] & 0xFFFFu);
            }
            __syncthreads();
        }
        // mn -> the global table; the next array in R: dist | jp with the start absorbing
        for (int i0 = tid; i0 < nn; i0 += 8 * T) {
            uint32_t w[8], sc[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                w[u] = i0 + u * T < nn ? R[i0 + u * T] : 0u;
                sc[u] = i0 + u * T < nn ? M.succ[i0 + u * T] : 0u;
            }
            // (a thread overwrites only the words it has just read: no barrier needed inside this loop)
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (i0 + u * T < nn) {
                    const uint32_t i = (uint32_t)(i0 + u * T), mn = w[u] >> 16;
                    M.link[i] = (unsigned long long)mn << 32;
                    R[i] = mn == i ? i : ((1u << 16) | sc[u]);
                }
        }
        __syncthreads();
        MSTAMP(); // 4: M4
        for (int rd = 0; rd < rounds; rd++) {
            for (int i0 = tid; i0 < nn; i0 += 8 * T) {
                uint32_t w[8], wt[8];
#pragma unroll
                for (int u = 0; u < 8; u++) w[u] = i0 + u * T < nn ? R[i0 + u * T] : 0u;
#pragma unroll
                for (int u = 0; u < 8; u++) wt[u] = R[w[u] & 0xFFFFu];
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (i0 + u * T < nn) R[i0 + u * T] = (((w[u] >> 16) + (wt[u] >> 16)) << 16) | (wt[u] & 0xFFFFu);
            }
            __syncthreads();
        }
        for (int i = tid; i < nn; i += T) M.dist[i] = (unsigned long long)(R[i] >> 16) << 32;
        __syncthreads();
    } else {
    for (int rd = 0; rd < rounds; rd++) {
        for (int i0 = tid; i0 < nn; i0 += 8 * T) {
            unsigned long long w[8], wt[8];
#pragma unroll
            for (int u = 0; u < 8; u++) w[u] = i0 + u * T < nn ? ld64(M.link + i0 + u * T) : 0ull;
#pragma unroll
            for (int u = 0; u < 8; u++) wt[u] = ld64(M.link + (uint32_t)w[u]);
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (i0 + u * T < nn) {
                    const unsigned long long mn = (w[u] >> 32) < (wt[u] >> 32) ? (w[u] >> 32) : (wt[u] >> 32);
                    st64(M.link + i0 + u * T, (mn << 32) | (uint32_t)wt[u]);
                }
        }
        __syncthreads();
    }
    MSTAMP(); // 4: M4
    // ---- M5: steps from every node FORWARD to its cycle's start (the start absorbs)
    for (int i0 = tid; i0 < nn; i0 += 8 * T) {
        unsigned long long w[8];
        uint32_t sc[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            w[u] = i0 + u * T < nn ? ld64(M.link + i0 + u * T) : 0ull;
            sc[u] = i0 + u * T < nn ? M.succ[i0 + u * T] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (i0 + u * T < nn) {
                const uint32_t i = (uint32_t)(i0 + u * T);
                M.dist[i] = (uint32_t)(w[u] >> 32) == i ? (unsigned long long)i : ((1ull << 32) | sc[u]);
            }
    }
    __syncthreads();
    for (int rd = 0; rd < rounds; rd++) {
        for (int i0 = tid; i0 < nn; i0 += 8 * T) {
            unsigned long long w[8], wt[8];
#pragma unroll
            for (int u = 0; u < 8; u++) w[u] = i0 + u * T < nn ? ld64(M.dist + i0 + u * T) : 0ull;
#pragma unroll
            for (int u = 0; u < 8; u++) wt[u] = ld64(M.dist + (uint32_t)w[u]);
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (i0 + u * T < nn) st64(M.dist + i0 + u * T, (((w[u] >> 32) + (wt[u] >> 32)) << 32) | (uint32_t)wt[u]);
        }
        __syncthreads();
    }
    }
    MSTAMP(); // 5: M5
    // ---- M6: candidates = cycles whose start visit contains the west neighbour, numbered in node order (raster order of the starts)
    int ncand;
    {
        const int per = (nn + T - 1) / T;
        const int lo = tid * per, hi = (lo + per < nn) ? lo + per : nn;
        int cnt = 0;
        for (int i0 = lo; i0 < hi; i0 += 8) {
            unsigned long long w[8];
            uint32_t p[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                w[u] = i0 + u < hi ? M.link[i0 + u] : ~0ull;
                p[u] = i0 + u < hi ? M.pxy[i0 + u] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) cnt += (i0 + u < hi) && ((uint32_t)(w[u] >> 32) == (uint32_t)(i0 + u)) && ((p[u] >> 28) & 1u);
        }
        int e = wg_scan_excl<T>(S.scan, tid, cnt, &ncand);
        if (ncand <= CAND_MID) {
            for (int i0 = lo; i0 < hi; i0 += 8) {
                unsigned long long w[8], dl_[8];
                uint32_t p[8], sc[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    w[u] = i0 + u < hi ? M.link[i0 + u] : ~0ull;
                    p[u] = i0 + u < hi ? M.pxy[i0 + u] : 0u;
                    sc[u] = i0 + u < hi ? M.succ[i0 + u] : 0u;
                }
#pragma unroll
                for (int u = 0; u < 8; u++) { // (only the starts' successors matter; the others read a harmless word)
                    const bool start = (i0 + u < hi) && (uint32_t)(w[u] >> 32) == (uint32_t)(i0 + u);
                    dl_[u] = M.dist[start ? sc[u] : 0u];
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int i = i0 + u;
                    if (i >= hi || (uint32_t)(w[u] >> 32) != (uint32_t)i) continue;
                    if ((p[u] >> 28) & 1u) {
                        M.cand[e] = (uint32_t)i;
                        M.klen[e] = (uint32_t)(dl_[u] >> 32) + 1u;
                        M.kacc[e] = 1u;
                        M.succ[i] = (uint32_t)e; // from here on: start node -> candidate index
                        e++;
                    } else M.succ[i] = 0xFFFFFFFFu; // a hole border
                }
            }
        }
    }
    if (ncand > CAND_MID) {
        if (tid == 0) S.flags |= FL;
        __syncthreads();
        return;
    }
    __syncthreads();
    MSTAMP(); // 6: M6
    // ---- M7: RETR_EXTERNAL as the fixed point of "label the accepted borders, revoke the starts whose nearest labelled pixel to
    // the left is positive" (see cycles_frame)
    for (int round = 0;; round++) {
        if (tid == 0) S.revoked = 0;
        for (int i = tid; i < nslots; i += T) { M.lab[i] = 0; M.neg[i] = 0; }
        __syncthreads();
        for (int i0 = tid; i0 < nn; i0 += 4 * T) {
            unsigned long long w[4];
            uint32_t p[4], e[4], ka[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                w[u] = i0 + u * T < nn ? M.link[i0 + u * T] : 0ull;
                p[u] = i0 + u * T < nn ? M.pxy[i0 + u * T] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) e[u] = M.succ[(uint32_t)(w[u] >> 32)];
#pragma unroll
            for (int u = 0; u < 4; u++) ka[u] = e[u] != 0xFFFFFFFFu ? M.kacc[e[u]] : 0u;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (i0 + u * T >= nn || !ka[u]) continue;
                const int x = (int)(p[u] & 0xFFFu), y = (int)((p[u] >> 12) & 0xFFFu);
                const int slot = LS.slot(y, x >> 6);
                atomicOr(&M.lab[slot], 1ull << (x & 63));
                if ((p[u] >> 27) & 1u) atomicOr(&M.neg[slot], 1ull << (x & 63));
            }
        }
        __syncthreads();
        for (int e0 = tid; e0 < ncand; e0 += 2 * T) {
            uint32_t ka[2], cn[2], p[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                ka[u] = e0 + u * T < ncand ? M.kacc[e0 + u * T] : 0u;
                cn[u] = e0 + u * T < ncand ? M.cand[e0 + u * T] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 2; u++) p[u] = M.pxy[cn[u]];
            unsigned long long l[2];
            int kk[2];
            uint32_t left[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int x0 = (int)(p[u] & 0xFFFu), y0 = (int)((p[u] >> 12) & 0xFFFu);
                kk[u] = x0 >> 6;
                // (the labels were OR-ed in by L2 atomics: read them past the vector L1, like the literal scanner does)
                l[u] = ka[u] ? (ld_l2(M.lab + LS.slot(y0, kk[u])) & ((1ull << (x0 & 63)) - 1)) : 0ull;
                left[u] = ka[u] ? (RT.rowmask[y0] & ((1u << kk[u]) - 1u)) : 0u;
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                if (!ka[u]) continue;
                const int y0 = (int)((p[u] >> 12) & 0xFFFu);
                while (!l[u] && left[u]) {
                    kk[u] = 31 - __clz((int)left[u]);
                    left[u] &= ~(1u << kk[u]);
                    l[u] = ld_l2(M.lab + LS.slot(y0, kk[u]));
                }
                if (l[u]) {
                    const int top = 63 - __clzll((long long)l[u]);
                    if (!((ld_l2(M.neg + LS.slot(y0, kk[u])) >> top) & 1ull)) { // positive: inside a hole of that border
                        M.kacc[e0 + u * T] = 0u;
                        S.revoked = 1;
                    }
                }
            }
        }
        __syncthreads();
        const int again = S.revoked;
        __syncthreads();
        if (!again) break;
        if (round >= 32) {
            if (tid == 0) S.flags |= FL;
            __syncthreads();
            return;
        }
    }
    MSTAMP(); // 7: M7
    // ---- M8: discovery rank and point offset of every accepted border (prefix sums in candidate order), then the points
    int nacc, npts;
    {
        const int per = (ncand + T - 1) / T;
        const int lo = tid * per, hi = (lo + per < ncand) ? lo + per : ncand;
        int cnt = 0, plen = 0;
        for (int e0 = lo; e0 < hi; e0 += 8) {
            uint32_t ka[8], kl[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                ka[u] = e0 + u < hi ? M.kacc[e0 + u] : 0u;
                kl[u] = e0 + u < hi ? M.klen[e0 + u] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (ka[u]) { cnt++; plen += (int)kl[u]; }
        }
        int rank = wg_scan_excl<T>(S.scan, tid, cnt, &nacc);
        int off = wg_scan_excl<T>(S.scan, tid, plen, &npts);
        if (nacc <= max_contours && npts <= max_points) {
            for (int e0 = lo; e0 < hi; e0 += 8) {
                uint32_t ka[8], kl[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    ka[u] = e0 + u < hi ? M.kacc[e0 + u] : 0u;
                    kl[u] = e0 + u < hi ? M.klen[e0 + u] : 0u;
                }
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (ka[u]) {
                        M.koff[e0 + u] = (uint32_t)off;
                        cs[rank] = off;
                        cl[rank] = (int32_t)kl[u];
                        off += (int)kl[u];
                        rank++;
                    }
            }
        }
    }
    if (nacc > max_contours || npts > max_points) { // the literal scanner reports the overflow the way it always did
        if (tid == 0) S.flags |= FL;
        __syncthreads();
        return;
    }
    __syncthreads();
    for (int i0 = tid; i0 < nn; i0 += 4 * T) {
        unsigned long long w[4], dd[4];
        uint32_t p[4], e[4], ka[4], kl[4], ko[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const bool ok = i0 + u * T < nn;
            w[u] = ok ? M.link[i0 + u * T] : 0ull;
            p[u] = ok ? M.pxy[i0 + u * T] : 0u;
            dd[u] = ok ? M.dist[i0 + u * T] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) e[u] = M.succ[(uint32_t)(w[u] >> 32)];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t ee = e[u] != 0xFFFFFFFFu ? e[u] : 0u;
            ka[u] = e[u] != 0xFFFFFFFFu ? M.kacc[ee] : 0u;
            kl[u] = M.klen[ee];
            ko[u] = M.koff[ee];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (i0 + u * T >= nn || !ka[u]) continue;
            rmcv_point q;
            q.x = (int)(p[u] & 0xFFFu);
            q.y = (int)((p[u] >> 12) & 0xFFFu);
            const int len = (int)kl[u], d = (int)(dd[u] >> 32), pos = d ? len - d : 0;
            if (pos >= 0 && pos < len) pts[ko[u] + pos] = q;
        }
    }
    MSTAMP(); // 8: M8
    // ---- V2 on the final labels: every unlabelled run start would have been rejected by the scanner (nearest labelled pixel to
    // its left positive).  (V1 -- every accepted start acceptable -- is what the last round of M7 established.)
    for (int r = tid; r < nrows; r += T) {
        const int y = RT.rows[r];
        const int64_t base = (int64_t)(y + 1) * prow + 1;
        uint32_t rem = RT.rowmask[y];
        int slot = RT.rowbase[y];
        uint64_t carry = 0;
        bool last_pos = false;
        int kprev = -2;
        while (rem) {
            const int k = __ffs((int)rem) - 1;
            rem &= rem - 1;
            const uint64_t fwd = F[base + k];
            if (k != kprev + 1) carry = 0;
            kprev = k;
            const unsigned long long l = ld_l2(M.lab + slot), ng = ld_l2(M.neg + slot);
            slot++;
            uint64_t cnd = fwd & ~((fwd << 1) | carry) & ~l;
            while (cnd) {
                const int b = __ffsll((long long)cnd) - 1;
                cnd &= cnd - 1;
                const uint64_t below = l & ((1ull << b) - 1);
                const bool pos = below ? !((ng >> (63 - __clzll((long long)below))) & 1ull) : last_pos;
                if (!pos) atomicOr(&S.flags, FL);
            }
            if (l) last_pos = !((ng >> (63 - __clzll((long long)l))) & 1ull);
            carry = fwd >> 63;
        }
    }
    __syncthreads();
    MSTAMP(); // 9: V2
#ifdef RMCV_PROFILE
    if (tid == 0 && blockIdx.x < 2)
        printf("[mid b%d nn=%d slots=%d cand=%d acc=%d rounds=%d] M0-1 %.1f M2 %.1f M3 %.1f M4 %.1f M5 %.1f M6 %.1f M7 %.1f M8 %.1f V2 %.1f us\n", (int)blockIdx.x, nn,
               nslots, ncand, nacc, rounds, (tm_[1] - tm_[0]) / 100.0, (tm_[2] - tm_[1]) / 100.0, (tm_[3] - tm_[2]) / 100.0, (tm_[4] - tm_[3]) / 100.0,
               (tm_[5] - tm_[4]) / 100.0, (tm_[6] - tm_[5]) / 100.0, (tm_[7] - tm_[6]) / 100.0, (tm_[8] - tm_[7]) / 100.0, (tm_[9] - tm_[8]) / 100.0);
#endif
    *nc_out = nacc;
    *np_out = npts;
}

} // namespace rmcv
