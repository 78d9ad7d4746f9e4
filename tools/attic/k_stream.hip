// k_stream.hip -- K1, second form: the pixel-streaming part of rm::extract_color
// (/root/reference/src/imgproc.cpp:52-69: split + saturating channel subtract + inRange + 3x3 morphology) as a
// WAVE-PRIVATE ROW STREAM.  Same results, bit for bit, as k_binary.hip (which stays for geometries this form does not take).
//
// HBM-bound kernel (3 B/px read + 1 B/px written), no MFMA.  What the first form left on the table (profiles/r02a_*):
// with its loads compiled out it still took 0.14-0.18 ms of the 0.28 -- ~165 vector instructions per 16-pixel item, most of
// them index arithmetic and predication around the 60 that threshold, and four workgroup barriers per strip during which the
// workgroup has nothing in flight.  Here
//   * ONE wavefront owns a band of rows of one frame and streams down it SR rows at a time; it never meets another wave, so
//     there is no s_barrier anywhere -- a CU's 8-16 resident waves interleave freely;
//   * the loads of step s+1 (and of the next band's first step: bands are claimed one ahead) are issued BEFORE the
//     morphology and the stores of step s: every wave always has 15 x 1 KiB in flight;
//   * the lane -> (row, 16-pixel group) map of a step is the same for every step and is computed once; the thresholded
//     16-bit masks go to LDS with ds_write_b16 (no lane merging), the store phase reads them back with ds_read_u16;
//   * bit planes live in three wave-private LDS rings (thresholded T, dilated D, result E rows; 8 rows x (ww + 2) words),
//     3.3 KiB per wave at 1280 px -- rows are padded with one word either side that reads 0 (T) / all ones (D), so the
//     horizontal carries need no bounds tests.
// Border semantics (cv::morphologyDefaultBorderValue): samples outside the image never win -- T rows outside the frame are
// written as zeros, D rows outside the frame as ones.
#include <stdlib.h>

#include "rmcv_internal.h"

namespace rmcv {

static constexpr int SU = 5;     // 16-pixel items per lane and step (5 x 48 B x 64 lanes = 15 KiB in flight per wave)
static constexpr int SRING = 8;  // rows per LDS ring; a step of SR <= 4 rows keeps SR + 2 rows alive

typedef uint32_t su32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short su16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t su32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t s_expand4(uint32_t nib) { return (((nib & 0xFu) * 0x00204081u) & 0x01010101u) * 0xFFu; }

// 16 pixels (12 dwords) -> 16-bit mask of (a - b >= lb); two pixels per packed-16 operation (see k_binary.hip thresh16)
template <int CA, int CB>
__device__ __forceinline__ uint32_t s_thresh16(const su32x4 v0, const su32x4 v1, const su32x4 v2, uint32_t K)
{
    const uint32_t d[12] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w};
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int ia = 6 * j + CA, ib = 6 * j + CB;
        const uint32_t sa = (uint32_t)(ia & 3) | (0x0cu << 8) | ((uint32_t)((ia & 3) + 3) << 16) | (0x0cu << 24);
        const uint32_t sb = (uint32_t)(ib & 3) | (0x0cu << 8) | ((uint32_t)((ib & 3) + 3) << 16) | (0x0cu << 24);
        const uint32_t A = __builtin_amdgcn_perm(d[(ia >> 2) + 1 < 12 ? (ia >> 2) + 1 : 11], d[ia >> 2], sa);
        const uint32_t B = __builtin_amdgcn_perm(d[(ib >> 2) + 1 < 12 ? (ib >> 2) + 1 : 11], d[ib >> 2], sb);
        su16x2 t = __builtin_bit_cast(su16x2, A) + __builtin_bit_cast(su16x2, K);
        t = t - __builtin_bit_cast(su16x2, B);
        acc |= ((__builtin_bit_cast(uint32_t, t) >> 15) & 0x00010001u) << j;
    }
    acc = (acc | (acc << 4)) & 0x0F0F0F0Fu;
    acc = (acc | (acc << 2)) & 0x33333333u;
    acc = (acc | (acc << 1)) & 0x55555555u;
    return (acc & 0xFFFFu) | ((acc >> 16) << 1);
}

struct StreamArgs {
    const uint8_t* frames;
    int64_t frame_pitch;
    int stride, w, h, ww;
    int lb, all_pass, morph;
    uint8_t* binary; // image output (unused when the kernel is instantiated without it: RMCV_STAGE_NO_IMAGE)
    uint64_t* bits;
    int prow;
    int64_t plane_pitch;
    uint32_t* rowmask;
    int* ctr;        // [8] per-XCD queue heads + [8] leavers (see k_binary.hip)
    int SR;          // rows per step (1..4)
    int BR;          // rows per band
    int bands_per_frame, n_units, n_frames;
    int dbg;         // dev knob (RMCV_KS_DBG): 1 no loads, 2 no image stores, 4 no plane / row-mask stores -- ablations, results are wrong
};

struct Band {
    int f, b0, b1;
};

// Every vector-memory instruction of a step is UNCONDITIONAL: lanes (or whole items) that have nothing to move use an offset
// beyond the buffer's extent, which the buffer hardware drops (stores) or answers with zeros (loads: a row outside the image
// thresholds to 0 by itself).  That is what keeps the prefetch a prefetch: vmcnt counts loads and stores in issue order, and
// with a store count the compiler can see (MI + 1 + NU per step, no branch around any of them) the wait in front of the next
// step's thresholds is s_waitcnt vmcnt(stores of this step) -- the loads issued BEFORE those stores are waited for, the
// stores themselves stay in flight.  With one store behind a branch the wait degrades to vmcnt(0): the wave then sits out
// the write acknowledgements of its own stores, step after step.
static constexpr uint32_t OOB = 0xFFFFFF00u; // voffset of a lane that moves nothing (every extent here is below 4 GiB - 256)
static constexpr int RSRC3 = 0x00020000;     // raw buffer, 32-bit data format (gfx9 family)

// NU: item slots per lane in use (ceil(SR * wq / 64)); MI: 64-word passes of the morphology (ceil(SR * ww / 64)); IMG: write the byte image
template <int CA, int CB, int NU, int MI, bool IMG>
__global__ __launch_bounds__(64) void k_stream(const StreamArgs A)
{
    extern __shared__ uint64_t smem[];
    const int lane = threadIdx.x;
    const int ww = A.ww, wq = ww * 4, pw = ww + 2; // pw: padded words per ring row
    const int SR = A.SR, halo = A.morph, h = A.h;
    uint64_t* const T = smem;                  // [SRING][pw]  thresholded rows, pads read 0
    uint64_t* const D = smem + SRING * pw;     // [SRING][pw]  dilated rows, pads read ~0 (they never win the erode)
    uint64_t* const E = smem + 2 * SRING * pw; // [SR][ww]     result rows of the step
    for (int i = lane; i < SRING; i += 64) {
        T[i * pw] = 0;
        T[i * pw + ww + 1] = 0;
        D[i * pw] = ~0ull;
        D[i * pw + ww + 1] = ~0ull;
    }
    const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(A.frames), 0, (int)((int64_t)(A.n_frames - 1) * A.frame_pitch + (int64_t)(h - 1) * A.stride + 3 * A.w), RSRC3);
    const __amdgpu_buffer_rsrc_t r_bin = __builtin_amdgcn_make_buffer_rsrc(A.binary, 0, IMG ? (int)((int64_t)A.n_frames * A.w * h) : 0, RSRC3);
    const __amdgpu_buffer_rsrc_t r_plane = __builtin_amdgcn_make_buffer_rsrc(A.bits, 0, (int)((int64_t)A.n_frames * A.plane_pitch * 8), RSRC3);
    const __amdgpu_buffer_rsrc_t r_mask = __builtin_amdgcn_make_buffer_rsrc(A.rowmask, 0, (int)((int64_t)A.n_frames * h * 4), RSRC3);
    // lane -> (row of the step, 16-pixel group of the row) of its NU items: the same in every step
    const int items = SR * wq, words = SR * ww;
    const uint32_t r_wq = (uint32_t)((0x100000000ull + wq - 1) / wq), r_ww = (uint32_t)((0x100000000ull + ww - 1) / ww);
    int it_r[NU], it_q[NU]; // a lane beyond the step's items: row -16384 (no range test admits it)
#pragma unroll
    for (int u = 0; u < NU; u++) {
        const int it = lane + 64 * u;
        const int r = (int)__umulhi((uint32_t)it, r_wq); // wq >= 4
        it_r[u] = it < items ? r : -16384;
        it_q[u] = it < items ? it - r * wq : 0;
    }
    int mw_r[MI], mw_k[MI]; // same for the morphology: lane -> (row of the step, word of the row)
#pragma unroll
    for (int j = 0; j < MI; j++) {
        const int idx = lane + 64 * j;
        const int r = ww == 1 ? idx : (int)__umulhi((uint32_t)idx, r_ww);
        mw_r[j] = idx < words ? r : 16384; // beyond every row count
        mw_k[j] = idx < words ? idx - r * ww : 0;
    }
    const uint32_t kk = (uint32_t)(0x8000 - A.lb) & 0xFFFFu;
    const uint32_t K = kk | (kk << 16);

    // ---- band queue (per XCD, bands claimed one ahead so that the first loads of the next band are in flight early)
    const int xcd = blockIdx.x & 7;
    const int per_xcd = (A.n_units + 7) >> 3;
    auto claim = [&](Band& b) -> bool {
        int j = 0;
        if (lane == 0) j = atomicAdd(&A.ctr[xcd], 1);
        j = __builtin_amdgcn_readfirstlane(j); // every lane of the wave is active here
        const int L = xcd * per_xcd + j;
        if (j >= per_xcd || L >= A.n_units) return false;
        const int f = L / A.bands_per_frame, band = L - f * A.bands_per_frame;
        b.f = f;
        b.b0 = band * A.BR;
        b.b1 = min(h, b.b0 + A.BR);
        return true;
    };
    su32x4 v[NU][3];
    auto prefetch = [&](int f, int t, int y_end) {
        const uint32_t fbase = (uint32_t)((int64_t)f * A.frame_pitch);
#pragma unroll
        for (int u = 0; u < NU; u++) {
            const int y = t + it_r[u];
            const uint32_t off = (y >= 0 && y < h && y < y_end && !(A.dbg & 1)) ? fbase + (uint32_t)y * (uint32_t)A.stride + (uint32_t)it_q[u] * 48u : OOB;
            v[u][0] = __builtin_amdgcn_raw_buffer_load_b128(r_in, off, 0, 0);
            v[u][1] = __builtin_amdgcn_raw_buffer_load_b128(r_in, off, 16, 0);
            v[u][2] = __builtin_amdgcn_raw_buffer_load_b128(r_in, off, 32, 0);
        }
    };

    Band cur, nxt;
    bool have = claim(cur);
    bool have_nxt = have ? claim(nxt) : false;
    if (have) {
        prefetch(cur.f, cur.b0 - halo, cur.b1 + halo);
        // as many (dropped) stores behind these first loads as every later step issues behind its prefetch: the compiler sizes
        // the wait in front of the thresholds by the path with the FEWEST younger operations, and this path would have none
#pragma unroll
        for (int j = 0; j < MI + 1 + (IMG ? NU : 0); j++) __builtin_amdgcn_raw_buffer_store_b32(0u, r_mask, OOB + 4u * j, 0, 0); // distinct offsets: identical stores would be merged
    }
    while (have) {
        int t = cur.b0 - halo;         // next row to threshold
        int d = cur.b0 - (halo - 1);   // next row to dilate (halo >= 1)
        int o = cur.b0;                // next row to put out
        const uint32_t bin_base = (uint32_t)((int64_t)cur.f * A.w * h);
        const uint32_t plane_base = (uint32_t)((int64_t)cur.f * A.plane_pitch);
        while (o < cur.b1) {
            // ---------------- threshold the rows [t, t + SR) that are in registers -> T ring
#pragma unroll
            for (int u = 0; u < NU; u++) {
                const int y = t + it_r[u];
                uint32_t m = s_thresh16<CA, CB>(v[u][0], v[u][1], v[u][2], K); // a row outside the image was answered with zeros: m = 0
                if (A.all_pass) m = (y >= 0 && y < h) ? 0xFFFFu : 0u;
                if (it_r[u] >= 0) reinterpret_cast<uint16_t*>(T + (y & (SRING - 1)) * pw + 1)[it_q[u]] = (uint16_t)m;
            }
            t += SR;
            // ---------------- the registers are free: loads of the next step (or of the next band's first step)
            {
                const bool more = t < cur.b1 + halo;
                if (more || have_nxt) prefetch(more ? cur.f : nxt.f, more ? t : nxt.b0 - halo, (more ? cur.b1 : nxt.b1) + halo);
            }
            __syncthreads(); // one wave per workgroup: orders the LDS writes above against the reads below, no s_barrier

            // ---------------- dilate rows [d, d1) -> D ring
            int o1;
            if (halo >= 1) {
                const int d1 = max(d, min(t - 1, cur.b1 + (halo - 1)));
#pragma unroll
                for (int j = 0; j < MI; j++) {
                    const int yy = d + mw_r[j], k = mw_k[j];
                    if (yy < d1) {
                        uint64_t dv = ~0ull; // a row outside the image never wins the erode
                        if (yy >= 0 && yy < h) {
                            const uint64_t* t0 = T + ((yy - 1) & (SRING - 1)) * pw + 1 + k;
                            const uint64_t* t1 = T + (yy & (SRING - 1)) * pw + 1 + k;
                            const uint64_t* t2 = T + ((yy + 1) & (SRING - 1)) * pw + 1 + k;
                            const uint64_t c = t0[0] | t1[0] | t2[0];
                            const uint64_t l = (t0[-1] | t1[-1] | t2[-1]) >> 63;
                            const uint64_t r = (t0[1] | t1[1] | t2[1]) << 63;
                            dv = c | (c << 1) | l | (c >> 1) | r;
                        }
                        D[(yy & (SRING - 1)) * pw + 1 + k] = dv;
                    }
                }
                d = d1;
                o1 = max(o, halo == 2 ? min(d1 - 1, cur.b1) : min(d1, cur.b1));
                __syncthreads();
            } else {
                o1 = min(t, cur.b1);
            }
            // ---------------- result rows [o, o1): erode of D (CLOSE), D (DILATE) or T (NONE) -> E + the frame's bit plane
#pragma unroll
            for (int j = 0; j < MI; j++) {
                const int y = o + mw_r[j], k = mw_k[j];
                uint64_t e = 0;
                if (y < o1) {
                    if (halo == 2) {
                        const uint64_t* d0 = D + ((y - 1) & (SRING - 1)) * pw + 1 + k;
                        const uint64_t* d1p = D + (y & (SRING - 1)) * pw + 1 + k;
                        const uint64_t* d2 = D + ((y + 1) & (SRING - 1)) * pw + 1 + k;
                        const uint64_t c = d0[0] & d1p[0] & d2[0];
                        const uint64_t l = (d0[-1] & d1p[-1] & d2[-1]) >> 63;
                        const uint64_t r = (d0[1] & d1p[1] & d2[1]) << 63;
                        e = c & ((c << 1) | l) & ((c >> 1) | r);
                    } else if (halo == 1) {
                        e = D[(y & (SRING - 1)) * pw + 1 + k];
                    } else {
                        e = T[(y & (SRING - 1)) * pw + 1 + k];
                    }
                    E[mw_r[j] * ww + k] = e;
                }
                const uint32_t poff = (y < o1 && !(A.dbg & 4)) ? (plane_base + (uint32_t)(y + 1) * (uint32_t)A.prow + 1u + (uint32_t)k) * 8u : OOB;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(su32x2, e), r_plane, poff, 0, 0);
            }
            __syncthreads();
            // ---------------- row masks for the contour stage + the 0/255 byte image
            {
                uint32_t m = 0;
                if (ww <= 32 && lane < o1 - o)
                    for (int k = 0; k < ww; k++) m |= (uint32_t)(E[lane * ww + k] != 0) << k;
                const uint32_t moff = (ww <= 32 && lane < o1 - o && !(A.dbg & 4)) ? (uint32_t)(cur.f * h + o + lane) * 4u : OOB;
                __builtin_amdgcn_raw_buffer_store_b32(m, r_mask, moff, 0, 0);
            }
            if (IMG) {
#pragma unroll
                for (int u = 0; u < NU; u++) {
                    const int y = o + it_r[u];
                    const bool ok = it_r[u] >= 0 && y < o1;
                    uint32_t m = 0;
                    if (ok) m = reinterpret_cast<const uint16_t*>(E + it_r[u] * ww)[it_q[u]];
                    const su32x4 px = {s_expand4(m), s_expand4(m >> 4), s_expand4(m >> 8), s_expand4(m >> 12)};
                    const uint32_t boff = (ok && !(A.dbg & 2)) ? bin_base + (uint32_t)y * (uint32_t)A.w + (uint32_t)it_q[u] * 16u : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(px, r_bin, boff, 0, 2 /* nt: written once, read by nobody here */);
                }
            }
            o = o1;
            __syncthreads(); // E and the ring slots are free for the next step
        }
        cur = nxt;
        have = have_nxt;
        have_nxt = have ? claim(nxt) : false;
    }
    // the last wave to leave zeroes the queue for the next launch (see k_binary.hip)
    {
        int left = 0;
        if (lane == 0) left = atomicAdd(&A.ctr[8], 1);
        left = __builtin_amdgcn_readfirstlane(left);
        if (left == (int)gridDim.x - 1 && lane < 9) atomicExch(&A.ctr[lane], 0);
    }
}

// returns hipErrorNotSupported when the geometry is not one this form takes (the caller then uses k_binary)
template <int CA, int CB>
static hipError_t launch_stream_t(const Geom& g, const Bufs& b, int lower_bound, int morph, bool image, int waves_per_cu, hipStream_t s)
{
    const int wq = g.ww * 4;
    int SR = (SU * 64) / wq;
    if (SR > 4) SR = 4;
    if (SR < 1 || g.ww > 64) return hipErrorNotSupported;
    // 32-bit buffer offsets: every extent must stay below 4 GiB
    const int64_t lim = 0xFFFFFF00ll;
    if ((int64_t)g.n_frames * g.frame_pitch >= lim || (int64_t)g.n_frames * g.plane_pitch * 8 >= lim || (int64_t)g.n_frames * g.w * g.h >= lim)
        return hipErrorNotSupported;
    StreamArgs A;
    A.frames = b.frames;
    A.frame_pitch = g.frame_pitch;
    A.stride = g.stride;
    A.w = g.w;
    A.h = g.h;
    A.ww = g.ww;
    int lb = lower_bound, all_pass = 0;
    if (lb <= 0) { all_pass = 1; lb = 1; }
    if (lb > 256) lb = 256;
    A.lb = lb;
    A.all_pass = all_pass;
    A.morph = morph;
    A.binary = b.binary;
    A.bits = b.bits;
    A.prow = g.prow;
    A.plane_pitch = g.plane_pitch;
    A.rowmask = b.rowmask;
    A.ctr = b.strip_ctr;
    A.SR = SR;
    A.n_frames = g.n_frames;
    static const int dbg_env = getenv("RMCV_KS_DBG") ? atoi(getenv("RMCV_KS_DBG")) : 0;
    A.dbg = dbg_env;
    static const int br_env = getenv("RMCV_K1_BAND") ? atoi(getenv("RMCV_K1_BAND")) : 0; // dev knob for A/B runs
    A.BR = br_env > 0 ? br_env : 64;
    A.bands_per_frame = (g.h + A.BR - 1) / A.BR;
    A.n_units = g.n_frames * A.bands_per_frame;
    static const int wpc_env = getenv("RMCV_K1_WPC") ? atoi(getenv("RMCV_K1_WPC")) : 0; // dev knob for A/B runs
    const int wpc = wpc_env > 0 ? wpc_env : waves_per_cu;
    int grid = (g.n_cu > 0 ? g.n_cu : 256) * wpc;
    if (grid > ((A.n_units + 7) & ~7)) grid = (A.n_units + 7) & ~7;
    grid = (grid + 7) & ~7;
    const size_t lds = (size_t)(2 * SRING * (g.ww + 2) + SR * g.ww) * sizeof(uint64_t);
    const int nu = (SR * wq + 63) / 64, mi = (SR * g.ww + 63) / 64; // nu 1..5, mi 1..2
#define RMCV_KS(NU_, MI_)                                                                                          \
    if (nu == NU_ && mi == MI_)                                                                                    \
        return image ? launch(k_stream<CA, CB, NU_, MI_, true>, dim3(grid), dim3(64), lds, s, A)                   \
                     : launch(k_stream<CA, CB, NU_, MI_, false>, dim3(grid), dim3(64), lds, s, A);
    RMCV_KS(5, 2) RMCV_KS(5, 1) RMCV_KS(4, 2) RMCV_KS(4, 1) RMCV_KS(3, 1) RMCV_KS(2, 1) RMCV_KS(1, 1)
#undef RMCV_KS
    return hipErrorNotSupported;
}

hipError_t launch_stream(const Geom& g, const Bufs& b, int camp, int lower_bound, int morph, bool image, int waves_per_cu, hipStream_t s)
{
    // imgproc.cpp:56-65: GUIDELIGHT G-R; BLUE B-R; everything else (RED, NEUTRAL) R-B.  BGR byte order.
    if (camp == RMCV_CAMP_GUIDELIGHT) return launch_stream_t<1, 2>(g, b, lower_bound, morph, image, waves_per_cu, s);
    if (camp == RMCV_CAMP_BLUE) return launch_stream_t<0, 2>(g, b, lower_bound, morph, image, waves_per_cu, s);
    return launch_stream_t<2, 0>(g, b, lower_bound, morph, image, waves_per_cu, s);
}

} // namespace rmcv
