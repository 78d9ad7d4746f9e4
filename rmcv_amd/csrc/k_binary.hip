// k_binary.hip -- K1: the pixel-streaming part of rm::extract_color
// (/root/reference/src/imgproc.cpp:52-69): split + saturating channel subtract + inRange + 3x3
// MORPH_CLOSE, fused into ONE pass over the BGR frame.
//
// HBM-bound (no MFMA: there is no contraction here).  Algorithmic traffic: 3 B/px read +
// 1 B/px written (the reference returns `binary`, imgproc.cpp:74) = 4 B/px; the bit plane the
// contour stage consumes adds 1/8 B/px.  The reference's CPU path makes ~8 full-frame passes
// (split x3, subtract, inRange, dilate, erode, findContours' copy); here every intermediate lives
// in registers or LDS:
//
//   phase 1  a wave loads a 256-px block of four rows with four coalesced dwordx3 (lane i: pixels 4i..4i+3 of each
//            row), thresholds its 16 px to a 16-bit mask, a lane quad transposes its 4x4 nibbles (2 DPP exchanges) and
//            each lane writes 16 bits of one row of the strip's bit plane T in LDS
//   phase 2  dilate on the bit plane  D = hdil(T[y-1] | T[y] | T[y+1])          (64 px / lane-op)
//   phase 3  erode                    E = hero(D[y-1] & D[y] & D[y+1])
//   phase 4  E -> 0/255 bytes, 16 px per lane, one coalesced dwordx4 store; E word -> bit plane
//
// A workgroup owns a strip of SR rows of one frame (+2 halo rows per side for CLOSE); strips are
// mapped so that consecutive strips of a frame land on the same XCD (blockIdx % 8 groups), where the
// halo rows they share are L2 hits.  Border semantics (OpenCV morphologyDefaultBorderValue): samples
// outside the image never win, i.e. they read 0 for the dilate and 1 for the erode.
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <type_traits>

#include "rmcv_internal.h"

namespace rmcv {

#ifndef RMCV_SR
#define RMCV_SR 32
#endif
#ifndef RMCV_K1_UNROLL
#define RMCV_K1_UNROLL 4
#endif
#ifndef RMCV_K1_STAUX
#define RMCV_K1_STAUX 2 // cache-policy bits of the byte-image stores (2 = nt)
#endif
#ifndef RMCV_K1_PLAIN_PLAUX
#define RMCV_K1_PLAIN_PLAUX 0 // cache-policy bits of the bit-plane stores (plain: the sparse kernel of the same batch finds the words in L2)
#endif
#ifndef RMCV_K1_HALOAUX
#define RMCV_K1_HALOAUX 0 // cache-policy bits of the loads of the row quads a strip shares with its neighbours (0 = cacheable: the neighbour finds them in L2)
#endif
#ifndef RMCV_K1_LDAUX
#define RMCV_K1_LDAUX 2 // cache-policy bits of the frame loads that no other workgroup shares (2 = nt)
#endif
static constexpr int SR = RMCV_SR; // strip rows per workgroup
static_assert(SR == STRIP_ROWS, "the sparse kernel's frame queues assume k_binary's strip height (rmcv_internal.h)");

__device__ __forceinline__ uint32_t expand4(uint32_t nib)
{ // 4 mask bits -> 4 bytes of 0x00/0xFF
    return (((nib & 0xFu) * 0x00204081u) & 0x01010101u) * 0xFFu;
}

// 16 pixels (48 bytes in 12 dwords) -> 16-bit mask of (a - b >= lb), two pixels per packed-16 operation:
//   v_perm_b32 gathers byte a of pixels p and p+8 (24 bytes = 6 dwords apart) into the two halves of a dword (same for b),
//   t = (A + (0x8000 - lb)) - B per half: bit 15 of a half is set  <=>  a - b - lb >= 0   (|a - b - lb| < 2^15),
// the flags are collected by shifting the accumulator (bit 15 -> pixels 0..7 end in bits 8..15, bit 31 -> pixels 8..15 in bits
// 24..31) and one last v_perm picks the two bytes: 6 operations per pair of pixels.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
template <int CA, int CB>
__device__ __forceinline__ uint32_t thresh16(const uint32_t d[12], int lb)
{
    const uint32_t kk = (uint32_t)(0x8000 - lb) & 0xFFFFu;
    const uint32_t K = kk | (kk << 16);
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int ia = 3 * j + CA, ib = 3 * j + CB; // byte offsets of pixel j; pixel j+8 is 24 bytes = 6 dwords further
        // v_perm_b32(S0, S1, sel): selector 0..3 = bytes of S1, 4..7 = bytes of S0, 0x0c = zero
        const uint32_t sa = (uint32_t)(ia & 3) | (0x0cu << 8) | ((uint32_t)((ia & 3) + 4) << 16) | (0x0cu << 24);
        const uint32_t sb = (uint32_t)(ib & 3) | (0x0cu << 8) | ((uint32_t)((ib & 3) + 4) << 16) | (0x0cu << 24);
        const uint32_t A = __builtin_amdgcn_perm(d[(ia >> 2) + 6], d[ia >> 2], sa);
        const uint32_t B = __builtin_amdgcn_perm(d[(ib >> 2) + 6], d[ib >> 2], sb);
        u16x2 t = __builtin_bit_cast(u16x2, A) + __builtin_bit_cast(u16x2, K);
        t = t - __builtin_bit_cast(u16x2, B);
        acc = (acc >> 1) | (__builtin_bit_cast(uint32_t, t) & 0x80008000u);
    }
    return __builtin_amdgcn_perm(0u, acc, 0x0c0c0301u); // byte 1 (pixels 0..7), byte 3 (pixels 8..15)
}

// n / d for n < 2^16 with a precomputed reciprocal r = ceil(2^32 / d) (exact in that range); d == 1 gives r == 0
__device__ __forceinline__ int div_r(int n, uint32_t r) { return r ? (int)__umulhi((uint32_t)n, r) : n; }

// lb is pre-clamped on the host to [1, 256]: lb <= 0 means "everything passes" (lb = -1 flag).
// FAST (w a multiple of 64, 16-byte aligned rows, every extent below 4 GiB): the vector-memory instructions are UNCONDITIONAL
// raw-buffer operations.  A lane (or item) that has nothing to move uses an offset beyond the buffer's extent: the hardware
// answers such a load with zeros -- a row outside the image thresholds to 0 by itself -- and drops such a store.  Round 1 had
// ordinary loads behind per-lane predicates: per 16-pixel item that was ~70 instructions of EXEC save/restore, branches and
// register zeroing around the 60 that threshold (profiles/r02a_k_binary_ablations.txt: 0.14-0.18 ms of the 0.28 with the
// loads compiled out).
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3v __attribute__((ext_vector_type(3)));
static constexpr uint32_t OOB = 0xFFFFFF00u; // voffset of a lane that moves nothing (extents are checked below 4 GiB - 256)
static constexpr int RSRC3 = 0x00020000;     // raw buffer descriptor word 3, gfx9 family

// Register budget: 6 waves per SIMD = at most 80 VGPRs.  Two launches of consecutive batches overlap (2 workgroups per CU each = 4
// waves per SIMD) next to one wave of the 4-wavefront sparse kernel (168 VGPRs): 4 x 80 + 168 <= 512.  At 88 the sparse kernel
// would no longer fit beside them and the batches in flight would take turns instead of sharing the CUs.
template <int CA, int CB, int FAST /* 0: byte-wise loader, 1: row-quad items, 2: linear items (rows contiguous in memory) */>
#ifndef RMCV_K1_MINBLOCKS
#define RMCV_K1_MINBLOCKS 6
#endif
__global__ __launch_bounds__(256, RMCV_K1_MINBLOCKS) void k_binary(const uint8_t* __restrict__ frames, int64_t frame_pitch, int stride, int n_frames,
                                                 int w, int h, int ww, int lb, int all_pass, int morph,
                                                 uint8_t* __restrict__ binary, uint64_t* __restrict__ bits, int prow,
                                                 int64_t plane_pitch, int strips, int n_blocks, uint32_t* __restrict__ rowmask,
                                                 int* __restrict__ strip_ctr, int taper_head, int taper_tail,
                                                 int halo_nt /* RMCV_OPT_PIXEL_HALO_NT */)
{
    extern __shared__ uint64_t smem[];
#ifdef RMCV_PROFILE_HANDOVER
    if (blockIdx.x == 0 && threadIdx.x == 0) printf("[kb start] %lld\n", (long long)wall_clock64());
#endif
#ifdef RMCV_K1_PRIO
    __builtin_amdgcn_s_setprio(RMCV_K1_PRIO); // dev knob (A/B of issue priorities against the sparse kernel's)
#endif
    const int halo = morph; // NONE 0, DILATE 1, CLOSE 2
    uint64_t* T = smem;
    uint64_t* D = smem + (size_t)(SR + 4) * ww;

    // Persistent workgroups: the grid is sized to a fixed number of workgroups per CU (leaving wave slots for the
    // sparse kernels of the previous batch that run on another stream) and every workgroup loops over strips.
    // XCD-aware order: workgroups b, b+8, b+16.. share an XCD; XCD x owns the contiguous strip range
    // [x*n/8, (x+1)*n/8) and its workgroups sweep it together, so neighbouring strips (which share halo rows)
    // are in flight on the same L2 at the same time.
    const int tid = threadIdx.x;
    const int wq = ww * 4; // 16-pixel groups per row
    const uint32_t r_wq = (uint32_t)((0x100000000ull + wq - 1) / wq), r_ww = (uint32_t)((0x100000000ull + ww - 1) / ww);
    const int xcd = blockIdx.x & 7; // gridDim.x is a multiple of 8
    const int per_xcd = (n_blocks + 7) >> 3;
    __shared__ int s_next;
    // A thread's items of a strip are tid, tid + 256, ...: their (row, group) pairs are stepped, not divided -- integer multiplies
    // (v_mul_lo/hi_u32) issue at a fraction of the rate of an add, and the FAST path is as much issue-bound as memory-bound
    const int q_first = tid - (int)div_r(tid, r_wq) * wq, r_first = div_r(tid, r_wq); // item tid = (r_first, q_first)
    const int q_step = 256 - (int)div_r(256, r_wq) * wq, r_step = div_r(256, r_wq);     // item + 256 = (r + r_step, q + q_step) or (r + r_step + 1, q + q_step - wq)
    const int k_first = tid - (int)div_r(tid, r_ww) * ww, s_first = div_r(tid, r_ww);   // the same for the strip's 64-pixel words
    const int k_step = 256 - (int)div_r(256, r_ww) * ww, s_step = div_r(256, r_ww);
    // 8 mask bits -> 8 bytes of 0/255: a 256-entry table in LDS instead of two multiplies per nibble (phase 4)
    __shared__ uint64_t s_lut[256];
    __shared__ uint16_t s_spare[256]; // where a lane without a place in the plane writes (no write sits behind a branch)
    if (FAST) s_lut[tid] = (uint64_t)expand4(tid) | ((uint64_t)expand4(tid >> 4) << 32);
    int ticket = 0;
    if (tid == 0) ticket = atomicAdd(&strip_ctr[xcd * CTR_STRIDE], 1);
    for (;;) {
    // dynamic strip queue per XCD: a workgroup takes the next strip of its XCD's range when it is done with the previous
    // one, so CUs that also host kernels of another stream simply take fewer strips (a static split made them the tail)
    __syncthreads(); // also: the LDS planes of the previous strip are free
    // Every launch finds the heads at 0: the workgroup that leaves last zeroes them (below), so there is no memset per step
    // and no host-side mirror of device state that a failed or foreign launch could put out of step.
    // Pieces: the first taper_head and the last taper_tail strips of an XCD's range are handed out as four 8-row pieces each.
    // Used by launches with fewer strips than half the CUs (one camera frame: the per-frame drop-in chain), which hand out EVERY
    // strip that way; as a ramp / tail shortener of full batches it measured nothing (round 3) and is not offered any more.
    const int n_mid = per_xcd - taper_head - taper_tail;
    const int n_queue = 4 * taper_head + n_mid + 4 * taper_tail;
    // The ticket for THIS strip was drawn while the previous strip was being processed (`ticket`, thread 0); the next one is
    // drawn now and not looked at until the next iteration: a draw is a device-scope atomic -- a round trip of microseconds to the
    // memory side, which used to sit on every strip's critical path between two barriers.  (The queue heads are CTR_STRIDE ints
    // apart: eight heads in one cache line served every draw of every XCD one after the other.)
    if (tid == 0) {
        s_next = ticket;
        ticket = atomicAdd(&strip_ctr[xcd * CTR_STRIDE], 1);
    }
    __syncthreads();
    const int j = s_next;
    if ((uint32_t)j >= (uint32_t)n_queue) break;
    int s_local, piece = 0, sr = SR;
    if (j < 4 * taper_head) { s_local = j >> 2; piece = j & 3; sr = SR / 4; }
    else if (j < 4 * taper_head + n_mid) { s_local = taper_head + (j - 4 * taper_head); }
    else { const int jj = j - 4 * taper_head - n_mid; s_local = taper_head + n_mid + (jj >> 2); piece = jj & 3; sr = SR / 4; }
    const int L = xcd * per_xcd + s_local;
    if (L >= n_blocks) continue; // tail of the last XCD's range: draw on, so that every head advances alike
    const int f = L / strips, strip = L - f * strips;
    const int y0 = strip * SR + piece * (SR / 4);
    if (y0 >= h) continue; // a piece of the frame's last strip that lies below the image (h % SR <= 24): nothing to load or store
    const int srh = sr + 2 * halo;
    const uint8_t* frame = frames + (int64_t)f * frame_pitch;

    // ---------------- phase 1: load + threshold -> T
    if (FAST) {
        // U items per wave per iteration: all 4*U loads are issued before the first threshold (memory-level
        // parallelism per wave); every 16-bit mask goes straight to its place in the LDS plane (ds_write_b16)
        constexpr int U = RMCV_K1_UNROLL;
        const int items = srh * wq;
        const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint8_t*>(frames), 0, (int)((int64_t)(n_frames - 1) * frame_pitch + (int64_t)(h - 1) * stride + 3 * w), RSRC3);
        const uint32_t fbase = (uint32_t)((int64_t)f * frame_pitch);
        // Wave-coalesced loads: an item is a 256-pixel block of FOUR rows; lane i loads pixels 4i..4i+3 of each row with one
        // dwordx3, so the wave reads 768 contiguous bytes = six whole cache lines per instruction and every line is touched by
        // exactly one instruction -- which is what lets the loads carry the non-temporal hint (round 2 measured it: per-lane
        // 48-byte loads touch a line with three instructions and lose 15-30 % with the hint; these gain 13 % with it,
        // profiles/r02d_k_binary_coalesced_x3.txt).  The 12 dwords of a lane are 16 whole pixels (thresh16): bit 4k+t = row k,
        // pixel 4i+t; the four lanes of a quad then transpose their 4x4 nibbles with two DPP exchanges, after which lane l of the
        // quad holds the 16 mask bits of row l and writes them with one ds_write_b16.  The item's (row quad, block) is
        // wave-uniform: its address arithmetic runs on the scalar unit.
        const int lane = tid & 63;
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int nb = (w + 255) >> 8, nq = (srh + 3) >> 2, n_it = nq * nb;
        const uint32_t r_nb = (uint32_t)((0x100000000ull + nb - 1) / nb);
        const uint32_t lane_off = (uint32_t)lane * 12u;
        const uint32_t lds_lane = (uint32_t)__umul24(lane & 3, ww) * 8u + (uint32_t)(lane >> 2) * 2u;
        const uint32_t M1 = (lane & 1) ? 0xF0F0u : 0x0F0Fu, S1 = (lane & 1) ? 12u : 4u;
        const uint32_t P2 = (lane & 2) ? 0x0c0c0105u : 0x0c0c0400u;
        constexpr uint32_t OOB_S = 0xFFFFFC00u; // scalar part of an offset that moves nothing (+ 63 * 12 stays out of extent)
        const int rr_lo = max(0, halo - y0), rr_hi = min(srh, h - y0 + halo); // the strip's rows that lie inside the image
        const uint32_t rr_span = (uint32_t)(rr_hi - rr_lo);
        const uint32_t strip_base = fbase + (uint32_t)(y0 - halo) * (uint32_t)stride; // wraps for the rows above the image: never used
        const int ragged = (w & 255) ? 1 : 0;
        const bool plain = rr_lo == 0 && rr_hi == srh && (srh & 3) == 0; // the strip's rows need no validity selects at all
        uint32_t dk1 = (uint32_t)stride, dk2 = 2u * (uint32_t)stride, dk3 = 3u * (uint32_t)stride;
        asm volatile("" : "+s"(dk1), "+s"(dk2), "+s"(dk3)); // opaque: otherwise every row's offset is re-derived with its own multiply
        if (all_pass) {
            int rq = r_first, q = q_first;
            for (int it = tid; it < items; it += 256) {
                const int y = y0 - halo + rq;
                reinterpret_cast<uint16_t*>(T + __umul24(rq, ww))[q] = (y >= 0 && y < h) ? 0xFFFFu : 0u;
                q += q_step;
                rq += r_step;
                if (q >= wq) { q -= wq; rq++; }
            }
        } else if (FAST == 2) {
            // LINEAR items (stride == 3 w: the strip's rows are ONE contiguous run in memory, as they are in the LDS plane since w % 64
            // == 0): the strip is a sequence of 256-pixel blocks -- block j = pixels [256 j, 256 j + 256) of that run = 768 contiguous
            // bytes = words [4 j, 4 j + 4) of T -- and an item is FOUR CONSECUTIVE blocks: the wave's four loads of an item read 3 KB
            // in one piece (row-quad items: four 768-byte pieces a row apart), and a row whose width is no multiple of 256 (1920 =
            // 7.5 blocks) wastes nothing: 270 blocks = 68 items per strip instead of 9 x 8 = 72 with every eighth half empty.
            // Lane i loads pixels 4 i .. 4 i + 3 of each block; after the quad transpose lane l of a quad holds 16 pixels of block l.
            const uint32_t px_total = (uint32_t)__umul24(srh, w);      // a multiple of 64; of 256 for the common sizes, not always of 1024
            const int n_blk = (int)((px_total + 255u) >> 8);
            const int n_itl = (n_blk + 3) >> 2;
            const uint32_t px_lo = (uint32_t)__umul24(rr_lo, w), px_span = (uint32_t)__umul24(rr_hi - rr_lo, w); // the run's pixels inside the image
            const uint32_t q_lo = (uint32_t)(4 * w), q_hi = (uint32_t)__umul24(srh - 4, w); // pixels of the first / last four rows: shared with the neighbours
            const bool whole = rr_lo == 0 && rr_hi == srh;             // every row of the strip is inside the image
            uint16_t* const T16 = reinterpret_cast<uint16_t*>(T);
            const uint32_t lds_l = (uint32_t)(lane & 3) * 16u + (uint32_t)(lane >> 2); // halfword of this lane's 16 pixels inside an item's 64 halfwords
            for (int it0 = wv; it0 < n_itl; it0 += 4 * U) {
                auto batch = [&](auto chk) {
                    constexpr bool CHK = decltype(chk)::value;
                    u32x3v v[U][4];
                    int itv[U];
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        const int it_ = it0 + 4 * u;
                        const int it = (L & 1) ? it_ : n_itl - 1 - it_; // sweep direction: neighbouring strips meet at their shared rows
                        itv[u] = (CHK && it_ >= n_itl) ? -1 : it;
                        const uint32_t blk0 = (uint32_t)it * 4u;
#ifdef RMCV_K1_NOLOAD
                        const uint32_t base = OOB_S - 2304u;
#else
                        const uint32_t base = strip_base + blk0 * 768u;
#endif
                        uint32_t vo[4];
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            vo[k] = base + (uint32_t)k * 768u + lane_off;
                            if (CHK) { // this lane's four pixels of block k: inside the strip's blocks and inside the image?
                                const uint32_t pix = (blk0 + (uint32_t)k) * 256u + (uint32_t)lane * 4u;
                                if (it_ >= n_itl || (int)(blk0 + k) >= n_blk || pix - px_lo >= px_span) vo[k] = OOB_S + lane_off;
                            }
                        }
                        const uint32_t p0 = blk0 * 256u;
                        if (halo && !halo_nt && (p0 < q_lo || p0 + 1024u > q_hi)) {
#pragma unroll
                            for (int k = 0; k < 4; k++) v[u][k] = __builtin_amdgcn_raw_buffer_load_b96(r_in, vo[k], 0, RMCV_K1_HALOAUX);
                        } else {
#pragma unroll
                            for (int k = 0; k < 4; k++) v[u][k] = __builtin_amdgcn_raw_buffer_load_b96(r_in, vo[k], 0, RMCV_K1_LDAUX);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        const uint32_t d[12] = {v[u][0].x, v[u][0].y, v[u][0].z, v[u][1].x, v[u][1].y, v[u][1].z,
                                                v[u][2].x, v[u][2].y, v[u][2].z, v[u][3].x, v[u][3].y, v[u][3].z};
                        const uint32_t m = thresh16<CA, CB>(d, lb);
                        const uint32_t p1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)m, 0xB1, 0xF, 0xF, true);
                        const uint32_t t1 = (m & M1) | (((p1 << 8) >> S1) & ~M1);
                        const uint32_t p2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)t1, 0x4E, 0xF, 0xF, true);
                        const uint32_t t2 = __builtin_amdgcn_perm(p2, t1, P2);
                        uint16_t* dst = T16 + (uint32_t)itv[u] * 64u + lds_l;
                        // an item beyond the strip's, or 16 pixels beyond the strip's last (the ragged end of its last item)
                        if (CHK && (itv[u] < 0 || (uint32_t)itv[u] * 1024u + (uint32_t)(lane & 3) * 256u + (uint32_t)(lane >> 2) * 16u >= px_total)) dst = s_spare + tid;
                        *dst = (uint16_t)t2;
                    }
                };
                // unchecked: every row inside the image, a full batch, and not the strip's last item if that one is ragged
                const bool has_last = (L & 1) ? (it0 + 4 * (U - 1) >= n_itl - 1) : (it0 == 0);
                if (whole && it0 + 4 * (U - 1) < n_itl && ((px_total & 1023u) == 0 || !has_last)) batch(std::false_type{});
                else batch(std::true_type{});
            }
        } else
        for (int it0 = wv; it0 < n_it; it0 += 4 * U) {
            // One batch = U items of the wave: all 4 * U loads are issued, then thresholded.  CHK = false is the common case (a strip
            // with every row inside the image, whole row quads, a full batch): no validity selects.
            auto batch = [&](auto chk) {
                constexpr bool CHK = decltype(chk)::value;
                u32x3v v[U][4];
                int info[U]; // LDS byte offset of the item's (row quad, block) | ragged-block flag; -1: beyond the strip's items
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int it = it0 + 4 * u; // wave-uniform: everything up to the four vector adds runs on the scalar unit
                    const int jq0 = div_r(it, r_nb), b = it - jq0 * nb;
                    const int jq = (L & 1) ? jq0 : nq - 1 - jq0; // sweep direction, see below
                    const int rr0 = 4 * jq;
                    info[u] = (int)(__umul24(rr0, ww) * 8u + (uint32_t)b * 32u) | (b == nb - 1 ? ragged : 0); // bit 0: ragged block
                    if (CHK && it >= n_it) info[u] = -1;
#ifdef RMCV_K1_NOLOAD
                    const uint32_t base = OOB_S - dk3; // ablation build: nothing is read
#else
                    const uint32_t base = strip_base + (uint32_t)rr0 * (uint32_t)stride + (uint32_t)b * 768u;
#endif
                    const uint32_t rowk[4] = {base, base + dk1, base + dk2, base + dk3};
                    uint32_t vo[4];
                    // rows rr_lo <= rr < rr_hi of the strip are inside the image; the others (and a whole item beyond the
                    // strip's) are "loaded" from beyond the extent: zeros, no traffic
                    const uint32_t t0 = (uint32_t)(rr0 - rr_lo), span = it < n_it ? rr_span : 0u;
#pragma unroll
                    for (int k = 0; k < 4; k++) vo[k] = (!CHK || t0 + (uint32_t)k < span ? rowk[k] : OOB_S) + lane_off;
                    // the first and the last row quad hold the rows this strip shares with its neighbours: those stay
                    // cacheable (the neighbour finds them in L2), everything else is read once and says so
                    // (RMCV_OPT_PIXEL_HALO_NT, a measurement knob: on some boxes the pixel kernels alone run at 0.2537 ms per launch
                    // with cacheable shared rows and at 0.2446 with the hint for them too, on the others the hint costs 1-5 %; the
                    // whole path hardly notices at 1280 px and loses 6 % at 1920 px: DESIGN.md 6g)
                    if (halo && !halo_nt && (jq == 0 || jq == nq - 1)) {
#pragma unroll
                        for (int k = 0; k < 4; k++) v[u][k] = __builtin_amdgcn_raw_buffer_load_b96(r_in, vo[k], 0, RMCV_K1_HALOAUX);
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; k++) v[u][k] = __builtin_amdgcn_raw_buffer_load_b96(r_in, vo[k], 0, RMCV_K1_LDAUX);
                    }
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const uint32_t d[12] = {v[u][0].x, v[u][0].y, v[u][0].z, v[u][1].x, v[u][1].y, v[u][1].z,
                                            v[u][2].x, v[u][2].y, v[u][2].z, v[u][3].x, v[u][3].y, v[u][3].z};
                    uint32_t m = thresh16<CA, CB>(d, lb);
                    const bool last_ragged = (info[u] & 1) != 0; // wave-uniform: the row's last block when w % 256 != 0
                    // pixels beyond the row's end: the lane has read the next row's bytes
                    if (last_ragged && lane * 4 >= w - ((nb - 1) << 8)) m = 0;
                    // 4x4 nibble transpose within the quad: exchange with lane^1 (nibbles), then with lane^2 (bytes)
                    const uint32_t p1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)m, 0xB1, 0xF, 0xF, true);
                    const uint32_t t1 = (m & M1) | (((p1 << 8) >> S1) & ~M1);
                    const uint32_t p2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)t1, 0x4E, 0xF, 0xF, true);
                    const uint32_t t2 = __builtin_amdgcn_perm(p2, t1, P2);
                    uint16_t* dst = reinterpret_cast<uint16_t*>(reinterpret_cast<uint8_t*>(T) + ((uint32_t)info[u] & ~1u) + lds_lane);
                    if (CHK && info[u] < 0) dst = s_spare + tid; // an item beyond the strip's
                    // the quad's 16 pixels of the last block may lie beyond the row: those go to a spare word
                    if (last_ragged && (lane >> 2) * 16 >= w - ((nb - 1) << 8)) dst = s_spare + tid;
                    *dst = (uint16_t)t2;
                }
            };
            if (plain && it0 + 4 * (U - 1) < n_it) batch(std::false_type{});
            else batch(std::true_type{});
        }
    } else {
        const int items = srh * wq;
        int rr = tid / wq, q = tid - rr * wq;
        const int dr = 256 / wq, dq = 256 - dr * wq;
        for (int it = tid; it < items; it += 256) {
            const int y = y0 - halo + rr;
            uint32_t m = 0;
            if (y >= 0 && y < h) {
                const uint8_t* row = frame + (int64_t)y * stride;
                for (int p = 0; p < 16; p++) {
                    int x = q * 16 + p;
                    if (x < w) {
                        int a = row[3 * x + CA], bb = row[3 * x + CB];
                        m |= (uint32_t)(all_pass || (a - bb >= lb)) << p;
                    }
                }
            }
            // merge the 4 lanes of a word (lanes are word-aligned: wq % 4 == 0, 256 % 4 == 0)
            uint32_t v = m << (16 * (q & 1));
            v |= __shfl_xor(v, 1);
            uint32_t o = __shfl_xor(v, 2);
            if ((q & 3) == 0) T[rr * ww + (q >> 2)] = ((uint64_t)o << 32) | v;
            rr += dr;
            q += dq;
            if (q >= wq) { q -= wq; rr++; }
        }
    }
    __syncthreads();

    const uint64_t last_valid = (w & 63) ? ((1ull << (w & 63)) - 1) : ~0ull; // valid bits of the last word
    uint64_t* R = T; // plane holding the result rows, result row s at R[(s + halo) * ww + k]

    if (morph != RMCV_MORPH_NONE) {
        // ---------------- phase 2: dilate -> D (rows 1 .. srh-2)
        const int items = (srh - 2) * ww;
        int r_ = s_first, k = k_first; // (row, word) of item it, stepped (see the kernel's prologue)
        for (int it = tid; it < items; it += 256, k += k_step, r_ += s_step) {
            if (k >= ww) { k -= ww; r_++; }
            const int rr = 1 + r_;
            const int y = y0 - halo + rr;
            const int row = FAST ? (int)__umul24(rr, ww) : rr * ww;
            uint64_t d;
            if (y < 0 || y >= h) {
                d = ~0ull; // outside the image: never wins the erode
            } else {
                const uint64_t* t0 = T + row - ww;
                const uint64_t* t1 = T + row;
                const uint64_t* t2 = T + row + ww;
                uint64_t c = t0[k] | t1[k] | t2[k];
                uint64_t l = (k > 0) ? (t0[k - 1] | t1[k - 1] | t2[k - 1]) >> 63 : 0;
                uint64_t r = (k < ww - 1) ? (t0[k + 1] | t1[k + 1] | t2[k + 1]) & 1 : 0;
                d = c | (c << 1) | l | (c >> 1) | (r << 63);
                if (k == ww - 1) {
                    d &= last_valid;
                    if (morph == RMCV_MORPH_CLOSE) d |= ~last_valid; // columns >= w never win the erode
                }
            }
            D[row + k] = d;
        }
        __syncthreads();
        R = D;
        if (morph == RMCV_MORPH_CLOSE) {
            // ---------------- phase 3: erode -> T (rows 2 .. srh-3 = the strip)
            const int items3 = sr * ww;
            int r3 = s_first, k = k_first;
            for (int it = tid; it < items3; it += 256, k += k_step, r3 += s_step) {
                if (k >= ww) { k -= ww; r3++; }
                const int rr = 2 + r3;
                const int row = FAST ? (int)__umul24(rr, ww) : rr * ww;
                const uint64_t* d0 = D + row - ww;
                const uint64_t* d1 = D + row;
                const uint64_t* d2 = D + row + ww;
                uint64_t c = d0[k] & d1[k] & d2[k];
                uint64_t l = (k > 0) ? (d0[k - 1] & d1[k - 1] & d2[k - 1]) >> 63 : 1;
                uint64_t r = (k < ww - 1) ? (d0[k + 1] & d1[k + 1] & d2[k + 1]) & 1 : 1;
                uint64_t e = c & ((c << 1) | l) & ((c >> 1) | (r << 63));
                if (k == ww - 1) e &= last_valid;
                T[row + k] = e;
            }
            __syncthreads();
            R = T;
        }
    }

    // ---------------- row masks for the contour stage: bit k = word k of the row is non-zero
    if (ww <= 32 && tid < sr && y0 + tid < h) {
        uint32_t m = 0;
        for (int k = 0; k < ww; k++) m |= (uint32_t)(R[(tid + halo) * ww + k] != 0) << k;
        rowmask[(int64_t)f * h + y0 + tid] = m;
    }
    // ---------------- phase 4: expand to bytes + bit plane
    if (FAST) {
        const __amdgpu_buffer_rsrc_t r_bin = __builtin_amdgcn_make_buffer_rsrc(binary, 0, binary ? (int)((int64_t)n_frames * w * h) : 0, RSRC3);
        const __amdgpu_buffer_rsrc_t r_plane = __builtin_amdgcn_make_buffer_rsrc(bits, 0, (int)((int64_t)n_frames * plane_pitch * 8), RSRC3);
        const uint32_t plane_base = (uint32_t)((int64_t)f * plane_pitch);
        // ... and the two pad words behind every row's last word (always zero), so that the plane's cache lines are written whole (see
        // k_binary_ws.inc: with a 16-byte hole in every line the plane costs 3-15 % of the kernel once it falls out of the Infinity Cache)
        if (tid < sr && y0 + tid < h) {
            const u32x2v z = {0u, 0u};
            const uint32_t pp = (plane_base + __umul24(y0 + tid + 1, prow) + 1u + (uint32_t)ww) * 8u;
            __builtin_amdgcn_raw_buffer_store_b64(z, r_plane, pp, 0, RMCV_K1_PLAIN_PLAUX);
            __builtin_amdgcn_raw_buffer_store_b64(z, r_plane, pp + 8u, 0, RMCV_K1_PLAIN_PLAUX);
        }
        { // the strip's words -> the frame's bit plane (8 contiguous bytes per lane)
            const int nw = sr * ww;
            int s_ = s_first, k = k_first;
            for (int it = tid; it - (tid & 63) < nw; it += 256) {
                const int y = y0 + s_;
                const bool ok = it < nw && y < h;
                uint64_t word = 0;
                if (ok) word = R[__umul24(s_ + halo, ww) + k];
                const uint32_t po = ok ? (plane_base + __umul24(y + 1, prow) + 1u + (uint32_t)k) * 8u : OOB;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, word), r_plane, po, 0, RMCV_K1_PLAIN_PLAUX); // (plain: the sparse kernel of the same batch finds the words in L2)
                k += k_step;
                s_ += s_step;
                if (k >= ww) { k -= ww; s_++; }
            }
        }
        if (binary) { // RMCV_STAGE_NO_IMAGE: the 0/255 byte image is not wanted
            // w % 64 == 0: the strip's rows are contiguous both in the LDS plane (ww * 64 == w bits per row) and in the byte
            // image, so the strip is ONE run of 16-pixel items: no (row, group) bookkeeping, and the loop bound is wave-uniform
            const int n_valid = min(sr, h - y0) * wq;
            const uint16_t* R16 = reinterpret_cast<const uint16_t*>(R + __umul24(halo, ww));
            const uint32_t out0 = (uint32_t)((int64_t)f * w * h) + (uint32_t)y0 * (uint32_t)w;
            const int lane = tid & 63;
            for (int base = __builtin_amdgcn_readfirstlane(tid - lane); base < n_valid; base += 256) {
                const int it = base + lane;
                const bool ok = it < n_valid;
                const uint32_t m = R16[ok ? it : 0];
                const uint64_t lo = s_lut[m & 0xFF], hi = s_lut[m >> 8];
                const u32x4v o = {(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
#ifdef RMCV_K1_NOSTORE
                const uint32_t off = OOB;
#else
                const uint32_t off = ok ? out0 + (uint32_t)it * 16u : OOB;
#endif
                __builtin_amdgcn_raw_buffer_store_b128(o, r_bin, off, 0, RMCV_K1_STAUX /* nt: written once, read by nobody here */);
            }
        }
    } else {
        const int items = sr * wq;
        int s = tid / wq, q = tid - s * wq;
        const int dr = 256 / wq, dq = 256 - dr * wq;
        uint8_t* bin = binary ? binary + (int64_t)f * w * h : nullptr;
        uint64_t* plane = bits + (int64_t)f * plane_pitch;
        for (int it = tid; it < items; it += 256) {
            const int y = y0 + s;
            if (y < h) {
                const uint64_t word = R[(s + halo) * ww + (q >> 2)];
                const uint32_t m = (uint32_t)(word >> (16 * (q & 3))) & 0xFFFFu;
                if (binary) { // RMCV_STAGE_NO_IMAGE: the 0/255 byte image is not wanted, only the bit plane below
                    for (int p = 0; p < 16; p++) {
                        int x = q * 16 + p;
                        if (x < w) bin[(int64_t)y * w + x] = ((m >> p) & 1) ? 255 : 0;
                    }
                }
                if ((q & 3) == 0) {
                    plane[(int64_t)(y + 1) * prow + 1 + (q >> 2)] = word;
                }
            }
            s += dr;
            q += dq;
            if (q >= wq) { q -= wq; s++; }
        }
    }
    } // strip loop
    // Leaving: this workgroup has drawn its last index.  strip_ctr[8] counts the leavers; the last one of the launch knows that
    // nobody will draw again and zeroes the eight heads and the count for the next launch (launches of one context are ordered:
    // rmcv_host.hip chains them with an event when the caller changes streams).
#ifdef RMCV_PROFILE_HANDOVER
    if (tid == 0) printf("[kbx] %d %lld\n", xcd, (long long)wall_clock64()); // when this workgroup left: the XCDs' tails
#endif
    if (tid < 64) {
        int left = 0;
        if (tid == 0) left = atomicAdd(&strip_ctr[8 * CTR_STRIDE], 1);
        left = __builtin_amdgcn_readfirstlane(left);
        if (left == (int)gridDim.x - 1 && tid < 9) atomicExch(&strip_ctr[tid * CTR_STRIDE], 0);
#ifdef RMCV_PROFILE_HANDOVER
        if (left == (int)gridDim.x - 1 && tid == 0) printf("[kb end] %lld\n", (long long)wall_clock64());
#endif
    }
}

#include "k_binary_ws.inc"

static std::atomic<int64_t> g_ws_launches{0};
int64_t pixel_ws_launches() { return g_ws_launches.load(std::memory_order_relaxed); }

template <int CA, int CB>
static hipError_t launch_binary_t(const Geom& g, const Bufs& b, int lower_bound, int morph, bool image, int groups,
                                  hipStream_t s)
{
    const int strips = (g.h + SR - 1) / SR;
    int lb = lower_bound, all_pass = 0;
    if (lb <= 0) { all_pass = 1; lb = 1; }
    if (lb > 256) lb = 256;
    const size_t planes = (size_t)2 * (SR + 4) * g.ww * sizeof(uint64_t);
    const bool aligned = (g.w % 64 == 0) && (g.stride % 16 == 0) && (g.frame_pitch % 16 == 0) && ((uintptr_t)b.frames % 16 == 0);
    // The FAST path addresses its buffers with 32-bit offsets, so one launch covers at most as many frames as keep every extent
    // (input, byte image, bit plane) below 4 GiB - 256; a larger batch (288 GB of HBM hold 70 000 frames) is a few launches in a
    // row on the same stream, each with its pointers advanced -- not a fall-back to the byte-wise loader.
    const int64_t lim = 0xFFFFF000ll;
    const int64_t per_frame = std::max<int64_t>(std::max<int64_t>(g.frame_pitch, g.plane_pitch * 8), (int64_t)g.w * g.h);
    const int chunk = aligned ? (int)std::min<int64_t>(g.n_frames, std::max<int64_t>(1, (lim - 1) / per_frame)) : g.n_frames;
    const bool fast = aligned && (int64_t)chunk * per_frame < lim;
    // rows contiguous in memory: the linear loader (Geom::pixel_rowquad, from RMCV_K1_LINEAR=0 when the context is made: the row-quad
    // loader everywhere -- a dev knob for A/B runs)
    const bool linear = fast && !g.pixel_rowquad && g.stride == 3 * g.w;
    // persistent grid: `groups` workgroups per CU (RMCV_OPT_PIXEL_GROUPS; RMCV_K1_BPC overrides for A/B runs): alone the kernel is
    // equally fast with 2 and 3 and slower with 4 and more; 2 leaves room on every CU for the kernels of the other batches in flight
    static const int bpc_env = getenv("RMCV_K1_BPC") ? atoi(getenv("RMCV_K1_BPC")) : 0;
    const int bpc = bpc_env > 0 ? bpc_env : groups;
    for (int f0 = 0; f0 < g.n_frames; f0 += chunk) {
        const int nf = std::min(chunk, g.n_frames - f0);
        const int n_blocks = nf * strips;
        int grid = (g.n_cu > 0 ? g.n_cu : 256) * (bpc > 0 ? bpc : 4); // n_cu: of the context's own device
        if (grid > ((n_blocks + 7) & ~7)) grid = (n_blocks + 7) & ~7;
        grid = (grid + 7) & ~7;
        const int per_xcd = (n_blocks + 7) >> 3;
        int taper_head = 0, taper_tail = 0;
        // A launch with fewer strips than half the CUs (one camera frame = 32 strips on 256 CUs: the per-frame drop-in chain) hands
        // EVERY strip out as four 8-row pieces: four times the workgroups, a quarter of the rows each (15 -> 7 us for one frame).
        if (n_blocks * 2 <= (g.n_cu > 0 ? g.n_cu : 256)) {
            taper_head = per_xcd;
            taper_tail = 0;
            grid = (4 * n_blocks + 7) & ~7;
        }
        const uint8_t* frames = b.frames + (int64_t)f0 * g.frame_pitch;
        uint8_t* binary = image ? b.binary + (int64_t)f0 * g.w * g.h : nullptr;
        uint64_t* bits = b.bits + (int64_t)f0 * g.plane_pitch;
        uint32_t* rowmask = b.rowmask + (int64_t)f0 * g.h;
        // beyond 64 KiB of dynamic LDS (frames wider than ~6700 pixels) the kernel has to be told; per device and instantiation
        static size_t lds_set[MAX_DEVICES][3] = {};
        const int mode = fast ? (linear ? 2 : 1) : 0;
        const int inst = mode;
        if (planes > 60 * 1024 && planes > lds_set[g.device][inst]) {
            const void* fn = mode == 2 ? reinterpret_cast<const void*>(k_binary<CA, CB, 2>) : mode == 1 ? reinterpret_cast<const void*>(k_binary<CA, CB, 1>) : reinterpret_cast<const void*>(k_binary<CA, CB, 0>);
            const hipError_t ea = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)planes);
            if (ea != hipSuccess) return ea;
            lds_set[g.device][inst] = planes;
        }
        // whole batches with contiguous rows, when the caller asks for it (RMCV_OPT_PIXEL_SHAPE; a pipeline does for its calm batches):
        // the wave-specialised kernel, ONE 1024-thread workgroup per CU -- 8 loader wavefronts with 2 items (8 loads) in flight each, 8 storers
        constexpr int WS_NL = 8, WS_NS = 8, WS_RING = 2, WS_AUX = 2 /* nt */;
        const size_t planes_ws = ((size_t)2 * (SR + 4) + SR) * g.ww * sizeof(uint64_t);
        if (g.pixel_ws && linear && !all_pass && taper_head == 0 && planes_ws <= 60 * 1024) {
            K1Args ka;
            ka.frames = frames; ka.frame_pitch = g.frame_pitch; ka.stride = g.stride; ka.n_frames = nf; ka.w = g.w; ka.h = g.h; ka.ww = g.ww;
            ka.lb = lb; ka.morph = morph; ka.binary = binary; ka.bits = bits; ka.prow = g.prow; ka.plane_pitch = g.plane_pitch;
            ka.strips = strips; ka.n_blocks = n_blocks; ka.rowmask = rowmask; ka.strip_ctr = b.strip_ctr;
            int grid_ws = ((g.n_cu > 0 ? g.n_cu : 256) + 7) & ~7;
            if (grid_ws > ((n_blocks + 7) & ~7)) grid_ws = (n_blocks + 7) & ~7;
            g_ws_launches.fetch_add(1, std::memory_order_relaxed);
            // (issue priority 3 for loaders and storers, the sparse kernel's own: in-process A/B against 0 / (2,1) / (3,0) / (1,1):
            // 0.991 / 1.008 / 1.017 / 1.006 of the step)
            // (in the pipeline, in-process A/B against this shape: ring of 3 items 1.005, of 4 1.005; 12 loaders + 4 storers 1.087, 10 + 4
            // 1.017, 8 + 4 1.024; loads without the nt hint 1.062)
            const hipError_t e = launch(k_binary_ws<CA, CB, WS_NL, WS_NS, WS_RING, WS_AUX, 3, 3>, dim3(grid_ws), dim3((WS_NL + WS_NS) * 64), planes_ws, s, ka);
            if (e != hipSuccess) return e;
            continue;
        }
#define RMCV_K1_LAUNCH(F)                                                                                                             \
    launch(k_binary<CA, CB, F>, dim3(grid), dim3(256), planes, s, frames, g.frame_pitch, g.stride, nf, g.w, g.h, g.ww, lb, all_pass, \
           morph, binary, bits, g.prow, g.plane_pitch, strips, n_blocks, rowmask, b.strip_ctr, taper_head, taper_tail,               \
           g.pixel_halo_nt)
        const hipError_t e = mode == 2 ? RMCV_K1_LAUNCH(2) : mode == 1 ? RMCV_K1_LAUNCH(1) : RMCV_K1_LAUNCH(0);
#undef RMCV_K1_LAUNCH
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// launch_binary_t's own rule, for the pipeline's hold-back of a burst's second launch: will the batch bound to (g, b) run as ONE launch of
// k_binary_ws with a workgroup on every CU?
bool binary_ws_full(const Geom& g, const Bufs& b, int lower_bound)
{
    const int strips = (g.h + SR - 1) / SR;
    const bool aligned = (g.w % 64 == 0) && (g.stride % 16 == 0) && (g.frame_pitch % 16 == 0) && ((uintptr_t)b.frames % 16 == 0);
    const int64_t lim = 0xFFFFF000ll;
    const int64_t per_frame = std::max<int64_t>(std::max<int64_t>(g.frame_pitch, g.plane_pitch * 8), (int64_t)g.w * g.h);
    const bool one_launch = aligned && (int64_t)g.n_frames * per_frame < lim;
    const bool linear = one_launch && !g.pixel_rowquad && g.stride == 3 * g.w;
    const int n_cu = g.n_cu > 0 ? g.n_cu : 256, n_blocks = g.n_frames * strips;
    const size_t planes_ws = ((size_t)2 * (SR + 4) + SR) * g.ww * sizeof(uint64_t);
    return g.pixel_ws && linear && lower_bound > 0 && n_blocks * 2 > n_cu && planes_ws <= 60 * 1024 && n_blocks >= n_cu;
}

hipError_t launch_binary(const Geom& g, const Bufs& b, int camp, int lower_bound, int morph, bool image, int groups, hipStream_t s)
{
    // imgproc.cpp:56-65: GUIDELIGHT G-R; BLUE B-R; everything else (RED, NEUTRAL) R-B.  BGR byte order.
    if (camp == RMCV_CAMP_GUIDELIGHT) return launch_binary_t<1, 2>(g, b, lower_bound, morph, image, groups, s);
    if (camp == RMCV_CAMP_BLUE) return launch_binary_t<0, 2>(g, b, lower_bound, morph, image, groups, s);
    return launch_binary_t<2, 0>(g, b, lower_bound, morph, image, groups, s);
}

// binary (0 / non-zero bytes) -> padded bit plane; used when a caller hands in its own binary image
__global__ void k_pack_bits(const uint8_t* __restrict__ binary, int w, int h, int ww, uint64_t* __restrict__ bits, int prow,
                            int64_t plane_pitch, uint32_t* __restrict__ rowmask)
{
    const int f = blockIdx.y;
    const int item = blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= h * ww) return;
    const int y = item / ww, k = item % ww;
    const uint8_t* row = binary + (int64_t)f * w * h + (int64_t)y * w;
    uint64_t word = 0;
    for (int bqt = 0; bqt < 64; bqt++) {
        int x = k * 64 + bqt;
        if (x < w && row[x]) word |= 1ull << bqt;
    }
    bits[(int64_t)f * plane_pitch + (int64_t)(y + 1) * prow + 1 + k] = word;
    if (word && k < 32) atomicOr(&rowmask[(int64_t)f * h + y], 1u << k); // caller zeroes the masks first
}

hipError_t launch_pack_bits(const Geom& g, const Bufs& b, hipStream_t s)
{
    const int items = g.h * g.ww;
    hipError_t e = hipMemsetAsync(b.rowmask, 0, (size_t)g.n_frames * g.h * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    return launch(k_pack_bits, dim3((items + 255) / 256, g.n_frames), dim3(256), 0, s, b.binary, g.w, g.h, g.ww, b.bits,
                       g.prow, g.plane_pitch, b.rowmask);
}

} // namespace rmcv
