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
// Fast path (cycles_frame, contours_device.h): every visit of the border follower to a pixel is a node, the follower's step a
// bijection on nodes, a border a cycle; all nodes of a frame are built at once, linked and ranked by pointer doubling.  The
// RETR_EXTERNAL bookkeeping is then verified on the merged labels; frames where it does not hold (nested components) take the
// literal scanner:
// k_contours_literal: one wavefront per frame.  Lane l owns row 64*band + l of the current band and
// keeps the first acceptable candidate of its row; the wave repeatedly takes the raster-first one,
// traces it (border following on a 3-row x 64-bit register window of F), ORs the labels into
// LAB/NEG, and re-scans only the rows the trace touched below the start.  Exact for every input.
//
// Output is kept in DISCOVERY order (cont_start/cont_len); findContours order is its reverse.
//
// The kernel body lives in k_contours_kernel.inc and is compiled twice, in two translation units: with 8 wavefronts per frame here,
// with 4 in k_contours_w4.hip.  One kernel per translation unit keeps every big device function at a single call site, i.e. inlined:
// as a template (or two kernels in one file) the shared out-of-line copies cost 21 VGPRs and a stack frame -- enough to lose the
// co-residency with the pixel kernel that the 8-wavefront form relies on.
#include "contours_device.h"

namespace rmcv {

#define KC_KERNEL k_contours_w8
#define KC_THREADS 512
#include "k_contours_kernel.inc"
#undef KC_KERNEL
#undef KC_THREADS

static_assert(sizeof(ContoursLds) >= (CT_THREADS_MAX / 64) * sizeof(WaveLds), "the fit rows reuse the contour tables");

// Geom::sparse_lean (a pipeline's dense mode) can be honoured: the mid tier applies to every frame (its scratch is there, the row tables
// cover the frame).
bool sparse_lean_applies(const Geom& g, const Bufs& b)
{
    return g.sparse_lean && !g.dense_defer && g.contour_tier == 0 && b.mid && g.h <= CT_MAXH && g.ww <= 32;
}

static hipError_t launch_contours_x(const Geom& g, const Bufs& b, const Limits& lim, const SparseTail& X, int waves, hipStream_t s)
{
    SparseSched Q;
    Q.order = b.frame_order;
    const int grid = g.n_frames;
    static const int force_literal = getenv("RMCV_CONTOURS_LITERAL") ? atoi(getenv("RMCV_CONTOURS_LITERAL")) : 0; // dev/test knob
    static bool attr_set[MAX_DEVICES] = {}; // hipFuncSetAttribute applies to the current device only (a process may drive several)
    if (!attr_set[g.device]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_contours_w8), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_bytes(CT_MAXH));
        if (e != hipSuccess) return e;
        attr_set[g.device] = true;
    }
    const int force = force_literal ? force_literal : g.contour_tier;
    // a pipeline's dense mode: the whole batch through the lean build (every frame on the mid tier, 61 KB of LDS: two workgroups per CU);
    // not with a classifier stage (the lean build has no room for its feature rows)
    if (waves == 4 && X.fused && !X.C.enabled && sparse_lean_applies(g, b) && force == 0) return launch_contours_lean(g, b, lim, X, 2, Q, grid, s);
    if (waves == 4) {
        // RMCV_OPT_DENSE_DEFER (off by default): the 4-wavefront launch leaves the frames beyond its LDS tables alone and the
        // 8-wavefront kernel takes them in a launch of its own right behind (1.4-1.5x faster per frame; a workgroup of any other
        // frame reads one word and ends).  Pays where every frame is dense, costs where a few are: DESIGN.md 5c.
        // g.dense_defer: 1 = both launches here, on this stream; 2 / 3 = the first / the second launch only (rmcv_pipeline.hip puts the
        // second on a stream of its own, so that a batch's few dense frames do not hold up the sparse stream)
        const bool defer = g.dense_defer && (force & 3) == 0 && b.mid;
        if (g.dense_defer == 3 && !defer) return hipSuccess; // (nothing was deferred: the first launch finished every frame)
        hipError_t e = g.dense_defer == 3 ? hipSuccess : launch_contours_w4(g, b, lim, X, force | (defer ? 4 : 0), Q, grid, s);
        if (e != hipSuccess || !defer || g.dense_defer == 2) return e;
        const SparseSched& Q2 = Q;
        static const bool second_w4 = getenv("RMCV_DEFER_W4") && atoi(getenv("RMCV_DEFER_W4")); // dev knob: the second launch with 4 wavefronts per frame too
        if (second_w4) return launch_contours_w4(g, b, lim, X, 2 | 8, Q2, grid, s);
        return launch(k_contours_w8, dim3(grid), dim3(512), lds_bytes(g.h), s, b.bits, b.rowmask, g.h, b.lab, b.neg, g.w,
                       g.h, g.ww, g.prow, g.plane_pitch, b.points, b.cont_start, b.cont_len, b.n_contours, b.n_points, b.status,
                       lim.max_contours, lim.max_points, 2 | 8, b.elig, b.n_elig, b.slot_kind, X, b.visit_xy, b.mid, b.mid_stride,
                       b.mid_slot_cap, Q2, lds_rows_cap(g.h));
    }
    return launch(k_contours_w8, dim3(grid), dim3(512), lds_bytes(g.h), s, b.bits, b.rowmask, g.h, b.lab, b.neg, g.w,
                       g.h, g.ww, g.prow, g.plane_pitch, b.points, b.cont_start, b.cont_len, b.n_contours, b.n_points, b.status,
                       lim.max_contours, lim.max_points, force, b.elig, b.n_elig, b.slot_kind, X, b.visit_xy, b.mid, b.mid_stride,
                       b.mid_slot_cap, Q, lds_rows_cap(g.h));
}

hipError_t launch_contours(const Geom& g, const Bufs& b, const Limits& lim, hipStream_t s)
{
    SparseTail X;
    memset(&X, 0, sizeof(X));
    return launch_contours_x(g, b, lim, X, 8, s);
}

// findContours + filter_lightblobs (+ filter_armours) of every frame in ONE launch
hipError_t launch_sparse(const Geom& g, const Bufs& b, const Limits& lim, const rmcv_params& p, bool pairs, bool identity, int waves, hipStream_t s)
{
    SparseTail X;
    memset(&X, 0, sizeof(X));
    X.fused = 1;
    X.G = {p.tilt_max, p.ratio_lo, p.ratio_hi, p.area_lo, p.area_hi, g.overloads};
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
    T.ov = g.overloads;
    if (identity && pairs) X.C = classify_args(g, b);
    return launch_contours_x(g, b, lim, X, waves, s);
}

// see ExportArgs (rmcv_internal.h): the lists and header words, by `nthreads` threads of which this is number `gtid`
__device__ __forceinline__ void export_lists(const ExportArgs& a, int gtid, int nthreads)
{
    if (gtid < 12 && a.hdr_src[gtid]) a.hdr_dst[gtid] = *a.hdr_src[gtid];
    for (int k = 0; k < a.n_sec; k++) {
        const ExportSec& e = a.sec[k];
        int n = e.count ? e.count[0] + e.count_add : e.max_elems;
        n = n < 0 ? 0 : (n > e.max_elems ? e.max_elems : n);
        const int ndw = n * (e.elem_bytes >> 2);
        const uint32_t* src = static_cast<const uint32_t*>(e.src);
        uint32_t* dst = static_cast<uint32_t*>(e.dst);
        for (int i = gtid; i < ndw; i += nthreads) dst[i] = src[i];
    }
}

__global__ void k_export(ExportArgs a)
{
    export_lists(a, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

hipError_t launch_export(const ExportArgs& a, hipStream_t s)
{
    return launch(k_export, dim3(8), dim3(256), 0, s, a);
}

// contours of every frame as CSR in findContours order (reverse discovery), for download
__global__ void k_pack_contours(const rmcv_point* __restrict__ points, const int32_t* __restrict__ cont_start,
                                const int32_t* __restrict__ cont_len, const int32_t* __restrict__ n_contours, int max_contours,
                                int max_points, rmcv_point* __restrict__ pts_out, int32_t* __restrict__ offs_out,
                                int32_t* __restrict__ hdr_out, const int32_t* __restrict__ status, ExportArgs ex)
{
    const int f = blockIdx.x;
    const int n = n_contours[f];
    const int32_t* cs = cont_start + (int64_t)f * max_contours;
    const int32_t* cl = cont_len + (int64_t)f * max_contours;
    int32_t* offs = offs_out + (int64_t)f * (max_contours + 1);
    // Offsets by a workgroup scan over the lengths, 1024 contours per round, kept in LDS for the copy that follows: as one thread
    // walking the list (two dependent global loads per contour) the kernel took 10-13 us for a camera frame's 20 contours.
    constexpr int PK = 1024, PER = PK / 256;
    __shared__ int s_len[PK], s_src[PK], s_o[PK], s_scan[256], s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_base = 0;
    for (int c0 = 0; c0 < n; c0 += PK) {
        const int m = n - c0 < PK ? n - c0 : PK;
        __syncthreads(); // s_base of the previous round; its tables are free
        int mine[PER], sum = 0;
#pragma unroll
        for (int u = 0; u < PER; u++) { // thread t owns entries PER*t .. PER*t + PER-1 of the round (findContours order = reverse discovery)
            const int i = PER * tid + u;
            int eff = 0, st = 0;
            if (i < m) {
                const int k = n - 1 - (c0 + i);
                st = cs[k];
                const int len = cl[k], room = max_points - st;
                eff = len < room ? len : (room > 0 ? room : 0);
                s_len[i] = eff;
                s_src[i] = st;
            }
            mine[u] = eff;
            sum += eff;
        }
        s_scan[tid] = sum;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) { // Hillis-Steele inclusive scan
            const int v = tid >= d ? s_scan[tid - d] : 0;
            __syncthreads();
            s_scan[tid] += v;
            __syncthreads();
        }
        int o = s_base + s_scan[tid] - sum;
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int i = PER * tid + u;
            if (i < m) { s_o[i] = o; offs[c0 + i] = o; }
            o += mine[u];
        }
        __syncthreads();
        for (int i = wave; i < m; i += 4) { // one wavefront per contour
            const int len = s_len[i];
            const rmcv_point* src = points + (int64_t)f * max_points + s_src[i];
            rmcv_point* dst = pts_out + (int64_t)f * max_points + s_o[i];
            for (int j = lane; j < len; j += 64) dst[j] = src[j];
        }
        __syncthreads();
        if (tid == 255) s_base += s_scan[255];
    }
    __syncthreads();
    if (tid == 0) {
        const int o = s_base;
        offs[n] = o;
        if (hdr_out && f == 0) { // the per-frame path reads sizes and status with one small copy
            hdr_out[0] = n;
            hdr_out[1] = o;
            hdr_out[2] = status[0];
        }
    }
    // the per-frame chain running ahead (one frame, everything of it computed already): this workgroup also sends the results home
    if (ex.hdr_dst && f == 0) {
        __threadfence_block(); // (what it exports it has written itself or an earlier kernel has: workgroup scope -- no L2 write-back)
        __syncthreads(); // the CSR this workgroup has just written is what it exports
        export_lists(ex, threadIdx.x, blockDim.x);
    }
}

__global__ void k_gather3(const int32_t* __restrict__ a, const int32_t* __restrict__ b, const int32_t* __restrict__ c,
                          int32_t* __restrict__ out)
{
    if (threadIdx.x == 0) out[0] = *a;
    if (threadIdx.x == 1) out[1] = *b;
    if (threadIdx.x == 2) out[2] = *c;
}

hipError_t launch_gather3(const int32_t* a, const int32_t* b, const int32_t* c, int32_t* d_out, hipStream_t s)
{
    return launch(k_gather3, dim3(1), dim3(64), 0, s, a, b, c, d_out);
}

hipError_t launch_pack_contours(const Geom& g, const Bufs& b, const Limits& lim, rmcv_point* d_pts_out, int32_t* d_offs_out,
                                int32_t* d_hdr, hipStream_t s, const ExportArgs* ex)
{
    ExportArgs none;
    memset(&none, 0, sizeof(none));
    return launch(k_pack_contours, dim3(g.n_frames), dim3(256), 0, s, b.points, b.cont_start, b.cont_len, b.n_contours,
                       lim.max_contours, lim.max_points, d_pts_out, d_offs_out, d_hdr, b.status, ex ? *ex : none);
}

} // namespace rmcv
