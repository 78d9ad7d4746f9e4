// rmcv_internal.h -- shared declarations of the HIP translation units (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <tuple>
#include <utility>

#include "../../include/rmcv_abi.h"

namespace rmcv {

// Geometry of the frames currently bound to a context and of the bit planes derived from them.
// Bit planes (foreground F, "labelled" LAB, "right-exit" NEG) hold one bit per pixel in u64 words,
// bit b of word k = pixel x = 64*k + b.  Each plane is padded with one zero word left and right of
// every row and one zero row above and below the image, so a 3x3 neighbourhood never needs a
// bounds check:  word(y, k) lives at (y + 1) * prow + (k + 1).
struct Geom {
    int device;          // HIP device of the owning context: per-device launch state (function attributes) is indexed by it
    int n_cu;            // compute units of that device (sizes the persistent grid of k_binary)
    int pixel_halo_nt;   // RMCV_OPT_PIXEL_HALO_NT: the row quads a strip shares with its neighbours are loaded non-temporal too
    int pixel_rowquad;   // dev knob (RMCV_K1_LINEAR=0 in the environment when the context is made): k_binary's row-quad loader even where rows are contiguous
    int pixel_ws;        // RMCV_OPT_PIXEL_SHAPE: whole batches with contiguous rows go to k_binary_ws (one 1024-thread workgroup per CU)
    int dense_defer;     // RMCV_OPT_DENSE_DEFER: frames beyond the LDS tables are left to a second launch with 8 wavefronts per frame
    int overloads;       // RMCV_OPT_OVERLOADS: SURVEY A.6, which functions the reference's unqualified abs / atan2 / sin / cos on floats are
    int sparse_lean;     // a pipeline's dense mode: the batch's sparse stage runs the lean build (k_contours_lean.hip), every frame on the mid tier
    int contour_tier;    // RMCV_OPT_CONTOUR_TIER: 0 = per frame (LDS tables, else mid tier, else literal scanner), 1 = literal, 2 = mid tier
    int n_frames;
    int w, h;
    int stride;          // bytes between rows of the BGR input
    int64_t frame_pitch; // bytes between frames of the BGR input
    int ww;              // words per row = ceil(w / 64)
    int prow;            // padded words per row = ww + 2
    int64_t plane_pitch; // words per frame = (h + 2) * prow
};

static constexpr int CTR_STRIDE = 32;   // ints between the heads of k_binary's strip queues (Bufs::strip_ctr): a 128-byte line each, 9 of them
#ifndef RMCV_SR
#define RMCV_SR 32
#endif
static constexpr int STRIP_ROWS = RMCV_SR;   // rows of a k_binary strip (k_binary.hip: SR); the sparse kernel's frame queues follow its strip order
static constexpr int VISIT_CAP = 4096; // border visits of one frame the contour stage holds in LDS (contours_device.h); more -> mid tier
static constexpr int NN_MID = 1 << 17;   // border visits of one frame the mid tier holds (tables in global memory); more -> literal scanner
static constexpr int CAND_MID = 1 << 15; // outer borders (before RETR_EXTERNAL drops the nested ones) the mid tier holds
// bytes of one frame slot's mid-tier scratch block (layout: contours_device.h, mid_tables)
inline size_t mid_bytes(int slot_cap)
{
    return (size_t)slot_cap * (6 * 8 + 2 * 4) + (size_t)NN_MID * (2 * 8 + 2 * 4) + (size_t)CAND_MID * 4 * 4;
}

struct Limits {
    int max_frames, max_width, max_height, max_contours, max_points, max_blobs, max_armours;
};

// Device buffers of one context (all sized by Limits at creation, reused by every call).
struct Bufs {
    const uint8_t* frames; // BGR input (owned upload buffer or borrowed)
    uint8_t* binary;       // [frame][h][w]          0/255           (imgproc.cpp:74 returns it)
    uint64_t* bits;        // [frame] padded plane F (closed binary as bits)
    int* strip_ctr;        // [8] per-XCD strip queue heads of k_binary + [8] = workgroups of the launch that have drawn their
                           // last strip; the last one to leave zeroes all nine, so every launch starts from 0 with no host mirror
    uint32_t* rowmask;     // [frame][h]  bit k: word k of row y of F is non-zero (rows are h apart; k_binary writes them)
    uint64_t* lab;         // [frame] padded plane: pixel was visited by a border trace
    uint64_t* neg;         // [frame] padded plane: ... and got the negative ("right exit") label
    // contours in DISCOVERY order; cv::findContours returns them reversed (oracle/rmcv_oracle.c)
    rmcv_point* points;    // [frame][max_points]
    int32_t* cont_start;   // [frame][max_contours]
    int32_t* cont_len;     // [frame][max_contours]
    int32_t* n_contours;   // [frame]
    int32_t* n_points;     // [frame]
    uint32_t* visit_xy;    // [frame][VISIT_CAP] scratch of the contour stage: packed (x, y, directions) of every border visit
    uint8_t* mid;          // [frame][mid_stride] scratch of the contour stage's mid tier (contours_device.h: MidTables)
    int64_t mid_stride;
    int mid_slot_cap;      // words per frame the mid tier's per-word tables hold (= every word of the largest frame)
    // light blobs (positive list, in findContours order) and the negative list (contour indices)
    rmcv_lightblob* blobs; // [frame][max_blobs]
    int32_t* blob_src;     // [frame][max_blobs]   contour index (findContours order)
    rmcv_rrect* ellipses;  // [frame][max_blobs]   the fitted ellipse of each positive
    int32_t* elig;         // [frame][max_contours] discovery indices of the contours with >= 6 points (fit work list)
    int32_t* n_elig;       // [frame]
    int32_t* slot_kind;    // [frame][max_contours] per contour: 0 skipped, 1 positive, 2 negative
    rmcv_rrect* slot_ell;  // [frame][max_contours] per contour: fitted ellipse
    int32_t* neg_idx;      // [frame][max_contours]
    int32_t* n_blobs;      // [frame]
    int32_t* n_neg;        // [frame]
    rmcv_armour* armours;  // [frame][max_armours]
    int32_t* n_armours;    // [frame]
    int32_t* status;       // [frame] RMCV_FRAME_* bits
    int32_t* frame_order;  // [frame] the frames in k_binary's completion order, interleaved over the XCDs (SparseSched::order)
    // icon classifier (BASELINE config 5); allocated by rmcv_svm_load
    float* svm_w;          // [n_df][1200]
    double* svm_rho;       // [n_df]
    int32_t* svm_labels;   // [n_class]
    int svm_classes;
    int32_t* identity;     // [frame][max_armours]
    uint8_t* icons;        // [frame][max_armours][1200]  rectified 20x20 BGR icons
    // armour pose; allocated by rmcv_pnp_load
    rmcv_pnp_config* pnp_cfg;
    double* base2gripper;  // [frame][16]
    double* poses;         // [frame][max_armours][9]  rvec | tvec | world position
};

// internal value of a frame's status word BETWEEN the two launches of the sparse stage (never seen by a caller: the second launch
// rewrites the word of every frame that carries it)
#define RMCV_FRAME_DEFERRED_ (1 << 30)

// The order in which the sparse kernel's workgroups take frames (Bufs::frame_order): the frames in the order k_binary completes
// them, interleaved over the XCDs the way workgroups are dealt to them -- a frame's planes are then read on the XCD whose L2 they
// were written through.
struct SparseSched {
    const int32_t* order; // [frame] workgroup b -> frame; null: identity
};

static constexpr int MAX_DEVICES = 64; // per-device launch state (hipFuncSetAttribute is per device) is kept in arrays of this size

// Launch a kernel and return the status of THIS launch.  hipLaunchKernelGGL reports errors only through the calling thread's
// sticky last-error slot, which may still hold the error of an unrelated earlier HIP call of the host application (one it
// handled by return code): reading that slot after a launch would turn a launch that ran into a reported failure.
template <typename... P, size_t... I>
inline hipError_t launch_tuple(void (*kernel)(P...), dim3 grid, dim3 block, size_t lds, hipStream_t s, std::tuple<P...>& vals,
                               std::index_sequence<I...>)
{
    void* ptrs[] = {static_cast<void*>(const_cast<typename std::remove_const<P>::type*>(&std::get<I>(vals)))...};
    return hipLaunchKernel(reinterpret_cast<const void*>(kernel), grid, block, ptrs, lds, s);
}
template <typename... P, typename... A>
inline hipError_t launch(void (*kernel)(P...), dim3 grid, dim3 block, size_t lds, hipStream_t s, A&&... a)
{
    static_assert(sizeof...(P) == sizeof...(A), "argument count differs from the kernel's parameter list");
    std::tuple<P...> vals{static_cast<P>(a)...};
    return launch_tuple(kernel, grid, block, lds, s, vals, std::index_sequence_for<P...>{});
}

int64_t pixel_ws_launches(); // launches of k_binary_ws by this process (rmcv_pixel_ws_launches)
// kernel launchers (each enqueues on `s` and returns the launch error)
hipError_t launch_match(const Geom& g, const Bufs& b, const Limits& lim, const rmcv_params& p, const rmcv_legacy_params& lp,
                        int mode, bool with_frames, bool pairs, hipStream_t s);
// identity: the frame's armours are classified by the same kernel (RMCV_STAGE_IDENTITY; needs pairs)
hipError_t launch_sparse(const Geom& g, const Bufs& b, const Limits& lim, const rmcv_params& p, bool pairs, bool identity, int waves, hipStream_t s);
hipError_t launch_pnp(const Geom& g, const Bufs& b, const Limits& lim, hipStream_t s);
hipError_t launch_binary(const Geom& g, const Bufs& b, int camp, int lower_bound, int morph, bool image, int groups, hipStream_t s);
bool binary_ws_full(const Geom& g, const Bufs& b, int lower_bound); // the batch will run as one launch of k_binary_ws with a workgroup on every CU
bool sparse_lean_applies(const Geom& g, const Bufs& b); // Geom::sparse_lean can be honoured for what is bound (k_contours.hip)
hipError_t launch_contours(const Geom& g, const Bufs& b, const Limits& lim, hipStream_t s);
hipError_t launch_blobs(const Geom& g, const Bufs& b, const Limits& lim, const rmcv_params& p, hipStream_t s);
hipError_t launch_armours(const Geom& g, const Bufs& b, const Limits& lim, const rmcv_params& p, hipStream_t s);
hipError_t launch_blobs_armours(const Geom& g, const Bufs& b, const Limits& lim, const rmcv_params& p, hipStream_t s);
// d_status_or (nullable): TWO words = the OR of the batch's per-frame status words, the number of frames with RMCV_FRAME_MID_PATH; hd_record (nullable, needs d_status_or): the
// record [frame_offs | ... status ... | armours at host_head] once more, in mapped pinned host memory (device address)
hipError_t launch_compact_armours(const Geom& g, const Bufs& b, const Limits& lim, rmcv_armour* d_out, int cap,
                                  int32_t* d_frame_offs, hipStream_t s, int32_t* d_status_or = nullptr, uint8_t* hd_record = nullptr, int host_head = 0);
hipError_t launch_status_clear(const Geom& g, const Bufs& b, int mask, hipStream_t s); // status[f] &= ~mask
hipError_t launch_classify(const Geom& g, const Bufs& b, const Limits& lim, hipStream_t s);
// stage-wise helpers: binary (host-supplied 0/255 image) -> bit plane
hipError_t launch_pack_bits(const Geom& g, const Bufs& b, hipStream_t s);
// contours in findContours order as CSR (for download); d_offs has max_contours+1 entries per frame
// The per-frame chain's results on their way to the host in ONE kernel: up to 8 lists (src on the device, dst in pinned host
// memory mapped into the device's address space) of `count[0] + count_add` elements, clamped to `max_elems`, and up to 12 header
// words.  Nine small device-to-host copies in a row, each a hand-over to the copy engine, cost the chain 40-60 us; the kernel's
// own stores cross PCIe as posted writes.
struct ExportSec {
    const void* src;
    void* dst;
    const int32_t* count; // null: max_elems elements
    int count_add, elem_bytes, max_elems;
};
struct ExportArgs {
    ExportSec sec[8];
    int n_sec;
    const int32_t* hdr_src[12]; // header word i = *hdr_src[i] (null: left alone)
    int32_t* hdr_dst;
};
hipError_t launch_export(const ExportArgs& a, hipStream_t s);
hipError_t launch_pack_contours(const Geom& g, const Bufs& b, const Limits& lim, rmcv_point* d_pts_out, int32_t* d_offs_out,
                                int32_t* d_hdr /* nullable: frame 0's {n_contours, n_points, status} */, hipStream_t s, const ExportArgs* ex = nullptr); // ex: frame 0's workgroup also exports (one-frame chains)
hipError_t launch_gather3(const int32_t* a, const int32_t* b, const int32_t* c, int32_t* d_out, hipStream_t s); // d_out[0..2] = *a, *b, *c

// ---- what rmcv_pipeline.hip needs of a context beyond the public ABI (rmcv_host.hip) ----
// external order: the pipeline chains a context's launches with its own events (it knows which stream ran what), so the context
// does not record / wait for its own ordering event around every launch (two HIP calls per launch); rmcv_batch_sync and the getters
// then wait for `done` (recorded by the pipeline behind the slot's last launch) instead
void ctx_external_order(rmcv_ctx* c, hipEvent_t done);
// rmcv_batch_compact_armours + the batch's OR-ed status word
int ctx_compact(rmcv_ctx* c, void* d_armours_out, int cap, void* d_frame_offs, void* d_status_or, hipStream_t s, void* hd_record = nullptr, int host_head = 0);
const Limits& ctx_limits(const rmcv_ctx* c);
// 1: whole batches with contiguous rows go to the wave-specialised pixel kernel (k_binary_ws.inc), 0: k_binary
void ctx_pixel_shape(rmcv_ctx* c, int shape);
// rmcv_batch_set_device_frames without a blocking call: a change of geometry (planes zeroed, frame order recomputed) is ENQUEUED on `s`,
// which the caller has made wait for the context's last batch
int ctx_bind_frames(rmcv_ctx* c, const void* d_frames, int n_frames, int w, int h, int stride, int64_t frame_pitch, hipStream_t s);
// everything binding a full batch would allocate (the mid tier's scratch), now
int ctx_prepare_ring(rmcv_ctx* c);
// allocations, host-side synchronisations and blocking copies this context has made while binding geometries
uint64_t ctx_blocking_calls(const rmcv_ctx* c);
int ctx_wait_timeout_ms(const rmcv_ctx* c);
bool pixel_ws_full(const rmcv_ctx* c, int lower_bound); // binary_ws_full of what is bound to the context
// waits that poll with a deadline instead of parking the thread in the runtime: 0 done, 1 deadline passed, -1 HIP error (*err)
int wait_stream_deadline(hipStream_t s, int timeout_ms, hipError_t* err);
int wait_event_deadline(hipEvent_t ev, int timeout_ms, hipError_t* err);
hipError_t launch_delay(unsigned long long ns, hipStream_t s); // holds `s` back for `ns` nanoseconds
// what rmcv_batch_run would refuse for (p, stages), checked without enqueuing anything
int ctx_check_stages(rmcv_ctx* c, const rmcv_params* p, int stages);
// Geom::dense_defer for the runs that follow: 0 off, 1 both launches on the run's stream (RMCV_OPT_DENSE_DEFER), 2 / 3 the first / second only
void ctx_defer_phase(rmcv_ctx* c, int phase);
// Geom::sparse_lean for the runs that follow
void ctx_sparse_lean(rmcv_ctx* c, int on);

} // namespace rmcv
