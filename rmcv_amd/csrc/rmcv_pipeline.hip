// rmcv_pipeline.hip -- rmcv_pipeline_*: the pipelined batch schedule behind the C-ABI (include/rmcv_abi.h).
//
// The reference's process_function (/root/reference/executable/main.cpp:163-209) is a loop: newest frame in, three detection calls,
// armours out.  On one MI355X the loop's batch form keeps `depth` batches in flight: the HBM-bound pixel kernel of batch i + 1 streams
// while the latency-bound per-frame kernel of batch i (contours, fits, pairing: a few waves per CU) runs beside it.  Rounds 1-3 had
// this schedule in bench.py (Python + torch streams and events); it lives here now, in the host language of the reference, and
// bench.py, tools/pipeline_bench.c and a C++ host all drive the same three calls.
//
// One slot of the ring = one context (own work buffers) + one record in HBM (frame_offs | status | armours: the payload of the
// multi-GPU gather) + its pinned host mirror.  Ticket t uses slot t % depth, pixel stream t % pixel_streams and sparse stream
// (t % depth) % sparse_streams -- a slot always meets the same sparse stream, so a record's rewrite is ordered behind its last
// reader on that stream by stream order alone.  Events per slot:
//     ev_done   behind the compaction: the slot's context buffers are free            -> waited for by the slot's next pixel kernel
//     ev_bin    behind the pixel kernel                                               -> waited for by the slot's sparse kernel
//     ev_host   behind the record's gather (rmcv_pipeline_set_gather)                   -> waited for by collect / wait (host)
//     ev_hook   (the hook's own, optional) the record has been read on another stream -> waited for by the slot's next compaction
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <new>
#include <vector>

#include "rmcv_internal.h"

using namespace rmcv;


struct rmcv_pipeline {
    int device = 0;
    rmcv_pipeline_config cfg{};
    Limits lim{};
    int64_t head_bytes = 0, record_bytes = 0;
    std::vector<rmcv_ctx*> ring;
    std::vector<hipStream_t> pix, sp, dn; // pixel streams, sparse streams, streams of the dense frames' second launch
    std::vector<hipEvent_t> ev_bin, ev_done, ev_host, ev_sp; // ev_sp: behind a slot's first sparse launch
    std::vector<void*> ev_hook;       // per slot: the event the hook handed back for the slot's last record (not owned), or null
    std::vector<uint8_t*> d_rec, h_rec, hd_rec; // the record in HBM, its pinned host mirror, the mirror's device address
    std::vector<uint64_t> slot_ticket; // ticket + 1 of the batch that lives in the slot (0: none yet)
    std::vector<int> slot_frames;
    std::vector<hipStream_t> slot_stream; // the stream the slot's record was finished on
    uint64_t next_ticket = 0, collected = 0;
    // Contexts in rotation.  What a batch writes with ordinary stores and reads back right away -- the pixel kernel's bit plane
    // (46 MB per batch at 1280x1024), the sparse kernel's planes, points and tables -- LIVES in the 256 MB Infinity Cache while few
    // enough contexts take turns: with 4 the bit planes are never written to HBM at all; with 8 every one of their cache lines is a
    // miss and the pixel kernel runs 3-15 % slower (profiles/r04f_*).  So while the batches are CALM (no frame of the record that last
    // came back went beyond findContours' LDS tables, and no classifier / pose stage is asked for) the batches use the first `hot`
    // contexts in turn -- the slot (record, events, streams, ticket window) is still one of `depth` -- and the wave-specialised pixel
    // kernel, which wants its CU to itself.  A context's next batch waits for its last one's list (ev_done of that slot): `hot` batches
    // back instead of `depth`, slack enough for sparse frames, a stall of the pixel stream for a batch of dense ones (0.5-1 ms of
    // sparse work) -- those run as before: every slot its own context, k_binary, the ring's full depth as slack.
    int hot = 0;                        // contexts in rotation while calm (0: a slot always uses its own)
    bool calm = false;
    uint64_t hot_seq = 0, hot_batches = 0;
    std::vector<int> ctx_last;          // per context: the slot of its last batch (-1: none)
    std::vector<int> slot_ctx;          // per slot: the context of its batch
    std::vector<hipEvent_t> ev_free;    // per slot: behind the LAST READER of its batch's pixel outputs (the sparse stage; before the compaction)
    // The back half of the NEWEST batch (sparse stage, compaction, events, hook) is enqueued by the next call, not by its own submit:
    // the next submit enqueues it as before -- nothing is lost, its first kernel waits for the pixel kernel anyway --, but a call that
    // WAITS for the newest batch (wait / collect of it, drain) finds that no pixel launch will be beside it and runs it with 8
    // wavefronts per frame: the last batch of a burst, or a host that submits one batch at a time, gets the latency kernel (0.09
    // against 0.18 ms alone).
    struct Pending {
        bool valid = false;
        uint64_t t = 0;
        size_t k = 0;
        rmcv_ctx* c = nullptr;
        rmcv_params p{};
        rmcv_legacy_params lp{};
        bool has_lp = false, used = false, heavy = false;
        int sparse = 0, n_frames = 0;
        hipStream_t B = nullptr;
    } pend;
    uint64_t latency_batches = 0;
    bool lazy_back = true;              // (dev knob RMCV_LAZY_BACK=0)
    bool chain_cold = true, was_cold = false; // (dev knob RMCV_CHAIN_COLD=0)
    int chain_cold_us = 60;             // (dev knob RMCV_CHAIN_COLD=n > 1: the delay in microseconds; 1: wait for the first launch's end instead)
    bool early_free = true;             // (dev knob RMCV_EARLY_FREE=0: a context's next batch waits for the whole list, as ev_done)
    // DENSE MODE (round 5).  While the records that come back say the batches are heavy -- more than an eighth of the frames beyond
    // findContours' LDS tables, or 1 500 border points per frame and more (a plain frame has 650) -- the stream is bound by its sparse
    // stage, not by the pixel kernel: one workgroup of the standard sparse kernel per CU, 0.2-0.6 ms per frame.  Such batches run the LEAN
    // build of the sparse kernel (k_contours_lean.hip: every frame on the mid tier, 61 KB of LDS instead of 80).  The way back: fewer than
    // 1 200 points per frame (in this mode every frame reports the mid tier, so only the points say what the stream is like).
    // Measured (tools/dense_mode_ab.sh, process against process on one box, ms per step off / on): dense2 0.320 / 0.291, dense3 0.396 /
    // 0.372, dense4 0.571 / 0.512; with ONE pixel workgroup per CU and launch as well (room for two lean workgroups per CU, but the pixel
    // kernel needs four resident workgroups to hide its latency): 0.314 / 0.368 / 0.504 -- not kept as the default.
    bool heavy = false;
    int heavy_pixel_groups = 0;        // pixel workgroups per CU and launch in dense mode; 0: as configured (dev knob RMCV_HEAVY_PG; -1: dense mode off)
    uint64_t heavy_batches = 0;
    std::vector<char> slot_lean;       // per slot: its batch ran in dense mode (every frame of its record reports the mid tier)
    bool split_now = false;            // the batches of the moment have a FEW dense frames: give those a launch and a stream of their own
    uint64_t split_batches = 0;        // batches submitted that way
    rmcv_pipeline_hook hook = nullptr;
    void* hook_user = nullptr;
    // built-in gather hook
    rmcv_comm* comm = nullptr;
    int root = 0, n_ranks = 0, rank = 0;
    std::vector<uint8_t*> d_recv;      // root: per slot, n_ranks x record_bytes
    hipEvent_t ev_gather = nullptr;    // behind the last gather: one communicator's operations run in ONE order on every rank
    bool gather_pending = false;
    // Nothing in submit blocks the host (round 5): the mid tier's scratch of every ring context is allocated at creation, a change of
    // geometry is enqueued (planes zeroed, frame order recomputed on the batch's pixel stream), a slot's change of finishing stream
    // is an event wait on the GPU.  The counter proves it: allocations, host-side synchronisations and blocking copies made inside
    // submit (by the pipeline or by the contexts' binding of a geometry) since the pipeline was created.
    uint64_t blocking_base = 0, own_blocking = 0, held_back = 0;
    std::vector<hipEvent_t> ev_chg;    // per slot: the tail of the stream the slot's record was finished on, when that stream changes
    int wait_timeout_ms = 5000;        // rmcv_pipeline_set_wait_timeout
    const char* last_what = "nothing"; // the enqueue made last (PCHK's label): named when a wait runs out
    int hot_cfg = 0;                   // rmcv_pipeline_config::hot_contexts as given (0: derived from the bound geometry)
    int64_t hot_plane_bytes = 0;       // ... the bit planes' bytes of a batch of the geometry `hot` was derived for
    bool ws_always = false;            // dev knob RMCV_WS_ALWAYS: the wave-specialised pixel kernel for every batch, hot rotation or not
    bool hot_identity = false;         // batches with a classifier stage take turns at the hot contexts too: measured in round 5 (three contexts at
                                       // 256 x 1920x1200: 0.514 against 0.426 ms per step), off; RMCV_HOT_IDENTITY=1 in a dev build
    double max_submit_us = 0;          // the longest single submit call (host time) since rmcv_pipeline_reset_stats
    char err[256] = {0};
};

static int finish_back(rmcv_pipeline* pl, bool latency);
static int hot_for(const rmcv_pipeline* pl, int n_frames, int w, int h);

static int pfail(rmcv_pipeline* pl, int code, const char* what, hipError_t e = hipSuccess)
{
    if (pl) {
        if (e != hipSuccess) snprintf(pl->err, sizeof(pl->err), "%s: %s", what, hipGetErrorString(e));
        else snprintf(pl->err, sizeof(pl->err), "%s", what);
    }
    if (e != hipSuccess) (void)hipGetLastError();
    return code;
}
#define PCHK(pl, call, what)                                               \
    do {                                                                   \
        (pl)->last_what = what;                                            \
        hipError_t e__ = (call);                                           \
        if (e__ != hipSuccess) return pfail((pl), RMCV_ERR_HIP, what, e__); \
    } while (0)
// a context call failed: its message is the pipeline's
static int cfail(rmcv_pipeline* pl, rmcv_ctx* c, int rc)
{
    snprintf(pl->err, sizeof(pl->err), "%s", rmcv_last_error(c));
    return rc;
}

// a wait with the pipeline's deadline: RMCV_ERR_TIMEOUT names the enqueue made last
static int pwait_event(rmcv_pipeline* pl, hipEvent_t ev, const char* what)
{
    hipError_t e = hipSuccess;
    const int rcw = wait_event_deadline(ev, pl->wait_timeout_ms, &e);
    if (rcw < 0) return pfail(pl, RMCV_ERR_HIP, what, e);
    if (rcw > 0) {
        snprintf(pl->err, sizeof(pl->err), "%s: not finished after %d ms (rmcv_pipeline_set_wait_timeout); enqueued last: %s", what, pl->wait_timeout_ms, pl->last_what);
        return RMCV_ERR_TIMEOUT;
    }
    return RMCV_OK;
}
static int pwait_stream(rmcv_pipeline* pl, hipStream_t st, const char* what)
{
    hipError_t e = hipSuccess;
    const int rcw = wait_stream_deadline(st, pl->wait_timeout_ms, &e);
    if (rcw < 0) return pfail(pl, RMCV_ERR_HIP, what, e);
    if (rcw > 0) {
        snprintf(pl->err, sizeof(pl->err), "%s: not finished after %d ms (rmcv_pipeline_set_wait_timeout); enqueued last: %s", what, pl->wait_timeout_ms, pl->last_what);
        return RMCV_ERR_TIMEOUT;
    }
    return RMCV_OK;
}
static uint64_t ring_blocking(const rmcv_pipeline* pl)
{
    uint64_t n = 0;
    for (auto c : pl->ring) n += ctx_blocking_calls(c);
    return n;
}

extern "C" {

void rmcv_default_pipeline_config(rmcv_pipeline_config* c)
{
    if (!c) return;
    memset(c, 0, sizeof(*c));
    c->depth = 8;          // measured by alternating regions of one process (round 3): 8 batches over 4 sparse streams run 4.3-4.5 % ahead of 4 over 2
    c->pixel_streams = 2;
    c->sparse_streams = 4;
    c->armour_cap = 0;     // resolved against the limits at creation: 8 per frame
    c->sparse_waves = 4;
    c->pixel_groups = 2;
    c->host_results = 1;
    c->dense_streams = 4;
    c->hot_contexts = 0;   // derived: as many contexts as keep the batches' bit planes inside the Infinity Cache (4 at 256 x 1280x1024: measured in
                           // round 4, process against process on five boxes: 4 < 5 << 3, 6; 0.236-0.243 against 0.244-0.268 ms per step)
}

void rmcv_pipeline_destroy(rmcv_pipeline* pl)
{
    if (!pl) return;
    hipSetDevice(pl->device);
    if (!pl->ring.empty()) (void)finish_back(pl, true);
    {   // with the deadline: batches that do not finish are not waited for without one -- the pipeline is leaked instead
        bool stuck = false;
        for (auto s : pl->pix) if (s) stuck |= pwait_stream(pl, s, "destroy") == RMCV_ERR_TIMEOUT;
        for (auto s : pl->sp) if (s) stuck |= pwait_stream(pl, s, "destroy") == RMCV_ERR_TIMEOUT;
        for (auto s : pl->dn) if (s) stuck |= pwait_stream(pl, s, "destroy") == RMCV_ERR_TIMEOUT;
        if (stuck) {
            fprintf(stderr, "rmcv_pipeline_destroy: %s; the pipeline's buffers are leaked\n", pl->err);
            return;
        }
    }
    for (auto c : pl->ring) rmcv_ctx_destroy(c);
    for (auto e : pl->ev_bin) if (e) hipEventDestroy(e);
    for (auto e : pl->ev_done) if (e) hipEventDestroy(e);
    for (auto e : pl->ev_host) if (e) hipEventDestroy(e);
    for (auto e : pl->ev_sp) if (e) hipEventDestroy(e);
    for (auto e : pl->ev_free) if (e) hipEventDestroy(e);
    for (auto e : pl->ev_chg) if (e) hipEventDestroy(e);
    if (pl->ev_gather) hipEventDestroy(pl->ev_gather);
    for (auto p : pl->d_rec) if (p) hipFree(p);
    for (auto p : pl->d_recv) if (p) hipFree(p);
    for (auto p : pl->h_rec) if (p) hipHostFree(p);
    for (auto s : pl->pix) if (s) hipStreamDestroy(s);
    for (auto s : pl->sp) if (s) hipStreamDestroy(s);
    for (auto s : pl->dn) if (s) hipStreamDestroy(s);
    delete pl;
}

int rmcv_pipeline_create(int device, const rmcv_limits* limits, const rmcv_pipeline_config* cfg, rmcv_pipeline** out)
{
    if (!out) return RMCV_ERR_BAD_ARG;
    *out = nullptr;
    rmcv_pipeline_config d;
    rmcv_default_pipeline_config(&d);
    if (cfg) {
        if (cfg->depth > 0) d.depth = cfg->depth;
        if (cfg->pixel_streams > 0) d.pixel_streams = cfg->pixel_streams;
        if (cfg->sparse_streams > 0) d.sparse_streams = cfg->sparse_streams;
        if (cfg->armour_cap > 0) d.armour_cap = cfg->armour_cap;
        // alone a batch has the CUs to itself: the latency settings (8 wavefronts per frame, 3 pixel workgroups per CU)
        d.sparse_waves = cfg->sparse_waves > 0 ? cfg->sparse_waves : (d.depth >= 3 ? 4 : 8);
        d.pixel_groups = cfg->pixel_groups > 0 ? cfg->pixel_groups : (d.depth >= 2 ? 2 : 3);
        if (cfg->host_results > 0) d.host_results = cfg->host_results;
        if (cfg->dense_streams != 0) d.dense_streams = cfg->dense_streams;
        if (cfg->hot_contexts != 0) d.hot_contexts = cfg->hot_contexts;
    }
    if (d.dense_streams < 0 || d.sparse_waves != 4 || d.host_results != 1) d.dense_streams = 0; // (the deferral exists for the 4-wavefront kernel; the policy reads the host mirror)
    if (d.depth > 64 || d.pixel_streams > 16 || d.sparse_streams > 16 || d.dense_streams > 16 || d.host_results > 2) return RMCV_ERR_BAD_ARG;
    if (d.pixel_streams > d.depth) d.pixel_streams = d.depth;
    if (d.sparse_streams > d.depth) d.sparse_streams = d.depth;
    if (d.dense_streams > d.depth) d.dense_streams = d.depth;
    // (what came back is read from the records' host mirror; fewer than 3 in rotation stall even sparse batches; the 4-wavefront sparse
    // kernel is the one that fits beside the wave-specialised pixel kernel)
    // hot_contexts: 0 = derived from the bound geometry (hot_for below), -1 = off, n = exactly n
    const int hot_given = cfg ? cfg->hot_contexts : 0;
    if (hot_given > 0 && (hot_given < 3 || hot_given >= d.depth)) d.hot_contexts = -1;
    if (d.depth < 4 || d.host_results != 1 || d.sparse_waves != 4) d.hot_contexts = -1;
    rmcv_pipeline* pl = new (std::nothrow) rmcv_pipeline();
    if (!pl) return RMCV_ERR_NOMEM;
    pl->device = device;
    pl->hot_cfg = d.hot_contexts < 0 ? -1 : (hot_given > 0 ? hot_given : 0);
    if (d.hot_contexts < 0) d.hot_contexts = 0;
    pl->cfg = d;
    int rc = RMCV_OK;
    for (int k = 0; k < d.depth && rc == RMCV_OK; k++) {
        rmcv_ctx* c = nullptr;
        rc = rmcv_ctx_create(device, limits, &c);
        if (rc == RMCV_OK) {
            pl->ring.push_back(c);
            rc = rmcv_ctx_set_option(c, RMCV_OPT_SPARSE_WAVES, d.sparse_waves);
            if (rc == RMCV_OK) rc = rmcv_ctx_set_option(c, RMCV_OPT_PIXEL_GROUPS, d.pixel_groups);
            // everything a batch will need is allocated NOW, for every context of the ring: rmcv_pipeline_submit never allocates
            if (rc == RMCV_OK) rc = ctx_prepare_ring(c);
        }
    }
    if (rc != RMCV_OK) {
        rmcv_pipeline_destroy(pl);
        return rc;
    }
    pl->lim = ctx_limits(pl->ring[0]);
    if (pl->cfg.armour_cap <= 0) pl->cfg.armour_cap = 8 * pl->lim.max_frames;
    pl->head_bytes = (((int64_t)pl->lim.max_frames + 3) * 4 + 15) / 16 * 16;
    pl->record_bytes = pl->head_bytes + (int64_t)pl->cfg.armour_cap * (int64_t)sizeof(rmcv_armour);
    hipError_t e = hipSetDevice(device);
    int lo = 0, hi = 0;
    if (e == hipSuccess) e = hipDeviceGetStreamPriorityRange(&lo, &hi); // hi = the numerically lowest = the highest priority
    // pixel streams at normal priority, sparse streams above them: the per-frame kernels are latency chains whose workgroups must be
    // placed as soon as their batch's planes are there, ahead of the next batches' streaming workgroups
    for (int i = 0; i < d.pixel_streams && e == hipSuccess; i++) {
        hipStream_t s = nullptr;
        e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, 0);
        pl->pix.push_back(s);
    }
    for (int i = 0; i < d.sparse_streams && e == hipSuccess; i++) {
        hipStream_t s = nullptr;
        e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi);
        pl->sp.push_back(s);
    }
    // Events: HIP's default (a system-scope release when the event fires).  hipEventDisableSystemFence / hipEventReleaseToDevice for the
    // device-only events ev_bin / ev_done measured the same as the default (round 4, alternating pipelines of one process against a
    // calibration pair: 1.049-1.063 against 1.051-1.061 for two identical pipelines), so nothing non-default is asked for.
    const unsigned dev_flags = hipEventDisableTiming;
    for (int i = 0; i < d.dense_streams && e == hipSuccess; i++) { // normal priority: a dense frame is long work, not a latency chain
        hipStream_t s = nullptr;
        e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, 0);
        pl->dn.push_back(s);
    }
    for (int k = 0; k < d.depth && e == hipSuccess; k++) {
        hipEvent_t a = nullptr, b = nullptr, h = nullptr, sp_ = nullptr;
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sp_, hipEventDisableTiming);
        pl->ev_sp.push_back(sp_);
        hipEvent_t fr_ = nullptr, chg_ = nullptr;
        if (e == hipSuccess) e = hipEventCreateWithFlags(&fr_, hipEventDisableTiming);
        pl->ev_free.push_back(fr_);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&chg_, hipEventDisableTiming);
        pl->ev_chg.push_back(chg_);
        uint8_t *dr = nullptr, *hr = nullptr;
        e = hipEventCreateWithFlags(&a, dev_flags);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&b, dev_flags); // (the host reads the record's mirror behind ev_done: it must stay a system-scope event)
        if (e == hipSuccess) e = hipEventCreateWithFlags(&h, hipEventDisableTiming);
        if (e == hipSuccess) e = hipMalloc((void**)&dr, (size_t)pl->record_bytes);
        if (e == hipSuccess) e = hipMemset(dr, 0, (size_t)pl->record_bytes);
        uint8_t* hdr = nullptr;
        if (e == hipSuccess && d.host_results == 1) {
            e = hipHostMalloc((void**)&hr, (size_t)pl->record_bytes, hipHostMallocMapped);
            if (e == hipSuccess) memset(hr, 0, (size_t)pl->record_bytes);
            if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&hdr, hr, 0);
        }
        pl->hd_rec.push_back(hdr);
        pl->ev_bin.push_back(a);
        pl->ev_done.push_back(b);
        pl->ev_host.push_back(h);
        pl->d_rec.push_back(dr);
        pl->h_rec.push_back(hr);
        pl->ev_hook.push_back(nullptr);
        pl->slot_ticket.push_back(0);
        pl->slot_frames.push_back(0);
        pl->slot_stream.push_back(nullptr);
        if (e == hipSuccess) ctx_external_order(pl->ring[(size_t)k], b);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&pl->ev_gather, hipEventDisableTiming);
    pl->ctx_last.assign((size_t)d.depth, -1);
    pl->slot_lean.assign((size_t)d.depth, 0);
    pl->slot_ctx.assign((size_t)d.depth, 0);
    pl->hot = hot_for(pl, pl->lim.max_frames, pl->lim.max_width, pl->lim.max_height); // (derived again for the geometry of every submit)
    pl->wait_timeout_ms = ctx_wait_timeout_ms(pl->ring[0]);
#ifdef RMCV_DEV_KNOBS // A/B switches of the schedule's parts (make EXTRA=-DRMCV_DEV_KNOBS): not in a product build
    pl->lazy_back = !(getenv("RMCV_LAZY_BACK") && atoi(getenv("RMCV_LAZY_BACK")) == 0);
    pl->chain_cold = !(getenv("RMCV_CHAIN_COLD") && atoi(getenv("RMCV_CHAIN_COLD")) == 0);
    if (getenv("RMCV_CHAIN_COLD")) pl->chain_cold_us = atoi(getenv("RMCV_CHAIN_COLD")) > 1 ? atoi(getenv("RMCV_CHAIN_COLD")) : 0;
    pl->early_free = !(getenv("RMCV_EARLY_FREE") && atoi(getenv("RMCV_EARLY_FREE")) == 0);
    if (getenv("RMCV_HOT_IDENTITY")) pl->hot_identity = atoi(getenv("RMCV_HOT_IDENTITY")) != 0;
    if (getenv("RMCV_WS_ALWAYS")) pl->ws_always = atoi(getenv("RMCV_WS_ALWAYS")) != 0;
    if (getenv("RMCV_HEAVY_PG")) pl->heavy_pixel_groups = atoi(getenv("RMCV_HEAVY_PG")) > 0 ? atoi(getenv("RMCV_HEAVY_PG")) : 0;
    if (getenv("RMCV_HEAVY_OFF") && atoi(getenv("RMCV_HEAVY_OFF"))) pl->heavy_pixel_groups = -1; // (dense mode off)
#endif
    pl->blocking_base = ring_blocking(pl);
    if (e != hipSuccess) {
        fprintf(stderr, "rmcv_pipeline_create: %s\n", hipGetErrorString(e));
        (void)hipGetLastError();
        rmcv_pipeline_destroy(pl);
        return e == hipErrorOutOfMemory ? RMCV_ERR_NOMEM : RMCV_ERR_HIP;
    }
    *out = pl;
    return RMCV_OK;
}

int rmcv_hw_queues_hint(void)
{
    setenv("GPU_MAX_HW_QUEUES", "12", 0);
    const char* q = getenv("GPU_MAX_HW_QUEUES");
    return q ? atoi(q) : 0;
}

const char* rmcv_pipeline_last_error(const rmcv_pipeline* pl) { return pl ? pl->err : "null pipeline"; }

int rmcv_pipeline_get_info(const rmcv_pipeline* pl, rmcv_pipeline_info* o)
{
    if (!pl || !o) return RMCV_ERR_BAD_ARG;
    memset(o, 0, sizeof(*o));
    o->depth = pl->cfg.depth;
    o->pixel_streams = pl->cfg.pixel_streams;
    o->sparse_streams = pl->cfg.sparse_streams;
    o->armour_cap = pl->cfg.armour_cap;
    o->sparse_waves = pl->cfg.sparse_waves;
    o->pixel_groups = pl->cfg.pixel_groups;
    o->host_results = pl->cfg.host_results;
    o->dense_streams = pl->cfg.dense_streams;
    o->max_frames = pl->lim.max_frames;
    const char* q = getenv("GPU_MAX_HW_QUEUES");
    o->hw_queues_env = q ? atoi(q) : 0;
    o->hw_queues_wanted = 1 + pl->cfg.pixel_streams + pl->cfg.sparse_streams + pl->cfg.dense_streams + (pl->comm ? 1 : 0);
    o->record_bytes = pl->record_bytes;
    o->armours_offset = pl->head_bytes;
    o->submitted = pl->next_ticket;
    o->collected = pl->collected;
    o->dense_split = pl->split_batches;
    o->hot_batches = pl->hot_batches;
    o->hot_contexts = pl->hot;
    o->latency_batches = pl->latency_batches;
    o->host_blocking_calls = pl->own_blocking;
    o->max_submit_us = pl->max_submit_us;
    o->wait_timeout_ms = pl->wait_timeout_ms;
    o->held_back = pl->held_back;
    o->heavy_batches = pl->heavy_batches;
    return RMCV_OK;
}

rmcv_ctx* rmcv_pipeline_context(rmcv_pipeline* pl, int slot)
{
    if (!pl || slot < 0 || slot >= pl->cfg.depth) return nullptr;
    return pl->ring[(size_t)slot];
}

int rmcv_pipeline_set_hot_contexts(rmcv_pipeline* pl, int n)
{
    if (!pl) return RMCV_ERR_BAD_ARG;
    if (n <= 0) { pl->hot = 0; pl->hot_cfg = -1; return RMCV_OK; }
    if (n < 3 || n >= pl->cfg.depth) return pfail(pl, RMCV_ERR_BAD_ARG, "hot_contexts: 3 .. depth - 1, or 0 / -1 for off");
    if (pl->cfg.host_results != 1 || pl->cfg.sparse_waves != 4) return pfail(pl, RMCV_ERR_BAD_ARG, "hot_contexts needs host_results = 1 and sparse_waves = 4");
    pl->hot = pl->hot_cfg = n;
    return RMCV_OK;
}

int rmcv_pipeline_set_wait_timeout(rmcv_pipeline* pl, int ms)
{
    if (!pl || ms < 0) return RMCV_ERR_BAD_ARG;
    pl->wait_timeout_ms = ms;
    for (auto c : pl->ring) rmcv_ctx_set_option(c, RMCV_OPT_WAIT_TIMEOUT_MS, ms);
    return RMCV_OK;
}

// How many contexts take turns while the batches are calm: as many as keep the bit planes of the batches in flight inside the 256 MB
// Infinity Cache (with the planes resident their writes never reach HBM; one context too many and every plane line is a
// miss).  Budget 200 MB of the 256 (frames and byte image stream past it with the nt hint; the sparse kernels' tables want the rest):
// 256 x 1280x1024 -> 46 MB per batch -> 4 (the measured optimum: 4 < 5 << 3, 6); 256 x 1920x1200 -> 79 MB -> 2, which is below the
// three a context's reuse needs as slack -> 3.
static int hot_for(const rmcv_pipeline* pl, int n_frames, int w, int h)
{
    if (pl->hot_cfg != 0) return pl->hot_cfg > 0 ? pl->hot_cfg : 0;
    const int64_t plane = (int64_t)n_frames * (h + 2) * ((w + 63) / 64 + 2) * 8;
    int n = (int)((200ll << 20) / (plane > 0 ? plane : 1));
    if (n < 3) n = 3;
    if (n > pl->cfg.depth - 1) n = pl->cfg.depth - 1;
    return n;
}

static int slot_of(rmcv_pipeline* pl, uint64_t ticket);
rmcv_ctx* rmcv_pipeline_context_of(rmcv_pipeline* pl, uint64_t ticket)
{
    if (!pl) return nullptr;
    if (pl->pend.valid) { hipSetDevice(pl->device); if (finish_back(pl, pl->pend.t == ticket)) return nullptr; }
    const int k = slot_of(pl, ticket);
    if (k < 0) { pfail(pl, RMCV_ERR_BAD_ARG, "no such ticket in flight (never issued, or its slot has been reused)"); return nullptr; }
    // (with the hot contexts a context is reused as early as ticket + hot_contexts, while the ticket's RECORD lives until ticket + depth)
    if (pl->ctx_last[(size_t)pl->slot_ctx[(size_t)k]] != k) { pfail(pl, RMCV_ERR_BAD_ARG, "the ticket's context has been reused by a later batch (its record is still there: rmcv_pipeline_collect)"); return nullptr; }
    return pl->ring[(size_t)pl->slot_ctx[(size_t)k]];
}

int rmcv_pipeline_set_hook(rmcv_pipeline* pl, rmcv_pipeline_hook fn, void* user)
{
    if (!pl) return RMCV_ERR_BAD_ARG;
    if (pl->pend.valid) { hipSetDevice(pl->device); const int rcb = finish_back(pl, false); if (rcb) return rcb; } // (the batch in hand keeps the hook it was submitted under)
    if (pl->comm && fn) return pfail(pl, RMCV_ERR_BAD_ARG, "the pipeline already gathers with rmcv_gather (rmcv_pipeline_set_gather): one hook at a time");
    pl->hook = fn;
    pl->hook_user = user;
    return RMCV_OK;
}

int rmcv_pipeline_set_gather(rmcv_pipeline* pl, rmcv_comm* comm, int root)
{
    if (pl && pl->pend.valid) { hipSetDevice(pl->device); const int rcb = finish_back(pl, false); if (rcb) return rcb; }
    if (!pl) return RMCV_ERR_BAD_ARG;
    if (pl->hook && comm) return pfail(pl, RMCV_ERR_BAD_ARG, "the pipeline already has a hook (rmcv_pipeline_set_hook): one at a time");
    int rc = rmcv_pipeline_drain(pl);
    if (rc) return rc;
    hipSetDevice(pl->device);
    for (auto& p : pl->d_recv) if (p) { hipFree(p); p = nullptr; }
    pl->d_recv.clear();
    pl->comm = nullptr;
    pl->gather_pending = false;
    if (!comm) return RMCV_OK;
    int32_t n = 0, r = 0;
    rc = rmcv_comm_info(comm, &n, &r);
    if (rc) return pfail(pl, rc, "rmcv_comm_info");
    if (root < 0 || root >= n) return pfail(pl, RMCV_ERR_BAD_ARG, "root out of range");
    if (r == root)
        for (int k = 0; k < pl->cfg.depth; k++) {
            uint8_t* p = nullptr;
            PCHK(pl, hipMalloc((void**)&p, (size_t)pl->record_bytes * (size_t)n), "hipMalloc (gather receive buffer)");
            pl->d_recv.push_back(p);
        }
    pl->comm = comm;
    pl->root = root;
    pl->n_ranks = n;
    pl->rank = r;
    return RMCV_OK;
}

// the back half of the newest batch (see rmcv_pipeline::Pending); latency: nothing will be launched beside it
static int finish_back(rmcv_pipeline* pl, bool latency)
{
    if (!pl->pend.valid) return RMCV_OK;
    pl->pend.valid = false;
    const uint64_t t = pl->pend.t;
    const size_t k = pl->pend.k;
    rmcv_ctx* c = pl->pend.c;
    const rmcv_params* p = &pl->pend.p;
    const rmcv_legacy_params* lp = pl->pend.has_lp ? &pl->pend.lp : nullptr;
    const bool used = pl->pend.used;
    const int sparse = pl->pend.sparse, n_frames = pl->pend.n_frames;
    hipStream_t B = pl->pend.B;
    int rc = RMCV_OK;
    const bool heavy = pl->pend.heavy;
    const bool w8 = latency && pl->cfg.sparse_waves == 4 && !lp && !heavy;
    if (w8) {
        rmcv_ctx_set_option(c, RMCV_OPT_SPARSE_WAVES, 8);
        pl->latency_batches++;
    }
    struct Restore { rmcv_ctx* c; bool on; ~Restore() { if (on) rmcv_ctx_set_option(c, RMCV_OPT_SPARSE_WAVES, 4); } } restore{c, w8};
    // Dense frames (beyond findContours' LDS tables: hundreds of borders, 0.5-1 ms on one workgroup) are left by the per-frame launch
    // to a second launch with 8 wavefronts per frame on a stream of its own, the compaction behind it: the sparse stream B is free
    // for the next batch when the batch's ordinary frames are through (one lit window per batch used to cost the whole loop 20-35 %).
    // When: while the batch that last left this slot had SOME such frames but not many (its count sits in the record's host mirror:
    // a camera's lit window stays for many batches).  A batch without any pays nothing (the second launch costs the plain stream
    // 1-3 %: 256 workgroups of 8 wavefronts and 80 KB of LDS to be placed just to find their frame is not marked); a batch full of
    // them is better off with every frame finished where it is (measured: 0.312 against 0.360 ms per step at 233 dense frames of 256).
    if (used && !pl->dn.empty() && hipEventQuery(pl->ev_done[k]) == hipSuccess) {
        const int32_t dense = reinterpret_cast<const int32_t*>(pl->h_rec[k])[pl->lim.max_frames + 2] & 0xFFFFF;
        pl->split_now = dense > 0 && dense * 8 <= pl->slot_frames[k];
    }
    (void)hipGetLastError(); // (hipErrorNotReady is not an error)
    const bool split = !w8 && !heavy && pl->split_now && !pl->dn.empty() && !lp && (sparse & RMCV_STAGE_CONTOURS) && (sparse & RMCV_STAGE_BLOBS);
    if (split) pl->split_batches++;
    hipStream_t T = split ? pl->dn[k % pl->dn.size()] : B; // the stream the batch's list is finished on
    // (a record's rewrite is ordered behind its readers by stream order: the slot meets the same stream every time -- unless the
    // caller mixes stage masks that finish on different streams, or a stream's dense frames come and go)
    // ... the new stream waits, on the GPU, for the tail of the old one
    if (used && pl->slot_stream[k] && pl->slot_stream[k] != T) {
        PCHK(pl, hipEventRecord(pl->ev_chg[k], pl->slot_stream[k]), "pipeline: change of the slot's stream (mark)");
        PCHK(pl, hipStreamWaitEvent(T, pl->ev_chg[k], 0), "pipeline: change of the slot's stream (wait)");
    }
    if (sparse) {
        if (split) {
            ctx_defer_phase(c, 2); // the first launch only: frames beyond the LDS tables are marked and left alone
            rc = rmcv_batch_run(c, p, sparse & ~RMCV_STAGE_POSE, B);
            if (rc == RMCV_OK) {
                PCHK(pl, hipEventRecord(pl->ev_sp[k], B), "pipeline: mark the first sparse launch");
                PCHK(pl, hipStreamWaitEvent(T, pl->ev_sp[k], 0), "pipeline: chain the dense frames");
                ctx_defer_phase(c, 3); // the second launch only (+ the pose stage, which needs every frame's armours)
                rc = rmcv_batch_run(c, p, sparse, T);
            }
            ctx_defer_phase(c, 0);
        } else {
            if (heavy) ctx_sparse_lean(c, 1); // (the launcher takes the lean build where it applies: fused stages, no classifier, the mid tier's scratch there)
            rc = lp ? rmcv_batch_run_legacy(c, p, lp, sparse, B) : rmcv_batch_run(c, p, sparse, B);
            if (heavy) ctx_sparse_lean(c, 0);
        }
        if (rc) return cfail(pl, c, rc);
    }
    PCHK(pl, hipEventRecord(pl->ev_free[k], T), "pipeline: mark the pixel outputs' last reader");
    // the record is rewritten: a reader on another stream (the hook's) must be through; readers on B are by stream order
    if (pl->ev_hook[k]) {
        PCHK(pl, hipStreamWaitEvent(T, (hipEvent_t)pl->ev_hook[k], 0), "pipeline: wait for the record's reader");
        pl->ev_hook[k] = nullptr;
    }
    int32_t* offs = reinterpret_cast<int32_t*>(pl->d_rec[k]);
    // host_results: the compaction kernel stores the record a second time, straight into the slot's pinned host mirror (posted
    // writes over PCIe, only the armours there are); the slot's event -- a default event: system-scope release -- makes them visible
    rc = ctx_compact(c, pl->d_rec[k] + pl->head_bytes, pl->cfg.armour_cap, offs, offs + pl->lim.max_frames + 1, T, pl->hd_rec[k], (int)pl->head_bytes);
    if (rc) return cfail(pl, c, rc);
    // the context's buffers are free from here on: the next pixel kernel of this slot does not wait for the hook
    PCHK(pl, hipEventRecord(pl->ev_done[k], T), "pipeline: mark the slot");
    pl->slot_ticket[k] = t + 1;
    pl->slot_frames[k] = n_frames;
    pl->slot_lean[k] = heavy ? 1 : 0;
    pl->slot_stream[k] = T;
    if (pl->comm) {
        // one communicator: its operations must execute in one order on every rank; they are issued in ticket order on alternating
        // streams, so each gather first waits (an event, on the GPU) for the one before
        if (pl->gather_pending) PCHK(pl, hipStreamWaitEvent(T, pl->ev_gather, 0), "pipeline: order the gathers");
        rc = rmcv_gather(pl->comm, pl->d_rec[k], pl->record_bytes, pl->rank == pl->root ? pl->d_recv[k] : nullptr, pl->root, T);
        if (rc) return pfail(pl, rc, rmcv_comm_last_error(pl->comm));
        PCHK(pl, hipEventRecord(pl->ev_gather, T), "pipeline: mark the gather");
        pl->gather_pending = true;
        PCHK(pl, hipEventRecord(pl->ev_host[k], T), "pipeline: mark the gather"); // wait / collect cover the gather too
    } else if (pl->hook) {
        void* done = nullptr;
        rc = pl->hook(pl->hook_user, t, pl->d_rec[k], pl->record_bytes, T, &done);
        if (rc) return pfail(pl, rc, "the pipeline hook failed");
        pl->ev_hook[k] = done;
    }
    return RMCV_OK;
}

static int submit(rmcv_pipeline* pl, const void* d_frames, int n_frames, int w, int h, int stride, int64_t frame_pitch, const rmcv_params* p,
                  const rmcv_legacy_params* lp, int stages, uint64_t* ticket)
{
    if (!pl || !d_frames || !p) return RMCV_ERR_BAD_ARG;
    if (!(stages & RMCV_STAGE_BINARY)) return pfail(pl, RMCV_ERR_BAD_ARG, "a pipelined batch starts at RMCV_STAGE_BINARY");
    hipSetDevice(pl->device);
    { const int rcb = finish_back(pl, false); if (rcb) return rcb; } // the batch before this one: a pixel launch follows it
    const uint64_t t = pl->next_ticket;
    const size_t k = (size_t)(t % (uint64_t)pl->cfg.depth);
    const bool used = pl->slot_ticket[k] != 0;
    pl->hot = hot_for(pl, n_frames, w, h);
    if (pl->cfg.host_results == 1) { // the newest record that has come back: did any of its frames go beyond the LDS tables?  how heavy was it?
        for (uint64_t d = 1; d <= (uint64_t)pl->cfg.depth && d <= t; d++) {
            const size_t s_ = (size_t)((t - d) % (uint64_t)pl->cfg.depth);
            if (pl->slot_ticket[s_] != t - d + 1 || !pl->h_rec[s_]) break;
            if (hipEventQuery(pl->ev_done[s_]) != hipSuccess) continue;
            const uint32_t word2 = reinterpret_cast<const uint32_t*>(pl->h_rec[s_])[pl->lim.max_frames + 2];
            const int dense = (int)(word2 & 0xFFFFFu), points = (int)(word2 >> 20) * 16; // frames beyond the LDS tables; border points per frame
            // (decided from the record alone -- a dense-mode record says "every frame on the mid tier" by construction: only its points count)
            pl->heavy = pl->slot_lean[s_] ? points >= 1200 : (dense * 8 > pl->slot_frames[s_] || points >= 1500);
            pl->calm = dense == 0 && !pl->heavy; // (measured once more in round 5: one 0.5 ms frame per batch in the hot contexts, its own launch or not: 0.424 ms per step against 0.27)
            break;
        }
        (void)hipGetLastError(); // (hipErrorNotReady is not an error)
    }
    // (dense mode needs the records on the host, the 4-wavefront kernel and two pixel streams to make up for the halved launches)
    const bool heavy = pl->heavy && pl->heavy_pixel_groups >= 0 && pl->cfg.host_results == 1 && pl->cfg.sparse_waves == 4 && pl->cfg.dense_streams >= 0 && !lp &&
                       !(stages & (RMCV_STAGE_IDENTITY | RMCV_STAGE_POSE)) && (stages & RMCV_STAGE_CONTOURS) && (stages & RMCV_STAGE_BLOBS);
    const bool fast = pl->hot && pl->calm && !lp && !(stages & RMCV_STAGE_POSE) && (pl->hot_identity || !(stages & RMCV_STAGE_IDENTITY));
    const size_t j = fast ? (size_t)(pl->hot_seq % (uint64_t)pl->hot) : k;
    rmcv_ctx* c = pl->ring[j];
    hipStream_t A = pl->pix[(size_t)(t % (uint64_t)pl->cfg.pixel_streams)], B = pl->sp[k % (size_t)pl->cfg.sparse_streams];
    int rc;
    // ---- waits first: stream A is behind everything that still uses the slot and the context when the binding below enqueues on it.
    // (Nothing of the pipeline's own state moves before the batch has been accepted: an error return leaves tickets, rotation and
    // context ownership as they were; the waits already enqueued on A are harmless.)
    // the slot's context buffers are free once its previous list is compacted
    if (used) PCHK(pl, hipStreamWaitEvent(A, pl->ev_done[k], 0), "pipeline: wait for the slot");
    // The context's last batch (another slot's, when the hot contexts take turns): the pixel kernel rewrites byte image, bit plane and row
    // masks, whose last reader is that batch's sparse stage -- its compaction reads the armour slots only, and those are rewritten by THIS
    // batch's sparse stage, which follows the compaction in stream order when both run on the same sparse stream.  So the pixel kernel
    // waits for ev_free (behind the sparse stage), not ev_done (behind the compaction: 35-95 us later beside the streaming kernels --
    // with four contexts in rotation the whole slack is ~80 us).
    if (pl->ctx_last[j] >= 0 && pl->ctx_last[j] != (int)k) {
        const size_t last = (size_t)pl->ctx_last[j];
        const bool early = pl->early_free && pl->slot_stream[last] == B;
        PCHK(pl, hipStreamWaitEvent(A, early ? pl->ev_free[last] : pl->ev_done[last], 0), "pipeline: wait for the context");
    }
    // ---- bind: a new geometry's work (planes zeroed, frame order) is ENQUEUED on A, nothing blocks
    rc = ctx_bind_frames(c, d_frames, n_frames, w, h, stride, frame_pitch, A);
    if (rc) return cfail(pl, c, rc);
    // a batch is several runs on several streams: everything that could refuse it is checked before the first launch
    if ((rc = ctx_check_stages(c, p, stages))) return cfail(pl, c, rc);
    const int pixel = stages & (RMCV_STAGE_BINARY | RMCV_STAGE_NO_IMAGE), sparse = stages & ~(RMCV_STAGE_BINARY | RMCV_STAGE_NO_IMAGE);
    ctx_external_order(c, pl->ev_done[k]);
    ctx_pixel_shape(c, (fast || pl->ws_always) ? 1 : 0);
    // A burst's SECOND pixel launch is held back (k_delay on its stream).  k_binary_ws is one workgroup per CU: when two launches
    // reach an empty machine 15 us apart, whether the first has taken every CU by then is a coin toss -- if not, the two split the
    // CUs, run side by side and END together, and so do the next pairs (each pair's ramp and tail in the open, both sparse kernels
    // at once) until they drift apart: 0.258 instead of 0.242 ms per step over a 20-batch burst, in 15 % of the bursts
    // (tools/trace_regions.py, profiles/r04k_burst_start.txt).  Held back, the second launch finds every CU taken and its workgroups
    // move in as the first one's leave -- the steady state -- at no cost: they would have waited anyway.  (Waiting for the first
    // launch's END instead puts the event's latency between the two: +1-3 %.)  Round 5: only where that reason exists -- the launch
    // WILL be k_binary_ws on every CU (launch_binary's own rule: pixel_ws_full) -- and for a quarter of the launch's expected time
    // (its bytes at 5.5 TB/s), 60 us at most, nothing below 100 us of launch: two 16-frame batches are not held back at all.
    bool cold = false;
    if (fast && pl->chain_cold) {
        cold = t == 0;
        const size_t s_ = t ? (size_t)((t - 1) % (uint64_t)pl->cfg.depth) : 0;
        if (t > 0) {
            cold = pl->slot_ticket[s_] == t && hipEventQuery(pl->ev_done[s_]) == hipSuccess;
            (void)hipGetLastError();
        }
        if (!cold && pl->was_cold && pl->slot_ticket[s_] == t && pixel_ws_full(c, p->lower_bound)) {
            const double launch_us = (double)n_frames * 4.0 * w * h / 5.5e6;
            const int hold_us = launch_us < 100.0 ? 0 : (int)(launch_us / 4.0 < pl->chain_cold_us ? launch_us / 4.0 : pl->chain_cold_us);
            if (pl->chain_cold_us > 0) { // hold the second launch back until the first one's workgroups have taken every CU
                if (hold_us > 0) {
                    PCHK(pl, launch_delay((unsigned long long)hold_us * 1000ull, A), "pipeline: k_delay");
                    pl->held_back++;
                }
            } else
                PCHK(pl, hipStreamWaitEvent(A, pl->ev_bin[s_], 0), "pipeline: chain a burst's second launch");
        }
    }
    if (heavy && pl->heavy_pixel_groups > 0) rmcv_ctx_set_option(c, RMCV_OPT_PIXEL_GROUPS, pl->heavy_pixel_groups);
    rc = rmcv_batch_run(c, p, pixel, A);
    if (heavy && pl->heavy_pixel_groups > 0) rmcv_ctx_set_option(c, RMCV_OPT_PIXEL_GROUPS, pl->cfg.pixel_groups);
    if (rc) return cfail(pl, c, rc);
    pl->last_what = "the pixel kernel (k_binary / k_binary_ws)";
    // ---- accepted: the pipeline's state moves
    pl->was_cold = cold;
    if (fast) { pl->hot_seq++; pl->hot_batches++; }
    if (heavy) pl->heavy_batches++;
    pl->ctx_last[j] = (int)k;
    pl->slot_ctx[k] = (int)j;
    PCHK(pl, hipEventRecord(pl->ev_bin[k], A), "pipeline: mark the pixel kernel");
    PCHK(pl, hipStreamWaitEvent(B, pl->ev_bin[k], 0), "pipeline: chain the sparse stages");
    pl->next_ticket = t + 1;
    if (ticket) *ticket = t;
    pl->pend.valid = true;
    pl->pend.t = t;
    pl->pend.k = k;
    pl->pend.c = c;
    pl->pend.p = *p;
    pl->pend.has_lp = lp != nullptr;
    if (lp) pl->pend.lp = *lp;
    pl->pend.used = used;
    pl->pend.heavy = heavy;
    pl->pend.sparse = sparse;
    pl->pend.n_frames = n_frames;
    pl->pend.B = B;
    // (a hook or the gather hands the record to a consumer the pipeline does not see waiting: its batches are finished here and now)
    if (!pl->lazy_back || pl->hook || pl->comm) return finish_back(pl, false);
    return RMCV_OK;
}

// submit + its own bookkeeping: the host time of the call, and the blocking calls the ring's contexts counted during it
static int submit_counted(rmcv_pipeline* pl, const void* d_frames, int n_frames, int w, int h, int stride, int64_t frame_pitch, const rmcv_params* p,
                          const rmcv_legacy_params* lp, int stages, uint64_t* ticket)
{
    if (!pl) return RMCV_ERR_BAD_ARG;
    timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    const uint64_t b0 = ring_blocking(pl);
    const int rc = submit(pl, d_frames, n_frames, w, h, stride, frame_pitch, p, lp, stages, ticket);
    pl->own_blocking += ring_blocking(pl) - b0;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double us = (t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3;
    if (us > pl->max_submit_us) pl->max_submit_us = us;
    return rc;
}

int rmcv_pipeline_reset_stats(rmcv_pipeline* pl)
{
    if (!pl) return RMCV_ERR_BAD_ARG;
    pl->max_submit_us = 0;
    return RMCV_OK;
}

int rmcv_pipeline_submit(rmcv_pipeline* pl, const void* d_frames, int n_frames, int w, int h, int stride, int64_t frame_pitch,
                         const rmcv_params* p, int stages, uint64_t* ticket)
{
    return submit_counted(pl, d_frames, n_frames, w, h, stride, frame_pitch, p, nullptr, stages, ticket);
}

int rmcv_pipeline_submit_legacy(rmcv_pipeline* pl, const void* d_frames, int n_frames, int w, int h, int stride, int64_t frame_pitch,
                                const rmcv_params* p, const rmcv_legacy_params* lp, int stages, uint64_t* ticket)
{
    if (!lp) return RMCV_ERR_BAD_ARG;
    return submit_counted(pl, d_frames, n_frames, w, h, stride, frame_pitch, p, lp, stages, ticket);
}

// slot of a live ticket, or -1
static int slot_of(rmcv_pipeline* pl, uint64_t ticket)
{
    if (ticket >= pl->next_ticket) return -1;
    const size_t k = (size_t)(ticket % (uint64_t)pl->cfg.depth);
    return pl->slot_ticket[k] == ticket + 1 ? (int)k : -1;
}

int rmcv_pipeline_wait(rmcv_pipeline* pl, uint64_t ticket)
{
    if (!pl) return RMCV_ERR_BAD_ARG;
    if (pl->pend.valid) { // (the newest batch's back half: with the latency kernel if it is the one waited for)
        hipSetDevice(pl->device);
        const int rcb = finish_back(pl, pl->pend.t == ticket);
        if (rcb) return rcb;
    }
    const int k = slot_of(pl, ticket);
    if (k < 0) return pfail(pl, RMCV_ERR_BAD_ARG, "no such ticket in flight (never issued, or its slot has been reused)");
    hipSetDevice(pl->device);
    int rcw = pwait_event(pl, pl->ev_done[(size_t)k], "rmcv_pipeline_wait");
    if (rcw) return rcw;
    if (pl->comm && (rcw = pwait_event(pl, pl->ev_host[(size_t)k], "rmcv_pipeline_wait (gather)"))) return rcw;
    return RMCV_OK;
}

int rmcv_pipeline_collect(rmcv_pipeline* pl, uint64_t ticket, rmcv_armour* armours_out, int cap, int32_t* frame_offs, int32_t* n_total)
{
    if (!pl || cap < 0) return RMCV_ERR_BAD_ARG;
    int rc = rmcv_pipeline_wait(pl, ticket);
    if (rc) return rc;
    const size_t k = (size_t)slot_of(pl, ticket);
    const int nf = pl->slot_frames[k];
    std::vector<uint8_t> tmp;
    const uint8_t* rec = pl->h_rec[k];
    if (pl->cfg.host_results != 1) { // lists stay on the device until asked for: the head first, then exactly the armours there are
        tmp.resize((size_t)pl->head_bytes);
        PCHK(pl, hipMemcpy(tmp.data(), pl->d_rec[k], (size_t)pl->head_bytes, hipMemcpyDeviceToHost), "pipeline: D2H head");
        rec = tmp.data();
    }
    const int32_t* offs = reinterpret_cast<const int32_t*>(rec);
    const int32_t total = offs[nf], st = offs[pl->lim.max_frames + 1];
    if (n_total) *n_total = total;
    if (frame_offs) memcpy(frame_offs, offs, (size_t)(nf + 1) * 4);
    pl->collected++;
    if (st & (RMCV_FRAME_OVF_CONTOURS | RMCV_FRAME_OVF_POINTS | RMCV_FRAME_OVF_BLOBS | RMCV_FRAME_OVF_ARMOURS))
        return pfail(pl, RMCV_ERR_CAPACITY, "context limits exceeded on at least one frame of the batch (rmcv_batch_counts on the slot's context names it)");
    if (total > pl->cfg.armour_cap) return pfail(pl, RMCV_ERR_CAPACITY, "the batch has more armours than the pipeline's armour_cap");
    if (total > cap) return pfail(pl, RMCV_ERR_CAPACITY, "output capacity exceeded");
    if (armours_out && total) {
        if (pl->cfg.host_results == 1) memcpy(armours_out, rec + pl->head_bytes, (size_t)total * sizeof(rmcv_armour));
        else PCHK(pl, hipMemcpy(armours_out, pl->d_rec[k] + pl->head_bytes, (size_t)total * sizeof(rmcv_armour), hipMemcpyDeviceToHost), "pipeline: D2H armours");
    }
    return RMCV_OK;
}

int rmcv_pipeline_drain(rmcv_pipeline* pl)
{
    if (!pl) return RMCV_ERR_BAD_ARG;
    hipSetDevice(pl->device);
    { const int rcb = finish_back(pl, true); if (rcb) return rcb; }
    int rcw;
    for (auto s : pl->pix) if ((rcw = pwait_stream(pl, s, "rmcv_pipeline_drain (pixel stream)"))) return rcw;
    for (auto s : pl->sp) if ((rcw = pwait_stream(pl, s, "rmcv_pipeline_drain (sparse stream)"))) return rcw;
    for (auto s : pl->dn) if ((rcw = pwait_stream(pl, s, "rmcv_pipeline_drain (dense stream)"))) return rcw;
    for (size_t k = 0; k < pl->ev_hook.size(); k++)
        if (pl->ev_hook[k]) {
            if ((rcw = pwait_event(pl, (hipEvent_t)pl->ev_hook[k], "rmcv_pipeline_drain (hook)"))) return rcw;
            pl->ev_hook[k] = nullptr;
        }
    return RMCV_OK;
}

int rmcv_pipeline_record(rmcv_pipeline* pl, uint64_t ticket, void** d_record, void** hip_stream)
{
    if (!pl) return RMCV_ERR_BAD_ARG;
    if (pl->pend.valid) { hipSetDevice(pl->device); const int rcb = finish_back(pl, pl->pend.t == ticket); if (rcb) return rcb; }
    const int k = slot_of(pl, ticket);
    if (k < 0) return pfail(pl, RMCV_ERR_BAD_ARG, "no such ticket in flight (never issued, or its slot has been reused)");
    if (d_record) *d_record = pl->d_rec[(size_t)k];
    if (hip_stream) *hip_stream = pl->slot_stream[(size_t)k];
    return RMCV_OK;
}

int rmcv_pipeline_gathered(rmcv_pipeline* pl, uint64_t ticket, void** d_recv, int64_t* bytes)
{
    if (pl && pl->pend.valid) { hipSetDevice(pl->device); const int rcb = finish_back(pl, pl->pend.t == ticket); if (rcb) return rcb; }
    if (!pl || !pl->comm) return RMCV_ERR_BAD_ARG;
    const int k = slot_of(pl, ticket);
    if (k < 0) return pfail(pl, RMCV_ERR_BAD_ARG, "no such ticket in flight (never issued, or its slot has been reused)");
    if (d_recv) *d_recv = pl->rank == pl->root ? pl->d_recv[(size_t)k] : nullptr;
    if (bytes) *bytes = pl->record_bytes * pl->n_ranks;
    return RMCV_OK;
}

// ---- device memory for hosts without HIP headers ----
int rmcv_device_alloc(int device, int64_t bytes, void** d_ptr)
{
    if (!d_ptr || bytes <= 0) return RMCV_ERR_BAD_ARG;
    *d_ptr = nullptr;
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return RMCV_ERR_NO_DEVICE; }
    const hipError_t e = hipMalloc(d_ptr, (size_t)bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); return e == hipErrorOutOfMemory ? RMCV_ERR_NOMEM : RMCV_ERR_HIP; }
    return RMCV_OK;
}

void rmcv_device_free(int device, void* d_ptr)
{
    if (!d_ptr) return;
    if (hipSetDevice(device) == hipSuccess) (void)hipFree(d_ptr);
    (void)hipGetLastError();
}

int rmcv_device_upload(int device, void* d_dst, const void* h_src, int64_t bytes)
{
    if (!d_dst || !h_src || bytes < 0) return RMCV_ERR_BAD_ARG;
    if (hipSetDevice(device) != hipSuccess || hipMemcpy(d_dst, h_src, (size_t)bytes, hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); return RMCV_ERR_HIP; }
    return RMCV_OK;
}

int rmcv_device_download(int device, void* h_dst, const void* d_src, int64_t bytes)
{
    if (!h_dst || !d_src || bytes < 0) return RMCV_ERR_BAD_ARG;
    if (hipSetDevice(device) != hipSuccess || hipMemcpy(h_dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return RMCV_ERR_HIP; }
    return RMCV_OK;
}

} // extern "C"
