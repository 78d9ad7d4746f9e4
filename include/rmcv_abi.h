/*
 * rmcv_abi.h -- C-ABI of the MI355X-native rmcv detection path (librmcv_hip.so).
 *
 * The reference has no plugin/FFI layer: its boundary is the C++ linkage of the
 * static library `rmcv` (/root/reference/CMakeLists.txt:13-16) consumed through
 * include/rmcv.h by executable/main.cpp:172-176.  This header is the thin C-ABI
 * that boundary is re-hosted on: plain pointers and sizes, PODs only, error
 * codes instead of exceptions.  include/rmcv_shim.hpp layers the reference's own
 * rm:: signatures on top of it where OpenCV headers exist.
 *
 * Every entry point runs on the GPU (hand-written HIP, gfx950).  There is no CPU
 * fallback: without a usable device rmcv_ctx_create() fails with
 * RMCV_ERR_NO_DEVICE.
 *
 * Threading (reference: one caller thread, executable/main.cpp:55): a context is
 * single-owner; calls on one context must not overlap.  Input memory is borrowed
 * read-only and must stay valid until the call (or, for the batch API, the
 * matching rmcv_batch_sync) returns.  The library never frees caller memory.
 */
#ifndef RMCV_ABI_H
#define RMCV_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RMCV_ABI_VERSION 1

/* error codes (0 = ok) */
#define RMCV_OK 0
#define RMCV_ERR_BAD_ARG (-1)   /* null pointer, bad size, unsupported value            */
#define RMCV_ERR_CAPACITY (-2)  /* an output did not fit; the counts report what is needed */
#define RMCV_ERR_NOMEM (-3)
#define RMCV_ERR_HIP (-4)       /* a HIP runtime call failed: see rmcv_last_error        */
#define RMCV_ERR_NO_DEVICE (-5) /* no gfx950 device: this library has no CPU path        */
#define RMCV_ERR_RCCL (-6)      /* librccl could not be loaded, or an RCCL call failed: see rmcv_comm_last_error */
#define RMCV_ERR_TIMEOUT (-7)   /* work on the GPU did not finish within the deadline (RMCV_OPT_WAIT_TIMEOUT_MS /
                                 * rmcv_pipeline_set_wait_timeout); the message names the kernel or copy enqueued last */

/* rm::camp -- include/core.h:20-23 */
#define RMCV_CAMP_RED 0
#define RMCV_CAMP_BLUE 1
#define RMCV_CAMP_GUIDELIGHT 2
#define RMCV_CAMP_NEUTRAL (-1)

/* morphology after the threshold.  The snapshot does MORPH_CLOSE (src/imgproc.cpp:68-69);
 * the older ExtractColor API (docs/namespacerm.html:854) did a dilate only, which is what
 * BASELINE.json config 2 names. */
#define RMCV_MORPH_NONE 0
#define RMCV_MORPH_DILATE 1
#define RMCV_MORPH_CLOSE 2

typedef struct { int32_t x, y; } rmcv_point;               /* cv::Point, rm::contour element (core.h:87) */
typedef struct { float cx, cy, w, h, angle; } rmcv_rrect;  /* cv::RotatedRect */

typedef struct {            /* rm::lightblob, include/core.h:89-99 */
    float   angle;          /* vertical = 90 */
    int32_t target;         /* rm::camp */
    float   center[2];
    float   vertices[4][2]; /* left-down, left-up, right-up, right-down (core.cpp:265-283) */
    float   size[2];        /* width = min side, height = max side */
} rmcv_lightblob;           /* 56 bytes */

typedef struct {            /* the per-frame data members of rm::armour, include/core.h:110-112 */
    float   icon[4][2];
    float   vertices[4][2]; /* the bit-exact deliverable */
    float   bbox[4];        /* cv::Rect2f bounding_box: x, y, width, height */
    int32_t blob_i, blob_j; /* indices of the two light blobs in the positive list */
} rmcv_armour;              /* 88 bytes */

typedef struct {            /* the literals of executable/main.cpp:172-176 are the defaults */
    int32_t camp;           /* enemy colour, CAMP_BLUE     */
    int32_t lower_bound;    /* 80                          */
    int32_t morph;          /* RMCV_MORPH_CLOSE            */
    float   tilt_max;       /* 70                          */
    float   ratio_lo, ratio_hi; /* 1.5, 80                 */
    double  area_lo, area_hi;   /* 10, 99999               */
    float   angle_diff_max; /* 12                          */
    float   shear_max;      /* 22                          */
    float   length_ratio_max; /* 0.4                       */
    int32_t _pad;
} rmcv_params;

typedef struct {            /* capacities of one context; 0 = default */
    int32_t max_frames;     /* frames per batch                  (256)   */
    int32_t max_width;      /*                                   (1920)  */
    int32_t max_height;     /*                                   (1200)  */
    int32_t max_contours;   /* contours per frame                (2048)  */
    int32_t max_points;     /* contour points per frame          (65536) */
    int32_t max_blobs;      /* positive light blobs per frame    (256)   */
    int32_t max_armours;    /* armours per frame                 (256)   */
    int32_t _pad;
} rmcv_limits;

/* stages of rmcv_batch_run (bit mask); each stage needs the ones below it */
#define RMCV_STAGE_BINARY 1   /* extract_color up to morphologyEx    src/imgproc.cpp:52-69    */
#define RMCV_STAGE_CONTOURS 2 /* findContours                        src/imgproc.cpp:71-72    */
#define RMCV_STAGE_BLOBS 4    /* filter_lightblobs                   src/objdetect.cpp:55-87  */
#define RMCV_STAGE_ARMOURS 8  /* filter_armours                      src/objdetect.cpp:114-166 */
#define RMCV_STAGE_ALL 15
#define RMCV_STAGE_POSE 32     /* solve_PnP + world position per armour, src/mobility.cpp:166-190, executable/main.cpp:183-192;
                                 needs rmcv_pnp_load */
#define RMCV_STAGE_NO_IMAGE 64 /* modifier of RMCV_STAGE_BINARY: do not write the 0/255 byte image (rmcv_batch_get_binary then returns
                                 stale data).  The reference returns `binary` from extract_color but only its debug view reads
                                 it (executable/main.cpp:200-201); a detection-only deployment saves 1 of the 4 bytes per pixel. */
#define RMCV_STAGE_IDENTITY 16 /* affine_correction + flatten + svm->predict per armour (BASELINE config 5),
                                 src/imgproc.cpp:9-35, src/core.cpp:202-216, executable/main.cpp:180-181; needs rmcv_svm_load */

/* per-frame status bits reported by rmcv_batch_counts */
#define RMCV_FRAME_OVF_CONTOURS 1
#define RMCV_FRAME_OVF_POINTS 2
#define RMCV_FRAME_OVF_BLOBS 4
#define RMCV_FRAME_OVF_ARMOURS 8
#define RMCV_FRAME_SLOW_PATH 16 /* informational: the sequential (literal) scanner was used -- the last resort: a frame beyond the mid tier
                                 * too (> 131072 border visits, wider than 2048 px or taller than the row tables) */
#define RMCV_FRAME_MID_PATH 64  /* informational: findContours of this frame ran on the mid tier (tables in global memory: the frame is
                                 * beyond the LDS tables -- > 4096 border visits, > 1024 non-empty words, > 512 contours -- but was not
                                 * handed to the sequential scanner) */
#define RMCV_FRAME_HULL 32      /* legacy matcher: a contour exceeded the hull tables (dimensions > 4096) or is not a closed border */

typedef struct rmcv_ctx rmcv_ctx;

int  rmcv_abi_version(void);
void rmcv_default_params(rmcv_params* p);
void rmcv_default_limits(rmcv_limits* l);

int  rmcv_ctx_create(int device, const rmcv_limits* limits /* nullable */, rmcv_ctx** out);
void rmcv_ctx_destroy(rmcv_ctx* ctx);
const char* rmcv_last_error(const rmcv_ctx* ctx);
/* tuning knobs of a context.  RMCV_OPT_SPARSE_WAVES: wavefronts per frame of the fused sparse kernel (findContours + fits +
 * pairing) in batch runs -- 8 (default): lowest latency of a lone batch; 4: highest throughput when several batches are in flight
 * on different streams (leaves register-file room on every CU for the pixel kernels of the next batches).  Results are identical. */
#define RMCV_OPT_SPARSE_WAVES 1
/* RMCV_OPT_PIXEL_GROUPS: persistent workgroups per CU of the pixel kernel, 1..8 -- 3 (default): fastest for a lone batch (2 is
 * 5 % slower, 4 within 3 %); 2: leaves wave slots and registers on every CU to the kernels of the other batches in flight
 * (two pixel launches of consecutive batches then overlap, i.e. 4 workgroups per CU are resident).  Results are identical. */
#define RMCV_OPT_PIXEL_GROUPS 2
/* RMCV_OPT_FRAME_UPLOAD: how rmcv_extract_color brings the caller's host frame to the device -- 0: the HIP runtime's
 * pageable copy (the fastest of the two safe ways when all is well: 0.186 ms per frame chain against 0.22); 1: through the context's
 * pinned staging buffer (a CPU copy, then DMA: nothing of it depends on the runtime pinning the caller's pages); 3 (default): 0, and
 * 1 WHILE 0 IS SLOW -- the library times the upload of every frame whose byte image it returns, moves to the staging buffer after three
 * slow frames in a row (the runtime's pageable copies were measured 120-250 us slower each for tens of seconds after a large GPU
 * process had exited: the chain 0.28-0.40 ms instead of 0.18), stays there for 512 frames and tries again; 2: the caller's buffer is pinned in place on first sight (hipHostRegister, kept for the context's
 * lifetime, at most 16 buffers) and read by DMA with no CPU copy (0.35 ms) -- for camera SDKs that hand out a fixed ring of
 * frame buffers (the reference's cameras do, hardware/src/daheng.cpp:83).  A pinning is keyed by ADDRESS (and size): the
 * buffers must stay mapped while the context lives, or be handed to rmcv_ctx_forget_frame_buffer BEFORE they are freed -- memory
 * that is freed and mapped again at the same address would otherwise be read through the stale pinning (the same contract
 * hipHostRegister itself has).  Results are identical. */
#define RMCV_OPT_FRAME_UPLOAD 3
/* RMCV_OPT_RUN_AHEAD: 1 (default): rmcv_extract_color also enqueues the blob and armour stages with the parameters the PREVIOUS
 * frame's rmcv_filter_lightblobs / rmcv_filter_armours calls used; when this frame's calls come with the same parameters and the
 * lists rmcv_extract_color / rmcv_filter_lightblobs returned, they hand over results that are already on the host -- the whole
 * chain of executable/main.cpp:172-176 is then one stream sequence with one synchronisation.  0: every call does its own work.
 * Results are identical. */
#define RMCV_OPT_RUN_AHEAD 4
/* RMCV_OPT_CONTOUR_TIER: which form of findContours a frame takes -- 0 (default): chosen per frame (tables in LDS; beyond their
 * capacity the same formulation with tables in global memory, RMCV_FRAME_MID_PATH; beyond that the sequential scanner,
 * RMCV_FRAME_SLOW_PATH); 1: the sequential scanner for every frame; 2: the mid tier for every frame.  A test / diagnosis knob:
 * results are identical. */
#define RMCV_OPT_CONTOUR_TIER 5
/* (option id 6 was RMCV_OPT_HANDOVER, the frame-level hand-over from the pixel kernel to the sparse kernel of rounds 3-4: correct,
 * tested, and measured equal or slower in every schedule, twice -- removed with its progress words, its write-through stores and
 * the second instantiation of every pixel kernel; HISTORY.md 5b has the design) */
/* RMCV_OPT_DENSE_DEFER: 1: with RMCV_OPT_SPARSE_WAVES = 4, a frame beyond the LDS tables of findContours (hundreds of borders:
 * RMCV_FRAME_MID_PATH) is left to a second launch with 8 wavefronts per frame right behind the first; 0 (default): every frame is
 * finished by the first launch.  Results are identical.  Measured: worth 14 % where EVERY frame is that dense, costs
 * 20-30 % where a few frames per batch are (they then run after the others instead of beside them). */
#define RMCV_OPT_DENSE_DEFER 7
/* (option ids 8-10 were round 3's measurement knobs PIXEL_STAGGER, SPARSE_PRIO, PIXEL_TAPER: every setting measured 1.000 or worse;
 * removed together with their kernel branches -- rmcv_ctx_set_option answers RMCV_ERR_BAD_ARG) */
/* RMCV_OPT_PIXEL_HALO_NT: cache policy of the pixel kernel's loads of the rows a strip shares with its neighbours -- 0 (default):
 * cacheable (the neighbouring strip finds them in L2); 1: non-temporal like every other load.  Which is faster depends on the
 * device the process finds itself on and on the workload (the pixel kernels alone gain 3.6 % with 1 on some boxes and
 * lose 1-5 % on others; the whole path moved by -0.4 % at 1280 px and +6 % at 1920 px on a box of the first kind).  A measurement
 * knob; results are identical. */
#define RMCV_OPT_PIXEL_HALO_NT 11
/* RMCV_OPT_OVERLOADS: which functions the reference's UNQUALIFIED abs / atan2 / sin / cos on floats are (src/objdetect.cpp:24, 79,
 * 131-143, 153, 157; src/core.cpp:335-337) -- that depends on the headers the reference's translation units see:
 * bit 0: abs(float) is int abs(int), the argument truncated towards zero (<cmath> alone under libstdc++); bit 1: atan2 / sin / cos
 * are the double functions, the arithmetic around them double.  0 (default): the float overloads everywhere.  Changes results --
 * by design: it follows the reference build it replaces.  INTEGRATION.md has the probe that tells which value a build needs. */
#define RMCV_OPT_OVERLOADS 13
/* RMCV_OPT_PIXEL_SHAPE: which kernel is the pixel stage of a WHOLE batch whose rows are contiguous (stride == 3 w, w % 64 == 0, more
 * strips than half the CUs, lb > 0; anything else is k_binary's whatever this says).
 * 1 (default): k_binary_ws -- ONE 1024-thread workgroup per CU, 8 wavefronts loading strip k+1 while 8 store strip k: alone 8 % faster
 * than k_binary (0.227 against 0.247 ms per 256 x 1280x1024, 5.9 TB/s of the bare copy's 6.25).  It fills the CU: a second launch
 * of it waits for the first, and the 8-wavefront sparse kernel cannot run beside it.
 * 0: k_binary -- 256-thread workgroups, 2-3 per CU, each loading, thresholding, closing and storing its strip in turn; launches of
 * consecutive batches and the sparse kernels share every CU.  A pipeline chooses per batch (rmcv_pipeline_config::hot_contexts)
 * and overrides this.  Same results. */
#define RMCV_OPT_PIXEL_SHAPE 14
/* RMCV_OPT_WAIT_TIMEOUT_MS: the deadline of every wait a call of this context makes for its work on the GPU, in milliseconds
 * (default 5000; 0: none).  The reference's process loop is real-time and latest-wins (executable/main.cpp:157, 169, 197): a drop-in
 * must not park its caller for good behind a kernel that never finishes.  No entry point calls hipStreamSynchronize /
 * hipEventSynchronize: waits poll the stream (spinning for the first 2 ms -- the runtime's own wait parks the thread after ~0.1 ms,
 * and its wake-up costs the per-frame chain another 0.1 ms --, then yielding, then sleeping 0.2 ms at a time) and give up at the
 * deadline with RMCV_ERR_TIMEOUT; rmcv_last_error then names the kernel or copy enqueued last.  After RMCV_ERR_TIMEOUT the work is
 * STILL IN FLIGHT: the buffers handed to the call (frame, binary_out) stay borrowed until a later call on the context succeeds
 * (every call first waits, with the same deadline, for what is in flight) or the context is destroyed; rmcv_ctx_destroy waits
 * once more and, if the work has still not finished, leaks the context's device memory instead of freeing it under a kernel. */
#define RMCV_OPT_WAIT_TIMEOUT_MS 15
/* RMCV_OPT_IMAGE_EXPORT: how rmcv_extract_color brings the byte image to `binary_out` -- 0: the HIP runtime's pageable
 * device-to-host copy, issued once the pixel kernel has finished (the library polls for that with the deadline first): fastest when all
 * is well (0.186 ms per 1280x1024 chain from a C host), but the copy happens INSIDE the runtime's call and was measured at 160-280 us
 * instead of 35 in some processes (bench.py's C-host child: the chain 0.28-0.40 ms); 1: a kernel on the library's side stream copies
 * the image into pinned host memory chunk by chunk, raising a flag per chunk that the host polls in memory, and the library copies
 * the chunks into `binary_out` as they arrive -- no runtime-internal wait anywhere in the chain: 0.190 ms alone, 0.19-0.20 ms where
 * the runtime's copy is slow; 2 (default): 0, and 1 while 0 is slow (three slow frames in a row -> 512 frames on the library's
 * path, then another try; see RMCV_OPT_FRAME_UPLOAD).  Same bytes. */
#define RMCV_OPT_IMAGE_EXPORT 17
/* RMCV_OPT_TEST_SLOW_US: microseconds added to what the library MEASURES of the runtime's two copies (not to the copies): the switch of
 * RMCV_OPT_FRAME_UPLOAD 3 / RMCV_OPT_IMAGE_EXPORT 2 on demand.  A test hook (tests/test_gpu_deadline.py). */
#define RMCV_OPT_TEST_SLOW_US 18
/* RMCV_OPT_TEST_DELAY_US: the next rmcv_extract_color / rmcv_batch_run of the context first holds its stream back for this many
 * microseconds (one sleeping wavefront): a stand-in for a kernel that does not finish in time.  One shot.  A test hook
 * (tests/test_gpu_deadline.py). */
#define RMCV_OPT_TEST_DELAY_US 16
int  rmcv_ctx_set_option(rmcv_ctx* ctx, int option, int value);
/* launches of k_binary_ws (RMCV_OPT_PIXEL_SHAPE 1) by this process so far: a diagnostic -- an option that is set but whose
 * conditions a batch does not meet falls back to k_binary silently (tests/test_gpu_pixel_shape.py) */
int64_t rmcv_pixel_ws_launches(void);
/* every device buffer of a context lies between two 4 KiB guard zones holding a fixed pattern: count the damaged ones (0 in a
 * correct build; rmcv_last_error names the first).  Synchronises the context.  A test/diagnosis hook (tests/test_gpu_canary.py). */
int  rmcv_ctx_check_guards(rmcv_ctx* ctx, int32_t* n_damaged);
/* where the last rmcv_extract_color spent its time on the HOST, seven figures in microseconds: us[0] waiting for earlier work, binding,
 * enqueuing the upload; us[1] enqueuing the kernels; us[2] until the byte image's first chunk is home (upload + pixel kernel + PCIe);
 * us[3] the other chunks, copied into binary_out as they arrive; us[4] (the runtime's copy where there is no mapped pinned memory);
 * us[5] waiting for the frame's kernels; us[6] handing the lists over.  cap >= 7; with cap >= 9 also us[7] = the upload path the frame
 * took (0 pageable, 1 pinned staging, 2 registered) and us[8] = the image path (0 the runtime's copy, 1 the export kernel).  A diagnosis
 * hook (tools/frame_chain.c prints the medians). */
int  rmcv_ctx_frame_timing(const rmcv_ctx* ctx, double* us, int cap);
/* drop the pinning RMCV_OPT_FRAME_UPLOAD = 2 made for `frame` (NULL: all of them); drains the context's stream first */
int  rmcv_ctx_forget_frame_buffer(rmcv_ctx* ctx, const void* frame);

/* ---- single frame, host buffers: one call per reference function ---------------------- */

/* rm::extract_color (include/imgproc.h:29, src/imgproc.cpp:50-75).  bgr: CV_8UC3, row pitch
 * `stride` bytes.  binary_out (h*w bytes, 0/255) may be NULL.  Contours come back as CSR in
 * cv::findContours(RETR_EXTERNAL, CHAIN_APPROX_NONE) order: offs_out has n_contours+1 entries. */
int rmcv_extract_color(rmcv_ctx* ctx, const uint8_t* bgr, int w, int h, int stride, int camp, int lower_bound,
                       int morph, uint8_t* binary_out, rmcv_point* pts_out, int pts_cap, int32_t* offs_out,
                       int contours_cap, int32_t* n_contours, int32_t* n_points);

/* rm::filter_lightblobs (include/objdetect.h:47-49, src/objdetect.cpp:55-87).  The negative
 * list is returned as contour indices.  blob_src (nullable) = contour index of each positive. */
int rmcv_filter_lightblobs(rmcv_ctx* ctx, const rmcv_point* pts, const int32_t* offs, int n_contours,
                           float tilt_max, float ratio_lo, float ratio_hi, double area_lo, double area_hi,
                           int enemy, rmcv_lightblob* blobs_out, int blobs_cap, int32_t* n_blobs,
                           int32_t* blob_src, int32_t* neg_idx_out, int32_t* n_neg);

/* rm::filter_armours (include/objdetect.h:70-71, src/objdetect.cpp:114-166) */
int rmcv_filter_armours(rmcv_ctx* ctx, const rmcv_lightblob* blobs, int n_blobs, float angle_diff_max,
                        float shear_max, float length_ratio_max, int enemy, rmcv_armour* armours_out,
                        int armours_cap, int32_t* n_armours);

/* cv::fitEllipseDirect on one contour (the step of src/objdetect.cpp:68), for stage-wise parity */
int rmcv_fit_ellipse(rmcv_ctx* ctx, const rmcv_point* pts, int n, rmcv_rrect* out);

/* ---- batch of independent frames, resident on the device ------------------------------ */

/* copy n_frames host frames (each h rows of `stride` bytes, frames `frame_pitch` bytes apart)
 * into the context's own HBM buffer */
int rmcv_batch_upload(rmcv_ctx* ctx, const uint8_t* frames, int n_frames, int w, int h, int stride,
                      int64_t frame_pitch);
/* or borrow frames that are already in HBM (e.g. a torch tensor's data_ptr) */
int rmcv_batch_set_device_frames(rmcv_ctx* ctx, const void* d_frames, int n_frames, int w, int h, int stride,
                                 int64_t frame_pitch);
/* enqueue the selected stages on `hip_stream` (a hipStream_t, NULL = the context's stream);
 * asynchronous: results are valid after rmcv_batch_sync */
int rmcv_batch_run(rmcv_ctx* ctx, const rmcv_params* p, int stages, void* hip_stream);
int rmcv_batch_sync(rmcv_ctx* ctx);
/* same, but brackets every kernel with HIP events on that stream and, after syncing, reports the
 * milliseconds of each: stage_ms[0]=binary, [1]=contours, [2]=blobs, [3]=armours, [4]=total */
int rmcv_batch_run_timed(rmcv_ctx* ctx, const rmcv_params* p, int stages, void* hip_stream, float stage_ms[5]);

/* per-frame result sizes (arrays of n_frames entries, any may be NULL) */
int rmcv_batch_counts(rmcv_ctx* ctx, int32_t* n_contours, int32_t* n_points, int32_t* n_blobs,
                      int32_t* n_armours, int32_t* status);
int rmcv_batch_get_binary(rmcv_ctx* ctx, int frame, uint8_t* binary_out);
int rmcv_batch_get_contours(rmcv_ctx* ctx, int frame, rmcv_point* pts_out, int pts_cap, int32_t* offs_out,
                            int contours_cap, int32_t* n_contours, int32_t* n_points);
int rmcv_batch_get_blobs(rmcv_ctx* ctx, int frame, rmcv_lightblob* blobs_out, int cap, int32_t* n_blobs,
                         int32_t* blob_src);
/* all armours of the batch, frame-major: frame_offs has n_frames+1 entries */
int rmcv_batch_get_armours(rmcv_ctx* ctx, rmcv_armour* armours_out, int cap, int32_t* frame_offs,
                           int32_t* n_total);
/* device views for a zero-copy hand-over to a collective (RCCL gather of the detections):
 * d_armours[frame][per_frame_cap], d_counts[frame] */
int rmcv_batch_device_views(rmcv_ctx* ctx, void** d_armours, void** d_counts, int32_t* per_frame_cap,
                            int32_t* n_frames);

/* ---- icon classifier: the "next" row of the path (executable/main.cpp:178-181) ------------------------ */
#define RMCV_SVM_FEATURES 1200 /* 20 x 20 x BGR, executable/main.cpp:180 ({20, 20}), core.cpp:202-216 */
/* linear one-vs-one C_SVC as cv::ml::SVM keeps it after training (executable/svm/optimizer.cpp:16-19): one weight
 * vector + rho per class pair (i<j, row-major), class labels in training order.  n_class <= 8. */
int rmcv_svm_load(rmcv_ctx* ctx, const float* weights /* [n_class*(n_class-1)/2][1200] */, const double* rho,
                  const int32_t* labels, int n_class);
/* single frame: identity_out[i] = predict(flatten(affine_correction(frame, armours[i].icon, {20,20})));
 * armours[i].icon is clamped to the frame in place, exactly as the reference does.  icons_out (n*1200 B) may be NULL. */
int rmcv_classify_armours(rmcv_ctx* ctx, const uint8_t* bgr, int w, int h, int stride, rmcv_armour* armours, int n,
                          int32_t* identity_out, uint8_t* icons_out);
/* batch: identities in the order of rmcv_batch_get_armours (after a run that included RMCV_STAGE_IDENTITY) */
int rmcv_batch_get_identities(rmcv_ctx* ctx, int32_t* identity_out, int cap, int32_t* n_total);
int rmcv_batch_get_icons(rmcv_ctx* ctx, int frame, uint8_t* icons_out, int cap_armours, int32_t* n_armours);

/* device-side, frame-major compaction into caller-provided HBM (e.g. torch tensors): d_armours_out has room
 * for `cap` armours, d_frame_offs for n_frames+1 int32 (last entry = total, which may exceed cap: then only the
 * first `cap` were written).  Asynchronous on hip_stream.  This is the payload of the multi-GPU gather. */
int rmcv_batch_compact_armours(rmcv_ctx* ctx, void* d_armours_out, int cap, void* d_frame_offs, void* hip_stream);

/* ---- multi-GPU: the gather of the armour lists (BASELINE config 4) ------------------------------------------------
 * One process per GPU; frames are independent, so there is no collective on the data path.  After rmcv_batch_run +
 * rmcv_batch_compact_armours every rank holds a fixed-size record in HBM ([frame_offs : n_frames+1 int32, padded to 16 B |
 * armours : cap x 88 B], the layout of rmcv_amd/dist.py); rmcv_gather moves the records of all ranks to the root with RCCL
 * point-to-point transfers (each peer's own xGMI link to the root), asynchronously on the caller's stream.
 * RCCL is loaded on first use (librccl.so.1); a single-GPU user never needs it.  The reference is single-process
 * (executable/main.cpp:45-107): this is the north star's addition, not a reference interface. */
#define RMCV_COMM_ID_BYTES 128
typedef struct rmcv_comm rmcv_comm;
/* rank 0: make the group's id; hand the 128 bytes to the other ranks by any means (file, socket, MPI, a launcher's store) */
int  rmcv_comm_unique_id(uint8_t id_out[RMCV_COMM_ID_BYTES]);
/* every rank, collectively: join the group of n_ranks on its own GPU `device` */
int  rmcv_comm_create(const uint8_t id[RMCV_COMM_ID_BYTES], int n_ranks, int rank, int device, rmcv_comm** out);
void rmcv_comm_destroy(rmcv_comm* comm);
int  rmcv_comm_info(const rmcv_comm* comm, int32_t* n_ranks, int32_t* rank);
const char* rmcv_comm_last_error(const rmcv_comm* comm /* NULL: the loader's message */);
/* every rank, collectively: d_record (record_bytes, device memory) -> root's d_recv (n_ranks x record_bytes, rank order; ignored
 * on the other ranks).  Enqueued on hip_stream; the root's buffer is complete when that stream reaches this point. */
int  rmcv_gather(rmcv_comm* comm, const void* d_record, int64_t record_bytes, void* d_recv, int root, void* hip_stream);

/* ---- legacy per-contour matcher: the "next" row SURVEY 8f-2 (src/objdetect.cpp:9-53, 89-112) ---------- */
typedef struct {            /* the float arguments of rm::MatchLightBlob / rm::FindLightBlobs, include/objdetect.h:22-37 */
    float   min_ratio, max_ratio; /* aspect-ratio bounds (strict compares, src/objdetect.cpp:20)                */
    float   tilt_angle;           /* maximal tilt, always judged on the fitted ellipse (src/objdetect.cpp:23-24) */
    float   min_area, max_area;   /* contourArea bounds (strict compares, src/objdetect.cpp:12)                  */
    int32_t fit_ellipse;          /* 1: box = cv::fitEllipseDirect, 0: box = cv::minAreaRect (src/objdetect.cpp:16) */
} rmcv_legacy_params;

/* cv::minAreaRect on one contour (the call of src/objdetect.cpp:16, :69): convex hull + rotating calipers.  pts must be
 * a border as cv::findContours returns it (8-connected, closed: every column of its bounding box holds a point);
 * anything else is RMCV_ERR_BAD_ARG. */
int rmcv_min_area_rect(rmcv_ctx* ctx, const rmcv_point* pts, int n, rmcv_rrect* out);
/* rm::MatchLightBlob (include/objdetect.h:22-23, src/objdetect.cpp:9-28): *matched = 1 and *box_out set when the
 * contour passes every gate */
int rmcv_match_lightblob(rmcv_ctx* ctx, const rmcv_point* pts, int n, const rmcv_legacy_params* lp, rmcv_rrect* box_out,
                         int32_t* matched);
/* rm::FindLightBlobs (include/objdetect.h:35-37, src/objdetect.cpp:30-53): every matching contour becomes a light blob
 * whose camp is voted from the mean B/G/R of `bgr` over the contour's bounding rectangle.  blob_src / boxes_out nullable. */
int rmcv_find_lightblobs(rmcv_ctx* ctx, const uint8_t* bgr, int w, int h, int stride, const rmcv_point* pts,
                         const int32_t* offs, int n_contours, const rmcv_legacy_params* lp, rmcv_lightblob* blobs_out,
                         int blobs_cap, int32_t* n_blobs, int32_t* blob_src, rmcv_rrect* boxes_out);
/* rm::LightBlobOverlap (include/objdetect.h:62, src/objdetect.cpp:89-112).  Host-side predicate over caller memory (a few
 * float compares, no device work).  right == n reads past the end in the reference (its bound check is off by one):
 * that case is RMCV_ERR_BAD_ARG here. */
int rmcv_lightblob_overlap(const rmcv_lightblob* blobs, int n, int left, int right, int32_t* overlap);
/* batch: like rmcv_batch_run, but RMCV_STAGE_BLOBS runs FindLightBlobs with `lp` on every frame's contours (blobs of all
 * camps, in findContours order); RMCV_STAGE_ARMOURS then pairs the blobs whose camp is p->camp. */
int rmcv_batch_run_legacy(rmcv_ctx* ctx, const rmcv_params* p, const rmcv_legacy_params* lp, int stages, void* hip_stream);

/* ---- armour pose: the "next" row SURVEY 8f-3 (src/mobility.cpp:166-190, executable/main.cpp:183-192) -------- */
typedef struct {               /* what the process loop hands to rm::solve_PnP and the world transform */
    double camera_matrix[9];   /* cammat, row-major 3x3                        executable/main.cpp:7-10  */
    double dist[5];            /* discof: k1 k2 p1 p2 k3                       executable/main.cpp:11-13 */
    double gripper2camera[16]; /* h_gripper2camera, row-major 4x4              executable/main.cpp:14-19 */
    float  square_w, square_h; /* exactSize                                    executable/main.cpp:184 {27, 27} */
} rmcv_pnp_config;
void rmcv_default_pnp_config(rmcv_pnp_config* c); /* the literals of executable/main.cpp:7-19, 184 */
int  rmcv_pnp_load(rmcv_ctx* ctx, const rmcv_pnp_config* cfg);
/* single frame: for each armour rvec/tvec = rm::solve_PnP(armour.vertices, cammat, discof, exactSize) and
 * position = base2gripper * (gripper2camera * [tvec; 1]) (executable/main.cpp:186-192).  base2gripper: row-major 4x4
 * (h_base2gripper of executable/main.cpp:170), NULL = identity.  Outputs n x 3 doubles each, any may be NULL. */
int  rmcv_locate_armours(rmcv_ctx* ctx, const rmcv_armour* armours, int n, const double* base2gripper, double* rvecs,
                         double* tvecs, double* positions);
/* batch: one base2gripper per frame (n_frames x 16 doubles, host; default identity), used by RMCV_STAGE_POSE */
int  rmcv_batch_set_base2gripper(rmcv_ctx* ctx, const double* mats, int n_frames);
/* batch: poses in the order of rmcv_batch_get_armours (after a run that included RMCV_STAGE_POSE) */
int  rmcv_batch_get_poses(rmcv_ctx* ctx, double* rvecs, double* tvecs, double* positions, int cap, int32_t* n_total);

/* ---- tracker: the "next" row SURVEY 8f-4 (src/core.cpp:51-162, executable/main.cpp:57-88) --------------------------------
 * Host-side functions over caller memory: the tracker is sequential per target by nature and works on a handful of armours. */
/* rm::armour::max_IoU (src/core.cpp:144-162): *index = the armour of `list` with the largest IoU of bounding boxes with `self`
 * (first on ties, -1 when none overlaps), *iou = that IoU */
int rmcv_max_iou(const rmcv_armour* self, const rmcv_armour* list, int n, int32_t* index, float* iou);
/* rm::armour::identity_max (src/core.cpp:124-142): soft-max vote over an identity histogram; ids ascending (the reference keeps a
 * std::map<int,int>); *max_id = -1 for an empty histogram */
int rmcv_identity_max(const int32_t* ids, const int32_t* counts, int n, int32_t* max_id, double* prob);

/* rm::armour as the tracking thread holds it, with every field readable (include/core.h:101-129).  The filter is
 * cv::KalmanFilter(6, 6, 0, CV_64F) (src/core.cpp:21): state [x y z vx vy vz], matrices row-major 6x6 doubles. */
#define RMCV_TRACK_IDS 32
typedef struct {
    rmcv_armour armour;      /* icon / vertices / bounding_box of the observation the target was created from (update() never refreshes them) */
    int64_t timestamp;       /* ticks of the last observation (core.h:114) */
    int32_t lost_count;      /* core.h:115 */
    int32_t identity;        /* core.h:117 */
    double  position[3];     /* core.h:116 */
    int32_t initialized;     /* core.h:107 */
    int32_t n_ids;           /* identity_history (core.h:103): ids ascending, as a std::map iterates */
    int32_t ids[RMCV_TRACK_IDS], counts[RMCV_TRACK_IDS];
    double  measurement[6];  /* core.h:106 */
    double  state_pre[6], state_post[6];
    double  transition[36], measurement_matrix[36], process_noise_cov[36], measurement_noise_cov[36];
    double  error_cov_pre[36], error_cov_post[36], gain[36];
} rmcv_track;
/* a freshly detected armour as executable/main.cpp:178-194 leaves it: constructor (filter initialised, src/core.cpp:21) +
 * identity, position, timestamp assigned; call rmcv_track_reset next, as main.cpp:195 does */
void rmcv_track_init(rmcv_track* t, const rmcv_armour* a, int32_t identity, int64_t timestamp, const double position[3]);
/* rm::armour::reset (src/core.cpp:51-72); the process loop uses (5e-5, 0.5, 0.05) */
void rmcv_track_reset(rmcv_track* t, double process_noise, double measurement_noise, double error);
/* rm::armour::update(const armour& new_observation) (src/core.cpp:74-108).  tick_frequency = cv::getTickFrequency() */
int  rmcv_track_update(rmcv_track* t, const rmcv_track* observation, double tick_frequency);
/* rm::armour::update(int64 new_timestamp) (src/core.cpp:110-122) */
int  rmcv_track_predict(rmcv_track* t, int64_t new_timestamp, double tick_frequency);
/* one pass of the tracking thread (executable/main.cpp:60-85) over this frame's observations: targets whose bounding box
 * overlaps an observation by IoU > 0.5 take it (and it leaves the list), the others age (dropped after 26 misses -- with
 * the reference's skip of the target behind an erased one) or coast; what is left of the observations becomes new targets.
 * *n_obs is 0 afterwards.  Any number of observations; RMCV_ERR_CAPACITY when the list the pass would leave behind (surviving
 * targets + unmatched observations, counted by a dry run before anything is changed) exceeds cap -- N targets re-observed by N
 * matching observations need cap >= N, as the reference's vectors do -- or when a target would see its 33rd distinct identity
 * (lists handed back consistent). */
int  rmcv_track_step(rmcv_track* tracking, int32_t* n_tracking, int cap, rmcv_track* observations, int32_t* n_obs, double tick_frequency);

/* ---- pipelined batches: the process loop behind the ABI ------------------------------------------------------------------------
 * The reference's process_function is a `while (1)` that takes the newest camera frame, runs the three detection calls and hands
 * the armours on (executable/main.cpp:163-209).  Its batch form on one MI355X: `depth` batches in flight, each in a context of its
 * own; the HBM-bound pixel kernels of consecutive batches alternate over `pixel_streams` HIP streams (two launches overlap: each
 * hides the other's ramp and tail), the latency-bound per-frame kernels run on `sparse_streams` higher-priority streams; a batch's
 * two halves and the reuse of its context are chained by events, the armour lists are compacted frame-major on the device and (by
 * default) copied to pinned host memory behind that.  This is the schedule bench.py's figure is measured on -- owned by the library:
 * a C or C++ host gets it with three calls.
 *
 *     rmcv_pipeline_create(0, &limits, NULL, &pl);
 *     for (;;) { rmcv_pipeline_submit(pl, d_frames, n, w, h, stride, pitch, &params, RMCV_STAGE_ALL, &ticket);
 *                if (ticket >= depth) rmcv_pipeline_collect(pl, ticket - depth + 1, armours, cap, frame_offs, &n_total); }
 *
 * Single-owner like a context: calls on one pipeline must not overlap.  The frames handed to submit are borrowed until the ticket
 * is collected (or waited for).  HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), read once
 * when the HIP runtime starts: the default schedule wants 8 or more (kernels of two batches that share a queue cannot overlap).
 * The library does NOT touch the process environment (round 4's load-time setenv is gone): a host that wants the pipelined schedule
 * calls rmcv_hw_queues_hint() before its first HIP call, or exports GPU_MAX_HW_QUEUES=12 itself; rmcv_pipeline_get_info reports
 * what the variable reads and what the schedule wants.
 *
 * rmcv_pipeline_submit NEVER BLOCKS the host (round 5): it allocates nothing (the ring's contexts hold everything a batch of the
 * limits' size needs from rmcv_pipeline_create on), synchronises nothing and copies nothing synchronously -- a change of geometry
 * (planes zeroed, frame order recomputed) and a slot's change of finishing stream are enqueued work and event waits on the GPU.
 * rmcv_pipeline_info::host_blocking_calls counts the exceptions: it stays 0. */
typedef struct rmcv_pipeline rmcv_pipeline;
typedef struct {            /* 0 in any field = the default; rmcv_default_pipeline_config fills them in */
    int32_t depth;          /* batches in flight = slots (records, tickets) = contexts in the ring (8)  */
    int32_t pixel_streams;  /*                                                                  (2)  */
    int32_t sparse_streams; /*                                                                  (4)  */
    int32_t armour_cap;     /* armours a batch's compacted list holds                (8 per frame)   */
    int32_t sparse_waves;   /* RMCV_OPT_SPARSE_WAVES of the ring's contexts   (4 if depth >= 3 else 8) */
    int32_t pixel_groups;   /* RMCV_OPT_PIXEL_GROUPS                          (2 if depth >= 2 else 3) */
    int32_t host_results;   /* 1: every list is copied to pinned host memory behind its compaction (collect then copies from there);
                             * 2: lists stay on the device until collected                      (1)  */
    int32_t dense_streams;  /* streams for a batch's frames beyond findContours' LDS tables (a lit window, hundreds of specks): WHILE a
                             * few frames per batch are that dense (1 .. max_frames / 8 in the batch that last left the slot; needs
                             * host_results = 1 and sparse_waves = 4), the per-frame launch leaves them to a second launch with 8
                              * wavefronts per frame that runs -- with the compaction behind it -- on one of these streams: the sparse
                             * stream goes on with the next batch instead of waiting for a 0.5-1 ms frame (one lit window per batch:
                             * 1.3 x the plain step time without, 1.01 x with).  Batches without such frames, and batches full of them,
                             * run as if this were off.  -1: off                                                              (4)  */
    int32_t hot_contexts;   /* WHILE the batches are calm -- no frame of the record that last came back went beyond findContours' LDS
                             * tables, no classifier / pose stage asked for -- the batches take turns at the first `hot_contexts`
                             * contexts of the ring (slot, record and ticket window stay `depth` deep) and run the wave-specialised pixel
                             * kernel (RMCV_OPT_PIXEL_SHAPE 1).  What a batch writes with ordinary stores and reads right back -- the
                             * 46 MB bit plane first of all -- then stays in the 256 MB Infinity Cache instead of going to HBM and
                             * back: 3-8 % on the plain stream.  A context's next batch waits for its last one's list, so dense
                             * batches (0.5-1 ms of sparse work) would stall the pixel stream: those run one context per slot, as
                             * with -1.  Needs host_results = 1 and depth >= 4.  0: DERIVED per geometry -- as many contexts as keep
                             * the batches' bit planes within 200 MB of the cache, 3 .. depth - 1 (4 at 256 x 1280x1024, 3 at
                             * 256 x 1920x1200); n: exactly n (3 .. depth - 1); -1: off                                       (0)  */
    int32_t _reserved;
} rmcv_pipeline_config;
typedef struct {
    int32_t depth, pixel_streams, sparse_streams, armour_cap, sparse_waves, pixel_groups, host_results, dense_streams;
    int32_t max_frames;
    int32_t hw_queues_env;     /* what GPU_MAX_HW_QUEUES reads in this process (0: unset) */
    int32_t hw_queues_wanted;  /* 1 + pixel_streams + sparse_streams + dense_streams (+ 1 with a communicator) */
    int32_t _pad;
    int64_t record_bytes;      /* a batch's record in HBM: [frame_offs: max_frames + 1 int32 | status: int32 | load: int32 = frames beyond the LDS tables
                                * (bits 0-19) + border points per frame / 16 (bits 20-31) | pad to 16 B | armours: armour_cap x 88 B] */
    int64_t armours_offset;    /* = the layout of rmcv_amd/dist.py, the payload of rmcv_gather */
    uint64_t submitted, collected;
    uint64_t dense_split;      /* batches whose dense frames were given a launch and a stream of their own (see dense_streams) */
    uint64_t hot_batches;      /* batches that ran in one of the hot contexts (see hot_contexts) */
    int32_t hot_contexts, _pad2;
    uint64_t latency_batches;  /* batches whose back half ran with the latency kernel (8 wavefronts per frame): the newest batch when a call waited
                                * for it -- wait / collect of it, drain -- before another submit (a burst's last batch; one batch at a time) */
    uint64_t host_blocking_calls; /* allocations, host-side synchronisations and blocking copies made inside rmcv_pipeline_submit since the
                                * pipeline was created: 0 (tests/test_gpu_pipeline.py asserts it over plain, dense and re-shaped streams) */
    int32_t wait_timeout_ms, _pad3; /* rmcv_pipeline_set_wait_timeout */
    double   max_submit_us;    /* host time of the longest single rmcv_pipeline_submit since creation / rmcv_pipeline_reset_stats, microseconds */
    uint64_t heavy_batches;    /* batches run in DENSE MODE: while the records that come back are heavy (more than an eighth of the frames beyond
                                * findContours' LDS tables, or >= 1 500 border points per frame; back below 1 200) the sparse stage runs its lean build
                                * (every frame on the mid tier, 61 KB of LDS instead of 80): dense streams 7-10 % faster */
    uint64_t held_back;        /* pixel launches held back behind a burst's first one (k_delay): only launches of the wave-specialised
                                * kernel on every CU, for a quarter of their expected time, 60 us at most, none below 100 us of launch */
} rmcv_pipeline_info;
/* GPU_MAX_HW_QUEUES=12 in the process environment unless the variable is set already; returns what it reads afterwards.  Effective
 * only BEFORE the process's first HIP call (the runtime reads the variable once); not thread-safe (setenv) -- call it first thing
 * in main.  12 = the default schedule's 2 + 4 + 4 streams + the null stream; more is harmful. */
int  rmcv_hw_queues_hint(void);
void rmcv_default_pipeline_config(rmcv_pipeline_config* c);
int  rmcv_pipeline_create(int device, const rmcv_limits* limits /* nullable */, const rmcv_pipeline_config* cfg /* nullable */, rmcv_pipeline** out);
void rmcv_pipeline_destroy(rmcv_pipeline* pl);   /* drains first */
const char* rmcv_pipeline_last_error(const rmcv_pipeline* pl);
int  rmcv_pipeline_get_info(const rmcv_pipeline* pl, rmcv_pipeline_info* out);
/* the context of ring slot `slot` (0 .. depth-1), for set-up that is per context -- rmcv_svm_load (RMCV_STAGE_IDENTITY),
 * rmcv_pnp_load (RMCV_STAGE_POSE), rmcv_ctx_set_option: do it for EVERY slot.  Owned by the pipeline. */
rmcv_ctx* rmcv_pipeline_context(rmcv_pipeline* pl, int slot);
/* the context a ticket's batch ran in, for the per-stage getters (binary image, contours, blobs, counts) on a ticket that has been
 * waited for.  The ticket's RECORD (rmcv_pipeline_collect / _record) lives until ticket + depth is submitted; its context's buffers
 * only until the context's next batch, which can be as early as ticket + hot_contexts (rmcv_pipeline_config): read them before
 * submitting that many more.  NULL for a ticket that is not live OR whose context a later batch has taken (rmcv_pipeline_last_error
 * says which). */
rmcv_ctx* rmcv_pipeline_context_of(rmcv_pipeline* pl, uint64_t ticket);
/* rmcv_pipeline_config::hot_contexts from the next submit on (3 .. depth - 1; 0 or -1: off).  Batches in flight are not touched. */
int  rmcv_pipeline_set_hot_contexts(rmcv_pipeline* pl, int n);
/* rmcv_pipeline_info::max_submit_us starts over (a diagnostic: bench.py reads it per timed region) */
int  rmcv_pipeline_reset_stats(rmcv_pipeline* pl);
/* the deadline of rmcv_pipeline_wait / _collect / _drain (and of the ring contexts' own waits), milliseconds; default 5000, 0: none.
 * A wait that runs out returns RMCV_ERR_TIMEOUT (the batch is still in flight; waiting again is allowed);
 * rmcv_pipeline_last_error names the enqueue made last. */
int  rmcv_pipeline_set_wait_timeout(rmcv_pipeline* pl, int ms);
/* enqueue one batch of n_frames frames that are resident in HBM (layout as rmcv_batch_set_device_frames); stages must include
 * RMCV_STAGE_BINARY.  Asynchronous; *ticket (0, 1, 2, ...) names the batch.  Without a hook, the batch's back half (sparse stage,
 * compaction) is enqueued by the NEXT call on the pipeline: another submit enqueues it as it always was; rmcv_pipeline_wait / _collect
 * of this very ticket, or _drain, know that nothing will run beside it and use the latency kernel (8 wavefronts per frame) -- the
 * last batch of a burst and a host that submits one batch at a time finish 0.03-0.08 ms earlier.  A submitted batch therefore
 * completes only once the pipeline is called again (any call that names it, the next submit, drain, destroy).  The slot's previous batch (ticket - depth) is
 * overwritten: collect it first. */
int  rmcv_pipeline_submit(rmcv_pipeline* pl, const void* d_frames, int n_frames, int w, int h, int stride, int64_t frame_pitch,
                          const rmcv_params* p, int stages, uint64_t* ticket);
/* the same with rm::FindLightBlobs as the blob stage (rmcv_batch_run_legacy) */
int  rmcv_pipeline_submit_legacy(rmcv_pipeline* pl, const void* d_frames, int n_frames, int w, int h, int stride, int64_t frame_pitch,
                                 const rmcv_params* p, const rmcv_legacy_params* lp, int stages, uint64_t* ticket);
/* block until the batch is through (its record complete in HBM and, with host_results, on the host) */
int  rmcv_pipeline_wait(rmcv_pipeline* pl, uint64_t ticket);
/* wait + hand the batch's armours over, frame-major, in submission order of the frames: frame_offs (nullable) has n_frames + 1
 * entries.  RMCV_ERR_CAPACITY: a frame exceeded a context limit, or the list exceeds armour_cap / cap (*n_total says what is
 * needed); RMCV_ERR_BAD_ARG: the ticket was never issued or its slot has been reused. */
int  rmcv_pipeline_collect(rmcv_pipeline* pl, uint64_t ticket, rmcv_armour* armours_out, int cap, int32_t* frame_offs, int32_t* n_total);
/* everything submitted so far is through */
int  rmcv_pipeline_drain(rmcv_pipeline* pl);
/* device view of a ticket's record and the stream it is produced on (work enqueued there runs behind the compaction) */
int  rmcv_pipeline_record(rmcv_pipeline* pl, uint64_t ticket, void** d_record, void** hip_stream);
/* Hook called by rmcv_pipeline_submit, on the submitting thread, right behind the enqueue of a batch's compaction: for a consumer
 * that lives on the device (a collective, a tracker kernel).  What the hook enqueues on `hip_stream` is ordered behind the record;
 * the record's next rewrite (ticket + depth) is ordered behind what the hook enqueued there.  A hook that moves the record on
 * ANOTHER stream returns an event through *done_event (a hipEvent_t it owns, recorded when the record has been read): the rewrite
 * then waits for it.  A non-zero return fails the submit. */
typedef int (*rmcv_pipeline_hook)(void* user, uint64_t ticket, void* d_record, int64_t record_bytes, void* hip_stream, void** done_event);
int  rmcv_pipeline_set_hook(rmcv_pipeline* pl, rmcv_pipeline_hook fn, void* user);
/* built-in hook for BASELINE config 4: every batch's record is gathered to `root` with rmcv_gather on `comm` (every rank submits the
 * same number of batches).  On the root, *d_recv of rmcv_pipeline_gathered is n_ranks x record_bytes in rank order, valid once the
 * ticket has been waited for, until ticket + depth is submitted. */
int  rmcv_pipeline_set_gather(rmcv_pipeline* pl, rmcv_comm* comm, int root);
int  rmcv_pipeline_gathered(rmcv_pipeline* pl, uint64_t ticket, void** d_recv, int64_t* bytes);

/* ---- device memory for hosts without HIP headers (tools/pipeline_bench.c): frames resident in HBM ------------------------------- */
int  rmcv_device_alloc(int device, int64_t bytes, void** d_ptr);
void rmcv_device_free(int device, void* d_ptr);
int  rmcv_device_upload(int device, void* d_dst, const void* h_src, int64_t bytes);   /* synchronous */
int  rmcv_device_download(int device, void* h_dst, const void* d_src, int64_t bytes); /* synchronous */

/* ---- synthetic stream: host generator, integer-only, bit-reproducible --- */
int      rmcv_synth_frame(uint8_t* bgr, int w, int h, int stride, uint64_t frame_index, int camp, int variant);
uint64_t rmcv_synth_checksum(const uint8_t* bgr, int w, int h, int stride);

#ifdef __cplusplus
}
#endif
#endif /* RMCV_ABI_H */
