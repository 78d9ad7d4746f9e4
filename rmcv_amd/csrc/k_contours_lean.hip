// k_contours_lean.hip -- the per-frame sparse kernel (k_contours_kernel.inc) for DENSE streams: 4 wavefronts per frame, every frame on the
// mid tier of findContours (tables in global memory), and an LDS footprint of 61 KB instead of 80 -- TWO workgroups per CU.
//
// Why (round 5).  cv::findContours has no bound (/root/reference/src/imgproc.cpp:71-72), and a scene of hundreds of specks or lit windows
// costs a frame 0.2-0.6 ms of sparse work on one workgroup where a plain frame costs 0.1.  Beside the pixel kernels only ONE workgroup of
// the standard build fits a CU (80 KB of LDS, 157 VGPRs beside four pixel workgroups): a batch of 256 such frames is 256 x 0.2-0.6 ms
// over 256 workgroup slots -- the stream becomes sparse-bound (dense4: 0.53 ms per step against the pixel kernel's 0.23).  A frame that
// takes the mid tier needs none of the LDS tier's tables (label planes, node tables for 4 096 visits, per-word tables for 1 664 words):
// compiled with those at token size the same kernel body needs the row tables, 36 KB for the mid tier's pointer doubling and staging
// (which the fused tail's wave-private rows overlay afterwards) and a work list.  The pipeline switches a stream to this kernel
// while the records that come back say the batches are heavy
// (rmcv_pipeline.hip: dense mode).  Same results: the mid tier is the same formulation as the LDS tier, bit for bit (tests/test_gpu_dense.py).
// Measured (round 5, tools/dense_mode_ab.sh, ms per step off / on): dense2 0.320 / 0.291, dense3 0.396 / 0.372, dense4 0.571 / 0.512 (with
// the pixel kernel's two workgroups per CU and launch; with one -- room for two of these per CU -- 0.314 / 0.368 / 0.504).  Tried and not kept: the same
// build WITHOUT the fused tail (114 VGPRs, 6 spilled SGPRs instead of 155 / 292 -- the fits are what the registers go to) with k_fit
// and k_pairs as launches of their own behind it: dense4 0.545 (rocprofv3: k_contours_lean 1.05 ms, k_fit 0.73 ms, k_pairs 0.08 ms per
// batch, overlapped: a dense4 frame's 13 lit windows are 13 general fits of 50 us each, as heavy as its contours); 24 KB instead of 34 for
// the mid tier's LDS (two of these beside FOUR pixel workgroups by LDS): dense2 -1..-3 %, dense3 -3 %, dense4 +14 % (its 8 000 visits then
// double their pointers in global memory); the register cap that would let two of them in by registers as well (96 VGPRs: 176 spilled,
// 280 B of scratch per lane) faulted the queue (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION) -- the deadline waits turned that into an
// error code after five seconds instead of a hung process -- and was not pursued.
#define RMCV_SLOT_CAP 64
#define RMCV_KEPT_CAP 64
#define RMCV_NN_CAP 2048
#define RMCV_CT_THREADS_MAX 256
#ifndef RMCV_LEAN_PAD
#define RMCV_LEAN_PAD 34816
#endif
#define RMCV_MID_PAD RMCV_LEAN_PAD
#include "contours_device.h"

#include <algorithm>

namespace rmcv {

#define KC_KERNEL k_contours_lean
#define KC_THREADS 256
#define KC_NO_CLASSIFY
#include "k_contours_kernel.inc"
#undef KC_NO_CLASSIFY
#undef KC_KERNEL
#undef KC_THREADS

static_assert(sizeof(ContoursLds) <= 54 * 1024, "two of these workgroups and two pixel workgroups share a CU's 160 KB");

// every frame on the mid tier (`flags`: 2, + 8 = only the frames the first launch marked as deferred)
hipError_t launch_contours_lean(const Geom& g, const Bufs& b, const Limits& lim, const SparseTail& X, int flags, const SparseSched& Q, int grid, hipStream_t s)
{
    static bool attr_set[MAX_DEVICES] = {}; // hipFuncSetAttribute applies to the current device only (a process may drive several)
    if (!attr_set[g.device]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_contours_lean), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(CT_MAXH));
        if (e != hipSuccess) return e;
        attr_set[g.device] = true;
    }
    return launch(k_contours_lean, dim3(grid), dim3(256), lds_bytes(g.h), s, b.bits, b.rowmask, g.h, b.lab, b.neg, g.w,
                  g.h, g.ww, g.prow, g.plane_pitch, b.points, b.cont_start, b.cont_len, b.n_contours, b.n_points, b.status,
                  lim.max_contours, lim.max_points, flags, b.elig, b.n_elig, b.slot_kind, X, b.visit_xy, b.mid, b.mid_stride,
                  b.mid_slot_cap, Q, lds_rows_cap(g.h));
}

} // namespace rmcv
