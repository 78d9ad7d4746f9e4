// k_contours_w4.hip -- the per-frame sparse kernel (k_contours_kernel.inc) with 4 wavefronts per frame: the throughput setting
// (RMCV_OPT_SPARSE_WAVES = 4).  Its own translation unit, see k_contours.hip.
#include "contours_device.h"

#include <algorithm>

namespace rmcv {

static constexpr size_t LDS_ONE_PER_CU = 84 * 1024;

#define KC_KERNEL k_contours_w4
#define KC_THREADS 256
#include "k_contours_kernel.inc"
#undef KC_KERNEL
#undef KC_THREADS

hipError_t launch_contours_w4(const Geom& g, const Bufs& b, const Limits& lim, const SparseTail& X, int force_literal, const SparseSched& Q, int grid,
                              hipStream_t s)
{
    static bool attr_set[MAX_DEVICES] = {}; // hipFuncSetAttribute applies to the current device only (a process may drive several)
    if (!attr_set[g.device]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_contours_w4), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)std::max(lds_bytes(CT_MAXH), LDS_ONE_PER_CU));
        if (e != hipSuccess) return e;
        attr_set[g.device] = true;
    }
    // Beside the wave-specialised pixel kernel (Geom::pixel_ws: a pipeline's calm batches) ONE of these workgroups per CU: two of them
    // (2 x 168 VGPRs per SIMD) leave no room for k_binary_ws's 4 x 72, and whenever two batches' sparse kernels reached the CUs in the gap
    // between two pixel launches the next pixel workgroups waited 0.1-0.25 ms for one of them to finish -- after which the launches ran
    // in lock-step pairs, 0.28-0.34 ms per step instead of 0.235 (rocprofv3 trace: profiles/r04h_pixel_stream_stalls.txt).  More than half
    // of the CU's 160 KB of LDS asked for = one per CU; k_binary_ws's 20 KB still fit beside it.
    static const bool one_per_cu = !(getenv("RMCV_W4_ONE_PER_CU") && atoi(getenv("RMCV_W4_ONE_PER_CU")) == 0); // dev knob (A/B)
    const size_t lds = (g.pixel_ws && one_per_cu) ? std::max(lds_bytes(g.h), LDS_ONE_PER_CU) : lds_bytes(g.h);
    return launch(k_contours_w4, dim3(grid), dim3(256), lds, s, b.bits, b.rowmask, g.h, b.lab, b.neg, g.w,
                       g.h, g.ww, g.prow, g.plane_pitch, b.points, b.cont_start, b.cont_len, b.n_contours, b.n_points, b.status,
                       lim.max_contours, lim.max_points, force_literal, b.elig, b.n_elig, b.slot_kind, X, b.visit_xy, b.mid, b.mid_stride,
                       b.mid_slot_cap, Q, lds_rows_cap(g.h));
}

} // namespace rmcv
