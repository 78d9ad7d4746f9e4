// k_contours_w4.hip -- the per-frame sparse kernel (k_contours_kernel.inc) with 4 wavefronts per frame: the throughput setting
// (RMCV_OPT_SPARSE_WAVES = 4).  Its own translation unit, see k_contours.hip.
#include "contours_device.h"

namespace rmcv {

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
                                           (int)lds_bytes(CT_MAXH));
        if (e != hipSuccess) return e;
        attr_set[g.device] = true;
    }
    return launch(k_contours_w4, dim3(grid), dim3(256), lds_bytes(g.h), s, b.bits, b.rowmask, g.h, b.lab, b.neg, g.w,
                       g.h, g.ww, g.prow, g.plane_pitch, b.points, b.cont_start, b.cont_len, b.n_contours, b.n_points, b.status,
                       lim.max_contours, lim.max_points, force_literal, b.elig, b.n_elig, b.slot_kind, X, b.visit_xy, b.mid, b.mid_stride,
                       b.mid_slot_cap, Q, lds_rows_cap(g.h));
}

} // namespace rmcv
