// k_detect.hip -- rm::filter_lightblobs (/root/reference/src/objdetect.cpp:55-87) and
// rm::filter_armours (/root/reference/src/objdetect.cpp:114-166) on the device.
//
// k_blobs:   one wavefront per frame; lane l takes contour 64*chunk + l (findContours order), runs the
//            size/area gate, the ellipse fit and the ratio/tilt tests on its own contour (the moment sums
//            are order dependent, so a contour is never split across lanes), then the wave compacts the
//            positives / negatives IN ORDER with a ballot + prefix popcount.
// k_armours: one wavefront per frame; for each i the lanes test 64 partners j > i at once and append the
//            accepted pairs in (i, j) lexicographic order, again by ballot + prefix popcount.
#include "device_fit.h"
#include "rmcv_internal.h"

namespace rmcv {

__device__ __forceinline__ int lanes_below(uint64_t m, int lane) { return __popcll(m & ((1ull << lane) - 1)); }

__global__ __launch_bounds__(64) void k_blobs(const rmcv_point* __restrict__ points, const int32_t* __restrict__ cont_start,
                                             const int32_t* __restrict__ cont_len, const int32_t* __restrict__ n_contours,
                                             int max_contours, int max_points, float tilt_max, float ratio_lo, float ratio_hi,
                                             double area_lo, double area_hi, int enemy, rmcv_lightblob* __restrict__ blobs,
                                             int32_t* __restrict__ blob_src, rmcv_rrect* __restrict__ ellipses,
                                             int32_t* __restrict__ neg_idx, int32_t* __restrict__ n_blobs,
                                             int32_t* __restrict__ n_neg, int32_t* __restrict__ status, int max_blobs)
{
    const int f = blockIdx.x, lane = threadIdx.x;
    const int n = n_contours[f];
    const rmcv_point* pts = points + (int64_t)f * max_points;
    const int32_t* cs = cont_start + (int64_t)f * max_contours;
    const int32_t* cl = cont_len + (int64_t)f * max_contours;
    rmcv_lightblob* ob = blobs + (int64_t)f * max_blobs;
    int32_t* osrc = blob_src + (int64_t)f * max_blobs;
    rmcv_rrect* oell = ellipses + (int64_t)f * max_blobs;
    int32_t* oneg = neg_idx + (int64_t)f * max_contours;
    int np = 0, nn = 0, st = 0;
    for (int base = 0; base < n; base += 64) {
        const int c = base + lane; // findContours order; discovery index is n-1-c
        int kind = 0;              // 0 skipped, 1 positive, 2 negative
        rmcv_rrect ell;
        if (c < n) {
            const int k = n - 1 - c;
            const int start = cs[k], len = cl[k];
            if (len >= 6 && start + len <= max_points) { // objdetect.cpp:64
                const rmcv_point* cp = pts + start;
                const double area = contour_area(cp, len);
                if (area >= area_lo && area <= area_hi) {
                    fit_ellipse_direct(cp, len, &ell); // :68  (:69 minAreaRect is dead code in the reference)
                    bool negative = false;
                    const float mx = ell.w > ell.h ? ell.w : ell.h, mn = ell.w < ell.h ? ell.w : ell.h;
                    const float ratio = mx / mn; // :71-73
                    if (!(ratio >= ratio_lo && ratio <= ratio_hi)) negative = true;
                    const float angle = ell.angle > 90 ? ell.angle - 90 : ell.angle + 90; // :78
                    if (__builtin_fabsf(angle - 90) > tilt_max) negative = true;          // :79
                    kind = negative ? 2 : 1;
                }
            }
        }
        const uint64_t mp = __ballot(kind == 1), mn_ = __ballot(kind == 2);
        if (kind == 1) {
            const int o = np + lanes_below(mp, lane);
            if (o < max_blobs) {
                make_lightblob(&ell, enemy, &ob[o]); // :83 -> core.cpp:9-19
                osrc[o] = c;
                oell[o] = ell;
            }
        } else if (kind == 2) {
            oneg[nn + lanes_below(mn_, lane)] = c; // :82
        }
        np += __popcll(mp);
        nn += __popcll(mn_);
    }
    if (np > max_blobs) { st |= RMCV_FRAME_OVF_BLOBS; np = max_blobs; }
    if (lane == 0) {
        n_blobs[f] = np;
        n_neg[f] = nn;
        if (st) atomicOr(&status[f], st);
    }
}

__device__ __forceinline__ bool pair_ok(const rmcv_lightblob& a, const rmcv_lightblob& b, float angle_diff_max,
                                        float shear_max, float length_ratio_max)
{
    const float angle_difference = __builtin_fabsf(a.angle - b.angle); // objdetect.cpp:131
    if (angle_difference > angle_diff_max) return false;
    const float y = __builtin_fabsf(a.center[1] - b.center[1]);
    const float x = __builtin_fabsf(a.center[0] - b.center[0]);
    const float rect_angle = pm_atan2f(y, x) * 180.0f / (float)RMCV_PI; // :137
    const float shear_i = __builtin_fabsf(a.angle > 90 ? __builtin_fabsf(a.angle - rect_angle) - 90
                                                       : __builtin_fabsf(180 - a.angle - rect_angle) - 90);
    const float shear_j = __builtin_fabsf(b.angle > 90 ? __builtin_fabsf(b.angle - rect_angle) - 90
                                                       : __builtin_fabsf(180 - b.angle - rect_angle) - 90);
    if (shear_i > shear_max || shear_j > shear_max) return false; // :144
    const float hi = a.size[1], hj = b.size[1];
    const float mn = hi < hj ? hi : hj, mx = hi < hj ? hj : hi;
    if (mn / mx < length_ratio_max) return false;                                                     // :149
    if (__builtin_fabsf(a.center[1] - b.center[1]) > (a.size[1] + b.size[1]) / 2) return false;       // :153
    if (__builtin_fabsf(a.center[0] - b.center[0]) > (a.size[1] + b.size[1]) * 2) return false;       // :157
    return true;
}

__global__ __launch_bounds__(64) void k_armours(const rmcv_lightblob* __restrict__ blobs, const int32_t* __restrict__ n_blobs,
                                               int max_blobs, float angle_diff_max, float shear_max,
                                               float length_ratio_max, int enemy, rmcv_armour* __restrict__ armours,
                                               int32_t* __restrict__ n_armours, int32_t* __restrict__ status, int max_armours)
{
    const int f = blockIdx.x, lane = threadIdx.x;
    const int n = n_blobs[f];
    const rmcv_lightblob* lb = blobs + (int64_t)f * max_blobs;
    rmcv_armour* out = armours + (int64_t)f * max_armours;
    int na = 0;
    if (n >= 2) { // :120
        for (int i = 0; i < n - 1; i++) {
            const rmcv_lightblob a = lb[i];
            if (a.target != enemy) continue; // :124
            for (int jb = i + 1; jb < n; jb += 64) {
                const int j = jb + lane;
                bool ok = false;
                rmcv_lightblob b;
                if (j < n) {
                    b = lb[j];
                    ok = (b.target == enemy) && pair_ok(a, b, angle_diff_max, shear_max, length_ratio_max);
                }
                const uint64_t m = __ballot(ok);
                if (ok) {
                    const int o = na + lanes_below(m, lane);
                    if (o < max_armours) {
                        make_armour(&a, &b, &out[o]); // :161 -> core.cpp:21-49
                        out[o].blob_i = i;
                        out[o].blob_j = j;
                    }
                }
                na += __popcll(m);
            }
        }
    }
    if (lane == 0) {
        if (na > max_armours) {
            atomicOr(&status[f], RMCV_FRAME_OVF_ARMOURS);
            na = max_armours;
        }
        n_armours[f] = na;
    }
}

// frame-major compaction of the per-frame armour slots into one list + offsets (the payload of the
// multi-GPU detection gather).  One workgroup; n_frames is a few hundred.
__global__ __launch_bounds__(256) void k_compact_armours(const rmcv_armour* __restrict__ armours,
                                                        const int32_t* __restrict__ n_armours, int n_frames, int max_armours,
                                                        rmcv_armour* __restrict__ out, int cap, int32_t* __restrict__ frame_offs)
{
    __shared__ int s_part[256];
    __shared__ int s_base;
    const int tid = threadIdx.x;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int f0 = 0; f0 < n_frames; f0 += 256) {
        const int f = f0 + tid;
        const int c = f < n_frames ? n_armours[f] : 0;
        s_part[tid] = c;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) { // inclusive Hillis-Steele scan
            int v = tid >= d ? s_part[tid - d] : 0;
            __syncthreads();
            s_part[tid] += v;
            __syncthreads();
        }
        const int excl = s_base + s_part[tid] - c;
        if (f < n_frames) {
            frame_offs[f] = excl;
            const uint32_t* src = reinterpret_cast<const uint32_t*>(armours + (int64_t)f * max_armours);
            uint32_t* dst = reinterpret_cast<uint32_t*>(out + excl);
            for (int k = 0; k < c; k++)
                if (excl + k < cap)
                    for (int w = 0; w < (int)(sizeof(rmcv_armour) / 4); w++) dst[k * (sizeof(rmcv_armour) / 4) + w] = src[k * (sizeof(rmcv_armour) / 4) + w];
        }
        __syncthreads();
        if (tid == 255) s_base += s_part[255];
        __syncthreads();
    }
    if (tid == 0) frame_offs[n_frames] = s_base;
}

hipError_t launch_compact_armours(const Geom& g, const Bufs& b, const Limits& lim, rmcv_armour* d_out, int cap,
                                  int32_t* d_frame_offs, hipStream_t s)
{
    hipLaunchKernelGGL(k_compact_armours, dim3(1), dim3(256), 0, s, b.armours, b.n_armours, g.n_frames, lim.max_armours, d_out,
                       cap, d_frame_offs);
    return hipGetLastError();
}

hipError_t launch_blobs(const Geom& g, const Bufs& b, const Limits& lim, const rmcv_params& p, hipStream_t s)
{
    hipLaunchKernelGGL(k_blobs, dim3(g.n_frames), dim3(64), 0, s, b.points, b.cont_start, b.cont_len, b.n_contours,
                       lim.max_contours, lim.max_points, p.tilt_max, p.ratio_lo, p.ratio_hi, p.area_lo, p.area_hi, p.camp,
                       b.blobs, b.blob_src, b.ellipses, b.neg_idx, b.n_blobs, b.n_neg, b.status, lim.max_blobs);
    return hipGetLastError();
}

hipError_t launch_armours(const Geom& g, const Bufs& b, const Limits& lim, const rmcv_params& p, hipStream_t s)
{
    hipLaunchKernelGGL(k_armours, dim3(g.n_frames), dim3(64), 0, s, b.blobs, b.n_blobs, lim.max_blobs, p.angle_diff_max,
                       p.shear_max, p.length_ratio_max, p.camp, b.armours, b.n_armours, b.status, lim.max_armours);
    return hipGetLastError();
}

} // namespace rmcv
