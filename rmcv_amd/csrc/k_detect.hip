// k_detect.hip -- rm::filter_lightblobs (/root/reference/src/objdetect.cpp:55-87) and
// rm::filter_armours (/root/reference/src/objdetect.cpp:114-166) on the device.
//
// k_fit:     one wavefront per contour (findContours order): size/area gate, ellipse fit, ratio/tilt tests.
// k_pairs:   one wavefront per frame appends the positives / negatives IN ORDER (ballot + prefix popcount),
//            builds the rm::lightblob PODs and runs the pair loop of filter_armours.
// k_armours: one wavefront per frame; for each i the lanes test 64 partners j > i at once and append the
//            accepted pairs in (i, j) lexicographic order, again by ballot + prefix popcount.
#include "device_hull.h"
#include "wave_detect.h"

namespace rmcv {

// kinds: 0 skipped, 1 positive, 2 negative.  grid (frames, FIT_CHUNKS), 4 wavefronts per block, one contour each.
static constexpr int FIT_CHUNKS = 4; // x 4 wavefronts = 16 contours of a frame at a time (a frame has ~10 with >= 6 points)
__global__ __launch_bounds__(256) void k_fit(const rmcv_point* __restrict__ points, const int32_t* __restrict__ cont_start,
                                            const int32_t* __restrict__ cont_len, const int32_t* __restrict__ n_contours,
                                            int max_contours, int max_points, float tilt_max, float ratio_lo, float ratio_hi,
                                            double area_lo, double area_hi, int32_t* __restrict__ slot_kind,
                                            rmcv_rrect* __restrict__ slot_ell, const int32_t* __restrict__ elig,
                                            const int32_t* __restrict__ n_elig, int ov)
{
    __shared__ WaveLds lds[4];
    __builtin_amdgcn_s_setprio(3); // latency-bound: issue ahead of the streaming pixel kernel of the next batch sharing the CU
    const int f = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    WaveLds& L = lds[wave];
    const int n = n_contours[f];
    const rmcv_point* pts = points + (int64_t)f * max_points;
    const int32_t* cs = cont_start + (int64_t)f * max_contours;
    const int32_t* cl = cont_len + (int64_t)f * max_contours;
    // the work list holds the discovery indices of the contours with >= 6 points (k_contours / load_contours wrote it
    // and marked the others "skipped"), so wavefronts are only spent on contours that reach the fit
    const int ne = n_elig[f];
    const FitGates G = {tilt_max, ratio_lo, ratio_hi, area_lo, area_hi, ov};
    for (int e = blockIdx.y * 4 + wave; e < ne; e += 4 * gridDim.y)
        fit_contour_slot(f, elig[(int64_t)f * max_contours + e], n, pts, cs, cl, max_contours, max_points, G, slot_kind, slot_ell, L, lane);
}

// ---- legacy matcher: rm::MatchLightBlob + the camp vote of rm::FindLightBlobs ----------------------
// One wavefront per contour of the frame's work list, one wavefront per workgroup (the hull tables take 16 B of LDS per
// frame column).  mode 0: the matcher (objdetect.cpp:9-28, 43-51); mode 1: cv::minAreaRect alone, every gate open (the
// stage-wise hook rmcv_min_area_rect).  Results go to the same per-contour slots k_fit uses, so k_pairs does the ordered
// compaction; a matching contour's slot word is 1 | (camp + 2) << 4.
struct MatchArgs {
    float min_ratio, max_ratio, tilt_angle, min_area, max_area;
    int fit_ellipse, mode;
    int ov; // SURVEY A.6 (device_fit.h: abs_ov)
    int wcap, hcap, ccap; // hull table sizes of this launch (device_hull.h)
    int pass;             // 0: all contours, small tables, what does not fit is marked MATCH_DEFERRED; 1: the marked ones, full tables
    const uint8_t* frames; // may be null: no camp vote (the blob gets CAMP_NEUTRAL)
    int64_t frame_pitch;
    int stride;
};

static constexpr int MATCH_CHUNKS = 16;
static constexpr int MATCH_DEFERRED = 0x100; // slot word: kind 0 (skipped by the compaction) + "repeat with full-size tables"
__global__ __launch_bounds__(64) void k_match(const rmcv_point* __restrict__ points, const int32_t* __restrict__ cont_start,
                                             const int32_t* __restrict__ cont_len, const int32_t* __restrict__ n_contours,
                                             int max_contours, int max_points, MatchArgs A, int32_t* __restrict__ slot_kind,
                                             rmcv_rrect* __restrict__ slot_ell, const int32_t* __restrict__ elig,
                                             const int32_t* __restrict__ n_elig, int32_t* __restrict__ status)
{
    extern __shared__ unsigned long long match_smem[];
    __builtin_amdgcn_s_setprio(3);
    const int f = blockIdx.x, lane = threadIdx.x;
    WaveLds& L = *reinterpret_cast<WaveLds*>(match_smem);
    HullLds H;
    hull_lds_carve(reinterpret_cast<unsigned char*>(match_smem) + sizeof(WaveLds), A.wcap, A.hcap, A.ccap, H);
    const bool final_pass = A.pass == 1 || A.mode == 1;
    const int n = n_contours[f];
    const rmcv_point* pts = points + (int64_t)f * max_points;
    const int32_t* cs = cont_start + (int64_t)f * max_contours;
    const int32_t* cl = cont_len + (int64_t)f * max_contours;
    const int ne = A.mode == 1 ? n : n_elig[f]; // the hook takes any contour, the matcher only those with >= 6 points
    for (int e = blockIdx.y; e < ne; e += gridDim.y) {
        const int k = A.mode == 1 ? e : elig[(int64_t)f * max_contours + e];
        const int c = n - 1 - k; // findContours order
        const int start = cs[k], len = cl[k];
        if (A.pass == 1 && slot_kind[(int64_t)f * max_contours + c] != MATCH_DEFERRED) continue;
        int word = 0;
        rmcv_rrect box = {0, 0, 0, 0, 0};
        if (len >= 1 && start + len <= max_points && (A.mode == 1 || len >= 6)) { // objdetect.cpp:12
            const rmcv_point* cp = pts + start;
            double a00 = 0;
            long long sx = 0, sy = 0;
            int minx = INT_MAX, maxx = INT_MIN, miny = INT_MAX, maxy = INT_MIN;
            for (int i = lane; i < len; i += 64) {
                const rmcv_point p = cp[i], q = cp[i == 0 ? len - 1 : i - 1];
                a00 += (double)(float)q.x * (float)p.y - (double)(float)q.y * (float)p.x;
                sx += p.x;
                sy += p.y;
                minx = p.x < minx ? p.x : minx;
                maxx = p.x > maxx ? p.x : maxx;
                miny = p.y < miny ? p.y : miny;
                maxy = p.y > maxy ? p.y : maxy;
            }
            a00 = wave_sum_f64(a00);
            sx = wave_sum_i64(sx);
            sy = wave_sum_i64(sy);
            for (int d = 32; d >= 1; d >>= 1) {
                const int a = __shfl_xor(minx, d), b = __shfl_xor(maxx, d), cc = __shfl_xor(miny, d), dd = __shfl_xor(maxy, d);
                minx = a < minx ? a : minx;
                maxx = b > maxx ? b : maxx;
                miny = cc < miny ? cc : miny;
                maxy = dd > maxy ? dd : maxy;
            }
            const int W = maxx - minx + 1, Hh = maxy - miny + 1; // cv::boundingRect on int points: inclusive box
            const double area = dabs(a00 * 0.5);
            int ovf = 0;
            if (minx < 0 || miny < 0 || W > A.wcap || maxy >= HULL_MAX_DIM || len > (1 << 20)) ovf = 1;
            if (A.mode == 1) {
                if (!ovf) min_area_rect_wave(cp, len, minx, W, H, lane, &box, &ovf);
                word = 1;
                if (ovf && lane == 0) atomicOr(&status[f], RMCV_FRAME_HULL);
            } else if (!(area < A.min_area || area > A.max_area)) { // :12
                rmcv_rrect ellipse;
                fit_ellipse_wave(cp, len, sx, sy, L, lane, &ellipse); // :15
                if (A.fit_ellipse) {
                    box = ellipse;
                    ovf = 0;
                } else if (!ovf) {
                    min_area_rect_wave(cp, len, minx, W, H, lane, &box, &ovf); // :16
                }
                if (ovf) { // the box is missing: repeat with the full-size tables, or give up (never for a findContours border)
                    word = final_pass ? 0 : MATCH_DEFERRED;
                    if (final_pass && lane == 0) atomicOr(&status[f], RMCV_FRAME_HULL);
                } else {
                    const float mx = box.w > box.h ? box.w : box.h, mn = box.w < box.h ? box.w : box.h;
                    const float ratio = mx / mn; // :19
                    bool ok = !(ratio > A.max_ratio || ratio < A.min_ratio);
                    const float angle = ellipse.angle > 90 ? ellipse.angle - 90 : ellipse.angle + 90; // :23
                    if (abs_ov(angle - 90, A.ov) > A.tilt_angle) ok = false;                           // :24
                    if (ok) {
                        int camp = RMCV_CAMP_NEUTRAL;
                        if (A.frames) // :43-51
                            camp = camp_from_mean_wave(A.frames + (int64_t)f * A.frame_pitch, A.stride, minx, miny, W, Hh, lane);
                        word = 1 | ((camp + 2) << 4);
                    }
                }
            }
        }
        if (lane == 0) {
            slot_kind[(int64_t)f * max_contours + c] = word;
            slot_ell[(int64_t)f * max_contours + c] = box;
        }
    }
}


// stand-alone pairing (the stage-wise rmcv_filter_armours entry point, or RMCV_STAGE_ARMOURS without RMCV_STAGE_BLOBS)
__global__ __launch_bounds__(64) void k_armours(const rmcv_lightblob* __restrict__ blobs, const int32_t* __restrict__ n_blobs,
                                               int max_blobs, float angle_diff_max, float shear_max,
                                               float length_ratio_max, int enemy, rmcv_armour* __restrict__ armours,
                                               int32_t* __restrict__ n_armours, int32_t* __restrict__ status, int max_armours, int ov)
{
    armours_frame(blockIdx.x, threadIdx.x, blobs, n_blobs[blockIdx.x], max_blobs, angle_diff_max, shear_max, length_ratio_max, enemy,
                  armours, n_armours, status, max_armours, ov);
}

// the tail of filter_lightblobs (ordered compaction) and filter_armours (pair loop) of one frame in one launch:
// one wavefront per frame.  (Kept out of k_fit on purpose: folded into k_fit's last-arriving workgroup it cost
// 26 VGPRs, one wave of occupancy and +0.13 ms on MI355X.)
__global__ __launch_bounds__(64) void k_pairs(const int32_t* __restrict__ slot_kind, const rmcv_rrect* __restrict__ slot_ell,
                                             const int32_t* __restrict__ n_contours, int max_contours, FitTail T)
{
    const int f = blockIdx.x, lane = threadIdx.x;
    __builtin_amdgcn_s_setprio(3);
    const int np = blob_compact_frame(f, lane, slot_kind, slot_ell, n_contours[f], max_contours, T.enemy, T.blobs, T.blob_src,
                                      T.ellipses, T.neg_idx, T.n_blobs, T.n_neg, T.status, T.max_blobs);
    if (T.do_pairs) {
        // the pair loop re-reads, across lanes, the blobs this wave has just written: workgroup scope is enough (one CU, one L1) -- at
            // device scope the fence is an L2 write-back + invalidate (buffer_wbl2 sc1, buffer_inv sc1) per frame, on the XCD whose
            // L2 the pixel kernels of the other batches are streaming through: it cost the step 3 %
            __threadfence_block();
        armours_frame(f, lane, T.blobs, np, T.max_blobs, T.angle_diff_max, T.shear_max, T.length_ratio_max, T.enemy, T.armours,
                      T.n_armours, T.status, T.max_armours, T.ov);
    }
}

// frame-major compaction of the per-frame armour slots into one list + offsets (the payload of the
// multi-GPU detection gather).  n_frames is a few hundred: EVERY workgroup scans all the counts (one read each + an LDS scan: a few
// microseconds) and then copies the armours of its own COMPACT_FRAMES consecutive frames, one wavefront per frame, dword per lane.
// (Round 1 had one workgroup whose threads each copied their frame's armours dword after dword: ~60 dependent global round trips,
// 8 us alone but 50 us beside the streaming kernels of the other batches -- on the critical path of every step's sparse chain.)
static constexpr int COMPACT_FRAMES = 16;
__global__ __launch_bounds__(256) void k_compact_armours(const rmcv_armour* __restrict__ armours,
                                                        const int32_t* __restrict__ n_armours, int n_frames, int max_armours,
                                                        rmcv_armour* __restrict__ out, int cap, int32_t* __restrict__ frame_offs,
                                                        const int32_t* __restrict__ status, int32_t* __restrict__ status_or,
                                                        uint8_t* __restrict__ host_rec, int host_head, const int32_t* __restrict__ n_points)
{
    // host_rec (nullable): the same record a second time, in pinned host memory mapped into the device's address space -- the
    // kernel's own stores cross PCIe as posted writes, and only the armours there are travel (a copy of the whole record behind the
    // kernel was a hand-over to the copy engine with a system-scope fence of its own, per step)
    int32_t* const h_offs = reinterpret_cast<int32_t*>(host_rec);
    rmcv_armour* const h_out = reinterpret_cast<rmcv_armour*>(host_rec + host_head);
    __shared__ int s_part[256];
    __shared__ int s_base, s_st, s_mid;
    __shared__ unsigned long long s_pts;
    __shared__ int s_off[COMPACT_FRAMES], s_cnt[COMPACT_FRAMES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f_begin = blockIdx.x * COMPACT_FRAMES; // this workgroup's frames
    if (tid == 0) { s_base = 0; s_st = 0; s_mid = 0; s_pts = 0; }
    __syncthreads();
    for (int f0 = 0; f0 < n_frames; f0 += 256) {
        const int f = f0 + tid;
        const int c = f < n_frames ? n_armours[f] : 0;
        // the batch's status bits OR-ed into one word (rmcv_pipeline_collect reads it with the list instead of n_frames words)
        // ... and the number of frames that were beyond findContours' LDS tables (the pipeline's schedule follows it)
        // ... and how heavy the batch's sparse work was: border points per frame (the pipeline's dense mode follows it)
        if (status_or && blockIdx.x == 0 && f < n_frames) {
            const int st = status[f];
            if (st) { atomicOr(&s_st, st); if (st & RMCV_FRAME_MID_PATH) atomicAdd(&s_mid, 1); }
            atomicAdd(&s_pts, (unsigned long long)(n_points[f] > 0 ? n_points[f] : 0));
        }
        s_part[tid] = c;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) { // inclusive Hillis-Steele scan
            int v = tid >= d ? s_part[tid - d] : 0;
            __syncthreads();
            s_part[tid] += v;
            __syncthreads();
        }
        const int excl = s_base + s_part[tid] - c;
        if (f < n_frames && f >= f_begin && f < f_begin + COMPACT_FRAMES) {
            frame_offs[f] = excl;
            if (host_rec) h_offs[f] = excl;
            s_off[f - f_begin] = excl;
            s_cnt[f - f_begin] = c;
        }
        __syncthreads();
        if (tid == 255) s_base += s_part[255];
        __syncthreads();
    }
    if (tid == 0 && blockIdx.x == 0) {
        frame_offs[n_frames] = s_base;
        // the record's second word: frames beyond the LDS tables (bits 0-19) | border points per frame / 16, capped (bits 20-31)
        const unsigned long long per16 = n_frames > 0 ? s_pts / (unsigned long long)n_frames / 16ull : 0ull;
        const int word2 = (s_mid & 0xFFFFF) | (int)((per16 > 4095ull ? 4095ull : per16) << 20);
        if (status_or) { status_or[0] = s_st; status_or[1] = word2; }
        if (host_rec) {
            h_offs[n_frames] = s_base;
            h_offs[status_or - frame_offs] = s_st; // (the status word's place in the record)
            h_offs[status_or - frame_offs + 1] = word2;
        }
    }
    constexpr int DW = (int)(sizeof(rmcv_armour) / 4);
    for (int fi = wave; fi < COMPACT_FRAMES && f_begin + fi < n_frames; fi += 4) {
        const int excl = s_off[fi], c = s_cnt[fi];
        const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(armours + (int64_t)(f_begin + fi) * max_armours);
        uint32_t* __restrict__ dst = reinterpret_cast<uint32_t*>(out + excl);
        const int nd = min(c, max(0, cap - excl)) * DW; // armours beyond the capacity of the list are dropped (the caller is told)
        uint32_t* __restrict__ hdst = reinterpret_cast<uint32_t*>(h_out + excl);
        if (host_rec) {
            for (int i = lane; i < nd; i += 64) { const uint32_t v = src[i]; dst[i] = v; hdst[i] = v; }
        } else {
            for (int i = lane; i < nd; i += 64) dst[i] = src[i];
        }
    }
}

__global__ void k_status_clear(int32_t* __restrict__ status, int n, int mask)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < n) status[f] &= ~mask;
}

hipError_t launch_status_clear(const Geom& g, const Bufs& b, int mask, hipStream_t s)
{
    return launch(k_status_clear, dim3((g.n_frames + 255) / 256), dim3(256), 0, s, b.status, g.n_frames, mask);
}

hipError_t launch_compact_armours(const Geom& g, const Bufs& b, const Limits& lim, rmcv_armour* d_out, int cap,
                                  int32_t* d_frame_offs, hipStream_t s, int32_t* d_status_or, uint8_t* hd_record, int host_head)
{
    if (hd_record && !d_status_or) return hipErrorInvalidValue; // (the host mirror has the record's layout, status word included)
    return launch(k_compact_armours, dim3(std::max(1, (g.n_frames + COMPACT_FRAMES - 1) / COMPACT_FRAMES)), dim3(256), 0, s, b.armours, b.n_armours, g.n_frames, lim.max_armours, d_out,
                       cap, d_frame_offs, b.status, d_status_or, hd_record, host_head, b.n_points);
}

static hipError_t launch_pairs_tail(const Geom& g, const Bufs& b, const Limits& lim, const rmcv_params& p, bool pairs, hipStream_t s)
{
    FitTail T;
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
    return launch(k_pairs, dim3(g.n_frames), dim3(64), 0, s, b.slot_kind, b.slot_ell, b.n_contours, lim.max_contours, T);
}

static hipError_t launch_fit(const Geom& g, const Bufs& b, const Limits& lim, const rmcv_params& p, bool pairs, hipStream_t s)
{
    hipError_t e = launch(k_fit, dim3(g.n_frames, FIT_CHUNKS), dim3(256), 0, s, b.points, b.cont_start, b.cont_len, b.n_contours,
                       lim.max_contours, lim.max_points, p.tilt_max, p.ratio_lo, p.ratio_hi, p.area_lo, p.area_hi, b.slot_kind,
                       b.slot_ell, b.elig, b.n_elig, g.overloads);
    if (e != hipSuccess) return e;
    return launch_pairs_tail(g, b, lim, p, pairs, s);
}

// legacy matcher: k_match fills the per-contour slots, k_pairs compacts them (and pairs the blobs of camp p.camp)
hipError_t launch_match(const Geom& g, const Bufs& b, const Limits& lim, const rmcv_params& p, const rmcv_legacy_params& lp,
                        int mode, bool with_frames, bool pairs, hipStream_t s)
{
    MatchArgs A;
    A.min_ratio = lp.min_ratio;
    A.max_ratio = lp.max_ratio;
    A.tilt_angle = lp.tilt_angle;
    A.ov = g.overloads;
    A.min_area = lp.min_area;
    A.max_area = lp.max_area;
    A.fit_ellipse = lp.fit_ellipse;
    A.mode = mode;
    A.frames = with_frames ? b.frames : nullptr;
    A.frame_pitch = g.frame_pitch;
    A.stride = g.stride;
    const int full_w = lim.max_width < HULL_MAX_DIM ? lim.max_width : HULL_MAX_DIM;
    const bool hull = mode == 1 || !lp.fit_ellipse;
    // pass 0 (every contour): small hull tables, ~10 KB of LDS per wavefront; pass 1 repeats the contours pass 0 marked
    // (wider than 256 columns, or a hull with more points than the small tables hold) with the full-size tables.  The hook
    // (mode 1) is a single full-size pass; with fitEllipse there is no hull at all.
    for (int pass = 0; pass < ((hull && mode == 0) ? 2 : 1); pass++) {
        const bool full = mode == 1 || pass == 1;
        A.pass = pass;
        A.wcap = !hull ? 1 : full ? full_w : (full_w < 256 ? full_w : 256);
        A.hcap = !hull ? 1 : full ? HULL_CAP : 128;
        A.ccap = !hull ? 1 : full ? HULL_CHAIN_CAP : 64;
        const size_t lds = sizeof(WaveLds) + hull_lds_bytes(A.wcap, A.hcap, A.ccap);
        static size_t lds_set_dev[MAX_DEVICES] = {}; // per device, like the attribute itself
        size_t& lds_set = lds_set_dev[g.device];
        if (lds > lds_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_match), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            lds_set = lds;
        }
        hipError_t e = launch(k_match, dim3(g.n_frames, pass ? 2 : MATCH_CHUNKS), dim3(64), lds, s, b.points, b.cont_start, b.cont_len,
                           b.n_contours, lim.max_contours, lim.max_points, A, b.slot_kind, b.slot_ell, b.elig, b.n_elig, b.status);
        if (e != hipSuccess) return e;
    }
    return launch_pairs_tail(g, b, lim, p, pairs, s);
}

hipError_t launch_blobs(const Geom& g, const Bufs& b, const Limits& lim, const rmcv_params& p, hipStream_t s)
{
    return launch_fit(g, b, lim, p, false, s);
}

// filter_lightblobs + filter_armours: k_fit, then one k_pairs launch for both tails
hipError_t launch_blobs_armours(const Geom& g, const Bufs& b, const Limits& lim, const rmcv_params& p, hipStream_t s)
{
    return launch_fit(g, b, lim, p, true, s);
}

hipError_t launch_armours(const Geom& g, const Bufs& b, const Limits& lim, const rmcv_params& p, hipStream_t s)
{
    return launch(k_armours, dim3(g.n_frames), dim3(64), 0, s, b.blobs, b.n_blobs, lim.max_blobs, p.angle_diff_max,
                       p.shear_max, p.length_ratio_max, p.camp, b.armours, b.n_armours, b.status, lim.max_armours, g.overloads);
}

} // namespace rmcv
