// device_classify.h -- per armour: rectify the icon ROI and classify it (rm::affine_correction, src/imgproc.cpp:9-35;
// rm::utils::flatten_image, src/core.cpp:202-216; svm->predict, executable/main.cpp:180-181).  One wavefront per armour; shared by
// the stand-alone k_classify (k_classify.hip, the stage-wise entry points) and the fused tail of the per-frame sparse kernel
// (k_contours_kernel.inc: a frame's armours are classified by the workgroup that has just built them, so BASELINE config 5 needs
// no launch of its own for it).
#pragma once
#include <float.h>

#include "device_fit.h"
#include "rmcv_internal.h"

namespace rmcv {

static constexpr int ICON = 20, NFEAT = ICON * ICON * 3;

__device__ __forceinline__ int cv_round_d(double v) { return (int)__builtin_rint(v); }      // round half to even
__device__ __forceinline__ int cv_round_f(float v) { return (int)__builtin_rintf(v); }
__device__ __forceinline__ int cv_floor_f(float v) { int i = (int)v; return i - (i > v); }

struct WarpCtx {
    const uint8_t* roi; // frame + by*stride + 3*bx
    int stride, bw, bh;
    double M[6]; // inverted map
};

// one channel triple of the warped ROI at (x, y): cv::warpAffine INTER_LINEAR, BORDER_CONSTANT 0, CV_8UC3
__device__ inline void warp_px(const WarpCtx& W, int x, int y, int out[3])
{
    const int AB_SCALE = 1 << 10, round_delta = 16;
    const int X0 = cv_round_d((W.M[1] * y + W.M[2]) * AB_SCALE) + round_delta;
    const int Y0 = cv_round_d((W.M[4] * y + W.M[5]) * AB_SCALE) + round_delta;
    const int adelta = cv_round_d(W.M[0] * x * AB_SCALE), bdelta = cv_round_d(W.M[3] * x * AB_SCALE);
    const int X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
    int sx = X >> 5, sy = Y >> 5;
    sx = sx > 32767 ? 32767 : (sx < -32768 ? -32768 : sx);
    sy = sy > 32767 ? 32767 : (sy < -32768 ? -32768 : sy);
    const int fx = X & 31, fy = Y & 31;
    int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    if (w00 > 32767) { w00 = 32767; w11 += 1; }
    const bool y0ok = sy >= 0 && sy < W.bh, y1ok = sy + 1 >= 0 && sy + 1 < W.bh;
    const bool x0ok = sx >= 0 && sx < W.bw, x1ok = sx + 1 >= 0 && sx + 1 < W.bw;
    const uint8_t* p = W.roi + (int64_t)sy * W.stride + 3 * sx;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const int v00 = (y0ok && x0ok) ? p[c] : 0, v01 = (y0ok && x1ok) ? p[3 + c] : 0;
        const int v10 = (y1ok && x0ok) ? p[W.stride + c] : 0, v11 = (y1ok && x1ok) ? p[W.stride + 3 + c] : 0;
        const int v = (v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11 + (1 << 14)) >> 15;
        out[c] = v < 0 ? 0 : (v > 255 ? 255 : v);
    }
}

struct ClassifyArgs { // what the classification of a frame's armours needs (Bufs: frames, svm_*, identity, icons)
    const uint8_t* frames;
    int64_t frame_pitch;
    int stride, w, h, n_class, enabled;
    const float* weights;
    const double* rho;
    const int32_t* labels;
    int32_t* identity;
    uint8_t* icons;
};

ClassifyArgs classify_args(const Geom& g, const Bufs& b); // k_classify.hip

// armours wave, wave + nwaves, ... of frame f by this wavefront.  feat: NFEAT floats, sums: 32 doubles of wave-private LDS.
__device__ inline void classify_frame(int f, int lane, int wave, int nwaves, int n, const ClassifyArgs& C, rmcv_armour* __restrict__ armours,
                                      int max_armours, float* feat, double* sums)
{
    const uint8_t* frame = C.frames + (int64_t)f * C.frame_pitch;
    const int w = C.w, h = C.h, stride = C.stride, n_class = C.n_class;
    const float* __restrict__ weights = C.weights;
    const double* __restrict__ rho = C.rho;
    const int32_t* __restrict__ labels = C.labels;
    int32_t* __restrict__ identity = C.identity;
    uint8_t* __restrict__ icons = C.icons;
    for (int a = wave; a < n; a += nwaves) {
        rmcv_armour* A = armours + (int64_t)f * max_armours + a;
        // ---- imgproc.cpp:11-15 clamp (in place, like the reference), :17 integer bounding box
        float ic[4][2];
        int minx = 0, maxx = 0, miny = 0, maxy = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float vx0 = A->icon[i][0], vy0 = A->icon[i][1];
            const float vx = vx0 < (float)w - 1 ? vx0 : (float)w - 1, vy = vy0 < (float)h - 1 ? vy0 : (float)h - 1;
            ic[i][0] = 0.0f > vx ? 0.0f : vx;
            ic[i][1] = 0.0f > vy ? 0.0f : vy;
            const int px = cv_round_f(ic[i][0]), py = cv_round_f(ic[i][1]);
            if (i == 0 || px < minx) minx = px;
            if (i == 0 || px > maxx) maxx = px;
            if (i == 0 || py < miny) miny = py;
            if (i == 0 || py > maxy) maxy = py;
        }
        const int bx = minx, by = miny, bw = maxx - minx + 1, bh = maxy - miny + 1;
        const bool degenerate = (bw <= 0 || bh <= 0 || bx < 0 || by < 0 || bx + bw > w || by + bh > h);
        // ---- :18-28 getAffineTransform: 6x6 LU with partial pivoting, matrix kept across the lanes
        WarpCtx W;
        W.roi = frame + (int64_t)by * stride + 3 * bx;
        W.stride = stride;
        W.bw = bw;
        W.bh = bh;
        {
            const float src[3][2] = {{ic[1][0] - (float)bx, ic[1][1] - (float)by},
                                     {ic[2][0] - (float)bx, ic[2][1] - (float)by},
                                     {ic[0][0] - (float)bx, ic[0][1] - (float)by}};
            const float dst[3][2] = {{0, 0}, {(float)bw, 0}, {0, (float)bh}};
            LaneVec Am(lane), bv(lane);
            { // row 2i: [x y 1 0 0 0], row 2i+1: [0 0 0 x y 1]   (selects, not lane-dependent subscripts: those put the arrays in scratch memory)
                const int r = lane / 6, c = lane - 6 * r, i = r >> 1;
                const float sxi = i == 0 ? src[0][0] : (i == 1 ? src[1][0] : src[2][0]);
                const float syi = i == 0 ? src[0][1] : (i == 1 ? src[1][1] : src[2][1]);
                double v = 0.0;
                if (lane < 36) {
                    const int cc = (r & 1) ? c - 3 : c;
                    if (cc >= 0 && cc < 3) v = cc == 0 ? (double)sxi : (cc == 1 ? (double)syi : 1.0);
                }
                Am.reg = v;
                // b = (dst0.x, dst0.y, dst1.x, dst1.y, dst2.x, dst2.y) = (0, 0, bw, 0, 0, bh)
                bv.reg = lane == 2 ? (double)dst[1][0] : (lane == 5 ? (double)dst[2][1] : 0.0);
            }
            bool ok = true;
            for (int i = 0; i < 6 && ok; i++) {
                int k = i;
                for (int j = i + 1; j < 6; j++)
                    if (dabs(Am[j * 6 + i]) > dabs(Am[k * 6 + i])) k = j;
                if (dabs(Am[k * 6 + i]) < DBL_EPSILON * 100) { ok = false; break; }
                if (k != i) {
                    for (int j = i; j < 6; j++) { const double t = Am[i * 6 + j]; Am[i * 6 + j] = (double)Am[k * 6 + j]; Am[k * 6 + j] = t; }
                    const double t = bv[i]; bv[i] = (double)bv[k]; bv[k] = t;
                }
                const double d = -1 / (double)Am[i * 6 + i];
                for (int j = i + 1; j < 6; j++) {
                    const double alpha = (double)Am[j * 6 + i] * d;
                    for (int kk = i + 1; kk < 6; kk++) Am[j * 6 + kk] += alpha * (double)Am[i * 6 + kk];
                    bv[j] += alpha * (double)bv[i];
                }
            }
            if (ok)
                for (int i = 5; i >= 0; i--) {
                    double s = bv[i];
                    for (int k = i + 1; k < 6; k++) s -= (double)Am[i * 6 + k] * (double)bv[k];
                    bv[i] = s / (double)Am[i * 6 + i];
                }
            double M[6];
#pragma unroll
            for (int i = 0; i < 6; i++) M[i] = ok ? bv.get(i) : 0.0;
            // warpAffine inverts the map (no WARP_INVERSE_MAP)
            double D = M[0] * M[4] - M[1] * M[3];
            D = D != 0 ? 1. / D : 0;
            const double A11 = M[4] * D, A22 = M[0] * D;
            M[0] = A11; M[1] *= -D;
            M[3] *= -D; M[4] = A22;
            const double b1 = -M[0] * M[2] - M[1] * M[5];
            const double b2 = -M[3] * M[2] - M[4] * M[5];
            M[2] = b1; M[5] = b2;
#pragma unroll
            for (int i = 0; i < 6; i++) W.M[i] = M[i];
        }
        // ---- :30-32 warpAffine (ROI size) + resize to 20x20, pixels on demand
        const bool area = (bw == 2 * ICON && bh == 2 * ICON);
        const double scale_x = (double)bw / ICON, scale_y = (double)bh / ICON;
        for (int o = lane; o < ICON * ICON; o += 64) {
            const int dy = o / ICON, dx = o - dy * ICON;
            int res[3] = {0, 0, 0};
            if (!degenerate) {
                if (area) {
                    int p00[3], p01[3], p10[3], p11[3];
                    warp_px(W, 2 * dx, 2 * dy, p00);
                    warp_px(W, 2 * dx + 1, 2 * dy, p01);
                    warp_px(W, 2 * dx, 2 * dy + 1, p10);
                    warp_px(W, 2 * dx + 1, 2 * dy + 1, p11);
#pragma unroll
                    for (int c = 0; c < 3; c++) res[c] = (p00[c] + p01[c] + p10[c] + p11[c] + 2) >> 2;
                } else {
                    float fx = (float)((dx + 0.5) * scale_x - 0.5);
                    int sx = cv_floor_f(fx);
                    fx -= sx;
                    if (sx < 0) { fx = 0; sx = 0; }
                    bool edge = false; // dx >= xmax: single tap weighted 2048
                    if (sx + 1 >= bw) {
                        edge = true;
                        if (sx >= bw - 1) { fx = 0; sx = bw - 1; }
                    }
                    const int a0 = (short)cv_round_f((1.f - fx) * 2048), a1 = (short)cv_round_f(fx * 2048);
                    float fy = (float)((dy + 0.5) * scale_y - 0.5);
                    const int sy = cv_floor_f(fy);
                    fy -= sy;
                    const int b0 = (short)cv_round_f((1.f - fy) * 2048), b1 = (short)cv_round_f(fy * 2048);
                    int sy0 = sy, sy1 = sy + 1;
                    sy0 = sy0 < 0 ? 0 : (sy0 > bh - 1 ? bh - 1 : sy0);
                    sy1 = sy1 < 0 ? 0 : (sy1 > bh - 1 ? bh - 1 : sy1);
                    int p00[3], p01[3] = {0, 0, 0}, p10[3], p11[3] = {0, 0, 0};
                    warp_px(W, sx, sy0, p00);
                    warp_px(W, sx, sy1, p10);
                    // xmax is the first dx whose right tap falls outside: every later dx is an edge column too
                    if (!edge) {
                        warp_px(W, sx + 1, sy0, p01);
                        warp_px(W, sx + 1, sy1, p11);
                    }
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const int r0 = edge ? p00[c] * 2048 : p00[c] * a0 + p01[c] * a1;
                        const int r1 = edge ? p10[c] * 2048 : p10[c] * a0 + p11[c] * a1;
                        const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
                        res[c] = v < 0 ? 0 : (v > 255 ? 255 : v);
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < 3; c++) {
                feat[o * 3 + c] = (float)res[c]; // flatten_image: reshape(1,1), CV_32FC1
                if (icons) icons[((int64_t)f * max_armours + a) * NFEAT + o * 3 + c] = (uint8_t)res[c];
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- main.cpp:181 svm->predict: one lane per decision function
        const int n_df = n_class * (n_class - 1) / 2;
        if (lane < n_df) {
            const float* wv = weights + (int64_t)lane * NFEAT;
            double s = 0;
            for (int k = 0; k < NFEAT; k += 4) {
                const float4 wq = *reinterpret_cast<const float4*>(wv + k);
                s += wq.x * feat[k] + wq.y * feat[k + 1] + wq.z * feat[k + 2] + wq.w * feat[k + 3];
            }
            const float kval = (float)(s * 1.0 + 0.0);
            sums[lane] = -rho[lane] + 1.0 * kval;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            // votes of the (at most 8) classes as eight 4-bit counters of one word (a class meets at most 7 others): a register,
            // where an int[8] indexed by the winner went to scratch memory
            uint32_t votes = 0;
            int dfi = 0;
            for (int i = 0; i < n_class; i++)
                for (int j = i + 1; j < n_class; j++, dfi++) votes += 1u << (4 * (sums[dfi] > 0 ? i : j));
            int best = 0, bv_ = (int)(votes & 15u);
#pragma unroll
            for (int q = 1; q < 8; q++) {
                const int vq = (int)((votes >> (4 * q)) & 15u);
                if (q < n_class && vq > bv_) { best = q; bv_ = vq; }
            }
            identity[(int64_t)f * max_armours + a] = labels[best];
#pragma unroll
            for (int i = 0; i < 4; i++) { A->icon[i][0] = ic[i][0]; A->icon[i][1] = ic[i][1]; }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

} // namespace rmcv
