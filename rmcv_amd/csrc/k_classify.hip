// k_classify.hip -- "next" row SURVEY 8f-1 / BASELINE config 5: per armour, rectify the icon ROI and classify it.
//   rm::affine_correction       /root/reference/src/imgproc.cpp:9-35
//   rm::utils::flatten_image    /root/reference/src/core.cpp:202-216
//   svm->predict                /root/reference/executable/main.cpp:180-181   (7-class linear C_SVC,
//                               model shape executable/svm/optimizer.cpp:9,16-19; svm.xml is not in the reference)
//
// One wavefront per armour.  The reference warps the whole ROI (warpAffine) and then shrinks it to 20x20 (resize);
// a bilinear resize only ever reads 2x2 source pixels per output pixel (or a 2x2 box for the exact 2:1 case), so the
// wave computes just those warped pixels on demand: 400 output pixels, 7 per lane, 16 frame taps each -- the ROI
// never exists in memory.  All arithmetic is OpenCV's 8-bit fixed point (1/32-px coordinates and 15-bit weights in
// warpAffine, 11-bit coefficients in resize), i.e. integers, bit-exact against the CPU restatement.  The 21 linear
// decision functions are 21 lanes, each a sequential float/double dot product in feature order (order dependent).
#include "device_classify.h"

namespace rmcv {

__global__ __launch_bounds__(256) void k_classify(ClassifyArgs C, rmcv_armour* __restrict__ armours, const int32_t* __restrict__ n_armours,
                                                 int max_armours)
{
    __shared__ float s_feat[4][NFEAT];
    __shared__ double s_sum[4][32];
    const int f = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    classify_frame(f, lane, wave, 4, n_armours[f], C, armours, max_armours, s_feat[wave], s_sum[wave]);
}

ClassifyArgs classify_args(const Geom& g, const Bufs& b)
{
    ClassifyArgs C;
    C.frames = b.frames;
    C.frame_pitch = g.frame_pitch;
    C.stride = g.stride;
    C.w = g.w;
    C.h = g.h;
    C.n_class = b.svm_classes;
    C.enabled = 1;
    C.weights = b.svm_w;
    C.rho = b.svm_rho;
    C.labels = b.svm_labels;
    C.identity = b.identity;
    C.icons = b.icons;
    return C;
}

hipError_t launch_classify(const Geom& g, const Bufs& b, const Limits& lim, hipStream_t s)
{
    return launch(k_classify, dim3(g.n_frames), dim3(256), 0, s, classify_args(g, b), b.armours, b.n_armours, lim.max_armours);
}

} // namespace rmcv
