// rmcv_shim.hpp -- the reference's own rm:: signatures re-hosted on the C-ABI (include/rmcv_abi.h).
//
// Drop-in for the three functions executable/main.cpp:172-176 calls:
//   rm::extract_color      include/imgproc.h:29       (body: src/imgproc.cpp:50-75)
//   rm::filter_lightblobs  include/objdetect.h:47-49  (body: src/objdetect.cpp:55-87)
//   rm::filter_armours     include/objdetect.h:70-71  (body: src/objdetect.cpp:114-166)
//
// Usage: compile this header into EXACTLY ONE translation unit of librmcv in place of
// those bodies, after including the reference's own "core.h" (it supplies rm::camp, rm::range,
// rm::contour, rm::lightblob, rm::armour and the cv:: types), and link librmcv_hip.so.
// The seven rm:: functions below are DEFINITIONS WITH EXTERNAL LINKAGE (they are what executable/main.cpp's
// undefined references resolve to, executable/CMakeLists.txt:1-2): including this header in two translation
// units of one program is an ODR violation by design, exactly as compiling src/objdetect.cpp twice would be.
// A header-only use (caller and shim in one TU) may define RMCV_SHIM_LINKAGE=inline before including it.
// Default arguments stay on the reference's declarations (include/objdetect.h:22-37, include/mobility.h:106-108).
// Also the legacy matcher rm::MatchLightBlob / rm::FindLightBlobs / rm::LightBlobOverlap (include/objdetect.h:22-37, 62)
// and rm::solve_PnP (include/mobility.h:106-108).
// The legacy names of the north star are aliased at the bottom (docs/core_8h_source.html:101,114).
//
// Every signature mentions cv:: types, so this header only compiles where OpenCV headers exist.
#pragma once
#if __has_include(<opencv2/opencv.hpp>)

#include <cstdlib>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "rmcv_abi.h"

#ifndef RMCV_CORE_H
#error "include the reference's core.h before rmcv_shim.hpp"
#endif

#ifndef RMCV_SHIM_LINKAGE
#define RMCV_SHIM_LINKAGE /* external: the backend TU emits rm::extract_color & co. whether or not it calls them */
#endif

namespace rm {
namespace hip_detail {

// one context per calling thread (the reference runs detection on a single process_thread,
// executable/main.cpp:55); created on first use on device RMCV_DEVICE (default 0)
inline rmcv_ctx* ctx()
{
    static thread_local rmcv_ctx* c = [] {
        rmcv_ctx* p = nullptr;
        rmcv_limits lim;
        rmcv_default_limits(&lim);
        lim.max_frames = 1;
        const char* dev = std::getenv("RMCV_DEVICE");
        const int rc = rmcv_ctx_create(dev ? std::atoi(dev) : 0, &lim, &p);
        if (rc != RMCV_OK) throw std::runtime_error("rmcv: no usable MI355X device (this build has no CPU path)");
        return p;
    }();
    return c;
}

inline void check(int rc)
{ // the reference would let a cv::Exception escape (no try/catch in main.cpp); keep that shape
    if (rc != RMCV_OK) throw std::runtime_error(std::string("rmcv: ") + rmcv_last_error(ctx()));
}

inline lightblob to_lightblob(const rmcv_lightblob& b)
{
    lightblob out(cv::RotatedRect(cv::Point2f(b.center[0], b.center[1]), cv::Size2f(b.size[0], b.size[1]), 0.f),
                  static_cast<camp>(b.target));
    out.angle = b.angle;
    out.target = static_cast<camp>(b.target);
    out.center = {b.center[0], b.center[1]};
    for (int i = 0; i < 4; i++) out.vertices[i] = {b.vertices[i][0], b.vertices[i][1]};
    out.size = {b.size[0], b.size[1]};
    return out;
}

inline rmcv_lightblob from_lightblob(const lightblob& b)
{
    rmcv_lightblob o{};
    o.angle = b.angle;
    o.target = static_cast<int32_t>(b.target);
    o.center[0] = b.center.x;
    o.center[1] = b.center.y;
    for (int i = 0; i < 4; i++) { o.vertices[i][0] = b.vertices[i].x; o.vertices[i][1] = b.vertices[i].y; }
    o.size[0] = b.size.width;
    o.size[1] = b.size.height;
    return o;
}

} // namespace hip_detail

RMCV_SHIM_LINKAGE std::tuple<std::vector<contour>, cv::Mat> extract_color(cv::InputArray image, camp target, int lower_bound)
{
    cv::Mat img = image.getMat();
    CV_Assert(img.type() == CV_8UC3);
    cv::Mat binary(img.rows, img.cols, CV_8UC1);
    static thread_local std::vector<rmcv_point> pts;
    static thread_local std::vector<int32_t> offs;
    rmcv_limits lim;
    rmcv_default_limits(&lim);
    pts.resize(lim.max_points);
    offs.resize(lim.max_contours + 1);
    int32_t nc = 0, np = 0;
    hip_detail::check(rmcv_extract_color(hip_detail::ctx(), img.data, img.cols, img.rows, (int)img.step, (int)target,
                                         lower_bound, RMCV_MORPH_CLOSE, binary.data, pts.data(), (int)pts.size(),
                                         offs.data(), (int)offs.size() - 1, &nc, &np));
    std::vector<contour> contours(nc);
    for (int i = 0; i < nc; i++) {
        contours[i].reserve(offs[i + 1] - offs[i]);
        for (int k = offs[i]; k < offs[i + 1]; k++) contours[i].emplace_back(pts[k].x, pts[k].y);
    }
    return {contours, binary};
}

RMCV_SHIM_LINKAGE auto filter_lightblobs(const std::vector<contour>& contours, const float tilt_max, const range<float> ratio_range,
                              const range<double> area_range, camp enemy)
    -> std::tuple<std::vector<lightblob>, std::vector<contour>>
{
    std::vector<rmcv_point> pts;
    std::vector<int32_t> offs(contours.size() + 1, 0);
    for (size_t i = 0; i < contours.size(); i++) {
        for (const auto& p : contours[i]) pts.push_back({p.x, p.y});
        offs[i + 1] = (int32_t)pts.size();
    }
    rmcv_limits lim;
    rmcv_default_limits(&lim);
    std::vector<rmcv_lightblob> blobs(lim.max_blobs);
    std::vector<int32_t> neg(contours.size() + 1);
    int32_t nb = 0, nn = 0;
    hip_detail::check(rmcv_filter_lightblobs(hip_detail::ctx(), pts.data(), offs.data(), (int)contours.size(), tilt_max,
                                             ratio_range.lower_bound, ratio_range.upper_bound, area_range.lower_bound,
                                             area_range.upper_bound, (int)enemy, blobs.data(), (int)blobs.size(), &nb, nullptr,
                                             neg.data(), &nn));
    std::vector<lightblob> positive;
    std::vector<contour> negative;
    positive.reserve(nb);
    for (int i = 0; i < nb; i++) positive.push_back(hip_detail::to_lightblob(blobs[i]));
    for (int i = 0; i < nn; i++) negative.push_back(contours[neg[i]]);
    return {positive, negative};
}

RMCV_SHIM_LINKAGE std::vector<armour> filter_armours(std::vector<lightblob>& lightblobs, const float angle_difference_max,
                                          const float shear_max, const float lenght_ratio_max, const camp enemy)
{
    std::vector<rmcv_lightblob> in;
    in.reserve(lightblobs.size());
    for (const auto& b : lightblobs) in.push_back(hip_detail::from_lightblob(b));
    rmcv_limits lim;
    rmcv_default_limits(&lim);
    std::vector<rmcv_armour> out(lim.max_armours);
    int32_t na = 0;
    hip_detail::check(rmcv_filter_armours(hip_detail::ctx(), in.data(), (int)in.size(), angle_difference_max, shear_max,
                                          lenght_ratio_max, (int)enemy, out.data(), (int)out.size(), &na));
    std::vector<armour> armours;
    armours.reserve(na);
    for (int k = 0; k < na; k++) {
        // rm::armour has one constructor, armour(std::vector<lightblob>) (include/core.h:117): handed anything but two light
        // blobs it returns right after allocating the per-target Kalman state (src/core.cpp:21-23) -- so an EMPTY list buys the
        // object (with its own filter matrices, as every reference armour has) without running the geometry of
        // src/core.cpp:25-48 on the CPU a second time; the device results are then stored into the public members.
        armour a{std::vector<lightblob>{}};
        for (int i = 0; i < 4; i++) {
            a.icon[i] = {out[k].icon[i][0], out[k].icon[i][1]};
            a.vertices[i] = {out[k].vertices[i][0], out[k].vertices[i][1]};
        }
        a.bounding_box = {out[k].bbox[0], out[k].bbox[1], out[k].bbox[2], out[k].bbox[3]};
        armours.push_back(a);
    }
    return armours;
}

// ---- legacy per-contour matcher (include/objdetect.h:22-37, 62; bodies src/objdetect.cpp:9-53, 89-112).  The default
// argument `fitEllipse = true` lives on the reference's declarations.
RMCV_SHIM_LINKAGE bool MatchLightBlob(const rm::contour& contour, float minRatio, float maxRatio, float tiltAngle, float minArea,
                           float maxArea, cv::RotatedRect& lightBlobBox, bool fitEllipse)
{
    std::vector<rmcv_point> pts;
    pts.reserve(contour.size());
    for (const auto& p : contour) pts.push_back({p.x, p.y});
    const rmcv_legacy_params lp = {minRatio, maxRatio, tiltAngle, minArea, maxArea, fitEllipse ? 1 : 0};
    rmcv_rrect box{};
    int32_t matched = 0;
    hip_detail::check(rmcv_match_lightblob(hip_detail::ctx(), pts.data(), (int)pts.size(), &lp, &box, &matched));
    if (!matched) return false;
    lightBlobBox = cv::RotatedRect(cv::Point2f(box.cx, box.cy), cv::Size2f(box.w, box.h), box.angle);
    return true;
}

RMCV_SHIM_LINKAGE void FindLightBlobs(std::vector<contour>& contours, std::vector<lightblob>& lightBlobs, float minRatio, float maxRatio,
                           float tiltAngle, float minArea, float maxArea, const cv::Mat& source, bool fitEllipse)
{
    lightBlobs.clear();
    if (source.channels() != 3) return; // src/objdetect.cpp:35
    std::vector<rmcv_point> pts;
    std::vector<int32_t> offs(contours.size() + 1, 0);
    for (size_t i = 0; i < contours.size(); i++) {
        for (const auto& p : contours[i]) pts.push_back({p.x, p.y});
        offs[i + 1] = (int32_t)pts.size();
    }
    const rmcv_legacy_params lp = {minRatio, maxRatio, tiltAngle, minArea, maxArea, fitEllipse ? 1 : 0};
    std::vector<rmcv_lightblob> blobs(contours.size() + 1);
    int32_t nb = 0;
    hip_detail::check(rmcv_find_lightblobs(hip_detail::ctx(), source.data, source.cols, source.rows, (int)source.step, pts.data(),
                                           offs.data(), (int)contours.size(), &lp, blobs.data(), (int)blobs.size(), &nb, nullptr,
                                           nullptr));
    lightBlobs.reserve(nb);
    for (int i = 0; i < nb; i++) lightBlobs.push_back(hip_detail::to_lightblob(blobs[i]));
}

RMCV_SHIM_LINKAGE bool LightBlobOverlap(const std::vector<rm::lightblob>& lightBlobs, int leftIndex, int rightIndex)
{
    std::vector<rmcv_lightblob> in;
    in.reserve(lightBlobs.size());
    for (const auto& b : lightBlobs) in.push_back(hip_detail::from_lightblob(b));
    int32_t overlap = 0;
    if (rmcv_lightblob_overlap(in.data(), (int)in.size(), leftIndex, rightIndex, &overlap) != RMCV_OK)
        throw std::out_of_range("rm::LightBlobOverlap: rightIndex == lightBlobs.size() (the reference reads past the end here)");
    return overlap != 0;
}

// ---- armour pose (include/mobility.h:106-108; body src/mobility.cpp:166-190).  The default argument ROI = {0,0,0,0} lives on
// the reference's declaration.  cameraMatrix: 3x3 CV_64F, distortionFactor: 1x5 (or 5x1) CV_64F, as executable/main.cpp:7-13.
RMCV_SHIM_LINKAGE std::tuple<cv::Mat, cv::Mat> solve_PnP(const cv::Point2f points_image[4], cv::InputArray cameraMatrix,
                                              cv::InputArray distortionFactor, const cv::Size2f& exactSize, const cv::Rect& ROI)
{
    const cv::Mat K = cameraMatrix.getMat(), D = distortionFactor.getMat();
    CV_Assert(K.type() == CV_64F && K.total() == 9 && D.type() == CV_64F && D.total() == 5 && K.isContinuous() && D.isContinuous());
    rmcv_pnp_config cfg;
    rmcv_default_pnp_config(&cfg);
    for (int i = 0; i < 9; i++) cfg.camera_matrix[i] = K.ptr<double>()[i];
    for (int i = 0; i < 5; i++) cfg.dist[i] = D.ptr<double>()[i];
    cfg.square_w = exactSize.width;
    cfg.square_h = exactSize.height;
    hip_detail::check(rmcv_pnp_load(hip_detail::ctx(), &cfg));
    rmcv_armour a{};
    const float ox = static_cast<float>(ROI.x), oy = static_cast<float>(ROI.y); // src/mobility.cpp:172, 182-185
    for (int i = 0; i < 4; i++) { a.vertices[i][0] = points_image[i].x + ox; a.vertices[i][1] = points_image[i].y + oy; }
    cv::Mat rotation_vector(3, 1, CV_64F), translation_vector(3, 1, CV_64F);
    hip_detail::check(rmcv_locate_armours(hip_detail::ctx(), &a, 1, nullptr, rotation_vector.ptr<double>(),
                                          translation_vector.ptr<double>(), nullptr));
    return {rotation_vector, translation_vector};
}

using LightBlob = lightblob; // pre-2024 API names used by the north star
using Armour = armour;

} // namespace rm

#endif // __has_include(<opencv2/opencv.hpp>)
