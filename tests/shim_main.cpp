// TEST-ONLY driver of include/rmcv_shim.hpp: the three calls of the reference's process loop
// (/root/reference/executable/main.cpp:172-176) through the rm:: signatures, on one synthetic frame.
// The rm:: data types below restate the DATA CONTRACT of the reference's include/core.h:20-44,87-130 (names,
// public members, constructor signatures) so that the shim's conversions are exercised; they carry no logic.
#include <cmath>
#include <cstdio>
#include <map>
#include <tuple>
#include <vector>

#include <opencv2/opencv.hpp>
#define RMCV_CORE_H
namespace rm {
enum camp { CAMP_RED = 0, CAMP_BLUE = 1, CAMP_GUIDELIGHT = 2, CAMP_NEUTRAL = -1 };
template <typename T> struct range {
    T lower_bound, upper_bound;
    range(T lower, T upper) : lower_bound(lower), upper_bound(upper) {}
};
typedef std::vector<cv::Point> contour;
class lightblob {
public:
    float angle = 0;
    camp target = CAMP_NEUTRAL;
    cv::Point2f center;
    cv::Point2f vertices[4];
    cv::Size2f size;
    explicit lightblob(cv::RotatedRect box, rm::camp c = rm::CAMP_NEUTRAL) : target(c), center(box.center) {}
};
class armour {
public:
    cv::Point2f icon[4];
    cv::Point2f vertices[4];
    cv::Rect2f bounding_box;
    explicit armour(std::vector<lightblob>) {}
};
} // namespace rm
#include "rmcv_shim.hpp"

int main(int argc, char** argv)
{
    const int w = 1280, h = 1024, index = argc > 1 ? atoi(argv[1]) : 0;
    cv::Mat frame(h, w, CV_8UC3);
    if (rmcv_synth_frame(frame.data, w, h, 3 * w, (uint64_t)index, 1, 0)) return 2;
    auto [contours, binary] = rm::extract_color(frame, rm::CAMP_BLUE, 80);
    auto [positive, negative] = rm::filter_lightblobs(contours, 70, {1.5f, 80.0f}, {10, 99999}, rm::CAMP_BLUE);
    auto armours = rm::filter_armours(positive, 12, 22, 0.4f, rm::CAMP_BLUE);
    size_t on = 0;
    for (size_t i = 0; i < (size_t)w * h; i++) on += binary.data[i] != 0;
    std::printf("contours %zu points %zu binary_on %zu positive %zu negative %zu armours %zu\n", contours.size(),
                [&] { size_t n = 0; for (auto& c : contours) n += c.size(); return n; }(), on, positive.size(), negative.size(),
                armours.size());
    for (auto& a : armours) {
        std::printf("armour");
        for (int i = 0; i < 4; i++) std::printf(" %a %a", a.vertices[i].x, a.vertices[i].y);
        std::printf("\n");
    }
    // the legacy matcher (include/objdetect.h:22-37, 62) through the same shim
    std::vector<rm::lightblob> legacy;
    rm::FindLightBlobs(contours, legacy, 1.5f, 80.0f, 70.0f, 10.0f, 99999.0f, frame, false);
    std::printf("legacy %zu", legacy.size());
    for (auto& b : legacy) std::printf(" %d %a %a", (int)b.target, b.size.width, b.size.height);
    std::printf("\n");
    size_t matched = 0;
    cv::RotatedRect box;
    for (auto& c : contours) matched += rm::MatchLightBlob(c, 1.5f, 80.0f, 70.0f, 10.0f, 99999.0f, box, true) ? 1 : 0;
    int overlaps = 0;
    for (int i = 0; i + 2 < (int)legacy.size(); i++) overlaps += rm::LightBlobOverlap(legacy, i, i + 2) ? 1 : 0;
    std::printf("matched %zu overlaps %d\n", matched, overlaps);
    // the pose of every armour: rm::solve_PnP with the camera constants of executable/main.cpp:7-13
    rmcv_pnp_config pc;
    rmcv_default_pnp_config(&pc);
    cv::Mat cammat(3, 3, CV_64F), discof(1, 5, CV_64F);
    for (int i = 0; i < 9; i++) cammat.ptr<double>()[i] = pc.camera_matrix[i];
    for (int i = 0; i < 5; i++) discof.ptr<double>()[i] = pc.dist[i];
    for (auto& a : armours) {
        auto [rvec, tvec] = rm::solve_PnP(a.vertices, cammat, discof, {27, 27}, cv::Rect(0, 0, 0, 0));
        std::printf("pose %a %a %a %a %a %a\n", rvec.ptr<double>()[0], rvec.ptr<double>()[1], rvec.ptr<double>()[2],
                    tvec.ptr<double>()[0], tvec.ptr<double>()[1], tvec.ptr<double>()[2]);
    }
    rm::LightBlob* alias_check = positive.empty() ? nullptr : &positive[0];
    (void)alias_check;
    return 0;
}
