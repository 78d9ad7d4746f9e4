// TEST-ONLY caller translation unit of the link test: sees the reference-style DECLARATIONS only (rm_contract.hpp stands in for
// "rmcv.h"), never the shim -- the position executable/main.cpp is in (executable/CMakeLists.txt:1-2).  It makes the three
// calls of the reference's process loop (executable/main.cpp:172-176) on one synthetic frame, then the legacy three and
// solve_PnP; every rm:: reference below must be resolved by backend.o at link time.
#include <cstdio>

#include "rm_contract.hpp"
#include "rmcv_abi.h" // rmcv_synth_frame / rmcv_default_pnp_config only (test input + the literals of main.cpp:7-13)

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
    for (auto& c : contours) matched += rm::MatchLightBlob(c, 1.5f, 80.0f, 70.0f, 10.0f, 99999.0f, box) /* default fitEllipse = true */ ? 1 : 0;
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
        auto [rvec, tvec] = rm::solve_PnP(a.vertices, cammat, discof, {27, 27}) /* default ROI */;
        std::printf("pose %a %a %a %a %a %a\n", rvec.ptr<double>()[0], rvec.ptr<double>()[1], rvec.ptr<double>()[2],
                    tvec.ptr<double>()[0], tvec.ptr<double>()[1], tvec.ptr<double>()[2]);
    }
    for (auto& a : armours)
        if (!a.has_filter_state()) return 3; // every returned armour owns its Kalman state, as a reference armour does
    return 0;
}
