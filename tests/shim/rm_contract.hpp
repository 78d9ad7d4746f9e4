// TEST-ONLY stand-in for the reference's public headers, for an image without OpenCV and a GPU box without /root/reference:
// the DATA CONTRACT of include/core.h:20-44,87-130 (names, members incl. the PRIVATE tracking state of rm::armour,
// include/core.h:103-107, constructor signatures) and the DECLARATIONS of include/imgproc.h:29, include/objdetect.h:22-37,47-49,62,70-71
// and include/mobility.h:106-108 with their default arguments.  No logic.  Both translation units of the link test include
// this file the way the reference's units include "rmcv.h"; only backend.cpp also includes rmcv_shim.hpp.
#pragma once
#include <cmath>
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
    explicit lightblob(cv::RotatedRect box, rm::camp camp = rm::CAMP_NEUTRAL); // defined in core_stub.cpp (≙ src/core.cpp)
};
class armour {
    std::map<int, int> identity_history = {};
    cv::KalmanFilter observer;
    cv::Mat measurement = cv::Mat::zeros(6, 1, CV_32F);
    bool initialized = false;

public:
    cv::Point2f icon[4];
    cv::Point2f vertices[4];
    cv::Rect2f bounding_box;
    int64 timestamp = 0;
    int lost_count = 0;
    cv::Point3d position;
    int identity = -1;
    explicit armour(std::vector<lightblob> lightblobs); // defined in core_stub.cpp (≙ src/core.cpp)
    bool has_filter_state() const { return observer.errorCovPost.rows == 6 && measurement.rows == 6 && !initialized; } // test probe
};

// include/imgproc.h:29
std::tuple<std::vector<contour>, cv::Mat> extract_color(cv::InputArray image, camp target, int lower_bound);
// include/objdetect.h:22-23, 35-37, 47-49, 62, 70-71
bool MatchLightBlob(const rm::contour& contour, float minRatio, float maxRatio, float tiltAngle, float minArea, float maxArea,
                    cv::RotatedRect& lightBlobBox, bool fitEllipse = true);
void FindLightBlobs(std::vector<contour>& contours, std::vector<lightblob>& lightBlobs, float minRatio, float maxRatio,
                    float tiltAngle, float minArea, float maxArea, const cv::Mat& source, bool fitEllipse = true);
auto filter_lightblobs(const std::vector<contour>& contours, float tilt_max, range<float> ratio_range, range<double> area_range,
                       camp enemy) -> std::tuple<std::vector<lightblob>, std::vector<contour>>;
bool LightBlobOverlap(const std::vector<rm::lightblob>& lightBlobs, int leftIndex, int rightIndex);
std::vector<armour> filter_armours(std::vector<lightblob>& lightblobs, float angle_difference_max, float shear_max,
                                   float lenght_ratio_max, camp enemy);
// include/mobility.h:106-108
std::tuple<cv::Mat, cv::Mat> solve_PnP(const cv::Point2f points_image[4], cv::InputArray cameraMatrix,
                                       cv::InputArray distortionFactor, const cv::Size2f& exactSize,
                                       const cv::Rect& ROI = {0, 0, 0, 0});
} // namespace rm
