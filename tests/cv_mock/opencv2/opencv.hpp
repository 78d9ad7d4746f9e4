// TEST-ONLY stand-in for the handful of OpenCV types include/rmcv_shim.hpp touches, so that the shim can be
// compiled and run in an image without OpenCV.  It is NOT used to build any reference source and is not part of
// the product; with real OpenCV installed the shim compiles against the real headers instead.
#pragma once
#include <cstddef>
#include <cstdlib>
#include <stdexcept>
#include <vector>
#define CV_8UC1 0
#define CV_8UC3 16
#define CV_32F 5
#define CV_64F 6
typedef long long int64;
#define CV_Assert(x) do { if (!(x)) throw std::runtime_error("CV_Assert: " #x); } while (0)
namespace cv {
typedef unsigned char uchar;
struct Point { int x, y; Point(int x_ = 0, int y_ = 0) : x(x_), y(y_) {} };
struct Point2f { float x, y; Point2f(float x_ = 0, float y_ = 0) : x(x_), y(y_) {} };
struct Size2f { float width, height; Size2f(float w = 0, float h = 0) : width(w), height(h) {} };
struct Rect { int x, y, width, height; Rect(int x_ = 0, int y_ = 0, int w = 0, int h = 0) : x(x_), y(y_), width(w), height(h) {} };
struct Rect2f { float x, y, width, height; Rect2f(float x_ = 0, float y_ = 0, float w = 0, float h = 0) : x(x_), y(y_), width(w), height(h) {} };
struct RotatedRect {
    Point2f center; Size2f size; float angle;
    RotatedRect(Point2f c = Point2f(), Size2f s = Size2f(), float a = 0) : center(c), size(s), angle(a) {}
};
struct Mat {
    int rows = 0, cols = 0, type_ = 0; size_t step = 0; uchar* data = nullptr;
    std::vector<uchar> own;
    Mat() {}
    Mat(int r, int c, int t) : rows(r), cols(c), type_(t) { step = (size_t)c * (t == CV_8UC3 ? 3 : t == CV_64F ? 8 : t == CV_32F ? 4 : 1); own.resize(step * r); data = own.data(); }
    Mat(int r, int c, int t, void* d) : rows(r), cols(c), type_(t), data((uchar*)d) { step = (size_t)c * (t == CV_8UC3 ? 3 : 1); }
    Mat(const Mat& o) : rows(o.rows), cols(o.cols), type_(o.type_), step(o.step), data(o.data), own(o.own) { if (!own.empty()) data = own.data(); }
    Mat& operator=(const Mat& o) { rows = o.rows; cols = o.cols; type_ = o.type_; step = o.step; own = o.own; data = own.empty() ? o.data : own.data(); return *this; }
    static Mat zeros(int r, int c, int t) { return Mat(r, c, t); } // own is value-initialised: all zero
    int type() const { return type_; }
    int channels() const { return type_ == CV_8UC3 ? 3 : 1; }
    bool isContinuous() const { return true; }
    size_t total() const { return (size_t)rows * cols; }
    template <typename T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + step * r); }
    template <typename T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + step * r); }
    template <typename T> T& at(int r, int c) { return ptr<T>(r)[c]; }
};
struct Point3d { double x, y, z; Point3d(double x_ = 0, double y_ = 0, double z_ = 0) : x(x_), y(y_), z(z_) {} };
// shape of cv::KalmanFilter(dynamParams, measureParams, controlParams, type): ten matrices allocated per object -- what makes
// constructing an rm::armour cost what it costs on the CPU (src/core.cpp:21); no filter logic here
struct KalmanFilter {
    Mat statePre, statePost, transitionMatrix, controlMatrix, measurementMatrix, processNoiseCov, measurementNoiseCov, errorCovPre,
        gain, errorCovPost;
    KalmanFilter() {}
    KalmanFilter(int dp, int mp, int cp = 0, int type = CV_32F)
        : statePre(dp, 1, type), statePost(dp, 1, type), transitionMatrix(dp, dp, type), controlMatrix(dp, cp > 0 ? cp : 0, type),
          measurementMatrix(mp, dp, type), processNoiseCov(dp, dp, type), measurementNoiseCov(mp, mp, type),
          errorCovPre(dp, dp, type), gain(dp, mp, type), errorCovPost(dp, dp, type) {}
};
struct _InputArray { const Mat* m; _InputArray(const Mat& mm) : m(&mm) {} Mat getMat() const { return *m; } };
typedef const _InputArray& InputArray;
} // namespace cv
