// mini_cv.h -- the few OpenCV types the ORBextractor / ORBmatcher interfaces mention, for building and testing the
// adapters in an image without OpenCV (this container has neither headers nor libraries).  With real OpenCV present
// include/orbfe_cv.h picks <opencv2/core/core.hpp> instead and this file is unused.  Only what include/ORBextractor.h,
// include/ORBmatcher.h, the adapters and their tests need: 8-bit matrices as byte containers, and a small float matrix
// algebra (views, product, sum, transpose, dot, norm) for the projection geometry of the guided searches.  The float
// algebra accumulates in float32 in index order; OpenCV's gemm may round differently, so geometry parity with the
// reference is by construction only in a build against real OpenCV.
#pragma once
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#ifndef CV_8U
#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5
#define CV_32FC1 5
#endif

namespace cv {

struct Point2f {
    float x = 0, y = 0;
    Point2f() {}
    Point2f(float x_, float y_) : x(x_), y(y_) {}
};

struct KeyPoint {   // field order and sizes of cv::KeyPoint: 28 bytes
    Point2f pt;
    float size = 0, angle = -1, response = 0;
    int octave = 0, class_id = -1;
};
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint layout");

class Mat {
public:
    int rows = 0, cols = 0;
    size_t step = 0;
    uint8_t *data = nullptr;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int type, void *ext, size_t step_ = 0)
        : rows(r), cols(c), step(step_ ? step_ : (size_t)c * esz(type)), data((uint8_t *)ext), type_(type) {}
    void create(int r, int c, int type) {
        rows = r; cols = c; type_ = type; step = (size_t)c * esz(type);
        store_ = std::shared_ptr<uint8_t>(new uint8_t[(size_t)r * step + 4](), std::default_delete<uint8_t[]>());
        data = store_.get();
    }
    void release() { rows = cols = 0; step = 0; data = nullptr; store_.reset(); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return type_; }
    size_t elemSize() const { return esz(type_); }
    uint8_t *ptr(int r = 0) { return data + (size_t)r * step; }
    const uint8_t *ptr(int r = 0) const { return data + (size_t)r * step; }
    template <typename T> T *ptr(int r = 0) { return (T *)(data + (size_t)r * step); }
    template <typename T> const T *ptr(int r = 0) const { return (const T *)(data + (size_t)r * step); }
    template <typename T> T &at(int i, int j) { return *(T *)(data + (size_t)i * step + (size_t)j * sizeof(T)); }
    template <typename T> const T &at(int i, int j) const { return *(const T *)(data + (size_t)i * step + (size_t)j * sizeof(T)); }
    // single index: element i of a row or column vector
    template <typename T> T &at(int i) { return rows == 1 ? at<T>(0, i) : at<T>(i, 0); }
    template <typename T> const T &at(int i) const { return rows == 1 ? at<T>(0, i) : at<T>(i, 0); }
    Mat roi(int x, int y, int w, int h) const {   // stand-in for Mat::operator()(Rect)
        Mat m;
        m.rows = h; m.cols = w; m.step = step; m.type_ = type_; m.data = data + (size_t)y * step + (size_t)x * esz(type_);
        m.store_ = store_;
        return m;
    }
    Mat row(int i) const { return roi(0, i, cols, 1); }
    Mat col(int j) const { return roi(j, 0, 1, rows); }
    Mat rowRange(int a, int b) const { return roi(0, a, cols, b - a); }
    Mat colRange(int a, int b) const { return roi(a, 0, b - a, rows); }
    Mat clone() const {
        Mat m;
        if (empty()) return m;
        m.create(rows, cols, type_);
        for (int i = 0; i < rows; i++) std::memcpy(m.ptr(i), ptr(i), (size_t)cols * esz(type_));
        return m;
    }
    Mat getMat() const { return *this; }
    // ---- float algebra (CV_32F only)
    Mat t() const {
        Mat m(cols, rows, CV_32F);
        for (int i = 0; i < rows; i++)
            for (int j = 0; j < cols; j++) m.at<float>(j, i) = at<float>(i, j);
        return m;
    }
    double dot(const Mat &o) const {   // cv::Mat::dot: sum of element products (accumulated in double, as OpenCV does)
        assert(rows * cols == o.rows * o.cols);
        double s = 0;
        const bool v = (rows == o.rows);
        for (int i = 0; i < rows; i++)
            for (int j = 0; j < cols; j++) s += (double)at<float>(i, j) * (double)(v ? o.at<float>(i, j) : o.at<float>(j, i));
        return s;
    }

private:
    static size_t esz(int type) { return type == CV_32F ? 4 : 1; }
    int type_ = CV_8U;
    std::shared_ptr<uint8_t> store_;
};

inline Mat operator*(const Mat &a, const Mat &b) {
    assert(a.cols == b.rows);
    Mat m(a.rows, b.cols, CV_32F);
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < b.cols; j++) {
            float s = 0;
            for (int k = 0; k < a.cols; k++) s += a.at<float>(i, k) * b.at<float>(k, j);
            m.at<float>(i, j) = s;
        }
    return m;
}
inline Mat elementwise(const Mat &a, const Mat &b, float sb) {
    assert(a.rows == b.rows && a.cols == b.cols);
    Mat m(a.rows, a.cols, CV_32F);
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < a.cols; j++) m.at<float>(i, j) = a.at<float>(i, j) + sb * b.at<float>(i, j);
    return m;
}
inline Mat operator+(const Mat &a, const Mat &b) { return elementwise(a, b, 1.0f); }
inline Mat operator-(const Mat &a, const Mat &b) { return elementwise(a, b, -1.0f); }
inline Mat scaled(const Mat &a, double s, bool divide) {
    Mat m(a.rows, a.cols, CV_32F);
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < a.cols; j++)
            m.at<float>(i, j) = divide ? (float)(a.at<float>(i, j) / s) : (float)(a.at<float>(i, j) * s);
    return m;
}
inline Mat operator-(const Mat &a) { return scaled(a, -1.0, false); }
inline Mat operator*(double s, const Mat &a) { return scaled(a, s, false); }
inline Mat operator*(const Mat &a, double s) { return scaled(a, s, false); }
inline Mat operator/(const Mat &a, double s) { return scaled(a, s, true); }
inline double norm(const Mat &a) {   // NORM_L2, accumulated in double
    double s = 0;
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < a.cols; j++) s += (double)a.at<float>(i, j) * (double)a.at<float>(i, j);
    return std::sqrt(s);
}

typedef const Mat &InputArray;
typedef Mat &OutputArray;

}  // namespace cv
