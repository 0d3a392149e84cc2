// mini_cv.h -- the few OpenCV types the ORBextractor interface mentions, for building and testing the adapter in an
// image without OpenCV (this container has neither headers nor libraries).  With real OpenCV present include/orbfe_cv.h
// picks <opencv2/core/core.hpp> instead and this file is unused.  Only what include/ORBextractor.h and its tests need.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#ifndef CV_8U
#define CV_8U 0
#define CV_8UC1 0
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
    Mat(int r, int c, int /*type*/) { create(r, c, CV_8U); }
    Mat(int r, int c, int /*type*/, void *ext, size_t step_) : rows(r), cols(c), step(step_), data((uint8_t *)ext) {}
    void create(int r, int c, int /*type*/) {
        rows = r; cols = c; step = (size_t)c;
        store_ = std::shared_ptr<uint8_t>(new uint8_t[(size_t)r * c + 1], std::default_delete<uint8_t[]>());
        data = store_.get();
    }
    void release() { rows = cols = 0; step = 0; data = nullptr; store_.reset(); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return CV_8UC1; }
    uint8_t *ptr(int r = 0) { return data + (size_t)r * step; }
    const uint8_t *ptr(int r = 0) const { return data + (size_t)r * step; }
    Mat roi(int x, int y, int w, int h) const {   // stand-in for Mat::operator()(Rect)
        Mat m;
        m.rows = h; m.cols = w; m.step = step; m.data = data + (size_t)y * step + x; m.store_ = store_;
        return m;
    }
    Mat getMat() const { return *this; }

private:
    std::shared_ptr<uint8_t> store_;
};

typedef const Mat &InputArray;
typedef Mat &OutputArray;

}  // namespace cv
