// refcheck_cv.h -- DECLARATIONS ONLY, for `make -C tools/refcheck`: a syntax check (-fsyntax-only, no object code) of this
// repository's two adapters against the reference's own class declarations (include/Frame.h, KeyFrame.h, MapPoint.h and
// the vendored DBoW2 headers they pull in), in an image without OpenCV.  The <opencv2/...> names below forward to
// include/mini_cv.h and add the few further cv:: names those headers mention.  Nothing here is linked, run, or used as an
// oracle, and no reference source is compiled: the check answers one question -- do the adapters still parse and
// type-check against the classes they claim to slot in beside.
#pragma once
#include <algorithm>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <limits>
#include <list>
#include <map>
#include <set>
#include <sstream>
#include <string>

#include "mini_cv.h"

namespace cv {
typedef std::string String;
template <class T> struct Point_ { T x, y; Point_() : x(0), y(0) {} Point_(T a, T b) : x(a), y(b) {} };
typedef Point_<int> Point;
typedef Point_<int> Point2i;
struct Size { int width, height; Size() : width(0), height(0) {} Size(int w, int h) : width(w), height(h) {} };
struct Scalar { double v[4]; Scalar(double a = 0, double b = 0, double c = 0, double d = 0) : v{a, b, c, d} {} };
class SparseMat;
class FileNode {
public:
    FileNode operator[](const char *) const;
    FileNode operator[](const std::string &) const;
    FileNode operator[](int) const;
    size_t size() const;
    operator int() const;
    operator float() const;
    operator double() const;
    operator std::string() const;
};
class FileStorage {
public:
    enum { READ = 0, WRITE = 1 };
    FileStorage();
    FileStorage(const std::string &, int);
    bool isOpened() const;
    void release();
    FileNode operator[](const char *) const;
    FileNode operator[](const std::string &) const;
};
FileStorage &operator<<(FileStorage &, const std::string &);
FileStorage &operator<<(FileStorage &, const char *);
FileStorage &operator<<(FileStorage &, int);
FileStorage &operator<<(FileStorage &, double);
}  // namespace cv
