// orbfe_cv.h -- OpenCV core types for the adapters: the real ones when OpenCV is installed, else the test shim.
#pragma once
#if __has_include(<opencv2/core/core.hpp>)
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>   // cv::KeyPoint lives here in OpenCV 2.4
#define ORBFE_HAVE_OPENCV 1
#else
#include "mini_cv.h"
#define ORBFE_HAVE_OPENCV 0
#endif
