// Stand-in for the reference's include/Frame.h: the members src/ORBmatcher.cc touches, same names and types (statics
// included).  Test infrastructure.
#ifndef FRAME_H
#define FRAME_H

#include <vector>

#include "KeyFrame.h"
#include "MapPoint.h"
#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"
#include "orbfe_cv.h"

namespace ORB_SLAM2 {

class Frame {
public:
    Frame() {}

    static float fx, fy, cx, cy;
    float mbf = 0, mb = 0;
    int N = 0;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    std::vector<float> mvuRight;
    DBoW2::FeatureVector mFeatVec;
    cv::Mat mDescriptors;
    std::vector<MapPoint *> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    static float mfGridElementWidthInv, mfGridElementHeightInv;
    cv::Mat mTcw;
    long unsigned int mnId = 0;
    int mnScaleLevels = 0;
    float mfScaleFactor = 0, mfLogScaleFactor = 0;
    std::vector<float> mvScaleFactors, mvInvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    static float mnMinX, mnMaxX, mnMinY, mnMaxY;
};

}  // namespace ORB_SLAM2

#endif
