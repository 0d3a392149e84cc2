// Stand-in for the reference's include/KeyFrame.h: the members src/ORBmatcher.cc touches, same names and types (the
// reference declares most of them const; the test driver fills them after construction).  Test infrastructure.
#ifndef KEYFRAME_H
#define KEYFRAME_H

#include <set>
#include <vector>

#include "MapPoint.h"
#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"
#include "orbfe_cv.h"

namespace ORB_SLAM2 {

class KeyFrame {
public:
    KeyFrame() {}

    cv::Mat GetRotation() { return Tcw.rowRange(0, 3).colRange(0, 3).clone(); }
    cv::Mat GetTranslation() { return Tcw.rowRange(0, 3).col(3).clone(); }
    cv::Mat GetCameraCenter() { return Ow.clone(); }
    std::vector<MapPoint *> GetMapPointMatches() { return mvpMapPoints; }
    MapPoint *GetMapPoint(const size_t &idx) { return mvpMapPoints[idx]; }
    std::set<MapPoint *> GetMapPoints() {   // src/KeyFrame.cc:267-279: every good point
        std::set<MapPoint *> s;
        for (MapPoint *p : mvpMapPoints)
            if (p && !p->isBad()) s.insert(p);
        return s;
    }
    void AddMapPoint(MapPoint *pMP, const size_t &idx) { mvpMapPoints[idx] = pMP; }
    void EraseMapPointMatch(const size_t &idx) { mvpMapPoints[idx] = nullptr; }
    void ReplaceMapPointMatch(const size_t &idx, MapPoint *pMP) { mvpMapPoints[idx] = pMP; }
    bool IsInImage(const float &x, const float &y) const { return (x >= mnMinX && x < mnMaxX && y >= mnMinY && y < mnMaxY); }

    long unsigned int mnId = 0;
    long unsigned int mnFrameId = 0;
    float mfGridElementWidthInv = 0, mfGridElementHeightInv = 0;
    float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0, mb = 0;
    int N = 0;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    std::vector<float> mvuRight;
    cv::Mat mDescriptors;
    DBoW2::FeatureVector mFeatVec;
    int mnScaleLevels = 0;
    float mfScaleFactor = 0, mfLogScaleFactor = 0;
    std::vector<float> mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    int mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0;

    // filled by the test driver
    cv::Mat Tcw, Ow;
    std::vector<MapPoint *> mvpMapPoints;
};

}  // namespace ORB_SLAM2

#endif
