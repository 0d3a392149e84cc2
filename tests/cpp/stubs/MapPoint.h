// Stand-in for the reference's include/MapPoint.h: exactly the members src/ORBmatcher.cc touches (the way include/mini_cv.h
// stands in for OpenCV), so that include/ORBmatcher.h + adapters/ORBmatcher_orbfe.cc compile and run here without the rest
// of ORB-SLAM2.  Test infrastructure.  Semantics follow src/MapPoint.cc for what the matcher can observe: Observations()
// counts 2 for a stereo keypoint (:98-109), Replace() moves the observations and marks the point bad (:177-215),
// PredictScale() is :385-417.
#ifndef MAPPOINT_H
#define MAPPOINT_H

#include <cmath>
#include <map>

#include "orbfe_cv.h"

namespace ORB_SLAM2 {

class KeyFrame;
class Frame;

class MapPoint {
public:
    MapPoint() {}

    cv::Mat GetWorldPos() { return mWorldPos.clone(); }
    cv::Mat GetNormal() { return mNormalVector.clone(); }
    cv::Mat GetDescriptor() { return mDescriptor.clone(); }
    int Observations() { return nObs; }
    bool isBad() { return mbBad; }
    float GetMinDistanceInvariance() { return 0.8f * mfMinDistance; }
    float GetMaxDistanceInvariance() { return 1.2f * mfMaxDistance; }
    int PredictScale(const float &currentDist, KeyFrame *pKF);
    int PredictScale(const float &currentDist, Frame *pF);
    bool IsInKeyFrame(KeyFrame *pKF) { return mObservations.count(pKF) != 0; }
    int GetIndexInKeyFrame(KeyFrame *pKF) { return mObservations.count(pKF) ? (int)mObservations[pKF] : -1; }
    void AddObservation(KeyFrame *pKF, size_t idx);
    void Replace(MapPoint *pMP);

    long unsigned int mnId = 0;
    int nObs = 0;
    // variables used by the tracking (filled by Frame::isInFrustum in the reference)
    float mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = 0;
    bool mbTrackInView = false;
    int mnTrackScaleLevel = 0;
    float mTrackViewCos = 0;

    // the test driver fills these directly
    cv::Mat mWorldPos, mNormalVector, mDescriptor;
    std::map<KeyFrame *, size_t> mObservations;
    bool mbBad = false;
    MapPoint *mpReplaced = nullptr;
    float mfMinDistance = 0, mfMaxDistance = 0;
};

}  // namespace ORB_SLAM2

#endif
