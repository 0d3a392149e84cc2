// ORBmatcher.h -- drop-in replacement for the reference's include/ORBmatcher.h (abhineet123/ORB_SLAM2).
//
// Same namespace, class name, constructor, static DescriptorDistance, the eleven Search* / Fuse methods with the
// reference's signatures and the three public constants (reference include/ORBmatcher.h:41-89), so src/Frame.cc,
// src/Tracking.cc, src/LocalMapping.cc, src/LoopClosing.cc and src/MapPoint.cc compile and link unchanged against it.
// adapters/ORBmatcher_orbfe.cc holds the member definitions: the projection geometry and the map bookkeeping stay host
// code written with the same cv::Mat expressions as the reference, the window searches (GetFeaturesInArea + the Hamming
// loops) and the vocabulary-node searches run in liborbfe.so (hand-written HIP for gfx950) through the C-ABI in orbfe.h.
// Frame.h / KeyFrame.h / MapPoint.h are the reference's own headers, untouched: the device copy of a frame's keypoint grid
// lives in a pool inside the adapter keyed by frame id, no class gains a member.
#ifndef ORBMATCHER_H
#define ORBMATCHER_H

#include <set>
#include <utility>
#include <vector>

#include "orbfe_cv.h"

#include "MapPoint.h"
#include "KeyFrame.h"
#include "Frame.h"

namespace ORB_SLAM2 {

class ORBmatcher {
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true);

    // Hamming distance between two ORB descriptors (32-byte rows)
    static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b);

    // Tracking::SearchLocalPoints: projected local MapPoints (mbTrackInView) against the frame's keypoints
    int SearchByProjection(Frame &F, const std::vector<MapPoint *> &vpMapPoints, const float th = 3);

    // Tracking::TrackWithMotionModel: MapPoints of the last frame projected into the current one
    int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono);

    // Tracking::Relocalization: MapPoints of a KeyFrame projected into the current frame
    int SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const std::set<MapPoint *> &sAlreadyFound, const float th,
                           const int ORBdist);

    // LoopClosing::ComputeSim3: MapPoints projected with a similarity transformation
    int SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, std::vector<MapPoint *> &vpMatched,
                           int th);

    // brute force inside shared vocabulary nodes (relocalisation, loop detection)
    int SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint *> &vpMapPointMatches);
    int SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12);

    // monocular map initialisation
    int SearchForInitialization(Frame &F1, Frame &F2, std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12,
                                int windowSize = 10);

    // LocalMapping::CreateNewMapPoints: unmatched keypoints of two KeyFrames under the epipolar constraint
    int SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t> > &vMatchedPairs,
                               const bool bOnlyStereo);

    // MapPoints of KF1 and KF2 related by the Sim3 [s12*R12 | t12]
    int SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12, const float &s12, const cv::Mat &R12,
                     const cv::Mat &t12, const float th);

    // project MapPoints into a KeyFrame and merge duplicates
    int Fuse(KeyFrame *pKF, const std::vector<MapPoint *> &vpMapPoints, const float th = 3.0);
    int Fuse(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, float th, std::vector<MapPoint *> &vpReplacePoint);

public:
    static const int TH_LOW;
    static const int TH_HIGH;
    static const int HISTO_LENGTH;

protected:
    bool CheckDistEpipolarLine(const cv::KeyPoint &kp1, const cv::KeyPoint &kp2, const cv::Mat &F12, const KeyFrame *pKF);
    float RadiusByViewingCos(const float &viewCos);
    void ComputeThreeMaxima(std::vector<int> *histo, const int L, int &ind1, int &ind2, int &ind3);

    float mfNNratio;
    bool mbCheckOrientation;
};

}  // namespace ORB_SLAM2

#endif  // ORBMATCHER_H
