// Out-of-line members of the stand-in classes (tests/cpp/stubs/*.h).  Test infrastructure.
#include "Frame.h"
#include "KeyFrame.h"
#include "MapPoint.h"

namespace ORB_SLAM2 {

float Frame::fx, Frame::fy, Frame::cx, Frame::cy;
float Frame::mfGridElementWidthInv, Frame::mfGridElementHeightInv;
float Frame::mnMinX, Frame::mnMaxX, Frame::mnMinY, Frame::mnMaxY;

// src/MapPoint.cc:385-417
template <class T>
static int predict(float maxDistance, float currentDist, T *p) {
    const float ratio = maxDistance / currentDist;
    int nScale = (int)std::ceil(std::log(ratio) / p->mfLogScaleFactor);
    if (nScale < 0)
        nScale = 0;
    else if (nScale >= p->mnScaleLevels)
        nScale = p->mnScaleLevels - 1;
    return nScale;
}
int MapPoint::PredictScale(const float &currentDist, KeyFrame *pKF) { return predict(mfMaxDistance, currentDist, pKF); }
int MapPoint::PredictScale(const float &currentDist, Frame *pF) { return predict(mfMaxDistance, currentDist, pF); }

// src/MapPoint.cc:98-109
void MapPoint::AddObservation(KeyFrame *pKF, size_t idx) {
    if (mObservations.count(pKF)) return;
    mObservations[pKF] = idx;
    nObs += pKF->mvuRight[idx] >= 0 ? 2 : 1;
}

// src/MapPoint.cc:177-215.  The reference then recomputes the surviving point's descriptor (:212); the stand-in inverts it,
// so a matcher that read a survivor's descriptor at the wrong moment would be caught.
void MapPoint::Replace(MapPoint *pMP) {
    if (pMP->mnId == this->mnId) return;
    std::map<KeyFrame *, size_t> obs = mObservations;
    mObservations.clear();
    mbBad = true;
    mpReplaced = pMP;
    for (std::map<KeyFrame *, size_t>::iterator mit = obs.begin(); mit != obs.end(); ++mit) {
        KeyFrame *pKF = mit->first;
        if (!pMP->IsInKeyFrame(pKF)) {
            pKF->ReplaceMapPointMatch(mit->second, pMP);
            pMP->AddObservation(pKF, mit->second);
        } else {
            pKF->EraseMapPointMatch(mit->second);
        }
    }
    for (int b = 0; b < 32; b++) pMP->mDescriptor.ptr()[b] = (unsigned char)~pMP->mDescriptor.ptr()[b];
}

}  // namespace ORB_SLAM2
