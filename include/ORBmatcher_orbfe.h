// ORBmatcher_orbfe.h -- the Hamming part of ORB_SLAM2::ORBmatcher on liborbfe (reference include/ORBmatcher.h:41-106).
//
// The reference's ORBmatcher.h keeps its class declaration (its Search* methods walk Frame / KeyFrame / MapPoint
// pointers and stay host code); what moves to liborbfe is the distance arithmetic underneath them:
//   * ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1647-1663): define it as `return orbfe_DescriptorDistance(a, b);`
//   * the row-band Hamming search of Frame::ComputeStereoMatches (src/Frame.cc:504-549): orbfe::StereoRowBand below
//   * Frame::ComputeStereoMatches as a whole (src/Frame.cc:466-640), SAD refinement and median gate included:
//     orbfe::StereoMatcher below -- no host copy of mvImagePyramid is read
//   * batched best/second-best for the BoW / projection searches: orbfe_match_knn2 / orbfe_match_csr in orbfe.h.
//   * SearchByBoW(KeyFrame*, Frame&, ...) (src/ORBmatcher.cc:159-288) and SearchByBoW(KeyFrame*, KeyFrame*, ...) (:522-655):
//     orbfe::BowMatcher below
//   * Frame::GetFeaturesInArea (src/Frame.cc:327-380) + the candidate loop of SearchByProjection(Frame&, vpMapPoints, th)
//     (src/ORBmatcher.cc:45-129): orbfe::ProjectionSearch below
//   * the other guided searches from their window search on -- SearchByProjection(KeyFrame*, Scw, ...) (:290-403),
//     SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) (:1472-1599), both Fuse overloads (:825-975, :977-1100),
//     SearchBySim3 (:1102-1326), SearchForInitialization (:405-520): further members of orbfe::ProjectionSearch;
//     SearchForTriangulation (:657-823): orbfe::TriangulationMatcher
// INTEGRATION.md shows the two-line patches.
#pragma once
#include <climits>
#include <cmath>
#include <cstring>
#include <map>
#include <utility>
#include <vector>

#include "orbfe.h"
#include "orbfe_context.h"
#include "orbfe_cv.h"

// body for ORBmatcher::DescriptorDistance(const cv::Mat &a, const cv::Mat &b)
static inline int orbfe_DescriptorDistance(const cv::Mat &a, const cv::Mat &b) { return orbfe_hamming(a.ptr(), b.ptr()); }

namespace orbfe {

// Device-side replacement for the loop at src/Frame.cc:504-549.  Inputs are the members Frame already holds
// (mvKeys, mDescriptors, mvKeysRight, mDescriptorsRight, mvScaleFactors, image rows, minD = 0, maxD = mbf/mb);
// outputs are bestDist / bestIdxR per left keypoint, exactly the values the reference's loop leaves in its locals.
// Runs on the calling thread's matcher context (orbfe_context.h); the staging is inside liborbfe.
class StereoRowBand {
public:
    int run(const std::vector<cv::KeyPoint> &kl, const cv::Mat &dl, const std::vector<cv::KeyPoint> &kr, const cv::Mat &dr,
            const std::vector<float> &scaleFactors, int nRows, float minD, float maxD, std::vector<int> &bestDist,
            std::vector<int> &bestIdxR) {
        const int nl = (int)kl.size(), nr = (int)kr.size();
        bestDist.assign(nl, 100);
        bestIdxR.assign(nl, 0);
        if (nl == 0) return ORBFE_OK;
        int rc = ORBFE_OK;
        orbfe_matcher *m = thread_matcher(&rc);
        if (!m) return rc;
        static_assert(sizeof(int) == sizeof(int32_t), "int is 32 bits");
        return orbfe_stereo_rowband_host(m, (const orbfe_keypoint *)kl.data(), dl.data, dl.step, nl, (const orbfe_keypoint *)kr.data(),
                                         dr.data, dr.step, nr, scaleFactors.data(), (int)scaleFactors.size(), nRows, minD, maxD,
                                         bestDist.data(), bestIdxR.data());
    }
};

// Frame::ComputeStereoMatches (src/Frame.cc:466-640) on the device.  `left` / `right` are the liborbfe handles of the
// two extractors that have just processed the stereo pair (ORBextractor::orbfeHandle()); keypoints and descriptors are
// the Frame members; outputs are Frame::mvuRight and Frame::mvDepth.
class StereoMatcher {
public:
    int ComputeStereoMatches(orbfe_extractor *left, orbfe_extractor *right, const std::vector<cv::KeyPoint> &kl, const cv::Mat &dl,
                             const std::vector<cv::KeyPoint> &kr, const cv::Mat &dr, float mbf, float mb,
                             std::vector<float> &mvuRight, std::vector<float> &mvDepth) {
        const int nl = (int)kl.size(), nr = (int)kr.size();
        mvuRight.assign(nl, -1.0f);   // :468-469
        mvDepth.assign(nl, -1.0f);
        if (nl == 0) return ORBFE_OK;
        int rc = ORBFE_OK;
        orbfe_matcher *m = thread_matcher(&rc);
        if (!m) return rc;
        return orbfe_stereo_match_host(m, left, 0, right, 0, (const orbfe_keypoint *)kl.data(), dl.data, dl.step, nl,
                                       (const orbfe_keypoint *)kr.data(), dr.data, dr.step, nr, mbf, mb, mvuRight.data(), mvDepth.data());
    }
};

// ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th) (src/ORBmatcher.cc:45-129) with the
// window query and the Hamming loop on the device.  build() is called where the reference calls AssignFeaturesToGrid
// (src/Frame.cc:96/151/205) with the same members; search() takes what the loop reads from each MapPoint that passes
// :54-58 (mbTrackInView, !isBad()).  All windows are searched in one launch against the associations that exist when the
// call is made; the loop's own assignments (:123) are then replayed on the host in the reference's order, and a query
// whose best or second keypoint was taken by an earlier MapPoint of the same call is searched again, alone, against the
// updated state -- so the outcome is the sequential loop's, association for association.
// ORBmatcher::ComputeThreeMaxima (:1601-1642)
inline void three_maxima(const std::vector<int> *histo, int L, int &ind1, int &ind2, int &ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = (int)histo[i].size();
        if (s > max1) {
            max3 = max2; max2 = max1; max1 = s;
            ind3 = ind2; ind2 = ind1; ind1 = i;
        } else if (s > max2) {
            max3 = max2; max2 = s;
            ind3 = ind2; ind2 = i;
        } else if (s > max3) {
            max3 = s;
            ind3 = i;
        }
    }
    if (max2 < 0.1f * (float)max1) {
        ind2 = -1;
        ind3 = -1;
    } else if (max3 < 0.1f * (float)max1) {
        ind3 = -1;
    }
}

struct ProjectedPoint {
    float projX, projY, projXR;   // mTrackProjX, mTrackProjY, mTrackProjXR
    int level;                    // mnTrackScaleLevel
    float viewCos;                // mTrackViewCos
    const unsigned char *desc;    // GetDescriptor(): 32 bytes
    bool observed = true;         // pMP->Observations() > 0: only then does its keypoint block later MapPoints (:87-89)
};
// One LastFrame MapPoint of SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono) (:1328-1470), after
// the projection the reference computes on the host (:1360-1381).
struct TrackedPoint {
    float u, v;                   // :1370-1371
    float ur;                     // u - mbf * invzc (:1409)
    float radius;                 // th * mvScaleFactors[nLastOctave] (:1381)
    int minLevel, maxLevel;       // :1385-1390: (oct, -1) forward, (0, oct) backward, (oct - 1, oct + 1) otherwise
    float angle;                  // LastFrame.mvKeysUn[i].angle
    const unsigned char *desc;    // pMP->GetDescriptor()
    bool observed;                // pMP->Observations() > 0 (false for the temporal points of UpdateLastFrame)
};
class ProjectionSearch {
public:
    // owning form: its own matcher context (stream + staging) and its own grid, filled by build()
    explicit ProjectionSearch(int device = 0) : dev_(device), m_(nullptr), g_(nullptr), owner_(true) {
        status_ = orbfe_matcher_create(device, &m_);
    }
    // borrowing form: a matcher context of the calling thread and a grid that somebody else keeps alive (the pool of
    // adapters/ORBmatcher_orbfe.cc).  n = keypoints in the grid.  With ownOrigin the cell range of a window is computed
    // from (minX, minY) instead of the bounds the grid was built with (KeyFrame::GetFeaturesInArea, src/KeyFrame.cc:574-586).
    ProjectionSearch(orbfe_matcher *m, orbfe_frame_grid *g, int n, bool ownOrigin = false, float minX = 0, float minY = 0)
        : dev_(0), m_(m), g_(g), owner_(false), n_(n), ownOrigin_(ownOrigin), originX_(minX), originY_(minY) {
        status_ = (m && g) ? ORBFE_OK : ORBFE_ERR_ARG;
    }
    ProjectionSearch(const ProjectionSearch &) = delete;
    ProjectionSearch &operator=(const ProjectionSearch &) = delete;
    ~ProjectionSearch() {
        if (!owner_) return;
        if (g_) orbfe_frame_grid_destroy(g_);
        orbfe_matcher_destroy(m_);
    }
    int status() const { return status_; }   // of the constructor: non-zero when no matcher context could be created
    int build(const std::vector<cv::KeyPoint> &keysUn, const cv::Mat &desc, const std::vector<float> &uRight, float minX, float minY,
              float gridElementWidthInv, float gridElementHeightInv) {
        if (status_ != ORBFE_OK || !owner_) return status_ != ORBFE_OK ? status_ : ORBFE_ERR_ARG;
        const int n = (int)keysUn.size();
        std::vector<float> kx(n), ky(n);
        std::vector<int32_t> oct(n);
        std::vector<unsigned char> d((size_t)n * 32);
        for (int i = 0; i < n; i++) {
            kx[i] = keysUn[i].pt.x;
            ky[i] = keysUn[i].pt.y;
            oct[i] = keysUn[i].octave;
            std::memcpy(&d[(size_t)i * 32], desc.ptr(i), 32);
        }
        n_ = n;
        const float *ur = uRight.empty() ? nullptr : uRight.data();
        if (g_)   // keep the device block: one copy on the context's stream instead of a hipFree / hipMalloc pair
            return orbfe_frame_grid_update(m_, g_, kx.data(), ky.data(), oct.data(), d.data(), ur, n, minX, minY, gridElementWidthInv,
                                           gridElementHeightInv);
        return orbfe_frame_grid_create(dev_, kx.data(), ky.data(), oct.data(), d.data(), ur, n, minX, minY, gridElementWidthInv,
                                       gridElementHeightInv, &g_);
    }
    // occupied[idx] != 0 where F.mvpMapPoints[idx] is set and has Observations() > 0 (:87-89); updated with the keypoints
    // this call associates (the MapPoints handed to this function are map points, i.e. observed).  matchIdx[i] is the
    // keypoint MapPoint i was written to (:123) or -1.  Returns a status; *nmatches is the reference's return value.
    int search(const std::vector<ProjectedPoint> &pts, const std::vector<float> &scaleFactors, float th, float nnRatio,
               std::vector<unsigned char> &occupied, std::vector<int> &matchIdx, int *nmatches) {
        const int nq = (int)pts.size();
        matchIdx.assign(nq, -1);
        *nmatches = 0;
        if (nq == 0 || n_ == 0) return ORBFE_OK;
        if ((int)occupied.size() != n_) return ORBFE_ERR_ARG;
        const bool bFactor = th != 1.0f;                                       // :49
        std::vector<orbfe_area_query> q(nq);
        std::vector<unsigned char> qd((size_t)nq * 32);
        for (int i = 0; i < nq; i++) {
            float r = pts[i].viewCos > 0.998 ? 2.5f : 4.0f;                    // RadiusByViewingCos, :131-137
            if (bFactor) r *= th;
            const float rs = r * scaleFactors[pts[i].level];                   // :69 and :94
            q[i].x = pts[i].projX;
            q[i].y = pts[i].projY;
            q[i].r = rs;
            q[i].min_level = pts[i].level - 1;
            q[i].max_level = pts[i].level;
            q[i].proj_xr = pts[i].projXR;
            q[i].er_max = rs;
            q[i].pad = 0;
            for (int b = 0; b < 32; b++) qd[(size_t)i * 32 + b] = pts[i].desc[b];
        }
        std::vector<orbfe_area_result> res(nq);
        orbfe_area_options opt = orbfe_area_options();
        opt.blocked = occupied.data();
        int rc = area(q.data(), qd.data(), nq, opt, res.data());
        if (rc != ORBFE_OK) return rc;
        std::vector<unsigned char> taken(n_, 0);                               // associations made by this call
        for (int i = 0; i < nq; i++) {
            orbfe_area_result r = res[i];
            if ((r.best_idx >= 0 && taken[r.best_idx]) || (r.best_idx2 >= 0 && taken[r.best_idx2])) {
                rc = area(&q[i], &qd[(size_t)i * 32], 1, opt, &r);
                if (rc != ORBFE_OK) return rc;
                requeried_++;
            }
            if (r.best_dist <= 100) {                                          // TH_HIGH, :118
                if (r.best_level == r.best_level2 && r.best_dist > nnRatio * r.best_dist2) continue;   // :120
                if (pts[i].observed) {                                         // :123, seen by :87-89 of later MapPoints
                    occupied[r.best_idx] = 1;
                    taken[r.best_idx] = 1;
                }
                matchIdx[i] = r.best_idx;
                (*nmatches)++;
            }
        }
        return ORBFE_OK;
    }
    // SearchByProjection(CurrentFrame, LastFrame, th, bMono) (:1328-1470) from the window search on: best match only,
    // accepted up to TH_HIGH, then the rotation histogram.  build() was called with CurrentFrame's members; curAngles =
    // CurrentFrame.mvKeysUn[*].angle; occupied as in search().  assigned[idx] = index into pts of the MapPoint that
    // CurrentFrame.mvpMapPoints[idx] ends up with, or -1 (a keypoint taken by an unobserved MapPoint can be overwritten by
    // a later one, as in the reference); *nmatches is the reference's return value (it counts such a keypoint twice).
    // removed (optional): the keypoints the rotation histogram un-matched (the reference writes NULL there, :1462).
    int searchLastFrame(const std::vector<TrackedPoint> &pts, const std::vector<float> &curAngles, bool checkOrientation,
                        std::vector<unsigned char> &occupied, std::vector<int> &assigned, int *nmatches,
                        std::vector<int> *removed = nullptr) {
        if (removed) removed->clear();
        const int nq = (int)pts.size();
        assigned.assign(n_, -1);
        *nmatches = 0;
        if (nq == 0 || n_ == 0) return ORBFE_OK;
        if ((int)occupied.size() != n_ || (int)curAngles.size() != n_) return ORBFE_ERR_ARG;
        std::vector<orbfe_area_query> q(nq);
        std::vector<unsigned char> qd((size_t)nq * 32);
        for (int i = 0; i < nq; i++) {
            q[i].x = pts[i].u;
            q[i].y = pts[i].v;
            q[i].r = pts[i].radius;
            q[i].min_level = pts[i].minLevel;
            q[i].max_level = pts[i].maxLevel;
            q[i].proj_xr = pts[i].ur;
            q[i].er_max = pts[i].radius;                                       // :1411
            q[i].pad = 0;
            for (int b = 0; b < 32; b++) qd[(size_t)i * 32 + b] = pts[i].desc[b];
        }
        std::vector<orbfe_area_result> res(nq);
        orbfe_area_options opt = orbfe_area_options();
        opt.blocked = occupied.data();
        int rc = area(q.data(), qd.data(), nq, opt, res.data());
        if (rc != ORBFE_OK) return rc;
        const int HISTO_LENGTH = 30;
        std::vector<int> rotHist[HISTO_LENGTH];
        const float factor = 1.0f / HISTO_LENGTH;
        std::vector<unsigned char> taken(n_, 0);
        int n = 0;
        for (int i = 0; i < nq; i++) {
            orbfe_area_result r = res[i];
            if (r.best_idx >= 0 && taken[r.best_idx]) {                        // only the best match is used here
                rc = area(&q[i], &qd[(size_t)i * 32], 1, opt, &r);
                if (rc != ORBFE_OK) return rc;
                requeried_++;
            }
            if (r.best_dist > 100) continue;                                   // TH_HIGH, :1426 (covers vIndices2.empty())
            assigned[r.best_idx] = i;                                          // :1428
            n++;
            if (pts[i].observed) {
                occupied[r.best_idx] = 1;
                taken[r.best_idx] = 1;
            }
            if (checkOrientation) {                                            // :1431-1441
                float rot = pts[i].angle - curAngles[r.best_idx];
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(r.best_idx);
            }
        }
        if (checkOrientation) {                                                // :1448-1467
            int ind1 = -1, ind2 = -1, ind3 = -1;
            three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
            for (int i = 0; i < HISTO_LENGTH; i++) {
                if (i == ind1 || i == ind2 || i == ind3) continue;
                for (size_t j = 0; j < rotHist[i].size(); j++) {
                    assigned[rotHist[i][j]] = -1;
                    occupied[rotHist[i][j]] = 0;
                    if (removed) removed->push_back(rotHist[i][j]);
                    n--;
                }
            }
        }
        *nmatches = n;
        return ORBFE_OK;
    }
    // ---- the remaining window searches.  One WindowPoint = one MapPoint (or F1 keypoint) that reached the window search of
    // its routine, with the values the reference has computed on the host by then.
    struct WindowPoint {
        float u, v;                   // projection (or vbPrevMatched[i1])
        float ur;                     // u - bf * invz, used by the Fuse gate only
        float radius;                 // th * mvScaleFactors[nPredictedLevel] (or windowSize)
        int level;                    // nPredictedLevel (or level1)
        float angle;                  // the keypoint angle the rotation histogram uses for this point
        const unsigned char *desc;    // pMP->GetDescriptor() (or F1.mDescriptors.row(i1))
        int index;                    // SearchBySim3 / SearchForInitialization: the keypoint of the other frame it belongs to
    };
    // SearchByProjection(KeyFrame *pKF, cv::Mat Scw, vpPoints, vpMatched, th) from :359 on; build() was called with pKF's
    // members.  matched[idx] != 0 where vpMatched[idx] is set (:375), updated as :396 does; matchIdx[i] = keypoint or -1.
    int searchSim3Projection(const std::vector<WindowPoint> &pts, std::vector<unsigned char> &matched, std::vector<int> &matchIdx,
                             int *nmatches) {
        return bestWithReplay(pts, -1, 0, 50 /* TH_LOW, :394 */, true, nullptr, 0, matched, matchIdx, nmatches);
    }
    // SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, sAlreadyFound, th, ORBdist) from :1525 on; build() was called
    // with CurrentFrame's members, curAngles = CurrentFrame.mvKeysUn[*].angle, pts[i].angle = pKF->mvKeysUn[i].angle.
    // occupied[i2] != 0 where CurrentFrame.mvpMapPoints[i2] is set (:1541); assigned[i2] = index into pts or -1.
    int searchRelocalization(const std::vector<WindowPoint> &pts, const std::vector<float> &curAngles, int ORBdist,
                             bool checkOrientation, std::vector<unsigned char> &occupied, std::vector<int> &assigned, int *nmatches,
                             std::vector<int> *removed = nullptr) {
        if (removed) removed->clear();
        assigned.assign(n_, -1);
        *nmatches = 0;
        if ((int)curAngles.size() != n_) return ORBFE_ERR_ARG;
        std::vector<int> matchIdx;
        int n = 0;
        const int rc = bestWithReplay(pts, -1, +1, ORBdist, true, nullptr, 0, occupied, matchIdx, &n);
        if (rc != ORBFE_OK) return rc;
        const int HISTO_LENGTH = 30;
        std::vector<int> rotHist[HISTO_LENGTH];
        const float factor = 1.0f / HISTO_LENGTH;
        for (size_t i = 0; i < pts.size(); i++) {
            if (matchIdx[i] < 0) continue;
            assigned[matchIdx[i]] = (int)i;                                    // :1557
            if (checkOrientation) {                                            // :1560-1570
                float rot = pts[i].angle - curAngles[matchIdx[i]];
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(matchIdx[i]);
            }
        }
        if (checkOrientation) {                                                // :1577-1596
            int ind1 = -1, ind2 = -1, ind3 = -1;
            three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
            for (int i = 0; i < HISTO_LENGTH; i++) {
                if (i == ind1 || i == ind2 || i == ind3) continue;
                for (size_t j = 0; j < rotHist[i].size(); j++) {
                    assigned[rotHist[i][j]] = -1;
                    occupied[rotHist[i][j]] = 0;
                    if (removed) removed->push_back(rotHist[i][j]);
                    n--;
                }
            }
        }
        *nmatches = n;
        return ORBFE_OK;
    }
    // The selection part of Fuse(KeyFrame *pKF, vpMapPoints, th) (:889-952; pass pKF->mvInvLevelSigma2) and of
    // Fuse(KeyFrame *pKF, Scw, vpPoints, th, vpReplacePoint) (:1048-1082; pass an empty vector: that overload has no
    // reprojection gate).  bestIdx[i] = the keypoint MapPoint i is fused into, or -1; what :954-970 / :1084-1095 then do to
    // the map stays with the caller.  No candidate is skipped because of earlier fusions, so one launch serves all points.
    int fuseSelect(const std::vector<WindowPoint> &pts, const std::vector<float> &invLevelSigma2, std::vector<int> &bestIdx,
                   int *nFused) {
        std::vector<unsigned char> none;
        return bestWithReplay(pts, -1, 0, 50 /* TH_LOW */, false, invLevelSigma2.empty() ? nullptr : invLevelSigma2.data(),
                              (int)invLevelSigma2.size(), none, bestIdx, nFused);
    }
    // SearchBySim3 (:1102-1326) from the window searches on.  kf1 / kf2 were built with pKF1's / pKF2's members; pts1 are the
    // MapPoints of KF1 projected into KF2 (index = i1), pts2 those of KF2 projected into KF1 (index = i2).
    // match12[i1] = idx2 where both directions agree (:1307-1323), else -1.
    static int SearchBySim3(ProjectionSearch &kf1, ProjectionSearch &kf2, const std::vector<WindowPoint> &pts1,
                            const std::vector<WindowPoint> &pts2, std::vector<int> &match12, int *nFound) {
        std::vector<unsigned char> none;
        std::vector<int> b1, b2, vnMatch1(kf1.n_, -1), vnMatch2(kf2.n_, -1);
        int dummy = 0;
        int rc = kf2.bestWithReplay(pts1, -1, 0, 100 /* TH_HIGH, :1221 */, false, nullptr, 0, none, b1, &dummy);
        if (rc == ORBFE_OK) rc = kf1.bestWithReplay(pts2, -1, 0, 100, false, nullptr, 0, none, b2, &dummy);
        if (rc != ORBFE_OK) return rc;
        for (size_t a = 0; a < pts1.size(); a++)
            if (b1[a] >= 0) vnMatch1[pts1[a].index] = b1[a];
        for (size_t a = 0; a < pts2.size(); a++)
            if (b2[a] >= 0) vnMatch2[pts2[a].index] = b2[a];
        match12.assign(kf1.n_, -1);
        *nFound = 0;
        for (int i1 = 0; i1 < kf1.n_; i1++) {
            const int idx2 = vnMatch1[i1];
            if (idx2 >= 0 && vnMatch2[idx2] == i1) {
                match12[i1] = idx2;
                (*nFound)++;
            }
        }
        return ORBFE_OK;
    }
    // SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (:405-520); build() was called with F2's members,
    // angles2 = F2.mvKeysUn[*].angle, pts2xy = F2.mvKeysUn[*].pt.  pts holds the level-0 keypoints of F1 in index order
    // (index = i1, u/v = vbPrevMatched[i1], radius = windowSize, level = 0).  All windows are searched in one launch with
    // vMatchedDistance = INT_MAX; the loop is then replayed in order, and a keypoint whose best or second candidate has
    // meanwhile been matched at a distance <= its own (:444) is searched again against the current vMatchedDistance.
    int searchForInitialization(const std::vector<WindowPoint> &pts, int n1, const std::vector<float> &angles2, float nnRatio,
                                bool checkOrientation, std::vector<int> &vnMatches12, int *nmatches) {
        vnMatches12.assign(n1, -1);
        *nmatches = 0;
        const int nq = (int)pts.size();
        if (nq == 0 || n_ == 0) return ORBFE_OK;
        if ((int)angles2.size() != n_) return ORBFE_ERR_ARG;
        std::vector<orbfe_area_query> q(nq);
        std::vector<unsigned char> qd((size_t)nq * 32);
        for (int i = 0; i < nq; i++) {
            q[i].x = pts[i].u; q[i].y = pts[i].v; q[i].r = pts[i].radius;
            q[i].min_level = pts[i].level; q[i].max_level = pts[i].level;   // :425
            q[i].proj_xr = 0; q[i].er_max = -1.0f; q[i].pad = 0;
            for (int b = 0; b < 32; b++) qd[(size_t)i * 32 + b] = pts[i].desc[b];
        }
        std::vector<orbfe_area_result> res(nq);
        orbfe_area_options opt = orbfe_area_options();
        int rc = area(q.data(), qd.data(), nq, opt, res.data());
        if (rc != ORBFE_OK) return rc;
        const int HISTO_LENGTH = 30, TH_LOW = 50;
        std::vector<int> rotHist[HISTO_LENGTH];
        const float factor = 1.0f / HISTO_LENGTH;
        std::vector<int32_t> vMatchedDistance(n_, INT_MAX);
        std::vector<int> vnMatches21(n_, -1);
        int n = 0;
        for (int i = 0; i < nq; i++) {
            orbfe_area_result r = res[i];
            if ((r.best_idx >= 0 && vMatchedDistance[r.best_idx] <= r.best_dist) ||
                (r.best_idx2 >= 0 && vMatchedDistance[r.best_idx2] <= r.best_dist2)) {
                opt.dist_limit = vMatchedDistance.data();
                rc = area(&q[i], &qd[(size_t)i * 32], 1, opt, &r);
                if (rc != ORBFE_OK) return rc;
                requeried_++;
            }
            const int bestDist = r.best_idx < 0 ? INT_MAX : r.best_dist;       // the loop starts from INT_MAX (:432-433)
            const int bestDist2 = r.best_idx2 < 0 ? INT_MAX : r.best_dist2;
            if (bestDist <= TH_LOW && bestDist < (float)bestDist2 * nnRatio) { // :459-461
                const int i1 = pts[i].index, bestIdx2 = r.best_idx;
                if (vnMatches21[bestIdx2] >= 0) {
                    vnMatches12[vnMatches21[bestIdx2]] = -1;
                    n--;
                }
                vnMatches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                n++;
                if (checkOrientation) {
                    float rot = pts[i].angle - angles2[bestIdx2];
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)std::round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back(i1);
                }
            }
        }
        if (checkOrientation) {                                                // :489-512
            int ind1 = -1, ind2 = -1, ind3 = -1;
            three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
            for (int i = 0; i < HISTO_LENGTH; i++) {
                if (i == ind1 || i == ind2 || i == ind3) continue;
                for (size_t j = 0; j < rotHist[i].size(); j++) {
                    const int idx1 = rotHist[i][j];
                    if (vnMatches12[idx1] >= 0) {
                        vnMatches12[idx1] = -1;
                        n--;
                    }
                }
            }
        }
        *nmatches = n;   // the caller then updates vbPrevMatched from vnMatches12 (:514-517)
        return ORBFE_OK;
    }
    long requeried() const { return requeried_; }

private:
    // best candidate per point in [level + dLo, level + dHi] with the reference's strict '<' scan, accepted up to maxDist;
    // blockAccepted: an accepted keypoint is skipped by later points (vpMatched / mvpMapPoints written inside the loop).
    int bestWithReplay(const std::vector<WindowPoint> &pts, int dLo, int dHi, int maxDist, bool blockAccepted,
                       const float *invLevelSigma2, int nlevels, std::vector<unsigned char> &occupied, std::vector<int> &matchIdx,
                       int *nmatches) {
        const int nq = (int)pts.size();
        matchIdx.assign(nq, -1);
        *nmatches = 0;
        if (nq == 0 || n_ == 0) return ORBFE_OK;
        if (blockAccepted && (int)occupied.size() != n_) return ORBFE_ERR_ARG;
        std::vector<orbfe_area_query> q(nq);
        std::vector<unsigned char> qd((size_t)nq * 32);
        for (int i = 0; i < nq; i++) {
            q[i].x = pts[i].u; q[i].y = pts[i].v; q[i].r = pts[i].radius;
            q[i].min_level = pts[i].level + dLo; q[i].max_level = pts[i].level + dHi;
            q[i].proj_xr = pts[i].ur; q[i].er_max = -1.0f; q[i].pad = 0;
            for (int b = 0; b < 32; b++) qd[(size_t)i * 32 + b] = pts[i].desc[b];
        }
        orbfe_area_options opt = orbfe_area_options();
        opt.blocked = blockAccepted ? occupied.data() : nullptr;
        opt.inv_level_sigma2 = invLevelSigma2;
        opt.nlevels = nlevels;
        opt.chi2_mono = 5.99;
        opt.chi2_stereo = 7.8;
        std::vector<orbfe_area_result> res(nq);
        int rc = area(q.data(), qd.data(), nq, opt, res.data());
        if (rc != ORBFE_OK) return rc;
        std::vector<unsigned char> taken(blockAccepted ? n_ : 0, 0);
        for (int i = 0; i < nq; i++) {
            orbfe_area_result r = res[i];
            if (blockAccepted && r.best_idx >= 0 && taken[r.best_idx]) {
                rc = area(&q[i], &qd[(size_t)i * 32], 1, opt, &r);
                if (rc != ORBFE_OK) return rc;
                requeried_++;
            }
            if (r.best_idx < 0 || r.best_dist > maxDist) continue;
            matchIdx[i] = r.best_idx;
            (*nmatches)++;
            if (blockAccepted) {
                occupied[r.best_idx] = 1;
                taken[r.best_idx] = 1;
            }
        }
        return ORBFE_OK;
    }
    // one launch over nq windows with this object's query origin
    int area(const orbfe_area_query *q, const unsigned char *qd, int nq, orbfe_area_options opt, orbfe_area_result *res) {
        if (status_ != ORBFE_OK) return status_;
        if (!g_) return ORBFE_ERR_ARG;
        opt.has_query_origin = ownOrigin_ ? 1 : 0;
        opt.query_min_x = originX_;
        opt.query_min_y = originY_;
        return orbfe_search_area_ex(m_, g_, q, qd, nq, &opt, res);
    }
    int dev_;
    orbfe_matcher *m_;
    orbfe_frame_grid *g_;
    bool owner_;
    int status_ = ORBFE_OK;
    int n_ = 0;
    bool ownOrigin_ = false;
    float originX_ = 0, originY_ = 0;
    long requeried_ = 0;
};

// ORBmatcher::SearchByBoW, both overloads.  Side 1 is the KeyFrame whose MapPoints are looked up (pKF / pKF1), side 2 the
// Frame (or pKF2).  fv1 / fv2 are the DBoW2::FeatureVector maps (node id -> feature indices); valid1[i] != 0 where side 1
// has a MapPoint that is not bad (:191-197, :558-562); blocked2 (may be empty) marks side-2 features that cannot be matched
// (KeyFrame / KeyFrame: no MapPoint or a bad one, :574-580).  maxDist = TH_LOW for the Frame overload (<=, :228) and
// TH_LOW - 1 for the KeyFrame overload (<, :598).  The merge walk (:180-264) and the rotation histogram with
// ComputeThreeMaxima (:236-246, :267-285, :1601-1642) run here on the host; the distance loops run on the device, one
// wavefront per shared node (orbfe_search_by_bow).  match12[i] = side-2 feature matched to side-1 feature i, or -1;
// *nmatches is the reference's return value.
class BowMatcher {
public:
    typedef std::map<unsigned int, std::vector<unsigned int> > FeatureVector;   // DBoW2::FeatureVector
    explicit BowMatcher(int device = 0) : m_(nullptr), owner_(true) { status_ = orbfe_matcher_create(device, &m_); }
    explicit BowMatcher(orbfe_matcher *borrowed) : m_(borrowed), owner_(false) { status_ = borrowed ? ORBFE_OK : ORBFE_ERR_ARG; }
    BowMatcher(const BowMatcher &) = delete;
    BowMatcher &operator=(const BowMatcher &) = delete;
    ~BowMatcher() { if (owner_) orbfe_matcher_destroy(m_); }
    int status() const { return status_; }
    int SearchByBoW(const FeatureVector &fv1, const cv::Mat &desc1, const std::vector<cv::KeyPoint> &keys1,
                    const std::vector<unsigned char> &valid1, const FeatureVector &fv2, const cv::Mat &desc2,
                    const std::vector<cv::KeyPoint> &keys2, const std::vector<unsigned char> &blocked2, int maxDist, float nnRatio,
                    bool checkOrientation, std::vector<int> &match12, int *nmatches) {
        const int n1 = (int)keys1.size(), n2 = (int)keys2.size();
        match12.assign(n1, -1);
        *nmatches = 0;
        if (status_ != ORBFE_OK) return status_;
        // the merge walk of :180-264: nodes present in both maps, in ascending node id
        std::vector<int32_t> off1(1, 0), off2(1, 0), idx1, idx2;
        FeatureVector::const_iterator it1 = fv1.begin(), it2 = fv2.begin();
        while (it1 != fv1.end() && it2 != fv2.end()) {
            if (it1->first == it2->first) {
                idx1.insert(idx1.end(), it1->second.begin(), it1->second.end());
                idx2.insert(idx2.end(), it2->second.begin(), it2->second.end());
                off1.push_back((int32_t)idx1.size());
                off2.push_back((int32_t)idx2.size());
                ++it1;
                ++it2;
            } else if (it1->first < it2->first) {
                it1 = fv1.lower_bound(it2->first);
            } else {
                it2 = fv2.lower_bound(it1->first);
            }
        }
        if (off1.size() == 1 || n1 == 0 || n2 == 0) return ORBFE_OK;
        std::vector<unsigned char> d1((size_t)n1 * 32), d2((size_t)n2 * 32);
        for (int i = 0; i < n1; i++) std::memcpy(&d1[(size_t)i * 32], desc1.ptr(i), 32);
        for (int i = 0; i < n2; i++) std::memcpy(&d2[(size_t)i * 32], desc2.ptr(i), 32);
        const int rc = orbfe_search_by_bow(m_, d1.data(), valid1.empty() ? nullptr : valid1.data(), n1, d2.data(),
                                           blocked2.empty() ? nullptr : blocked2.data(), n2, off1.data(), idx1.data(), off2.data(),
                                           idx2.data(), (int)off1.size() - 1, maxDist, nnRatio, match12.data());
        if (rc != ORBFE_OK) return rc;
        // rotation consistency (:236-246, :267-285)
        const int HISTO_LENGTH = 30;
        std::vector<int> rotHist[HISTO_LENGTH];
        const float factor = 1.0f / HISTO_LENGTH;
        int n = 0;
        for (int i = 0; i < n1; i++) {
            if (match12[i] < 0) continue;
            n++;
            if (!checkOrientation) continue;
            float rot = keys1[i].angle - keys2[match12[i]].angle;
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)std::round(rot * factor);
            if (bin == HISTO_LENGTH) bin = 0;
            rotHist[bin].push_back(i);
        }
        if (checkOrientation) {
            int ind1 = -1, ind2 = -1, ind3 = -1;
            ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
            for (int i = 0; i < HISTO_LENGTH; i++) {
                if (i == ind1 || i == ind2 || i == ind3) continue;
                for (size_t j = 0; j < rotHist[i].size(); j++) {
                    match12[rotHist[i][j]] = -1;
                    n--;
                }
            }
        }
        *nmatches = n;
        return ORBFE_OK;
    }
    static void ComputeThreeMaxima(const std::vector<int> *histo, int L, int &ind1, int &ind2, int &ind3) {
        three_maxima(histo, L, ind1, ind2, ind3);
    }

private:
    orbfe_matcher *m_;
    bool owner_;
    int status_ = ORBFE_OK;
};

// ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:657-823).  fv1 / fv2: pKF1->mFeatVec / pKF2->mFeatVec; hasMp1 / hasMp2:
// GetMapPoint(idx) != NULL; uRight1 / uRight2: mvuRight; keys1 / keys2: mvKeysUn; F12 as the reference's 3x3 CV_32F matrix in
// row-major order; ex, ey: the epipole the reference computes at :664-670; scaleFactors2 / levelSigma2: pKF2's tables.
class TriangulationMatcher {
public:
    typedef std::map<unsigned int, std::vector<unsigned int> > FeatureVector;
    explicit TriangulationMatcher(int device = 0) : m_(nullptr), owner_(true) { status_ = orbfe_matcher_create(device, &m_); }
    explicit TriangulationMatcher(orbfe_matcher *borrowed) : m_(borrowed), owner_(false) { status_ = borrowed ? ORBFE_OK : ORBFE_ERR_ARG; }
    TriangulationMatcher(const TriangulationMatcher &) = delete;
    TriangulationMatcher &operator=(const TriangulationMatcher &) = delete;
    ~TriangulationMatcher() { if (owner_) orbfe_matcher_destroy(m_); }
    int status() const { return status_; }
    int SearchForTriangulation(const FeatureVector &fv1, const cv::Mat &desc1, const std::vector<cv::KeyPoint> &keys1,
                               const std::vector<unsigned char> &hasMp1, const std::vector<float> &uRight1, const FeatureVector &fv2,
                               const cv::Mat &desc2, const std::vector<cv::KeyPoint> &keys2, const std::vector<unsigned char> &hasMp2,
                               const std::vector<float> &uRight2, const float F12[9], float ex, float ey,
                               const std::vector<float> &scaleFactors2, const std::vector<float> &levelSigma2, bool bOnlyStereo,
                               bool checkOrientation, std::vector<std::pair<size_t, size_t> > &vMatchedPairs, int *nmatches) {
        const int n1 = (int)keys1.size(), n2 = (int)keys2.size();
        vMatchedPairs.clear();
        *nmatches = 0;
        if (status_ != ORBFE_OK) return status_;
        std::vector<int32_t> off1(1, 0), off2(1, 0), idx1, idx2;                // the merge walk of :691-789
        FeatureVector::const_iterator it1 = fv1.begin(), it2 = fv2.begin();
        while (it1 != fv1.end() && it2 != fv2.end()) {
            if (it1->first == it2->first) {
                idx1.insert(idx1.end(), it1->second.begin(), it1->second.end());
                idx2.insert(idx2.end(), it2->second.begin(), it2->second.end());
                off1.push_back((int32_t)idx1.size());
                off2.push_back((int32_t)idx2.size());
                ++it1;
                ++it2;
            } else if (it1->first < it2->first) {
                it1 = fv1.lower_bound(it2->first);
            } else {
                it2 = fv2.lower_bound(it1->first);
            }
        }
        if (off1.size() == 1 || n1 == 0 || n2 == 0) return ORBFE_OK;
        std::vector<unsigned char> d1((size_t)n1 * 32), d2((size_t)n2 * 32);
        std::vector<float> xy1((size_t)n1 * 2), xy2((size_t)n2 * 2);
        std::vector<int32_t> oct2(n2);
        for (int i = 0; i < n1; i++) {
            for (int b = 0; b < 32; b++) d1[(size_t)i * 32 + b] = desc1.ptr(i)[b];
            xy1[2 * i] = keys1[i].pt.x; xy1[2 * i + 1] = keys1[i].pt.y;
        }
        for (int i = 0; i < n2; i++) {
            for (int b = 0; b < 32; b++) d2[(size_t)i * 32 + b] = desc2.ptr(i)[b];
            xy2[2 * i] = keys2[i].pt.x; xy2[2 * i + 1] = keys2[i].pt.y;
            oct2[i] = keys2[i].octave;
        }
        std::vector<int32_t> vMatches12(n1, -1);
        const int rc = orbfe_search_for_triangulation(m_, d1.data(), hasMp1.data(), uRight1.data(), xy1.data(), n1, d2.data(),
                                                      hasMp2.data(), uRight2.data(), xy2.data(), oct2.data(), n2, off1.data(),
                                                      idx1.data(), off2.data(), idx2.data(), (int)off1.size() - 1, F12, ex, ey,
                                                      scaleFactors2.data(), levelSigma2.data(), (int)scaleFactors2.size(),
                                                      bOnlyStereo ? 1 : 0, vMatches12.data());
        if (rc != ORBFE_OK) return rc;
        const int HISTO_LENGTH = 30;
        std::vector<int> rotHist[HISTO_LENGTH];
        const float factor = 1.0f / HISTO_LENGTH;
        int n = 0;
        // the histogram is filled in the order the reference's loop visits the side-1 features (node by node); the bins'
        // contents are only counted and cleared, so index order gives the same result
        for (int i = 0; i < n1; i++) {
            if (vMatches12[i] < 0) continue;
            n++;
            if (!checkOrientation) continue;
            float rot = keys1[i].angle - keys2[vMatches12[i]].angle;           // :764-774
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)std::round(rot * factor);
            if (bin == HISTO_LENGTH) bin = 0;
            rotHist[bin].push_back(i);
        }
        if (checkOrientation) {                                                // :791-810
            int ind1 = -1, ind2 = -1, ind3 = -1;
            three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
            for (int i = 0; i < HISTO_LENGTH; i++) {
                if (i == ind1 || i == ind2 || i == ind3) continue;
                for (size_t j = 0; j < rotHist[i].size(); j++) {
                    vMatches12[rotHist[i][j]] = -1;
                    n--;
                }
            }
        }
        vMatchedPairs.reserve(n);                                              // :812-820
        for (int i = 0; i < n1; i++)
            if (vMatches12[i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)vMatches12[i]));
        *nmatches = n;
        return ORBFE_OK;
    }

private:
    orbfe_matcher *m_;
    bool owner_;
    int status_ = ORBFE_OK;
};

}  // namespace orbfe
