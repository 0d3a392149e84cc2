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
// INTEGRATION.md shows the two-line patches.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <map>
#include <vector>

#include "orbfe.h"
#include "orbfe_cv.h"

// body for ORBmatcher::DescriptorDistance(const cv::Mat &a, const cv::Mat &b)
static inline int orbfe_DescriptorDistance(const cv::Mat &a, const cv::Mat &b) { return orbfe_hamming(a.ptr(), b.ptr()); }

namespace orbfe {

// Device-side replacement for the loop at src/Frame.cc:504-549.  Inputs are the members Frame already holds
// (mvKeys, mDescriptors, mvKeysRight, mDescriptorsRight, mvScaleFactors, image rows, minD = 0, maxD = mbf/mb);
// outputs are bestDist / bestIdxR per left keypoint, exactly the values the reference's loop leaves in its locals.
class StereoRowBand {
public:
    explicit StereoRowBand(int device = 0) : m_(nullptr) { orbfe_matcher_create(device, &m_); }
    ~StereoRowBand() {
        for (void *p : bufs_) if (p) (void)hipFree(p);
        orbfe_matcher_destroy(m_);
    }
    int run(const std::vector<cv::KeyPoint> &kl, const cv::Mat &dl, const std::vector<cv::KeyPoint> &kr, const cv::Mat &dr,
            const std::vector<float> &scaleFactors, int nRows, float minD, float maxD, std::vector<int> &bestDist,
            std::vector<int> &bestIdxR) {
        const int nl = (int)kl.size(), nr = (int)kr.size();
        bestDist.assign(nl, 100);
        bestIdxR.assign(nl, 0);
        if (nl == 0) return ORBFE_OK;
        const size_t need[6] = {(size_t)nl * 28, (size_t)nl * 32, (size_t)(nr ? nr : 1) * 28, (size_t)(nr ? nr : 1) * 32,
                                (size_t)nl * 4, (size_t)nl * 4};
        for (int i = 0; i < 6; i++)
            if (need[i] > size_[i]) {
                if (bufs_[i]) (void)hipFree(bufs_[i]);
                if (hipMalloc(&bufs_[i], need[i]) != hipSuccess) return ORBFE_ERR_DEVICE;
                size_[i] = need[i];
            }
        (void)hipMemcpy(bufs_[0], kl.data(), need[0], hipMemcpyHostToDevice);
        for (int i = 0; i < nl; i++) (void)hipMemcpy((char *)bufs_[1] + (size_t)i * 32, dl.ptr(i), 32, hipMemcpyHostToDevice);
        if (nr) {
            (void)hipMemcpy(bufs_[2], kr.data(), (size_t)nr * 28, hipMemcpyHostToDevice);
            for (int i = 0; i < nr; i++) (void)hipMemcpy((char *)bufs_[3] + (size_t)i * 32, dr.ptr(i), 32, hipMemcpyHostToDevice);
        }
        int rc = orbfe_stereo_rowband(m_, (const orbfe_keypoint *)bufs_[0], (const uint8_t *)bufs_[1], nl,
                                      (const orbfe_keypoint *)bufs_[2], (const uint8_t *)bufs_[3], nr, scaleFactors.data(),
                                      (int)scaleFactors.size(), nRows, minD, maxD, (int32_t *)bufs_[4], (int32_t *)bufs_[5], 1);
        if (rc != ORBFE_OK) return rc;
        (void)hipMemcpy(bestDist.data(), bufs_[4], (size_t)nl * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(bestIdxR.data(), bufs_[5], (size_t)nl * 4, hipMemcpyDeviceToHost);
        return ORBFE_OK;
    }

private:
    orbfe_matcher *m_;
    void *bufs_[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t size_[6] = {0, 0, 0, 0, 0, 0};
};

// Frame::ComputeStereoMatches (src/Frame.cc:466-640) on the device.  `left` / `right` are the liborbfe handles of the
// two extractors that have just processed the stereo pair (ORBextractor::orbfeHandle()); keypoints and descriptors are
// the Frame members; outputs are Frame::mvuRight and Frame::mvDepth.
class StereoMatcher {
public:
    explicit StereoMatcher(int device = 0) : m_(nullptr) { orbfe_matcher_create(device, &m_); }
    ~StereoMatcher() {
        for (void *p : bufs_) if (p) (void)hipFree(p);
        orbfe_matcher_destroy(m_);
    }
    int ComputeStereoMatches(orbfe_extractor *left, orbfe_extractor *right, const std::vector<cv::KeyPoint> &kl, const cv::Mat &dl,
                             const std::vector<cv::KeyPoint> &kr, const cv::Mat &dr, float mbf, float mb,
                             std::vector<float> &mvuRight, std::vector<float> &mvDepth) {
        const int nl = (int)kl.size(), nr = (int)kr.size();
        mvuRight.assign(nl, -1.0f);   // :468-469
        mvDepth.assign(nl, -1.0f);
        if (nl == 0) return ORBFE_OK;
        const size_t need[6] = {(size_t)nl * 28, (size_t)nl * 32, (size_t)(nr ? nr : 1) * 28, (size_t)(nr ? nr : 1) * 32,
                                (size_t)nl * 4, (size_t)nl * 4};
        for (int i = 0; i < 6; i++)
            if (need[i] > size_[i]) {
                if (bufs_[i]) (void)hipFree(bufs_[i]);
                bufs_[i] = nullptr;
                size_[i] = 0;
                if (hipMalloc(&bufs_[i], need[i]) != hipSuccess) return ORBFE_ERR_DEVICE;
                size_[i] = need[i];
            }
        (void)hipMemcpy(bufs_[0], kl.data(), need[0], hipMemcpyHostToDevice);
        (void)hipMemcpy2D(bufs_[1], 32, dl.data, dl.step, 32, nl, hipMemcpyHostToDevice);
        if (nr) {
            (void)hipMemcpy(bufs_[2], kr.data(), (size_t)nr * 28, hipMemcpyHostToDevice);
            (void)hipMemcpy2D(bufs_[3], 32, dr.data, dr.step, 32, nr, hipMemcpyHostToDevice);
        }
        int rc = orbfe_stereo_match(m_, left, 0, right, 0, (const orbfe_keypoint *)bufs_[0], (const uint8_t *)bufs_[1], nl,
                                    (const orbfe_keypoint *)bufs_[2], (const uint8_t *)bufs_[3], nr, mbf, mb, (float *)bufs_[4],
                                    (float *)bufs_[5], 1);
        if (rc != ORBFE_OK) return rc;
        (void)hipMemcpy(mvuRight.data(), bufs_[4], (size_t)nl * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(mvDepth.data(), bufs_[5], (size_t)nl * 4, hipMemcpyDeviceToHost);
        return ORBFE_OK;
    }

private:
    orbfe_matcher *m_;
    void *bufs_[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t size_[6] = {0, 0, 0, 0, 0, 0};
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
    explicit ProjectionSearch(int device = 0) : dev_(device), m_(nullptr), g_(nullptr) { orbfe_matcher_create(device, &m_); }
    ~ProjectionSearch() {
        if (g_) orbfe_frame_grid_destroy(g_);
        orbfe_matcher_destroy(m_);
    }
    int build(const std::vector<cv::KeyPoint> &keysUn, const cv::Mat &desc, const std::vector<float> &uRight, float minX, float minY,
              float gridElementWidthInv, float gridElementHeightInv) {
        const int n = (int)keysUn.size();
        std::vector<float> kx(n), ky(n);
        std::vector<int32_t> oct(n);
        std::vector<unsigned char> d((size_t)n * 32);
        for (int i = 0; i < n; i++) {
            kx[i] = keysUn[i].pt.x;
            ky[i] = keysUn[i].pt.y;
            oct[i] = keysUn[i].octave;
            for (int b = 0; b < 32; b++) d[(size_t)i * 32 + b] = desc.ptr(i)[b];
        }
        if (g_) orbfe_frame_grid_destroy(g_);
        g_ = nullptr;
        n_ = n;
        return orbfe_frame_grid_create(dev_, kx.data(), ky.data(), oct.data(), d.data(), uRight.empty() ? nullptr : uRight.data(), n,
                                       minX, minY, gridElementWidthInv, gridElementHeightInv, &g_);
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
        int rc = orbfe_search_area(m_, g_, q.data(), qd.data(), nq, occupied.data(), res.data());
        if (rc != ORBFE_OK) return rc;
        std::vector<unsigned char> taken(n_, 0);                               // associations made by this call
        for (int i = 0; i < nq; i++) {
            orbfe_area_result r = res[i];
            if ((r.best_idx >= 0 && taken[r.best_idx]) || (r.best_idx2 >= 0 && taken[r.best_idx2])) {
                rc = orbfe_search_area(m_, g_, &q[i], &qd[(size_t)i * 32], 1, occupied.data(), &r);
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
    int searchLastFrame(const std::vector<TrackedPoint> &pts, const std::vector<float> &curAngles, bool checkOrientation,
                        std::vector<unsigned char> &occupied, std::vector<int> &assigned, int *nmatches) {
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
        int rc = orbfe_search_area(m_, g_, q.data(), qd.data(), nq, occupied.data(), res.data());
        if (rc != ORBFE_OK) return rc;
        const int HISTO_LENGTH = 30;
        std::vector<int> rotHist[HISTO_LENGTH];
        const float factor = 1.0f / HISTO_LENGTH;
        std::vector<unsigned char> taken(n_, 0);
        int n = 0;
        for (int i = 0; i < nq; i++) {
            orbfe_area_result r = res[i];
            if (r.best_idx >= 0 && taken[r.best_idx]) {                        // only the best match is used here
                rc = orbfe_search_area(m_, g_, &q[i], &qd[(size_t)i * 32], 1, occupied.data(), &r);
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
                    n--;
                }
            }
        }
        *nmatches = n;
        return ORBFE_OK;
    }
    long requeried() const { return requeried_; }

private:
    int dev_;
    orbfe_matcher *m_;
    orbfe_frame_grid *g_;
    int n_ = 0;
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
    explicit BowMatcher(int device = 0) : m_(nullptr) { orbfe_matcher_create(device, &m_); }
    ~BowMatcher() { orbfe_matcher_destroy(m_); }
    int SearchByBoW(const FeatureVector &fv1, const cv::Mat &desc1, const std::vector<cv::KeyPoint> &keys1,
                    const std::vector<unsigned char> &valid1, const FeatureVector &fv2, const cv::Mat &desc2,
                    const std::vector<cv::KeyPoint> &keys2, const std::vector<unsigned char> &blocked2, int maxDist, float nnRatio,
                    bool checkOrientation, std::vector<int> &match12, int *nmatches) {
        const int n1 = (int)keys1.size(), n2 = (int)keys2.size();
        match12.assign(n1, -1);
        *nmatches = 0;
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
        for (int i = 0; i < n1; i++) for (int b = 0; b < 32; b++) d1[(size_t)i * 32 + b] = desc1.ptr(i)[b];
        for (int i = 0; i < n2; i++) for (int b = 0; b < 32; b++) d2[(size_t)i * 32 + b] = desc2.ptr(i)[b];
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
};

}  // namespace orbfe
