// ORBmatcher_orbfe.cc -- member definitions of ORB_SLAM2::ORBmatcher (include/ORBmatcher.h of this repository) on top of
// the liborbfe C-ABI.  Compile this INSTEAD of the reference's src/ORBmatcher.cc and link liborbfe.so; Frame.h, KeyFrame.h
// and MapPoint.h stay the reference's.
//
// Every routine of the reference has the same three parts: (1) host geometry per MapPoint -- projection, depth / viewing
// angle / scale-range gates, PredictScale -- (2) a window search, GetFeaturesInArea + a Hamming loop over the candidates,
// or a brute-force search inside shared vocabulary nodes, and (3) bookkeeping on Frame / KeyFrame / MapPoint.  Here (1)
// runs first for ALL points of a call, with the reference's cv::Mat expressions; (2) is one launch over all of them in
// liborbfe (include/ORBmatcher_orbfe.h: orbfe::ProjectionSearch, BowMatcher, TriangulationMatcher); (3) is replayed on the
// host in the reference's order, and a point whose search result can depend on an association made earlier in the same
// call is searched again against the updated state -- so the outcome is the sequential loop's, association for
// association (tests/test_matcher_class.py drives all eleven methods against the oracle's sequential replays).
//
// The device copy of a frame's keypoint grid (Frame::mGrid / KeyFrame::mGrid, 64 x 48 cells) lives in a pool keyed by the
// frame id (Frame::mnId; KeyFrame::mnFrameId -- a KeyFrame holds the keypoints of the Frame it was made from), recycled
// least-recently-used, so neither class needs a new member.  Pool size: environment ORBFE_GRID_POOL (default 256 grids,
// about 120 KB of HBM each at 2000 keypoints).
//
// Errors: the reference has no error channel; as in adapters/ORBextractor_orbfe.cc a liborbfe failure prints the C-ABI
// message and aborts.
#include "ORBmatcher.h"

#include <climits>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <list>
#include <mutex>

#include "ORBmatcher_orbfe.h"

namespace ORB_SLAM2 {

const int ORBmatcher::TH_HIGH = 100;
const int ORBmatcher::TH_LOW = 50;
const int ORBmatcher::HISTO_LENGTH = 30;

namespace {

typedef orbfe::ProjectionSearch::WindowPoint WindowPoint;

void check(int rc, const char *what) {
    if (rc == ORBFE_OK) return;
    std::fprintf(stderr, "ORBmatcher(orbfe): %s: %s\n", what, orbfe_last_error());
    std::abort();
}

orbfe_matcher *context() {
    int rc = ORBFE_OK;
    orbfe_matcher *m = orbfe::thread_matcher(&rc);
    check(m ? ORBFE_OK : (rc ? rc : ORBFE_ERR_DEVICE), "matcher context");
    return m;
}

// ---- pool of device grids, keyed by frame id --------------------------------------------------------------------------
struct PooledGrid {
    orbfe_frame_grid *g = nullptr;
    unsigned long key = 0;
    unsigned long print = 0;   // fingerprint of the keypoints it was built from: frame ids restart after Tracking::Reset
    int n = 0;
    int users = 0;             // leases outstanding: a grid in use is never recycled
    bool building = false;     // reserved, its upload still running outside the pool's lock
};

// a few keypoints and descriptor rows, enough to tell two frames that were given the same id apart
unsigned long fingerprint(const std::vector<cv::KeyPoint> &keys, const cv::Mat &desc) {
    unsigned long h = 1469598103934665603ul;
    auto mix = [&h](const void *p, size_t bytes) {
        const unsigned char *b = (const unsigned char *)p;
        for (size_t i = 0; i < bytes; i++) h = (h ^ b[i]) * 1099511628211ul;
    };
    const size_t n = keys.size();
    mix(&n, sizeof(n));
    if (n == 0) return h;
    const size_t probe[4] = {0, n / 3, (2 * n) / 3, n - 1};
    for (size_t k = 0; k < 4; k++) {
        mix(&keys[probe[k]].pt, sizeof(cv::Point2f));
        mix(desc.ptr((int)probe[k]), 32);
    }
    return h;
}

class GridPool {
public:
    static GridPool &instance() {
        static GridPool *pool = new GridPool;   // never destroyed: worker threads may still search at process exit
        return *pool;
    }
    // the grid of frame `key`, built from the given members on first use (Frame::AssignFeaturesToGrid, src/Frame.cc:230-246).
    // The pool's mutex covers the bookkeeping only: the entry is reserved (users = 1, building) under the lock, the host
    // repack, the copy to the device and its synchronisation run outside it, and the entry is published through the
    // condition variable -- so Tracking, LocalMapping and LoopClosing do not queue behind one another's uploads; a thread
    // that wants a grid another thread is still building waits for that one entry.
    PooledGrid *acquire(orbfe_matcher *m, unsigned long key, const std::vector<cv::KeyPoint> &keysUn, const cv::Mat &desc,
                        const std::vector<float> &uRight, float minX, float minY, float wInv, float hInv) {
        const int n = (int)keysUn.size();
        const unsigned long print = fingerprint(keysUn, desc);
        PooledGrid *e = nullptr;
        {
            std::unique_lock<std::mutex> lock(mutex_);
            for (std::list<PooledGrid *>::iterator it = lru_.begin(); it != lru_.end(); ++it)
                if ((*it)->key == key && (*it)->n == n && (*it)->print == print) {
                    PooledGrid *hit = *it;
                    lru_.erase(it);
                    lru_.push_front(hit);
                    hit->users++;           // leased before waiting: a grid in use is never recycled
                    built_.wait(lock, [hit] { return !hit->building; });
                    return hit;
                }
            if (lru_.size() >= capacity_)
                for (std::list<PooledGrid *>::reverse_iterator it = lru_.rbegin(); it != lru_.rend(); ++it)
                    if ((*it)->users == 0) {
                        e = *it;
                        lru_.erase(std::next(it).base());
                        break;
                    }
            if (!e) e = new PooledGrid;   // room left, or every grid is in use right now (release() trims the excess)
            e->key = key;
            e->print = print;
            e->n = n;
            e->users = 1;
            e->building = true;
            lru_.push_front(e);
        }
        std::vector<float> kx(n), ky(n);
        std::vector<int32_t> oct(n);
        std::vector<unsigned char> d((size_t)n * 32 + 1);
        for (int i = 0; i < n; i++) {
            kx[i] = keysUn[i].pt.x;
            ky[i] = keysUn[i].pt.y;
            oct[i] = keysUn[i].octave;
            std::memcpy(&d[(size_t)i * 32], desc.ptr(i), 32);
        }
        const float *ur = (int)uRight.size() == n && n > 0 ? uRight.data() : nullptr;
        if (e->g)
            check(orbfe_frame_grid_update(m, e->g, kx.data(), ky.data(), oct.data(), d.data(), ur, n, minX, minY, wInv, hInv), "grid update");
        else
            check(orbfe_frame_grid_create(orbfe::default_device(), kx.data(), ky.data(), oct.data(), d.data(), ur, n, minX, minY, wInv,
                                          hInv, &e->g), "grid create");
        {
            std::lock_guard<std::mutex> lock(mutex_);
            e->building = false;
        }
        built_.notify_all();
        return e;
    }
    void release(PooledGrid *e) {
        PooledGrid *drop = nullptr;
        {
            std::lock_guard<std::mutex> lock(mutex_);
            e->users--;
            // the pool grows past its capacity only while every grid is leased; give the excess back as leases end
            if (e->users == 0 && lru_.size() > capacity_) {
                lru_.remove(e);
                drop = e;
            }
        }
        if (drop) {
            if (drop->g) (void)orbfe_frame_grid_destroy(drop->g);
            delete drop;
        }
    }
    size_t size() {
        std::lock_guard<std::mutex> lock(mutex_);
        return lru_.size();
    }

private:
    GridPool() {
        const char *c = std::getenv("ORBFE_GRID_POOL");
        capacity_ = c ? (size_t)std::max(2, std::atoi(c)) : 256;
    }
    std::mutex mutex_;
    std::condition_variable built_;
    std::list<PooledGrid *> lru_;
    size_t capacity_;
};

// a leased grid + the window searches on it for the calling thread
struct GridSearch {
    PooledGrid *lease;
    orbfe::ProjectionSearch ps;
    GridSearch(orbfe_matcher *m, PooledGrid *e, bool ownOrigin, float ox, float oy) : lease(e), ps(m, e->g, e->n, ownOrigin, ox, oy) {}
    ~GridSearch() { GridPool::instance().release(lease); }
};

// Frame: cells and windows both use the static image bounds (src/Frame.cc:332-346, :384-385)
PooledGrid *lease_of(orbfe_matcher *m, const Frame &F) {
    return GridPool::instance().acquire(m, F.mnId, F.mvKeysUn, F.mDescriptors, F.mvuRight, Frame::mnMinX, Frame::mnMinY,
                                        Frame::mfGridElementWidthInv, Frame::mfGridElementHeightInv);
}
// KeyFrame: its mGrid is the Frame's (src/KeyFrame.cc:31-57 copies it), so the cells are assigned with the Frame's float
// bounds; the windows subtract the KeyFrame's own int mnMinX / mnMinY (src/KeyFrame.cc:574-586)
PooledGrid *lease_of(orbfe_matcher *m, const KeyFrame *pKF) {
    return GridPool::instance().acquire(m, pKF->mnFrameId, pKF->mvKeysUn, pKF->mDescriptors, pKF->mvuRight, Frame::mnMinX, Frame::mnMinY,
                                        pKF->mfGridElementWidthInv, pKF->mfGridElementHeightInv);
}
#define ORBFE_FRAME_SEARCH(name, m, F) GridSearch name((m), lease_of((m), (F)), false, 0.0f, 0.0f)
#define ORBFE_KEYFRAME_SEARCH(name, m, pKF) GridSearch name((m), lease_of((m), (pKF)), true, (float)(pKF)->mnMinX, (float)(pKF)->mnMinY)

// ---- host geometry shared by the projection routines -------------------------------------------------------------------
struct Sim3Camera {   // [sR | st] split as at src/ORBmatcher.cc:299-303 and :986-990
    cv::Mat Rcw, tcw, Ow;
    explicit Sim3Camera(const cv::Mat &Scw) {
        cv::Mat sRcw = Scw.rowRange(0, 3).colRange(0, 3);
        const float scw = std::sqrt(sRcw.row(0).dot(sRcw.row(0)));
        Rcw = sRcw / scw;
        tcw = Scw.rowRange(0, 3).col(3) / scw;
        Ow = -(Rcw.t() * tcw);
    }
};

// pinhole projection the way the KeyFrame routines write it (:331-336, :859-864, :1019-1024, :1166-1171): x = X * invz first
inline void pinhole(const cv::Mat &p3Dc, float invz, float fx, float fy, float cx, float cy, float &u, float &v) {
    const float x = p3Dc.at<float>(0) * invz;
    const float y = p3Dc.at<float>(1) * invz;
    u = fx * x + cx;
    v = fy * y + cy;
}

// the distance / viewing-angle gates and the predicted level of :342-357, :872-887, :1030-1046.  PO = p3Dw - Ow.
template <class Host>
inline bool scale_gate(MapPoint *pMP, const cv::Mat &PO, bool checkNormal, Host *host, float &dist3D, int &level) {
    const float maxDistance = pMP->GetMaxDistanceInvariance();
    const float minDistance = pMP->GetMinDistanceInvariance();
    dist3D = cv::norm(PO);
    if (dist3D < minDistance || dist3D > maxDistance) return false;
    if (checkNormal) {
        cv::Mat Pn = pMP->GetNormal();
        if (PO.dot(Pn) < 0.5 * dist3D) return false;
    }
    level = pMP->PredictScale(dist3D, host);
    return true;
}

// One candidate of a batched window search: the MapPoint, where it sits in the caller's container, and its descriptor
// (kept alive here: GetDescriptor() returns a clone).
struct Candidate {
    MapPoint *mp;
    int slot;
    cv::Mat desc;
};

inline WindowPoint window(const Candidate &c, float u, float v, float ur, float radius, int level, float angle, int index) {
    WindowPoint w;
    w.u = u; w.v = v; w.ur = ur; w.radius = radius; w.level = level; w.angle = angle;
    w.desc = c.desc.ptr();
    w.index = index;
    return w;
}

inline std::vector<float> angles_of(const std::vector<cv::KeyPoint> &keys) {
    std::vector<float> a(keys.size());
    for (size_t i = 0; i < keys.size(); i++) a[i] = keys[i].angle;
    return a;
}

}  // namespace

ORBmatcher::ORBmatcher(float nnratio, bool checkOri) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

// src/ORBmatcher.cc:1647-1663
int ORBmatcher::DescriptorDistance(const cv::Mat &a, const cv::Mat &b) { return orbfe_hamming(a.ptr(), b.ptr()); }

float ORBmatcher::RadiusByViewingCos(const float &viewCos) { return viewCos > 0.998 ? 2.5f : 4.0f; }   // :131-137

void ORBmatcher::ComputeThreeMaxima(std::vector<int> *histo, const int L, int &ind1, int &ind2, int &ind3) {   // :1601-1642
    orbfe::three_maxima(histo, L, ind1, ind2, ind3);
}

// :140-157 (host form; the batched SearchForTriangulation evaluates the same expression on the device)
bool ORBmatcher::CheckDistEpipolarLine(const cv::KeyPoint &kp1, const cv::KeyPoint &kp2, const cv::Mat &F12, const KeyFrame *pKF2) {
    const float a = kp1.pt.x * F12.at<float>(0, 0) + kp1.pt.y * F12.at<float>(1, 0) + F12.at<float>(2, 0);
    const float b = kp1.pt.x * F12.at<float>(0, 1) + kp1.pt.y * F12.at<float>(1, 1) + F12.at<float>(2, 1);
    const float c = kp1.pt.x * F12.at<float>(0, 2) + kp1.pt.y * F12.at<float>(1, 2) + F12.at<float>(2, 2);
    const float num = a * kp2.pt.x + b * kp2.pt.y + c;
    const float den = a * a + b * b;
    if (den == 0) return false;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * pKF2->mvLevelSigma2[kp2.octave];
}

// ---- :45-129, Tracking::SearchLocalPoints ------------------------------------------------------------------------------
int ORBmatcher::SearchByProjection(Frame &F, const std::vector<MapPoint *> &vpMapPoints, const float th) {
    std::vector<Candidate> cand;
    std::vector<orbfe::ProjectedPoint> pts;
    for (size_t iMP = 0; iMP < vpMapPoints.size(); iMP++) {
        MapPoint *pMP = vpMapPoints[iMP];
        if (!pMP->mbTrackInView || pMP->isBad()) continue;                     // :54-58
        Candidate c = {pMP, (int)iMP, pMP->GetDescriptor()};
        cand.push_back(c);
    }
    if (cand.empty() || F.N == 0) return 0;
    pts.resize(cand.size());
    for (size_t i = 0; i < cand.size(); i++) {
        MapPoint *pMP = cand[i].mp;
        pts[i].projX = pMP->mTrackProjX;
        pts[i].projY = pMP->mTrackProjY;
        pts[i].projXR = pMP->mTrackProjXR;
        pts[i].level = pMP->mnTrackScaleLevel;
        pts[i].viewCos = pMP->mTrackViewCos;
        pts[i].desc = cand[i].desc.ptr();
        pts[i].observed = pMP->Observations() > 0;
    }
    std::vector<unsigned char> occupied(F.N, 0);                               // :87-89
    for (int i = 0; i < F.N; i++) occupied[i] = F.mvpMapPoints[i] && F.mvpMapPoints[i]->Observations() > 0;
    orbfe_matcher *m = context();
    ORBFE_FRAME_SEARCH(gs, m, F);
    std::vector<int> matchIdx;
    int nmatches = 0;
    check(gs.ps.search(pts, F.mvScaleFactors, th, mfNNratio, occupied, matchIdx, &nmatches), "SearchByProjection(Frame, MapPoints)");
    for (size_t i = 0; i < cand.size(); i++)
        if (matchIdx[i] >= 0) F.mvpMapPoints[matchIdx[i]] = cand[i].mp;      // :123, in the loop's order
    return nmatches;
}

// ---- :159-288, relocalisation and reference-keyframe tracking ------------------------------------------------------------
int ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint *> &vpMapPointMatches) {
    const std::vector<MapPoint *> vpMapPointsKF = pKF->GetMapPointMatches();
    vpMapPointMatches = std::vector<MapPoint *>(F.N, static_cast<MapPoint *>(NULL));
    std::vector<unsigned char> valid1(vpMapPointsKF.size()), none;
    for (size_t i = 0; i < valid1.size(); i++) valid1[i] = vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad();   // :191-197
    orbfe::BowMatcher bow(context());
    std::vector<int> match12;
    int nmatches = 0;
    check(bow.SearchByBoW(pKF->mFeatVec, pKF->mDescriptors, pKF->mvKeysUn, valid1, F.mFeatVec, F.mDescriptors, F.mvKeys, none, TH_LOW,
                          mfNNratio, mbCheckOrientation, match12, &nmatches), "SearchByBoW(KeyFrame, Frame)");
    for (size_t i = 0; i < match12.size(); i++)
        if (match12[i] >= 0) vpMapPointMatches[match12[i]] = vpMapPointsKF[i];
    return nmatches;
}

// ---- :290-403, LoopClosing::ComputeSim3 -----------------------------------------------------------------------------------
int ORBmatcher::SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, std::vector<MapPoint *> &vpMatched,
                                   int th) {
    const float &fx = pKF->fx, &fy = pKF->fy, &cx = pKF->cx, &cy = pKF->cy;
    const Sim3Camera cam(Scw);
    std::set<MapPoint *> spAlreadyFound(vpMatched.begin(), vpMatched.end());
    spAlreadyFound.erase(static_cast<MapPoint *>(NULL));

    std::vector<Candidate> cand;
    std::vector<WindowPoint> pts;
    for (int iMP = 0, iendMP = (int)vpPoints.size(); iMP < iendMP; iMP++) {
        MapPoint *pMP = vpPoints[iMP];
        if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
        cv::Mat p3Dw = pMP->GetWorldPos();
        cv::Mat p3Dc = cam.Rcw * p3Dw + cam.tcw;
        if (p3Dc.at<float>(2) < 0.0) continue;
        const float invz = 1 / p3Dc.at<float>(2);
        float u, v;
        pinhole(p3Dc, invz, fx, fy, cx, cy, u, v);
        if (!pKF->IsInImage(u, v)) continue;
        cv::Mat PO = p3Dw - cam.Ow;
        float dist;
        int nPredictedLevel;
        if (!scale_gate(pMP, PO, true, pKF, dist, nPredictedLevel)) continue;
        const float radius = th * pKF->mvScaleFactors[nPredictedLevel];
        Candidate c = {pMP, iMP, pMP->GetDescriptor()};
        cand.push_back(c);
        pts.push_back(window(cand.back(), u, v, 0.0f, radius, nPredictedLevel, 0.0f, iMP));
    }
    if (cand.empty() || pKF->N == 0) return 0;
    for (size_t i = 0; i < cand.size(); i++) pts[i].desc = cand[i].desc.ptr();   // cand may have been reallocated
    std::vector<unsigned char> matched(vpMatched.size());
    for (size_t i = 0; i < matched.size(); i++) matched[i] = vpMatched[i] != NULL;   // :375
    orbfe_matcher *m = context();
    ORBFE_KEYFRAME_SEARCH(gs, m, pKF);
    std::vector<int> matchIdx;
    int nmatches = 0;
    check(gs.ps.searchSim3Projection(pts, matched, matchIdx, &nmatches), "SearchByProjection(KeyFrame, Scw)");
    for (size_t i = 0; i < cand.size(); i++)
        if (matchIdx[i] >= 0) vpMatched[matchIdx[i]] = cand[i].mp;            // :396
    return nmatches;
}

// ---- :405-520, monocular initialisation ------------------------------------------------------------------------------------
int ORBmatcher::SearchForInitialization(Frame &F1, Frame &F2, std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12,
                                        int windowSize) {
    const int n1 = (int)F1.mvKeysUn.size();
    vnMatches12 = std::vector<int>(n1, -1);
    std::vector<WindowPoint> pts;
    for (int i1 = 0; i1 < n1; i1++) {
        const int level1 = F1.mvKeysUn[i1].octave;
        if (level1 > 0) continue;                                              // :421-423
        WindowPoint w;
        w.u = vbPrevMatched[i1].x; w.v = vbPrevMatched[i1].y; w.ur = 0.0f;
        w.radius = (float)windowSize;
        w.level = level1;
        w.angle = F1.mvKeysUn[i1].angle;
        w.desc = F1.mDescriptors.ptr(i1);
        w.index = i1;
        pts.push_back(w);
    }
    int nmatches = 0;
    if (!pts.empty() && !F2.mvKeysUn.empty()) {
        orbfe_matcher *m = context();
        ORBFE_FRAME_SEARCH(gs, m, F2);
        check(gs.ps.searchForInitialization(pts, n1, angles_of(F2.mvKeysUn), mfNNratio, mbCheckOrientation, vnMatches12, &nmatches),
              "SearchForInitialization");
    }
    for (size_t i1 = 0, iend1 = vnMatches12.size(); i1 < iend1; i1++)          // :514-517
        if (vnMatches12[i1] >= 0) vbPrevMatched[i1] = F2.mvKeysUn[vnMatches12[i1]].pt;
    return nmatches;
}

// ---- :522-655, LoopClosing::ComputeSim3 -------------------------------------------------------------------------------------
int ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12) {
    const std::vector<MapPoint *> vpMapPoints1 = pKF1->GetMapPointMatches();
    const std::vector<MapPoint *> vpMapPoints2 = pKF2->GetMapPointMatches();
    vpMatches12 = std::vector<MapPoint *>(vpMapPoints1.size(), static_cast<MapPoint *>(NULL));
    std::vector<unsigned char> valid1(vpMapPoints1.size()), blocked2(vpMapPoints2.size());
    for (size_t i = 0; i < valid1.size(); i++) valid1[i] = vpMapPoints1[i] && !vpMapPoints1[i]->isBad();        // :558-562
    for (size_t i = 0; i < blocked2.size(); i++) blocked2[i] = !vpMapPoints2[i] || vpMapPoints2[i]->isBad();    // :574-580
    orbfe::BowMatcher bow(context());
    std::vector<int> match12;
    int nmatches = 0;
    check(bow.SearchByBoW(pKF1->mFeatVec, pKF1->mDescriptors, pKF1->mvKeysUn, valid1, pKF2->mFeatVec, pKF2->mDescriptors, pKF2->mvKeysUn,
                          blocked2, TH_LOW - 1 /* strict '<' at :598 */, mfNNratio, mbCheckOrientation, match12, &nmatches),
          "SearchByBoW(KeyFrame, KeyFrame)");
    for (size_t i = 0; i < match12.size() && i < vpMatches12.size(); i++)
        if (match12[i] >= 0) vpMatches12[i] = vpMapPoints2[match12[i]];
    return nmatches;
}

// ---- :657-823, LocalMapping::CreateNewMapPoints ------------------------------------------------------------------------------
int ORBmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t> > &vMatchedPairs,
                                       const bool bOnlyStereo) {
    // epipole in the second image (:664-670)
    cv::Mat Cw = pKF1->GetCameraCenter();
    cv::Mat R2w = pKF2->GetRotation();
    cv::Mat t2w = pKF2->GetTranslation();
    cv::Mat C2 = R2w * Cw + t2w;
    const float invz = 1.0f / C2.at<float>(2);
    const float ex = pKF2->fx * C2.at<float>(0) * invz + pKF2->cx;
    const float ey = pKF2->fy * C2.at<float>(1) * invz + pKF2->cy;

    std::vector<unsigned char> hasMp1(pKF1->N), hasMp2(pKF2->N);
    for (int i = 0; i < pKF1->N; i++) hasMp1[i] = pKF1->GetMapPoint(i) != NULL;   // :699-703
    for (int i = 0; i < pKF2->N; i++) hasMp2[i] = pKF2->GetMapPoint(i) != NULL;   // :722-726
    float f12[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) f12[3 * r + c] = F12.at<float>(r, c);
    orbfe::TriangulationMatcher tri(context());
    int nmatches = 0;
    check(tri.SearchForTriangulation(pKF1->mFeatVec, pKF1->mDescriptors, pKF1->mvKeysUn, hasMp1, pKF1->mvuRight, pKF2->mFeatVec,
                                     pKF2->mDescriptors, pKF2->mvKeysUn, hasMp2, pKF2->mvuRight, f12, ex, ey, pKF2->mvScaleFactors,
                                     pKF2->mvLevelSigma2, bOnlyStereo, mbCheckOrientation, vMatchedPairs, &nmatches),
          "SearchForTriangulation");
    return nmatches;
}

// ---- :825-975, LocalMapping::SearchInNeighbors -----------------------------------------------------------------------------
int ORBmatcher::Fuse(KeyFrame *pKF, const std::vector<MapPoint *> &vpMapPoints, const float th) {
    cv::Mat Rcw = pKF->GetRotation();
    cv::Mat tcw = pKF->GetTranslation();
    const float &fx = pKF->fx, &fy = pKF->fy, &cx = pKF->cx, &cy = pKF->cy, &bf = pKF->mbf;
    cv::Mat Ow = pKF->GetCameraCenter();

    // geometry for every point that passes the entry gates now; the gates that the loop's own fusions can close
    // (isBad, IsInKeyFrame -- both only ever turn true inside the loop) are tested again when the loop is replayed
    std::vector<Candidate> cand;
    std::vector<WindowPoint> pts;
    const int nMPs = (int)vpMapPoints.size();
    for (int i = 0; i < nMPs; i++) {
        MapPoint *pMP = vpMapPoints[i];
        if (!pMP) continue;
        if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
        cv::Mat p3Dw = pMP->GetWorldPos();
        cv::Mat p3Dc = Rcw * p3Dw + tcw;
        if (p3Dc.at<float>(2) < 0.0f) continue;
        const float invz = 1 / p3Dc.at<float>(2);
        float u, v;
        pinhole(p3Dc, invz, fx, fy, cx, cy, u, v);
        if (!pKF->IsInImage(u, v)) continue;
        const float ur = u - bf * invz;
        cv::Mat PO = p3Dw - Ow;
        float dist3D;
        int nPredictedLevel;
        if (!scale_gate(pMP, PO, true, pKF, dist3D, nPredictedLevel)) continue;
        const float radius = th * pKF->mvScaleFactors[nPredictedLevel];
        Candidate c = {pMP, i, pMP->GetDescriptor()};
        cand.push_back(c);
        pts.push_back(window(cand.back(), u, v, ur, radius, nPredictedLevel, 0.0f, i));
    }
    if (cand.empty() || pKF->N == 0) return 0;
    for (size_t i = 0; i < cand.size(); i++) pts[i].desc = cand[i].desc.ptr();
    orbfe_matcher *m = context();
    std::vector<int> bestIdx;
    int selected = 0;
    {
        ORBFE_KEYFRAME_SEARCH(gs, m, pKF);
        check(gs.ps.fuseSelect(pts, pKF->mvInvLevelSigma2, bestIdx, &selected), "Fuse(KeyFrame, MapPoints)");
    }
    int nFused = 0;
    for (size_t i = 0; i < cand.size(); i++) {                                 // :951-971 in the loop's order
        MapPoint *pMP = cand[i].mp;
        if (bestIdx[i] < 0) continue;
        if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;                  // closed by an earlier fusion of this call
        MapPoint *pMPinKF = pKF->GetMapPoint(bestIdx[i]);
        if (pMPinKF) {
            if (!pMPinKF->isBad()) {
                if (pMPinKF->Observations() > pMP->Observations())
                    pMP->Replace(pMPinKF);
                else
                    pMPinKF->Replace(pMP);
            }
        } else {
            pMP->AddObservation(pKF, bestIdx[i]);
            pKF->AddMapPoint(pMP, bestIdx[i]);
        }
        nFused++;
    }
    return nFused;
}

// ---- :977-1100, LoopClosing::SearchAndFuse ------------------------------------------------------------------------------------
int ORBmatcher::Fuse(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, float th, std::vector<MapPoint *> &vpReplacePoint) {
    const float &fx = pKF->fx, &fy = pKF->fy, &cx = pKF->cx, &cy = pKF->cy;
    const Sim3Camera cam(Scw);
    const std::set<MapPoint *> spAlreadyFound = pKF->GetMapPoints();

    std::vector<Candidate> cand;
    std::vector<WindowPoint> pts;
    const int nPoints = (int)vpPoints.size();
    for (int iMP = 0; iMP < nPoints; iMP++) {
        MapPoint *pMP = vpPoints[iMP];
        if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
        cv::Mat p3Dw = pMP->GetWorldPos();
        cv::Mat p3Dc = cam.Rcw * p3Dw + cam.tcw;
        if (p3Dc.at<float>(2) < 0.0f) continue;
        const float invz = 1.0 / p3Dc.at<float>(2);
        float u, v;
        pinhole(p3Dc, invz, fx, fy, cx, cy, u, v);
        if (!pKF->IsInImage(u, v)) continue;
        cv::Mat PO = p3Dw - cam.Ow;
        float dist3D;
        int nPredictedLevel;
        if (!scale_gate(pMP, PO, true, pKF, dist3D, nPredictedLevel)) continue;
        const float radius = th * pKF->mvScaleFactors[nPredictedLevel];
        Candidate c = {pMP, iMP, pMP->GetDescriptor()};
        cand.push_back(c);
        pts.push_back(window(cand.back(), u, v, 0.0f, radius, nPredictedLevel, 0.0f, iMP));
    }
    if (cand.empty() || pKF->N == 0) return 0;
    for (size_t i = 0; i < cand.size(); i++) pts[i].desc = cand[i].desc.ptr();
    orbfe_matcher *m = context();
    std::vector<int> bestIdx;
    int selected = 0;
    {
        ORBFE_KEYFRAME_SEARCH(gs, m, pKF);
        check(gs.ps.fuseSelect(pts, std::vector<float>(), bestIdx, &selected), "Fuse(KeyFrame, Scw)");
    }
    int nFused = 0;
    for (size_t i = 0; i < cand.size(); i++) {                                 // :1081-1096
        if (bestIdx[i] < 0) continue;
        MapPoint *pMP = cand[i].mp;
        MapPoint *pMPinKF = pKF->GetMapPoint(bestIdx[i]);
        if (pMPinKF) {
            if (!pMPinKF->isBad()) vpReplacePoint[cand[i].slot] = pMPinKF;
        } else {
            pMP->AddObservation(pKF, bestIdx[i]);
            pKF->AddMapPoint(pMP, bestIdx[i]);
        }
        nFused++;
    }
    return nFused;
}

// ---- :1102-1326, LoopClosing::ComputeSim3 ----------------------------------------------------------------------------------------
int ORBmatcher::SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12, const float &s12, const cv::Mat &R12,
                             const cv::Mat &t12, const float th) {
    const float &fx = pKF1->fx, &fy = pKF1->fy, &cx = pKF1->cx, &cy = pKF1->cy;
    cv::Mat R1w = pKF1->GetRotation(), t1w = pKF1->GetTranslation();
    cv::Mat R2w = pKF2->GetRotation(), t2w = pKF2->GetTranslation();
    cv::Mat sR12 = s12 * R12;
    cv::Mat sR21 = (1.0 / s12) * R12.t();
    cv::Mat t21 = -(sR21 * t12);

    const std::vector<MapPoint *> vpMapPoints1 = pKF1->GetMapPointMatches();
    const int N1 = (int)vpMapPoints1.size();
    const std::vector<MapPoint *> vpMapPoints2 = pKF2->GetMapPointMatches();
    const int N2 = (int)vpMapPoints2.size();
    std::vector<bool> vbAlreadyMatched1(N1, false), vbAlreadyMatched2(N2, false);
    for (int i = 0; i < N1; i++) {                                             // :1132-1142
        MapPoint *pMP = vpMatches12[i];
        if (pMP) {
            vbAlreadyMatched1[i] = true;
            int idx2 = pMP->GetIndexInKeyFrame(pKF2);
            if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true;
        }
    }
    // one direction: the MapPoints of `from` (in its camera frame through Rw, tw) moved by [sR | t] and projected into `into`
    struct Direction {
        std::vector<Candidate> cand;
        std::vector<WindowPoint> pts;
    } d12, d21;
    auto collect = [&](Direction &d, const std::vector<MapPoint *> &vpFrom, const std::vector<bool> &done, const cv::Mat &Rw,
                       const cv::Mat &tw, const cv::Mat &sR, const cv::Mat &t, KeyFrame *into) {
        for (int i = 0; i < (int)vpFrom.size(); i++) {
            MapPoint *pMP = vpFrom[i];
            if (!pMP || done[i]) continue;
            if (pMP->isBad()) continue;
            cv::Mat p3Dw = pMP->GetWorldPos();
            cv::Mat p3Dfrom = Rw * p3Dw + tw;
            cv::Mat p3Dinto = sR * p3Dfrom + t;
            if (p3Dinto.at<float>(2) < 0.0) continue;
            const float invz = 1.0 / p3Dinto.at<float>(2);
            float u, v;
            pinhole(p3Dinto, invz, fx, fy, cx, cy, u, v);
            if (!into->IsInImage(u, v)) continue;
            float dist3D;
            int nPredictedLevel;
            if (!scale_gate(pMP, p3Dinto, false, into, dist3D, nPredictedLevel)) continue;   // norm of the camera-frame point (:1179)
            const float radius = th * into->mvScaleFactors[nPredictedLevel];
            Candidate c = {pMP, i, pMP->GetDescriptor()};
            d.cand.push_back(c);
            d.pts.push_back(window(d.cand.back(), u, v, 0.0f, radius, nPredictedLevel, 0.0f, i));
        }
        for (size_t i = 0; i < d.cand.size(); i++) d.pts[i].desc = d.cand[i].desc.ptr();
    };
    collect(d12, vpMapPoints1, vbAlreadyMatched1, R1w, t1w, sR21, t21, pKF2);   // :1148-1225
    collect(d21, vpMapPoints2, vbAlreadyMatched2, R2w, t2w, sR12, t12, pKF1);   // :1228-1305
    if (pKF1->N == 0 || pKF2->N == 0) return 0;

    orbfe_matcher *m = context();
    ORBFE_KEYFRAME_SEARCH(g1, m, pKF1);
    ORBFE_KEYFRAME_SEARCH(g2, m, pKF2);
    std::vector<int> match12;
    int nFound = 0;
    check(orbfe::ProjectionSearch::SearchBySim3(g1.ps, g2.ps, d12.pts, d21.pts, match12, &nFound), "SearchBySim3");
    for (int i1 = 0; i1 < N1 && i1 < (int)match12.size(); i1++)               // :1310-1323
        if (match12[i1] >= 0) vpMatches12[i1] = vpMapPoints2[match12[i1]];
    return nFound;
}

// ---- :1328-1470, Tracking::TrackWithMotionModel ----------------------------------------------------------------------------------
int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono) {
    const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
    const cv::Mat twc = -(Rcw.t() * tcw);
    const cv::Mat Rlw = LastFrame.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tlw = LastFrame.mTcw.rowRange(0, 3).col(3);
    const cv::Mat tlc = Rlw * twc + tlw;
    const bool bForward = tlc.at<float>(2) > CurrentFrame.mb && !bMono;
    const bool bBackward = -tlc.at<float>(2) > CurrentFrame.mb && !bMono;

    std::vector<Candidate> cand;
    std::vector<orbfe::TrackedPoint> pts;
    for (int i = 0; i < LastFrame.N; i++) {
        MapPoint *pMP = LastFrame.mvpMapPoints[i];
        if (!pMP || LastFrame.mvbOutlier[i]) continue;
        cv::Mat x3Dw = pMP->GetWorldPos();
        cv::Mat x3Dc = Rcw * x3Dw + tcw;
        const float xc = x3Dc.at<float>(0);
        const float yc = x3Dc.at<float>(1);
        const float invzc = 1.0 / x3Dc.at<float>(2);
        if (invzc < 0) continue;
        float u = CurrentFrame.fx * xc * invzc + CurrentFrame.cx;
        float v = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
        if (u < CurrentFrame.mnMinX || u > CurrentFrame.mnMaxX) continue;
        if (v < CurrentFrame.mnMinY || v > CurrentFrame.mnMaxY) continue;
        const int nLastOctave = LastFrame.mvKeys[i].octave;
        orbfe::TrackedPoint p;
        p.u = u; p.v = v;
        p.ur = u - CurrentFrame.mbf * invzc;                                   // :1409
        p.radius = th * CurrentFrame.mvScaleFactors[nLastOctave];              // :1381
        if (bForward) { p.minLevel = nLastOctave; p.maxLevel = -1; }           // :1385-1390
        else if (bBackward) { p.minLevel = 0; p.maxLevel = nLastOctave; }
        else { p.minLevel = nLastOctave - 1; p.maxLevel = nLastOctave + 1; }
        p.angle = LastFrame.mvKeysUn[i].angle;
        p.desc = NULL;
        p.observed = pMP->Observations() > 0;
        Candidate c = {pMP, i, pMP->GetDescriptor()};
        cand.push_back(c);
        pts.push_back(p);
    }
    if (cand.empty() || CurrentFrame.N == 0) return 0;
    for (size_t i = 0; i < cand.size(); i++) pts[i].desc = cand[i].desc.ptr();
    std::vector<unsigned char> occupied(CurrentFrame.N, 0);                    // :1403-1405
    for (int i = 0; i < CurrentFrame.N; i++)
        occupied[i] = CurrentFrame.mvpMapPoints[i] && CurrentFrame.mvpMapPoints[i]->Observations() > 0;
    orbfe_matcher *m = context();
    ORBFE_FRAME_SEARCH(gs, m, CurrentFrame);
    std::vector<int> assigned, removed;
    int nmatches = 0;
    check(gs.ps.searchLastFrame(pts, angles_of(CurrentFrame.mvKeysUn), mbCheckOrientation, occupied, assigned, &nmatches, &removed),
          "SearchByProjection(Frame, Frame)");
    for (int i2 = 0; i2 < CurrentFrame.N; i2++)
        if (assigned[i2] >= 0) CurrentFrame.mvpMapPoints[i2] = cand[assigned[i2]].mp;                  // :1428
    for (size_t k = 0; k < removed.size(); k++) CurrentFrame.mvpMapPoints[removed[k]] = static_cast<MapPoint *>(NULL);   // :1462
    return nmatches;
}

// ---- :1472-1599, Tracking::Relocalization ------------------------------------------------------------------------------------------
int ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const std::set<MapPoint *> &sAlreadyFound, const float th,
                                   const int ORBdist) {
    const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
    const cv::Mat Ow = -(Rcw.t() * tcw);
    const std::vector<MapPoint *> vpMPs = pKF->GetMapPointMatches();

    std::vector<Candidate> cand;
    std::vector<WindowPoint> pts;
    for (size_t i = 0, iend = vpMPs.size(); i < iend; i++) {
        MapPoint *pMP = vpMPs[i];
        if (!pMP) continue;
        if (pMP->isBad() || sAlreadyFound.count(pMP)) continue;
        cv::Mat x3Dw = pMP->GetWorldPos();
        cv::Mat x3Dc = Rcw * x3Dw + tcw;
        const float xc = x3Dc.at<float>(0);
        const float yc = x3Dc.at<float>(1);
        const float invzc = 1.0 / x3Dc.at<float>(2);
        const float u = CurrentFrame.fx * xc * invzc + CurrentFrame.cx;
        const float v = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
        if (u < CurrentFrame.mnMinX || u > CurrentFrame.mnMaxX) continue;
        if (v < CurrentFrame.mnMinY || v > CurrentFrame.mnMaxY) continue;
        cv::Mat PO = x3Dw - Ow;
        float dist3D;
        int nPredictedLevel;
        if (!scale_gate(pMP, PO, false, &CurrentFrame, dist3D, nPredictedLevel)) continue;
        const float radius = th * CurrentFrame.mvScaleFactors[nPredictedLevel];
        Candidate c = {pMP, (int)i, pMP->GetDescriptor()};
        cand.push_back(c);
        pts.push_back(window(cand.back(), u, v, 0.0f, radius, nPredictedLevel, pKF->mvKeysUn[i].angle, (int)i));
    }
    if (cand.empty() || CurrentFrame.N == 0) return 0;
    for (size_t i = 0; i < cand.size(); i++) pts[i].desc = cand[i].desc.ptr();
    std::vector<unsigned char> occupied(CurrentFrame.N, 0);                    // :1541
    for (int i = 0; i < CurrentFrame.N; i++) occupied[i] = CurrentFrame.mvpMapPoints[i] != NULL;
    orbfe_matcher *m = context();
    ORBFE_FRAME_SEARCH(gs, m, CurrentFrame);
    std::vector<int> assigned, removed;
    int nmatches = 0;
    check(gs.ps.searchRelocalization(pts, angles_of(CurrentFrame.mvKeysUn), ORBdist, mbCheckOrientation, occupied, assigned, &nmatches,
                                     &removed),
          "SearchByProjection(Frame, KeyFrame)");
    for (int i2 = 0; i2 < CurrentFrame.N; i2++)
        if (assigned[i2] >= 0) CurrentFrame.mvpMapPoints[i2] = cand[assigned[i2]].mp;                  // :1557
    for (size_t k = 0; k < removed.size(); k++) CurrentFrame.mvpMapPoints[removed[k]] = NULL;        // :1591
    return nmatches;
}

}  // namespace ORB_SLAM2
