// matcher_main.cc -- drives ORB_SLAM2::ORBmatcher (include/ORBmatcher.h + adapters/ORBmatcher_orbfe.cc) through its eleven
// Search* / Fuse methods the way Tracking, LocalMapping and LoopClosing call them (reference src/Tracking.cc:765-768,
// :886-892, :1192, :1378, :1453; src/LocalMapping.cc:268, :489-514; src/LoopClosing.cc:265, :323, :375, :600), on a small
// world (views, map points, observations) read from a file written by tests/test_matcher_class.py, and dumps what each
// method leaves behind so that the pytest driver can compare it with the oracle's sequential replay.
// Frame / KeyFrame / MapPoint are the stand-ins of tests/cpp/stubs (the members src/ORBmatcher.cc touches).
//   usage: matcher_main world.bin out.bin
//          matcher_main --concurrent REPS world1.bin out1.bin world2.bin out2.bin ...
// The second form runs every case on a host thread of its own, REPS times each, all threads released together -- the way
// Tracking, LocalMapping and LoopClosing use ORBmatcher temporaries at once (reference src/System.cc:92-102): one liborbfe
// matcher context per thread (include/orbfe_context.h) and ONE shared pool of device grids (adapters/ORBmatcher_orbfe.cc,
// ORBFE_GRID_POOL sets its size).  The worlds must share the camera (the Frame statics are process-wide, as in the reference).
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <thread>
#include <vector>

#include "ORBmatcher.h"

using namespace ORB_SLAM2;

namespace {

struct Reader {
    FILE *f;
    bool ok = true;
    void raw(void *p, size_t bytes) {
        if (bytes && std::fread(p, 1, bytes, f) != bytes) ok = false;
    }
    int i32() { int v = 0; raw(&v, 4); return v; }
    float f32() { float v = 0; raw(&v, 4); return v; }
    template <class T> std::vector<T> vec(size_t n) {
        std::vector<T> v(n);
        raw(v.data(), n * sizeof(T));
        return v;
    }
};

struct Writer {
    FILE *f;
    void i32(int v) { std::fwrite(&v, 4, 1, f); }
    template <class T> void vec(const std::vector<T> &v) { if (!v.empty()) std::fwrite(v.data(), sizeof(T), v.size(), f); }
};

cv::Mat mat_from(const float *p, int rows, int cols) {
    cv::Mat m(rows, cols, CV_32F);
    for (int i = 0; i < rows; i++)
        for (int j = 0; j < cols; j++) m.at<float>(i, j) = p[i * cols + j];
    return m;
}

struct Scales {
    int nlevels;
    float scaleFactor, logScaleFactor;
    std::vector<float> sf, inv, sigma2, invSigma2;
};

struct View {   // one image: usable as a Frame and as the KeyFrame made from it
    Frame F;
    KeyFrame KF;
    std::vector<int> mp;   // map point index per keypoint, -1 = none
};

struct World {
    std::vector<View *> views;
    std::vector<MapPoint *> points;
    int index_of(MapPoint *p) const {
        if (!p) return -1;
        for (size_t i = 0; i < points.size(); i++)
            if (points[i] == p) return (int)i;
        return -2;
    }
    std::vector<int> indices(const std::vector<MapPoint *> &v) const {
        std::vector<int> out(v.size());
        for (size_t i = 0; i < v.size(); i++) out[i] = index_of(v[i]);
        return out;
    }
    std::vector<MapPoint *> pointers(const std::vector<int> &v) const {
        std::vector<MapPoint *> out(v.size());
        for (size_t i = 0; i < v.size(); i++) out[i] = v[i] >= 0 ? points[v[i]] : static_cast<MapPoint *>(NULL);
        return out;
    }
};

}  // namespace

namespace {

void set_camera_statics(const float *cam) {
    Frame::fx = cam[0]; Frame::fy = cam[1]; Frame::cx = cam[2]; Frame::cy = cam[3];
    Frame::mnMinX = cam[6]; Frame::mnMaxX = cam[7]; Frame::mnMinY = cam[8]; Frame::mnMaxY = cam[9];
    Frame::mfGridElementWidthInv = cam[10]; Frame::mfGridElementHeightInv = cam[11];
}

bool camera_statics_are(const float *cam) {
    return Frame::fx == cam[0] && Frame::fy == cam[1] && Frame::cx == cam[2] && Frame::cy == cam[3] && Frame::mnMinX == cam[6] &&
           Frame::mnMaxX == cam[7] && Frame::mnMinY == cam[8] && Frame::mnMaxY == cam[9] &&
           Frame::mfGridElementWidthInv == cam[10] && Frame::mfGridElementHeightInv == cam[11];
}

// one case: load the world, run its operation through the class, dump what it left behind.  setStatics: single-case mode
// writes the camera into the Frame statics; the concurrent mode set them before the threads started and only checks.
int run_case(const char *inPath, const char *outPath, bool setStatics, bool quiet) {
    Reader r{std::fopen(inPath, "rb")};
    if (!r.f) return 3;
    const int op = r.i32(), nViews = r.i32(), nPoints = r.i32();
    float cam[12];
    r.raw(cam, sizeof(cam));   // fx fy cx cy bf b minX maxX minY maxY wInv hInv
    if (setStatics) set_camera_statics(cam);
    else if (!camera_statics_are(cam)) return 5;
    Scales S;
    S.nlevels = r.i32();
    S.scaleFactor = r.f32();
    S.logScaleFactor = r.f32();
    S.sf = r.vec<float>(S.nlevels);
    S.sigma2 = r.vec<float>(S.nlevels);
    S.invSigma2 = r.vec<float>(S.nlevels);
    S.inv.resize(S.nlevels);
    for (int i = 0; i < S.nlevels; i++) S.inv[i] = 1.0f / S.sf[i];
    if (!r.ok || nViews < 1 || nViews > 8 || nPoints < 0) return 3;

    World W;
    for (int k = 0; k < nPoints; k++) {
        MapPoint *p = new MapPoint;
        float g[8];
        r.raw(g, sizeof(g));   // pos[3] normal[3] minDist maxDist
        p->mWorldPos = mat_from(g, 3, 1);
        p->mNormalVector = mat_from(g + 3, 3, 1);
        p->mfMinDistance = g[6];
        p->mfMaxDistance = g[7];
        p->mDescriptor.create(1, 32, CV_8U);
        r.raw(p->mDescriptor.ptr(), 32);
        p->mnId = (unsigned long)k;
        p->mbBad = r.i32() != 0;
        p->nObs = r.i32();
        p->mbTrackInView = r.i32() != 0;
        p->mTrackProjX = r.f32(); p->mTrackProjY = r.f32(); p->mTrackProjXR = r.f32();
        p->mnTrackScaleLevel = r.i32();
        p->mTrackViewCos = r.f32();
        W.points.push_back(p);
    }
    for (int vi = 0; vi < nViews; vi++) {
        View *V = new View;
        const int N = r.i32(), frameId = r.i32();
        float T[16];
        r.raw(T, sizeof(T));
        std::vector<float> x = r.vec<float>(N), y = r.vec<float>(N);
        std::vector<int> oct = r.vec<int>(N);
        std::vector<float> ang = r.vec<float>(N), ur = r.vec<float>(N);
        cv::Mat D(N > 0 ? N : 1, 32, CV_8U);
        r.raw(D.ptr(), (size_t)N * 32);
        V->mp = r.vec<int>(N);
        std::vector<unsigned char> outlier = r.vec<unsigned char>(N);
        const int nFv = r.i32();
        std::vector<int> fv = r.vec<int>((size_t)nFv * 2);
        if (!r.ok) return 3;
        std::vector<cv::KeyPoint> keys(N);
        for (int i = 0; i < N; i++) {
            keys[i].pt.x = x[i]; keys[i].pt.y = y[i]; keys[i].octave = oct[i]; keys[i].angle = ang[i];
        }
        DBoW2::FeatureVector featVec;
        for (int i = 0; i < nFv; i++) featVec.addFeature((unsigned)fv[2 * i], (unsigned)fv[2 * i + 1]);
        cv::Mat Tcw = mat_from(T, 4, 4);
        cv::Mat Rcw = Tcw.rowRange(0, 3).colRange(0, 3), tcw = Tcw.rowRange(0, 3).col(3);
        cv::Mat Ow = -(Rcw.t() * tcw);
        std::vector<MapPoint *> mps(N);
        for (int i = 0; i < N; i++) mps[i] = V->mp[i] >= 0 ? W.points[V->mp[i]] : static_cast<MapPoint *>(NULL);

        Frame &F = V->F;
        F.mbf = cam[4]; F.mb = cam[5];
        F.N = N;
        F.mvKeys = keys; F.mvKeysUn = keys;
        F.mvuRight = ur;
        F.mFeatVec = featVec;
        F.mDescriptors = D;
        F.mvpMapPoints = mps;
        F.mvbOutlier.assign(N, false);
        for (int i = 0; i < N; i++) F.mvbOutlier[i] = outlier[i] != 0;
        F.mTcw = Tcw;
        F.mnId = (unsigned long)frameId;
        F.mnScaleLevels = S.nlevels; F.mfScaleFactor = S.scaleFactor; F.mfLogScaleFactor = S.logScaleFactor;
        F.mvScaleFactors = S.sf; F.mvInvScaleFactors = S.inv; F.mvLevelSigma2 = S.sigma2; F.mvInvLevelSigma2 = S.invSigma2;

        KeyFrame &K = V->KF;
        K.mnId = (unsigned long)(1000 + vi);
        K.mnFrameId = (unsigned long)frameId;
        K.mfGridElementWidthInv = cam[10]; K.mfGridElementHeightInv = cam[11];
        K.fx = cam[0]; K.fy = cam[1]; K.cx = cam[2]; K.cy = cam[3]; K.mbf = cam[4]; K.mb = cam[5];
        K.N = N;
        K.mvKeys = keys; K.mvKeysUn = keys;
        K.mvuRight = ur;
        K.mDescriptors = D;
        K.mFeatVec = featVec;
        K.mnScaleLevels = S.nlevels; K.mfScaleFactor = S.scaleFactor; K.mfLogScaleFactor = S.logScaleFactor;
        K.mvScaleFactors = S.sf; K.mvLevelSigma2 = S.sigma2; K.mvInvLevelSigma2 = S.invSigma2;
        K.mnMinX = (int)cam[6]; K.mnMaxX = (int)cam[7]; K.mnMinY = (int)cam[8]; K.mnMaxY = (int)cam[9];   // src/KeyFrame.cc:40-41
        K.Tcw = Tcw;
        K.Ow = Ow;
        K.mvpMapPoints = mps;
        W.views.push_back(V);
    }
    // observations of the KeyFrames: a map point knows the keyframe keypoints it sits on
    for (size_t vi = 0; vi < W.views.size(); vi++)
        for (int i = 0; i < W.views[vi]->KF.N; i++)
            if (W.views[vi]->mp[i] >= 0) W.points[W.views[vi]->mp[i]]->mObservations[&W.views[vi]->KF] = (size_t)i;

    int ip[8];
    float fp[8];
    r.raw(ip, sizeof(ip));   // ip[0] = checkOrientation; the rest per operation
    r.raw(fp, sizeof(fp));   // fp[0] = nnratio, fp[1] = th
    if (!r.ok) return 3;
    ORBmatcher matcher(fp[0], ip[0] != 0);
    Writer w{std::fopen(outPath, "wb")};
    if (!w.f) return 4;
    View &A = *W.views[0];
    View &B = *W.views[nViews > 1 ? 1 : 0];
    int result = 0;

    switch (op) {
    case 1: {   // Tracking::SearchLocalPoints
        const std::vector<int> list = r.vec<int>((size_t)ip[1]);
        result = matcher.SearchByProjection(A.F, W.pointers(list), fp[1]);
        w.i32(result);
        w.vec(W.indices(A.F.mvpMapPoints));
        break;
    }
    case 2: {   // Tracking::TrackWithMotionModel: A = current, B = last
        result = matcher.SearchByProjection(A.F, B.F, fp[1], ip[1] != 0);
        w.i32(result);
        w.vec(W.indices(A.F.mvpMapPoints));
        break;
    }
    case 3: {   // Tracking::Relocalization: A = current frame, B = candidate keyframe
        const std::vector<int> found = r.vec<int>((size_t)ip[1]);
        std::set<MapPoint *> sFound;
        for (int k : found) sFound.insert(W.points[k]);
        result = matcher.SearchByProjection(A.F, &B.KF, sFound, fp[1], ip[2]);
        w.i32(result);
        w.vec(W.indices(A.F.mvpMapPoints));
        break;
    }
    case 4: {   // LoopClosing::ComputeSim3: projection with Scw
        const std::vector<float> scw = r.vec<float>(16);
        const std::vector<int> list = r.vec<int>((size_t)ip[1]);
        const std::vector<int> matchedIn = r.vec<int>((size_t)A.KF.N);
        std::vector<MapPoint *> vpMatched = W.pointers(matchedIn);
        result = matcher.SearchByProjection(&A.KF, mat_from(scw.data(), 4, 4), W.pointers(list), vpMatched, ip[2]);
        w.i32(result);
        w.vec(W.indices(vpMatched));
        break;
    }
    case 5: {   // KeyFrame A against Frame B by vocabulary node
        std::vector<MapPoint *> matches;
        result = matcher.SearchByBoW(&A.KF, B.F, matches);
        w.i32(result);
        w.i32((int)matches.size());
        w.vec(W.indices(matches));
        break;
    }
    case 6: {   // KeyFrame A against KeyFrame B by vocabulary node
        std::vector<MapPoint *> matches;
        result = matcher.SearchByBoW(&A.KF, &B.KF, matches);
        w.i32(result);
        w.i32((int)matches.size());
        w.vec(W.indices(matches));
        break;
    }
    case 7: {   // monocular initialisation: A = F1, B = F2
        const std::vector<float> prev = r.vec<float>((size_t)A.F.N * 2);
        std::vector<cv::Point2f> vbPrevMatched(A.F.N);
        for (int i = 0; i < A.F.N; i++) vbPrevMatched[i] = cv::Point2f(prev[2 * i], prev[2 * i + 1]);
        std::vector<int> vnMatches12;
        result = matcher.SearchForInitialization(A.F, B.F, vbPrevMatched, vnMatches12, ip[1]);
        w.i32(result);
        w.vec(vnMatches12);
        std::vector<float> after((size_t)A.F.N * 2);
        for (int i = 0; i < A.F.N; i++) { after[2 * i] = vbPrevMatched[i].x; after[2 * i + 1] = vbPrevMatched[i].y; }
        w.vec(after);
        break;
    }
    case 8: {   // LocalMapping::CreateNewMapPoints
        const std::vector<float> f12 = r.vec<float>(9);
        std::vector<std::pair<size_t, size_t> > pairs;
        result = matcher.SearchForTriangulation(&A.KF, &B.KF, mat_from(f12.data(), 3, 3), pairs, ip[1] != 0);
        w.i32(result);
        w.i32((int)pairs.size());
        for (size_t k = 0; k < pairs.size(); k++) { w.i32((int)pairs[k].first); w.i32((int)pairs[k].second); }
        break;
    }
    case 9: {   // LoopClosing::ComputeSim3: SearchBySim3
        const std::vector<float> rt = r.vec<float>(12);   // R12 (9), t12 (3)
        const std::vector<int> in12 = r.vec<int>((size_t)A.KF.N);
        std::vector<MapPoint *> vpMatches12 = W.pointers(in12);
        const float s12 = fp[2];
        result = matcher.SearchBySim3(&A.KF, &B.KF, vpMatches12, s12, mat_from(rt.data(), 3, 3), mat_from(rt.data() + 9, 3, 1), fp[1]);
        w.i32(result);
        w.vec(W.indices(vpMatches12));
        break;
    }
    case 10: {   // LocalMapping::SearchInNeighbors
        const std::vector<int> list = r.vec<int>((size_t)ip[1]);
        result = matcher.Fuse(&A.KF, W.pointers(list), fp[1]);
        w.i32(result);
        w.vec(W.indices(A.KF.mvpMapPoints));
        std::vector<int> state;
        for (MapPoint *p : W.points) {
            state.push_back(p->mbBad ? 1 : 0);
            state.push_back(p->nObs);
            state.push_back(W.index_of(p->mpReplaced));
            state.push_back(p->GetIndexInKeyFrame(&A.KF));
        }
        w.vec(state);
        break;
    }
    case 11: {   // LoopClosing::SearchAndFuse
        const std::vector<float> scw = r.vec<float>(16);
        const std::vector<int> list = r.vec<int>((size_t)ip[1]);
        std::vector<MapPoint *> vpReplace(list.size(), static_cast<MapPoint *>(NULL));
        result = matcher.Fuse(&A.KF, mat_from(scw.data(), 4, 4), W.pointers(list), fp[1], vpReplace);
        w.i32(result);
        w.vec(W.indices(vpReplace));
        w.vec(W.indices(A.KF.mvpMapPoints));
        break;
    }
    case 12: {   // the static helpers and constants other files use (src/Frame.cc:471,522,541; src/MapPoint.cc:281)
        w.i32(ORBmatcher::TH_LOW); w.i32(ORBmatcher::TH_HIGH); w.i32(ORBmatcher::HISTO_LENGTH);
        w.i32(A.F.N > 1 ? ORBmatcher::DescriptorDistance(A.F.mDescriptors.row(0), A.F.mDescriptors.row(1)) : -1);
        break;
    }
    default:
        return 2;
    }
    if (!r.ok) return 3;
    std::fclose(w.f);
    std::fclose(r.f);
    for (View *V : W.views) delete V;
    for (MapPoint *p : W.points) delete p;
    if (!quiet) std::printf("matcher_main ok: op %d, result %d\n", op, result);
    return 0;
}

}  // namespace

int main(int argc, char **argv) {
    if (argc >= 5 && std::strcmp(argv[1], "--concurrent") == 0) {
        const int reps = std::atoi(argv[2]);
        const int nCases = (argc - 3) / 2;
        if (reps < 1 || nCases < 1 || (argc - 3) % 2) return 2;
        {
            Reader r{std::fopen(argv[3], "rb")};
            if (!r.f) return 3;
            r.i32(); r.i32(); r.i32();
            float cam[12];
            r.raw(cam, sizeof(cam));
            if (!r.ok) return 3;
            set_camera_statics(cam);
            std::fclose(r.f);
        }
        std::atomic<int> ready(0), failed(0);
        std::vector<std::thread> threads;
        for (int c = 0; c < nCases; c++)
            threads.emplace_back([&, c] {
                ready++;
                while (ready.load() < nCases) std::this_thread::yield();   // all threads enter their first search together
                for (int k = 0; k < reps; k++) {
                    const int rc = run_case(argv[3 + 2 * c], argv[4 + 2 * c], false, true);
                    if (rc) {
                        failed = rc;
                        return;
                    }
                }
            });
        for (std::thread &t : threads) t.join();
        if (failed.load()) return failed.load();
        std::printf("matcher_main ok: %d cases x %d repetitions on %d threads\n", nCases, reps, nCases);
        return 0;
    }
    if (argc < 3) return 2;
    return run_case(argv[1], argv[2], true, false);
}
