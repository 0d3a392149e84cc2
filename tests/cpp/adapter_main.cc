// adapter_main.cc -- exercises ORB_SLAM2::ORBextractor (include/ORBextractor.h + adapters/ORBextractor_orbfe.cc) the way
// Frame::ExtractORB does (reference src/Frame.cc:247-253), on a raw 8-bit image file, and dumps the outputs so the
// pytest driver can compare them with the oracle.  usage: adapter_main in.raw w h nfeatures out.bin
//                                                          adapter_main stereo left.raw right.raw w h nfeatures bf b out.bin
//                                                          adapter_main bow voc.txt desc.raw n levelsup out.bin
//                                                          adapter_main project case.bin out.bin
//                                                          adapter_main bowmatch case.bin out.bin
//                                                          adapter_main track case.bin out.bin
//                                                          adapter_main guided case.bin out.bin
//                                                          adapter_main triangulate case.bin out.bin
// The guided mode runs the remaining window searches of orbfe::ProjectionSearch (searchSim3Projection, searchRelocalization,
// fuseSelect, SearchBySim3, searchForInitialization) and the triangulate mode orbfe::TriangulationMatcher; layouts in
// tests/test_guided_search.py.
// The track mode runs orbfe::ProjectionSearch::searchLastFrame: SearchByProjection(CurrentFrame, LastFrame, th, bMono)
// (src/ORBmatcher.cc:1328-1470) from the window search on.
// The bowmatch mode runs orbfe::BowMatcher::SearchByBoW (src/ORBmatcher.cc:159-288 / :522-655) on two feature sets with
// their FeatureVectors read from case.bin (layout in tests/test_gpu_adapter.py).
// The project mode replays ORBmatcher::SearchByProjection(Frame&, vpMapPoints, th) (src/ORBmatcher.cc:45-129) on a frame
// and a list of projected map points read from case.bin (layout in tests/test_gpu_adapter.py).
// The bow mode loads a vocabulary in the ORBvoc.txt format and runs Frame::ComputeBoW's transform (src/Frame.cc:395-402).
// The stereo mode follows the Frame stereo constructor (src/Frame.cc:59-93): two extractors, then ComputeStereoMatches.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <map>

#include "ORBextractor.h"
#include "ORBVocabulary_orbfe.h"
#include "ORBmatcher_orbfe.h"

static bool read_raw(const char *path, std::vector<unsigned char> &buf) {
    FILE *fi = std::fopen(path, "rb");
    if (!fi) return false;
    const bool ok = std::fread(buf.data(), 1, buf.size(), fi) == buf.size();
    std::fclose(fi);
    return ok;
}

static int stereo_main(char **argv) {
    const int w = std::atoi(argv[4]), h = std::atoi(argv[5]), nf = std::atoi(argv[6]);
    const float bf = (float)std::atof(argv[7]), b = (float)std::atof(argv[8]);
    std::vector<unsigned char> bl((size_t)w * h), br((size_t)w * h);
    if (!read_raw(argv[2], bl) || !read_raw(argv[3], br)) return 3;
    cv::Mat imL(h, w, CV_8UC1, bl.data(), (size_t)w), imR(h, w, CV_8UC1, br.data(), (size_t)w);
    ORB_SLAM2::ORBextractor exL(nf, 1.2f, 8, 20, 7), exR(nf, 1.2f, 8, 20, 7);
    std::vector<cv::KeyPoint> kl, kr;
    cv::Mat dl, dr;
    exL(imL, cv::Mat(), kl, dl);
    exR(imR, cv::Mat(), kr, dr);
    orbfe::StereoMatcher sm;
    std::vector<float> uRight, depth;
    const int rc = sm.ComputeStereoMatches(exL.orbfeHandle(), exR.orbfeHandle(), kl, dl, kr, dr, bf, b, uRight, depth);
    if (rc != 0) { std::fprintf(stderr, "stereo: %s\n", orbfe_last_error()); return 5; }
    FILE *fo = std::fopen(argv[9], "wb");
    if (!fo) return 4;
    const int n = (int)kl.size();
    std::fwrite(&n, 4, 1, fo);
    std::fwrite(uRight.data(), 4, n, fo);
    std::fwrite(depth.data(), 4, n, fo);
    std::fclose(fo);
    int matched = 0;
    for (float d : depth) matched += d > 0;
    std::printf("adapter stereo ok: %d left keypoints, %d with depth\n", n, matched);
    return 0;
}

static int bow_main(char **argv) {
    const int n = std::atoi(argv[4]), levelsup = std::atoi(argv[5]);
    std::vector<unsigned char> buf((size_t)n * 32);
    if (!read_raw(argv[3], buf)) return 3;
    orbfe::Vocabulary voc;
    if (!voc.loadFromTextFile(argv[2])) { std::fprintf(stderr, "bow: cannot load %s: %s\n", argv[2], orbfe_last_error()); return 5; }
    cv::Mat desc(n, 32, CV_8UC1, buf.data(), (size_t)32);
    std::map<unsigned, double> bow;                          // DBoW2::BowVector
    std::map<unsigned, std::vector<unsigned> > fv;           // DBoW2::FeatureVector
    if (voc.transform(desc, bow, fv, levelsup) != 0) { std::fprintf(stderr, "bow: %s\n", orbfe_last_error()); return 6; }
    FILE *fo = std::fopen(argv[6], "wb");
    if (!fo) return 4;
    int nb = (int)bow.size(), nf = 0;
    for (auto &kv : fv) nf += (int)kv.second.size();
    std::fwrite(&nb, 4, 1, fo);
    std::fwrite(&nf, 4, 1, fo);
    for (auto &kv : bow) { int id = (int)kv.first; std::fwrite(&id, 4, 1, fo); }
    for (auto &kv : bow) std::fwrite(&kv.second, 8, 1, fo);
    for (auto &kv : fv) for (unsigned ft : kv.second) { int a = (int)kv.first, b = (int)ft; std::fwrite(&a, 4, 1, fo); std::fwrite(&b, 4, 1, fo); }
    std::fclose(fo);
    std::printf("adapter bow ok: %d words, %d features in %d nodes\n", nb, nf, (int)fv.size());
    return 0;
}

static int project_main(char **argv) {
    FILE *fi = std::fopen(argv[2], "rb");
    if (!fi) return 3;
    int n = 0, nq = 0, nl = 0;
    float f[6];
    bool ok = std::fread(&n, 4, 1, fi) == 1 && std::fread(&nq, 4, 1, fi) == 1 && std::fread(f, 4, 6, fi) == 6 &&
              std::fread(&nl, 4, 1, fi) == 1;
    if (!ok || n < 0 || nq < 0 || nl < 1 || nl > 64) return 3;
    std::vector<float> sf(nl), kx(n), ky(n), ur(n), px(nq), py(nq), pxr(nq), vc(nq);
    std::vector<int> oct(n), lvl(nq);
    std::vector<unsigned char> desc((size_t)n * 32 + 1), occ(n), qd((size_t)nq * 32 + 1);
    auto rd = [&](void *p, size_t bytes) { return bytes == 0 || std::fread(p, 1, bytes, fi) == bytes; };
    ok = rd(sf.data(), 4 * (size_t)nl) && rd(kx.data(), 4 * (size_t)n) && rd(ky.data(), 4 * (size_t)n) && rd(oct.data(), 4 * (size_t)n) &&
         rd(ur.data(), 4 * (size_t)n) && rd(desc.data(), 32 * (size_t)n) && rd(occ.data(), (size_t)n) && rd(px.data(), 4 * (size_t)nq) &&
         rd(py.data(), 4 * (size_t)nq) && rd(pxr.data(), 4 * (size_t)nq) && rd(lvl.data(), 4 * (size_t)nq) && rd(vc.data(), 4 * (size_t)nq) &&
         rd(qd.data(), 32 * (size_t)nq);
    std::fclose(fi);
    if (!ok) return 3;
    std::vector<cv::KeyPoint> keys(n);
    for (int i = 0; i < n; i++) {
        keys[i].pt.x = kx[i];
        keys[i].pt.y = ky[i];
        keys[i].octave = oct[i];
    }
    cv::Mat D(n, 32, CV_8UC1, desc.data(), (size_t)32);
    orbfe::ProjectionSearch ps;
    if (ps.build(keys, D, ur, f[0], f[1], f[2], f[3]) != 0) { std::fprintf(stderr, "project: %s\n", orbfe_last_error()); return 5; }
    std::vector<orbfe::ProjectedPoint> pts(nq);
    for (int i = 0; i < nq; i++) pts[i] = {px[i], py[i], pxr[i], lvl[i], vc[i], qd.data() + (size_t)i * 32};
    std::vector<int> matchIdx;
    int nmatches = 0;
    if (ps.search(pts, sf, f[4], f[5], occ, matchIdx, &nmatches) != 0) { std::fprintf(stderr, "project: %s\n", orbfe_last_error()); return 6; }
    FILE *fo = std::fopen(argv[3], "wb");
    if (!fo) return 4;
    const int rq = (int)ps.requeried();
    std::fwrite(&nmatches, 4, 1, fo);
    std::fwrite(&rq, 4, 1, fo);
    std::fwrite(matchIdx.data(), 4, nq, fo);
    std::fwrite(occ.data(), 1, n, fo);
    std::fclose(fo);
    std::printf("adapter project ok: %d map points, %d matches, %d searched again\n", nq, nmatches, rq);
    return 0;
}

static int track_main(char **argv) {
    FILE *fi = std::fopen(argv[2], "rb");
    if (!fi) return 3;
    int n = 0, nq = 0, checkOri = 0;
    float f[4];
    if (std::fread(&n, 4, 1, fi) != 1 || std::fread(&nq, 4, 1, fi) != 1 || std::fread(&checkOri, 4, 1, fi) != 1 ||
        std::fread(f, 4, 4, fi) != 4 || n < 0 || nq < 0)
        return 3;
    std::vector<float> kx(n), ky(n), kur(n), kang(n), pu(nq), pv(nq), pur(nq), prad(nq), pang(nq);
    std::vector<int> oct(n), pmin(nq), pmax(nq);
    std::vector<unsigned char> desc((size_t)n * 32 + 1), occ(n), pobs(nq), qd((size_t)nq * 32 + 1);
    auto rd = [&](void *p, size_t bytes) { return bytes == 0 || std::fread(p, 1, bytes, fi) == bytes; };
    const bool ok = rd(kx.data(), 4 * (size_t)n) && rd(ky.data(), 4 * (size_t)n) && rd(oct.data(), 4 * (size_t)n) && rd(kur.data(), 4 * (size_t)n) &&
                    rd(kang.data(), 4 * (size_t)n) && rd(desc.data(), 32 * (size_t)n) && rd(occ.data(), (size_t)n) && rd(pu.data(), 4 * (size_t)nq) &&
                    rd(pv.data(), 4 * (size_t)nq) && rd(pur.data(), 4 * (size_t)nq) && rd(prad.data(), 4 * (size_t)nq) && rd(pmin.data(), 4 * (size_t)nq) &&
                    rd(pmax.data(), 4 * (size_t)nq) && rd(pang.data(), 4 * (size_t)nq) && rd(pobs.data(), (size_t)nq) && rd(qd.data(), 32 * (size_t)nq);
    std::fclose(fi);
    if (!ok) return 3;
    std::vector<cv::KeyPoint> keys(n);
    for (int i = 0; i < n; i++) {
        keys[i].pt.x = kx[i];
        keys[i].pt.y = ky[i];
        keys[i].octave = oct[i];
        keys[i].angle = kang[i];
    }
    cv::Mat D(n, 32, CV_8UC1, desc.data(), (size_t)32);
    orbfe::ProjectionSearch ps;
    if (ps.build(keys, D, kur, f[0], f[1], f[2], f[3]) != 0) { std::fprintf(stderr, "track: %s\n", orbfe_last_error()); return 5; }
    std::vector<orbfe::TrackedPoint> pts(nq);
    for (int i = 0; i < nq; i++)
        pts[i] = {pu[i], pv[i], pur[i], prad[i], pmin[i], pmax[i], pang[i], qd.data() + (size_t)i * 32, pobs[i] != 0};
    std::vector<int> assigned;
    int nmatches = 0;
    if (ps.searchLastFrame(pts, kang, checkOri != 0, occ, assigned, &nmatches) != 0) {
        std::fprintf(stderr, "track: %s\n", orbfe_last_error());
        return 6;
    }
    FILE *fo = std::fopen(argv[3], "wb");
    if (!fo) return 4;
    const int rq = (int)ps.requeried();
    std::fwrite(&nmatches, 4, 1, fo);
    std::fwrite(&rq, 4, 1, fo);
    std::fwrite(assigned.data(), 4, n, fo);
    std::fwrite(occ.data(), 1, n, fo);
    std::fclose(fo);
    std::printf("adapter track ok: %d points, %d matches, %d searched again\n", nq, nmatches, rq);
    return 0;
}

static int bowmatch_main(char **argv) {
    FILE *fi = std::fopen(argv[2], "rb");
    if (!fi) return 3;
    int hdr[6];   // n1, n2, number of (node, feature) entries on side 1 and 2, maxDist, checkOrientation
    float ratio = 0;
    if (std::fread(hdr, 4, 6, fi) != 6 || std::fread(&ratio, 4, 1, fi) != 1) return 3;
    const int n1 = hdr[0], n2 = hdr[1], e1 = hdr[2], e2 = hdr[3];
    std::vector<unsigned char> d1((size_t)n1 * 32 + 1), d2((size_t)n2 * 32 + 1), valid1(n1), blocked2(n2);
    std::vector<float> ang1(n1), ang2(n2);
    std::vector<int> ent1((size_t)e1 * 2 + 1), ent2((size_t)e2 * 2 + 1);   // (node id, feature index) in insertion order
    auto rd = [&](void *p, size_t bytes) { return bytes == 0 || std::fread(p, 1, bytes, fi) == bytes; };
    const bool ok = rd(d1.data(), (size_t)n1 * 32) && rd(d2.data(), (size_t)n2 * 32) && rd(valid1.data(), n1) && rd(blocked2.data(), n2) &&
                    rd(ang1.data(), 4 * (size_t)n1) && rd(ang2.data(), 4 * (size_t)n2) && rd(ent1.data(), 8 * (size_t)e1) &&
                    rd(ent2.data(), 8 * (size_t)e2);
    std::fclose(fi);
    if (!ok) return 3;
    orbfe::BowMatcher::FeatureVector fv1, fv2;
    for (int i = 0; i < e1; i++) fv1[(unsigned)ent1[2 * i]].push_back((unsigned)ent1[2 * i + 1]);   // FeatureVector::addFeature
    for (int i = 0; i < e2; i++) fv2[(unsigned)ent2[2 * i]].push_back((unsigned)ent2[2 * i + 1]);
    std::vector<cv::KeyPoint> k1(n1), k2(n2);
    for (int i = 0; i < n1; i++) k1[i].angle = ang1[i];
    for (int i = 0; i < n2; i++) k2[i].angle = ang2[i];
    cv::Mat D1(n1, 32, CV_8UC1, d1.data(), (size_t)32), D2(n2, 32, CV_8UC1, d2.data(), (size_t)32);
    orbfe::BowMatcher bm;
    std::vector<int> match12;
    int nmatches = 0;
    if (bm.SearchByBoW(fv1, D1, k1, valid1, fv2, D2, k2, blocked2, hdr[4], ratio, hdr[5] != 0, match12, &nmatches) != 0) {
        std::fprintf(stderr, "bowmatch: %s\n", orbfe_last_error());
        return 6;
    }
    FILE *fo = std::fopen(argv[3], "wb");
    if (!fo) return 4;
    std::fwrite(&nmatches, 4, 1, fo);
    std::fwrite(match12.data(), 4, n1, fo);
    std::fclose(fo);
    std::printf("adapter bowmatch ok: %d x %d features, %d matches\n", n1, n2, nmatches);
    return 0;
}

namespace {
struct GuidedSide {   // one frame / keyframe with its grid, plus the points searched in it
    int n = 0, nq = 0;
    std::vector<float> kx, ky, kur, kang, u, v, ur, radius, angle;
    std::vector<int> oct, level, index;
    std::vector<unsigned char> desc, occ, qd;
    std::vector<cv::KeyPoint> keys;
    std::vector<orbfe::ProjectionSearch::WindowPoint> pts;
    bool read(FILE *fi) {
        auto rd = [&](void *p, size_t bytes) { return bytes == 0 || std::fread(p, 1, bytes, fi) == bytes; };
        if (!rd(&n, 4) || !rd(&nq, 4) || n < 0 || nq < 0) return false;
        kx.resize(n); ky.resize(n); kur.resize(n); kang.resize(n); oct.resize(n); desc.resize((size_t)n * 32 + 1); occ.resize(n);
        u.resize(nq); v.resize(nq); ur.resize(nq); radius.resize(nq); angle.resize(nq); level.resize(nq); index.resize(nq);
        qd.resize((size_t)nq * 32 + 1);
        const size_t a = 4 * (size_t)n, b = 4 * (size_t)nq;
        if (!(rd(kx.data(), a) && rd(ky.data(), a) && rd(oct.data(), a) && rd(kur.data(), a) && rd(kang.data(), a) &&
              rd(desc.data(), 32 * (size_t)n) && rd(occ.data(), n) && rd(u.data(), b) && rd(v.data(), b) && rd(ur.data(), b) &&
              rd(radius.data(), b) && rd(level.data(), b) && rd(angle.data(), b) && rd(index.data(), b) && rd(qd.data(), 32 * (size_t)nq)))
            return false;
        keys.resize(n);
        for (int i = 0; i < n; i++) {
            keys[i].pt.x = kx[i]; keys[i].pt.y = ky[i]; keys[i].octave = oct[i]; keys[i].angle = kang[i];
        }
        pts.resize(nq);
        for (int i = 0; i < nq; i++)
            pts[i] = {u[i], v[i], ur[i], radius[i], level[i], angle[i], qd.data() + (size_t)i * 32, index[i]};
        return true;
    }
};
}  // namespace

static int guided_main(char **argv) {
    FILE *fi = std::fopen(argv[2], "rb");
    if (!fi) return 3;
    int hdr[4];      // kind, checkOrientation, ORBdist / n1 (initialisation), use the Fuse gate
    float f[3];      // mfGridElementWidthInv, mfGridElementHeightInv, nnratio
    float invSigma2[8];
    if (std::fread(hdr, 4, 4, fi) != 4 || std::fread(f, 4, 3, fi) != 3 || std::fread(invSigma2, 4, 8, fi) != 8) return 3;
    GuidedSide A, B;
    if (!A.read(fi)) return 3;
    const int kind = hdr[0];
    if (kind == 4 && !B.read(fi)) return 3;
    std::fclose(fi);
    orbfe::ProjectionSearch psA, psB;
    cv::Mat DA(A.n, 32, CV_8UC1, A.desc.data(), (size_t)32);
    if (psA.build(A.keys, DA, A.kur, 0.0f, 0.0f, f[0], f[1]) != 0) { std::fprintf(stderr, "guided: %s\n", orbfe_last_error()); return 5; }
    std::vector<int> out1;
    std::vector<unsigned char> out2;
    int count = 0, rc = 0;
    if (kind == 1) {
        out2 = A.occ;
        rc = psA.searchSim3Projection(A.pts, out2, out1, &count);
    } else if (kind == 2) {
        out2 = A.occ;
        rc = psA.searchRelocalization(A.pts, A.kang, hdr[2], hdr[1] != 0, out2, out1, &count);
    } else if (kind == 3) {
        std::vector<float> s2;
        if (hdr[3]) s2.assign(invSigma2, invSigma2 + 8);
        rc = psA.fuseSelect(A.pts, s2, out1, &count);
    } else if (kind == 4) {
        cv::Mat DB(B.n, 32, CV_8UC1, B.desc.data(), (size_t)32);
        if (psB.build(B.keys, DB, B.kur, 0.0f, 0.0f, f[0], f[1]) != 0) return 5;
        // A.pts: MapPoints of KF1 (= side A) searched in KF2's grid (side B); B.pts the other way round
        rc = orbfe::ProjectionSearch::SearchBySim3(psA, psB, A.pts, B.pts, out1, &count);
    } else if (kind == 5) {
        rc = psA.searchForInitialization(A.pts, hdr[2], A.kang, f[2], hdr[1] != 0, out1, &count);
    } else {
        return 2;
    }
    if (rc != 0) { std::fprintf(stderr, "guided: %s\n", orbfe_last_error()); return 6; }
    FILE *fo = std::fopen(argv[3], "wb");
    if (!fo) return 4;
    const int rq = (int)psA.requeried(), n1 = (int)out1.size(), n2 = (int)out2.size();
    std::fwrite(&count, 4, 1, fo);
    std::fwrite(&rq, 4, 1, fo);
    std::fwrite(&n1, 4, 1, fo);
    std::fwrite(&n2, 4, 1, fo);
    std::fwrite(out1.data(), 4, n1, fo);
    std::fwrite(out2.data(), 1, n2, fo);
    std::fclose(fo);
    std::printf("adapter guided ok: kind %d, %d points, count %d, %d searched again\n", kind, A.nq, count, rq);
    return 0;
}

static int triangulate_main(char **argv) {
    FILE *fi = std::fopen(argv[2], "rb");
    if (!fi) return 3;
    int hdr[6];      // n1, n2, entries on side 1 and 2, bOnlyStereo, checkOrientation
    float F[9], e[2], sf[8], s2[8];
    if (std::fread(hdr, 4, 6, fi) != 6 || std::fread(F, 4, 9, fi) != 9 || std::fread(e, 4, 2, fi) != 2 || std::fread(sf, 4, 8, fi) != 8 ||
        std::fread(s2, 4, 8, fi) != 8)
        return 3;
    const int n1 = hdr[0], n2 = hdr[1], e1 = hdr[2], e2 = hdr[3];
    std::vector<unsigned char> d1((size_t)n1 * 32 + 1), d2((size_t)n2 * 32 + 1), mp1(n1), mp2(n2);
    std::vector<float> x1(n1), y1(n1), a1(n1), u1(n1), x2(n2), y2(n2), a2(n2), u2(n2);
    std::vector<int> o2(n2), ent1((size_t)e1 * 2 + 1), ent2((size_t)e2 * 2 + 1);
    auto rd = [&](void *p, size_t bytes) { return bytes == 0 || std::fread(p, 1, bytes, fi) == bytes; };
    const size_t a = 4 * (size_t)n1, b = 4 * (size_t)n2;
    const bool ok = rd(d1.data(), 32 * (size_t)n1) && rd(mp1.data(), n1) && rd(u1.data(), a) && rd(x1.data(), a) && rd(y1.data(), a) &&
                    rd(a1.data(), a) && rd(d2.data(), 32 * (size_t)n2) && rd(mp2.data(), n2) && rd(u2.data(), b) && rd(x2.data(), b) &&
                    rd(y2.data(), b) && rd(a2.data(), b) && rd(o2.data(), b) && rd(ent1.data(), 8 * (size_t)e1) &&
                    rd(ent2.data(), 8 * (size_t)e2);
    std::fclose(fi);
    if (!ok) return 3;
    orbfe::TriangulationMatcher::FeatureVector fv1, fv2;
    for (int i = 0; i < e1; i++) fv1[(unsigned)ent1[2 * i]].push_back((unsigned)ent1[2 * i + 1]);
    for (int i = 0; i < e2; i++) fv2[(unsigned)ent2[2 * i]].push_back((unsigned)ent2[2 * i + 1]);
    std::vector<cv::KeyPoint> k1(n1), k2(n2);
    for (int i = 0; i < n1; i++) { k1[i].pt.x = x1[i]; k1[i].pt.y = y1[i]; k1[i].angle = a1[i]; }
    for (int i = 0; i < n2; i++) { k2[i].pt.x = x2[i]; k2[i].pt.y = y2[i]; k2[i].angle = a2[i]; k2[i].octave = o2[i]; }
    cv::Mat D1(n1, 32, CV_8UC1, d1.data(), (size_t)32), D2(n2, 32, CV_8UC1, d2.data(), (size_t)32);
    orbfe::TriangulationMatcher tm;
    std::vector<std::pair<size_t, size_t> > pairs;
    int nmatches = 0;
    if (tm.SearchForTriangulation(fv1, D1, k1, mp1, u1, fv2, D2, k2, mp2, u2, F, e[0], e[1], std::vector<float>(sf, sf + 8),
                                  std::vector<float>(s2, s2 + 8), hdr[4] != 0, hdr[5] != 0, pairs, &nmatches) != 0) {
        std::fprintf(stderr, "triangulate: %s\n", orbfe_last_error());
        return 6;
    }
    FILE *fo = std::fopen(argv[3], "wb");
    if (!fo) return 4;
    const int np = (int)pairs.size();
    std::fwrite(&nmatches, 4, 1, fo);
    std::fwrite(&np, 4, 1, fo);
    for (auto &pr : pairs) { int v[2] = {(int)pr.first, (int)pr.second}; std::fwrite(v, 4, 2, fo); }
    std::fclose(fo);
    std::printf("adapter triangulate ok: %d x %d features, %d pairs\n", n1, n2, np);
    return 0;
}

// latency mode: wall clock around ORBextractor::operator() on one host image, as Examples/Monocular/mono_tum.cc:88-104 times
// TrackMonocular (steady_clock, median over the frames); once with the mvImagePyramid mirror and once without.
//   adapter_main latency in.raw w h nfeatures reps
static int latency_main(char **argv) {
    const int w = std::atoi(argv[3]), h = std::atoi(argv[4]), nf = std::atoi(argv[5]), reps = std::atoi(argv[6]);
    std::vector<unsigned char> buf((size_t)w * h);
    if (!read_raw(argv[2], buf)) return 3;
    cv::Mat im(h, w, CV_8UC1, buf.data(), (size_t)w);
    double med[2] = {0, 0};
    int nkp = 0;
    for (int mirror = 1; mirror >= 0; mirror--) {
        setenv("ORBFE_PYRAMID_MIRROR", mirror ? "1" : "0", 1);
        ORB_SLAM2::ORBextractor ex(nf, 1.2f, 8, 20, 7);
        std::vector<cv::KeyPoint> keys;
        cv::Mat desc;
        for (int i = 0; i < 5; i++) ex(im, cv::Mat(), keys, desc);   // warm-up: workspace allocation, code object load
        std::vector<double> t(reps);
        for (int i = 0; i < reps; i++) {
            const auto t0 = std::chrono::steady_clock::now();
            ex(im, cv::Mat(), keys, desc);
            t[i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
        std::sort(t.begin(), t.end());
        med[1 - mirror] = t[reps / 2];
        nkp = (int)keys.size();
    }
    std::printf("{\"median_ms_with_pyramid_mirror\": %.4f, \"median_ms_without_pyramid_mirror\": %.4f, \"keypoints\": %d, \"reps\": %d}\n",
                med[0], med[1], nkp, reps);
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 7 && std::strcmp(argv[1], "latency") == 0) return latency_main(argv);
    if (argc >= 4 && std::strcmp(argv[1], "guided") == 0) return guided_main(argv);
    if (argc >= 4 && std::strcmp(argv[1], "triangulate") == 0) return triangulate_main(argv);
    if (argc >= 4 && std::strcmp(argv[1], "bowmatch") == 0) return bowmatch_main(argv);
    if (argc >= 4 && std::strcmp(argv[1], "track") == 0) return track_main(argv);
    if (argc >= 4 && std::strcmp(argv[1], "project") == 0) return project_main(argv);
    if (argc >= 7 && std::strcmp(argv[1], "bow") == 0) return bow_main(argv);
    if (argc >= 10 && std::strcmp(argv[1], "stereo") == 0) return stereo_main(argv);
    if (argc < 6) return 2;
    const int w = std::atoi(argv[2]), h = std::atoi(argv[3]), nf = std::atoi(argv[4]);
    std::vector<unsigned char> buf((size_t)w * h);
    FILE *fi = std::fopen(argv[1], "rb");
    if (!fi || std::fread(buf.data(), 1, buf.size(), fi) != buf.size()) return 3;
    std::fclose(fi);
    cv::Mat im(h, w, CV_8UC1, buf.data(), (size_t)w);
    ORB_SLAM2::ORBextractor ex(nf, 1.2f, 8, 20, 7);
    std::vector<cv::KeyPoint> keys;
    cv::Mat desc;
    ex(im, cv::Mat(), keys, desc);
    FILE *fo = std::fopen(argv[5], "wb");
    if (!fo) return 4;
    int n = (int)keys.size();
    std::fwrite(&n, 4, 1, fo);
    std::fwrite(keys.data(), sizeof(cv::KeyPoint), n, fo);
    for (int i = 0; i < n; i++) std::fwrite(desc.ptr(i), 1, 32, fo);
    // mvImagePyramid[3]: ROI view + border, as Frame::ComputeStereoMatches reads it
    const cv::Mat &p = ex.mvImagePyramid[3];
    std::fwrite(&p.cols, 4, 1, fo);
    std::fwrite(&p.rows, 4, 1, fo);
    for (int y = -19; y < p.rows + 19; y++) std::fwrite(p.data + (long)y * (long)p.step - 19, 1, p.cols + 38, fo);
    int d = orbfe_DescriptorDistance(desc.roi(0, 0, 32, 1), desc.roi(0, 1, 32, 1));
    std::fwrite(&d, 4, 1, fo);
    std::fclose(fo);
    std::printf("adapter ok: %d keypoints, levels %d, scale %.3f\n", n, ex.GetLevels(), ex.GetScaleFactor());
    return 0;
}
