/*
 * orb_oracle.cpp -- CPU ORACLE of the ORB front end.  TEST INFRASTRUCTURE ONLY
 * (see orb_oracle.h).  PARITY UNPINNED: the reference has no tests or golden
 * vectors and cannot be compiled here (OpenCV absent).
 *
 * Every function cites the reference lines (relative to /root/reference) or
 * the OpenCV 2.4.11/3.2 generic-path semantics that it
 * restates.  Written from scratch; the reference keeps its data in cv::Mat /
 * cv::KeyPoint / std::list<ExtractorNode>, this file uses flat arrays and its
 * own node type, but visits, splits and emits in the same order.
 *
 * Build: g++ -O3 -ffp-contract=off -fno-fast-math (contraction off: the float
 * expressions x*b + y*a of src/ORBextractor.cc:119-120 are pinned to their
 * non-FMA value).
 *
 * Intentional, documented pins where the reference itself is not a function
 * of its inputs:
 *   (1) DistributeOctTree sorts pair<int,ExtractorNode*> (src/ORBextractor.cc:684)
 *       -- ties between equal-size nodes are broken by heap address.  The
 *       oracle breaks them by creation order (later-created node first).
 *   (2) cos/sin of the keypoint angle resolve to std::cos(float)/std::sin(float)
 *       (`using namespace std`, src/ORBextractor.cc:67,113), i.e. libm
 *       cosf/sinf; the oracle calls the same libm functions of this host
 *       (glibc 2.35 here; tools/sincos_census.c compares the device's routine
 *       with them over every float angle in [0, 360): none differs).
 */
#include "orb_oracle.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <list>
#include <thread>
#include <utility>
#include <map>
#include <vector>

namespace {

static const int8_t kPattern[1024] = {
#include "pattern31.inc"
};

const int PATCH_SIZE = 31;       // src/ORBextractor.cc:72
const int HALF_PATCH_SIZE = 15;  // :73
const int EDGE_THRESHOLD = 19;   // :74

/* cvRound: round to nearest, ties to even (SSE2 cvtsd2si under the default rounding mode). Appendix A.1 */
inline int cv_round(double v) { return (int)std::nearbyint(v); }
inline int cv_floor(double v) { return (int)std::floor(v); }
inline int cv_ceil(double v) { return (int)std::ceil(v); }
inline short sat_s16(float v) {
    int i = cv_round(v);
    return (short)(i < -32768 ? -32768 : i > 32767 ? 32767 : i);
}
inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
inline int refl101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * (n - 1) - p;
    return p;
}

struct Img {
    int w = 0, h = 0;
    std::vector<uint8_t> d;
    void alloc(int w_, int h_) { w = w_; h = h_; d.assign((size_t)w * h, 0); }
    const uint8_t *row(int y) const { return d.data() + (size_t)y * w; }
    uint8_t *row(int y) { return d.data() + (size_t)y * w; }
};

struct Kp {  // the subset of cv::KeyPoint the extractor touches
    float x, y, size, angle, response;
    int octave;
};

/* ------------------------------------------------------------------ resize */
/* cv::resize INTER_LINEAR coefficient set-up, Appendix A.2 */
void resize_tables(int ssize, int dsize, std::vector<int> &ofs, std::vector<short> &coef, bool is_x) {
    double inv_scale = (double)dsize / ssize;
    double scale = 1. / inv_scale;
    ofs.resize(dsize);
    coef.resize((size_t)dsize * 2);
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = cv_floor(f);
        f -= s;
        if (is_x) {  // only the x set-up clamps the fraction; rows are clipped when fetched
            if (s < 0) { f = 0; s = 0; }
            if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        }
        ofs[d] = s;
        coef[2 * d + 0] = sat_s16((1.f - f) * 2048);
        coef[2 * d + 1] = sat_s16(f * 2048);
    }
}

void resize_linear_u8(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst, int dw, int dh, int dstride) {
    std::vector<int> xofs, yofs;
    std::vector<short> alpha, beta;
    resize_tables(sw, dw, xofs, alpha, true);
    resize_tables(sh, dh, yofs, beta, false);
    std::vector<int> h0(dw), h1(dw);
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = yofs[dy];
        int r0 = sy0 < 0 ? 0 : sy0 >= sh ? sh - 1 : sy0;          // clip(sy0 + k, 0, sh)
        int r1 = sy0 + 1 < 0 ? 0 : sy0 + 1 >= sh ? sh - 1 : sy0 + 1;
        const uint8_t *S0 = src + (size_t)r0 * sstride, *S1 = src + (size_t)r1 * sstride;
        for (int dx = 0; dx < dw; dx++) {
            int sx = xofs[dx];
            if (sx + 1 < sw) {
                h0[dx] = S0[sx] * alpha[2 * dx] + S0[sx + 1] * alpha[2 * dx + 1];
                h1[dx] = S1[sx] * alpha[2 * dx] + S1[sx + 1] * alpha[2 * dx + 1];
            } else {  // dx >= xmax: single tap times INTER_RESIZE_COEF_SCALE
                h0[dx] = S0[sx] * 2048;
                h1[dx] = S1[sx] * 2048;
            }
        }
        int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
        uint8_t *D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; dx++)
            D[dx] = (uint8_t)((((b0 * (h0[dx] >> 4)) >> 16) + ((b1 * (h1[dx] >> 4)) >> 16) + 2) >> 2);
    }
}

/* -------------------------------------------------------------------- blur */
/* getGaussianKernel(7, 2.0, CV_32F) then convertTo(CV_32S, 256): Appendix A.6 */
void gauss7_kernel(int k7[7]) {
    float cf[7];
    double sum = 0;
    const double sigma = 2.0, scale2X = -0.5 / (sigma * sigma);
    for (int i = 0; i < 7; i++) {
        double x = i - 3;
        double t = std::exp(scale2X * x * x);
        cf[i] = (float)t;
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < 7; i++) cf[i] = (float)(cf[i] * sum);
    for (int i = 0; i < 7; i++) k7[i] = cv_round(cf[i] * 256.f);
}

void gauss7_u8(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride) {
    int k[7];
    gauss7_kernel(k);
    std::vector<int> R((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t *s = src + (size_t)y * sstride;
        for (int x = 0; x < w; x++) {
            int acc = 0;
            for (int i = 0; i < 7; i++) acc += k[i] * s[refl101(x + i - 3, w)];
            R[(size_t)y * w + x] = acc;
        }
    }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int acc = 0;
            for (int j = 0; j < 7; j++) acc += k[j] * R[(size_t)refl101(y + j - 3, h) * w + x];
            dst[(size_t)y * dstride + x] = sat_u8((acc + (1 << 15)) >> 16);
        }
}

/* -------------------------------------------------------------------- FAST */
/* cv::FAST_t<16> with nonmax suppression, Appendix A.4 */
void fast_offsets(int pixel[25], int stride) {
    static const int off[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
                                   {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};
    for (int k = 0; k < 16; k++) pixel[k] = off[k][0] + off[k][1] * stride;
    for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
}

int fast_corner_score(const uint8_t *ptr, const int pixel[25], int threshold) {
    const int N = 25;
    int v = ptr[0];
    short d[N];
    for (int k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min((int)d[k + 1], (int)d[k + 2]);
        a = std::min(a, (int)d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, (int)d[k + 4]);
        a = std::min(a, (int)d[k + 5]);
        a = std::min(a, (int)d[k + 6]);
        a = std::min(a, (int)d[k + 7]);
        a = std::min(a, (int)d[k + 8]);
        a0 = std::max(a0, std::min(a, (int)d[k]));
        a0 = std::max(a0, std::min(a, (int)d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max((int)d[k + 1], (int)d[k + 2]);
        b = std::max(b, (int)d[k + 3]);
        b = std::max(b, (int)d[k + 4]);
        b = std::max(b, (int)d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, (int)d[k + 6]);
        b = std::max(b, (int)d[k + 7]);
        b = std::max(b, (int)d[k + 8]);
        b0 = std::min(b0, std::max(b, (int)d[k]));
        b0 = std::min(b0, std::max(b, (int)d[k + 9]));
    }
    return -b0 - 1;
}

struct FastPt { int x, y, score; };

void fast9(const uint8_t *img, int cols, int rows, int stride, int threshold, std::vector<FastPt> &out) {
    out.clear();
    if (cols < 7 || rows < 7) return;  // nothing can be scanned (loops below are empty)
    const int K = 8, N = 25;
    int pixel[25];
    fast_offsets(pixel, stride);
    threshold = std::min(std::max(threshold, 0), 255);
    uint8_t tab[512];
    for (int i = -255; i <= 255; i++) tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);
    std::vector<uint8_t> buf((size_t)cols * 3, 0);
    std::vector<int> cpos((size_t)(cols + 1) * 3, 0);
    int ncs[3] = {0, 0, 0};
    for (int i = 3; i < rows - 2; i++) {
        const uint8_t *ptr = img + (size_t)i * stride + 3;
        uint8_t *curr = &buf[(size_t)((i - 3) % 3) * cols];
        int *cornerpos = &cpos[(size_t)((i - 3) % 3) * (cols + 1)];
        memset(curr, 0, cols);
        int ncorners = 0;
        if (i < rows - 3) {
            for (int j = 3; j < cols - 3; j++, ptr++) {
                int v = ptr[0];
                const uint8_t *t = &tab[0] - v + 255;
                int d = t[ptr[pixel[0]]] | t[ptr[pixel[8]]];
                if (d == 0) continue;
                d &= t[ptr[pixel[2]]] | t[ptr[pixel[10]]];
                d &= t[ptr[pixel[4]]] | t[ptr[pixel[12]]];
                d &= t[ptr[pixel[6]]] | t[ptr[pixel[14]]];
                if (d == 0) continue;
                d &= t[ptr[pixel[1]]] | t[ptr[pixel[9]]];
                d &= t[ptr[pixel[3]]] | t[ptr[pixel[11]]];
                d &= t[ptr[pixel[5]]] | t[ptr[pixel[13]]];
                d &= t[ptr[pixel[7]]] | t[ptr[pixel[15]]];
                if (d & 1) {
                    int vt = v - threshold, count = 0;
                    for (int k = 0; k < N; k++) {
                        int x = ptr[pixel[k]];
                        if (x < vt) {
                            if (++count > K) {
                                cornerpos[ncorners++] = j;
                                curr[j] = (uint8_t)fast_corner_score(ptr, pixel, threshold);
                                break;
                            }
                        } else
                            count = 0;
                    }
                }
                if (d & 2) {
                    int vt = v + threshold, count = 0;
                    for (int k = 0; k < N; k++) {
                        int x = ptr[pixel[k]];
                        if (x > vt) {
                            if (++count > K) {
                                cornerpos[ncorners++] = j;
                                curr[j] = (uint8_t)fast_corner_score(ptr, pixel, threshold);
                                break;
                            }
                        } else
                            count = 0;
                    }
                }
            }
        }
        ncs[(i - 3) % 3] = ncorners;
        if (i == 3) continue;
        const uint8_t *prev = &buf[(size_t)((i - 4 + 3) % 3) * cols];
        const uint8_t *pprev = &buf[(size_t)((i - 5 + 3) % 3) * cols];
        cornerpos = &cpos[(size_t)((i - 4 + 3) % 3) * (cols + 1)];
        ncorners = ncs[(i - 4 + 3) % 3];
        for (int k = 0; k < ncorners; k++) {
            int j = cornerpos[k];
            int score = prev[j];
            if (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] && score > pprev[j] &&
                score > pprev[j + 1] && score > curr[j - 1] && score > curr[j] && score > curr[j + 1])
                out.push_back({j, i - 1, score});
        }
    }
}

/* cell loop of ComputeKeyPointsOctTree, src/ORBextractor.cc:771-829; emits (x,y) relative to minBorder */
void fast_cells(const uint8_t *img, int cols, int rows, int stride, int iniTh, int minTh, std::vector<Kp> &out) {
    out.clear();
    const float W = 30;
    const int minBorderX = EDGE_THRESHOLD - 3;
    const int minBorderY = minBorderX;
    const int maxBorderX = cols - EDGE_THRESHOLD + 3;
    const int maxBorderY = rows - EDGE_THRESHOLD + 3;
    const float width = (float)(maxBorderX - minBorderX);
    const float height = (float)(maxBorderY - minBorderY);
    const int nCols = (int)(width / W);
    const int nRows = (int)(height / W);
    if (nCols <= 0 || nRows <= 0) return;  // the reference would divide by zero here; levels this small are rejected upstream
    const int wCell = (int)std::ceil(width / nCols);
    const int hCell = (int)std::ceil(height / nRows);
    std::vector<FastPt> cell;
    for (int i = 0; i < nRows; i++) {
        const float iniY = (float)(minBorderY + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBorderY - 3) continue;
        if (maxY > maxBorderY) maxY = (float)maxBorderY;
        for (int j = 0; j < nCols; j++) {
            const float iniX = (float)(minBorderX + j * wCell);
            float maxX = iniX + wCell + 6;
            if (iniX >= maxBorderX - 6) continue;
            if (maxX > maxBorderX) maxX = (float)maxBorderX;
            const int y0 = (int)iniY, y1 = (int)maxY, x0 = (int)iniX, x1 = (int)maxX;
            const uint8_t *win = img + (size_t)y0 * stride + x0;
            fast9(win, x1 - x0, y1 - y0, stride, iniTh, cell);
            if (cell.empty()) fast9(win, x1 - x0, y1 - y0, stride, minTh, cell);
            for (const FastPt &p : cell) {
                Kp k;
                k.x = (float)p.x + j * wCell;
                k.y = (float)p.y + i * hCell;
                k.size = 7.f;
                k.angle = -1.f;
                k.response = (float)p.score;
                k.octave = 0;
                out.push_back(k);
            }
        }
    }
}

/* ---------------------------------------------------------------- quadtree */
struct Node {  // ExtractorNode, include/ORBextractor.h:32-43
    std::vector<Kp> keys;
    int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
    std::list<Node>::iterator lit;
    bool noMore = false;
    long seq = 0;  // creation order: replaces the heap address in the size sort
};

/* ExtractorNode::DivideNode, src/ORBextractor.cc:481-537 */
void divide_node(const Node &p, Node &n1, Node &n2, Node &n3, Node &n4) {
    const int halfX = (int)std::ceil(static_cast<float>(p.URx - p.ULx) / 2);
    const int halfY = (int)std::ceil(static_cast<float>(p.BRy - p.ULy) / 2);
    n1.ULx = p.ULx; n1.ULy = p.ULy;
    n1.URx = p.ULx + halfX; n1.URy = p.ULy;
    n1.BLx = p.ULx; n1.BLy = p.ULy + halfY;
    n1.BRx = p.ULx + halfX; n1.BRy = p.ULy + halfY;
    n2.ULx = n1.URx; n2.ULy = n1.URy;
    n2.URx = p.URx; n2.URy = p.URy;
    n2.BLx = n1.BRx; n2.BLy = n1.BRy;
    n2.BRx = p.URx; n2.BRy = p.ULy + halfY;
    n3.ULx = n1.BLx; n3.ULy = n1.BLy;
    n3.URx = n1.BRx; n3.URy = n1.BRy;
    n3.BLx = p.BLx; n3.BLy = p.BLy;
    n3.BRx = n1.BRx; n3.BRy = p.BLy;
    n4.ULx = n3.URx; n4.ULy = n3.URy;
    n4.URx = n2.BRx; n4.URy = n2.BRy;
    n4.BLx = n3.BRx; n4.BLy = n3.BRy;
    n4.BRx = p.BRx; n4.BRy = p.BRy;
    for (const Kp &kp : p.keys) {
        if (kp.x < n1.URx) {
            if (kp.y < n1.BRy) n1.keys.push_back(kp);
            else n3.keys.push_back(kp);
        } else if (kp.y < n1.BRy)
            n2.keys.push_back(kp);
        else
            n4.keys.push_back(kp);
    }
    if (n1.keys.size() == 1) n1.noMore = true;
    if (n2.keys.size() == 1) n2.noMore = true;
    if (n3.keys.size() == 1) n3.noMore = true;
    if (n4.keys.size() == 1) n4.noMore = true;
}

typedef std::pair<int, std::pair<long, Node *>> SizeNode;  // (size, (creation seq, node))

/* DistributeOctTree, src/ORBextractor.cc:539-763 */
std::vector<Kp> distribute_octree(const std::vector<Kp> &in, int minX, int maxX, int minY, int maxY, int N) {
    const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
    std::vector<Kp> result;
    if (nIni <= 0) return result;  // reference: division by zero for portrait levels; rejected upstream
    const float hX = static_cast<float>(maxX - minX) / nIni;
    std::list<Node> nodes;
    std::vector<Node *> ini(nIni);
    long seq = 0;
    for (int i = 0; i < nIni; i++) {
        Node ni;
        ni.ULx = (int)(hX * static_cast<float>(i)); ni.ULy = 0;
        ni.URx = (int)(hX * static_cast<float>(i + 1)); ni.URy = 0;
        ni.BLx = ni.ULx; ni.BLy = maxY - minY;
        ni.BRx = ni.URx; ni.BRy = maxY - minY;
        ni.seq = seq++;
        nodes.push_back(ni);
        ini[i] = &nodes.back();
    }
    for (const Kp &kp : in) ini[(size_t)(kp.x / hX)]->keys.push_back(kp);
    for (auto lit = nodes.begin(); lit != nodes.end();) {
        if (lit->keys.size() == 1) { lit->noMore = true; ++lit; }
        else if (lit->keys.empty()) lit = nodes.erase(lit);
        else ++lit;
    }
    bool finish = false;
    std::vector<SizeNode> sizeAndNode;
    auto push_child = [&](Node &n, int *nToExpand) {
        if (n.keys.size() > 0) {
            n.seq = seq++;
            nodes.push_front(n);
            if (n.keys.size() > 1) {
                if (nToExpand) (*nToExpand)++;
                sizeAndNode.push_back(std::make_pair((int)n.keys.size(), std::make_pair(nodes.front().seq, &nodes.front())));
                nodes.front().lit = nodes.begin();
            }
        }
    };
    while (!finish) {
        int prevSize = (int)nodes.size();
        auto lit = nodes.begin();
        int nToExpand = 0;
        sizeAndNode.clear();
        while (lit != nodes.end()) {
            if (lit->noMore) { ++lit; continue; }
            Node n1, n2, n3, n4;
            divide_node(*lit, n1, n2, n3, n4);
            push_child(n1, &nToExpand);
            push_child(n2, &nToExpand);
            push_child(n3, &nToExpand);
            push_child(n4, &nToExpand);
            lit = nodes.erase(lit);
        }
        if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) {
            finish = true;
        } else if (((int)nodes.size() + nToExpand * 3) > N) {
            while (!finish) {
                prevSize = (int)nodes.size();
                std::vector<SizeNode> prev = sizeAndNode;
                sizeAndNode.clear();
                std::sort(prev.begin(), prev.end());  // (size, seq) ascending; walked from the back
                for (int j = (int)prev.size() - 1; j >= 0; j--) {
                    Node n1, n2, n3, n4;
                    Node *pn = prev[j].second.second;
                    divide_node(*pn, n1, n2, n3, n4);
                    push_child(n1, nullptr);
                    push_child(n2, nullptr);
                    push_child(n3, nullptr);
                    push_child(n4, nullptr);
                    nodes.erase(pn->lit);
                    if ((int)nodes.size() >= N) break;
                }
                if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) finish = true;
            }
        }
    }
    result.reserve(nodes.size());
    for (auto &n : nodes) {
        const Kp *best = &n.keys[0];
        float maxResponse = best->response;
        for (size_t k = 1; k < n.keys.size(); k++)
            if (n.keys[k].response > maxResponse) { best = &n.keys[k]; maxResponse = n.keys[k].response; }
        result.push_back(*best);
    }
    return result;
}

/* ------------------------------------------------------------- orientation */
/* cv::fastAtan2 scalar path, SURVEY 8a-5 / Appendix A.5 */
float fast_atan2(float y, float x) {
    static const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
    static const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
    static const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
    static const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
    float ax = std::abs(x), ay = std::abs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* IC_Angle, src/ORBextractor.cc:77-104 (centre already rounded by the caller) */
float ic_angle(const uint8_t *img, int stride, int cx, int cy, const int *umax) {
    int m_01 = 0, m_10 = 0;
    const uint8_t *center = img + (size_t)cy * stride + cx;
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0;
        int d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * stride], val_minus = center[u - v * stride];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return fast_atan2((float)m_01, (float)m_10);
}

/* -------------------------------------------------------------- descriptor */
const float factorPI = (float)(M_PI / 180.f);  // src/ORBextractor.cc:107

/* computeOrbDescriptor, src/ORBextractor.cc:108-147 */
void orb_descriptor(const uint8_t *img, int step, int cx, int cy, float kp_angle, uint8_t *desc) {
    float angle = (float)kp_angle * factorPI;
    float a = (float)std::cos(angle), b = (float)std::sin(angle);  // float overloads: see header note (2)
    const uint8_t *center = img + (size_t)cy * step + cx;
    const int8_t *pat = kPattern;
    auto get = [&](int idx) -> int {
        int px = pat[2 * idx], py = pat[2 * idx + 1];
        float fr = px * b + py * a;
        float fc = px * a - py * b;
        return center[cv_round(fr) * step + cv_round(fc)];
    };
    for (int i = 0; i < 32; ++i, pat += 32) {
        int val = 0;
        for (int j = 0; j < 8; j++) {
            int t0 = get(2 * j), t1 = get(2 * j + 1);
            val |= (t0 < t1) << j;
        }
        desc[i] = (uint8_t)val;
    }
}

/* --------------------------------------------------------------- Hamming */
/* ORBmatcher::DescriptorDistance, src/ORBmatcher.cc:1647-1663 */
int hamming(const uint8_t *a, const uint8_t *b) {
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t x, y;
        memcpy(&x, a + 4 * i, 4);
        memcpy(&y, b + 4 * i, 4);
        unsigned int v = x ^ y;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}
inline int hamming_popcnt(const uint8_t *a, const uint8_t *b) {
    int dist = 0;
    for (int i = 0; i < 4; i++) {
        uint64_t x, y;
        memcpy(&x, a + 8 * i, 8);
        memcpy(&y, b + 8 * i, 8);
        dist += __builtin_popcountll(x ^ y);
    }
    return dist;
}

}  // namespace

/* ========================================================================= */
struct orc_extractor {
    int nfeatures, nlevels, iniThFAST, minThFAST;
    double scaleFactor;  // include/ORBextractor.h:98 (double member initialised from the float argument)
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    std::vector<int> mnFeaturesPerLevel, umax;
    // per-call state
    std::vector<Img> pyramid, blurred;
    std::vector<std::vector<Kp>> candidates, selected;
};

extern "C" {

int orc_cv_round(double v) { return cv_round(v); }

/* ORBextractor::ORBextractor, src/ORBextractor.cc:410-470 */
orc_extractor *orc_create(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST) {
    orc_extractor *e = new orc_extractor;
    e->nfeatures = _nfeatures;
    e->scaleFactor = _scaleFactor;
    e->nlevels = _nlevels;
    e->iniThFAST = _iniThFAST;
    e->minThFAST = _minThFAST;
    const int nlevels = _nlevels;
    e->mvScaleFactor.resize(nlevels);
    e->mvLevelSigma2.resize(nlevels);
    e->mvScaleFactor[0] = 1.0f;
    e->mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
        e->mvScaleFactor[i] = (float)(e->mvScaleFactor[i - 1] * e->scaleFactor);
        e->mvLevelSigma2[i] = e->mvScaleFactor[i] * e->mvScaleFactor[i];
    }
    e->mvInvScaleFactor.resize(nlevels);
    e->mvInvLevelSigma2.resize(nlevels);
    for (int i = 0; i < nlevels; i++) {
        e->mvInvScaleFactor[i] = 1.0f / e->mvScaleFactor[i];
        e->mvInvLevelSigma2[i] = 1.0f / e->mvLevelSigma2[i];
    }
    e->mnFeaturesPerLevel.resize(nlevels);
    float factor = (float)(1.0f / e->scaleFactor);
    float nDesiredFeaturesPerScale =
        e->nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sumFeatures = 0;
    for (int level = 0; level < nlevels - 1; level++) {
        e->mnFeaturesPerLevel[level] = cv_round(nDesiredFeaturesPerScale);
        sumFeatures += e->mnFeaturesPerLevel[level];
        nDesiredFeaturesPerScale *= factor;
    }
    e->mnFeaturesPerLevel[nlevels - 1] = std::max(e->nfeatures - sumFeatures, 0);
    e->umax.resize(HALF_PATCH_SIZE + 1);
    int v, v0, vmax = cv_floor(HALF_PATCH_SIZE * sqrt(2.f) / 2 + 1);
    int vmin = cv_ceil(HALF_PATCH_SIZE * sqrt(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) e->umax[v] = cv_round(sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (e->umax[v0] == e->umax[v0 + 1]) ++v0;
        e->umax[v] = v0;
        ++v0;
    }
    return e;
}

void orc_destroy(orc_extractor *e) { delete e; }
int orc_nlevels(const orc_extractor *e) { return e->nlevels; }
void orc_scale_tables(const orc_extractor *e, float *s, float *is, float *s2, float *is2) {
    for (int i = 0; i < e->nlevels; i++) {
        if (s) s[i] = e->mvScaleFactor[i];
        if (is) is[i] = e->mvInvScaleFactor[i];
        if (s2) s2[i] = e->mvLevelSigma2[i];
        if (is2) is2[i] = e->mvInvLevelSigma2[i];
    }
}
void orc_features_per_level(const orc_extractor *e, int *out) {
    for (int i = 0; i < e->nlevels; i++) out[i] = e->mnFeaturesPerLevel[i];
}
void orc_umax(const orc_extractor *e, int *out) {
    for (int i = 0; i < 16; i++) out[i] = e->umax[i];
}

/* operator(), src/ORBextractor.cc:1043-1105 with ComputePyramid :1107-1132 and
 * ComputeKeyPointsOctTree :765-853 inlined as stages. */
int orc_extract(orc_extractor *e, const uint8_t *img, int w, int h, int stride, orc_keypoint *kps, int cap,
                uint8_t *desc32, int *n_out) {
    const int nlevels = e->nlevels;
    if (n_out) *n_out = 0;
    if (!img || w <= 0 || h <= 0) return 0;  // _image.empty(): outputs untouched
    /* ComputePyramid.  The 19-px reflect-101 frame (copyMakeBorder :1122-1128) is never read by any
     * stage below, so only the ROI is materialised. */
    e->pyramid.assign(nlevels, Img());
    e->blurred.assign(nlevels, Img());
    for (int level = 0; level < nlevels; ++level) {
        float scale = e->mvInvScaleFactor[level];
        int sw = cv_round((float)w * scale), sh = cv_round((float)h * scale);
        Img &L = e->pyramid[level];
        L.alloc(sw, sh);
        if (level != 0) {
            const Img &P = e->pyramid[level - 1];
            resize_linear_u8(P.d.data(), P.w, P.h, P.w, L.d.data(), sw, sh, sw);
        } else {
            for (int y = 0; y < h; y++) memcpy(L.row(y), img + (size_t)y * stride, w);
        }
    }
    /* ComputeKeyPointsOctTree */
    e->candidates.assign(nlevels, std::vector<Kp>());
    e->selected.assign(nlevels, std::vector<Kp>());
    for (int level = 0; level < nlevels; ++level) {
        const Img &L = e->pyramid[level];
        const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
        const int maxBorderX = L.w - EDGE_THRESHOLD + 3, maxBorderY = L.h - EDGE_THRESHOLD + 3;
        fast_cells(L.d.data(), L.w, L.h, L.w, e->iniThFAST, e->minThFAST, e->candidates[level]);
        std::vector<Kp> &keypoints = e->selected[level];
        keypoints = distribute_octree(e->candidates[level], minBorderX, maxBorderX, minBorderY, maxBorderY,
                                      e->mnFeaturesPerLevel[level]);
        const int scaledPatchSize = (int)(PATCH_SIZE * e->mvScaleFactor[level]);
        for (Kp &k : keypoints) {
            k.x += minBorderX;
            k.y += minBorderY;
            k.octave = level;
            k.size = (float)scaledPatchSize;
        }
    }
    for (int level = 0; level < nlevels; ++level) {
        const Img &L = e->pyramid[level];
        for (Kp &k : e->selected[level])
            k.angle = ic_angle(L.d.data(), L.w, cv_round(k.x), cv_round(k.y), e->umax.data());
    }
    int nkeypoints = 0;
    for (int level = 0; level < nlevels; ++level) nkeypoints += (int)e->selected[level].size();
    if (n_out) *n_out = nkeypoints;
    if (nkeypoints > cap) return -1;
    int offset = 0;
    for (int level = 0; level < nlevels; ++level) {
        std::vector<Kp> &keypoints = e->selected[level];
        int n = (int)keypoints.size();
        if (n == 0) continue;
        const Img &L = e->pyramid[level];
        Img &B = e->blurred[level];
        B.alloc(L.w, L.h);
        gauss7_u8(L.d.data(), L.w, L.h, L.w, B.d.data(), L.w);
        for (int i = 0; i < n; i++)
            orb_descriptor(B.d.data(), B.w, cv_round(keypoints[i].x), cv_round(keypoints[i].y), keypoints[i].angle,
                           desc32 + (size_t)(offset + i) * 32);
        float scale = e->mvScaleFactor[level];
        for (int i = 0; i < n; i++) {
            orc_keypoint &o = kps[offset + i];
            const Kp &k = keypoints[i];
            o.x = k.x; o.y = k.y;
            if (level != 0) { o.x *= scale; o.y *= scale; }
            o.size = k.size; o.angle = k.angle; o.response = k.response;
            o.octave = k.octave; o.class_id = -1;
        }
        offset += n;
    }
    return 0;
}

int orc_level_size(const orc_extractor *e, int level, int *w, int *h) {
    if (level < 0 || level >= (int)e->pyramid.size()) return -1;
    *w = e->pyramid[level].w;
    *h = e->pyramid[level].h;
    return 0;
}
const uint8_t *orc_level_ptr(const orc_extractor *e, int level) { return e->pyramid[level].d.data(); }
const uint8_t *orc_blur_ptr(const orc_extractor *e, int level) {
    return e->blurred[level].d.empty() ? nullptr : e->blurred[level].d.data();
}
int orc_level_candidates(const orc_extractor *e, int level, float *xyr, int cap) {
    const std::vector<Kp> &c = e->candidates[level];
    for (int i = 0; i < (int)c.size() && i < cap; i++) {
        xyr[3 * i] = c[i].x; xyr[3 * i + 1] = c[i].y; xyr[3 * i + 2] = c[i].response;
    }
    return (int)c.size();
}
int orc_level_selected(const orc_extractor *e, int level, float *xyra, int cap) {
    const std::vector<Kp> &c = e->selected[level];
    for (int i = 0; i < (int)c.size() && i < cap; i++) {
        xyra[4 * i] = c[i].x; xyra[4 * i + 1] = c[i].y; xyra[4 * i + 2] = c[i].response; xyra[4 * i + 3] = c[i].angle;
    }
    return (int)c.size();
}

void orc_resize_linear_u8(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst, int dw, int dh, int dstride) {
    resize_linear_u8(src, sw, sh, sstride, dst, dw, dh, dstride);
}
void orc_resize_tables(int ssize, int dsize, int *ofs, int16_t *coef) {
    std::vector<int> o;
    std::vector<short> c;
    resize_tables(ssize, dsize, o, c, true);
    memcpy(ofs, o.data(), sizeof(int) * dsize);
    memcpy(coef, c.data(), sizeof(short) * 2 * dsize);
}
void orc_gauss7_u8(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride) {
    gauss7_u8(src, w, h, sstride, dst, dstride);
}
void orc_gauss7_kernel(int *k7) { gauss7_kernel(k7); }
int orc_fast9(const uint8_t *win, int w, int h, int stride, int threshold, float *xyr, int cap) {
    std::vector<FastPt> pts;
    fast9(win, w, h, stride, threshold, pts);
    for (int i = 0; i < (int)pts.size() && i < cap; i++) {
        xyr[3 * i] = (float)pts[i].x; xyr[3 * i + 1] = (float)pts[i].y; xyr[3 * i + 2] = (float)pts[i].score;
    }
    return (int)pts.size();
}
int orc_fast_cells(const uint8_t *img, int w, int h, int stride, int ini_th, int min_th, float *xyr, int cap) {
    std::vector<Kp> out;
    fast_cells(img, w, h, stride, ini_th, min_th, out);
    for (int i = 0; i < (int)out.size() && i < cap; i++) {
        xyr[3 * i] = out[i].x; xyr[3 * i + 1] = out[i].y; xyr[3 * i + 2] = out[i].response;
    }
    return (int)out.size();
}
float orc_fast_atan2(float y, float x) { return fast_atan2(y, x); }
float orc_ic_angle(const uint8_t *img, int stride, int cx, int cy, const int *umax16) {
    return ic_angle(img, stride, cx, cy, umax16);
}
void orc_descriptor(const uint8_t *blurred, int stride, int cx, int cy, float angle_deg, uint8_t *desc32) {
    orb_descriptor(blurred, stride, cx, cy, angle_deg, desc32);
}
int orc_distribute_octree(const float *xyr, int n, int minX, int maxX, int minY, int maxY, int N, float *out_xyr, int cap) {
    std::vector<Kp> in(n);
    for (int i = 0; i < n; i++) {
        in[i].x = xyr[3 * i]; in[i].y = xyr[3 * i + 1]; in[i].response = xyr[3 * i + 2];
        in[i].size = 7.f; in[i].angle = -1.f; in[i].octave = 0;
    }
    std::vector<Kp> out = distribute_octree(in, minX, maxX, minY, maxY, N);
    for (int i = 0; i < (int)out.size() && i < cap; i++) {
        out_xyr[3 * i] = out[i].x; out_xyr[3 * i + 1] = out[i].y; out_xyr[3 * i + 2] = out[i].response;
    }
    return (int)out.size();
}

int orc_hamming(const uint8_t *a32, const uint8_t *b32) { return hamming(a32, b32); }

/* best/second update rule of src/ORBmatcher.cc:216-225 (SURVEY Appendix B "Common") */
void orc_knn2(const uint8_t *q, int nq, const uint8_t *r, int nr, int32_t *best_dist, int32_t *best_idx, int32_t *second_dist) {
    for (int i = 0; i < nq; i++) {
        int best1 = 256, best2 = 256, idx = -1;
        for (int j = 0; j < nr; j++) {
            int d = hamming(q + (size_t)i * 32, r + (size_t)j * 32);
            if (d < best1) { best2 = best1; best1 = d; idx = j; }
            else if (d < best2) best2 = d;
        }
        best_dist[i] = best1; best_idx[i] = idx; second_dist[i] = best2;
    }
}
void orc_match_csr(const uint8_t *q, int nq, const uint8_t *r, const int32_t *cand_off, const int32_t *cand_idx,
                   int32_t *best_dist, int32_t *best_idx, int32_t *second_dist) {
    for (int i = 0; i < nq; i++) {
        int best1 = 256, best2 = 256, idx = -1;
        for (int c = cand_off[i]; c < cand_off[i + 1]; c++) {
            int j = cand_idx[c];
            int d = hamming(q + (size_t)i * 32, r + (size_t)j * 32);
            if (d < best1) { best2 = best1; best1 = d; idx = j; }
            else if (d < best2) best2 = d;
        }
        best_dist[i] = best1; best_idx[i] = idx; second_dist[i] = best2;
    }
}

/* Frame::ComputeStereoMatches row table + Hamming scan, src/Frame.cc:476-549 */
static void stereo_rowband(const orc_keypoint *kl, const uint8_t *dl, int nl, const orc_keypoint *kr, const uint8_t *dr, int nr,
                           const float *scale_factors, int nrows, float minD, float maxD, int32_t *best_dist, int32_t *best_idx) {
    std::vector<std::vector<size_t>> vRowIndices(nrows);
    for (int iR = 0; iR < nr; iR++) {
        const float kpY = kr[iR].y;
        const float r = 2.0f * scale_factors[kr[iR].octave];
        const int maxr = (int)std::ceil(kpY + r);
        const int minr = (int)std::floor(kpY - r);
        for (int yi = minr; yi <= maxr; yi++)
            if (yi >= 0 && yi < nrows) vRowIndices[yi].push_back(iR);  // the reference indexes unchecked; rows outside do not exist
    }
    for (int iL = 0; iL < nl; iL++) {
        best_dist[iL] = 100;  // ORBmatcher::TH_HIGH
        best_idx[iL] = 0;
        const int levelL = kl[iL].octave;
        const float vL = kl[iL].y, uL = kl[iL].x;
        const std::vector<size_t> &cands = vRowIndices[(size_t)vL];
        if (cands.empty()) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = 100;
        size_t bestIdxR = 0;
        for (size_t iC = 0; iC < cands.size(); iC++) {
            const size_t iR = cands[iC];
            if (kr[iR].octave < levelL - 1 || kr[iR].octave > levelL + 1) continue;
            const float uR = kr[iR].x;
            if (uR >= minU && uR <= maxU) {
                const int dist = hamming(dl + (size_t)iL * 32, dr + iR * 32);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        best_dist[iL] = bestDist;
        best_idx[iL] = (int32_t)bestIdxR;
    }
}

void orc_stereo_rowband(const orc_keypoint *kl, const uint8_t *dl, int nl, const orc_keypoint *kr, const uint8_t *dr, int nr,
                        const float *scale_factors, int nrows, float min_d, float max_d, int32_t *best_dist, int32_t *best_idx) {
    stereo_rowband(kl, dl, nl, kr, dr, nr, scale_factors, nrows, min_d, max_d, best_dist, best_idx);
}

/* Frame::ComputeStereoMatches, src/Frame.cc:466-640 (pyramids taken from the two extractors' last call) */
void orc_stereo_full(const orc_extractor *el, const orc_extractor *er, const orc_keypoint *kl, const uint8_t *dl, int nl,
                     const orc_keypoint *kr, const uint8_t *dr, int nr, float mbf, float mb, float *mvuRight, float *mvDepth) {
    for (int i = 0; i < nl; i++) { mvuRight[i] = -1.0f; mvDepth[i] = -1.0f; }
    const int thOrbDist = (100 + 50) / 2;
    const int nRows = el->pyramid[0].h;
    const float minZ = mb, minD = 0, maxD = mbf / minZ;
    std::vector<int32_t> bd(nl), bi(nl);
    stereo_rowband(kl, dl, nl, kr, dr, nr, el->mvScaleFactor.data(), nRows, minD, maxD, bd.data(), bi.data());
    std::vector<std::pair<int, int>> vDistIdx;
    for (int iL = 0; iL < nl; iL++) {
        if (!(bd[iL] < thOrbDist)) continue;
        const int oct = kl[iL].octave;
        const float uL = kl[iL].x;
        const float uR0 = kr[bi[iL]].x;
        const float scaleFactor = el->mvInvScaleFactor[oct];
        const float scaleduL = std::round(kl[iL].x * scaleFactor);
        const float scaledvL = std::round(kl[iL].y * scaleFactor);
        const float scaleduR0 = std::round(uR0 * scaleFactor);
        const int w = 5, L = 5;
        const Img &PL = el->pyramid[oct], &PR = er->pyramid[oct];
        int bestDist = INT_MAX, bestincR = 0;
        float vDists[2 * L + 1];
        const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
        if (iniu < 0 || endu >= PR.w) continue;
        const int cy = (int)scaledvL, cxl = (int)scaleduL;
        for (int incR = -L; incR <= +L; incR++) {
            const int cxr = (int)(scaleduR0 + incR);
            float dist = 0;  // cv::norm(IL, IR, NORM_L1) of centre-subtracted float patches
            double acc = 0;
            for (int dy = -w; dy <= w; dy++)
                for (int dx = -w; dx <= w; dx++) {
                    float a = (float)PL.row(cy + dy)[cxl + dx] - (float)PL.row(cy)[cxl];
                    float b = (float)PR.row(cy + dy)[cxr + dx] - (float)PR.row(cy)[cxr];
                    acc += std::fabs(a - b);  // all terms are small integers: exact in any accumulation order
                }
            dist = (float)acc;
            if (dist < bestDist) { bestDist = (int)dist; bestincR = incR; }
            vDists[L + incR] = dist;
        }
        if (bestincR == -L || bestincR == L) continue;
        const float dist1 = vDists[L + bestincR - 1], dist2 = vDists[L + bestincR], dist3 = vDists[L + bestincR + 1];
        const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
        if (deltaR < -1 || deltaR > 1) continue;
        float bestuR = el->mvScaleFactor[oct] * ((float)scaleduR0 + (float)bestincR + deltaR);
        float disparity = (uL - bestuR);
        if (disparity >= minD && disparity < maxD) {
            if (disparity <= 0) { disparity = (float)0.01; bestuR = (float)((double)uL - 0.01); }   // double literals, :616-617
            mvDepth[iL] = mbf / disparity;
            mvuRight[iL] = bestuR;
            vDistIdx.push_back(std::pair<int, int>(bestDist, iL));
        }
    }
    if (vDistIdx.empty()) return;  // the reference reads vDistIdx[0] of an empty vector here
    std::sort(vDistIdx.begin(), vDistIdx.end());
    const float median = (float)vDistIdx[vDistIdx.size() / 2].first;
    const float thDist = 1.5f * 1.4f * median;
    for (int i = (int)vDistIdx.size() - 1; i >= 0; i--) {
        if (vDistIdx[i].first < thDist) break;
        mvuRight[vDistIdx[i].second] = -1;
        mvDepth[vDistIdx[i].second] = -1;
    }
}

/* ------------------------------------------------------------ DBoW2 transform */
/* FORB::distance, Thirdparty/DBoW2/DBoW2/FORB.cpp:81-101 (the same SWAR popcount as ORBmatcher::DescriptorDistance) */
static void bow_descend_one(const int32_t *child_off, const int32_t *child_idx, const uint8_t *node_desc, int L, const uint8_t *feature,
                            int levelsup, int32_t *leaf, int32_t *nid) {
    /* TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup), TemplatedVocabulary.h:1218-1259 */
    const int nid_level = L - levelsup;
    *nid = 0;   /* :1227 sets it only when nid_level <= 0; otherwise :1250 does, if the descent reaches that level */
    int32_t final_id = 0;
    int current_level = 0;
    do {
        ++current_level;
        const int32_t c0 = child_off[final_id], c1 = child_off[final_id + 1];
        if (c0 == c1) break;   /* a root without children: the reference would index an empty vector */
        int32_t best = child_idx[c0];
        double best_d = hamming(feature, node_desc + (size_t)best * 32);
        for (int32_t c = c0 + 1; c < c1; c++) {
            const int32_t id = child_idx[c];
            const double d = hamming(feature, node_desc + (size_t)id * 32);
            if (d < best_d) { best_d = d; best = id; }
        }
        final_id = best;
        if (current_level == nid_level) *nid = final_id;
    } while (child_off[final_id] != child_off[final_id + 1]);
    *leaf = final_id;
}

void orc_bow_descend(const int32_t *child_off, const int32_t *child_idx, const uint8_t *node_desc, int L, const uint8_t *desc, int n,
                     int levelsup, int32_t *leaf, int32_t *nid) {
    for (int i = 0; i < n; i++) bow_descend_one(child_off, child_idx, node_desc, L, desc + (size_t)i * 32, levelsup, &leaf[i], &nid[i]);
}

void orc_bow_transform(const int32_t *child_off, const int32_t *child_idx, const uint8_t *node_desc, const int32_t *word_id,
                       const double *weight, int L, int weighting, int norm, const uint8_t *desc, int n, int levelsup,
                       int32_t *bow_ids, double *bow_vals, int32_t *n_bow, int32_t *fv_node, int32_t *fv_feat, int32_t *n_fv) {
    /* TemplatedVocabulary::transform(features, v, fv, levelsup), TemplatedVocabulary.h:1127-1200 */
    std::map<unsigned, double> v;                       /* BowVector */
    std::map<unsigned, std::vector<unsigned>> fv;       /* FeatureVector */
    const bool tf = weighting == 0 || weighting == 1;   /* TF_IDF or TF */
    for (int i = 0; i < n; i++) {
        int32_t leaf, nid;
        bow_descend_one(child_off, child_idx, node_desc, L, desc + (size_t)i * 32, levelsup, &leaf, &nid);
        const unsigned id = (unsigned)word_id[leaf];
        const double w = weight[leaf];
        if (w > 0) {   /* not stopped */
            if (tf) v[id] += w;                          /* BowVector::addWeight, BowVector.cpp:34-46 */
            else v.insert(std::make_pair(id, w));        /* addIfNotExist, :50-58 */
            fv[(unsigned)nid].push_back((unsigned)i);    /* FeatureVector::addFeature, FeatureVector.cpp:29-43 */
        }
    }
    if (tf && !v.empty() && norm == 0) {                 /* :1160-1166 */
        const double nd = (double)v.size();
        for (auto &kv : v) kv.second /= nd;
    }
    if (norm != 0) {                                     /* BowVector::normalize, BowVector.cpp:62-84 */
        double s = 0.0;
        if (norm == 1) { for (auto &kv : v) s += std::fabs(kv.second); }
        else { for (auto &kv : v) s += kv.second * kv.second; s = std::sqrt(s); }
        if (s > 0.0) for (auto &kv : v) kv.second /= s;
    }
    int k = 0;
    for (auto &kv : v) { bow_ids[k] = (int32_t)kv.first; bow_vals[k] = kv.second; k++; }
    *n_bow = k;
    k = 0;
    for (auto &kv : fv) for (unsigned f : kv.second) { fv_node[k] = (int32_t)kv.first; fv_feat[k] = (int32_t)f; k++; }
    *n_fv = k;
}

/* ------------------------------------------------------------ MapPoint::ComputeDistinctiveDescriptors */
void orc_distinctive(const uint8_t *desc, const int32_t *off, int n_points, int32_t *best) {
    for (int p = 0; p < n_points; p++) {
        const int N = off[p + 1] - off[p];
        if (N <= 0) { best[p] = -1; continue; }   /* src/MapPoint.cc:257-258, 270-271 */
        const uint8_t *d = desc + (size_t)off[p] * 32;
        std::vector<float> D((size_t)N * N);       /* float Distances[N][N], :276 */
        for (int i = 0; i < N; i++) {
            D[(size_t)i * N + i] = 0;
            for (int j = i + 1; j < N; j++) {
                const int dij = hamming(d + (size_t)i * 32, d + (size_t)j * 32);
                D[(size_t)i * N + j] = (float)dij;
                D[(size_t)j * N + i] = (float)dij;
            }
        }
        int BestMedian = INT_MAX, BestIdx = 0;     /* :289-302 */
        for (int i = 0; i < N; i++) {
            std::vector<int> vDists(D.begin() + (size_t)i * N, D.begin() + (size_t)(i + 1) * N);
            std::sort(vDists.begin(), vDists.end());
            const int median = vDists[(size_t)(0.5 * (N - 1))];
            if (median < BestMedian) { BestMedian = median; BestIdx = i; }
        }
        best[p] = BestIdx;
    }
}

/* ------------------------------------------------------------ Frame grid + windowed search */
struct orc_grid {
    std::vector<float> kx, ky;
    std::vector<int32_t> koct;
    float minX, minY, wInv, hInv;
    float qMinX, qMinY;                  /* origin GetFeaturesInArea subtracts: the Frame's bounds, or a KeyFrame's int copies */
    std::vector<int32_t> cell[64][48];   /* mGrid[FRAME_GRID_COLS][FRAME_GRID_ROWS] */
};

orc_grid *orc_grid_create(const float *kx, const float *ky, const int32_t *koct, int n, float min_x, float min_y,
                          float width_inv, float height_inv) {
    orc_grid *g = new orc_grid;
    g->kx.assign(kx, kx + n); g->ky.assign(ky, ky + n); g->koct.assign(koct, koct + n);
    g->minX = min_x; g->minY = min_y; g->wInv = width_inv; g->hInv = height_inv;
    g->qMinX = min_x; g->qMinY = min_y;
    for (int i = 0; i < n; i++) {   /* Frame::AssignFeaturesToGrid :230-246, PosInGrid :382-392 */
        const int posX = (int)std::round((kx[i] - min_x) * width_inv);
        const int posY = (int)std::round((ky[i] - min_y) * height_inv);
        if (posX < 0 || posX >= 64 || posY < 0 || posY >= 48) continue;
        g->cell[posX][posY].push_back(i);
    }
    return g;
}

void orc_grid_destroy(orc_grid *g) { delete g; }

/* A KeyFrame walks the cells its Frame assigned (src/KeyFrame.cc:31-57 copies mGrid) but subtracts its own mnMinX / mnMinY,
 * which are ints (include/KeyFrame.h:185-186, src/KeyFrame.cc:574-586): the window origin can differ from the cell origin. */
void orc_grid_set_query_origin(orc_grid *g, float min_x, float min_y) { g->qMinX = min_x; g->qMinY = min_y; }

int orc_features_in_area(const orc_grid *g, float x, float y, float r, int minLevel, int maxLevel, int32_t *out) {
    int n = 0;   /* Frame::GetFeaturesInArea :327-380 */
    const int nMinCellX = std::max(0, (int)std::floor((x - g->qMinX - r) * g->wInv));
    if (nMinCellX >= 64) return 0;
    const int nMaxCellX = std::min(64 - 1, (int)std::ceil((x - g->qMinX + r) * g->wInv));
    if (nMaxCellX < 0) return 0;
    const int nMinCellY = std::max(0, (int)std::floor((y - g->qMinY - r) * g->hInv));
    if (nMinCellY >= 48) return 0;
    const int nMaxCellY = std::min(48 - 1, (int)std::ceil((y - g->qMinY + r) * g->hInv));
    if (nMaxCellY < 0) return 0;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            const std::vector<int32_t> &vCell = g->cell[ix][iy];
            for (size_t j = 0; j < vCell.size(); j++) {
                const int idx = vCell[j];
                if (bCheckLevels) {
                    if (g->koct[idx] < minLevel) continue;
                    if (maxLevel >= 0 && g->koct[idx] > maxLevel) continue;
                }
                const float distx = g->kx[idx] - x, disty = g->ky[idx] - y;
                if (std::fabs(distx) < r && std::fabs(disty) < r) out[n++] = idx;
            }
        }
    return n;
}

void orc_search_area(const orc_grid *g, const uint8_t *kdesc, const float *uright, const uint8_t *blocked, float x, float y,
                     float r, int minLevel, int maxLevel, float projXR, float erMax, const uint8_t *qdesc, int32_t res[7]) {
    std::vector<int32_t> vIndices(g->kx.size() + 1);
    const int n = orc_features_in_area(g, x, y, r, minLevel, maxLevel, vIndices.data());
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;   /* src/ORBmatcher.cc:71-75 */
    int bestIdx2 = -1;   /* bookkeeping the reference does not keep: which keypoint holds bestDist2 / bestLevel2 */
    for (int k = 0; k < n; k++) {
        const int idx = vIndices[k];
        if (blocked && blocked[idx]) continue;                                    /* :83-85 */
        if (erMax >= 0 && uright && uright[idx] > 0) {                            /* :87-92 */
            const float er = std::fabs(projXR - uright[idx]);
            if (er > erMax) continue;
        }
        const int dist = hamming(qdesc, kdesc + (size_t)idx * 32);
        if (dist < bestDist) {
            bestDist2 = bestDist; bestDist = dist;
            bestLevel2 = bestLevel; bestLevel = g->koct[idx];
            bestIdx2 = bestIdx; bestIdx = idx;
        } else if (dist < bestDist2) {
            bestLevel2 = g->koct[idx];
            bestDist2 = dist;
            bestIdx2 = idx;
        }
    }
    res[0] = bestDist; res[1] = bestIdx; res[2] = bestLevel; res[3] = bestDist2; res[4] = bestLevel2; res[5] = n;
    res[6] = bestIdx2;
}

/* The candidate loops of the remaining guided searches, on the same grid (KeyFrame::GetFeaturesInArea, src/KeyFrame.cc:569-608,
 * is Frame::GetFeaturesInArea without a level test).  best only: these loops keep no second place. */
static int window_best(const orc_grid *g, const uint8_t *kdesc, const uint8_t *blocked, float u, float v, float radius,
                       int levelLo, int levelHi, bool levelsInWindow, const uint8_t *dMP, int *bestDistOut) {
    std::vector<int32_t> vIndices(g->kx.size() + 1);
    /* Frame version filters the levels inside GetFeaturesInArea (:1528); the KeyFrame loops filter after it (:380, :911, ...) */
    const int n = levelsInWindow ? orc_features_in_area(g, u, v, radius, levelLo, levelHi, vIndices.data())
                                 : orc_features_in_area(g, u, v, radius, -1, -1, vIndices.data());
    int bestDist = 256, bestIdx = -1;
    for (int k = 0; k < n; k++) {
        const int idx = vIndices[k];
        if (blocked && blocked[idx]) continue;
        if (!levelsInWindow) {
            const int kpLevel = g->koct[idx];
            if (kpLevel < levelLo || kpLevel > levelHi) continue;
        }
        const int dist = hamming(dMP, kdesc + (size_t)idx * 32);
        if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
    }
    *bestDistOut = bestDist;
    return bestIdx;
}

/* ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) from the window search on (src/ORBmatcher.cc:359-399):
 * query i is a MapPoint that passed the projection tests (:317-355) with its u, v, radius = th * scale[nPredictedLevel];
 * matched[idx] != 0 where vpMatched[idx] is set (updated here as :396 does); match_idx[i] = keypoint or -1. */
int orc_search_by_projection_sim3(const orc_grid *g, const uint8_t *kdesc, int nq, const float *u, const float *v,
                                  const float *radius, const int32_t *level, const uint8_t *qdesc, uint8_t *matched,
                                  int32_t *match_idx) {
    int nmatches = 0;
    for (int i = 0; i < nq; i++) {
        match_idx[i] = -1;
        int bestDist;
        const int bestIdx = window_best(g, kdesc, matched, u[i], v[i], radius[i], level[i] - 1, level[i], false,
                                        qdesc + (size_t)i * 32, &bestDist);
        if (bestDist <= 50) {            /* TH_LOW, :394 */
            matched[bestIdx] = 1;        /* :396 */
            match_idx[i] = bestIdx;
            nmatches++;
        }
    }
    return nmatches;
}

/* ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame*, sAlreadyFound, th, ORBdist) from the window search on
 * (:1525-1598): occupied[i2] != 0 where CurrentFrame.mvpMapPoints[i2] is set; assigned[i2] = query whose MapPoint the keypoint
 * ends up with, or -1; qangle = pKF->mvKeysUn[i].angle, kangle = CurrentFrame.mvKeysUn[*].angle. */
int orc_search_by_projection_reloc(const orc_grid *g, const uint8_t *kdesc, const float *kangle, int nq, const float *u,
                                   const float *v, const float *radius, const int32_t *level, const float *qangle,
                                   const uint8_t *qdesc, int orb_dist, int check_orientation, uint8_t *occupied,
                                   int32_t *assigned) {
    const int HISTO_LENGTH = 30;
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<std::vector<int>> rotHist(HISTO_LENGTH);
    const int n = (int)g->kx.size();
    for (int i = 0; i < n; i++) assigned[i] = -1;
    int nmatches = 0;
    for (int i = 0; i < nq; i++) {
        int bestDist;
        const int bestIdx2 = window_best(g, kdesc, occupied, u[i], v[i], radius[i], level[i] - 1, level[i] + 1, true,
                                         qdesc + (size_t)i * 32, &bestDist);
        if (bestIdx2 >= 0 && bestDist <= orb_dist) {   /* :1555 (bestIdx2 < 0 only with bestDist = 256 > any ORBdist in use) */
            occupied[bestIdx2] = 1;                    /* :1557 */
            assigned[bestIdx2] = i;
            nmatches++;
            if (check_orientation) {
                float rot = qangle[i] - kangle[bestIdx2];
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (check_orientation) {
        int32_t sizes[30], ind[3];
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int32_t)rotHist[i].size();
        orc_three_maxima(sizes, HISTO_LENGTH, ind);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind[0] && i != ind[1] && i != ind[2])
                for (int j : rotHist[i]) {
                    occupied[j] = 0;                   /* :1591 */
                    assigned[j] = -1;
                    nmatches--;
                }
    }
    return nmatches;
}

/* The selection part of ORBmatcher::Fuse(KeyFrame*, vpMapPoints, th) (:889-952, use_chi2 != 0) and of
 * Fuse(KeyFrame*, Scw, vpPoints, th, vpReplacePoint) (:1048-1082, use_chi2 == 0): best_idx[i] = the keypoint the MapPoint is
 * fused into (bestDist <= TH_LOW) or -1.  What happens to the map afterwards (:954-970, :1084-1095) is the caller's. */
int orc_fuse_select(const orc_grid *g, const uint8_t *kdesc, const float *uright, const float *inv_level_sigma2, int use_chi2,
                    int nq, const float *u, const float *v, const float *ur, const float *radius, const int32_t *level,
                    const uint8_t *qdesc, int32_t *best_idx) {
    int nFused = 0;
    std::vector<int32_t> vIndices(g->kx.size() + 1);
    for (int i = 0; i < nq; i++) {
        best_idx[i] = -1;
        const int n = orc_features_in_area(g, u[i], v[i], radius[i], -1, -1, vIndices.data());
        const int nPredictedLevel = level[i];
        int bestDist = 256, bestIdx = -1;
        for (int k = 0; k < n; k++) {
            const int idx = vIndices[k];
            const int kpLevel = g->koct[idx];
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            if (use_chi2) {
                if (uright && uright[idx] >= 0) {            /* :914-927 */
                    const float ex = u[i] - g->kx[idx], ey = v[i] - g->ky[idx], er = ur[i] - uright[idx];
                    const float e2 = ex * ex + ey * ey + er * er;
                    if (e2 * inv_level_sigma2[kpLevel] > 7.8) continue;
                } else {                                     /* :928-938 */
                    const float ex = u[i] - g->kx[idx], ey = v[i] - g->ky[idx];
                    const float e2 = ex * ex + ey * ey;
                    if (e2 * inv_level_sigma2[kpLevel] > 5.99) continue;
                }
            }
            const int dist = hamming(qdesc + (size_t)i * 32, kdesc + (size_t)idx * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist <= 50) {   /* TH_LOW, :952 / :1082 */
            best_idx[i] = bestIdx;
            nFused++;
        }
    }
    return nFused;
}

/* ORBmatcher::SearchBySim3 from the window searches on (:1188-1325).  Side-1 query a belongs to keypoint q1_index[a] of KF1 and
 * is searched in KF2's grid (g2), side-2 queries the other way round; match12[i1] = idx2 for mutual best matches, else -1. */
int orc_search_by_sim3(const orc_grid *g1, const uint8_t *kdesc1, const orc_grid *g2, const uint8_t *kdesc2, int nq1,
                       const int32_t *q1_index, const float *u1, const float *v1, const float *radius1, const int32_t *level1,
                       const uint8_t *qdesc1, int nq2, const int32_t *q2_index, const float *u2, const float *v2,
                       const float *radius2, const int32_t *level2, const uint8_t *qdesc2, int32_t *match12) {
    const int N1 = (int)g1->kx.size(), N2 = (int)g2->kx.size();
    std::vector<int> vnMatch1(N1, -1), vnMatch2(N2, -1);
    for (int a = 0; a < nq1; a++) {   /* :1188-1224 */
        int bestDist;
        const int bestIdx = window_best(g2, kdesc2, nullptr, u1[a], v1[a], radius1[a], level1[a] - 1, level1[a], false,
                                        qdesc1 + (size_t)a * 32, &bestDist);
        if (bestIdx >= 0 && bestDist <= 100) vnMatch1[q1_index[a]] = bestIdx;   /* TH_HIGH */
    }
    for (int a = 0; a < nq2; a++) {   /* :1268-1304 */
        int bestDist;
        const int bestIdx = window_best(g1, kdesc1, nullptr, u2[a], v2[a], radius2[a], level2[a] - 1, level2[a], false,
                                        qdesc2 + (size_t)a * 32, &bestDist);
        if (bestIdx >= 0 && bestDist <= 100) vnMatch2[q2_index[a]] = bestIdx;
    }
    int nFound = 0;                   /* :1307-1323 */
    for (int i1 = 0; i1 < N1; i1++) {
        match12[i1] = -1;
        const int idx2 = vnMatch1[i1];
        if (idx2 >= 0 && vnMatch2[idx2] == i1) {
            match12[i1] = idx2;
            nFound++;
        }
    }
    return nFound;
}

/* ORBmatcher::SearchForInitialization (:405-520).  g2 / kdesc2 / kangle2 / kxy2: F2's grid, descriptors, angles, mvKeysUn[*].pt;
 * side 1: F1.mvKeysUn octaves and angles, F1.mDescriptors; prev_xy = vbPrevMatched (updated at the end, :514-517). */
int orc_search_for_initialization(const orc_grid *g2, const uint8_t *kdesc2, const float *kangle2, int n1, const int32_t *oct1,
                                  const float *angle1, const uint8_t *desc1, float *prev_xy, int window_size, float nn_ratio,
                                  int check_orientation, int32_t *vnMatches12) {
    const int HISTO_LENGTH = 30;
    const int n2 = (int)g2->kx.size();
    int nmatches = 0;
    for (int i = 0; i < n1; i++) vnMatches12[i] = -1;
    std::vector<std::vector<int>> rotHist(HISTO_LENGTH);
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<int> vMatchedDistance(n2, INT_MAX), vnMatches21(n2, -1);
    std::vector<int32_t> vIndices2(n2 + 1);
    for (int i1 = 0; i1 < n1; i1++) {
        const int level1 = oct1[i1];
        if (level1 > 0) continue;
        const int n = orc_features_in_area(g2, prev_xy[2 * i1], prev_xy[2 * i1 + 1], (float)window_size, level1, level1, vIndices2.data());
        if (n == 0) continue;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int k = 0; k < n; k++) {
            const int i2 = vIndices2[k];
            const int dist = hamming(desc1 + (size_t)i1 * 32, kdesc2 + (size_t)i2 * 32);
            if (vMatchedDistance[i2] <= dist) continue;
            if (dist < bestDist) {
                bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2;
            } else if (dist < bestDist2) {
                bestDist2 = dist;
            }
        }
        if (bestDist <= 50) {
            if (bestDist < (float)bestDist2 * nn_ratio) {
                if (vnMatches21[bestIdx2] >= 0) {
                    vnMatches12[vnMatches21[bestIdx2]] = -1;
                    nmatches--;
                }
                vnMatches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (check_orientation) {
                    float rot = angle1[i1] - kangle2[bestIdx2];
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)std::round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back(i1);
                }
            }
        }
    }
    if (check_orientation) {
        int32_t sizes[30], ind[3];
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int32_t)rotHist[i].size();
        orc_three_maxima(sizes, HISTO_LENGTH, ind);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind[0] || i == ind[1] || i == ind[2]) continue;
            for (int idx1 : rotHist[i])
                if (vnMatches12[idx1] >= 0) {
                    vnMatches12[idx1] = -1;
                    nmatches--;
                }
        }
    }
    for (int i1 = 0; i1 < n1; i1++)      /* :514-517 */
        if (vnMatches12[i1] >= 0) {
            prev_xy[2 * i1] = g2->kx[vnMatches12[i1]];
            prev_xy[2 * i1 + 1] = g2->ky[vnMatches12[i1]];
        }
    return nmatches;
}

/* ORBmatcher::CheckDistEpipolarLine (:140-157); F12 row-major 3x3 */
static bool check_dist_epipolar_line(float x1, float y1, float x2, float y2, int oct2, const float *F12, const float *levelSigma2) {
    const float a = x1 * F12[0] + y1 * F12[3] + F12[6];
    const float b = x1 * F12[1] + y1 * F12[4] + F12[7];
    const float c = x1 * F12[2] + y1 * F12[5] + F12[8];
    const float num = a * x2 + b * y2 + c;
    const float den = a * a + b * b;
    if (den == 0) return false;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * levelSigma2[oct2];
}

/* ORBmatcher::SearchForTriangulation (:657-823) after the epipole has been computed (:664-670: ex, ey).  The shared nodes
 * come as in orc_search_by_bow.  has_mp1 / has_mp2: GetMapPoint(idx) != NULL; uright1 / uright2: mvuRight; xy*: mvKeysUn[*].pt;
 * vMatches12[idx1] = idx2 or -1 (the vMatchedPairs of :812-820 are its non-negative entries in index order). */
int orc_search_for_triangulation(const uint8_t *desc1, const uint8_t *has_mp1, const float *uright1, const float *xy1,
                                 const float *angle1, int n1, const uint8_t *desc2, const uint8_t *has_mp2, const float *uright2,
                                 const float *xy2, const int32_t *oct2, const float *angle2, int n2, const int32_t *off1,
                                 const int32_t *idx1, const int32_t *off2, const int32_t *idx2, int n_nodes, const float *F12,
                                 float ex, float ey, const float *scale_factors, const float *level_sigma2, int only_stereo,
                                 int check_orientation, int32_t *vMatches12) {
    const int HISTO_LENGTH = 30;
    const int TH_LOW = 50;
    int nmatches = 0;
    std::vector<char> vbMatched2((size_t)std::max(n2, 1), 0);   /* never set in the reference either (:677, :725) */
    for (int i = 0; i < n1; i++) vMatches12[i] = -1;
    std::vector<std::vector<int>> rotHist(HISTO_LENGTH);
    const float factor = 1.0f / HISTO_LENGTH;
    for (int k = 0; k < n_nodes; k++) {
        for (int a = off1[k]; a < off1[k + 1]; a++) {
            const int i1 = idx1[a];
            if (has_mp1[i1]) continue;                                   /* :699-703 */
            const bool bStereo1 = uright1[i1] >= 0;
            if (only_stereo && !bStereo1) continue;
            int bestDist = TH_LOW, bestIdx2 = -1;
            for (int b = off2[k]; b < off2[k + 1]; b++) {
                const int i2 = idx2[b];
                if (vbMatched2[i2] || has_mp2[i2]) continue;             /* :725 */
                const bool bStereo2 = uright2[i2] >= 0;
                if (only_stereo && !bStereo2) continue;
                const int dist = hamming(desc1 + (size_t)i1 * 32, desc2 + (size_t)i2 * 32);
                if (dist > TH_LOW || dist > bestDist) continue;          /* :738 */
                if (!bStereo1 && !bStereo2) {                            /* :743-749 */
                    const float distex = ex - xy2[2 * i2], distey = ey - xy2[2 * i2 + 1];
                    if (distex * distex + distey * distey < 100 * scale_factors[oct2[i2]]) continue;
                }
                if (check_dist_epipolar_line(xy1[2 * i1], xy1[2 * i1 + 1], xy2[2 * i2], xy2[2 * i2 + 1], oct2[i2], F12, level_sigma2)) {
                    bestIdx2 = i2;
                    bestDist = dist;
                }
            }
            if (bestIdx2 >= 0) {
                vMatches12[i1] = bestIdx2;
                nmatches++;
                if (check_orientation) {
                    float rot = angle1[i1] - angle2[bestIdx2];
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)std::round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back(i1);
                }
            }
        }
    }
    if (check_orientation) {
        int32_t sizes[30], ind[3];
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int32_t)rotHist[i].size();
        orc_three_maxima(sizes, HISTO_LENGTH, ind);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind[0] || i == ind[1] || i == ind[2]) continue;
            for (int j : rotHist[i]) {
                vMatches12[j] = -1;
                nmatches--;
            }
        }
    }
    return nmatches;
}

/* ------------------------------------------------------------ SearchByBoW (src/ORBmatcher.cc:159-288, :522-655) */
/* The loops over one shared node after the other, as the reference's merge walk visits them; see include/orbfe.h
 * (orbfe_search_by_bow) for the argument meaning. */
void orc_search_by_bow(const uint8_t *desc1, const uint8_t *valid1, int n1, const uint8_t *desc2, const uint8_t *blocked2, int n2,
                       const int32_t *off1, const int32_t *idx1, const int32_t *off2, const int32_t *idx2, int n_nodes,
                       int max_dist, float nn_ratio, int32_t *match1) {
    std::vector<char> matched2((size_t)std::max(n2, 1), 0);                     /* vpMapPointMatches[realIdxF] != NULL / vbMatched2 */
    for (int i = 0; i < n1; i++) match1[i] = -1;
    for (int k = 0; k < n_nodes; k++) {
        for (int a = off1[k]; a < off1[k + 1]; a++) {                           /* :187 / :554 */
            const int realIdx1 = idx1[a];
            if (valid1 && !valid1[realIdx1]) continue;                          /* :193-197 / :559-562 */
            int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
            for (int b = off2[k]; b < off2[k + 1]; b++) {                       /* :205 / :570 */
                const int realIdx2 = idx2[b];
                if (matched2[realIdx2] || (blocked2 && blocked2[realIdx2])) continue;   /* :209 / :576-580 */
                const int dist = hamming(desc1 + (size_t)realIdx1 * 32, desc2 + (size_t)realIdx2 * 32);
                if (dist < bestDist1) {
                    bestDist2 = bestDist1;
                    bestDist1 = dist;
                    bestIdx2 = realIdx2;
                } else if (dist < bestDist2) {
                    bestDist2 = dist;
                }
            }
            if (bestDist1 <= max_dist) {                                        /* :228 (<= TH_LOW) / :598 (< TH_LOW) */
                if (static_cast<float>(bestDist1) < nn_ratio * static_cast<float>(bestDist2)) {
                    match1[realIdx1] = bestIdx2;
                    matched2[bestIdx2] = 1;
                }
            }
        }
    }
}

/* ComputeThreeMaxima (:1601-1642) on the bin sizes */
void orc_three_maxima(const int32_t *sizes, int L, int32_t ind[3]) {
    int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = sizes[i];
        if (s > max1) {
            max3 = max2; max2 = max1; max1 = s;
            ind3 = ind2; ind2 = ind1; ind1 = i;
        } else if (s > max2) {
            max3 = max2; max2 = s;
            ind3 = ind2; ind2 = i;
        } else if (s > max3) {
            max3 = s; ind3 = i;
        }
    }
    if (max2 < 0.1f * (float)max1) {
        ind2 = -1; ind3 = -1;
    } else if (max3 < 0.1f * (float)max1) {
        ind3 = -1;
    }
    ind[0] = ind1; ind[1] = ind2; ind[2] = ind3;
}

/* the rotation-consistency step (:236-246, :267-285): matches whose angle difference falls outside the three fullest of
 * the 30 bins are dropped.  Returns the number of matches left. */
int orc_rotation_filter(const float *angle1, const float *angle2, int32_t *match1, int n1) {
    const int HISTO_LENGTH = 30;
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<std::vector<int>> rotHist(HISTO_LENGTH);
    int nmatches = 0;
    for (int i = 0; i < n1; i++) {
        if (match1[i] < 0) continue;
        float rot = angle1[i] - angle2[match1[i]];
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)std::round(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        rotHist[bin].push_back(i);
        nmatches++;
    }
    int32_t sizes[30], ind[3];
    for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int32_t)rotHist[i].size();
    orc_three_maxima(sizes, HISTO_LENGTH, ind);
    for (int i = 0; i < HISTO_LENGTH; i++) {
        if (i == ind[0] || i == ind[1] || i == ind[2]) continue;
        for (int j : rotHist[i]) {
            match1[j] = -1;
            nmatches--;
        }
    }
    return nmatches;
}

/* ------------------------------------------------------------ CPU baseline */
double orc_time_extract(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th, const uint8_t *imgs, int n_frames,
                        int w, int h, int stride, long frame_stride, int threads, int reps, long *total_kps) {
    if (threads < 1) threads = 1;
    std::atomic<int> next(0);
    std::atomic<long> kpsum(0);
    const int total = n_frames * reps;
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++)
        pool.emplace_back([&]() {
            orc_extractor *e = orc_create(nfeatures, scale_factor, nlevels, ini_th, min_th);
            std::vector<orc_keypoint> kps(nfeatures * 2 + 64);
            std::vector<uint8_t> desc((size_t)kps.size() * 32);
            for (;;) {
                int i = next.fetch_add(1);
                if (i >= total) break;
                int n = 0;
                orc_extract(e, imgs + (size_t)(i % n_frames) * frame_stride, w, h, stride, kps.data(), (int)kps.size(),
                            desc.data(), &n);
                kpsum += n;
            }
            orc_destroy(e);
        });
    for (auto &th : pool) th.join();
    auto t1 = std::chrono::steady_clock::now();
    if (total_kps) *total_kps = kpsum.load();
    return std::chrono::duration<double>(t1 - t0).count();
}

double orc_time_knn2(const uint8_t *q, int nq, const uint8_t *r, int nr, int threads, int swar) {
    if (threads < 1) threads = 1;
    std::vector<int32_t> bd(nq), bi(nq), sd(nq);
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++)
        pool.emplace_back([&, t]() {
            int lo = (int)((long)nq * t / threads), hi = (int)((long)nq * (t + 1) / threads);
            for (int i = lo; i < hi; i++) {
                int best1 = 256, best2 = 256, idx = -1;
                for (int j = 0; j < nr; j++) {
                    int d = swar ? hamming(q + (size_t)i * 32, r + (size_t)j * 32)
                                 : hamming_popcnt(q + (size_t)i * 32, r + (size_t)j * 32);
                    if (d < best1) { best2 = best1; best1 = d; idx = j; }
                    else if (d < best2) best2 = d;
                }
                bd[i] = best1; bi[i] = idx; sd[i] = best2;
            }
        });
    for (auto &th : pool) th.join();
    auto t1 = std::chrono::steady_clock::now();
    volatile int sink = bd[0] + bi[0] + sd[0];
    (void)sink;
    return std::chrono::duration<double>(t1 - t0).count();
}

}  // extern "C"
