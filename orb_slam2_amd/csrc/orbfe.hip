// orbfe.hip -- host side of liborbfe.so: the C-ABI of include/orbfe.h on top of the gfx950 kernels in k_*.h.
//
// Mirrors ORB_SLAM2::ORBextractor (reference include/ORBextractor.h:45-111, src/ORBextractor.cc) and the Hamming part
// of ORB_SLAM2::ORBmatcher (include/ORBmatcher.h:41-106, src/ORBmatcher.cc).  Compiled with -ffp-contract=off so the
// float arithmetic restated from the reference (scale tables, resize coefficients) is evaluated as written.
#include "../../include/orbfe.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.h"
#include "k_blur.h"
#include "k_describe.h"
#include "k_fast.h"
#include "k_match.h"
#include "k_pyramid.h"
#include "k_quadtree.h"
#include "quadtree_host.h"

using namespace orbfe;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

#define HIPCHK(expr)                                                                                          \
    do {                                                                                                      \
        hipError_t _e = (expr);                                                                               \
        if (_e != hipSuccess)                                                                                 \
            return fail(ORBFE_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e) + " (" __FILE__ ":" + \
                                              std::to_string(__LINE__) + ")");                                \
    } while (0)

inline int round_up(int v, int a) { return (v + a - 1) / a * a; }
inline int cv_round(double v) { return (int)std::nearbyint(v); }   // cvRound: ties to even

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t need) {
        if (need <= bytes) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        size_t want = need + need / 4;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) bytes = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};
struct PinBuf {
    void *p = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t need) {
        if (need <= bytes) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
        size_t want = need + need / 4;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e == hipSuccess) bytes = want;
        return e;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
    }
};

}  // namespace

// =====================================================================================================================
struct orbfe_extractor {
    orbfe_params p;
    int device = 0;
    hipStream_t stream = nullptr;
    // side stream: the blur runs here while the (latency-bound, 2 workgroups per CU) quadtree runs on `stream`
    hipStream_t stream2 = nullptr;
    hipEvent_t evFastDone = nullptr, evBlurDone = nullptr;
    // pyramid stream: with asynchronous batches queued back to back, the pyramid of batch k+1 (seven dependent, latency-bound
    // launches) runs on its own stream beside the quadtree / blur / descriptors of batch k -- never beside a FAST kernel (it
    // starts after FAST of batch k and FAST of batch k+1 waits for it).  Two pyramid buffers alternate.
    hipStream_t streamPyr = nullptr;
    hipEvent_t evPyrDone = nullptr, evDescDone[2] = {nullptr, nullptr};
    bool pyrBufBusy[2] = {false, false}, fastRecorded = false;
    int pyrFlip = 0;
    bool overlapPyr = true;             // ORBFE_OVERLAP_PYRAMID
    bool overlap = true;
    // constructor tables (reference src/ORBextractor.cc:415-469)
    double scaleFactor = 1.0;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    std::vector<int> mnFeaturesPerLevel, umax;
    // geometry for the current input size
    Geom g;
    bool haveGeom = false;
    std::vector<RszTab> hRsz;           // all levels' x and y tables
    std::vector<int> rszRows;           // output rows per wavefront of the resize kernel, per level
    std::vector<char> rszWide;          // 0: 8-byte source window, 1: 12-byte window (scale factor ~2), 2: gather kernel
    std::vector<BlurTile> hBlurTiles;
    std::vector<FastTile> hFastTiles;
    DevBuf dRsz, dBlurTiles, dFastTiles, dDescMask;
    // per-batch workspace
    DevBuf dPyr, dPyr2, dBlur, dCand, dSel, dSelCount;
    int numCUs = 256;
    // status word, per-level candidate counts and the cell table share one allocation: one memset per batch zeroes them
    DevBuf dCounters;
    int *pStatus = nullptr, *pLevelCount = nullptr;
    int2 *pCellTab = nullptr;
    size_t countersBytes = 0;
    PinBuf hCand, hLevelCount, hCellTab, hSel, hSelCount;
    // host-sequence path (orbfe_extract_sequence): two chunks in flight
    DevBuf seqIn[2], seqKps[2], seqDesc[2], seqCnt[2];
    hipStream_t seqInStream = nullptr, seqOutStream = nullptr;
    hipEvent_t seqEvIn[2] = {nullptr, nullptr}, seqEvComp[2] = {nullptr, nullptr}, seqEvOut[2] = {nullptr, nullptr};
    // single-image path
    DevBuf dImg, dKps, dDesc, dCounts;
    PinBuf hImg, hKps, hDesc, hCounts;
    PinBuf hMirror;                     // orbfe_pyramid_mirror: level 0 + the frame's pyramid block
    // last call (for orbfe_pyramid_level / orbfe_debug_stage)
    ImgSrc last{};
    int lastFrames = 0;
    bool overflow = false;
    // FAST kernel variant: 1 = dense (k_fast.h), 2 = the same kernel with the large per-wave survivor list (one wave per
    // workgroup).  A tile that overflows the normal list reports it, and the handle takes the large list for a while.  (A sparse
    // variant -- score network only where a compass pre-test passes -- was built twice and lost at every corner density:
    // profiles/r2c_fast_sparse_ab.txt, profiles/r3_fast_sparse_ab.txt.)
    int fastVariant = 1;
    int fastVariantBase = 1;            // the variant the handle was created with (ORBFE_FAST_VARIANT)
    // The large-list variant is a detour, not a state: after `demoteAfter` batches on it the handle tries its own variant
    // again (a frame of noise is an event, not a property of the sequence); if that overflows once more the batch is rerun
    // as before and the next detour lasts twice as long (16, 32, ... 4096 batches), so a sequence that does consist of noise
    // pays one rerun per detour, not one per batch.
    int bigListBatches = 0, demoteAfter = 16, promotions = 0;
    struct BatchCall { const uint8_t *imgs; int F, w, h, stride; size_t frameStride; orbfe_keypoint *kps; uint8_t *desc;
                       int32_t *counts; int cap; };
    std::vector<BatchCall> inFlight;    // every orbfe_extract_batch call since the last synchronisation, in order: the status
                                        // word cannot say which of them overflowed, so all of them are run again
    bool rerunPending = false;          // an overflow of queued batches was seen outside orbfe_synchronize (workspace growth)
    int batchesInFlight = 0;            // = inFlight.size() (kept for the paths that enqueue through run_batch directly)
    int quadtreeMode = ORBFE_QUADTREE_DEVICE;   // orbfe_set_quadtree_mode
    bool lastOnHost = false;                    // did the last call bring candidates / kept keypoints to the host
    PinBuf hStatus;
    // optional per-stage timing with HIP events on the handle's stream (orbfe_set_profiling)
    int profiling = 0;   // 0 off, 1 every stage, 2 FAST kernel only (two events per call)
    // profiling never blocks the stream: each call records one ProfSet; sets are folded into the sums when the profile is
    // read (orbfe_get_profile), when profiling is switched, or when kMaxPendingProf sets are outstanding
    struct ProfSet {
        hipEvent_t ev[6] = {};
        int *counts = nullptr;   // pinned: [2][F][kMaxLevels] (level counts, kept counts)
        int F = 0, capF = 0, nlevels = 0, mode = 1;
        bool hostTree = false;
    };
    std::vector<ProfSet> profPool;
    size_t profPending = 0;
    hipEvent_t *ev = nullptr;   // the set being recorded by the current call
    double stageMs[ORBFE_STAGE_COUNT] = {};
    long stageLaunches[ORBFE_STAGE_COUNT] = {};
    long stageFrames = 0;
    long candTotal = 0, kpTotal = 0;
};

constexpr size_t kMaxQueuedBatches = 4096;
constexpr int kRerunBigList = 1;   // internal: eval_status asks for the batch to be run again (next FAST kernel variant)
static int fetch_status(orbfe_extractor *e);

namespace {

// ORBextractor::ORBextractor, reference src/ORBextractor.cc:410-470
void build_tables(orbfe_extractor *e) {
    const int nlevels = e->p.nlevels;
    e->scaleFactor = e->p.scale_factor;   // double member initialised from the float argument (include/ORBextractor.h:98)
    e->mvScaleFactor.assign(nlevels, 1.0f);
    e->mvLevelSigma2.assign(nlevels, 1.0f);
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
    const float factor = (float)(1.0f / e->scaleFactor);
    float nDesired = e->p.nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int level = 0; level < nlevels - 1; level++) {
        e->mnFeaturesPerLevel[level] = cv_round(nDesired);
        sum += e->mnFeaturesPerLevel[level];
        nDesired *= factor;
    }
    e->mnFeaturesPerLevel[nlevels - 1] = std::max(e->p.nfeatures - sum, 0);
    // umax (:454-469)
    e->umax.assign(kHalfPatch + 1, 0);
    int v, v0;
    const int vmax = (int)std::floor(kHalfPatch * std::sqrt(2.f) / 2 + 1);
    const int vmin = (int)std::ceil(kHalfPatch * std::sqrt(2.f) / 2);
    const double hp2 = kHalfPatch * kHalfPatch;
    for (v = 0; v <= vmax; ++v) e->umax[v] = cv_round(std::sqrt(hp2 - v * v));
    for (v = kHalfPatch, v0 = 0; v >= vmin; --v) {
        while (e->umax[v0] == e->umax[v0 + 1]) ++v0;
        e->umax[v] = v0;
        ++v0;
    }
}

// cv::resize INTER_LINEAR coefficient set-up (OpenCV 2.4/3.2 generic path)
void resize_table(int ssize, int dsize, bool isX, int padTo, std::vector<RszTab> &out) {
    const double inv_scale = (double)dsize / ssize;
    const double scale = 1. / inv_scale;
    const size_t base = out.size();
    for (int d = 0; d < dsize; d++) {
        float fr = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(fr);
        fr -= s;
        if (isX) {
            if (s < 0) { fr = 0; s = 0; }
            if (s >= ssize - 1) { fr = 0; s = ssize - 1; }
        }
        auto sat16 = [](float v) { int i = cv_round(v); return i < -32768 ? -32768 : i > 32767 ? 32767 : i; };
        const int c0 = sat16((1.f - fr) * 2048), c1 = sat16(fr * 2048);
        RszTab t;
        t.ofs = s;
        t.coef = (c0 & 0xffff) | (c1 << 16);
        out.push_back(t);
    }
    const int padded = round_up(dsize, padTo);
    for (int d = dsize; d < padded; d++) out.push_back(out[base + dsize - 1]);
}

int build_geometry(orbfe_extractor *e, int w, int h) {
    if (e->haveGeom && e->g.w0 == w && e->g.h0 == h) return ORBFE_OK;
    Geom &g = e->g;
    memset(&g, 0, sizeof(g));
    const int nl = e->p.nlevels;
    g.nlevels = nl;
    g.w0 = w;
    g.h0 = h;
    e->hRsz.clear();
    e->hBlurTiles.clear();
    e->hFastTiles.clear();
    e->rszRows.assign(nl, 1);
    e->rszWide.assign(nl, 0);
    int pyrOff = 0, blurOff = 0, cellBase = 0, candOff = 0, selOff = 0;
    for (int l = 0; l < nl; l++) {
        LevelGeom &L = g.lv[l];
        const float scale = e->mvInvScaleFactor[l];
        L.w = cv_round((float)w * scale);   // reference src/ORBextractor.cc:1111-1112
        L.h = cv_round((float)h * scale);
        L.pitch = round_up(L.w + kPitchSlack, kPitchAlign);
        if (L.w > 65535 || L.h > 65535) return fail(ORBFE_ERR_UNSUPPORTED, "image larger than 65535 pixels on a side");
        L.pyrOff = pyrOff;
        if (l > 0) pyrOff += round_up(L.pitch * L.h, 256);
        L.blurOff = blurOff;
        blurOff += round_up(L.pitch * L.h, 256);
        // FAST cell grid (:773-787)
        const int maxBX = L.w - kEdge + 3, maxBY = L.h - kEdge + 3;
        const float width = (float)(maxBX - kMinBorder), height = (float)(maxBY - kMinBorder);
        L.nCols = (int)(width / 30.f);
        L.nRows = (int)(height / 30.f);
        if (L.nCols < 1 || L.nRows < 1)
            return fail(ORBFE_ERR_UNSUPPORTED, "pyramid level " + std::to_string(l) + " is smaller than one 30-px FAST cell "
                                               "plus borders (the reference divides by zero there)");
        L.wCell = (int)std::ceil(width / L.nCols);
        L.wCellRcp = (65536 + L.wCell - 1) / L.wCell;
        // the FAST kernels divide strip-relative pixel positions (one wavefront: < 4 * 64 + 8) by wCell with this reciprocal
        for (int n = 0; n < 4 * 64 + 8; n++)
            if ((int)(((uint32_t)n * (uint32_t)L.wCellRcp) >> 16) != n / L.wCell)
                return fail(ORBFE_ERR_UNSUPPORTED, "internal: reciprocal of the FAST cell width is inexact inside a strip");
        L.hCell = (int)std::ceil(height / L.nRows);
        if (L.wCell > 64 || L.hCell > 64)   // cannot happen for 30-px cells (wCell < 60), kept as a guard for the select kernel
            return fail(ORBFE_ERR_UNSUPPORTED, "FAST cell wider than a wavefront");
        {
            // FAST tiles: a wavefront takes whole cells, at most 8 of them and 62 dword columns (+ 1 halo lane each side).
            // Full strips get a wave each; the remainder strips of cell rows with the same height are packed together.
            L.fastCellsPerStrip = std::max(1, std::min(kFastMaxCells, (4 * kFastStripMaxDw) / L.wCell));
            auto seg_lanes = [&](int c0, int n) {
                const int xs = kEdge + c0 * L.wCell, xe = std::min(kEdge + (c0 + n) * L.wCell, L.w - kEdge);
                return (xe - xs + 3) / 4 + 2;
            };
            FastTile open{};           // the tile collecting remainder segments
            int openLanes = 0, openCells = 0, openDh = -1;
            auto flush_open = [&]() {
                if (open.nSeg) e->hFastTiles.push_back(open);
                open = FastTile{};
                openLanes = openCells = 0;
                openDh = -1;
            };
            for (int i = 0; i < L.nRows; i++) {
                const int y0 = kEdge + i * L.hCell;
                if (y0 >= L.h - kEdge) continue;
                const int dh = std::min(y0 + L.hCell, L.h - kEdge) - y0;
                for (int c0 = 0; c0 < L.nCols; c0 += L.fastCellsPerStrip) {
                    if (kEdge + c0 * L.wCell >= L.w - kEdge) continue;
                    const int n = std::min(L.fastCellsPerStrip, L.nCols - c0);
                    const int lanes = seg_lanes(c0, n);
                    FastSeg sg{};
                    sg.cellRow = (uint16_t)i; sg.c0 = (uint16_t)c0; sg.nCells = (uint8_t)n; sg.nLanes = (uint8_t)lanes;
                    if (n == L.fastCellsPerStrip || lanes > 32) {   // nothing else of this level fits beside it
                        FastTile t{};
                        t.level = (uint16_t)l; t.nSeg = 1;
                        t.seg[0] = sg;
                        e->hFastTiles.push_back(t);
                        continue;
                    }
                    if (open.nSeg && (dh != openDh || openLanes + lanes > 64 || openCells + n > kFastMaxCells ||
                                      open.nSeg == kFastMaxSeg))
                        flush_open();
                    sg.lane0 = (uint8_t)openLanes;
                    open.level = (uint16_t)l;
                    open.seg[open.nSeg++] = sg;
                    openLanes += lanes;
                    openCells += n;
                    openDh = dh;
                }
            }
            flush_open();
        }
        L.cellBase = cellBase;
        cellBase += L.nCols * L.nRows;
        {
            // candidate capacity = the most the level can yield: strict 3x3 NMS keeps at most one pixel of any 2x2 block, and
            // the cells' scan ranges tile the detection area (the reference's vector just grows; this one cannot overflow)
            int sx = 0, sy = 0;
            for (int j = 0; j < L.nCols; j++) {
                const int xs = kEdge + j * L.wCell, xe = std::min(xs + L.wCell, L.w - kEdge);
                if (xe > xs) sx += (xe - xs + 1) / 2;
            }
            for (int i = 0; i < L.nRows; i++) {
                const int ys = kEdge + i * L.hCell, ye = std::min(ys + L.hCell, L.h - kEdge);
                if (ye > ys) sy += (ye - ys + 1) / 2;
            }
            L.candCap = std::max(64, sx * sy);
        }
        L.candOff = candOff;
        candOff += L.candCap;
        // quadtree roots (:543-545)
        L.quota = e->mnFeaturesPerLevel[l];
        L.nIni = (int)std::round(static_cast<float>(maxBX - kMinBorder) / (maxBY - kMinBorder));
        if (L.nIni < 1) return fail(ORBFE_ERR_UNSUPPORTED, "portrait level: the reference's nIni rounds to 0");
        L.hX = static_cast<float>(maxBX - kMinBorder) / L.nIni;
        L.selCap = std::max(L.quota + 3, 4 * L.nIni) + 1;
        L.selOff = selOff;
        selOff += L.selCap;
        L.scale = e->mvScaleFactor[l];
        L.patch = (float)(int)(31 * e->mvScaleFactor[l]);   // scaledPatchSize (:837)
        if (l > 0) {
            const LevelGeom &P = g.lv[l - 1];
            L.rszOff = (int)e->hRsz.size();
            resize_table(P.w, L.w, true, kRszStripPx, e->hRsz);
            L.rszOffY = (int)e->hRsz.size();
            resize_table(P.h, L.h, false, 64, e->hRsz);
            for (int i = 0; i < 64; i++) e->hRsz.push_back(e->hRsz.back());   // a row block reads 64 entries from its first row
            // output rows per wavefront: the largest block whose source rows fit the kernel's register window
            const RszTab *xt = &e->hRsz[L.rszOff], *yt = &e->hRsz[L.rszOffY];
            auto srow = [&](int v) { return std::min(std::max(v, 0), P.h - 1); };
            int rows = 0;
            for (int r = 16; r >= 1 && !rows; r--) {
                bool ok = true;
                for (int y0 = 0; y0 < L.h && ok; y0 += r) {
                    const int yl = std::min(y0 + r, L.h) - 1;
                    ok = srow(yt[yl].ofs + 1) - srow(yt[y0].ofs) + 1 <= kRszSrcRows;
                }
                if (ok) rows = r;
            }
            // the lane's 4 source pixel pairs must lie inside its 8-byte window (scale factors below ~2) or, with the wide
            // variant of the kernel, pixels 0..2 inside 8 bytes and pixel 3 inside bytes 3..10 of a 12-byte window
            bool narrow = rows > 0 && P.w >= 12, wide = narrow;
            for (int x0 = 0; x0 < L.w && wide; x0 += 4) {
                const int l2 = std::min(x0 + 2, L.w - 1), l3 = std::min(x0 + 3, L.w - 1);
                const int end3 = std::min(xt[l3].ofs + 1, P.w - 1), end2 = std::min(xt[l2].ofs + 1, P.w - 1);
                const int b8 = std::min(xt[x0].ofs, round_up(P.w, 16) - 8), b12 = std::min(xt[x0].ofs, round_up(P.w, 16) - 12);
                if (xt[l3].ofs < xt[x0].ofs) wide = narrow = false;
                if (end3 - b8 > 7) narrow = false;
                // (pixel 3 past the last column is a padded table entry whose result lands in the row padding)
                if (end2 - b12 > 7 || end3 - b12 > 10 || (x0 + 3 < L.w && xt[l3].ofs - b12 < 3)) wide = false;
            }
            e->rszWide[l] = narrow ? 0 : wide ? 1 : 2;   // 2: the gather kernel, any ratio
            if (L.h > 65535) return fail(ORBFE_ERR_UNSUPPORTED, "image larger than 65535 pixels on a side");
            e->rszRows[l] = rows;
        }
        // blur strips: a lane column is one dword; interior strips write lanes 1..62 (lanes 0 and 63 are halo columns), the
        // strip at the left image edge also writes lane 0 and the one at the right edge may end in lane 63 (reflection
        // needs no neighbour there): 1241 px = 311 dwords = 63 + 3 x 62 + 63 take 5 strips, not 6
        const int wd = (L.w + 3) / 4;
        for (int ty = 0; ty * kBlurRows < L.h; ty++)
            for (int d0 = 0; d0 < wd;) {
                BlurTile t;
                t.level = (uint16_t)l; t.ty = (uint16_t)ty;
                const int first = d0 == 0 ? 0 : 1;          // lane of dword column d0
                const int room = 64 - first;                // lanes first..63
                const int left = wd - d0;
                const int n = left <= room ? left : room - 1;   // not the last strip: lane 63 stays a halo column
                t.xdw0 = (int16_t)(d0 - first);
                t.lane0 = (uint8_t)first; t.lane1 = (uint8_t)(first + n - 1);
                e->hBlurTiles.push_back(t);
                d0 += n;
            }
    }
    g.totalCells = cellBase;
    g.pyrFrameBytes = std::max(pyrOff, 256);
    g.blurFrameBytes = blurOff;
    g.candFrameCap = candOff;
    g.selFrameCap = selOff;
    HIPCHK(e->dRsz.ensure(e->hRsz.size() * sizeof(RszTab) + 16));
    if (!e->hRsz.empty())
        HIPCHK(hipMemcpyAsync(e->dRsz.p, e->hRsz.data(), e->hRsz.size() * sizeof(RszTab), hipMemcpyHostToDevice, e->stream));
    HIPCHK(e->dBlurTiles.ensure(e->hBlurTiles.size() * sizeof(BlurTile)));
    HIPCHK(hipMemcpyAsync(e->dBlurTiles.p, e->hBlurTiles.data(), e->hBlurTiles.size() * sizeof(BlurTile),
                          hipMemcpyHostToDevice, e->stream));
    HIPCHK(e->dFastTiles.ensure(e->hFastTiles.size() * sizeof(FastTile)));
    HIPCHK(hipMemcpyAsync(e->dFastTiles.p, e->hFastTiles.data(), e->hFastTiles.size() * sizeof(FastTile),
                          hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->haveGeom = true;
    return ORBFE_OK;
}

int ensure_workspace(orbfe_extractor *e, int F, bool twoPyramids) {
    const Geom &g = e->g;
    HIPCHK(e->dPyr.ensure((size_t)F * g.pyrFrameBytes));
    if (twoPyramids) HIPCHK(e->dPyr2.ensure((size_t)F * g.pyrFrameBytes));
    HIPCHK(e->dBlur.ensure((size_t)F * g.blurFrameBytes));
    HIPCHK(e->dCand.ensure((size_t)F * g.candFrameCap * sizeof(Cand)));
    {
        const size_t lcBytes = (size_t)F * kMaxLevels * sizeof(int), ctBytes = (size_t)F * g.totalCells * sizeof(int2);
        e->countersBytes = 64 + lcBytes + ctBytes;
        if (e->countersBytes > e->dCounters.bytes) {   // about to move: whatever an unfinished batch reported goes first
            // (an overflow of the batches already queued is remembered for orbfe_synchronize, which runs them again)
            const int rc = fetch_status(e);
            if (rc == kRerunBigList) e->rerunPending = true;
            else if (rc) return rc;
        }
        HIPCHK(e->dCounters.ensure(e->countersBytes));
        if (e->pStatus != (int *)e->dCounters.p) HIPCHK(hipMemsetAsync(e->dCounters.p, 0, 64, e->stream));
        e->pStatus = (int *)e->dCounters.p;
        e->pLevelCount = (int *)((uint8_t *)e->dCounters.p + 64);
        e->pCellTab = (int2 *)((uint8_t *)e->dCounters.p + 64 + lcBytes);
    }
    HIPCHK(e->dSel.ensure((size_t)F * g.selFrameCap * sizeof(SelKp)));
    HIPCHK(e->dSelCount.ensure((size_t)F * kMaxLevels * sizeof(int)));
    HIPCHK(e->hStatus.ensure(16));
    return ORBFE_OK;
}

// pinned host mirrors of the candidate / selection arrays: only the host quadtree and the debug exports read them
int ensure_host_mirrors(orbfe_extractor *e, int F) {
    const Geom &g = e->g;
    HIPCHK(e->hCand.ensure((size_t)F * g.candFrameCap * sizeof(Cand)));
    HIPCHK(e->hLevelCount.ensure((size_t)F * kMaxLevels * sizeof(int)));
    HIPCHK(e->hCellTab.ensure((size_t)F * g.totalCells * sizeof(int2)));
    HIPCHK(e->hSel.ensure((size_t)F * g.selFrameCap * sizeof(SelKp)));
    HIPCHK(e->hSelCount.ensure((size_t)F * kMaxLevels * sizeof(int)));
    return ORBFE_OK;
}

// candidates of (frame f, level l) in the reference's order: cells row-major, inside a cell row-major (:789-829)
void ordered_candidates(const orbfe_extractor *e, int f, int l, std::vector<Cand> &out) {
    const Geom &g = e->g;
    const LevelGeom &L = g.lv[l];
    const int2 *tab = (const int2 *)e->hCellTab.p + (size_t)f * g.totalCells + L.cellBase;
    const Cand *cand = (const Cand *)e->hCand.p + (size_t)f * g.candFrameCap + L.candOff;
    out.clear();
    for (int c = 0; c < L.nCols * L.nRows; c++) {
        const int off = tab[c].x, n = tab[c].y;
        for (int k = 0; k < n && off + k < L.candCap; k++) out.push_back(cand[off + k]);
    }
}

// copyMakeBorder(..., BORDER_REFLECT_101) (:1122-1128) around a w x h ROI that sits kEdge pixels inside `buf` (row pitch
// w + 2 kEdge): the frame is API surface only, built on the host
void reflect101_frame(uint8_t *buf, int w, int h) {
    const int bw = w + 2 * kEdge;
    uint8_t *roi = buf + (size_t)kEdge * bw + kEdge;
    auto refl = [](int p, int n) { p = p < 0 ? -p : p; return p >= n ? 2 * (n - 1) - p : p; };
    for (int y = 0; y < h; y++) {
        uint8_t *row = roi + (size_t)y * bw;
        for (int k = 1; k <= kEdge; k++) {
            row[-k] = row[refl(-k, w)];
            row[w - 1 + k] = row[refl(w - 1 + k, w)];
        }
    }
    for (int k = 1; k <= kEdge; k++) {
        memcpy(buf + (size_t)(kEdge - k) * bw, buf + (size_t)(kEdge + refl(-k, h)) * bw, bw);
        memcpy(buf + (size_t)(kEdge + h - 1 + k) * bw, buf + (size_t)(kEdge + refl(h - 1 + k, h)) * bw, bw);
    }
}

constexpr int kQtMaxNodes = 1024;   // live quadtree nodes the device kernel's LDS tables hold

// The device quadtree needs max(quota+3, 4*nIni) <= kQtMaxNodes on every level; larger quotas (nfeatures beyond ~4700 at
// scale 1.2) use the host quadtree, as does an explicit orbfe_set_quadtree_mode(ORBFE_QUADTREE_HOST).
bool quadtree_on_host(const orbfe_extractor *e) {
    if (e->quadtreeMode == ORBFE_QUADTREE_HOST) return true;
    for (int l = 0; l < e->g.nlevels; l++)
        if (std::max(e->g.lv[l].quota + 3, 4 * e->g.lv[l].nIni) > kQtMaxNodes || e->g.lv[l].nIni > 64) return true;
    return false;
}

// candidates to the host, DistributeOctTree on host threads (one (frame, level) task each), kept keypoints back
int host_quadtree_stage(orbfe_extractor *e, int F) {
    const Geom &g = e->g;
    hipStream_t st = e->stream;
    if (int rc = ensure_host_mirrors(e, F)) return rc;
    HIPCHK(hipMemcpyAsync(e->hLevelCount.p, e->pLevelCount, (size_t)F * kMaxLevels * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(e->hCellTab.p, e->pCellTab, (size_t)F * g.totalCells * sizeof(int2), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(e->hCand.p, e->dCand.p, (size_t)F * g.candFrameCap * sizeof(Cand), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    auto t0 = std::chrono::steady_clock::now();
    {
        const int *lc = (const int *)e->hLevelCount.p;
        for (int f = 0; f < F; f++)
            for (int l = 0; l < g.nlevels; l++)
                if (lc[f * kMaxLevels + l] > g.lv[l].candCap) e->overflow = true;
    }
    const int tasks = F * g.nlevels;
    int nthreads = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
    nthreads = std::min(nthreads, tasks);
    std::atomic<int> next(0);
    auto worker = [&]() {
        QuadtreeHost qt;
        std::vector<Cand> kept, ordered;
        for (;;) {
            const int t = next.fetch_add(1);
            if (t >= tasks) break;
            const int f = t / g.nlevels, l = t % g.nlevels;
            const LevelGeom &L = g.lv[l];
            ordered_candidates(e, f, l, ordered);
            const int width = (L.w - kEdge + 3) - kMinBorder, height = (L.h - kEdge + 3) - kMinBorder;
            qt.run(ordered.data(), (int)ordered.size(), width, height, L.quota, L.nIni, L.hX, kept);
            SelKp *sel = (SelKp *)e->hSel.p + (size_t)f * g.selFrameCap + L.selOff;
            const int n = std::min((int)kept.size(), L.selCap);
            for (int k = 0; k < n; k++) {
                sel[k].x = (uint16_t)(kept[k].x + kMinBorder);   // :843-844
                sel[k].y = (uint16_t)(kept[k].y + kMinBorder);
                sel[k].resp = kept[k].resp;
                sel[k].level = (uint16_t)l;
            }
            ((int *)e->hSelCount.p)[f * kMaxLevels + l] = n;
        }
    };
    if (nthreads <= 1) {
        worker();
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nthreads; t++) pool.emplace_back(worker);
        for (auto &th : pool) th.join();
    }
    if (e->profiling == 1)
        e->stageMs[ORBFE_STAGE_QUADTREE] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    HIPCHK(hipMemcpyAsync(e->dSel.p, e->hSel.p, (size_t)F * g.selFrameCap * sizeof(SelKp), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(e->dSelCount.p, e->hSelCount.p, (size_t)F * kMaxLevels * sizeof(int), hipMemcpyHostToDevice, st));
    return ORBFE_OK;
}

constexpr size_t kMaxPendingProf = 256;

// fold every recorded-but-unread event set into the sums (drains the stream first)
int prof_resolve(orbfe_extractor *e) {
    if (e->profPending == 0) return ORBFE_OK;
    HIPCHK(hipStreamSynchronize(e->stream));
    for (size_t i = 0; i < e->profPending; i++) {
        orbfe_extractor::ProfSet &p = e->profPool[i];
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, p.ev[1], p.ev[2])); e->stageMs[ORBFE_STAGE_FAST] += ms;
        e->stageLaunches[ORBFE_STAGE_FAST] += 1;
        if (p.mode == 1) {
            HIPCHK(hipEventElapsedTime(&ms, p.ev[0], p.ev[1])); e->stageMs[ORBFE_STAGE_PYRAMID] += ms;
            HIPCHK(hipEventElapsedTime(&ms, p.ev[2], p.ev[3])); e->stageMs[ORBFE_STAGE_BLUR] += ms;
            if (!p.hostTree) { HIPCHK(hipEventElapsedTime(&ms, p.ev[3], p.ev[4])); e->stageMs[ORBFE_STAGE_QUADTREE] += ms; }
            HIPCHK(hipEventElapsedTime(&ms, p.ev[4], p.ev[5])); e->stageMs[ORBFE_STAGE_DESCRIBE] += ms;
            e->stageLaunches[ORBFE_STAGE_PYRAMID] += p.nlevels - 1;
            e->stageLaunches[ORBFE_STAGE_BLUR] += 1;
            e->stageLaunches[ORBFE_STAGE_QUADTREE] += 1;
            e->stageLaunches[ORBFE_STAGE_DESCRIBE] += 1;
        }
        e->stageFrames += p.F;
        if (p.mode == 1) {   // candidate / keypoint counts come back only with full profiling (mode 2 adds no copies)
            const int *lc = p.counts, *sc = p.counts + (size_t)p.F * kMaxLevels;
            for (int f = 0; f < p.F; f++)
                for (int l = 0; l < p.nlevels; l++) { e->candTotal += lc[f * kMaxLevels + l]; e->kpTotal += sc[f * kMaxLevels + l]; }
        }
    }
    e->profPending = 0;
    return ORBFE_OK;
}

int prof_acquire(orbfe_extractor *e, int F, orbfe_extractor::ProfSet **out) {
    if (e->profPending >= kMaxPendingProf) {
        int rc = prof_resolve(e);
        if (rc) return rc;
    }
    if (e->profPending >= e->profPool.size()) {
        orbfe_extractor::ProfSet p;
        for (auto &ev : p.ev) HIPCHK(hipEventCreate(&ev));
        e->profPool.push_back(p);
    }
    orbfe_extractor::ProfSet &p = e->profPool[e->profPending];
    if (p.capF < F) {
        if (p.counts) (void)hipHostFree(p.counts);
        p.counts = nullptr;
        HIPCHK(hipHostMalloc((void **)&p.counts, (size_t)2 * F * kMaxLevels * sizeof(int), hipHostMallocDefault));
        p.capF = F;
    }
    p.F = F;
    p.nlevels = e->g.nlevels;
    *out = &p;
    return ORBFE_OK;
}

// The pipeline for F frames resident on the device.  Outputs on the device.
int run_batch(orbfe_extractor *e, const uint8_t *dImgs, int F, int w, int h, int stride, size_t frameStride,
              orbfe_keypoint *dKps, uint8_t *dDesc, int32_t *dCounts, int cap, bool queued = false) {
    int rc = build_geometry(e, w, h);
    if (rc) return rc;
    // `queued`: an asynchronous orbfe_extract_batch call whose input frames are ready before the call (no copy of this library
    // precedes it on the stream) -- its pyramid may run ahead on the pyramid stream
    const bool forkPyr = queued && e->overlap && e->overlapPyr && e->profiling != 1 && !quadtree_on_host(e) && e->g.nlevels > 1;
    rc = ensure_workspace(e, F, forkPyr);
    if (rc) return rc;
    const Geom &g = e->g;
    hipStream_t st = e->stream;
    const int pb = forkPyr ? (e->pyrFlip ^= 1) : 0;
    ImgSrc src;
    src.img0 = dImgs;
    src.pitch0 = stride;
    src.frameStride0 = frameStride;
    src.pyr = (uint8_t *)(pb ? e->dPyr2.p : e->dPyr.p);
    src.blur = (uint8_t *)e->dBlur.p;
    e->last = src;
    e->lastFrames = F;
    e->overflow = false;

    const bool profAny = e->profiling != 0, prof = e->profiling == 1;
    orbfe_extractor::ProfSet *ps = nullptr;
    if (profAny) {
        rc = prof_acquire(e, F, &ps);
        if (rc) return rc;
        e->ev = ps->ev;
        ps->mode = e->profiling;
        if (prof) HIPCHK(hipEventRecord(e->ev[0], st));
    }
    // K1: pyramid, level by level (level l reads level l-1)
    hipStream_t spyr = forkPyr ? e->streamPyr : st;
    if (forkPyr) {
        if (e->fastRecorded) HIPCHK(hipStreamWaitEvent(spyr, e->evFastDone, 0));        // after the previous batch's FAST kernel
        if (e->pyrBufBusy[pb]) HIPCHK(hipStreamWaitEvent(spyr, e->evDescDone[pb], 0));  // the buffer's last reader has finished
    } else if (e->pyrBufBusy[0]) {
        // a batch that does not fork writes buffer 0 on the main stream: stream order already puts it behind every earlier batch
        e->pyrBufBusy[0] = false;
    }
    for (int l = 1; l < g.nlevels; l++) {
        const LevelGeom &L = g.lv[l], &P = g.lv[l - 1];
        ResizeArgs a;
        int sp;
        a.src = level_ptr(src, g, 0, l - 1, &sp);
        a.sPitch = sp;
        a.sFrameStride = (l - 1 == 0) ? frameStride : (size_t)g.pyrFrameBytes;
        a.sw = P.w; a.sh = P.h;
        a.dst = src.pyr + L.pyrOff;
        a.dPitch = L.pitch;
        a.dFrameStride = g.pyrFrameBytes;
        a.dw = L.w; a.dh = L.h;
        a.xtab = (const RszTab *)e->dRsz.p + L.rszOff;
        a.ytab = (const RszTab *)e->dRsz.p + L.rszOffY;
        a.nStrips = (L.w + kRszStripPx - 1) / kRszStripPx;
        a.rows = e->rszRows[l];
        a.nRowBlocks = (L.h + a.rows - 1) / a.rows;
        if (e->rszWide[l] == 2)
            hipLaunchKernelGGL(pyr_resize_generic_kernel, dim3((L.w + 1023) / 1024, L.h, F), dim3(256), 0, spyr, a);
        else if (e->rszWide[l])
            hipLaunchKernelGGL(pyr_resize_kernel<true>, dim3((a.nStrips * a.nRowBlocks + 3) / 4, F), dim3(256), 0, spyr, a);
        else
            hipLaunchKernelGGL(pyr_resize_kernel<false>, dim3((a.nStrips * a.nRowBlocks + 3) / 4, F), dim3(256), 0, spyr, a);
    }
    if (forkPyr) {
        HIPCHK(hipEventRecord(e->evPyrDone, spyr));
        HIPCHK(hipStreamWaitEvent(st, e->evPyrDone, 0));
    }
    // K3: FAST over every cell of every level
    // (the status word in the first 64 bytes stays: it collects over batches until the host has read it)
    HIPCHK(hipMemsetAsync((uint8_t *)e->dCounters.p + 64, 0, e->countersBytes - 64, st));
    if (profAny) HIPCHK(hipEventRecord(e->ev[1], st));
    {
        const int iniTh = std::min(std::max(e->p.ini_th_fast, 0), 255), minTh = std::min(std::max(e->p.min_th_fast, 0), 255);
        FastArgs a;
        a.g = g; a.s = src;
        a.tiles = (const FastTile *)e->dFastTiles.p;
        a.nTiles = (int)e->hFastTiles.size();
        a.cand = (Cand *)e->dCand.p;
        a.levelCount = e->pLevelCount;
        a.cellTab = e->pCellTab;
        a.status = e->pStatus;
        // iniTh < minTh would make the fallback a no-op in the reference too (FAST at the lower threshold runs first)
        a.iniTh = std::max(iniTh, minTh);
        a.minTh = std::min(iniTh, minTh);
        if (iniTh < minTh) a.iniTh = a.minTh = iniTh;
        // Threshold 0 gives what threshold 1 gives: cv::FAST stores S - 1 as a corner's score and its NMS wants that
        // strictly above the 0 of a non-corner, so a corner with S = 1 never survives; the kernel compares raw S, so the
        // same rule is "S > max(t, 1)" (for t >= 1 the two readings agree, see k_fast.h).
        a.iniTh = std::max(a.iniTh, 1);
        a.minTh = std::max(a.minTh, 1);
        if (e->fastVariant >= 2 && e->fastVariantBase < 2 && ++e->bigListBatches > e->demoteAfter) {
            e->fastVariant = e->fastVariantBase;   // the detour is over: back to the handle's own variant
            e->bigListBatches = 0;
        }
        if (e->fastVariant >= 2)   // a strip of an earlier batch produced more survivor records than the normal variant holds
            hipLaunchKernelGGL((fast_cells_kernel<kFastListCapMax, 1>), dim3(a.nTiles, F), dim3(64), 0, st, a);
        else
            hipLaunchKernelGGL((fast_cells_kernel<kFastListCap, 4>), dim3((a.nTiles + 3) / 4, F), dim3(256), 0, st, a);
    }
    if (profAny) HIPCHK(hipEventRecord(e->ev[2], st));
    const bool hostTree = quadtree_on_host(e);
    e->lastOnHost = hostTree;
    // K5: blur of every level.  It depends on the pyramid only and the descriptors are its only consumer, so outside
    // per-stage profiling it runs on the side stream, started when FAST has finished: blur (VALU-bound) then shares the
    // GPU with the quadtree (latency-bound, one workgroup per frame and level) instead of running after it.
    const bool fork = e->overlap && !prof && !hostTree;
    hipStream_t sb = fork ? e->stream2 : st;
    if (fork) {
        HIPCHK(hipEventRecord(e->evFastDone, st));
        e->fastRecorded = true;
        HIPCHK(hipStreamWaitEvent(sb, e->evFastDone, 0));
    }
    hipLaunchKernelGGL(gauss7_kernel, dim3(((unsigned)e->hBlurTiles.size() + 3) / 4, F), dim3(256), 0, sb, g, src,
                       (const BlurTile *)e->dBlurTiles.p, (int)e->hBlurTiles.size());
    if (fork) HIPCHK(hipEventRecord(e->evBlurDone, sb));
    if (prof) HIPCHK(hipEventRecord(e->ev[3], st));
    HIPCHK(hipGetLastError());

    if (hostTree) {
        rc = host_quadtree_stage(e, F);
        if (rc) return rc;
    } else {
        // N1: DistributeOctTree on the device, one workgroup per (frame, level); nothing leaves HBM
        QuadtreeArgs a;
        a.g = g;
        a.cand = (Cand *)e->dCand.p;
        a.levelCount = e->pLevelCount;
        a.sel = (SelKp *)e->dSel.p;
        a.selCount = (int *)e->dSelCount.p;
        a.status = e->pStatus;
        // node tables sized to what the largest level needs: smaller tables = more trees resident per CU
        int need = 0;
        for (int l = 0; l < g.nlevels; l++) need = std::max(need, std::max(g.lv[l].quota + 3, 4 * g.lv[l].nIni));
        if (need <= 256)
            hipLaunchKernelGGL(quadtree_kernel<256>, dim3(g.nlevels, F), dim3(256), sizeof(QtShared<256>), st, a);
        else if (need <= 512)
            hipLaunchKernelGGL(quadtree_kernel<512>, dim3(g.nlevels, F), dim3(256), sizeof(QtShared<512>), st, a);
        else
            hipLaunchKernelGGL(quadtree_kernel<kQtMaxNodes>, dim3(g.nlevels, F), dim3(256), sizeof(QtShared<kQtMaxNodes>), st, a);
        HIPCHK(hipGetLastError());
    }
    // K4 + K6
    if (fork) HIPCHK(hipStreamWaitEvent(st, e->evBlurDone, 0));
    if (prof) HIPCHK(hipEventRecord(e->ev[4], st));
    {
        DescribeArgs a;
        a.g = g; a.s = src;
        a.sel = (const SelKp *)e->dSel.p;
        a.selCount = (const int *)e->dSelCount.p;
        a.kps = (KeyPointOut *)dKps;
        a.desc = dDesc;
        a.counts = dCounts;
        a.cap = cap;
        const int maxKp = std::min(g.selFrameCap, cap);
        a.nFrames = F;
        a.maskTab = (const uint4 *)e->dDescMask.p;
        a.groupsPerFrame = (maxKp + kDescKpPerGroup - 1) / kDescKpPerGroup;
        hipLaunchKernelGGL(orient_describe_kernel, dim3((unsigned)a.groupsPerFrame * ((F + 7) / 8 * 8)), dim3(256), 0, st, a);
    }
    HIPCHK(hipGetLastError());
    if (forkPyr) {   // the pyramid buffer's last reader: the moments of the descriptor kernel
        HIPCHK(hipEventRecord(e->evDescDone[pb], st));
        e->pyrBufBusy[pb] = true;
    }
    if (profAny) {
        if (prof) HIPCHK(hipEventRecord(e->ev[5], st));
        ps->hostTree = hostTree;
        if (prof) HIPCHK(hipMemcpyAsync(ps->counts, e->pLevelCount, (size_t)F * kMaxLevels * sizeof(int), hipMemcpyDeviceToHost, st));
        if (prof) HIPCHK(hipMemcpyAsync(ps->counts + (size_t)F * kMaxLevels, e->dSelCount.p, (size_t)F * kMaxLevels * sizeof(int),
                              hipMemcpyDeviceToHost, st));
        e->profPending++;
    }
    return ORBFE_OK;
}

}  // namespace

// =====================================================================================================================
extern "C" {

const char *orbfe_last_error(void) { return g_err.c_str(); }

int orbfe_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int orbfe_extractor_create(const orbfe_params *params, int device, orbfe_extractor **out) {
    if (!params || !out) return fail(ORBFE_ERR_ARG, "null argument");
    if (params->nlevels < 1 || params->nlevels > kMaxLevels) return fail(ORBFE_ERR_ARG, "nlevels must be in [1,16]");
    if (!(params->scale_factor > 1.0f) || !(params->scale_factor <= 16.0f))   // 1 makes the reference's quota formula 0/0
        return fail(ORBFE_ERR_UNSUPPORTED, "scale_factor must be in (1, 16]");
    if (params->nfeatures < 1) return fail(ORBFE_ERR_ARG, "nfeatures must be positive");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(ORBFE_ERR_DEVICE, "no HIP device: liborbfe has no CPU path");
    if (device < 0 || device >= ndev) return fail(ORBFE_ERR_ARG, "device index out of range");
    HIPCHK(hipSetDevice(device));
    orbfe_extractor *e = new orbfe_extractor;
    e->p = *params;
    e->device = device;
    build_tables(e);
    static const int kUmax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
    if (!std::equal(e->umax.begin(), e->umax.end(), kUmax)) {
        delete e;
        return fail(ORBFE_ERR_DEVICE, "umax table mismatch");
    }
    hipError_t er = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
    if (er != hipSuccess) {
        delete e;
        return fail(ORBFE_ERR_DEVICE, std::string("hipStreamCreate: ") + hipGetErrorString(er));
    }
    er = hipStreamCreateWithFlags(&e->stream2, hipStreamNonBlocking);
    if (er == hipSuccess) er = hipStreamCreateWithFlags(&e->streamPyr, hipStreamNonBlocking);
    if (er == hipSuccess) er = hipEventCreateWithFlags(&e->evFastDone, hipEventDisableTiming);
    if (er == hipSuccess) er = hipEventCreateWithFlags(&e->evBlurDone, hipEventDisableTiming);
    if (er == hipSuccess) er = hipEventCreateWithFlags(&e->evPyrDone, hipEventDisableTiming);
    if (er == hipSuccess) er = hipEventCreateWithFlags(&e->evDescDone[0], hipEventDisableTiming);
    if (er == hipSuccess) er = hipEventCreateWithFlags(&e->evDescDone[1], hipEventDisableTiming);
    if (er != hipSuccess) {
        orbfe_extractor_destroy(e);
        return fail(ORBFE_ERR_DEVICE, std::string("side stream: ") + hipGetErrorString(er));
    }
    {
        // the descriptor kernel's disc mask table, from this object's umax (checked above against the reference's values)
        std::vector<uint2> tab((size_t)kDescMaskTabU4 * 2);
        desc_mask_table(e->umax.data(), tab.data());
        er = e->dDescMask.ensure(tab.size() * sizeof(uint2));
        if (er == hipSuccess) er = hipMemcpy(e->dDescMask.p, tab.data(), tab.size() * sizeof(uint2), hipMemcpyHostToDevice);
        if (er != hipSuccess) {
            orbfe_extractor_destroy(e);
            return fail(ORBFE_ERR_DEVICE, std::string("mask table: ") + hipGetErrorString(er));
        }
    }
    if (const char *ov = getenv("ORBFE_OVERLAP")) e->overlap = atoi(ov) != 0;
    if (const char *ov = getenv("ORBFE_OVERLAP_PYRAMID")) e->overlapPyr = atoi(ov) != 0;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) e->numCUs = cus;
    }
    if (const char *fv = getenv("ORBFE_FAST_VARIANT")) e->fastVariant = std::min(std::max(atoi(fv), 1), 2);
    e->fastVariantBase = e->fastVariant;
    *out = e;
    return ORBFE_OK;
}

int orbfe_set_profiling(orbfe_extractor *e, int enable) {
    if (!e) return fail(ORBFE_ERR_ARG, "null extractor");
    HIPCHK(hipSetDevice(e->device));
    int rc = prof_resolve(e);
    if (rc) return rc;
    e->profiling = enable == 2 ? 2 : (enable != 0 ? 1 : 0);
    for (int i = 0; i < ORBFE_STAGE_COUNT; i++) { e->stageMs[i] = 0; e->stageLaunches[i] = 0; }
    e->stageFrames = 0;
    e->candTotal = e->kpTotal = 0;
    return ORBFE_OK;
}

int orbfe_get_profile(orbfe_extractor *e, orbfe_profile *out) {
    if (!e || !out) return fail(ORBFE_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(e->device));
    int rc = prof_resolve(e);
    if (rc) return rc;
    for (int i = 0; i < ORBFE_STAGE_COUNT; i++) { out->stage_ms[i] = e->stageMs[i]; out->stage_launches[i] = e->stageLaunches[i]; }
    out->frames = e->stageFrames;
    out->candidates = e->candTotal;
    out->keypoints = e->kpTotal;
    return ORBFE_OK;
}

int orbfe_extractor_destroy(orbfe_extractor *e) {
    if (!e) return ORBFE_OK;
    (void)hipSetDevice(e->device);
    if (e->stream) {
        (void)hipStreamSynchronize(e->stream);
        (void)hipStreamDestroy(e->stream);
    }
    if (e->stream2) {
        (void)hipStreamSynchronize(e->stream2);
        (void)hipStreamDestroy(e->stream2);
    }
    if (e->streamPyr) {
        (void)hipStreamSynchronize(e->streamPyr);
        (void)hipStreamDestroy(e->streamPyr);
    }
    for (hipEvent_t ev : {e->evFastDone, e->evBlurDone, e->evPyrDone, e->evDescDone[0], e->evDescDone[1]})
        if (ev) (void)hipEventDestroy(ev);
    for (auto &p : e->profPool) {
        for (auto &ev : p.ev)
            if (ev) (void)hipEventDestroy(ev);
        if (p.counts) (void)hipHostFree(p.counts);
    }
    for (hipStream_t st : {e->seqInStream, e->seqOutStream})
        if (st) {
            (void)hipStreamSynchronize(st);
            (void)hipStreamDestroy(st);
        }
    for (int b = 0; b < 2; b++)
        for (hipEvent_t ev : {e->seqEvIn[b], e->seqEvComp[b], e->seqEvOut[b]})
            if (ev) (void)hipEventDestroy(ev);
    for (DevBuf *b : {&e->dRsz, &e->dBlurTiles, &e->dFastTiles, &e->dDescMask, &e->dPyr, &e->dPyr2, &e->dBlur, &e->dCand, &e->dCounters, &e->dSel,
                      &e->dSelCount, &e->dImg, &e->dKps, &e->dDesc, &e->dCounts, &e->seqIn[0], &e->seqIn[1], &e->seqKps[0], &e->seqKps[1],
                      &e->seqDesc[0], &e->seqDesc[1], &e->seqCnt[0], &e->seqCnt[1]})
        b->release();
    for (PinBuf *b : {&e->hImg, &e->hCand, &e->hLevelCount, &e->hCellTab, &e->hSel, &e->hSelCount, &e->hKps, &e->hDesc, &e->hCounts, &e->hStatus, &e->hMirror})
        b->release();
    delete e;
    return ORBFE_OK;
}

int orbfe_get_tables(const orbfe_extractor *e, float *scale, float *inv_scale, float *sigma2, float *inv_sigma2,
                     int32_t *features_per_level, int32_t *umax) {
    if (!e) return fail(ORBFE_ERR_ARG, "null extractor");
    for (int i = 0; i < e->p.nlevels; i++) {
        if (scale) scale[i] = e->mvScaleFactor[i];
        if (inv_scale) inv_scale[i] = e->mvInvScaleFactor[i];
        if (sigma2) sigma2[i] = e->mvLevelSigma2[i];
        if (inv_sigma2) inv_sigma2[i] = e->mvInvLevelSigma2[i];
        if (features_per_level) features_per_level[i] = e->mnFeaturesPerLevel[i];
    }
    if (umax)
        for (int i = 0; i < 16; i++) umax[i] = e->umax[i];
    return ORBFE_OK;
}

#ifdef ORBFE_QT_TIMING
int orbfe_debug_qt_times(unsigned long long *out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(orbfe::g_qtTimes), sizeof(unsigned long long) * std::min(n, 256)) == hipSuccess ? 0 : -1;
}
#endif

void *orbfe_extractor_stream(orbfe_extractor *e) { return e ? (void *)e->stream : nullptr; }

int orbfe_debug_fast_variant(const orbfe_extractor *e) { return e ? e->fastVariant : -1; }


// after the stream has drained: fold the device status word into the handle.  The word is sticky on the device (batches
// enqueued back to back all report into it) and cleared here.  kRerunBigList: a FAST strip overflowed the normal survivor
// list; the handle has been switched to the large-list kernel and the caller runs the batch again.
static int eval_status(orbfe_extractor *e) {
    const int st = *(const int *)e->hStatus.p;
    if (st && getenv("ORBFE_DEBUG_STATUS")) fprintf(stderr, "[orbfe] status word 0x%x (FAST variant %d)\n", st, e->fastVariant);
    if (st) (void)hipMemsetAsync(e->pStatus, 0, sizeof(int), e->stream);
    *(int *)e->hStatus.p = 0;
    if (st & 1) e->overflow = true;
    if (st & 2) return fail(ORBFE_ERR_CAPACITY, "quadtree node table overflow");
    if (st & 4) {
        if (e->fastVariant >= 2) return fail(ORBFE_ERR_CAPACITY, "FAST survivor list overflowed its upper bound (internal error)");
        e->fastVariant++;
        if (e->fastVariant >= 2) {
            e->bigListBatches = 0;
            e->demoteAfter = e->promotions++ ? std::min(e->demoteAfter * 2, 4096) : 16;
        }
        return kRerunBigList;
    }
    if (e->overflow) return fail(ORBFE_ERR_CAPACITY, "FAST candidate buffer overflow on at least one level");
    return ORBFE_OK;
}

static int fetch_status(orbfe_extractor *e) {
    if (!e->pStatus || e->lastFrames == 0) return ORBFE_OK;
    HIPCHK(hipMemcpyAsync(e->hStatus.p, e->pStatus, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return eval_status(e);
}

int orbfe_synchronize(orbfe_extractor *e) {
    if (!e) return fail(ORBFE_ERR_ARG, "null extractor");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    int rc = fetch_status(e);
    if (rc == ORBFE_OK && e->rerunPending) rc = kRerunBigList;
    e->rerunPending = false;
    const std::vector<orbfe_extractor::BatchCall> pending = std::move(e->inFlight);
    e->inFlight.clear();
    e->batchesInFlight = 0;
    for (int attempt = 0; attempt < 2 && rc == kRerunBigList; attempt++) {
        // Asynchronous batches: their arguments are the caller's to keep alive until this call, so they can be run again.  The
        // status word is one per handle: every batch queued since the last synchronisation is run again, in order, on the next
        // kernel variant.
        if (pending.empty()) return fail(ORBFE_ERR_CAPACITY, "a FAST strip overflowed its survivor list and no batch is recorded to run again");
        for (const orbfe_extractor::BatchCall &c : pending) {
            rc = run_batch(e, c.imgs, c.F, c.w, c.h, c.stride, c.frameStride, c.kps, c.desc, c.counts, c.cap);
            if (rc) return rc;
        }
        HIPCHK(hipStreamSynchronize(e->stream));
        rc = fetch_status(e);
    }
    return rc;
}

int orbfe_set_quadtree_mode(orbfe_extractor *e, int mode) {
    if (!e) return fail(ORBFE_ERR_ARG, "null extractor");
    if (mode != ORBFE_QUADTREE_DEVICE && mode != ORBFE_QUADTREE_HOST) return fail(ORBFE_ERR_ARG, "unknown quadtree mode");
    e->quadtreeMode = mode;
    return ORBFE_OK;
}

int orbfe_extract_batch(orbfe_extractor *e, const uint8_t *dev_imgs, int n_frames, int w, int h, int stride,
                        size_t frame_stride, orbfe_keypoint *dev_kps, uint8_t *dev_desc, int32_t *dev_counts, int cap,
                        int sync) {
    if (!e || !dev_imgs || !dev_kps || !dev_desc || !dev_counts) return fail(ORBFE_ERR_ARG, "null argument");
    if (n_frames < 1 || w < 1 || h < 1 || cap < 1) return fail(ORBFE_ERR_ARG, "bad size");
    if (stride < w || (stride & 15) || (frame_stride & 15) || ((uintptr_t)dev_imgs & 15) ||
        frame_stride < (size_t)stride * h)
        return fail(ORBFE_ERR_ARG, "dev_imgs, stride and frame_stride must be multiples of 16 (stride >= w)");
    HIPCHK(hipSetDevice(e->device));
    // the record of queued calls (kept for the overflow rerun) is bounded: a caller that never synchronises gets a
    // synchronisation from the library every kMaxQueuedBatches calls, with the usual status handling
    if (e->inFlight.size() >= kMaxQueuedBatches) {
        const int rc0 = orbfe_synchronize(e);
        if (rc0) return rc0;
    }
    int rc = run_batch(e, dev_imgs, n_frames, w, h, stride, frame_stride, dev_kps, dev_desc, dev_counts, cap, sync == 0);
    if (rc) return rc;
    e->inFlight.push_back({dev_imgs, n_frames, w, h, stride, frame_stride, dev_kps, dev_desc, dev_counts, cap});
    e->batchesInFlight++;
    return sync ? orbfe_synchronize(e) : ORBFE_OK;
}

int orbfe_extract_sequence(orbfe_extractor *e, const uint8_t *host_imgs, int n_frames, int w, int h, int stride,
                           size_t frame_stride, orbfe_keypoint *host_kps, uint8_t *host_desc, int32_t *host_counts, int cap,
                           int chunk_frames) {
    if (!e || !host_imgs || !host_kps || !host_desc || !host_counts) return fail(ORBFE_ERR_ARG, "null argument");
    if (n_frames < 1 || w < 1 || h < 1 || cap < 1) return fail(ORBFE_ERR_ARG, "bad size");
    if (stride < w || (stride & 15) || (frame_stride & 15) || frame_stride < (size_t)stride * h)
        return fail(ORBFE_ERR_ARG, "stride and frame_stride must be multiples of 16 (stride >= w)");
    HIPCHK(hipSetDevice(e->device));
    if (e->batchesInFlight) {   // asynchronous batches of the caller come first, with their own status handling
        const int rc0 = orbfe_synchronize(e);
        if (rc0) return rc0;
    }
    const int chunk = std::min(n_frames, chunk_frames > 0 ? chunk_frames : 256);
    if (!e->seqInStream) {
        HIPCHK(hipStreamCreateWithFlags(&e->seqInStream, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&e->seqOutStream, hipStreamNonBlocking));
        for (int b = 0; b < 2; b++) {
            HIPCHK(hipEventCreateWithFlags(&e->seqEvIn[b], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&e->seqEvComp[b], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&e->seqEvOut[b], hipEventDisableTiming));
        }
    }
    for (int b = 0; b < 2; b++) {
        HIPCHK(e->seqIn[b].ensure((size_t)chunk * frame_stride + 256));
        HIPCHK(e->seqKps[b].ensure((size_t)chunk * cap * sizeof(orbfe_keypoint)));
        HIPCHK(e->seqDesc[b].ensure((size_t)chunk * cap * 32));
        HIPCHK(e->seqCnt[b].ensure((size_t)chunk * sizeof(int32_t)));
    }
    const int nChunks = (n_frames + chunk - 1) / chunk;
    for (int attempt = 0; attempt < 3; attempt++) {
        // chunk k: copy in on seqInStream, extract on the handle's stream, copy out on seqOutStream; buffers alternate, so the
        // copy-in of chunk k+1 and the copy-out of chunk k-1 run under the kernels of chunk k
        for (int k = 0; k < nChunks; k++) {
            const int b = k & 1, f0 = k * chunk, F = std::min(chunk, n_frames - f0);
            if (k >= 2) HIPCHK(hipStreamWaitEvent(e->seqInStream, e->seqEvComp[b], 0));    // input buffer b has been consumed
            HIPCHK(hipMemcpyAsync(e->seqIn[b].p, host_imgs + (size_t)f0 * frame_stride, (size_t)F * frame_stride, hipMemcpyHostToDevice,
                                  e->seqInStream));
            HIPCHK(hipEventRecord(e->seqEvIn[b], e->seqInStream));
            HIPCHK(hipStreamWaitEvent(e->stream, e->seqEvIn[b], 0));
            if (k >= 2) HIPCHK(hipStreamWaitEvent(e->stream, e->seqEvOut[b], 0));         // output buffers b have been copied out
            const int rc = run_batch(e, (const uint8_t *)e->seqIn[b].p, F, w, h, stride, frame_stride, (orbfe_keypoint *)e->seqKps[b].p,
                                     (uint8_t *)e->seqDesc[b].p, (int32_t *)e->seqCnt[b].p, cap);
            if (rc) return rc;
            HIPCHK(hipEventRecord(e->seqEvComp[b], e->stream));
            HIPCHK(hipStreamWaitEvent(e->seqOutStream, e->seqEvComp[b], 0));
            HIPCHK(hipMemcpyAsync(host_kps + (size_t)f0 * cap, e->seqKps[b].p, (size_t)F * cap * sizeof(orbfe_keypoint),
                                  hipMemcpyDeviceToHost, e->seqOutStream));
            HIPCHK(hipMemcpyAsync(host_desc + (size_t)f0 * cap * 32, e->seqDesc[b].p, (size_t)F * cap * 32, hipMemcpyDeviceToHost,
                                  e->seqOutStream));
            HIPCHK(hipMemcpyAsync(host_counts + f0, e->seqCnt[b].p, (size_t)F * sizeof(int32_t), hipMemcpyDeviceToHost, e->seqOutStream));
            HIPCHK(hipEventRecord(e->seqEvOut[b], e->seqOutStream));
        }
        HIPCHK(hipStreamSynchronize(e->seqOutStream));
        HIPCHK(hipStreamSynchronize(e->stream));
        const int rc = fetch_status(e);
        e->batchesInFlight = 0;
        if (rc != kRerunBigList) return rc;   // otherwise the whole sequence once more with the large FAST survivor list
    }
    return fail(ORBFE_ERR_CAPACITY, "FAST survivor list overflowed its upper bound (internal error)");
}

int orbfe_extract(orbfe_extractor *e, const uint8_t *img, int w, int h, int stride, orbfe_keypoint *kps, int cap,
                  uint8_t *desc32, int *n_out) {
    if (!e) return fail(ORBFE_ERR_ARG, "null extractor");
    if (n_out) *n_out = 0;
    if (!img || w <= 0 || h <= 0) return ORBFE_OK;   // _image.empty(): return with outputs untouched (:1046-1047)
    if (stride < w) return fail(ORBFE_ERR_ARG, "stride < w");
    HIPCHK(hipSetDevice(e->device));
    if (e->batchesInFlight) {   // asynchronous batches of the caller come first, with their own status handling
        const int rc0 = orbfe_synchronize(e);
        if (rc0) return rc0;
    }
    int rc = build_geometry(e, w, h);
    if (rc) return rc;
    const int pitch = round_up(w, kPitchAlign);
    const size_t fs = (size_t)pitch * h;
    const int icap = e->g.selFrameCap;
    HIPCHK(e->dImg.ensure(fs + 256));
    HIPCHK(e->dKps.ensure((size_t)icap * sizeof(KeyPointOut)));
    HIPCHK(e->dDesc.ensure((size_t)icap * 32));
    HIPCHK(e->dCounts.ensure(sizeof(int) * 4));
    HIPCHK(e->hKps.ensure((size_t)icap * sizeof(KeyPointOut)));
    HIPCHK(e->hDesc.ensure((size_t)icap * 32));
    HIPCHK(e->hCounts.ensure(sizeof(int) * 4));
    // the image goes through a pinned staging buffer in the device pitch: one DMA instead of a pitched copy from
    // pageable memory (which the runtime splits into one transfer per row)
    HIPCHK(e->hImg.ensure(fs));
    for (int y = 0; y < h; y++) memcpy((uint8_t *)e->hImg.p + (size_t)y * pitch, img + (size_t)y * stride, (size_t)w);
    HIPCHK(hipMemcpyAsync(e->dImg.p, e->hImg.p, fs, hipMemcpyHostToDevice, e->stream));
    int n = 0;
    for (int attempt = 0; attempt < 3; attempt++) {
        rc = run_batch(e, (const uint8_t *)e->dImg.p, 1, w, h, pitch, round_up((int)fs, 16), (orbfe_keypoint *)e->dKps.p,
                       (uint8_t *)e->dDesc.p, (int32_t *)e->dCounts.p, icap);
        if (rc) return rc;
        // count, keypoints and descriptors come back together (at most selFrameCap entries, ~130 KB): one synchronisation
        HIPCHK(hipMemcpyAsync(e->hCounts.p, e->dCounts.p, sizeof(int), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipMemcpyAsync(e->hKps.p, e->dKps.p, (size_t)icap * sizeof(KeyPointOut), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipMemcpyAsync(e->hDesc.p, e->dDesc.p, (size_t)icap * 32, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipMemcpyAsync(e->hStatus.p, e->pStatus, sizeof(int), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        n = *(int *)e->hCounts.p;
        rc = eval_status(e);
        if (rc != kRerunBigList) break;   // otherwise once more with the large survivor list (the image is still in dImg)
    }
    if (n_out) *n_out = n;
    if (rc) return rc;
    if (n > cap) return fail(ORBFE_ERR_CAPACITY, "keypoint buffer too small");
    if (n == 0) return ORBFE_OK;
    if (!kps || !desc32) return fail(ORBFE_ERR_ARG, "null output");
    memcpy(kps, e->hKps.p, (size_t)n * sizeof(KeyPointOut));
    memcpy(desc32, e->hDesc.p, (size_t)n * 32);
    return ORBFE_OK;
}

int orbfe_pyramid_level(orbfe_extractor *e, int frame, int level, uint8_t *dst, size_t dst_bytes, int *w, int *h) {
    if (!e || !e->haveGeom || e->lastFrames == 0) return fail(ORBFE_ERR_ARG, "no extraction has run on this handle");
    if (frame < 0 || frame >= e->lastFrames || level < 0 || level >= e->g.nlevels) return fail(ORBFE_ERR_ARG, "bad index");
    const LevelGeom &L = e->g.lv[level];
    if (w) *w = L.w;
    if (h) *h = L.h;
    if (!dst) return ORBFE_OK;
    const int bw = L.w + 2 * kEdge, bh = L.h + 2 * kEdge;
    if (dst_bytes < (size_t)bw * bh) return fail(ORBFE_ERR_CAPACITY, "destination too small");
    HIPCHK(hipSetDevice(e->device));
    int pitch;
    const uint8_t *src = level_ptr(e->last, e->g, frame, level, &pitch);
    uint8_t *roi = dst + (size_t)kEdge * bw + kEdge;
    HIPCHK(hipMemcpy2DAsync(roi, bw, src, pitch, L.w, L.h, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    reflect101_frame(dst, L.w, L.h);
    return ORBFE_OK;
}

int orbfe_pyramid_mirror(orbfe_extractor *e, int frame, uint8_t *const *dst, const size_t *dst_bytes, int nlevels) {
    if (!e || !e->haveGeom || e->lastFrames == 0) return fail(ORBFE_ERR_ARG, "no extraction has run on this handle");
    if (frame < 0 || frame >= e->lastFrames || !dst || !dst_bytes || nlevels != e->g.nlevels) return fail(ORBFE_ERR_ARG, "bad argument");
    const Geom &g = e->g;
    for (int l = 0; l < nlevels; l++)
        if (!dst[l] || dst_bytes[l] < (size_t)(g.lv[l].w + 2 * kEdge) * (g.lv[l].h + 2 * kEdge))
            return fail(ORBFE_ERR_CAPACITY, "destination too small");
    HIPCHK(hipSetDevice(e->device));
    // level 0 (the caller's frame) and the frame's block of levels 1.. come over in two copies into pinned memory, one wait
    const size_t l0Bytes = (size_t)e->last.pitch0 * g.h0, pyrBytes = nlevels > 1 ? (size_t)g.pyrFrameBytes : 0;
    const size_t l0Pad = (l0Bytes + 63) & ~(size_t)63;
    HIPCHK(e->hMirror.ensure(l0Pad + pyrBytes + 64));
    uint8_t *h0 = (uint8_t *)e->hMirror.p, *hp = h0 + l0Pad;
    HIPCHK(hipMemcpyAsync(h0, e->last.img0 + (size_t)frame * e->last.frameStride0, l0Bytes, hipMemcpyDeviceToHost, e->stream));
    if (pyrBytes)
        HIPCHK(hipMemcpyAsync(hp, e->last.pyr + (size_t)frame * g.pyrFrameBytes, pyrBytes, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int l = 0; l < nlevels; l++) {
        const LevelGeom &L = g.lv[l];
        const int bw = L.w + 2 * kEdge;
        const uint8_t *src = l == 0 ? h0 : hp + L.pyrOff;
        const int pitch = l == 0 ? e->last.pitch0 : L.pitch;
        uint8_t *roi = dst[l] + (size_t)kEdge * bw + kEdge;
        for (int y = 0; y < L.h; y++) memcpy(roi + (size_t)y * bw, src + (size_t)y * pitch, (size_t)L.w);
        reflect101_frame(dst[l], L.w, L.h);
    }
    return ORBFE_OK;
}

int orbfe_debug_stage(orbfe_extractor *e, int frame, int level, int what, void *dst, size_t dst_bytes, size_t *bytes) {
    if (!e || !e->haveGeom || e->lastFrames == 0) return fail(ORBFE_ERR_ARG, "no extraction has run on this handle");
    if (frame < 0 || frame >= e->lastFrames || level < 0 || level >= e->g.nlevels) return fail(ORBFE_ERR_ARG, "bad index");
    const Geom &g = e->g;
    const LevelGeom &L = g.lv[level];
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (what == 0 || what == 1) {
        const size_t need = (size_t)L.w * L.h;
        if (bytes) *bytes = need;
        if (!dst || dst_bytes < need) return ORBFE_OK;
        int pitch;
        const uint8_t *src;
        if (what == 0) {
            src = level_ptr(e->last, g, frame, level, &pitch);
        } else {
            src = e->last.blur + (size_t)frame * g.blurFrameBytes + L.blurOff;
            pitch = L.pitch;
        }
        HIPCHK(hipMemcpy2D(dst, L.w, src, pitch, L.w, L.h, hipMemcpyDeviceToHost));
        return ORBFE_OK;
    }
    if (what == 2) {
        // candidates in the reference's order: read the level's dense array + cell table back and walk the cells
        const int F = e->lastFrames;
        if (int rc = ensure_host_mirrors(e, F)) return rc;
        HIPCHK(hipMemcpy(e->hLevelCount.p, e->pLevelCount, (size_t)F * kMaxLevels * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(e->hCellTab.p, e->pCellTab, (size_t)F * g.totalCells * sizeof(int2), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy((Cand *)e->hCand.p + (size_t)frame * g.candFrameCap + L.candOff,
                         (const Cand *)e->dCand.p + (size_t)frame * g.candFrameCap + L.candOff, (size_t)L.candCap * sizeof(Cand),
                         hipMemcpyDeviceToHost));
        std::vector<Cand> c;
        ordered_candidates(e, frame, level, c);
        const size_t need = c.size() * 3 * sizeof(float);
        if (bytes) *bytes = need;
        if (!dst || dst_bytes < need) return ORBFE_OK;
        float *o = (float *)dst;
        for (size_t i = 0; i < c.size(); i++) {
            o[3 * i] = (float)c[i].x; o[3 * i + 1] = (float)c[i].y; o[3 * i + 2] = (float)c[i].resp;
        }
        return ORBFE_OK;
    }
    if (what == 3) {
        int n = 0;
        HIPCHK(hipMemcpy(&n, (const int *)e->dSelCount.p + frame * kMaxLevels + level, sizeof(int), hipMemcpyDeviceToHost));
        const size_t need = (size_t)n * 4 * sizeof(float);
        if (bytes) *bytes = need;
        if (!dst || dst_bytes < need) return ORBFE_OK;
        std::vector<SelKp> sel(n);
        if (n)
            HIPCHK(hipMemcpy(sel.data(), (const SelKp *)e->dSel.p + (size_t)frame * g.selFrameCap + L.selOff, (size_t)n * sizeof(SelKp),
                             hipMemcpyDeviceToHost));
        float *o = (float *)dst;
        for (int i = 0; i < n; i++) {
            o[4 * i] = (float)sel[i].x; o[4 * i + 1] = (float)sel[i].y; o[4 * i + 2] = (float)sel[i].resp; o[4 * i + 3] = -1.f;
        }
        return ORBFE_OK;
    }
    return fail(ORBFE_ERR_ARG, "unknown stage");
}

int orbfe_debug_rotation(orbfe_extractor *e, const float *angles_deg, int n, float *cos_sin) {
    if (!e || n < 0 || (n && (!angles_deg || !cos_sin))) return fail(ORBFE_ERR_ARG, "bad argument");
    if (n == 0) return ORBFE_OK;
    HIPCHK(hipSetDevice(e->device));
    DevBuf in, out;
    HIPCHK(in.ensure((size_t)n * sizeof(float)));
    if (hipError_t err = out.ensure((size_t)n * sizeof(float2))) {
        in.release();
        return fail(ORBFE_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(err));
    }
    hipError_t err = hipMemcpyAsync(in.p, angles_deg, (size_t)n * sizeof(float), hipMemcpyHostToDevice, e->stream);
    if (err == hipSuccess) {
        debug_rotation_kernel<<<(n + 255) / 256, 256, 0, e->stream>>>((const float *)in.p, n, (float2 *)out.p);
        err = hipGetLastError();
    }
    if (err == hipSuccess) err = hipMemcpyAsync(cos_sin, out.p, (size_t)n * sizeof(float2), hipMemcpyDeviceToHost, e->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
    in.release();
    out.release();
    if (err != hipSuccess) return fail(ORBFE_ERR_DEVICE, std::string("orbfe_debug_rotation: ") + hipGetErrorString(err));
    return ORBFE_OK;
}

// ------------------------------------------------------------------------------------------------ matcher ----------
int orbfe_hamming(const uint8_t *a32, const uint8_t *b32) {
    // ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1647-1663): popcount of the XOR of 8 dwords
    int d = 0;
    for (int i = 0; i < 4; i++) {
        uint64_t x, y;
        memcpy(&x, a32 + 8 * i, 8);
        memcpy(&y, b32 + 8 * i, 8);
        d += __builtin_popcountll(x ^ y);
    }
    return d;
}

}  // extern "C"

#include "matcher_host.inc"
