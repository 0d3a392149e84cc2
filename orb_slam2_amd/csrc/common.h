// common.h -- shared host/device definitions of the orbfe library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace orbfe {

constexpr int kMaxLevels = 16;
constexpr int kEdge = 19;        // EDGE_THRESHOLD, reference src/ORBextractor.cc:74
constexpr int kMinBorder = 16;   // EDGE_THRESHOLD-3, :773
constexpr int kHalfPatch = 15;   // HALF_PATCH_SIZE, :73
constexpr int kPitchAlign = 64;  // row pitch of every workspace image, bytes
constexpr int kPitchSlack = 16;  // pitch >= w + kPitchSlack: the descriptor kernel's 16-byte window loads may run past w

// Geometry of one pyramid level for a given input size (host-built, passed to kernels by value).
struct LevelGeom {
    int w, h, pitch;        // level size; pitch of the workspace copy (levels >= 1) and of the blurred copy (all levels)
    int pyrOff;             // byte offset of the level inside one frame's pyramid block (level 0: unused)
    int blurOff;            // byte offset inside one frame's blurred block
    int nCols, nRows;       // FAST cell grid (reference src/ORBextractor.cc:784-787)
    int wCell, hCell;
    int wCellRcp;           // ceil(65536 / wCell): n / wCell == (n * wCellRcp) >> 16 for STRIP-RELATIVE n only (0 <= n < 264, the
                            // lanes of one wavefront; build_geometry checks every such n) -- not for absolute x (first miss: 62 @ 1115)
    int cellBase;           // first cell id of this level in the frame-wide cell numbering
    int candOff, candCap;   // candidate slots of this level inside one frame's candidate block
    int selOff, selCap;     // kept-keypoint slots of this level inside one frame's block
    int quota;              // mnFeaturesPerLevel[level]
    int nIni;               // quadtree roots (:543)
    float hX;               // root width (:545)
    float scale;            // mvScaleFactor[level]
    float patch;            // (float)(int)(31*scale), :837
    int rszOff;             // offset of this level's x table / y table in the resize-table arrays (level >= 1)
    int rszOffY;
    int fastCellsPerStrip;  // whole FAST cells one wavefront's strip covers (<= 8 cells, <= 62 dword columns)
};

struct Geom {
    int nlevels;
    int w0, h0;
    int totalCells;
    int pyrFrameBytes;      // levels 1..n-1
    int blurFrameBytes;     // levels 0..n-1
    int candFrameCap;       // candidates per frame (all levels)
    int selFrameCap;        // kept keypoints per frame (all levels)
    LevelGeom lv[kMaxLevels];
};

// FAST candidate: position relative to the level's minBorder (16,16), as the reference hands them to DistributeOctTree.
struct Cand {
    uint16_t x, y;
    uint16_t resp;   // FAST score - 1 (cv::KeyPoint::response)
    uint16_t pad;
};
static_assert(sizeof(Cand) == 8, "Cand");

// Keypoint kept by the quadtree, level coordinates including the border.
struct SelKp {
    uint16_t x, y;
    uint16_t resp;
    uint16_t level;
};
static_assert(sizeof(SelKp) == 8, "SelKp");

struct KeyPointOut {  // == orbfe_keypoint == cv::KeyPoint
    float x, y, size, angle, response;
    int32_t octave, class_id;
};
static_assert(sizeof(KeyPointOut) == 28, "KeyPointOut");

// Where level images live.  Level 0 is read in place from the caller's frames.
struct ImgSrc {
    const uint8_t *img0;    // frame 0 of the input batch
    int pitch0;
    size_t frameStride0;
    uint8_t *pyr;           // [F][pyrFrameBytes]
    uint8_t *blur;          // [F][blurFrameBytes]
};

__host__ __device__ inline const uint8_t *level_ptr(const ImgSrc &s, const Geom &g, int f, int l, int *pitch) {
    if (l == 0) {
        *pitch = s.pitch0;
        return s.img0 + (size_t)f * s.frameStride0;
    }
    *pitch = g.lv[l].pitch;
    return s.pyr + (size_t)f * g.pyrFrameBytes + g.lv[l].pyrOff;
}

}  // namespace orbfe
