// k_describe_tiled.h -- K4 + K6 from LDS-staged tiles: the same arithmetic as orient_describe_kernel (k_describe.h), with
// the keypoints of a frame bucketed by image tile so that every patch row is fetched from L2 once per TILE instead of once
// per keypoint (gfx950).  OPT-IN (ORBFE_DESCRIBE_TILED=1): it cuts the L1 <- L2 requests from 89 M to 32 M per 512 frames, but
// not the time (0.655 ms against 0.676 ms at 512 frames per launch, 0.094 against 0.072 at 64; profiles/r2d_describe_tiled_ab.txt).
//
// orient_describe_kernel is bound by the path between L1 and L2: a keypoint's two patches (31 x 36 bytes of the level, 37 x 40
// bytes of the blurred level) touch ~87 128-byte lines of which 36-48 bytes each are used (11 KB fetched for 2.1 KB), and
// keypoints come in quadtree order, so nothing is shared.  With every patch resident the same kernel takes half the time
// (measured: 0.34 ms against 0.675 ms per 512 frames).  Here
//   1. kp_bucket_kernel (one workgroup per frame and level) counting-sorts the level's kept keypoints by tile of kDtW x kDtH
//      keypoint positions; an entry carries the keypoint and its slot in the output (the reference's order), the tile table
//      carries (first entry, count);
//   2. orient_describe_tiled_kernel: persistent workgroups walk the (frame, tile) jobs.  Per job the tile plus an 18-pixel
//      apron of the LEVEL plane is copied into LDS with row-coalesced 16-byte loads (issued while the previous job computes),
//      each wavefront sums the moments of its share of the tile's keypoints (dword-per-lane mapping, v_dot4 against a mask /
//      weight table), turns them into angle, cos, sin and KeyPoint records for all its keypoints at once, the SAME buffer is
//      refilled with the BLURRED plane (loads issued before the moments), and the 512 taps are read straight from it.
//      Barriers order LDS only (lds_barrier), so the prefetches stay in flight across them.
// Results go to the keypoint's own output slot, so the output order does not depend on the bucketing.
#pragma once
#include "k_describe.h"

namespace orbfe {

constexpr int kDtW = 128, kDtH = 64;                 // tile, in keypoint positions
constexpr int kDtApron = 18;                         // the rotated taps reach 18 pixels; the moments 15 (+3 of alignment)
constexpr int kDtRows = kDtH + 2 * kDtApron;         // staged rows
constexpr int kDtSteps = (kDtRows + 19) / 20;        // the workgroup stages 20 rows per step (5 per wave)
constexpr int kDtPieces = 12;                        // 16-byte pieces per staged row: 15 (alignment) + 18 + 128 + 22 <= 192
constexpr int kDtPitch = 16 * (kDtPieces + 1);       // LDS row pitch (208: rows fall on different banks)
constexpr int kDtGroupsPerCU = 4;                    // resident workgroups per CU (LDS: 16 KB masks + the tile + 3 KB)
constexpr int kDtMaxTiles = 1024;                    // tiles of one level the bucketing kernel's LDS tables hold

struct DescTileGeom {
    int tilesX[kMaxLevels];
    int tileBase[kMaxLevels + 1];   // first tile of a level in the frame-wide numbering; [nlevels] = tiles per frame
};

struct SortedKp {   // 16 bytes
    uint16_t x, y, resp, level;
    int32_t out;    // slot in the frame's output arrays = index in the reference's order
    int32_t pad;
};
static_assert(sizeof(SortedKp) == 16, "SortedKp");

struct BucketArgs {
    Geom g;
    DescTileGeom tg;
    const SelKp *sel;        // [F][selFrameCap]
    const int *selCount;     // [F][kMaxLevels]
    SortedKp *sorted;        // [F][selFrameCap], level segments at lv[l].selOff, tiles contiguous inside a level
    int2 *tileTab;           // [F][tiles per frame]: (first entry inside the frame's sorted block, count)
    int *counts;             // [F]
};

__global__ __launch_bounds__(256) void kp_bucket_kernel(BucketArgs a) {
    __shared__ int sHist[kDtMaxTiles];
    __shared__ int sCursor[kDtMaxTiles];
    __shared__ int sScan[256];
    const int l = blockIdx.x, f = blockIdx.y, tid = threadIdx.x;
    const LevelGeom &L = a.g.lv[l];
    const int n = min(a.selCount[f * kMaxLevels + l], L.selCap);
    int outBase = 0, total = 0;
    for (int q = 0; q < a.g.nlevels; q++) {
        const int nq = min(a.selCount[f * kMaxLevels + q], a.g.lv[q].selCap);
        if (q < l) outBase += nq;
        total += nq;
    }
    if (l == 0 && tid == 0) a.counts[f] = total;
    const int tX = a.tg.tilesX[l], nT = a.tg.tileBase[l + 1] - a.tg.tileBase[l];
    for (int t = tid; t < nT; t += 256) sHist[t] = 0;
    __syncthreads();
    const SelKp *sel = a.sel + (size_t)f * a.g.selFrameCap + L.selOff;
    auto tile_of = [&](const SelKp &kp) {
        const int cx = min(max((int)kp.x, kEdge), L.w - kEdge - 1), cy = min(max((int)kp.y, kEdge), L.h - kEdge - 1);
        return ((cy - kEdge) / kDtH) * tX + (cx - kEdge) / kDtW;
    };
    for (int i = tid; i < n; i += 256) atomicAdd(&sHist[tile_of(sel[i])], 1);
    __syncthreads();
    // exclusive prefix over the tiles: 4 per thread, then a block scan
    int loc[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int t = 4 * tid + k;
        loc[k] = t < nT ? sHist[t] : 0;
        sum += loc[k];
    }
    sScan[tid] = sum;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const int v = tid >= d ? sScan[tid - d] : 0;
        __syncthreads();
        sScan[tid] += v;
        __syncthreads();
    }
    int run = sScan[tid] - sum;
    int2 *tab = a.tileTab + (size_t)f * a.tg.tileBase[a.g.nlevels] + a.tg.tileBase[l];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int t = 4 * tid + k;
        if (t < nT) {
            sCursor[t] = run;
            tab[t] = make_int2(L.selOff + run, loc[k]);
        }
        run += loc[k];
    }
    __syncthreads();
    SortedKp *out = a.sorted + (size_t)f * a.g.selFrameCap + L.selOff;
    for (int i = tid; i < n; i += 256) {
        const SelKp kp = sel[i];
        const int pos = atomicAdd(&sCursor[tile_of(kp)], 1);
        SortedKp s;
        s.x = kp.x; s.y = kp.y; s.resp = kp.resp; s.level = (uint16_t)l;
        s.out = outBase + i;
        s.pad = 0;
        out[pos] = s;
    }
}

// Mask / weight table of the tiled kernel's moment sum.  The 31 rows x 9 dwords of the patch (36 bytes from the dword-aligned
// xa = (cx - 15) & ~3) are spread over the lanes dword by dword: item it = lane + 64 * slot (slot 0..4) is dword it % 9 of row
// it / 9, so that consecutive lanes read consecutive LDS dwords.  For patch alignment al = (cx - 15) & 3 the entry holds, for
// the dword's 4 bytes, 1 where the pixel lies inside the disc (|u| <= umax[|v|]) and u + 18 there, 0 elsewhere.
constexpr int kDtMomSlots = 5;
constexpr int kDtMaskTabU4 = 4 * kDtMomSlots * 64 / 2;   // the table in 16-byte units (640)
inline void desc_mask_table_dwords(const int *umax, uint2 *tab) {
    for (int al = 0; al < 4; al++)
        for (int slot = 0; slot < kDtMomSlots; slot++)
            for (int ln = 0; ln < 64; ln++) {
                const int it = ln + 64 * slot, row = it / 9, dc = it - row * 9, v = row - kHalfPatch;
                const int dmax = row < 31 ? umax[v < 0 ? -v : v] : -1;
                uint32_t m01 = 0, mU = 0;
                for (int b = 0; b < 4; b++) {
                    const int u = 4 * dc - kHalfPatch - al + b;
                    const bool ok = u >= -dmax && u <= dmax;
                    m01 |= (ok ? 1u : 0u) << (8 * b);
                    mU |= (ok ? (uint32_t)(u + 18) : 0u) << (8 * b);
                }
                tab[(al * kDtMomSlots + slot) * 64 + ln] = make_uint2(m01, mU);
            }
}

struct DescribeTiledArgs {
    Geom g;
    ImgSrc s;
    DescTileGeom tg;
    const SortedKp *sorted;
    const int2 *tileTab;
    KeyPointOut *kps;        // [F][cap]
    uint8_t *desc;           // [F][cap][32]
    int cap;
    int nFrames;
    const uint4 *maskTab;    // desc_mask_table_dwords()
};

// four dwords at a dword-aligned LDS address
__device__ __forceinline__ uint4 lds_u32x4(const uint8_t *p) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    struct __attribute__((packed, aligned(4))) V { u32x4 v; };
    const u32x4 v = reinterpret_cast<const V *>(p)->v;
    return make_uint4(v.x, v.y, v.z, v.w);
}

// Barrier that orders LDS traffic only: __syncthreads() also waits for every global load and store in flight (vmcnt(0)),
// which would drain the prefetches this kernel keeps outstanding across its phases.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

constexpr int kDtPass = 128;   // keypoints of a tile handled per staging pass (a tile with more is staged again)

// Persistent workgroups: gridDim.x (a multiple of 8) workgroups walk the (frame, tile) jobs; workgroup b serves the frames
// f = b mod 8 (mod 8), so a frame's tiles stay in one XCD's L2, and neighbouring workgroups take neighbouring tiles of one
// frame.  Per workgroup the mask table and the pattern are set up once.  Per job: the tile of the LEVEL plane is staged, the
// moments are summed (keypoint k by wave k mod 4), every thread turns one keypoint's moments into angle, cos, sin and its
// KeyPoint record, the SAME LDS buffer is refilled with the tile of the BLURRED plane, and the descriptors are read from it.
__global__ __launch_bounds__(256) void orient_describe_tiled_kernel(DescribeTiledArgs a) {
    __shared__ __align__(16) uint2 sMask[4][kDtMomSlots][64];
    __shared__ __align__(16) uint8_t sTile[kDtRows * kDtPitch];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tilesPerFrame = a.tg.tileBase[a.g.nlevels];
    for (int k = threadIdx.x; k < kDtMaskTabU4; k += 256) reinterpret_cast<uint4 *>(&sMask[0][0][0])[k] = a.maskTab[k];
    int mOfs[kDtMomSlots], mV[kDtMomSlots];   // the lane's dword inside the 31-row patch (LDS byte offset) and its row v
#pragma unroll
    for (int sl = 0; sl < kDtMomSlots; sl++) {
        const int it = lane + 64 * sl, row = it / 9, dc = it - row * 9;
        const bool in = row < 31;
        mOfs[sl] = in ? row * kDtPitch + 4 * dc : 0;   // items past the patch read a legal address and carry zero masks
        mV[sl] = in ? row - kHalfPatch : 0;
        asm volatile("" : "+v"(mOfs[sl]), "+v"(mV[sl]));
    }
    float px0[4], py0[4], px1[4], py1[4];
    {
        const uint32_t *pat32 = reinterpret_cast<const uint32_t *>(c_pattern);
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const uint32_t pp = pat32[lane * 4 + t];
            px0[t] = (float)(int8_t)(pp & 0xFF);
            py0[t] = (float)(int8_t)((pp >> 8) & 0xFF);
            px1[t] = (float)(int8_t)((pp >> 16) & 0xFF);
            py1[t] = (float)(int8_t)(pp >> 24);
        }
    }
    // staging: a wave copies 5 rows x 12 pieces per step (lanes 60..63 idle), the four waves 20 rows
    const int stSub = lane / kDtPieces, stPc = lane - stSub * kDtPieces;
    const bool stLane = lane < 5 * kDtPieces;

    const int xcd = blockIdx.x & 7, step = gridDim.x >> 3;
    // A job = up to kDtPass keypoints of one tile.  Everything a job needs before its first barrier (tile table entry, the
    // thread's keypoint, the thread's pieces of the level tile) is fetched while the PREVIOUS job computes its descriptors,
    // and the pieces of the blurred tile while the job sums its moments: no phase waits for global memory it asked for itself.
    struct Job {
        int j, k0, n, first, count, f, lvl, XA, Y0, pitchI, pitchB, hL, wL;
        const uint8_t *img, *blr;
    };
    // job index j (and keypoint offset k0 inside its tile) -> the job, with the tile's table entry `tab`
    auto make_job = [&](int j, int k0, int2 tab, Job &o) {
        o.j = j; o.k0 = k0;
        o.f = 8 * (j / tilesPerFrame) + xcd;
        const int tile = j % tilesPerFrame;
        o.first = tab.x; o.count = tab.y;
        o.n = min(kDtPass, tab.y - k0);
        int lvl = 0;
        for (int l = 1; l < a.g.nlevels; l++) lvl += tile >= a.tg.tileBase[l] ? 1 : 0;
        o.lvl = lvl;
        const LevelGeom &L = a.g.lv[lvl];
        const int tl = tile - a.tg.tileBase[lvl], tX = a.tg.tilesX[lvl];
        const int ty = tl / tX, tx = tl - ty * tX;
        o.XA = (kEdge + kDtW * tx - kDtApron) & ~15;
        o.Y0 = kEdge + kDtH * ty - kDtApron;   // >= 1
        o.img = level_ptr(a.s, a.g, o.f, lvl, &o.pitchI);
        o.blr = a.s.blur + (size_t)o.f * a.g.blurFrameBytes + L.blurOff;
        o.pitchB = L.pitch; o.hL = L.h; o.wL = L.w;
    };
    auto in_range = [&](int j) { return 8 * (j / tilesPerFrame) + xcd < a.nFrames; };
    auto load_tab = [&](int j) { return a.tileTab[(size_t)(8 * (j / tilesPerFrame) + xcd) * tilesPerFrame + j % tilesPerFrame]; };
    // the job after (j, k0, count): the rest of the same tile, else the next tile with keypoints (a synchronous look at the
    // tile table per empty tile: rare)
    auto next_job = [&](int j, int k0, int count, int2 tabAhead, Job &o) -> bool {
        if (k0 + kDtPass < count) { make_job(j, k0 + kDtPass, make_int2(o.first, count), o); return true; }
        j += step;
        if (!in_range(j)) return false;
        int2 tab = tabAhead;
        while (tab.y == 0) {
            j += step;
            if (!in_range(j)) return false;
            tab = load_tab(j);
        }
        make_job(j, 0, tab, o);
        return true;
    };
    // rows past the image repeat its last row and pieces past the row's pitch its last piece: no keypoint of the tile reads
    // them with a non-zero weight (a patch ends at w - 1, h - 1)
    // the thread's 16-byte pieces of a tile (one per staging step), in registers between their load and their LDS store
    typedef uint32_t dt_u32x4 __attribute__((ext_vector_type(4)));   // (HIP's uint4 in an array ends up in scratch)
    dt_u32x4 q[kDtSteps];
#define ORBFE_DT_LOADS(jb, blurred)                                                                                        \
    do {                                                                                                                   \
        const uint8_t *plane_ = (blurred) ? (jb).blr : (jb).img;                                                           \
        const int pitch_ = (blurred) ? (jb).pitchB : (jb).pitchI;                                                          \
        const int xo_ = min((jb).XA + 16 * stPc, pitch_ - 16), r_ = 5 * wv + stSub, yl_ = (jb).hL - 1 - (jb).Y0;          \
        _Pragma("unroll") for (int i_ = 0; i_ < kDtSteps; i_++) /* idle lanes and rows past the tile load a legal address */ \
            q[i_] = *reinterpret_cast<const dt_u32x4 *>(plane_ + (size_t)((jb).Y0 + min(r_ + 20 * i_, yl_)) * pitch_ + xo_);  \
    } while (0)
#define ORBFE_DT_STORES()                                                                                                  \
    do {                                                                                                                   \
        const int r_ = 5 * wv + stSub;                                                                                     \
        uint8_t *d_ = sTile + r_ * kDtPitch + 16 * stPc;                                                                   \
        _Pragma("unroll") for (int i_ = 0; i_ < kDtSteps; i_++)                                                            \
            if (stLane && r_ + 20 * i_ < kDtRows) *reinterpret_cast<dt_u32x4 *>(d_ + 20 * i_ * kDtPitch) = q[i_];             \
    } while (0)
    auto load_kp = [&](const Job &jb) {
        SortedKp kp = {};
        const int k = wv + 4 * lane;   // the wave's i-th keypoint goes to lane i
        if (k < jb.n) kp = a.sorted[(size_t)jb.f * a.g.selFrameCap + jb.first + jb.k0 + k];
        return kp;
    };

    Job cur;
    {
        int j = blockIdx.x >> 3;
        if (!in_range(j)) return;   // whole workgroup
        cur.first = 0;
        if (!next_job(j - step, 0, 0, load_tab(j), cur)) return;
    }
    ORBFE_DT_LOADS(cur, false);
    SortedKp kp = load_kp(cur);
    for (;;) {
        const int n = cur.n, lvl = cur.lvl;
        const LevelGeom &L = a.g.lv[lvl];
        const int nW = n > wv ? (n - wv + 3) >> 2 : 0;   // this wave's keypoints: wv, wv + 4, ...; the i-th sits in lane i
        lds_barrier();   // the previous job has finished with sTile
        ORBFE_DT_STORES();
        const int cxV = min(max((int)kp.x, kEdge), cur.wL - kEdge - 1), cyV = min(max((int)kp.y, kEdge), cur.hL - kEdge - 1);
        const int lxV = cxV - cur.XA, lyV = cyV - cur.Y0, outV = kp.out, respV = kp.resp;
        ORBFE_DT_LOADS(cur, true);   // the blurred tile: arrives while the moments are summed
        const bool moreTiles = in_range(cur.j + step);
        int2 tabAhead = make_int2(0, 0);
        if (moreTiles) tabAhead = load_tab(cur.j + step);
        lds_barrier();
        // ---- phase 1: moments of the wave's keypoints, one after the other; the sums land in the keypoint's lane
        int m10V = 0, m01V = 0;
        {
            auto moment_loads = [&](int i, uint32_t (&dw)[kDtMomSlots]) {
                const int lx = rl(lxV, i), ly = rl(lyV, i);
                const uint8_t *base = sTile + (ly - kHalfPatch) * kDtPitch + ((lx - kHalfPatch) & ~3);   // XA is a multiple of 16
#pragma unroll
                for (int sl = 0; sl < kDtMomSlots; sl++) dw[sl] = *reinterpret_cast<const uint32_t *>(base + mOfs[sl]);
            };
            uint32_t dw[kDtMomSlots], nx[kDtMomSlots];
            if (nW > 0) moment_loads(0, dw);
            for (int i = 0; i < nW; i++) {
                if (i + 1 < nW) moment_loads(i + 1, nx);
                const int al = (rl(lxV, i) - kHalfPatch) & 3;
                uint32_t A = 0;
                int rsSum = 0, m01 = 0;
#pragma unroll
                for (int sl = 0; sl < kDtMomSlots; sl++) {
                    const uint2 mk = sMask[al][sl][lane];
                    const uint32_t rs = __builtin_amdgcn_udot4(dw[sl], mk.x, 0u, false);
                    A = __builtin_amdgcn_udot4(dw[sl], mk.y, A, false);
                    rsSum += (int)rs;
                    m01 += __mul24(mV[sl], (int)rs);
                }
                const int m10 = (int)A - 18 * rsSum;
                const int s10 = wave_sum_uniform(m10), s01 = wave_sum_uniform(m01);
                m10V = lane == i ? s10 : m10V;
                m01V = lane == i ? s01 : m01V;
#pragma unroll
                for (int sl = 0; sl < kDtMomSlots; sl++) dw[sl] = nx[sl];
            }
        }
        lds_barrier();   // every wave has its moments; nobody reads the level tile any more
        ORBFE_DT_STORES();   // the blurred tile into the same buffer
        // ---- phase 2: one keypoint per lane
        const float angle = fast_atan2_deg((float)m01V, (float)m10V);
        const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
        float caV, sbV;
        sincosf_glibc(__fmul_rn(angle, factorPI), &sbV, &caV);
        if (lane < nW && outV < a.cap) {
            KeyPointOut o;
            o.x = (float)cxV;
            o.y = (float)cyV;
            if (lvl != 0) { o.x = __fmul_rn(o.x, L.scale); o.y = __fmul_rn(o.y, L.scale); }
            o.size = L.patch;
            o.angle = angle;
            o.response = (float)respV;
            o.octave = lvl;
            o.class_id = -1;
            a.kps[(size_t)cur.f * a.cap + outV] = o;
        }
        // ---- the next job's level tile and keypoints: on their way while this job's descriptors are read
        Job nxt = cur;
        const bool haveNext = next_job(cur.j, cur.k0, cur.count, tabAhead, nxt);
        const size_t descBase = (size_t)cur.f * a.cap;
        if (haveNext) {
            ORBFE_DT_LOADS(nxt, false);
            kp = load_kp(nxt);
        }
        lds_barrier();
        // ---- phase 3: rBRIEF, taps from the staged blurred tile
        const int caI = __builtin_bit_cast(int, caV), sbI = __builtin_bit_cast(int, sbV);
        for (int i = 0; i < nW; i++) {
            const float ca = __builtin_bit_cast(float, rl(caI, i)), sb = __builtin_bit_cast(float, rl(sbI, i));
            const f32x2 sc = {sb, ca}, cs = {ca, -sb};
            const int origin = rl(lyV, i) * kDtPitch + rl(lxV, i);
            const int originM = origin - 0x4B400000;
            uint32_t nib = 0;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const f32x2 kMagic = {12582912.f, 12582912.f};
                const f32x2 a0 = f32x2{px0[t], px0[t]} * sc, b0 = f32x2{py0[t], py0[t]} * cs;
                const f32x2 a1 = f32x2{px1[t], px1[t]} * sc, b1 = f32x2{py1[t], py1[t]} * cs;
                const f32x2 rc0 = (a0 + b0) + kMagic, rc1 = (a1 + b1) + kMagic;
                const float r0f = rc0.x, c0f = rc0.y, r1f = rc1.x, c1f = rc1.y;
                const int r0 = __builtin_bit_cast(int, r0f), c0 = __builtin_bit_cast(int, c0f);
                const int r1 = __builtin_bit_cast(int, r1f), c1 = __builtin_bit_cast(int, c1f);
                const int t0 = sTile[mad_i16(r0, kDtPitch, c0 + originM)];
                const int t1 = sTile[mad_i16(r1, kDtPitch, c1 + originM)];
                nib |= (uint32_t)(t0 < t1) << t;
            }
            uint32_t v = nib << (4 * (lane & 7));
            v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);
            v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);
            v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false);
            const int out = rl(outV, i);
            if (out < a.cap && (lane & 7) == 0)
                *reinterpret_cast<uint32_t *>(a.desc + (descBase + out) * 32 + (lane >> 3) * 4) = v;
        }
        if (!haveNext) break;
        cur = nxt;
    }
#undef ORBFE_DT_LOADS
#undef ORBFE_DT_STORES
}

}  // namespace orbfe
