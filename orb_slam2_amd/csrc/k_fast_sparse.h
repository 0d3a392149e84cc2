// k_fast_sparse.h -- K3, sparse form: the same result as fast_cells_kernel (k_fast.h) with the arc-score network run only
// where a corner is possible (gfx950).
//
// Why: the dense kernel spends 150 of its ~200 VALU instructions per wave-row on the exact score of EVERY pixel, and it is
// issue-bound (97 %).  A pixel can have S > t only if two ADJACENT compass points of its ring (k = 0, 4, 8, 12) are both
// brighter than v + t or both darker than v - t: every 9-arc of the 16-ring contains two compass points 4 apart.  With
//     max over adjacent pairs of min(c_i, c_i+4) = min(max(c0, c8), max(c4, c12))      (one of each axis; dually for max/min)
// that test is 11 packed instructions per pixel pair.  On textured frames 13 % of the pixels pass it at minThFAST = 7 (3 % do
// have S > 7); a pixel that fails has S <= minTh, is no corner at either threshold, and as a neighbour in the 3x3 test it can
// never beat a survivor (which has S > minTh) -- so its score may be taken as 0 and the result is unchanged.
//
// One wavefront owns the same tile as in the dense kernel (a cell row x a strip of whole cells, lanes = dwords, halo lanes,
// packed segments; FastTile) and walks down its rows:
//   per row   the lane's dword goes to an LDS ring of the last 16 image rows (u8, row = 64 dwords); the compass test runs in
//             pair form in registers on rows c-3, c, c+3 (neighbour columns by DPP); the passing pixels of the row are
//             appended, in x order, to an LDS queue of (row, x) entries (wave prefix sum by DPP);
//   per batch as soon as 128 entries wait, every lane takes two of them, gathers their 16 ring values + centre from the LDS
//             ring as bytes (the first 6 ring rows are mirrored behind the 16th, so one base address and immediate offsets
//             reach all 7 rows), packs them into the dense kernel's pair form and runs the SAME score function
//             (fast_arc_score_pk); scores above minTh go to a u8 score plane in LDS (ring of 16 rows, zero elsewhere) and to a
//             list of pending pixels;
//   NMS       a pending pixel is decided once the rows around it are complete: 8 byte reads from the score plane, the
//             neighbours across a cell edge masked (per-pixel edge table), strict '>' -- survivors become records of the form
//             the dense kernel writes, in row-major order, and the tile ends in the same fast_flush.
// Entries wait at most 8 rows (a forced drain keeps ring and plane rows alive for everything pending).  Every list is bounded;
// a tile whose pending list or record list overflows (dense noise) sets status bit 2 and the host runs the batch again with
// the dense kernel, as it already does for the dense kernel's own list overflow.
#pragma once
#include "k_fast.h"

namespace orbfe {

constexpr int kFsRingRows = 16, kFsRingMirror = 6;
constexpr int kFsRingBytes = (kFsRingRows + kFsRingMirror) * 256;
constexpr int kFsPlaneBytes = 16 * 256;
constexpr int kFsQueueCap = 512;     // (row, x) entries of 16 bits waiting for their score
constexpr int kFsPendCap = 256;      // scored pixels above minTh waiting for the rows around them
constexpr int kFsListCap = 512;      // survivor records of a tile (textured frames: ~150)
constexpr int kFsMaxWait = 8;        // rows an entry may wait (ring: rows c-9 .. c+6 of the tile are alive at row c)
constexpr int kFsWaves = 2;         // wavefronts per workgroup (16.5 KB of LDS each)
constexpr int kFsMaxRows = 62;       // rows of a tile the 6-bit row field and the list bounds are good for

__device__ __forceinline__ int lanes_below(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
}

// compass test of a pixel pair (plain 0..255 in 16-bit lanes): non-zero half <=> S > t is possible for that pixel
__device__ __forceinline__ uint32_t fast_compass_pk(u16x2 v, u16x2 c0, u16x2 c8, u16x2 c4, u16x2 c12, u16x2 t) {
    const u16x2 pmin = pkmin(pkmax(c0, c8), pkmax(c4, c12));     // best "both brighter" pair
    const u16x2 pmax = pkmax(pkmin(c0, c8), pkmin(c4, c12));     // best "both darker" pair
    return as_u32(pksubsat(pmin, v + t)) | as_u32(pksubsat(pksubsat(v, t), pmax));
}

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void fast_cells_sparse_kernel(FastArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t sRing[WAVES][kFsRingBytes];
    __shared__ __attribute__((aligned(16))) uint8_t sPlane[WAVES][kFsPlaneBytes];
    __shared__ uint32_t sQueue[WAVES][kFsQueueCap / 2];
    __shared__ uint32_t sPend[WAVES][kFsPendCap];
    __shared__ uint2 sList[WAVES][kFsListCap];
    __shared__ uint32_t sTag[WAVES][64];
    __shared__ uint32_t sEdge[WAVES][64];
    __shared__ int sCellMax[WAVES][2 * kFastMaxCells];
    __shared__ int4 sSeg[WAVES][kFastMaxSeg];
    const int lane = threadIdx.x & 63;
    const int wvi = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ti = blockIdx.x * WAVES + wvi;
    if (ti >= a.nTiles) return;
    const FastTile &t = a.tiles[ti];
    const int f = blockIdx.y;
    const int level = t.level, nSeg = t.nSeg;
    const LevelGeom &L = a.g.lv[level];
    const int w = L.w, h = L.h;
    int pitch;
    const uint8_t *img = level_ptr(a.s, a.g, f, level, &pitch);
    // ---- the lane's place in the tile: as in fast_cells_kernel
    int Xs = kEdge, Xe = kEdge, Y0 = kEdge, segLane0 = 0, nCellsTotal = 0, segCell0 = 0;
    for (int sgi = 0; sgi < nSeg; sgi++) {
        const FastSeg sg = t.seg[sgi];
        const int sXs = kEdge + sg.c0 * L.wCell, sXe = min(kEdge + (sg.c0 + sg.nCells) * L.wCell, w - kEdge);
        const bool mine = lane >= sg.lane0 && lane < sg.lane0 + sg.nLanes;
        if (mine || sgi == 0) {
            Xs = sXs; Y0 = kEdge + sg.cellRow * L.hCell; segLane0 = sg.lane0;
            Xe = mine ? sXe : sXs;
        }
        if (mine) segCell0 = nCellsTotal;
        if (lane == sgi) sSeg[wvi][sgi] = make_int4(sg.lane0, nCellsTotal, 3 + sg.c0 * L.wCell, 3 + sg.cellRow * L.hCell);
        nCellsTotal += sg.nCells;
    }
    const int dh = min(kEdge + t.seg[0].cellRow * L.hCell + L.hCell, h - kEdge) - (kEdge + t.seg[0].cellRow * L.hCell);
    if (dh <= 0) return;
    if (dh > kFsMaxRows) {   // taller cells than the row field holds: the dense kernel takes the batch
        if (lane == 0) atomicOr(a.status, 4);
        return;
    }
    const int x0 = Xs - 4 + 4 * (lane - segLane0);
    const int xl = min(max(x0, 0), pitch - 4);
    const uint32_t rowOff = (uint32_t)((Y0 - 3) * pitch + xl);
    uint32_t own01 = 0, recTag = (uint32_t)lane, edge = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const int x = x0 + b, rel = x - kEdge, relS = x - Xs;
        own01 |= (x >= Xs && x < Xe) ? (1u << (8 * b)) : 0u;
        int cis = 0;
#pragma unroll
        for (int c = 1; c < kFastMaxCells; c++) cis += relS >= c * L.wCell ? 1 : 0;
        recTag |= (uint32_t)min(segCell0 + cis, kFastMaxCells - 1) << (6 + 3 * b);
        const int cc = rel >= 0 ? rel % L.wCell : 0;
        const bool leftIn = rel > 0 && cc != 0, rightIn = rel >= 0 && cc != L.wCell - 1 && x != w - kEdge - 1;
        edge |= ((leftIn ? 1u : 0u) | (rightIn ? 2u : 0u)) << (8 * b);
    }
    uint32_t *ring32 = reinterpret_cast<uint32_t *>(sRing[wvi]);
    uint32_t *plane32 = reinterpret_cast<uint32_t *>(sPlane[wvi]);
    uint8_t *ring8 = sRing[wvi], *plane8 = sPlane[wvi];
    typedef uint16_t __attribute__((may_alias)) u16_alias;
    u16_alias *queue16 = reinterpret_cast<u16_alias *>(sQueue[wvi]);
    sTag[wvi][lane] = recTag;
    sEdge[wvi][lane] = edge;
    if (lane < 2 * kFastMaxCells) sCellMax[wvi][lane] = 0;
    plane32[lane] = 0;          // row -1 (slot 0): above the cell row
    plane32[64 + lane] = 0;     // row 0
    // rows Y0-3 .. Y0+2 of the image = tile rows j = 0..5 (always inside the image), ring slot j and its mirror 16 + j
#pragma unroll
    for (int u = 0; u < 6; u++) {
        const uint32_t raw = *reinterpret_cast<const uint32_t *>(img + (size_t)u * pitch + rowOff);
        ring32[u * 64 + lane] = raw;
        ring32[(kFsRingRows + u) * 64 + lane] = raw;
    }
    uint32_t rawNext = *reinterpret_cast<const uint32_t *>(img + (size_t)6 * pitch + rowOff);
    const u16x2 thPair = as_pk((uint32_t)a.minTh | ((uint32_t)a.minTh << 16));
    const u16x2 one = as_pk(0x00010001u);
    const int minTh = a.minTh;

    // wave-uniform state
    int qHead = 0, qTail = 0, qFirstRow = 0;      // queue: entries [qHead, qTail), qHead even; row of the oldest (lower bound)
    int pHead = 0, pTail = 0, pFirstRow = 0;      // pending pixels [pHead, pTail) (indices taken mod kFsPendCap)
    int listCount = 0;
    bool overflow = false;

    for (int c = 0; c <= dh; c++) {               // c == dh: no row, only the final drain
        bool drainAll = c == dh;
        if (c < dh) {
            const uint32_t rD = rawNext;          // image row of tile row j = c + 6 (detection row c + 3)
            if (c + 1 < dh) rawNext = *reinterpret_cast<const uint32_t *>(img + (size_t)(7 + c) * pitch + rowOff);
            const int jd = (c + 6) & (kFsRingRows - 1);
            ring32[jd * 64 + lane] = rD;
            if (jd < kFsRingMirror) ring32[(kFsRingRows + jd) * 64 + lane] = rD;
            plane32[((c + 2) & 15) * 64 + lane] = 0;   // the plane row of detection row c + 1
            const uint32_t rC = ring32[((c + 3) & (kFsRingRows - 1)) * 64 + lane], rU = ring32[(c & (kFsRingRows - 1)) * 64 + lane];
            // pair form without bias: (p0, p2) and (p1, p3)
            const uint32_t eC = rC & 0x00FF00FFu, oC = __builtin_amdgcn_perm(0u, rC, 0x0c030c01u);
            const uint32_t eU = rU & 0x00FF00FFu, oU = __builtin_amdgcn_perm(0u, rU, 0x0c030c01u);
            const uint32_t eD = rD & 0x00FF00FFu, oD = __builtin_amdgcn_perm(0u, rD, 0x0c030c01u);
            const uint32_t eL = dpp_from_left(eC), oL = dpp_from_left(oC), eR = dpp_from_right(eC), oR = dpp_from_right(oC);
            // even pair (p0, p2): x-3 = (p-3, p-1) = oL, x+3 = (p3, p5);  odd pair (p1, p3): x-3 = (p-2, p0), x+3 = (p4, p6) = eR
            const uint32_t dE = fast_compass_pk(as_pk(eC), as_pk(eD), as_pk(eU), as_pk(__builtin_amdgcn_alignbit(oR, oC, 16)), as_pk(oL), thPair);
            const uint32_t dO = fast_compass_pk(as_pk(oC), as_pk(oD), as_pk(oU), as_pk(eR), as_pk(__builtin_amdgcn_alignbit(eC, eL, 16)), thPair);
            // bytes of 0 / 1 in pixel order, own detection pixels only
            const uint32_t tb = (as_u32(pkmin(as_pk(dE), one)) | (as_u32(pkmin(as_pk(dO), one)) << 8)) & own01;
            if (__ballot(tb != 0) != 0) {   // wave-uniform
                const int k = __builtin_popcount(tb);
                const int incl = wave_incl_scan(k);
                const int total = __builtin_amdgcn_readlane(incl, 63);
                const uint32_t pre = (tb << 8) + (tb << 16) + (tb << 24);   // byte b: passing pixels of this dword before b
                const int base = qTail + incl - k;
                const uint32_t ent = ((uint32_t)c << 8) | (uint32_t)(4 * lane);
#pragma unroll
                for (int b = 0; b < 4; b++)
                    if ((tb >> (8 * b)) & 1u) queue16[(base + (int)((pre >> (8 * b)) & 0xFFu)) & (kFsQueueCap - 1)] = (uint16_t)(ent + b);
                if (qTail == qHead) qFirstRow = c;
                qTail += total;
            }
            drainAll = (qTail > qHead && qFirstRow <= c - kFsMaxWait) || (pTail > pHead && pFirstRow <= c - kFsMaxWait);
        }
        // ---- score what waits: full batches of 128, or everything at a deadline / at the end
        bool scored = false;
        int lastRow = 0;
        while (qTail - qHead >= 128 || (drainAll && qTail > qHead)) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int n = min(128, qTail - qHead);
            uint32_t e32 = sQueue[wvi][((qHead >> 1) + lane) & (kFsQueueCap / 2 - 1)];
            const bool vLo = 2 * lane < n, vHi = 2 * lane + 1 < n;
            uint32_t eLo = e32 & 0xFFFFu, eHi = e32 >> 16;
            eLo = vLo ? eLo : 4u;     // lanes without an entry gather somewhere legal
            eHi = vHi ? eHi : 4u;
            // ring offset of (row - 3, x - 3): tile row j = row is image row (detection row - 3)
            const uint8_t *gLo = ring8 + (eLo & 0xFFFu) - 3, *gHi = ring8 + (eHi & 0xFFFu) - 3;
            // ring k -> (dx + 3) + 256 * (dy + 3), dy > 0 = down
            constexpr int kOff[16] = {3 + 256 * 6, 4 + 256 * 6, 5 + 256 * 5, 6 + 256 * 4, 6 + 256 * 3, 6 + 256 * 2, 5 + 256 * 1, 4 + 256 * 0,
                                      3 + 256 * 0, 2 + 256 * 0, 1 + 256 * 1, 0 + 256 * 2, 0 + 256 * 3, 0 + 256 * 4, 1 + 256 * 5, 2 + 256 * 6};
            u16x2 r[16];
#pragma unroll
            for (int k = 0; k < 16; k++)
                r[k] = as_pk(((uint32_t)gLo[kOff[k]] | ((uint32_t)gHi[kOff[k]] << 16)) | 0x40004000u);
            const u16x2 v = as_pk(((uint32_t)gLo[3 + 256 * 3] | ((uint32_t)gHi[3 + 256 * 3] << 16)) | 0x40004000u);
            const uint32_t S = as_u32(fast_arc_score_pk(v, r));
            const int sLo = (int)(S & 0xFFFFu), sHi = (int)(S >> 16);
            const bool kLo = vLo && sLo > minTh, kHi = vHi && sHi > minTh;
            // the plane row of detection row r is slot (r + 1) & 15
            if (kLo) plane8[(eLo + 0x100u) & 0xFFFu] = (uint8_t)sLo;
            if (kHi) plane8[(eHi + 0x100u) & 0xFFFu] = (uint8_t)sHi;
            const unsigned long long bLo = __ballot(kLo), bHi = __ballot(kHi);
            const int add = __popcll(bLo) + __popcll(bHi);
            if (add) {
                const int pos = pTail + lanes_below(bLo) + lanes_below(bHi);
                if (kLo) sPend[wvi][pos & (kFsPendCap - 1)] = ((uint32_t)sLo << 16) | eLo;
                if (kHi) sPend[wvi][(pos + (kLo ? 1 : 0)) & (kFsPendCap - 1)] = ((uint32_t)sHi << 16) | eHi;
                if (pTail == pHead) pFirstRow = (int)((__builtin_amdgcn_readfirstlane(e32) >> 8) & 63u);
                pTail += add;
                overflow = overflow || (pTail - pHead > kFsPendCap);
            }
            {
                const uint32_t eLast = (uint32_t)__builtin_amdgcn_readlane((int)e32, (n - 1) >> 1);
                lastRow = (int)((((n - 1) & 1) ? eLast >> 24 : eLast >> 8) & 63u);
            }
            qHead += n;
            if (qHead == qTail) qHead = qTail = 0;
            else qFirstRow = lastRow;
            scored = true;
        }
        // ---- 3x3 test of the pending pixels whose surroundings are complete
        if ((scored || drainAll) && pTail > pHead) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // every detection row <= through has all its scores in the plane
            const int through = qTail > qHead ? lastRow - 1 : (c < dh ? c : dh);
            const int lim = through - 1;   // rows whose lower neighbours are complete too (row dh stays zero: below the cell row)
            while (pTail > pHead) {
                const int i = pHead + lane;
                const uint32_t pe = sPend[wvi][i & (kFsPendCap - 1)];
                const int row = (int)((pe >> 8) & 63u), xp = (int)(pe & 0xFFu), sc = (int)(pe >> 16);
                const bool ok = i < pTail && row <= lim;
                const unsigned long long bOk = __ballot(ok);
                if (bOk == 0) break;
                // plane rows row-1, row, row+1 = slots row, row+1, row+2 (mod 16)
                const uint8_t *pu = plane8 + ((row & 15) << 8) + xp, *pc = plane8 + (((row + 1) & 15) << 8) + xp,
                              *pd = plane8 + (((row + 2) & 15) << 8) + xp;
                const int ed = (int)(reinterpret_cast<const uint8_t *>(sEdge[wvi])[xp]);
                int nl = max(max((int)pu[-1], (int)pc[-1]), (int)pd[-1]);
                int nr = max(max((int)pu[1], (int)pc[1]), (int)pd[1]);
                const int nm = max((int)pu[0], (int)pd[0]);
                nl = (ed & 1) ? nl : 0;
                nr = (ed & 2) ? nr : 0;
                const bool keep = ok && sc > max(max(nl, nr), nm);
                const unsigned long long bK = __ballot(keep);
                if (bK != 0) {
                    const int pos = listCount + lanes_below(bK);
                    if (keep && pos < kFsListCap)
                        sList[wvi][pos] = make_uint2((uint32_t)sc << (8 * (xp & 3)), sTag[wvi][xp >> 2] | ((uint32_t)row << 18));
                    listCount += __popcll(bK);
                }
                const int done = __popcll(bOk);
                pHead += done;
                if (done < 64) break;
            }
            if (pTail > pHead) pFirstRow = max(lim + 1, 0);
            else pHead = pTail = 0;
        }
    }
    if (overflow) listCount = kFsListCap + 1;   // reported by fast_flush as a list overflow: the host takes the dense kernel
    fast_flush<kFsListCap>(a, t, L, f, level, nSeg, nCellsTotal, lane, listCount, sList[wvi], sCellMax[wvi], sSeg[wvi]);
}

}  // namespace orbfe
