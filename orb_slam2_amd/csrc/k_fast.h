// k_fast.h -- K3: per-cell FAST-9/16 with the iniThFAST -> minThFAST fallback, 3x3 NMS and ordered emission (gfx950).
//
// Replaces the cell loop of ORBextractor::ComputeKeyPointsOctTree (reference src/ORBextractor.cc:771-829) and the two
// cv::FAST(cell, kps, th, true) calls inside it (:809-816; OpenCV FAST_t<16>).
//
// Facts used (checked against the literal OpenCV algorithm in tests/test_oracle_kat.py):
//   * arc score  S = max over the 16 nine-arcs A of max(min_{k in A}(v - ring_k), min_{k in A}(ring_k - v))
//                  = max(v - min_A max_A ring, max_A min_A ring - v);   corner at threshold t <=> S > t;  response = S - 1;
//   * cv::FAST's NMS keeps p iff its score is strictly greater than its 8 neighbours' scores, where a neighbour that is
//     no corner at t or lies outside the cell's scan range counts 0.  A survivor has S > t, so comparing against RAW
//     neighbour scores gives the same answer: survivors(t) = {strict 3x3 local maxima inside the cell with S > t} for
//     every t.  One score pass therefore serves both thresholds, and "re-run with minThFAST iff the post-NMS list is
//     empty" (:812-816) becomes: threshold = (any survivor with S > iniTh) ? iniTh : minTh, per cell.
//
// One kernel, fast_cells_kernel, register-sliding like the blur: one wavefront owns (one cell row) x (a strip of whole
// cells, <= 56 dword columns) of one level and walks down the rows.  Per row each lane loads ONE aligned dword (coalesced
// 256-B segments) and keeps the last 7 rows; left/right neighbour dwords come by DPP wave shifts.  The 16 ring values of
// the dword's four pixels are cut out as byte windows (v_alignbyte) and processed two pixels at a time in packed 16-bit
// lanes: the sixteen 9-windows' maxima/minima come from prefix/suffix maxima of the two ring halves (28 + 16
// v_pk_max_u16 instead of 64), then a 15-op tree.  The strict 3x3 local-maximum test runs in the same registers (cell
// edges masked).  Survivors above minTh (about 1 % of the pixels) are appended, in row-major order, to a per-wave LDS
// list together with a per-cell running maximum.  Because a wave owns whole cells, the threshold decision is wave-local:
// at the end of the strip the wave filters its list per cell, reserves space in the level's dense candidate array with
// ONE atomicAdd and writes the candidates grouped by cell in cv::FAST's order; the cell's (offset, count) goes to a cell
// table so a consumer can walk cells in the reference's order.  No barriers, no score plane in HBM.
#pragma once
#include "common.h"

namespace orbfe {

constexpr int kFastStripMaxDw = 60;   // writer lanes 2..61; lanes 0,1 and 62,63 are halo

struct FastTile { uint16_t level, strip, cellRow, pad; };

constexpr int kFastListCap = 2048;    // survivors a wave can hold for its strip (dense texture: ~150)

struct FastArgs {
    Geom g;
    ImgSrc s;
    const FastTile *tiles;
    int nTiles;
    Cand *cand;              // [F][candFrameCap]
    int *levelCount;         // [F][kMaxLevels]  (zeroed before launch)
    int2 *cellTab;           // [F][totalCells]  (offset inside the level's array, count; zeroed before launch)
    int *status;             // bit 2: a wave's survivor list overflowed
    int iniTh, minTh;
};

typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u16x2 as_pk(uint32_t v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ uint32_t as_u32(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ u16x2 pkmax(u16x2 a, u16x2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ u16x2 pkmin(u16x2 a, u16x2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ u16x2 pksubsat(u16x2 a, u16x2 b) { return __builtin_elementwise_sub_sat(a, b); }

// even pixels (bytes 0,2) / odd pixels (bytes 1,3) of a 4-pixel byte window as two 16-bit lanes
__device__ __forceinline__ u16x2 pair_even(uint32_t w) { return as_pk(w & 0x00FF00FFu); }
__device__ __forceinline__ u16x2 pair_odd(uint32_t w) { return as_pk(__builtin_amdgcn_perm(0u, w, 0x0c030c01u)); }

// arc score of two pixels at once: v = centres, r[k] = ring values, all in 16-bit lanes holding 0..255
__device__ __forceinline__ u16x2 fast_arc_score_pk(u16x2 v, const u16x2 (&r)[16]) {
    u16x2 preMx[16], sufMx[16], preMn[16], sufMn[16];
#pragma unroll
    for (int b = 0; b < 16; b += 8) {
        preMx[b] = r[b];
        preMn[b] = r[b];
        sufMx[b + 7] = r[b + 7];
        sufMn[b + 7] = r[b + 7];
#pragma unroll
        for (int j = 1; j < 8; j++) {
            preMx[b + j] = pkmax(preMx[b + j - 1], r[b + j]);
            preMn[b + j] = pkmin(preMn[b + j - 1], r[b + j]);
            sufMx[b + 7 - j] = pkmax(sufMx[b + 8 - j], r[b + 7 - j]);
            sufMn[b + 7 - j] = pkmin(sufMn[b + 8 - j], r[b + 7 - j]);
        }
    }
    // window of 9 starting at k (k < 8): ring[k..7] u ring[8..8+k];  starting at 8+k: ring[8+k..15] u ring[0..k]
    u16x2 minOfMax = as_pk(0x00FF00FFu), maxOfMin = as_pk(0u);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        minOfMax = pkmin(minOfMax, pkmax(sufMx[k], preMx[8 + k]));
        minOfMax = pkmin(minOfMax, pkmax(sufMx[8 + k], preMx[k]));
        maxOfMin = pkmax(maxOfMin, pkmin(sufMn[k], preMn[8 + k]));
        maxOfMin = pkmax(maxOfMin, pkmin(sufMn[8 + k], preMn[k]));
    }
    return pkmax(pksubsat(v, minOfMax), pksubsat(maxOfMin, v));   // max(S, 0)
}

__device__ __forceinline__ uint32_t dpp_from_left(uint32_t v) {    // lane i <- lane i-1 (0 into lane 0)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t dpp_from_right(uint32_t v) {   // lane i <- lane i+1 (0 into lane 63)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, false);
}

struct FastRowS {      // one row of scores in pair form, with what the 3x3 test needs from it
    u16x2 se, so;      // scores of pixels (0,2) and (1,3)
    u16x2 he, ho;      // max of the left and right neighbours in this row (cell edges masked)
    u16x2 me, mo;      // max of left, centre, right in this row (what the rows above/below see)
};

__global__ __launch_bounds__(256) void fast_cells_kernel(FastArgs a) {
    __shared__ uint32_t sList[4][kFastListCap];
    __shared__ int sCellMax[4][8];
    const int lane = threadIdx.x & 63;
    const int ti = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ti >= a.nTiles) return;
    const FastTile t = a.tiles[ti];
    const int f = blockIdx.y;
    const LevelGeom &L = a.g.lv[t.level];
    const int w = L.w, h = L.h;
    const int Y0 = kEdge + t.cellRow * L.hCell;
    const int dh = min(Y0 + L.hCell, h - kEdge) - Y0;
    if (dh <= 0) return;
    int pitch;
    const uint8_t *img = level_ptr(a.s, a.g, f, t.level, &pitch);
    // strip = cells [c0, c1) of this cell row; lane 2 holds the dword with the strip's first detection pixel
    const int wvi = threadIdx.x >> 6;
    const int c0 = t.strip * L.fastCellsPerStrip, c1 = min(c0 + L.fastCellsPerStrip, L.nCols);
    const int Xs = kEdge + c0 * L.wCell, Xe = min(kEdge + c1 * L.wCell, w - kEdge);
    if (Xe <= Xs) return;
    const int d = (Xs >> 2) + lane - 2;
    const int x0 = 4 * d;
    const int xl = min(max(x0, 0), pitch - 4);   // halo lanes beyond the row: any legal address, value unused
    // per byte: is the pixel owned by this strip, and its (cell in strip, x in cell) packed for the survivor record
    uint32_t ownMask = 0, recBase[4];
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const int x = x0 + b;
        const bool own = x >= Xs && x < Xe;
        const int cis = own ? (x - Xs) / L.wCell : 0;
        recBase[b] = ((uint32_t)cis << 8) | ((uint32_t)(own ? x - Xs - cis * L.wCell : 0) << 11);
        ownMask |= own ? (0xFFu << (8 * b)) : 0u;
    }
    if (lane < 8) sCellMax[wvi][lane] = 0;
    int listCount = 0;   // wave-uniform
    // per-lane cell-edge masks (16-bit lanes of 0xFFFF where the neighbour on that side is inside the pixel's cell)
    uint32_t mLe, mLo, mRe, mRo;
    {
        uint32_t ml[4], mr[4];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int x = x0 + b, rel = x - kEdge;
            const int c = rel >= 0 ? rel % L.wCell : 0;
            ml[b] = (rel > 0 && c != 0) ? 0xFFFFu : 0u;
            mr[b] = (rel >= 0 && c != L.wCell - 1 && x != w - kEdge - 1) ? 0xFFFFu : 0u;
        }
        mLe = ml[0] | (ml[2] << 16); mLo = ml[1] | (ml[3] << 16);
        mRe = mr[0] | (mr[2] << 16); mRo = mr[1] | (mr[3] << 16);
    }
    const u16x2 thPair = as_pk((uint32_t)a.minTh | ((uint32_t)a.minTh << 16));

    uint32_t A[7], B[7], C[7];
    // prologue: rows Y0-3 .. Y0+2 into window slots 0..5 (16 <= row <= h-17: always inside the image)
#pragma unroll
    for (int u = 0; u < 6; u++) {
        B[u] = *reinterpret_cast<const uint32_t *>(img + (size_t)(Y0 - 3 + u) * pitch + xl);
        A[u] = dpp_from_left(B[u]);
        C[u] = dpp_from_right(B[u]);
    }
    A[6] = B[6] = C[6] = 0;
    FastRowS cur, prev;                 // score rows c (newest) and c-1
    u16x2 ppMe = as_pk(0u), ppMo = as_pk(0u);   // row c-2, vertical contribution only
    prev.se = prev.so = prev.he = prev.ho = prev.me = prev.mo = as_pk(0u);
    cur = prev;
    const u16x2 one = as_pk(0x00010001u);

    // append the kept scores of one output row (bytes of Pd) to the wave's survivor list, row-major
    auto emit_row = [&](uint32_t Pd, int o) {
        if (__ballot(Pd != 0) != 0) {   // wave-uniform: about one row step in ten has no survivor at all
            const unsigned long long B0 = __ballot((Pd & 0xFFu) != 0), B1 = __ballot((Pd & 0xFF00u) != 0),
                                     B2 = __ballot((Pd & 0xFF0000u) != 0), B3 = __ballot((Pd & 0xFF000000u) != 0);
            // lanes ascending, then the lane's bytes ascending
            int pos = listCount + __builtin_amdgcn_mbcnt_hi((uint32_t)(B0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)B0, 0)) +
                      __builtin_amdgcn_mbcnt_hi((uint32_t)(B1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)B1, 0)) +
                      __builtin_amdgcn_mbcnt_hi((uint32_t)(B2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)B2, 0)) +
                      __builtin_amdgcn_mbcnt_hi((uint32_t)(B3 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)B3, 0));
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const uint32_t sv = (Pd >> (8 * b)) & 0xFFu;
                if (sv) {
                    if (pos < kFastListCap) {
                        sList[wvi][pos] = sv | recBase[b] | ((uint32_t)o << 17);
                        atomicMax(&sCellMax[wvi][(recBase[b] >> 8) & 7], (int)sv);
                    }
                    pos++;
                }
            }
            listCount += __popcll(B0) + __popcll(B1) + __popcll(B2) + __popcll(B3);
        }
    };

    // steady state, one detection row c per step: load row Y0+3+c into slot (6+c)%7, score row c, emit row c-1
    for (int cb = 0; cb < dh; cb += 7) {
#pragma unroll
        for (int u7 = 0; u7 < 7; u7++) {
            const int c = cb + u7;
            if (c >= dh) break;
            const int u = (6 + u7) % 7;   // slot of the newest row
            B[u] = *reinterpret_cast<const uint32_t *>(img + (size_t)(Y0 + 3 + c) * pitch + xl);
            A[u] = dpp_from_left(B[u]);
            C[u] = dpp_from_right(B[u]);
            // window slots: row c-3 = (u+1)%7 ... row c = (u+4)%7 ... row c+3 = u
            const int s0 = (u + 1) % 7, s1 = (u + 2) % 7, s2 = (u + 3) % 7, s3 = (u + 4) % 7, s4 = (u + 5) % 7,
                      s5 = (u + 6) % 7, s6 = u;
            uint32_t W[16];
            // ring offsets (dx,dy), k = 0..15: (0,3),(1,3),(2,2),(3,1),(3,0),(3,-1),(2,-2),(1,-3),(0,-3),(-1,-3),(-2,-2),
            //                                  (-3,-1),(-3,0),(-3,1),(-2,2),(-1,3)
            W[0] = B[s6];
            W[1] = __builtin_amdgcn_alignbyte(C[s6], B[s6], 1);
            W[2] = __builtin_amdgcn_alignbyte(C[s5], B[s5], 2);
            W[3] = __builtin_amdgcn_alignbyte(C[s4], B[s4], 3);
            W[4] = __builtin_amdgcn_alignbyte(C[s3], B[s3], 3);
            W[5] = __builtin_amdgcn_alignbyte(C[s2], B[s2], 3);
            W[6] = __builtin_amdgcn_alignbyte(C[s1], B[s1], 2);
            W[7] = __builtin_amdgcn_alignbyte(C[s0], B[s0], 1);
            W[8] = B[s0];
            W[9] = __builtin_amdgcn_alignbyte(B[s0], A[s0], 3);
            W[10] = __builtin_amdgcn_alignbyte(B[s1], A[s1], 2);
            W[11] = __builtin_amdgcn_alignbyte(B[s2], A[s2], 1);
            W[12] = __builtin_amdgcn_alignbyte(B[s3], A[s3], 1);
            W[13] = __builtin_amdgcn_alignbyte(B[s4], A[s4], 1);
            W[14] = __builtin_amdgcn_alignbyte(B[s5], A[s5], 2);
            W[15] = __builtin_amdgcn_alignbyte(B[s6], A[s6], 3);
            u16x2 r[16];
#pragma unroll
            for (int k = 0; k < 16; k++) r[k] = pair_even(W[k]);
            const u16x2 se = fast_arc_score_pk(pair_even(B[s3]), r);
#pragma unroll
            for (int k = 0; k < 16; k++) r[k] = pair_odd(W[k]);
            const u16x2 so = fast_arc_score_pk(pair_odd(B[s3]), r);
            // shift the score-row window
            ppMe = prev.me; ppMo = prev.mo;
            prev = cur;
            cur.se = se; cur.so = so;
            // horizontal neighbours: even pixels (0,2) see (-1,1) and (1,3); odd pixels (1,3) see (0,2) and (2,4)
            const uint32_t soL = dpp_from_left(as_u32(so)), seR = dpp_from_right(as_u32(se));
            const u16x2 le = as_pk(__builtin_amdgcn_alignbit(as_u32(so), soL, 16) & mLe);
            const u16x2 re = as_pk(as_u32(so) & mRe);
            const u16x2 lo = as_pk(as_u32(se) & mLo);
            const u16x2 ro = as_pk(__builtin_amdgcn_alignbit(seR, as_u32(se), 16) & mRo);
            cur.he = pkmax(le, re);
            cur.ho = pkmax(lo, ro);
            cur.me = pkmax(cur.he, se);
            cur.mo = pkmax(cur.ho, so);
            // ---- output row o = c-1 (`prev`): rows o-1 = pp (masked on the cell's first row), o+1 = cur.
            //      For c = 0 there is no row yet: prev is all zero, so nothing is kept.
            const uint32_t topOK = c > 1 ? 0xFFFFFFFFu : 0u;
            u16x2 ne = pkmax(prev.he, as_pk(as_u32(ppMe) & topOK)), no = pkmax(prev.ho, as_pk(as_u32(ppMo) & topOK));
            ne = pkmax(ne, cur.me);
            no = pkmax(no, cur.mo);
            // keep S iff S > max(neighbours, minTh)
            const u16x2 ke = pkmin(pksubsat(prev.se, pkmax(ne, thPair)), one), ko = pkmin(pksubsat(prev.so, pkmax(no, thPair)), one);
            const uint32_t pe = as_u32(prev.se * ke), po = as_u32(prev.so * ko);
            emit_row((pe | (po << 8)) & ownMask, c - 1);
        }
    }
    {
        // ---- last row o = dh-1 (`cur`): row above = prev (masked if the cell row has a single row), nothing below
        const uint32_t topOK = dh > 1 ? 0xFFFFFFFFu : 0u;
        const u16x2 ne = pkmax(cur.he, as_pk(as_u32(prev.me) & topOK)), no = pkmax(cur.ho, as_pk(as_u32(prev.mo) & topOK));
        const u16x2 ke = pkmin(pksubsat(cur.se, pkmax(ne, thPair)), one), ko = pkmin(pksubsat(cur.so, pkmax(no, thPair)), one);
        const uint32_t pe = as_u32(cur.se * ke), po = as_u32(cur.so * ko);
        emit_row((pe | (po << 8)) & ownMask, dh - 1);
    }

    // ---- flush: per-cell threshold, one reservation in the level's dense array, candidates grouped by cell
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (listCount > kFastListCap) {
        if (lane == 0) atomicOr(a.status, 4);
        listCount = kFastListCap;
    }
    const int nCells = c1 - c0;                      // <= 8
    int thCell[8], cntCell[8];
#pragma unroll
    for (int c = 0; c < 8; c++) {
        thCell[c] = (c < nCells && sCellMax[wvi][c] > a.iniTh) ? a.iniTh : a.minTh;
        cntCell[c] = 0;
    }
    for (int i0 = 0; i0 < listCount; i0 += 64) {
        const int i = i0 + lane;
        const uint32_t rec = i < listCount ? sList[wvi][i] : 0u;
        const int cis = (rec >> 8) & 7, sv = rec & 0xFF;
#pragma unroll
        for (int c = 0; c < 8; c++) cntCell[c] += __popcll(__ballot(i < listCount && cis == c && sv > thCell[c]));
    }
    int offCell[8], total = 0;
#pragma unroll
    for (int c = 0; c < 8; c++) { offCell[c] = total; total += cntCell[c]; }
    int base = 0;
    if (lane == 0 && total) base = atomicAdd(&a.levelCount[f * kMaxLevels + t.level], total);
    base = __shfl(base, 0);
    {
        int2 *tab = a.cellTab + (size_t)f * a.g.totalCells + L.cellBase + t.cellRow * L.nCols + c0;
#pragma unroll
        for (int c = 0; c < 8; c++)
            if (lane == c && c < nCells) tab[c] = make_int2(base + offCell[c], cntCell[c]);
    }
    Cand *out = a.cand + (size_t)f * a.g.candFrameCap + L.candOff;
    int run[8];
#pragma unroll
    for (int c = 0; c < 8; c++) run[c] = base + offCell[c];
    for (int i0 = 0; i0 < listCount; i0 += 64) {
        const int i = i0 + lane;
        const uint32_t rec = i < listCount ? sList[wvi][i] : 0u;
        const int cis = (rec >> 8) & 7, sv = rec & 0xFF;
        int pos = -1;
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const bool mine = i < listCount && cis == c && sv > thCell[c];
            const unsigned long long Bc = __ballot(mine);
            if (mine) pos = run[c] + __builtin_amdgcn_mbcnt_hi((uint32_t)(Bc >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)Bc, 0));
            run[c] += __popcll(Bc);
        }
        if (pos >= 0 && pos < L.candCap) {
            // cv::FAST reports window coordinates (detection pixel + 3); the reference adds (j*wCell, i*hCell) (:822-823)
            Cand c;
            c.x = (uint16_t)(((rec >> 11) & 63) + 3 + (c0 + cis) * L.wCell);
            c.y = (uint16_t)((rec >> 17) + 3 + t.cellRow * L.hCell);
            c.resp = (uint16_t)(sv - 1);
            c.pad = 0;
            out[pos] = c;
        }
    }
}

}  // namespace orbfe
