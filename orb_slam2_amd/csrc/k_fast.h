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
//     every t >= 1.  (t = 0 behaves as t = 1: the score cv::FAST stores is S - 1, which for S = 1 is not above the 0 of a
//     non-corner; the host passes max(t, 1), orbfe.hip run_batch.)  One score pass therefore serves both thresholds, and "re-run with minThFAST iff the post-NMS list is
//     empty" (:812-816) becomes: threshold = (any survivor with S > iniTh) ? iniTh : minTh, per cell.
//
// One kernel, fast_cells_kernel, register-sliding like the blur: one wavefront owns (one cell row) x (a strip of whole
// cells, <= 62 dword columns) of one level and walks down the rows.  Per row each lane loads ONE dword (a coalesced 256-B
// segment that starts 4 pixels left of the strip, at any alignment) and immediately converts it to PAIR FORM: two pixels per register in 16-bit lanes, for the eight
// horizontal offsets -3..+4 the ring needs (2 masks/permutes, 4 DPP wave shifts for the neighbour lanes' pairs, 4
// v_alignbit).  The last 7 rows stay in registers in that form, so the sixteen ring values of both pixel pairs of every
// row are plain register reads.  Arc score per pair: windows k and k+1 share ring[k+1..k+8], so
//     min(max W_k, max W_k+1) = max(M8[k+1], min(ring[k], ring[k+9]))
// with the eight M8 (maxima of 8 consecutive ring values from odd starts) built by doubling: 47 v_pk_max/min_u16 per
// side, 97 per pixel pair.  The strict 3x3 local-maximum test runs in the same registers (cell edges masked).
// A lane whose dword holds a survivor above minTh (about 1 % of the pixels) appends ONE record (the dword of kept
// scores, lane, row) to a per-wave LDS list; records are in row-major order.  Because a wave owns whole cells, the
// threshold decision is wave-local: at the end of the strip the wave takes each cell's maximum, filters its list per
// cell, reserves space in the level's dense candidate array with ONE atomicAdd and writes the candidates grouped by cell
// in cv::FAST's order; the cell's (offset, count) goes to a cell table so a consumer can walk cells in the reference's
// order.  No barriers, no score plane in HBM.
#pragma once
#include "common.h"

namespace orbfe {

constexpr int kFastStripMaxDw = 62;   // detection lanes of a segment that fills a wave; one halo lane on each side
constexpr int kFastMaxSeg = 6;        // segments (pieces of different cell rows of one level) packed into one wavefront
constexpr int kFastMaxCells = 8;      // cells per wavefront, over all its segments

// A segment is nCells whole cells [c0, c0 + nCells) of cell row `cellRow`; it occupies nLanes consecutive lanes from
// lane0: a left halo lane, ceil(pixels / 4) detection lanes, a right halo lane.  Strips that do not fill a wave (the
// remainder of a cell row) are packed together, so that narrow levels do not leave most of a wavefront idle.  All
// segments of a tile belong to one level and have the same number of rows.
struct FastSeg { uint16_t cellRow, c0; uint8_t nCells, lane0, nLanes, pad; };
struct FastTile {
    uint16_t level, nSeg;
    uint16_t pad[2];
    FastSeg seg[kFastMaxSeg];
};
static_assert(sizeof(FastTile) == 56, "FastTile");

constexpr int kFastListCap = 1024;    // survivor records (one per lane and row) a wave can hold for its strip (dense texture: ~150)
// The most a strip can produce: every lane of every row (one kept pixel per 2x2 block still allows a record per dword and
// row) plus one split record per inner cell edge: (64 + 7) x 60 rows.  The host switches a handle to the kernel variant
// with this capacity (one wave per workgroup, 34 KB LDS) when a wave of the normal variant reports an overflow.
constexpr int kFastListCapMax = 4288;

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
// per 16-bit lane: s where s > lim, else 0 -- saturating subtract, min with 1, multiply.  Spelled out: written with the
// elementwise builtins the compiler rewrites it into 16-bit compares and selects per half (11 instructions per pixel pair
// instead of 3).
__device__ __forceinline__ uint32_t pk_keep_above(u16x2 s, u16x2 lim, u16x2 one) {
    uint32_t r;
    asm("v_pk_sub_u16 %0, %1, %2 clamp\n\tv_pk_min_u16 %0, %0, %3\n\tv_pk_mul_lo_u16 %0, %1, %0"
        : "=&v"(r)
        : "v"(__builtin_bit_cast(uint32_t, s)), "v"(__builtin_bit_cast(uint32_t, lim)), "v"(__builtin_bit_cast(uint32_t, one)));
    return r;
}

// even pixels (bytes 0,2) / odd pixels (bytes 1,3) of a 4-pixel byte window as two 16-bit lanes
__device__ __forceinline__ u16x2 pair_even(uint32_t w) { return as_pk(w & 0x00FF00FFu); }
__device__ __forceinline__ u16x2 pair_odd(uint32_t w) { return as_pk(__builtin_amdgcn_perm(0u, w, 0x0c030c01u)); }

// gfx950 has three-input packed min / max (v_pk_minimum3_f16 / v_pk_maximum3_f16) at the rate of the two-input packed
// ops (measured, profiles/r1_valu_rates_ubench.txt).  Pixel pairs carry a bias of 0x4000 in each 16-bit lane (fast_row_pairs),
// which makes them normal half-precision numbers 2.0 + p/512 ordered like p, so the float min / max select exactly what
// the integer ones would, whatever the denormal mode; differences of two biased values are the unbiased differences.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 pkmax3(u16x2 a, u16x2 b, u16x2 c) {
    const f16x2 m = __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b)),
                                                  __builtin_bit_cast(f16x2, c));
    return __builtin_bit_cast(u16x2, m);
}
__device__ __forceinline__ u16x2 pkmin3(u16x2 a, u16x2 b, u16x2 c) {
    const f16x2 m = __builtin_elementwise_minimum(__builtin_elementwise_minimum(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b)),
                                                  __builtin_bit_cast(f16x2, c));
    return __builtin_bit_cast(u16x2, m);
}

// arc score of two pixels at once: v = centres, r[k] = ring values, all in 16-bit lanes holding 0x4000 + (0..255)
__device__ __forceinline__ u16x2 fast_arc_score_pk(u16x2 v, const u16x2 (&r)[16]) {
    u16x2 x2[8], n2[8], x4[8], n4[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {      // ring[j], ring[j+1], j = 2i+1
        x2[i] = pkmax(r[2 * i + 1], r[(2 * i + 2) & 15]);
        n2[i] = pkmin(r[2 * i + 1], r[(2 * i + 2) & 15]);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {      // ring[j..j+3]
        x4[i] = pkmax(x2[i], x2[(i + 1) & 7]);
        n4[i] = pkmin(n2[i], n2[(i + 1) & 7]);
    }
    // windows k = 2i and k+1 share ring[k+1..k+8] = M8[j = k+1]:  min(maxW_k, maxW_k+1) = max(M8, min(r[k], r[k+9]))
    u16x2 wx[8], wn[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int k = 2 * i;
        wx[i] = pkmax3(x4[i], x4[(i + 2) & 7], pkmin(r[k], r[(k + 9) & 15]));
        wn[i] = pkmin3(n4[i], n4[(i + 2) & 7], pkmax(r[k], r[(k + 9) & 15]));
    }
    const u16x2 minOfMax = pkmin(pkmin3(pkmin3(wx[0], wx[1], wx[2]), pkmin3(wx[3], wx[4], wx[5]), wx[6]), wx[7]);
    const u16x2 maxOfMin = pkmax(pkmax3(pkmax3(wn[0], wn[1], wn[2]), pkmax3(wn[3], wn[4], wn[5]), wn[6]), wn[7]);
    return pkmax(pksubsat(v, minOfMax), pksubsat(maxOfMin, v));   // max(S, 0)
}

__device__ __forceinline__ uint32_t dpp_from_left(uint32_t v) {    // lane i <- lane i-1 (0 into lane 0)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t dpp_from_right(uint32_t v) {   // lane i <- lane i+1 (0 into lane 63)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true);
}

struct FastRowS {      // one row of scores in pair form, with what the 3x3 test needs from it
    u16x2 se, so;      // scores of pixels (0,2) and (1,3)
    u16x2 he, ho;      // max of the left and right neighbours in this row (cell edges masked)
    u16x2 me, mo;      // max of left, centre, right in this row (what the rows above/below see)
};

// one image row in pair form: q[o + 3] = pixels (x0 + o, x0 + o + 2) of the lane's dword, o = -3..4, each 16-bit lane
// biased by 0x4000 (see pkmax3)
__device__ __forceinline__ void fast_row_pairs(uint32_t raw, uint32_t (&q)[8]) {
    const uint32_t E = (raw & 0x00FF00FFu) | 0x40004000u;                   // (p0, p2)
    const uint32_t O = __builtin_amdgcn_perm(0x40404040u, raw, 0x04030401u);   // (p1, p3)
    const uint32_t EL = dpp_from_left(E), OL = dpp_from_left(O);            // (p-4, p-2), (p-3, p-1)
    const uint32_t ER = dpp_from_right(E), OR = dpp_from_right(O);          // (p4, p6),  (p5, p7)
    q[0] = OL;
    q[1] = __builtin_amdgcn_alignbit(E, EL, 16);                            // (p-2, p0)
    q[2] = __builtin_amdgcn_alignbit(O, OL, 16);                            // (p-1, p1)
    q[3] = E;
    q[4] = O;
    q[5] = __builtin_amdgcn_alignbit(ER, E, 16);                            // (p2, p4)
    q[6] = __builtin_amdgcn_alignbit(OR, O, 16);                            // (p3, p5)
    q[7] = ER;
}

// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ int wave_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);    // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);    // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);    // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);    // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return v;
}

// ---- the end of a tile: `list` holds the wave's survivor records in row-major
// order (x = kept scores, one byte per pixel of the lane's dword; y = lane | wave-cell of each of the 4 pixels << 6 | row
// << 18), cellMax[16] is zero, seg[] describes the tile's segments (first lane, first wave-cell, x and y origin).
template <int CAP>
__device__ __forceinline__ void fast_flush(const FastArgs &a, const FastTile &t, const LevelGeom &L, int f, int level, int nSeg,
                                           int nCellsTotal, int lane, int listCount, uint2 *list, int *cellMax, const int4 *seg) {
    // ---- flush: per-cell maximum -> threshold, one reservation in the level's dense array, candidates grouped by cell
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (listCount > CAP) {
        if (lane == 0) atomicOr(a.status, 4);
        listCount = CAP;
    }
    // a record's lane gives its segment (sSeg), the segment gives the cell numbering and the output origin
    auto rec_seg = [&](uint32_t recLane) {
        int si = 0;
        for (int k = 1; k < nSeg; k++) si += (int)recLane >= seg[k].x ? 1 : 0;   // segments are in lane order
        return seg[si];
    };
    // pass A: per cell the maximum kept score and two counters, kept scores (low half) and kept scores above iniTh (high
    // half): the cell's threshold is iniTh iff its maximum is above iniTh, and its candidates are then the second counter
    for (int i0 = 0; i0 < listCount; i0 += 64) {
        const int i = i0 + lane;
        if (i < listCount) {
            const uint2 rec = list[i];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int sv = (rec.x >> (8 * b)) & 0xFF;
                if (sv) {
                    const int cw = (rec.y >> (6 + 3 * b)) & 7;
                    atomicMax(&cellMax[cw], sv);
                    atomicAdd(&cellMax[kFastMaxCells + cw], sv > a.iniTh ? 0x10001 : 1);
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int cntCell[kFastMaxCells];
    uint32_t iniCells = 0;   // bit c: cell c has a survivor above iniTh, so iniTh is its threshold
#pragma unroll
    for (int c = 0; c < kFastMaxCells; c++) {
        const bool ini = c < nCellsTotal && cellMax[c] > a.iniTh;
        const int both = cellMax[kFastMaxCells + c];
        iniCells |= ini ? (1u << c) : 0u;
        cntCell[c] = c < nCellsTotal ? (ini ? both >> 16 : both & 0xFFFF) : 0;
    }
    // A record holds at most two kept scores (one cell's, never adjacent pixels).  slots(): the candidates of a record as two slots,
    // (wave-cell, byte) of the first and of the second byte that passes its cell's threshold; cell 15 = none.
    auto slots = [&](const uint2 &rec, int &c0, int &b0, int &c1, int &b1) {
        uint32_t pass = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int sv = (rec.x >> (8 * b)) & 0xFF, cw = (rec.y >> (6 + 3 * b)) & 7;
            const int th = (iniCells >> cw) & 1 ? a.iniTh : a.minTh;
            pass |= sv > th ? (1u << b) : 0u;
        }
        b0 = pass ? __builtin_ctz(pass) : 0;
        b1 = pass ? 31 - __builtin_clz(pass) : 0;
        c0 = pass ? (int)((rec.y >> (6 + 3 * b0)) & 7) : 15;
        c1 = (pass && b1 != b0) ? (int)((rec.y >> (6 + 3 * b1)) & 7) : 15;
    };
    int offCell[kFastMaxCells], total = 0;
#pragma unroll
    for (int c = 0; c < kFastMaxCells; c++) { offCell[c] = total; total += cntCell[c]; }
    int base = 0;
    if (lane == 0 && total) base = atomicAdd(&a.levelCount[f * kMaxLevels + level], total);
    base = __shfl(base, 0);
    {
        // lane c writes the table entry of wave-cell c: find its segment, then (cell row, cell column)
        int tabIdx = -1;
        int cellsBefore = 0;
        for (int sgi = 0; sgi < nSeg; sgi++) {
            const FastSeg sg = t.seg[sgi];
            if (lane >= cellsBefore && lane < cellsBefore + sg.nCells) tabIdx = sg.cellRow * L.nCols + sg.c0 + (lane - cellsBefore);
            cellsBefore += sg.nCells;
        }
        int2 *tab = a.cellTab + (size_t)f * a.g.totalCells + L.cellBase;
#pragma unroll
        for (int c = 0; c < kFastMaxCells; c++)
            if (lane == c && tabIdx >= 0) tab[tabIdx] = make_int2(base + offCell[c], cntCell[c]);
    }
    Cand *out = a.cand + (size_t)f * a.g.candFrameCap + L.candOff;
    // lane c (c < 8) carries the next output position of wave-cell c
    int runV = 0;
#pragma unroll
    for (int c = 0; c < kFastMaxCells; c++) runV = lane == c ? base + offCell[c] : runV;
    // pass C: write, records ascending then bytes ascending inside each cell (= cv::FAST's row-major order).  The rank of a
    // slot among the chunk's slots of the same cell comes from ONE prefix sum over the lanes of eight byte counters (a lane
    // adds at most 2, a chunk at most 128 per cell), instead of two ballots per cell.
    for (int i0 = 0; i0 < listCount; i0 += 64) {
        const int i = i0 + lane;
        const uint2 rec = i < listCount ? list[i] : make_uint2(0u, 0u);
        int c0, b0, c1, b1;
        slots(rec, c0, b0, c1, b1);
        const uint32_t one0 = c0 < 8 ? 1u << (8 * (c0 & 3)) : 0u, one1 = c1 < 8 ? 1u << (8 * (c1 & 3)) : 0u;
        const uint32_t lo = (c0 < 4 ? one0 : 0u) + (c1 < 4 ? one1 : 0u), hi = (c0 >= 4 ? one0 : 0u) + (c1 >= 4 ? one1 : 0u);
        const uint32_t inLo = (uint32_t)wave_incl_scan((int)lo), inHi = (uint32_t)wave_incl_scan((int)hi);
        const uint32_t exLo = inLo - lo, exHi = inHi - hi;
        const int rank0 = (int)(((c0 < 4 ? exLo : exHi) >> (8 * (c0 & 3))) & 0xFFu);
        const int rank1 = (int)(((c1 < 4 ? exLo : exHi) >> (8 * (c1 & 3))) & 0xFFu) + (c1 == c0 ? 1 : 0);
        // (every lane takes part in the permutes: a lane switched off could not be read by the others)
        const int run0 = __builtin_amdgcn_ds_bpermute(4 * (c0 & 7), runV), run1 = __builtin_amdgcn_ds_bpermute(4 * (c1 & 7), runV);
        const int pos0 = c0 < 8 ? run0 + rank0 : -1;
        const int pos1 = c1 < 8 ? run1 + rank1 : -1;
        {
            const uint32_t tLo = (uint32_t)__builtin_amdgcn_readlane((int)inLo, 63), tHi = (uint32_t)__builtin_amdgcn_readlane((int)inHi, 63);
            runV += lane < kFastMaxCells ? (int)(((lane < 4 ? tLo : tHi) >> (8 * (lane & 3))) & 0xFFu) : 0;
        }
        if (pos0 >= 0) {
            const int4 sg = rec_seg(rec.y & 63);
            const int rel0 = 4 * ((int)(rec.y & 63) - sg.x) - 4;   // x - Xs of byte 0
            const int yo = (int)(rec.y >> 18) + sg.w;
            // cv::FAST reports window coordinates (detection pixel + 3); the reference adds (j*wCell, i*hCell) (:822-823)
            if (pos0 < L.candCap) {
                Cand cnd;
                cnd.x = (uint16_t)(rel0 + b0 + sg.z);
                cnd.y = (uint16_t)yo;
                cnd.resp = (uint16_t)(((rec.x >> (8 * b0)) & 0xFF) - 1);
                cnd.pad = 0;
                out[pos0] = cnd;
            }
            if (pos1 >= 0 && pos1 < L.candCap) {
                Cand cnd;
                cnd.x = (uint16_t)(rel0 + b1 + sg.z);
                cnd.y = (uint16_t)yo;
                cnd.resp = (uint16_t)(((rec.x >> (8 * b1)) & 0xFF) - 1);
                cnd.pad = 0;
                out[pos1] = cnd;
            }
        }
    }
}

template <int CAP, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void fast_cells_kernel(FastArgs a) {
    __shared__ uint2 sList[WAVES][CAP];
    __shared__ int sCellMax[WAVES][2 * kFastMaxCells];   // per wave-cell: maximum kept score; (kept, kept above iniTh) counters
    __shared__ int4 sSeg[WAVES][kFastMaxSeg];   // per segment: first lane, first cell (in wave), x and y of its origin in the output
    const int lane = threadIdx.x & 63;
    const int wvi = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: tile record and row tests on the scalar unit
    const int ti = blockIdx.x * WAVES + wvi;
    if (ti >= a.nTiles) return;
    const FastTile &t = a.tiles[ti];
    const int f = blockIdx.y;
    const int level = t.level, nSeg = t.nSeg;
    const LevelGeom &L = a.g.lv[level];
    const int w = L.w, h = L.h;
    int pitch;
    const uint8_t *img = level_ptr(a.s, a.g, f, level, &pitch);
    // the lane's segment.  Its 4 pixels start at x0 = Xs - 4 + 4 * (lane - lane0) whatever the alignment (unaligned dword
    // loads): lane0 is the left halo, then the detection lanes, then the right halo.  Lanes outside every segment idle.
    int Xs = kEdge, Xe = kEdge, Y0 = kEdge, segLane0 = 0, nCellsTotal = 0;
    bool inSeg = false;
    for (int sgi = 0; sgi < nSeg; sgi++) {   // wave-uniform trip count
        const FastSeg sg = t.seg[sgi];
        const int sXs = kEdge + sg.c0 * L.wCell, sXe = min(kEdge + (sg.c0 + sg.nCells) * L.wCell, w - kEdge);
        const bool mine = lane >= sg.lane0 && lane < sg.lane0 + sg.nLanes;
        if (mine || sgi == 0) {              // lanes of no segment read segment 0's rows (legal addresses) and own nothing
            Xs = sXs; Y0 = kEdge + sg.cellRow * L.hCell; segLane0 = sg.lane0;
            Xe = mine ? sXe : sXs;
        }
        inSeg = inSeg || mine;
        if (lane == sgi) sSeg[wvi][sgi] = make_int4(sg.lane0, nCellsTotal, 3 + sg.c0 * L.wCell, 3 + sg.cellRow * L.hCell);
        nCellsTotal += sg.nCells;
    }
    const int dh = min(kEdge + t.seg[0].cellRow * L.hCell + L.hCell, h - kEdge) - (kEdge + t.seg[0].cellRow * L.hCell);   // same for all segments
    if (dh <= 0) return;
    const int x0 = Xs - 4 + 4 * (lane - segLane0);
    // lanes past the row (or left of it, for idle lanes): a legal address, value unused (needed pixels are in [16, w-16))
    const int xl = min(max(x0, 0), pitch - 4);
    const uint32_t rowOff = (uint32_t)((Y0 - 3) * pitch + xl);
    // Per byte b of the lane's dword (pixel x = x0 + b): is it a detection pixel of the lane's segment (ownMask), which
    // wave-cell it lies in (recTag: lane | cell (0..7) of each pixel << 6, 3 bits each; the row goes to bit 18), and whether its
    // left / right neighbour lies in the same cell (16-bit lanes of 0xFFFF in mL*, mR*).  One division by the cell width per
    // lane: x0 - Xs is a multiple of 4 and the cells are wider than 4 pixels, so the four pixels are in the cell of byte 0 or
    // in the next one.  Pixels left of the segment (the halo lane, idle lanes) count as cell 0 without neighbours.
    uint32_t ownMask = 0, recTag = (uint32_t)lane, sameCell = 0xFFu, mLe, mLo, mRe, mRo;
    {
        int segCell0 = 0, cellsBefore = 0;
        for (int sgi = 0; sgi < nSeg; sgi++) {
            const FastSeg sg = t.seg[sgi];
            if (lane >= sg.lane0 && lane < sg.lane0 + sg.nLanes) segCell0 = cellsBefore;
            cellsBefore += sg.nCells;
        }
        const int rel0 = x0 - Xs;                                   // -4 for the left halo lane
        const int n0 = max(rel0, 0);
        const int q0 = (int)(((uint32_t)n0 * (uint32_t)L.wCellRcp) >> 16), r0 = n0 - q0 * L.wCell;
        uint32_t ml[4], mr[4];
        int cellB0 = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int x = x0 + b;
            const bool in = rel0 >= 0;
            const bool wrap = r0 + b >= L.wCell;
            const int cis = in ? q0 + (wrap ? 1 : 0) : 0, cc = in ? (wrap ? r0 + b - L.wCell : r0 + b) : 0;
            const int cell = min(segCell0 + cis, kFastMaxCells - 1);
            if (b == 0) cellB0 = cell;
            ownMask |= (x >= Xs && x < Xe) ? (0xFFu << (8 * b)) : 0u;
            recTag |= (uint32_t)cell << (6 + 3 * b);
            if (b) sameCell |= cell == cellB0 ? (0xFFu << (8 * b)) : 0u;
            // (x - kEdge) and (x - Xs) differ by whole cells: the same position inside the cell
            ml[b] = (in && x > kEdge && cc != 0) ? 0xFFFFu : 0u;
            mr[b] = (in && cc != L.wCell - 1 && x != w - kEdge - 1) ? 0xFFFFu : 0u;
        }
        mLe = ml[0] | (ml[2] << 16); mLo = ml[1] | (ml[3] << 16);
        mRe = mr[0] | (mr[2] << 16); mRo = mr[1] | (mr[3] << 16);
    }
    if (lane < 2 * kFastMaxCells) sCellMax[wvi][lane] = 0;
    int listCount = 0;   // wave-uniform
    const u16x2 thPair = as_pk((uint32_t)a.minTh | ((uint32_t)a.minTh << 16));

    uint32_t Q[7][8];
    // prologue: rows Y0-3 .. Y0+2 into window slots 0..5 (16 <= row <= h-17: always inside the image)
#pragma unroll
    for (int u = 0; u < 6; u++)
        fast_row_pairs(*reinterpret_cast<const uint32_t *>(img + (size_t)u * pitch + rowOff), Q[u]);
#pragma unroll
    for (int k = 0; k < 8; k++) Q[6][k] = 0;
    FastRowS cur, prev;                 // score rows c (newest) and c-1
    u16x2 ppMe = as_pk(0u), ppMo = as_pk(0u);   // row c-2, vertical contribution only
    prev.se = prev.so = prev.he = prev.ho = prev.me = prev.mo = as_pk(0u);
    cur = prev;
    const u16x2 one = as_pk(0x00010001u);

    // a lane whose dword holds a kept score appends one record (kept scores, recTag | row << 18); row-major order
    // Two horizontally adjacent pixels can both be kept only across a cell edge (NMS does not look over it); a lane with
    // kept scores on both sides of an edge writes two records, so that a record holds one cell's scores: at most two.
    const unsigned long long edgeLanes = __ballot(sameCell != 0xFFFFFFFFu);   // lanes whose dword is cut by a cell edge
    auto emit_row = [&](uint32_t Pd, int o) {
        const unsigned long long Bm = __ballot(Pd != 0);
        if (Bm != 0) {   // wave-uniform
            const uint32_t PdA = Pd & sameCell, PdB = Pd & ~sameCell;
            unsigned long long Bs = 0;
            if (Bm & edgeLanes) Bs = __ballot(PdA != 0 && PdB != 0);   // scalar test: only an edge lane can hold two cells' scores
            int pos = listCount + __builtin_amdgcn_mbcnt_hi((uint32_t)(Bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)Bm, 0));
            if (Bs != 0) {   // wave-uniform, rare
                pos += __builtin_amdgcn_mbcnt_hi((uint32_t)(Bs >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)Bs, 0));
                if (PdA != 0 && PdB != 0) {
                    if (pos < CAP) sList[wvi][pos] = make_uint2(PdA, recTag | ((uint32_t)o << 18));
                    if (pos + 1 < CAP) sList[wvi][pos + 1] = make_uint2(PdB, recTag | ((uint32_t)o << 18));
                } else if (Pd != 0 && pos < CAP) {
                    sList[wvi][pos] = make_uint2(Pd, recTag | ((uint32_t)o << 18));
                }
                listCount += __popcll(Bm) + __popcll(Bs);
            } else {
                if (Pd != 0 && pos < CAP) sList[wvi][pos] = make_uint2(Pd, recTag | ((uint32_t)o << 18));
                listCount += __popcll(Bm);
            }
        }
    };

    // steady state, one detection row c per step: load row Y0+3+c into slot (6+c)%7, score row c, emit row c-1
    for (int cb = 0; cb < dh; cb += 7) {
#pragma unroll
        for (int u7 = 0; u7 < 7; u7++) {
            const int c = cb + u7;
            if (c >= dh) break;
            const int u = (6 + u7) % 7;   // slot of the newest row
            fast_row_pairs(*reinterpret_cast<const uint32_t *>(img + (size_t)(6 + c) * pitch + rowOff), Q[u]);
            // window slots: row c-3 = (u+1)%7 ... row c = (u+4)%7 ... row c+3 = u
            const int s0 = (u + 1) % 7, s1 = (u + 2) % 7, s2 = (u + 3) % 7, s3 = (u + 4) % 7, s4 = (u + 5) % 7,
                      s5 = (u + 6) % 7, s6 = u;
            // ring offsets (dx,dy), k = 0..15: (0,3),(1,3),(2,2),(3,1),(3,0),(3,-1),(2,-2),(1,-3),(0,-3),(-1,-3),(-2,-2),
            //                                  (-3,-1),(-3,0),(-3,1),(-2,2),(-1,3);  even pair: q[dx+3], odd pair: q[dx+4]
            u16x2 r[16];
#define ORBFE_RING(P)                                                                                                    \
            r[0] = as_pk(Q[s6][3 + P]); r[1] = as_pk(Q[s6][4 + P]); r[2] = as_pk(Q[s5][5 + P]); r[3] = as_pk(Q[s4][6 + P]);     \
            r[4] = as_pk(Q[s3][6 + P]); r[5] = as_pk(Q[s2][6 + P]); r[6] = as_pk(Q[s1][5 + P]); r[7] = as_pk(Q[s0][4 + P]);     \
            r[8] = as_pk(Q[s0][3 + P]); r[9] = as_pk(Q[s0][2 + P]); r[10] = as_pk(Q[s1][1 + P]); r[11] = as_pk(Q[s2][0 + P]);   \
            r[12] = as_pk(Q[s3][0 + P]); r[13] = as_pk(Q[s4][0 + P]); r[14] = as_pk(Q[s5][1 + P]); r[15] = as_pk(Q[s6][2 + P]);
            ORBFE_RING(0)
            const u16x2 se = fast_arc_score_pk(as_pk(Q[s3][3]), r);
            ORBFE_RING(1)
            const u16x2 so = fast_arc_score_pk(as_pk(Q[s3][4]), r);
#undef ORBFE_RING
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
            cur.me = pkmax3(cur.he, se, thPair);   // minTh rides along: every real row's (me, mo) is >= minTh, so the 3x3
            cur.mo = pkmax3(cur.ho, so, thPair);   // maximum below already holds max(neighbours, minTh)
            // ---- output row o = c-1 (`prev`): rows o-1 = pp, o+1 = cur.  The score rows start out as zeros, so for c = 0
            //      (no row yet: prev is all zero, nothing is kept) and c = 1 (the cell's first row: pp is still the zero row,
            //      as NMS must not look above the cell) no masking is needed.
            // (scores are 0..255: as half-precision bit patterns these are subnormals, which v_pk_maximum3_f16 orders like the
            // integers as long as f16 denormals are not flushed -- the kernel's default mode; every GPU parity test checks it)
            const u16x2 ne = pkmax3(prev.he, ppMe, cur.me), no = pkmax3(prev.ho, ppMo, cur.mo);
            // keep S iff S > max(neighbours, minTh)
            const uint32_t pe = pk_keep_above(prev.se, ne, one), po = pk_keep_above(prev.so, no, one);
            emit_row((pe | (po << 8)) & ownMask, c - 1);
        }
    }
    {
        // ---- last row o = dh-1 (`cur`): row above = prev (the zero row if the cell row has a single row), nothing below
        const u16x2 ne = pkmax(cur.he, prev.me), no = pkmax(cur.ho, prev.mo);
        const uint32_t pe = pk_keep_above(cur.se, pkmax(ne, thPair), one), po = pk_keep_above(cur.so, pkmax(no, thPair), one);
        emit_row((pe | (po << 8)) & ownMask, dh - 1);
    }

    fast_flush<CAP>(a, t, L, f, level, nSeg, nCellsTotal, lane, listCount, sList[wvi], sCellMax[wvi], sSeg[wvi]);
}

}  // namespace orbfe
