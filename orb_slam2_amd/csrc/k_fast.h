// k_fast.h -- K3: per-cell FAST-9/16 with the iniThFAST -> minThFAST fallback, 3x3 NMS and ordered emission (gfx950).
//
// Replaces the cell loop of ORBextractor::ComputeKeyPointsOctTree (reference src/ORBextractor.cc:771-829) and the two
// cv::FAST(cell, kps, th, true) calls inside it (:809-816; OpenCV FAST_t<16>).
//
// One 256-thread workgroup = one cell of one level of one frame (the reference's unit of work: NMS and the threshold
// fallback never look across cell borders, so no inter-workgroup exchange is needed).
//   1. stage the cell window (<= wCell+6 x hCell+6 pixels) in LDS with aligned dword row loads;
//   2. per detection pixel compute the arc score  S = max over the 16 nine-arcs A of
//        max( min_{k in A}(v - ring_k), min_{k in A}(ring_k - v) ) = max(v - min_A max_A ring, max_A min_A ring - v)
//      with a max3/min3 network on the raw ring bytes (p is a corner at threshold t  <=>  S > t; response = S - 1);
//   3. strict 3x3 local-maximum test inside the cell (a neighbour outside the cell's scan range counts as 0).  Since a
//      survivor has S > t, comparing against raw neighbour scores equals comparing against thresholded ones, hence
//      survivors(t) = {local maxima with S > t} for every t: one score pass serves both thresholds;
//   4. threshold = any survivor at iniTh ? iniTh : minTh  (the reference re-runs FAST iff the post-NMS list is empty);
//   5. emit survivors in row-major order (the order cv::FAST pushes them) into the level's dense candidate array; the
//      cell's (offset, count) goes to a cell table so the consumer can walk cells in the reference's order.
#pragma once
#include "common.h"

namespace orbfe {

constexpr int kFastMaxWin = 72;                 // max window edge: wCell <= 59 (nCols = floor(width/30)) + 6, padded
constexpr int kFastMaxDet = kFastMaxWin - 6;    // max detection edge
constexpr int kFastWinPitch = kFastMaxWin + 8;  // staged row pitch (dword aligned start costs up to 3 extra bytes)

struct FastArgs {
    Geom g;
    ImgSrc s;
    Cand *cand;          // [F][candFrameCap]
    int *levelCount;     // [F][kMaxLevels]  (zeroed before launch)
    int2 *cellTab;       // [F][totalCells]  (offset inside the level's array, count)
    int iniTh, minTh;
};

__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }
__device__ __forceinline__ int min3i(int a, int b, int c) { return min(min(a, b), c); }

// arc score from the centre value and the 16 ring values
__device__ __forceinline__ int fast_arc_score(int v, const int (&r)[16]) {
    int mx3[16], mn3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        mx3[k] = max3i(r[k], r[(k + 1) & 15], r[(k + 2) & 15]);
        mn3[k] = min3i(r[k], r[(k + 1) & 15], r[(k + 2) & 15]);
    }
    int minOfMax = 255, maxOfMin = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int mx9 = max3i(mx3[k], mx3[(k + 3) & 15], mx3[(k + 6) & 15]);
        const int mn9 = min3i(mn3[k], mn3[(k + 3) & 15], mn3[(k + 6) & 15]);
        minOfMax = min(minOfMax, mx9);
        maxOfMin = max(maxOfMin, mn9);
    }
    return max(v - minOfMax, maxOfMin - v);
}

__global__ __launch_bounds__(256) void fast_cells_kernel(FastArgs a) {
    __shared__ __align__(16) uint8_t win[kFastMaxWin][kFastWinPitch];
    __shared__ uint8_t sc[kFastMaxDet + 2][kFastMaxDet + 2 + 2];   // score plane with a zero frame
    __shared__ uint64_t rowMask[kFastMaxDet];                       // survivors per detection row
    __shared__ int rowBase[kFastMaxDet + 1];
    __shared__ int sBase;

    const int tid = threadIdx.x;
    const int f = blockIdx.y;
    const int cell = blockIdx.x;
    int lvl = 0;
    while (lvl + 1 < a.g.nlevels && cell >= a.g.lv[lvl + 1].cellBase) lvl++;
    const LevelGeom &L = a.g.lv[lvl];
    const int ci = (cell - L.cellBase) / L.nCols, cj = (cell - L.cellBase) - ci * L.nCols;
    const int iniX = kMinBorder + cj * L.wCell, iniY = kMinBorder + ci * L.hCell;
    const int maxX = min(iniX + L.wCell + 6, L.w - kMinBorder), maxY = min(iniY + L.hCell + 6, L.h - kMinBorder);
    const int cw = maxX - iniX, ch = maxY - iniY;
    int2 *tab = a.cellTab + (size_t)f * a.g.totalCells + cell;
    if (cw < 7 || ch < 7) {   // reference: cell skipped (:794,:803) or FAST scans nothing
        if (tid == 0) *tab = make_int2(0, 0);
        return;
    }
    const int dw = cw - 6, dh = ch - 6;   // detection area: window x in [3,cw-3), y in [3,ch-3)

    int pitch;
    const uint8_t *img = level_ptr(a.s, a.g, f, lvl, &pitch);
    const int ax = iniX & ~3, shift = iniX - ax;
    const int nd = ((maxX - 1 - ax) >> 2) + 1;
    for (int i = tid; i < nd * ch; i += 256) {
        const int r = i / nd, c = i - r * nd;
        const uint32_t v = *reinterpret_cast<const uint32_t *>(img + (size_t)(iniY + r) * pitch + ax + 4 * c);
        *reinterpret_cast<uint32_t *>(&win[r][4 * c]) = v;
    }
    for (int i = tid; i < (dh + 2) * (kFastMaxDet + 4); i += 256) (&sc[0][0])[i] = 0;
    if (tid < kFastMaxDet) rowMask[tid] = 0;
    __syncthreads();

    // ---- score pass: 32 x 8 pixel patches
    const int tx = tid & 31, ty = tid >> 5;
    for (int by = 0; by < dh; by += 8) {
        for (int bx = 0; bx < dw; bx += 32) {
            const int x = bx + tx, y = by + ty;
            if (x < dw && y < dh) {
                const uint8_t *p = &win[y + 3][x + 3 + shift];
                const int v = p[0];
                int r[16];
                r[0] = p[3 * kFastWinPitch];          r[1] = p[3 * kFastWinPitch + 1];
                r[2] = p[2 * kFastWinPitch + 2];      r[3] = p[kFastWinPitch + 3];
                r[4] = p[3];                          r[5] = p[-kFastWinPitch + 3];
                r[6] = p[-2 * kFastWinPitch + 2];     r[7] = p[-3 * kFastWinPitch + 1];
                r[8] = p[-3 * kFastWinPitch];         r[9] = p[-3 * kFastWinPitch - 1];
                r[10] = p[-2 * kFastWinPitch - 2];    r[11] = p[-kFastWinPitch - 3];
                r[12] = p[-3];                        r[13] = p[kFastWinPitch - 3];
                r[14] = p[2 * kFastWinPitch - 2];     r[15] = p[3 * kFastWinPitch - 1];
                const int S = fast_arc_score(v, r);
                sc[y + 1][x + 1] = (uint8_t)max(S, 0);
            }
        }
    }
    __syncthreads();

    // ---- strict 3x3 local maxima above minTh; a wave covers two detection rows of 32 columns
    int any20 = 0;
    for (int by = 0; by < dh; by += 8) {
        for (int bx = 0; bx < dw; bx += 32) {
            const int x = bx + tx, y = by + ty;
            bool keep = false;
            int s = 0;
            if (x < dw && y < dh) {
                s = sc[y + 1][x + 1];
                if (s > a.minTh) {
                    keep = s > sc[y][x] && s > sc[y][x + 1] && s > sc[y][x + 2] && s > sc[y + 1][x] && s > sc[y + 1][x + 2] &&
                           s > sc[y + 2][x] && s > sc[y + 2][x + 1] && s > sc[y + 2][x + 2];
                }
            }
            any20 |= (keep && s > a.iniTh);
            const uint64_t m = __ballot(keep);
            if ((tid & 63) == 0) {
                const int yy = by + (tid >> 6) * 2;
                if (yy < dh) atomicOr((unsigned long long *)&rowMask[yy], (unsigned long long)(m & 0xffffffffull) << bx);
                if (yy + 1 < dh) atomicOr((unsigned long long *)&rowMask[yy + 1], (unsigned long long)(m >> 32) << bx);
            }
        }
    }
    const int th = __syncthreads_or(any20) ? a.iniTh : a.minTh;

    // ---- per-row survivor masks at the chosen threshold, row offsets by a serial scan (<= 66 rows)
    if (tid < dh) {
        uint64_t m = rowMask[tid], keepm = 0;
        while (m) {
            const int x = __ffsll((long long)m) - 1;
            m &= m - 1;
            if (sc[tid + 1][x + 1] > th) keepm |= 1ull << x;
        }
        rowMask[tid] = keepm;
    }
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int y = 0; y < dh; y++) { rowBase[y] = acc; acc += __popcll(rowMask[y]); }
        rowBase[dh] = acc;
        int base = 0;
        if (acc) base = atomicAdd(&a.levelCount[f * kMaxLevels + lvl], acc);
        sBase = base;
        *tab = make_int2(base, acc);
    }
    __syncthreads();
    if (tid < dh) {
        uint64_t m = rowMask[tid];
        int o = sBase + rowBase[tid];
        Cand *out = a.cand + (size_t)f * a.g.candFrameCap + L.candOff;
        // cv::FAST reports window coordinates; the reference adds (j*wCell, i*hCell) (:822-823)
        const int yAbs = tid + 3 + ci * L.hCell;
        while (m) {
            const int x = __ffsll((long long)m) - 1;
            m &= m - 1;
            if (o < L.candCap) {
                Cand c;
                c.x = (uint16_t)(x + 3 + cj * L.wCell);
                c.y = (uint16_t)yAbs;
                c.resp = (uint16_t)(sc[tid + 1][x + 1] - 1);
                c.pad = 0;
                out[o] = c;
            }
            o++;
        }
    }
}

}  // namespace orbfe
