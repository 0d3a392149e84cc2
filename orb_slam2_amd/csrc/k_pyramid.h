// k_pyramid.h -- K1: one pyramid level from the previous one, fixed-point bilinear (gfx950).
//
// Replaces cv::resize(mvImagePyramid[level-1], mvImagePyramid[level], sz, 0, 0, INTER_LINEAR) at reference
// src/ORBextractor.cc:1120 (OpenCV 2.4/3.2 generic 8UC1 path).  All float work (the Q11
// coefficient tables) is done once on the host with the exact float/double sequence; the kernel is pure integer:
//   H[r][dx] = S[r][sx]*a0 + S[r][sx+1]*a1                         (int32)
//   D[dy][dx] = (((b0*(H0>>4))>>16) + ((b1*(H1>>4))>>16) + 2) >> 2
//
// Register-sliding, no LDS: one wavefront owns 256 output columns (4 per lane) x `rows` output rows.
//   * set-up, once per wave: each lane reads its 4 column-table entries and turns them into byte-permute selectors
//     relative to its own source window, which starts at the first column's source pixel: for scale factors in (1,2]
//     the 4 pixel pairs (sx, sx+1) lie inside 8 bytes;
//   * all source rows the row block needs (<= kRszSrcRows) are fetched up front, ONE 8-byte load per lane and row at the
//     lane's (unaligned) window address -- every load of the wave is in flight together;
//   * a source row's horizontal pass is computed once: per pixel one v_perm_b32 builds (S[sx], S[sx+1]) as a 16-bit pair
//     and one v_dot2_u32_u16 against (a0, a1) gives H; the previous row's H values slide along in registers;
//   * an output row is emitted when its lower source row arrives: two 24-bit multiplies per pixel, one dword store per
//     lane (256-B coalesced rows).
#pragma once
#include "common.h"

namespace orbfe {

constexpr int kRszStripPx = 256;     // output columns per wavefront
constexpr int kRszSrcRows = 12;      // source rows a row block may touch (host picks `rows` accordingly)

struct RszTab {      // per output column / row
    int ofs;         // source index (x: clamped to [0, sw-1]; y: unclamped as in the reference)
    int coef;        // lo16 = weight of ofs, hi16 = weight of ofs+1  (Q11, both in [0, 2048])
};

struct ResizeArgs {
    const uint8_t *src; int sPitch; size_t sFrameStride; int sw, sh;
    uint8_t *dst; int dPitch; size_t dFrameStride; int dw, dh;
    const RszTab *xtab;   // padded to a multiple of kRszStripPx entries (last entry replicated)
    const RszTab *ytab;   // padded to a multiple of 64 entries
    int nStrips, nRowBlocks, rows;   // rows <= 64 output rows per wavefront, chosen so that a block touches <= kRszSrcRows rows
};

struct __attribute__((packed, aligned(1))) RszWindow { uint32_t lo, hi; };
struct __attribute__((packed, aligned(1))) RszWindow3 { uint32_t lo, hi, top; };

typedef uint16_t rsz_u16x2 __attribute__((ext_vector_type(2)));

// high 32 bits of the 48-bit product of two values known to fit 24 bits (v_mul_hi_u32_u24)
__device__ __forceinline__ uint32_t rsz_mulhi24(uint32_t x, uint32_t y) {
    uint32_t r;
    asm("v_mul_hi_u32_u24_e32 %0, %1, %2" : "=v"(r) : "s"(x), "v"(y));   // x: wave-uniform row weight
    return r;
}

// WIDE = false: the lane's four pixel pairs lie inside 8 bytes (scale factors up to ~1.99).  WIDE = true: 12-byte window,
// the fourth pixel reads a view shifted by 3 bytes (two v_alignbyte_b32 per source row), good up to scale factors of 3.
template <bool WIDE>
__global__ __launch_bounds__(256) void pyr_resize_kernel(ResizeArgs a) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: row loop in SGPRs
    if (tile >= a.nStrips * a.nRowBlocks) return;
    const int strip = tile % a.nStrips, rb = tile / a.nStrips;   // the 4 waves of a workgroup: neighbouring strips
    const int f = blockIdx.y;
    const int x0 = strip * kRszStripPx + 4 * lane;
    const int y0 = rb * a.rows, y1 = min(y0 + a.rows, a.dh);

    // ---- per-lane column set-up
    const uint4 t01 = *reinterpret_cast<const uint4 *>(a.xtab + x0), t23 = *reinterpret_cast<const uint4 *>(a.xtab + x0 + 2);
    const int ofs[4] = {(int)t01.x, (int)t01.z, (int)t23.x, (int)t23.z};
    const uint32_t coef[4] = {t01.y, t01.w, t23.y, t23.w};
    const int base = min(ofs[0], a.sPitch - (WIDE ? 12 : 8));   // the window never leaves the row's pitch
    uint32_t sel[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int view = (WIDE && i == 3) ? 3 : 0;   // pixel 3 of a wide window selects from bytes 3..10
        const int o0 = ofs[i] - base - view, o1 = min(ofs[i] + 1, a.sw - 1) - base - view;   // sx == sw-1: right weight is 0
        sel[i] = (uint32_t)o0 | 0x0c00u | ((uint32_t)o1 << 16) | 0x0c000000u;
    }
    // ---- the row block's row-table entries, one per lane, and its source rows
    const RszTab ytl = a.ytab[y0 + lane];
    const int ylOfs = ytl.ofs, ylCoef = ytl.coef;
    const int sy0 = min(max(__builtin_amdgcn_readfirstlane(ylOfs), 0), a.sh - 1);
    const uint8_t *srcF = a.src + (size_t)f * a.sFrameStride + base;
    RszWindow3 raw[kRszSrcRows];
#pragma unroll
    for (int j = 0; j < kRszSrcRows; j++) {
        const uint8_t *rp = srcF + (size_t)min(sy0 + j, a.sh - 1) * a.sPitch;
        if (WIDE) {
            raw[j] = *reinterpret_cast<const RszWindow3 *>(rp);
        } else {
            const RszWindow v = *reinterpret_cast<const RszWindow *>(rp);
            raw[j].lo = v.lo; raw[j].hi = v.hi; raw[j].top = 0;
        }
    }

    uint8_t *dstF = a.dst + (size_t)f * a.dFrameStride + x0;
    const bool writer = x0 < a.dw;   // dPitch is a multiple of 64 >= round_up(dw, 4): the dword store stays inside the row
    int dy = y0;
    uint32_t Hp[4] = {0, 0, 0, 0}, Hc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < kRszSrcRows; j++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            Hp[i] = Hc[i];
            uint32_t lo = raw[j].lo, hi = raw[j].hi;
            if (WIDE && i == 3) {
                lo = __builtin_amdgcn_alignbyte(raw[j].hi, raw[j].lo, 3);
                hi = __builtin_amdgcn_alignbyte(raw[j].top, raw[j].hi, 3);
            }
            const uint32_t pr = __builtin_amdgcn_perm(hi, lo, sel[i]);
            // H >> 4 is kept in place (low 4 bits cleared): the vertical pass folds the shift into its multiplier
            Hc[i] = __builtin_amdgcn_udot2(__builtin_bit_cast(rsz_u16x2, pr), __builtin_bit_cast(rsz_u16x2, coef[i]), 0u, false) & ~15u;
        }
        while (dy < y1) {
            const int yo = __builtin_amdgcn_readlane(ylOfs, dy - y0), yc = __builtin_amdgcn_readlane(ylCoef, dy - y0);
            const int r0 = min(max(yo, 0), a.sh - 1) - sy0, r1 = min(max(yo + 1, 0), a.sh - 1) - sy0;
            if (r1 != j) break;            // its lower source row comes later
            // (b * (H >> 4)) >> 16 as the high half of one 24-bit multiply: (b << 12) * (H & ~15) = (b * (H >> 4)) << 16,
            // b <= 2048 and H < 2^19, so both factors fit 24 bits and the product's bits 32.. are the wanted floor
            const uint32_t b0 = ((uint32_t)yc & 0xffffu) << 12, b1 = ((uint32_t)yc >> 16) << 12;
            if (r0 == j) {                 // both rows clamp to the same source row (image top / bottom): wave-uniform
#pragma unroll
                for (int i = 0; i < 4; i++) Hp[i] = Hc[i];   // Hp is overwritten by the next source row anyway
                asm volatile("; top / bottom clamp" ::: "memory");   // keeps this a (rarely taken) scalar branch, not four selects per row
            }
            uint32_t s[4];
#pragma unroll
            for (int i = 0; i < 4; i++) s[i] = rsz_mulhi24(b0, Hp[i]) + rsz_mulhi24(b1, Hc[i]) + 2u;   // <= 1023
            // >> 2 on both 16-bit halves at once, then the four low bytes into one dword
            const rsz_u16x2 q01 = __builtin_bit_cast(rsz_u16x2, (s[1] << 16) | s[0]) >> (uint16_t)2;
            const rsz_u16x2 q23 = __builtin_bit_cast(rsz_u16x2, (s[3] << 16) | s[2]) >> (uint16_t)2;
            const uint32_t out = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, q23), __builtin_bit_cast(uint32_t, q01), 0x06040200u);
            if (writer) *reinterpret_cast<uint32_t *>(dstF + (size_t)dy * a.dPitch) = out;
            dy++;
        }
    }
}

// Any other ratio (scale factors of 2 whose rounded level sizes push the ratio past the 12-byte window, factors >= 3,
// up-sampling): one lane per 4 output pixels of one row, byte gathers.  Same arithmetic, no window assumption.
__global__ __launch_bounds__(256) void pyr_resize_generic_kernel(ResizeArgs a) {
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int dy = blockIdx.y, f = blockIdx.z;
    if (x0 >= a.dw) return;
    const RszTab yt = a.ytab[dy];
    const int r0 = min(max(yt.ofs, 0), a.sh - 1), r1 = min(max(yt.ofs + 1, 0), a.sh - 1);
    const uint32_t b0 = (uint32_t)yt.coef & 0xffffu, b1 = (uint32_t)yt.coef >> 16;
    const uint8_t *p0 = a.src + (size_t)f * a.sFrameStride + (size_t)r0 * a.sPitch;
    const uint8_t *p1 = a.src + (size_t)f * a.sFrameStride + (size_t)r1 * a.sPitch;
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const RszTab xt = a.xtab[x0 + i];                    // padded past dw with the last entry
        const int s0 = xt.ofs, s1 = min(xt.ofs + 1, a.sw - 1);
        const uint32_t a0 = (uint32_t)xt.coef & 0xffffu, a1 = (uint32_t)xt.coef >> 16;
        const uint32_t H0 = (p0[s0] * a0 + p0[s1] * a1) >> 4, H1 = (p1[s0] * a0 + p1[s1] * a1) >> 4;
        const uint32_t d = (((b0 * H0) >> 16) + ((b1 * H1) >> 16) + 2u) >> 2;
        out |= (d & 0xffu) << (8 * i);
    }
    *reinterpret_cast<uint32_t *>(a.dst + (size_t)f * a.dFrameStride + (size_t)dy * a.dPitch + x0) = out;
}

}  // namespace orbfe
