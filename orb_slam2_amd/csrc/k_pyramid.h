// k_pyramid.h -- K1: one pyramid level from the previous one, fixed-point bilinear (gfx950).
//
// Replaces cv::resize(mvImagePyramid[level-1], mvImagePyramid[level], sz, 0, 0, INTER_LINEAR) at reference
// src/ORBextractor.cc:1120 (OpenCV 2.4/3.2 generic 8UC1 path).  All float work (the Q11
// coefficient tables) is done once on the host with the exact float/double sequence; the kernel is pure integer:
//   H[r][dx] = S[r][sx]*a0 + S[r][sx+1]*a1                         (int32)
//   D[dy][dx] = (((b0*(H0>>4))>>16) + ((b1*(H1>>4))>>16) + 2) >> 2
//
// Work split: one 256-thread workgroup produces a 128x16 output tile.  The source footprint of the tile
// (<= 2*128+2 columns, <= 2*16+2 rows for scale factors in (1,2]) is staged in LDS with aligned dword loads of whole
// row segments (coalesced HBM rows), then every thread produces 8 horizontally adjacent pixels = one 8-byte store.
#pragma once
#include "common.h"

namespace orbfe {

constexpr int kRszTW = 128;
constexpr int kRszTH = 16;

struct RszTab {      // per output column / row
    int ofs;         // source index (x: clamped to [0, sw-1]; y: unclamped as in the reference)
    int coef;        // lo16 = weight of ofs, hi16 = weight of ofs+1  (Q11, sum ~2048)
};

struct ResizeArgs {
    const uint8_t *src; int sPitch; size_t sFrameStride; int sw, sh;
    uint8_t *dst; int dPitch; size_t dFrameStride; int dw, dh;
    const RszTab *xtab;   // padded to a multiple of kRszTW entries (last entry replicated)
    const RszTab *ytab;   // padded to a multiple of kRszTH entries
    int ldsPitch;         // bytes per staged source row (multiple of 4)
    int ldsRows;
};

__global__ __launch_bounds__(256) void pyr_resize_kernel(ResizeArgs a) {
    extern __shared__ __align__(16) uint8_t lds[];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * kRszTW, y0 = blockIdx.y * kRszTH;
    const int f = blockIdx.z;
    const uint8_t *src = a.src + (size_t)f * a.sFrameStride;
    uint8_t *dst = a.dst + (size_t)f * a.dFrameStride;

    // source footprint of this tile
    const int xl = min(x0 + kRszTW, a.dw) - 1, yl = min(y0 + kRszTH, a.dh) - 1;
    const int sx0 = a.xtab[x0].ofs;
    const int sx1 = min(a.xtab[xl].ofs + 1, a.sw - 1);
    const int sy0 = min(max(a.ytab[y0].ofs, 0), a.sh - 1);
    const int sy1 = min(a.ytab[yl].ofs + 1, a.sh - 1);
    const int ax0 = sx0 & ~3;
    const int nd = ((sx1 - ax0) >> 2) + 1;     // dwords per staged row
    const int nr = sy1 - sy0 + 1;
    // host guarantees nd*4 <= ldsPitch and nr <= ldsRows
    for (int i = tid; i < nd * nr; i += 256) {
        int r = i / nd, c = i - r * nd;
        uint32_t v = *reinterpret_cast<const uint32_t *>(src + (size_t)(sy0 + r) * a.sPitch + ax0 + 4 * c);
        *reinterpret_cast<uint32_t *>(lds + r * a.ldsPitch + 4 * c) = v;
    }
    __syncthreads();

    const int ry = tid >> 4, cx = (tid & 15) * 8;
    const int dy = y0 + ry;
    if (dy >= a.dh || x0 + cx >= a.dw) return;
    const RszTab yt = a.ytab[dy];
    const int r0 = min(max(yt.ofs, 0), a.sh - 1) - sy0;
    const int r1 = min(max(yt.ofs + 1, 0), a.sh - 1) - sy0;
    const int b0 = (short)(yt.coef & 0xffff), b1 = yt.coef >> 16;
    const uint8_t *S0 = lds + r0 * a.ldsPitch - ax0;
    const uint8_t *S1 = lds + r1 * a.ldsPitch - ax0;
    uint32_t out[2] = {0, 0};
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const RszTab xt = a.xtab[x0 + cx + i];
        const int sx = xt.ofs, sxp = min(sx + 1, a.sw - 1);   // when sx == sw-1 the right weight is 0
        const int a0 = (short)(xt.coef & 0xffff), a1 = xt.coef >> 16;
        const int h0 = S0[sx] * a0 + S0[sxp] * a1;
        const int h1 = S1[sx] * a0 + S1[sxp] * a1;
        const int d = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
        out[i >> 2] |= (uint32_t)(d & 0xff) << (8 * (i & 3));
    }
    // dPitch is a multiple of 64 and >= round_up(dw, 8): the full 8-byte store stays inside the row
    *reinterpret_cast<uint2 *>(dst + (size_t)dy * a.dPitch + x0 + cx) = make_uint2(out[0], out[1]);
}

}  // namespace orbfe
