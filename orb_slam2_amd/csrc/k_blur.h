// k_blur.h -- K5: 7x7 sigma=2 Gaussian blur, BORDER_REFLECT_101, 8-bit fixed point (gfx950).
//
// Replaces GaussianBlur(workingMat, workingMat, Size(7,7), 2, 2, BORDER_REFLECT_101) on a clone of each level at
// reference src/ORBextractor.cc:1085-1086 (OpenCV pre-3.4.1 generic 8UC1 path):
//   R[y][x]   = sum_i k[i] * src[y][refl(x+i-3)]                      k = {18,34,49,55,49,34,18} (Q8, sum 257)
//   dst[y][x] = sat_u8((sum_j k[j] * R[refl(y+j-3)][x] + 32768) >> 16)
//
// Register-sliding design, no LDS and no barriers.  One wavefront owns a vertical strip of 62 dword columns (248 px)
// x kBlurRows rows of one level of one frame and walks down it:
//   * per row every lane loads ONE aligned dword (a fully coalesced 256-B row segment); lanes 0 and 63 are the halo
//     columns, the left/right neighbour dwords arrive by DPP wave shifts, not by re-loading;
//   * row pass: the seven taps of the four pixels of a dword are two v_dot4_u32_u8 each on byte windows cut with
//     v_alignbyte_b32 (weights 18,34,49,55 | 49,34,18,0);
//   * column pass: the last seven row results stay in registers (the loop is unrolled by 7 so the window rotates by
//     renaming); the symmetric kernel needs 3 adds + 4 multiply-adds per pixel; one dword store per lane and row.
// Reflection is branch-free: the image's edge lanes patch their dwords with byte permutes (v_perm_b32).
// One launch covers every level of every frame: blockIdx.x walks a host-built (level, strip, row-block) list, 4 entries
// per 256-thread workgroup, blockIdx.y is the frame.
#pragma once
#include "common.h"

namespace orbfe {

constexpr int kBlurStripDw = 62;       // output dword columns per wave (lanes 1..62)
constexpr int kBlurRows = 64;          // output rows per wave; kBlurRows + 6 must be a multiple of 7
static_assert((kBlurRows + 6) % 7 == 0, "row loop is unrolled by 7");

struct BlurTile { uint16_t level, tx, ty, pad; };

__device__ __forceinline__ int refl101(int p, int n) {
    p = p < 0 ? -p : p;
    return p >= n ? 2 * (n - 1) - p : p;
}

__device__ __forceinline__ uint32_t dpp_lane_m1(uint32_t v) {   // lane i <- lane i-1 (wave_shr:1), 0 into lane 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t dpp_lane_p1(uint32_t v) {   // lane i <- lane i+1 (wave_shl:1), 0 into lane 63
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, false);
}

__global__ __launch_bounds__(256) void gauss7_kernel(Geom g, ImgSrc s, const BlurTile *tiles, int nTiles) {
    const int lane = threadIdx.x & 63;
    const int ti = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ti >= nTiles) return;
    const BlurTile t = tiles[ti];
    const int f = blockIdx.y;
    const LevelGeom &L = g.lv[t.level];
    int pitch;
    const uint8_t *img = level_ptr(s, g, f, t.level, &pitch);
    uint8_t *dst = s.blur + (size_t)f * g.blurFrameBytes + L.blurOff;
    const int w = L.w, h = L.h;
    const int x0 = 4 * (t.tx * kBlurStripDw + lane - 1);
    const int y0 = t.ty * kBlurRows;
    const bool writer = lane >= 1 && lane <= kBlurStripDw && x0 < w;

    // Horizontal BORDER_REFLECT_101 without byte loads or branches.  Every lane loads an aligned dword from a clamped,
    // always legal address; the two places where reflected pixels are needed are patched with byte permutes:
    //   left edge  (lane with x0 == 0): its left neighbour dword holds pixels (4,3,2,1) -> a permute of its own dword;
    //   right edge (lane holding the last pixel xe = w-1, byte r = xe & 3): bytes r+1..3 of its dword and the first
    //              bytes of its right neighbour are pixels xe-1, xe-2, ... taken from {own dword, left neighbour dword}.
    const int xe = w - 1, r = xe & 3;
    const bool isLeftEdge = x0 == 0, isRightEdge = x0 == (xe & ~3);
    uint32_t selC = 0, selR = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const int tc = b <= r ? b : 2 * r - b;             // source offset relative to the edge dword, -3..3
        const int tr = max(2 * r - 4 - b, -4);             // right neighbour: pixel xe - (4 + b - r); only b < r is used
        selC |= (uint32_t)(tc + 4) << (8 * b);
        selR |= (uint32_t)(tr + 4) << (8 * b);
    }
    const uint32_t selL = 0x01020300u;
    const uint32_t xl = (uint32_t)min(max(x0, 0), pitch - 4);

    constexpr uint32_t W1 = 18u | (34u << 8) | (49u << 16) | (55u << 24);
    constexpr uint32_t W2 = 49u | (34u << 8) | (18u << 16);
    uint32_t R[7][4];
#pragma unroll
    for (int u = 0; u < 7; u++)
#pragma unroll
        for (int k = 0; k < 4; k++) R[u][k] = 0;

    for (int tb = 0; tb < kBlurRows + 6; tb += 7) {
        if (y0 + tb - 6 >= h) break;   // nothing below the image (wave-uniform)
#pragma unroll
        for (int u = 0; u < 7; u++) {
            const int tt = tb + u;
            const int yy = refl101(min(y0 - 3 + tt, h + 2), h);
            const uint8_t *rowp = img + (size_t)yy * pitch;     // wave-uniform base, 32-bit per-lane offset
            const uint32_t Craw = *reinterpret_cast<const uint32_t *>(rowp + xl);
            const uint32_t Lraw = dpp_lane_m1(Craw);
            const uint32_t C = isRightEdge ? __builtin_amdgcn_perm(Craw, Lraw, selC) : Craw;
            const uint32_t Lf = isLeftEdge ? __builtin_amdgcn_perm(C, C, selL) : Lraw;   // lanes left of the right edge are unpatched
            const uint32_t Rnb = dpp_lane_p1(C);
            const uint32_t Rt = isRightEdge ? __builtin_amdgcn_perm(Craw, Lraw, selR) : Rnb;
            // row pass for the dword's four pixels: bytes x-3..x+3 of pixel x
            const uint32_t a0 = __builtin_amdgcn_alignbyte(C, Lf, 1), a1 = __builtin_amdgcn_alignbyte(C, Lf, 2),
                           a2 = __builtin_amdgcn_alignbyte(C, Lf, 3);
            const uint32_t b0 = __builtin_amdgcn_alignbyte(Rt, C, 1), b1 = __builtin_amdgcn_alignbyte(Rt, C, 2),
                           b2 = __builtin_amdgcn_alignbyte(Rt, C, 3);
            R[u][0] = __builtin_amdgcn_udot4(b0, W2, __builtin_amdgcn_udot4(a0, W1, 0u, false), false);
            R[u][1] = __builtin_amdgcn_udot4(b1, W2, __builtin_amdgcn_udot4(a1, W1, 0u, false), false);
            R[u][2] = __builtin_amdgcn_udot4(b2, W2, __builtin_amdgcn_udot4(a2, W1, 0u, false), false);
            R[u][3] = __builtin_amdgcn_udot4(Rt, W2, __builtin_amdgcn_udot4(C, W1, 0u, false), false);
            const int y = y0 + tt - 6;   // the row whose seven row results are now complete
            if (tt >= 6 && y < h) {      // wave-uniform
                uint32_t out = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    // window, oldest to newest: R[(u+1)%7] ... R[u]
                    // row results are <= 255*257 = 65535, pair sums < 2^17: 24-bit multiplies are exact
                    const uint32_t acc = __umul24(18u, R[(u + 1) % 7][k] + R[u][k]) + __umul24(34u, R[(u + 2) % 7][k] + R[(u + 6) % 7][k]) +
                                         __umul24(49u, R[(u + 3) % 7][k] + R[(u + 5) % 7][k]) + __umul24(55u, R[(u + 4) % 7][k]);
                    const uint32_t d = min((acc + 32768u) >> 16, 255u);
                    out |= d << (8 * k);
                }
                if (writer) *reinterpret_cast<uint32_t *>(dst + (size_t)y * L.pitch + x0) = out;   // pitch >= round_up(w,4)
            }
        }
    }
}

}  // namespace orbfe
