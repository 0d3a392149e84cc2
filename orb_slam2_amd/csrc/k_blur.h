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
// Reflection only costs on the image's own edges (edge lanes assemble their dword from reflected bytes).
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

__device__ __forceinline__ uint32_t blur_load_dword(const uint8_t *row, int x0, int w) {
    if (x0 >= 0 && x0 + 3 < w) return *reinterpret_cast<const uint32_t *>(row + x0);
    uint32_t v = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
        int xx = x0 + b;
        // columns further than 3 px outside the image never reach a valid output; the clamp only keeps the read legal
        xx = xx < -3 ? 0 : (xx > w + 2 ? w - 1 : refl101(xx, w));
        v |= (uint32_t)row[xx] << (8 * b);
    }
    return v;
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
            const uint32_t C = blur_load_dword(img + (size_t)yy * pitch, x0, w);
            // neighbours: lane-1 / lane+1 (wave_shr:1 / wave_shl:1); the wave's edge lanes are halo and never write
            const uint32_t Lf = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)C, 0x138, 0xf, 0xf, false);
            const uint32_t Rt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)C, 0x130, 0xf, 0xf, false);
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
            if (tt >= 6 && y < h && writer) {
                uint32_t out = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    // window, oldest to newest: R[(u+1)%7] ... R[u]
                    // row results are <= 255*257 = 65535, pair sums < 2^17: 24-bit multiplies (full rate) are exact
                    const uint32_t acc = __umul24(18u, R[(u + 1) % 7][k] + R[u][k]) + __umul24(34u, R[(u + 2) % 7][k] + R[(u + 6) % 7][k]) +
                                         __umul24(49u, R[(u + 3) % 7][k] + R[(u + 5) % 7][k]) + __umul24(55u, R[(u + 4) % 7][k]);
                    const uint32_t d = min((acc + 32768u) >> 16, 255u);
                    out |= d << (8 * k);
                }
                *reinterpret_cast<uint32_t *>(dst + (size_t)y * L.pitch + x0) = out;   // pitch >= round_up(w,4)
            }
        }
    }
}

}  // namespace orbfe
