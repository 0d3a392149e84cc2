// k_blur.h -- K5: 7x7 sigma=2 Gaussian blur, BORDER_REFLECT_101, 8-bit fixed point (gfx950).
//
// Replaces GaussianBlur(workingMat, workingMat, Size(7,7), 2, 2, BORDER_REFLECT_101) on a clone of each level at
// reference src/ORBextractor.cc:1085-1086 (OpenCV pre-3.4.1 generic 8UC1 path):
//   R[y][x]   = sum_i k[i] * src[y][refl(x+i-3)]                      k = {18,34,49,55,49,34,18} (Q8, sum 257)
//   dst[y][x] = sat_u8((sum_j k[j] * R[refl(y+j-3)][x] + 32768) >> 16)
// One launch covers every level of every frame of the batch: blockIdx.x walks a host-built tile list
// (level, tile x, tile y), blockIdx.y is the frame.  A 256-thread workgroup stages a (128+8) x (16+6) source window
// in LDS with aligned dword loads (reflection only on border tiles), runs the row pass into a u16 LDS plane
// (R <= 255*257 fits 16 bits) and the column pass from there; each thread stores 4 adjacent pixels as one dword.
#pragma once
#include "common.h"

namespace orbfe {

constexpr int kBlurTW = 128;
constexpr int kBlurTH = 16;
constexpr int kBlurSW = kBlurTW + 8;    // staged columns: x0-4 .. x0+TW+3
constexpr int kBlurSH = kBlurTH + 6;    // staged rows:    y0-3 .. y0+TH+2

struct BlurTile { uint16_t level, tx, ty, pad; };

__device__ __forceinline__ int refl101(int p, int n) {
    p = p < 0 ? -p : p;
    return p >= n ? 2 * (n - 1) - p : p;
}

__global__ __launch_bounds__(256) void gauss7_kernel(Geom g, ImgSrc s, const BlurTile *tiles) {
    __shared__ __align__(16) uint8_t src[kBlurSH][kBlurSW];
    __shared__ __align__(16) uint16_t rowp[kBlurSH][kBlurTW];
    const int tid = threadIdx.x;
    const BlurTile t = tiles[blockIdx.x];
    const int f = blockIdx.y;
    const LevelGeom &L = g.lv[t.level];
    int pitch;
    const uint8_t *img = level_ptr(s, g, f, t.level, &pitch);
    uint8_t *dst = s.blur + (size_t)f * g.blurFrameBytes + L.blurOff;
    const int w = L.w, h = L.h;
    const int x0 = t.tx * kBlurTW, y0 = t.ty * kBlurTH;

    // stage: kBlurSH rows x (kBlurSW/4 = 34) dwords
    for (int i = tid; i < kBlurSH * (kBlurSW / 4); i += 256) {
        const int r = i / (kBlurSW / 4), c = i - r * (kBlurSW / 4);
        const int sy = refl101(y0 - 3 + r, h);
        const int sx = x0 - 4 + 4 * c;
        uint32_t v;
        if (sx >= 0 && sx + 3 < w) {
            v = *reinterpret_cast<const uint32_t *>(img + (size_t)sy * pitch + sx);
        } else {
            v = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                int xx = sx + b;
                // columns further than 3 px outside the image are never used by a valid output; clamp keeps the read legal
                xx = xx < -3 ? 0 : (xx > w + 2 ? w - 1 : refl101(xx, w));
                v |= (uint32_t)img[(size_t)sy * pitch + xx] << (8 * b);
            }
        }
        *reinterpret_cast<uint32_t *>(&src[r][4 * c]) = v;
    }
    __syncthreads();

    // row pass: kBlurSH x kBlurTW values, 4 adjacent per thread-iteration
    for (int i = tid; i < kBlurSH * (kBlurTW / 4); i += 256) {
        const int r = i / (kBlurTW / 4), c4 = (i - r * (kBlurTW / 4)) * 4;
        const uint8_t *p = &src[r][c4 + 1];   // tap x-3 of output column c4 sits at staged column c4+4-3
        int v[10];
#pragma unroll
        for (int k = 0; k < 10; k++) v[k] = p[k];
        uint16_t o[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
            o[k] = (uint16_t)(18 * (v[k] + v[k + 6]) + 34 * (v[k + 1] + v[k + 5]) + 49 * (v[k + 2] + v[k + 4]) + 55 * v[k + 3]);
        *reinterpret_cast<uint2 *>(&rowp[r][c4]) =
            make_uint2((uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16));
    }
    __syncthreads();

    // column pass: kBlurTH x kBlurTW outputs, 4 adjacent per thread-iteration
    for (int i = tid; i < kBlurTH * (kBlurTW / 4); i += 256) {
        const int r = i / (kBlurTW / 4), c4 = (i - r * (kBlurTW / 4)) * 4;
        const int y = y0 + r, x = x0 + c4;
        if (y >= h || x >= w) continue;
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int acc = 18 * (rowp[r][c4 + k] + rowp[r + 6][c4 + k]) + 34 * (rowp[r + 1][c4 + k] + rowp[r + 5][c4 + k]) +
                      49 * (rowp[r + 2][c4 + k] + rowp[r + 4][c4 + k]) + 55 * rowp[r + 3][c4 + k];
            int d = (acc + 32768) >> 16;
            d = d > 255 ? 255 : d;
            out |= (uint32_t)d << (8 * k);
        }
        *reinterpret_cast<uint32_t *>(dst + (size_t)y * L.pitch + x) = out;   // pitch >= round_up(w,4)
    }
}

}  // namespace orbfe
