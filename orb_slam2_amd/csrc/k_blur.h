// k_blur.h -- K5: 7x7 sigma=2 Gaussian blur, BORDER_REFLECT_101, 8-bit fixed point (gfx950).
//
// Replaces GaussianBlur(workingMat, workingMat, Size(7,7), 2, 2, BORDER_REFLECT_101) on a clone of each level at
// reference src/ORBextractor.cc:1085-1086 (OpenCV pre-3.4.1 generic 8UC1 path):
//   R[y][x]   = sum_i k[i] * src[y][refl(x+i-3)]                      k = {18,34,49,55,49,34,18} (Q8, sum 257)
//   dst[y][x] = sat_u8((sum_j k[j] * R[refl(y+j-3)][x] + 32768) >> 16)
//
// The two passes commute exactly (no intermediate rounding; every partial sum is below 2^16 resp. 2^25), so the kernel
// runs the COLUMN pass first, where two pixels fit one register:
//   V[y][x]   = sum_j k[j] * src[refl(y+j-3)][x]                       <= 257*255 = 65535: packed u16, 2 pixels per op
//   dst[y][x] = sat_u8((sum_i k[i] * V[y][refl(x+i-3)] + 32768) >> 16)  v_dot2_u32_u16 on pairs of V
//
// Register-sliding design, no LDS and no barriers.  One wavefront owns a vertical strip of 62 dword columns (248 px)
// x kBlurRows rows of one level of one frame and walks down it:
//   * per row every lane loads ONE aligned dword (a fully coalesced 256-B row segment) and splits it into the pixel
//     pairs (p0,p2), (p1,p3) in 16-bit lanes; lanes 0 and 63 are the halo columns, except at the image's left and right
//     edge, where the edge lane reflects its own data and writes too (host-built strip list: BlurTile);
//   * column pass: the last seven rows stay in registers in pair form (the loop is unrolled by 7 so the window rotates
//     by renaming); the symmetric kernel needs 3 v_pk_add_u16 + 1 v_pk_mul_lo_u16 + 3 v_pk_mad_u16 per pixel pair;
//   * row pass: the left/right neighbour lanes' V pairs arrive by DPP wave shifts; the 7 taps of the dword's four pixels
//     are 18 v_dot2_u32_u16 (the rounding constant is the first accumulator); clamp, pack, one dword store per lane.
// Reflection is branch-free: the image's edge lanes patch their V pairs with byte permutes (v_perm_b32).
// One launch covers every level of every frame: blockIdx.x walks a host-built (level, strip, row-block) list, 4 entries
// per 256-thread workgroup, blockIdx.y is the frame.
#pragma once
#include "common.h"

namespace orbfe {

constexpr int kBlurStripDw = 62;       // output dword columns per wave (lanes 1..62)
constexpr int kBlurRows = 64;          // output rows per wave; kBlurRows + 6 must be a multiple of 7
static_assert((kBlurRows + 6) % 7 == 0, "row loop is unrolled by 7");

struct BlurTile { uint16_t level; int16_t xdw0; uint16_t ty; uint8_t lane0, lane1; };   // xdw0: dword column of lane 0; lanes lane0..lane1 write

__device__ __forceinline__ int refl101(int p, int n) {
    p = p < 0 ? -p : p;
    return p >= n ? 2 * (n - 1) - p : p;
}

__device__ __forceinline__ uint32_t dpp_lane_m1(uint32_t v) {   // lane i <- lane i-1 (wave_shr:1), 0 into lane 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t dpp_lane_p1(uint32_t v) {   // lane i <- lane i+1 (wave_shl:1), 0 into lane 63
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true);
}

typedef uint16_t blur_u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ blur_u16x2 bpk(uint32_t v) { return __builtin_bit_cast(blur_u16x2, v); }
__device__ __forceinline__ uint32_t bu32(blur_u16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t bdot2(uint32_t v, uint32_t w, uint32_t acc) {
    return __builtin_amdgcn_udot2(bpk(v), bpk(w), acc, false);
}
constexpr uint32_t bw(uint32_t lo, uint32_t hi) { return lo | (hi << 16); }

__global__ __launch_bounds__(256) void gauss7_kernel(Geom g, ImgSrc s, const BlurTile *tiles, int nTiles) {
    const int lane = threadIdx.x & 63;
    const int ti = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: tile record, row addresses and row tests on the scalar unit
    if (ti >= nTiles) return;
    const BlurTile t = tiles[ti];
    const int f = blockIdx.y;
    const LevelGeom &L = g.lv[t.level];
    int pitch;
    const uint8_t *img = level_ptr(s, g, f, t.level, &pitch);
    uint8_t *dst = s.blur + (size_t)f * g.blurFrameBytes + L.blurOff;
    const int w = L.w, h = L.h;
    const int x0 = 4 * (t.xdw0 + lane);
    const int y0 = t.ty * kBlurRows;
    const bool writer = lane >= t.lane0 && lane <= t.lane1 && x0 < w;

    // Horizontal BORDER_REFLECT_101 on the column sums V, without branches.  VE = (V0,V2), VO = (V1,V3) are the lane's
    // pairs, VEL/VOL the left neighbour's (V-4,V-2), (V-3,V-1), VER/VOR the right neighbour's (V4,V6), (V5,V7).
    //   left edge  (x0 == 0):  V-k = Vk         -> VOL = (V3,V1) = VO rotated, VEL = (., V2) = VE
    //   right edge (lane holding the last pixel xe = w-1, byte r = xe & 3):  V(r+k) = V(r-k); the pairs that contain
    //              pixels past xe are rebuilt from {own pair, left neighbour's pair} by one byte permute each.  Only
    //              outputs 0..r of that lane are real pixels; they read V up to r+3.
    const int xe = w - 1, r = xe & 3;
    const bool isLeftEdge = x0 == 0, isRightEdge = x0 == (xe & ~3);
    // v_perm_b32(left pair, own pair, sel): bytes 0..3 = own (lo half 0,1; hi half 2,3), bytes 4..7 = left neighbour's
    const uint32_t kSelE[4] = {0x07060100u, 0x01000100u, 0x03020100u, 0x03020100u};    // VE'
    const uint32_t kSelO[4] = {0x05040706u, 0x07060100u, 0x01000100u, 0x03020100u};    // VO'
    const uint32_t kSelER[4] = {0x03020100u, 0x03020706u, 0x03020100u, 0x01000302u};   // VER' (r = 0: unused)
    const uint32_t kSelOR[4] = {0x03020100u, 0x03020504u, 0x03020706u, 0x03020100u};   // VOR' (r = 0: unused)
    const uint32_t selE = isRightEdge ? kSelE[r] : 0x03020100u, selO = isRightEdge ? kSelO[r] : 0x03020100u;
    const uint32_t selER = kSelER[r], selOR = kSelOR[r];
    const uint32_t xl = (uint32_t)min(max(x0, 0), pitch - 4);
    const bool hasEdge = __ballot(isLeftEdge || isRightEdge) != 0;
    const int dPitch = L.pitch;                        // (L lives in the kernel arguments: read it once, not per row)
    const uint32_t xw = (uint32_t)max(x0, 0);          // writers have x0 >= 0

    uint32_t E[7], O[7];
#pragma unroll
    for (int u = 0; u < 7; u++) E[u] = O[u] = 0;
    const blur_u16x2 k18 = bpk(bw(18, 18)), k34 = bpk(bw(34, 34)), k49 = bpk(bw(49, 49)), k55 = bpk(bw(55, 55));

    // seven source rows are fetched at a time, one block ahead of the arithmetic: 7 loads per wave stay in flight
    auto load7 = [&](int tb, uint32_t (&raw)[7]) {
#pragma unroll
        for (int u = 0; u < 7; u++) {
            const int yy = refl101(min(y0 - 3 + tb + u, h + 2), h);
            raw[u] = *reinterpret_cast<const uint32_t *>(img + (size_t)yy * pitch + xl);   // wave-uniform row, 32-bit lane offset
        }
    };
    uint32_t cur[7], nxt[7];
    load7(0, cur);
    for (int tb = 0; tb < kBlurRows + 6; tb += 7) {
        if (y0 + tb - 6 >= h) break;   // nothing below the image (wave-uniform)
        if (tb + 7 < kBlurRows + 6 && y0 + tb + 1 < h) load7(tb + 7, nxt);
#pragma unroll
        for (int u = 0; u < 7; u++) {
            const int tt = tb + u;
            const uint32_t raw = cur[u];
            E[u] = raw & 0x00FF00FFu;                                  // (p0, p2)
            O[u] = __builtin_amdgcn_perm(0u, raw, 0x0c030c01u);        // (p1, p3)
            const int y = y0 + tt - 6;   // the row whose seven source rows are now in the window
            if (tt >= 6 && y < h) {      // wave-uniform
                // column pass; window, oldest to newest: slot (u+1)%7 ... slot u
                const int s0 = (u + 1) % 7, s1 = (u + 2) % 7, s2 = (u + 3) % 7, s3 = (u + 4) % 7, s4 = (u + 5) % 7, s5 = (u + 6) % 7;
                blur_u16x2 ve = bpk(E[s3]) * k55, vo = bpk(O[s3]) * k55;
                ve += (bpk(E[s0]) + bpk(E[u])) * k18;
                vo += (bpk(O[s0]) + bpk(O[u])) * k18;
                ve += (bpk(E[s1]) + bpk(E[s5])) * k34;
                vo += (bpk(O[s1]) + bpk(O[s5])) * k34;
                ve += (bpk(E[s2]) + bpk(E[s4])) * k49;
                vo += (bpk(O[s2]) + bpk(O[s4])) * k49;
                const uint32_t VE0 = bu32(ve), VO0 = bu32(vo);
                uint32_t VEL = dpp_lane_m1(VE0), VOL = dpp_lane_m1(VO0);
                uint32_t VE = VE0, VO = VO0, VER, VOR;
                if (hasEdge) {   // wave-uniform: only the first and the last strip of a level hold an image edge
                    VOL = isLeftEdge ? __builtin_amdgcn_alignbit(VO0, VO0, 16) : VOL;
                    VEL = isLeftEdge ? VE0 : VEL;
                    VE = __builtin_amdgcn_perm(VEL, VE0, selE);
                    VO = __builtin_amdgcn_perm(VOL, VO0, selO);
                    VER = dpp_lane_p1(VE);
                    VOR = dpp_lane_p1(VO);
                    VER = isRightEdge ? __builtin_amdgcn_perm(VEL, VE0, selER) : VER;
                    VOR = isRightEdge ? __builtin_amdgcn_perm(VOL, VO0, selOR) : VOR;
                } else {
                    VER = dpp_lane_p1(VE0);
                    VOR = dpp_lane_p1(VO0);
                }
                // row pass: out_j = sum_i k[i] * V[j+i-3] + 32768
                uint32_t o0 = bdot2(VOL, bw(18, 49), 32768u);
                o0 = bdot2(VEL, bw(0, 34), o0);
                o0 = bdot2(VE, bw(55, 34), o0);
                o0 = bdot2(VO, bw(49, 18), o0);
                uint32_t o1 = bdot2(VEL, bw(0, 18), 32768u);
                o1 = bdot2(VOL, bw(0, 34), o1);
                o1 = bdot2(VE, bw(49, 49), o1);
                o1 = bdot2(VO, bw(55, 34), o1);
                o1 = bdot2(VER, bw(18, 0), o1);
                uint32_t o2 = bdot2(VOL, bw(0, 18), 32768u);
                o2 = bdot2(VE, bw(34, 55), o2);
                o2 = bdot2(VO, bw(49, 49), o2);
                o2 = bdot2(VER, bw(34, 0), o2);
                o2 = bdot2(VOR, bw(18, 0), o2);
                uint32_t o3 = bdot2(VE, bw(18, 49), 32768u);
                o3 = bdot2(VO, bw(34, 55), o3);
                o3 = bdot2(VER, bw(49, 18), o3);
                o3 = bdot2(VOR, bw(34, 0), o3);
                // o >> 16 (<= 257) of two pixels side by side in 16-bit halves, saturated to 255 by one packed minimum per
                // pair, then the four low bytes into one dword
                const blur_u16x2 k255 = bpk(bw(255, 255));
                const blur_u16x2 h01 = __builtin_elementwise_min(bpk(__builtin_amdgcn_perm(o1, o0, 0x07060302u)), k255);
                const blur_u16x2 h23 = __builtin_elementwise_min(bpk(__builtin_amdgcn_perm(o3, o2, 0x07060302u)), k255);
                const uint32_t out = __builtin_amdgcn_perm(bu32(h23), bu32(h01), 0x06040200u);
                if (writer) *reinterpret_cast<uint32_t *>(dst + (size_t)y * dPitch + (size_t)xw) = out;   // wave-uniform row + 32-bit lane offset; pitch >= round_up(w,4)
            }
        }
#pragma unroll
        for (int u = 0; u < 7; u++) cur[u] = nxt[u];
    }
}

}  // namespace orbfe
