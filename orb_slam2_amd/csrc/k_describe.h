// k_describe.h -- K4 + K6: intensity-centroid orientation and 256-bit rBRIEF (gfx950).
//
// Replaces IC_Angle / computeOrientation (reference src/ORBextractor.cc:77-104, 472-479, 851-852), computeOrbDescriptor /
// computeDescriptors (:108-147, 1034-1041) and the output assembly of operator() (:1072-1104: pt *= scale for level > 0,
// level-major concatenation).
//
// One wavefront handles kDescKpPerWave (8) consecutive keypoints of a frame, in three phases, so that the per-keypoint
// scalar math (fastAtan2, cos, sin) is done ONCE for the 8 keypoints with one keypoint per lane instead of 64 times
// redundantly.  8 rather than 16 per wave: the keypoints resident on an XCD at one time then belong to 3 frames instead of
// 6 and more of the patches stay in its 4-MiB L2 (measured TCC misses per 512-frame launch: 29.3 M at 16 keypoints per
// wave, 12.2 M at 4); 8 % faster than 16 although the per-wave fixed work is spread over half as many keypoints, and
// as fast as 4, where that fixed work eats the gain.
//   1. moments, one keypoint at a time: the 31 disc rows of the UN-blurred level are fetched as 31 x 9 aligned dwords
//      (5 per lane); per dword two v_dot4_u32_u8 against a mask / weight pair from an LDS table indexed by the patch's
//      alignment (cx-15)&3 give sum(I) and sum((u+18) I) over the bytes inside the disc; m01 += v*sum(I); a DPP
//      reduction folds the 64 lanes and the totals land in lane k of (m10, m01);
//   2. lanes 0..15: angle = cv::fastAtan2's polynomial in float32 with every operation individually rounded, a = cosf and
//      b = sinf by glibc's algorithm (sincosf_glibc below: the reference calls libm cosf/sinf, and the device's floats
//      equal glibc 2.35's for every float angle, DESIGN.md section 2 pin 3); the KeyPoint records are written here;
//   3. descriptor, one keypoint at a time: the 37 x 37 window of the blurred level that the rotated taps can reach
//      (|tap| <= round(18.38) = 18) is staged into LDS with 6 row-coalesced dword loads per lane, then lane i evaluates
//      test pairs 4i..4i+3: tap offsets are cvRound(px*b + py*a), cvRound(px*a - py*b) with un-fused float
//      multiplies/adds and round-half-even conversion, the 8 taps are LDS byte reads (no scattered global gathers).
#pragma once
#include "common.h"

namespace orbfe {

struct DescribeArgs {
    Geom g;
    ImgSrc s;
    const SelKp *sel;        // [F][selFrameCap], level segments at lv[l].selOff
    const int *selCount;     // [F][kMaxLevels]
    KeyPointOut *kps;        // [F][cap]
    uint8_t *desc;           // [F][cap][32]
    int *counts;             // [F]
    int cap;
    int nFrames, groupsPerFrame;
    const uint4 *maskTab;    // desc_mask_table(): [4 alignments][2 slots][64 lanes][4 dwords] x (01-mask, (u+18)-weights)
};

__constant__ __attribute__((aligned(16))) int8_t c_pattern[1024] = {
#include "pattern31.inc"
};

__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    // cv::fastAtan2 (OpenCV 2.4/3.x scalar path).  p1..p7 = coefficient * (float)(180/CV_PI), folded by the compiler in
    // float exactly as the reference's static initialisers do.
    const float k = (float)(180.0 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * k, p3 = -0.3258083974640975f * k, p5 = 0.1555786518463281f * k,
                p7 = -0.04432655554792128f * k;
    const float eps = 2.2204460492503131e-16f;   // (float)DBL_EPSILON
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, eps));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, eps));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// sum over the 64 lanes, result wave-uniform: four DPP butterfly steps give every row of 16 its sum, then four readlanes
__device__ __forceinline__ int wave_sum_uniform(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);   // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);   // row_mirror
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}

// cosf(y) and sinf(y) as glibc 2.35 computes them (sysdeps/ieee754/flt-32/s_sincosf.h, s_sinf.c, s_cosf.c: the reference
// calls them at src/ORBextractor.cc:112-113), for 0 <= y < 2^23: reduction n = round(y * 2/pi) through the 2^24-scaled
// product, x = y - n * (pi/2) in double, then the degree-7 odd / degree-8 even polynomials in double and ONE rounding to
// float.  The quadrant sign enters glibc's polynomials through x*sign and a table with negated cosine coefficients; both
// are exact negations of the value computed here, so they are applied to the result.  tools/sincos_census.c checks the host
// mirror of this routine against libm over every float angle in [0, 360): 0 of 1 135 869 952 differ (with and without
// fused multiply-adds: the contraction glibc's build may apply does not change any rounded float on this domain).
__device__ __forceinline__ void sincosf_glibc(float y, float *sOut, float *cOut) {
    const double hpiInv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5, c3 = -0x1.6c087e89a359dp-10,
                 c4 = 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    const double xd = (double)y;
    const int n = ((int)(xd * hpiInv) + 0x800000) >> 24;      // 0 for y < pi/4: glibc's no-reduction branch is the same code
    const double x = __builtin_fma(-(double)n, hpi, xd);
    const double x2 = x * x;
    // sine polynomial
    const double x3 = x * x2;
    const double sa = __builtin_fma(x2, s3, s2);
    const double x7 = x3 * x2;
    const double sb = __builtin_fma(x3, s1, x);
    float S = (float)__builtin_fma(x7, sa, sb);
    // cosine polynomial
    const double x4 = x2 * x2;
    const double ca = __builtin_fma(x2, c4, c3);
    const double cb = __builtin_fma(x2, c2, c1);
    const double x6 = x4 * x2;
    const double cc = __builtin_fma(x2, cb, c0);
    float C = (float)__builtin_fma(x6, ca, cc);
    // |y| < 2^-12: glibc returns y and 1.0f without evaluating
    const bool tiny = ((__float_as_uint(y) >> 20) & 0x7ff) < ((__float_as_uint(0x1p-12f) >> 20) & 0x7ff);
    S = tiny ? y : S;
    C = tiny ? 1.0f : C;
    const float sq = (n & 1) ? C : S, cq = (n & 1) ? S : C;
    *sOut = (n & 2) ? -sq : sq;
    *cOut = ((n + 1) & 2) ? -cq : cq;
}

// The rotation the descriptor kernel derives from a keypoint angle, alone: out[i] = (cos, sin) of angleDeg[i] * factorPI,
// computed by the same device function (orbfe_debug_rotation; the census angles of tests/golden/sincos_census.json).
__global__ void __launch_bounds__(256) debug_rotation_kernel(const float *angleDeg, int n, float2 *out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    float s, c;
    sincosf_glibc(__fmul_rn(angleDeg[i], factorPI), &s, &c);
    out[i] = make_float2(c, s);
}

constexpr int kDescKpPerWave = 8;
constexpr int kDescKpPerGroup = 4 * kDescKpPerWave;
constexpr int kDescWinR = 18;                              // rotated taps reach at most round(hypot(13,13)) = 18 pixels
constexpr int kDescWinRows = 2 * kDescWinR + 1;            // 37
constexpr int kDescWinDw = 12;                             // LDS row: 3 x 16 bytes (37 bytes + up to 3 of alignment slack used)
constexpr int kDescWinItems = kDescWinRows * 3;            // 16-byte loads of the blurred window (111: two per lane)
constexpr int kDescMomentItems = 31 * 3;                   // 16-byte loads of the 31 x 36 moment window (93: two per lane)

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int rl(int v, int k) { return __builtin_amdgcn_readlane(v, k); }

// (int16)a * (int16)b + c
__device__ __forceinline__ int mad_i16(int a, int b, int c) {
    int d;
    asm("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

typedef const __attribute__((address_space(1))) uint8_t *gptr_u8;
// dword at (wave-uniform base) + (32-bit per-lane offset): global_load_dword with an SGPR base
__device__ __forceinline__ uint32_t gload_u32(gptr_u8 base, uint32_t off) {
    return *reinterpret_cast<const __attribute__((address_space(1))) uint32_t *>(base + off);
}
__device__ __forceinline__ uint4 gload_u32x4(gptr_u8 base, uint32_t off) {   // dword-aligned 16 bytes
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    struct __attribute__((packed, aligned(4))) V { u32x4 v; };
    const u32x4 v = reinterpret_cast<const __attribute__((address_space(1))) V *>(base + off)->v;
    return make_uint4(v.x, v.y, v.z, v.w);
}
// wave-uniform global pointer from two lanes' halves
__device__ __forceinline__ gptr_u8 gptr_from(int lo, int hi) {
    return (gptr_u8)(((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo);
}

constexpr int kDescMaskTabU4 = 4 * 2 * 64 * 4 / 2;   // the table in 16-byte units (1024)

// The 31 x 36-byte moment window is fetched as three 16-byte pieces per row, at dwords 0, 4 and 5 of the row (the third
// overlaps the second so that nothing is read past byte 35; the overlapped dwords get zero masks).  For patch alignment
// al = (cx - 15) & 3, load item it = lane + 64 * slot (row it / 3, piece it % 3) and dword q of the piece, the table
// holds for the 4 bytes: 1 where the pixel lies inside the disc (|u| <= umax[|v|]) and u + 18 there, 0 elsewhere.
inline void desc_mask_table(const int *umax, uint2 *tab) {
    for (int al = 0; al < 4; al++)
        for (int slot = 0; slot < 2; slot++)
            for (int ln = 0; ln < 64; ln++) {
                const int it = ln + 64 * slot, row = it / 3, ch = it - row * 3, v = row - kHalfPatch;
                const int dmax = it < kDescMomentItems ? umax[v < 0 ? -v : v] : -1;
                const int dw0 = ch == 0 ? 0 : ch == 1 ? 4 : 5;
                for (int q = 0; q < 4; q++) {
                    const int dc = dw0 + q;
                    const bool dup = ch == 2 && q < 3;   // dwords 5..7 already belong to piece 1
                    uint32_t m01 = 0, mU = 0;
                    for (int b = 0; b < 4; b++) {
                        const int u = 4 * dc - kHalfPatch - al + b;
                        const bool ok = !dup && u >= -dmax && u <= dmax;
                        m01 |= (ok ? 1u : 0u) << (8 * b);
                        mU |= (ok ? (uint32_t)(u + 18) : 0u) << (8 * b);
                    }
                    tab[((al * 2 + slot) * 64 + ln) * 4 + q] = make_uint2(m01, mU);
                }
            }
}

__global__ __launch_bounds__(256) void orient_describe_kernel(DescribeArgs a) {
    // [alignment][slot][lane][dword]: (01-mask, (u+18)-weights) of the disc for the dwords of the lane's 16-byte load
    __shared__ __align__(16) uint2 sMask[4][2][64][4];
    __shared__ __align__(16) uint32_t sWin[4][kDescWinRows * kDescWinDw];   // per wave: the blurred window, 48-byte rows
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // XCD-aware block mapping: workgroups are dealt round-robin over the 8 XCDs (private L2s), so block id % 8 selects
    // the XCD.  All blocks of frame f get ids with id % 8 == f % 8: a frame's patches then stay in ONE 4-MiB L2 instead
    // of being pulled into all eight.
    const int bid = blockIdx.x;
    const int f = 8 * ((bid >> 3) / a.groupsPerFrame) + (bid & 7);
    const int grp = (bid >> 3) % a.groupsPerFrame;
    if (f >= a.nFrames) return;   // whole workgroup

    // the disc's mask / weight table (desc_mask_table below, built once on the host): 16 KB into LDS
    {
        uint4 *dst = reinterpret_cast<uint4 *>(&sMask[0][0][0][0]);
#pragma unroll
        for (int k = 0; k < kDescMaskTabU4 / 256; k++) dst[threadIdx.x + 256 * k] = a.maskTab[threadIdx.x + 256 * k];
    }
    __syncthreads();

    // ---- the wave's keypoints, one per lane (the other lanes mirror them): level, slot, geometry
    const int iBase = grp * kDescKpPerGroup + wv * kDescKpPerWave;
    const int i = iBase + (lane & (kDescKpPerWave - 1));   // keypoint index inside the frame, level-major
    int lvl = -1, j = i, total = 0, selOff = 0, pitchV = 0, bpV = 0, wV = 2 * kEdge + 1, hV = 2 * kEdge + 1;
    unsigned long long imgV = 0, blurV = 0;
    float scaleV = 1.f, patchV = 0.f;
    for (int l = 0; l < a.g.nlevels; l++) {
        const LevelGeom &L = a.g.lv[l];
        const int n = min(a.selCount[f * kMaxLevels + l], L.selCap);
        if (lvl < 0 && j < n) {
            lvl = l;
            wV = L.w;
            hV = L.h;
            selOff = L.selOff;
            int pl;
            imgV = (unsigned long long)level_ptr(a.s, a.g, f, l, &pl);
            pitchV = pl;
            blurV = (unsigned long long)(a.s.blur + (size_t)f * a.g.blurFrameBytes + L.blurOff);
            bpV = L.pitch;
            scaleV = L.scale;
            patchV = L.patch;
        }
        if (lvl < 0) j -= n;
        total += n;
    }
    if (iBase == 0 && lane == 0) a.counts[f] = total;
    const bool valid = lvl >= 0 && i < a.cap;
    const int nValid = __popcll(__ballot(valid && lane < kDescKpPerWave));   // valid keypoints are a prefix
    if (nValid == 0) return;
    SelKp kp = {};
    if (valid) kp = a.sel[(size_t)f * a.g.selFrameCap + selOff + j];
    // keypoints lie in [19, w-20] x [19, h-20] by construction; the clamp only makes a corrupted entry harmless (no
    // out-of-bounds read) instead of a GPU fault
    const int cxV = min(max((int)kp.x, kEdge), wV - kEdge - 1), cyV = min(max((int)kp.y, kEdge), hV - kEdge - 1);
    const int imgLo = (int)(uint32_t)imgV, imgHi = (int)(uint32_t)(imgV >> 32);
    const int blurLo = (int)(uint32_t)blurV, blurHi = (int)(uint32_t)(blurV >> 32);

    // ---- phase 1: moments
    int mRow[2], mOff[2], mV[2];
#pragma unroll
    for (int sl = 0; sl < 2; sl++) {
        const int it = lane + 64 * sl, row = it / 3, ch = it - row * 3;
        const bool in = it < kDescMomentItems;
        mRow[sl] = in ? row : 0;          // items past the window read a legal address and carry zero masks
        mOff[sl] = in ? (ch == 0 ? 0 : ch == 1 ? 16 : 20) : 0;
        mV[sl] = in ? row - kHalfPatch : 0;
        asm volatile("" : "+v"(mRow[sl]), "+v"(mOff[sl]), "+v"(mV[sl]));   // hold them, do not re-derive per keypoint
    }
    int m10V = 0, m01V = 0;
    auto moment_loads = [&](int k, uint4 (&dw)[2]) {
        const int cx = rl(cxV, k), cy = rl(cyV, k), pitch = rl(pitchV, k);
        const gptr_u8 base = gptr_from(rl(imgLo, k), rl(imgHi, k)) + (size_t)(cy - kHalfPatch) * pitch +
                             ((cx - kHalfPatch) & ~3);   // xa >= 4; xa + 35 <= cx + 20 <= w
#pragma unroll
        for (int sl = 0; sl < 2; sl++)
            dw[sl] = gload_u32x4(base, __umul24((uint32_t)mRow[sl], (uint32_t)pitch) + (uint32_t)mOff[sl]);
    };
    {
        uint4 dw[2], nx[2];
        moment_loads(0, dw);
        for (int k = 0; k < nValid; k++) {
            if (k + 1 < nValid) moment_loads(k + 1, nx);
            const int al = (rl(cxV, k) - kHalfPatch) & 3;
            uint32_t A = 0;
            int rsSum = 0, m01 = 0;
#pragma unroll
            for (int sl = 0; sl < 2; sl++) {
                const uint4 *mk = reinterpret_cast<const uint4 *>(sMask[al][sl][lane]);
                const uint4 m0 = mk[0], m1 = mk[1];   // (m01, mU) of dwords 0,1 and 2,3
                uint32_t rs = __builtin_amdgcn_udot4(dw[sl].x, m0.x, 0u, false);
                rs = __builtin_amdgcn_udot4(dw[sl].y, m0.z, rs, false);
                rs = __builtin_amdgcn_udot4(dw[sl].z, m1.x, rs, false);
                rs = __builtin_amdgcn_udot4(dw[sl].w, m1.z, rs, false);
                A = __builtin_amdgcn_udot4(dw[sl].x, m0.y, A, false);
                A = __builtin_amdgcn_udot4(dw[sl].y, m0.w, A, false);
                A = __builtin_amdgcn_udot4(dw[sl].z, m1.y, A, false);
                A = __builtin_amdgcn_udot4(dw[sl].w, m1.w, A, false);
                rsSum += (int)rs;
                m01 += __mul24(mV[sl], (int)rs);
            }
            const int m10 = (int)A - 18 * rsSum;
            const int s10 = wave_sum_uniform(m10), s01 = wave_sum_uniform(m01);
            m10V = lane == k ? s10 : m10V;
            m01V = lane == k ? s01 : m01V;
#pragma unroll
            for (int sl = 0; sl < 2; sl++) dw[sl] = nx[sl];
        }
    }

    // ---- phase 2: one keypoint per lane
    const float angle = fast_atan2_deg((float)m01V, (float)m10V);
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    const float ang = __fmul_rn(angle, factorPI);
    float caV, sbV;
    sincosf_glibc(ang, &sbV, &caV);
    if (valid && lane < kDescKpPerWave) {
        KeyPointOut o;
        o.x = (float)cxV;
        o.y = (float)cyV;
        if (lvl != 0) { o.x = __fmul_rn(o.x, scaleV); o.y = __fmul_rn(o.y, scaleV); }
        o.size = patchV;
        o.angle = angle;
        o.response = (float)kp.resp;
        o.octave = lvl;
        o.class_id = -1;
        a.kps[(size_t)f * a.cap + i] = o;
    }

    // ---- phase 3: rBRIEF on the blurred level
    // the 37 x 40-byte window of the blurred level: three 16-byte pieces per row (the blurred rows have >= 16 bytes of
    // pitch slack, so the last piece may run past byte 39), written to 48-byte LDS rows
    int wRow[2], wOff[2];
#pragma unroll
    for (int sl = 0; sl < 2; sl++) {
        const int it = lane + 64 * sl, row = it / 3, ch = it - row * 3;
        const bool in = it < kDescWinItems;
        wRow[sl] = in ? row : 0;
        wOff[sl] = in ? 16 * ch : 0;
        asm volatile("" : "+v"(wRow[sl]), "+v"(wOff[sl]));
    }
    float px0[4], py0[4], px1[4], py1[4];
    {
        const uint32_t *pat32 = reinterpret_cast<const uint32_t *>(c_pattern);
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const uint32_t pp = pat32[lane * 4 + t];   // (x0, y0, x1, y1) as int8
            px0[t] = (float)(int8_t)(pp & 0xFF);
            py0[t] = (float)(int8_t)((pp >> 8) & 0xFF);
            px1[t] = (float)(int8_t)((pp >> 16) & 0xFF);
            py1[t] = (float)(int8_t)(pp >> 24);
        }
    }
    const int caI = __builtin_bit_cast(int, caV), sbI = __builtin_bit_cast(int, sbV);
    auto window_loads = [&](int k, uint4 (&dw)[2]) {
        const int cx = rl(cxV, k), cy = rl(cyV, k), bp = rl(bpV, k);
        const int xa = (cx - kDescWinR) & ~3;           // >= 0; xa + 47 <= w + 9 < pitch (common.h: kPitchSlack)
        const gptr_u8 base = gptr_from(rl(blurLo, k), rl(blurHi, k)) + (size_t)(cy - kDescWinR) * bp + xa;
#pragma unroll
        for (int sl = 0; sl < 2; sl++) dw[sl] = gload_u32x4(base, __umul24((uint32_t)wRow[sl], (uint32_t)bp) + (uint32_t)wOff[sl]);
    };
    {
        uint4 dw[2], nx[2];
        window_loads(0, dw);
        const uint8_t *win = reinterpret_cast<const uint8_t *>(sWin[wv]);
        for (int k = 0; k < nValid; k++) {
            {
                uint4 *w4 = reinterpret_cast<uint4 *>(sWin[wv]);   // item it = 3 * row + piece sits at 16 * it bytes
                w4[lane] = dw[0];
                if (lane + 64 < kDescWinItems) w4[lane + 64] = dw[1];
            }
            if (k + 1 < nValid) window_loads(k + 1, nx);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const float ca = __builtin_bit_cast(float, rl(caI, k)), sb = __builtin_bit_cast(float, rl(sbI, k));
            const f32x2 sc = {sb, ca}, cs = {ca, -sb};   // py * (-sb) = -(py * sb) exactly
            const int cx = rl(cxV, k);
            const int origin = kDescWinR * (4 * kDescWinDw) + cx - ((cx - kDescWinR) & ~3);   // byte of the keypoint in the window
            const int originM = origin - 0x4B400000;                                          // minus the bits of 1.5*2^23
            uint32_t nib = 0;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                // cvRound by the 1.5*2^23 trick: for |x| < 2^22 the float add rounds x to the nearest-even integer and leaves
                // it in the low mantissa bits; the tap's byte address r*40 + c + origin then takes one v_mad_i32_i16
                // on the low halves (bits 0..15 of the sum are r as int16) plus one add
                // two floats per instruction (v_pk_mul_f32 / v_pk_add_f32, each half individually rounded): (row, col) of a tap
                // = (px, px) * (sb, ca) + (py, py) * (ca, -sb) as two products and one sum, then the rounding constant
                const f32x2 kMagic = {12582912.f, 12582912.f};
                const f32x2 a0 = f32x2{px0[t], px0[t]} * sc, b0 = f32x2{py0[t], py0[t]} * cs;
                const f32x2 a1 = f32x2{px1[t], px1[t]} * sc, b1 = f32x2{py1[t], py1[t]} * cs;
                const f32x2 rc0 = (a0 + b0) + kMagic, rc1 = (a1 + b1) + kMagic;
                // (elements copied to scalars first: __builtin_bit_cast applied to a vector element reads element 0)
                const float r0f = rc0.x, c0f = rc0.y, r1f = rc1.x, c1f = rc1.y;
                const int r0 = __builtin_bit_cast(int, r0f), c0 = __builtin_bit_cast(int, c0f);
                const int r1 = __builtin_bit_cast(int, r1f), c1 = __builtin_bit_cast(int, c1f);
                const int t0 = win[mad_i16(r0, 4 * kDescWinDw, c0 + originM)];
                const int t1 = win[mad_i16(r1, 4 * kDescWinDw, c1 + originM)];
                nib |= (uint32_t)(t0 < t1) << t;
            }
            // lane i holds bits 4i..4i+3: assemble dwords over groups of 8 lanes
            uint32_t v = nib << (4 * (lane & 7));
            v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);
            v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);
            v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false);
            uint8_t *dout = a.desc + ((size_t)f * a.cap + iBase + k) * 32;
            if ((lane & 7) == 0) *reinterpret_cast<uint32_t *>(dout + (lane >> 3) * 4) = v;
            __builtin_amdgcn_wave_barrier();   // the next keypoint's window overwrites sWin
            dw[0] = nx[0];
            dw[1] = nx[1];
        }
    }
}

}  // namespace orbfe
