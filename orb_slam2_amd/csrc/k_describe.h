// k_describe.h -- K4 + K6: intensity-centroid orientation and 256-bit rBRIEF, one wavefront per keypoint (gfx950).
//
// Replaces IC_Angle / computeOrientation (reference src/ORBextractor.cc:77-104, 472-479, 851-852), computeOrbDescriptor /
// computeDescriptors (:108-147, 1034-1041) and the output assembly of operator() (:1072-1104: pt *= scale for level > 0,
// level-major concatenation).
//
// Orientation: lanes 0..30 each own one row v = lane-15 of the 749-pixel disc on the UN-blurred level and accumulate
// (sum u*I, v*sum I) in int32; a 64-lane butterfly reduces (m10, m01); the angle is cv::fastAtan2's polynomial in
// float32 with every operation individually rounded (the file is compiled with -ffp-contract=off).
// Descriptor: lane i evaluates test pairs 4i..4i+3 (8 rotated taps on the blurred level); a = cos, b = sin of the angle
// are taken in double and rounded to float (the reference calls libm cosf/sinf); tap offsets are
// cvRound(px*b + py*a), cvRound(px*a - py*b) with un-fused float multiplies/adds and round-half-even conversion.
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
};

// umax[v] of the reference constructor (src/ORBextractor.cc:454-469) for HALF_PATCH_SIZE = 15; the host re-derives the
// table with the reference's arithmetic at create time and refuses to run if it differs.
__constant__ int c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

__constant__ int8_t c_pattern[1024] = {
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

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(256) void orient_describe_kernel(DescribeArgs a) {
    const int lane = threadIdx.x & 63;
    const int f = blockIdx.y;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);   // keypoint index inside the frame, level-major
    // level prefix (<= 16 entries, wave-uniform)
    int lvl = -1, j = i, total = 0;
    for (int l = 0; l < a.g.nlevels; l++) {
        const int n = min(a.selCount[f * kMaxLevels + l], a.g.lv[l].selCap);
        if (lvl < 0 && j < n) lvl = l;
        if (lvl < 0) j -= n;
        total += n;
    }
    if (i == 0 && lane == 0) a.counts[f] = total;
    if (lvl < 0 || i >= a.cap) return;
    const LevelGeom &L = a.g.lv[lvl];
    const SelKp kp = a.sel[(size_t)f * a.g.selFrameCap + L.selOff + j];
    // keypoints lie in [19, w-20] x [19, h-20] by construction; the clamp only makes a corrupted entry harmless (no
    // out-of-bounds gather) instead of a GPU fault
    const int cx = min(max((int)kp.x, kEdge), L.w - kEdge - 1), cy = min(max((int)kp.y, kEdge), L.h - kEdge - 1);

    // ---- IC_Angle on the un-blurred level
    int pitch;
    const uint8_t *img = level_ptr(a.s, a.g, f, lvl, &pitch);
    int m10 = 0, m01 = 0;
    if (lane < 31) {
        const int v = lane - kHalfPatch;
        const int d = c_umax[v < 0 ? -v : v];
        const uint8_t *row = img + (size_t)(cy + v) * pitch + cx;
        int s = 0;
        for (int u = -d; u <= d; ++u) {
            const int I = row[u];
            m10 += u * I;
            s += I;
        }
        m01 = v * s;
    }
    m10 = wave_sum(m10);
    m01 = wave_sum(m01);
    const float angle = fast_atan2_deg((float)m01, (float)m10);

    // ---- rBRIEF on the blurred level
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    const float ang = __fmul_rn(angle, factorPI);
    const float ca = (float)cos((double)ang), sb = (float)sin((double)ang);
    const uint8_t *blur = a.s.blur + (size_t)f * a.g.blurFrameBytes + L.blurOff + (size_t)cy * L.pitch + cx;
    const int bp = L.pitch;
    uint32_t nib = 0;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int8_t *pp = &c_pattern[(lane * 4 + t) * 4];
        const float x0 = (float)pp[0], y0 = (float)pp[1], x1 = (float)pp[2], y1 = (float)pp[3];
        const int r0 = __float2int_rn(__fadd_rn(__fmul_rn(x0, sb), __fmul_rn(y0, ca)));
        const int c0 = __float2int_rn(__fsub_rn(__fmul_rn(x0, ca), __fmul_rn(y0, sb)));
        const int r1 = __float2int_rn(__fadd_rn(__fmul_rn(x1, sb), __fmul_rn(y1, ca)));
        const int c1 = __float2int_rn(__fsub_rn(__fmul_rn(x1, ca), __fmul_rn(y1, sb)));
        const int t0 = blur[r0 * bp + c0], t1 = blur[r1 * bp + c1];
        nib |= (uint32_t)(t0 < t1) << t;
    }
    // lane i holds bits 4i..4i+3: assemble bytes (2 lanes), then dwords (8 lanes)
    uint32_t v = nib << (4 * (lane & 7));
    v |= __shfl_xor(v, 1);
    v |= __shfl_xor(v, 2);
    v |= __shfl_xor(v, 4);
    uint8_t *dout = a.desc + ((size_t)f * a.cap + i) * 32;
    if ((lane & 7) == 0) *reinterpret_cast<uint32_t *>(dout + (lane >> 3) * 4) = v;
    if (lane == 0) {
        KeyPointOut o;
        o.x = (float)cx;
        o.y = (float)cy;
        if (lvl != 0) { o.x = __fmul_rn(o.x, L.scale); o.y = __fmul_rn(o.y, L.scale); }
        o.size = L.patch;
        o.angle = angle;
        o.response = (float)kp.resp;
        o.octave = lvl;
        o.class_id = -1;
        a.kps[(size_t)f * a.cap + i] = o;
    }
}

}  // namespace orbfe
