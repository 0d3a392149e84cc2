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
    int nFrames, groupsPerFrame;
};

// umax[v] of the reference constructor (src/ORBextractor.cc:454-469) for HALF_PATCH_SIZE = 15; the host re-derives the
// table with the reference's arithmetic at create time and refuses to run if it differs.
__constant__ int c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

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

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// sin and cos of x in [0, 2*pi] in double, good to ~1e-15 absolute: quadrant reduction with a two-term pi/2, then Taylor
// polynomials on |r| <= pi/4 (remainders (pi/4)^17/17! and (pi/4)^16/16! are below 1e-16).  About 30 double operations
// instead of the ~170 of the general-purpose library routine; every operation is an explicit fma/mul/add.
__device__ __forceinline__ void sincos_0_2pi(double x, double *s, double *c) {
    const double twoOverPi = 0.63661977236758134308, pio2Hi = 1.57079632679489655800e+00, pio2Lo = 6.12323399573676603587e-17;
    const double qd = __builtin_rint(x * twoOverPi);
    const int q = (int)qd;
    double r = __builtin_fma(-qd, pio2Hi, x);
    r = __builtin_fma(-qd, pio2Lo, r);
    const double z = r * r;
    double ps = -7.6471637318198164759e-13;                // -1/15!
    ps = __builtin_fma(ps, z, 1.6059043836821614599e-10);  //  1/13!
    ps = __builtin_fma(ps, z, -2.5052108385441718775e-08); // -1/11!
    ps = __builtin_fma(ps, z, 2.7557319223985890653e-06);  //  1/9!
    ps = __builtin_fma(ps, z, -1.9841269841269841270e-04); // -1/7!
    ps = __builtin_fma(ps, z, 8.3333333333333333333e-03);  //  1/5!
    ps = __builtin_fma(ps, z, -1.6666666666666666667e-01); // -1/3!
    const double sr = __builtin_fma(ps * z, r, r);
    double pc = 4.7794773323873852974e-14;                 //  1/16!
    pc = __builtin_fma(pc, z, -1.1470745597729724714e-11); // -1/14!
    pc = __builtin_fma(pc, z, 2.0876756987868098979e-09);  //  1/12!
    pc = __builtin_fma(pc, z, -2.7557319223985890653e-07); // -1/10!
    pc = __builtin_fma(pc, z, 2.4801587301587301587e-05);  //  1/8!
    pc = __builtin_fma(pc, z, -1.3888888888888888889e-03); // -1/6!
    pc = __builtin_fma(pc, z, 4.1666666666666666667e-02);  //  1/4!
    pc = __builtin_fma(pc, z, -0.5);
    const double cr = __builtin_fma(pc, z, 1.0);
    // rotate by the quadrant
    const double s0 = (q & 1) ? cr : sr, c0 = (q & 1) ? sr : cr;
    *s = (q & 2) ? -s0 : s0;
    *c = ((q + 1) & 2) ? -c0 : c0;
}

__global__ __launch_bounds__(256) void orient_describe_kernel(DescribeArgs a) {
    const int lane = threadIdx.x & 63;
    // XCD-aware block mapping: workgroups are dealt round-robin over the 8 XCDs (private L2s), so block id % 8 selects
    // the XCD.  All blocks of frame f get ids with id % 8 == f % 8: a frame's patches (about 3 MB of level + blurred
    // bytes, gathered ~2.5 KB per keypoint) then stay in ONE 4-MiB L2 instead of being pulled into all eight.
    const int bid = blockIdx.x;
    const int f = 8 * ((bid >> 3) / a.groupsPerFrame) + (bid & 7);
    const int grp = (bid >> 3) % a.groupsPerFrame;
    if (f >= a.nFrames) return;
    const int i = grp * 4 + (threadIdx.x >> 6);   // keypoint index inside the frame, level-major
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

    // ---- IC_Angle on the un-blurred level: the 31 disc rows are fetched as 31 x 9 aligned dwords (5 per lane, all
    //      in flight together); each lane folds its own bytes into (m10, m01)
    int pitch;
    const uint8_t *img = level_ptr(a.s, a.g, f, lvl, &pitch);
    const int xa = (cx - kHalfPatch) & ~3;   // >= 4; xa + 35 <= cx + 20 < pitch
    uint32_t dwv[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int it = lane + 64 * k;
        const int row = it / 9, dc = it - row * 9;
        dwv[k] = (it < 31 * 9) ? *reinterpret_cast<const uint32_t *>(img + (size_t)(cy - kHalfPatch + row) * pitch + xa + 4 * dc) : 0u;
    }
    int m10 = 0, m01 = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int it = lane + 64 * k;
        const int row = it / 9, dc = it - row * 9;
        const int v = row - kHalfPatch;
        const int dmax = (it < 31 * 9) ? c_umax[v < 0 ? -v : v] : -1;
        int rs = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int u = xa + 4 * dc + b - cx;
            const int I = (int)((dwv[k] >> (8 * b)) & 0xFFu);
            const int Iin = (u >= -dmax && u <= dmax) ? I : 0;
            m10 += u * Iin;
            rs += Iin;
        }
        m01 += v * rs;
    }
    m10 = wave_sum(m10);
    m01 = wave_sum(m01);
    const float angle = fast_atan2_deg((float)m01, (float)m10);

    // ---- rBRIEF on the blurred level
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    const float ang = __fmul_rn(angle, factorPI);
    double sd, cd;
    sincos_0_2pi((double)ang, &sd, &cd);
    const float ca = (float)cd, sb = (float)sd;
    const uint8_t *blur = a.s.blur + (size_t)f * a.g.blurFrameBytes + L.blurOff + (size_t)cy * L.pitch + cx;
    const int bp = L.pitch;
    uint32_t nib = 0;
    const uint32_t *pat32 = reinterpret_cast<const uint32_t *>(c_pattern);
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const uint32_t pp = pat32[lane * 4 + t];   // (x0, y0, x1, y1) as int8
        const float x0 = (float)(int8_t)(pp & 0xFF), y0 = (float)(int8_t)((pp >> 8) & 0xFF), x1 = (float)(int8_t)((pp >> 16) & 0xFF),
                    y1 = (float)(int8_t)(pp >> 24);
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
