/* sincos_census.c -- the rotation of computeOrbDescriptor over EVERY float keypoint angle.
 *
 * The reference rotates the 512 pattern taps by a = cosf(ang), b = sinf(ang), ang = angle * factorPI in float
 * (src/ORBextractor.cc:111-113; `using namespace std` at :67 makes them the float overloads, i.e. libm's cosf / sinf).
 * A descriptor is bit-exact only if the device produces the same two floats, or different ones that never move a tap.
 * This program settles that by enumeration of the 1 135 869 952 floats in [0, 360):
 *
 *   routine "taylor" -- the device routine of round 1 (double Taylor kernel on [-pi/4, pi/4], rounded to float once):
 *                       how many angles give a different cos or sin, and for how many of those does any of the 512 taps
 *                       (row = cvRound(px*b + py*a), col = cvRound(px*a - py*b), float products and sums rounded one
 *                       by one) land elsewhere;
 *   routine "glibc"  -- the device routine since round 2 (k_describe.h sincosf_glibc, the algorithm of glibc 2.35's
 *                       sinf / cosf restated): must differ for NO angle.
 *
 * Both host mirrors use the same IEEE double operations as the device code (explicit fma / mul; build with -mfma and
 * -ffp-contract=off so nothing else is fused).  Prints JSON: the counts, the angles where a tap moved under "taylor", and
 * a deterministic sample of the angles where "taylor" differed (bit pattern % 1024 == 0), each with libm's (cos, sin) bits
 * -- the known-answer set of tests/test_gpu_extract.py::test_rotation_known_answers.
 *
 * build: gcc -O2 -mfma -ffp-contract=off -fopenmp tools/sincos_census.c -o /tmp/sincos_census -lm
 * usage: sincos_census [stride]     (stride > 1: every stride-th float, for a quick run; the committed file is stride 1)
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static const signed char kPattern[1024] = {
#include "../oracle/pattern31.inc"
};

static uint32_t bits_of(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

/* round-1 device routine */
static void sincos_taylor(double x, float *sOut, float *cOut) {
    const double twoOverPi = 0.63661977236758134308, pio2Hi = 1.57079632679489655800e+00, pio2Lo = 6.12323399573676603587e-17;
    const double qd = __builtin_rint(x * twoOverPi);
    const int q = (int)qd;
    double r = __builtin_fma(-qd, pio2Hi, x);
    r = __builtin_fma(-qd, pio2Lo, r);
    const double z = r * r;
    double ps = -7.6471637318198164759e-13;
    ps = __builtin_fma(ps, z, 1.6059043836821614599e-10);
    ps = __builtin_fma(ps, z, -2.5052108385441718775e-08);
    ps = __builtin_fma(ps, z, 2.7557319223985890653e-06);
    ps = __builtin_fma(ps, z, -1.9841269841269841270e-04);
    ps = __builtin_fma(ps, z, 8.3333333333333333333e-03);
    ps = __builtin_fma(ps, z, -1.6666666666666666667e-01);
    const double sr = __builtin_fma(ps * z, r, r);
    double pc = 4.7794773323873852974e-14;
    pc = __builtin_fma(pc, z, -1.1470745597729724714e-11);
    pc = __builtin_fma(pc, z, 2.0876756987868098979e-09);
    pc = __builtin_fma(pc, z, -2.7557319223985890653e-07);
    pc = __builtin_fma(pc, z, 2.4801587301587301587e-05);
    pc = __builtin_fma(pc, z, -1.3888888888888888889e-03);
    pc = __builtin_fma(pc, z, 4.1666666666666666667e-02);
    pc = __builtin_fma(pc, z, -0.5);
    const double cr = __builtin_fma(pc, z, 1.0);
    const double s0 = (q & 1) ? cr : sr, c0 = (q & 1) ? sr : cr;
    *sOut = (float)((q & 2) ? -s0 : s0);
    *cOut = (float)(((q + 1) & 2) ? -c0 : c0);
}

/* host mirror of orbfe::sincosf_glibc (orb_slam2_amd/csrc/k_describe.h), operation for operation */
static void sincosf_glibc(float y, float *sOut, float *cOut) {
    const double hpiInv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5, c3 = -0x1.6c087e89a359dp-10,
                 c4 = 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    const double xd = (double)y;
    const int n = ((int)(xd * hpiInv) + 0x800000) >> 24;
    const double x = __builtin_fma(-(double)n, hpi, xd);
    const double x2 = x * x;
    const double x3 = x * x2;
    const double sa = __builtin_fma(x2, s3, s2);
    const double x7 = x3 * x2;
    const double sb = __builtin_fma(x3, s1, x);
    float S = (float)__builtin_fma(x7, sa, sb);
    const double x4 = x2 * x2;
    const double ca = __builtin_fma(x2, c4, c3);
    const double cb = __builtin_fma(x2, c2, c1);
    const double x6 = x4 * x2;
    const double cc = __builtin_fma(x2, cb, c0);
    float C = (float)__builtin_fma(x6, ca, cc);
    const int tiny = ((bits_of(y) >> 20) & 0x7ff) < ((bits_of(0x1p-12f) >> 20) & 0x7ff);
    S = tiny ? y : S;
    C = tiny ? 1.0f : C;
    const float sq = (n & 1) ? C : S, cq = (n & 1) ? S : C;
    *sOut = (n & 2) ? -sq : sq;
    *cOut = ((n + 1) & 2) ? -cq : cq;
}

static int cv_round(float v) { return (int)nearbyintf(v); }   /* ties to even under the default rounding mode */

static int taps_differ(float a1, float b1, float a2, float b2) {
    for (int i = 0; i < 512; i++) {
        const float px = (float)kPattern[2 * i], py = (float)kPattern[2 * i + 1];
        volatile float t;
        t = px * b1; float r1 = t; t = py * a1; r1 = r1 + t;
        t = px * a1; float c1 = t; t = py * b1; c1 = c1 - t;
        t = px * b2; float r2 = t; t = py * a2; r2 = r2 + t;
        t = px * a2; float c2 = t; t = py * b2; c2 = c2 - t;
        if (cv_round(r1) != cv_round(r2) || cv_round(c1) != cv_round(c2)) return 1;
    }
    return 0;
}

static int cmp_u32(const void *a, const void *b) {
    const uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : x > y;
}

#define MAX_TAP 8192
#define MAX_SAMPLE 8192

int main(int argc, char **argv) {
    const unsigned stride = argc > 1 ? (unsigned)atoi(argv[1]) : 1u;
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    const uint32_t n = bits_of(360.0f);   /* floats in [0, 360) are the bit patterns 0 .. bits(360) - 1 */
    long long tCos = 0, tSin = 0, tAny = 0, tTap = 0, gCos = 0, gSin = 0, gAny = 0, evaluated = 0;
    static uint32_t tapList[MAX_TAP], sample[MAX_SAMPLE];
    int nTap = 0, nSample = 0;
#pragma omp parallel for schedule(static) reduction(+ : tCos, tSin, tAny, tTap, gCos, gSin, gAny, evaluated)
    for (long long k = 0; k < (long long)n; k += stride) {
        union { float f; uint32_t u; } a;
        a.u = (uint32_t)k;
        volatile float angv = a.f * factorPI;
        const float ang = angv;
        const float c1 = cosf(ang), s1 = sinf(ang);
        float c2, s2, c3, s3;
        sincos_taylor((double)ang, &s2, &c2);
        sincosf_glibc(ang, &s3, &c3);
        evaluated++;
        if (bits_of(c1) != bits_of(c3) || bits_of(s1) != bits_of(s3)) {
            gAny++;
            gCos += bits_of(c1) != bits_of(c3);
            gSin += bits_of(s1) != bits_of(s3);
        }
        if (c1 != c2 || s1 != s2) {
            tAny++;
            tCos += c1 != c2;
            tSin += s1 != s2;
            const int t = taps_differ(c1, s1, c2, s2);
            tTap += t;
            if (t || (a.u & 1023u) == 0) {
#pragma omp critical
                {
                    if (t && nTap < MAX_TAP) tapList[nTap++] = a.u;
                    if ((a.u & 1023u) == 0 && nSample < MAX_SAMPLE) sample[nSample++] = a.u;
                }
            }
        }
    }
    qsort(tapList, nTap, sizeof(uint32_t), cmp_u32);
    qsort(sample, nSample, sizeof(uint32_t), cmp_u32);
    printf("{\"floats_in_0_360\": %u, \"stride\": %u, \"angles_evaluated\": %lld,\n", n, stride, evaluated);
    printf(" \"taylor\": {\"angles_where_cos_or_sin_differs\": %lld, \"cos_differs\": %lld, \"sin_differs\": %lld, "
           "\"angles_where_a_tap_moves\": %lld},\n", tAny, tCos, tSin, tTap);
    printf(" \"glibc\": {\"angles_where_cos_or_sin_differs\": %lld, \"cos_differs\": %lld, \"sin_differs\": %lld},\n", gAny, gCos, gSin);
    for (int pass = 0; pass < 2; pass++) {
        const uint32_t *list = pass ? sample : tapList;
        const int cnt = pass ? nSample : nTap;
        /* rows of [angle bits, libm cos bits, libm sin bits] */
        printf(" \"%s\": [", pass ? "sample_of_differing_angles" : "tap_moving_angles");
        for (int i = 0; i < cnt; i++) {
            union { float f; uint32_t u; } a;
            a.u = list[i];
            volatile float angv = a.f * factorPI;
            const float ang = angv;
            printf("%s[%u, %u, %u]", i ? ", " : "", a.u, bits_of(cosf(ang)), bits_of(sinf(ang)));
        }
        printf("]%s\n", pass ? "}" : ",");
    }
    return 0;
}
