// k_match.h -- K7/K8/K9: Hamming matching kernels (gfx950): XOR + v_bcnt_u32_b32, best / second-best in key space.
//
// Replaces the distance loops around ORBmatcher::DescriptorDistance (reference src/ORBmatcher.cc:1647-1663):
//   knn2   -- brute-force best/second of SearchByBoW's inner loops (:205-226, :570-596) and the 1M-descriptor config;
//   csr    -- candidate-list argmin of SearchByProjection / Fuse / SearchBySim3 (:83-115, :372-392, :1400-1424, ...);
//   stereo -- row-band search of Frame::ComputeStereoMatches (src/Frame.cc:476-549).
//
// Selection rule everywhere in the reference:  if d<best {second=best; best=d; idx=j} else if d<second {second=d},
// scanning j ascending.  With key = (d << 22) | j (d <= 256, j < 2^22) that rule is exactly
//     second = med3(best, key, second);  best = min(best, key)          (two VALU ops, unsigned)
// because keys are totally ordered by (distance, index): the smallest key is the first minimum, and the distance of the
// second-smallest key is the second-smallest distance of the multiset.
//
// Layout: one query per lane, its 8 dwords in VGPRs for the whole scan; reference descriptors are addressed uniformly
// across the wave, so they arrive through the scalar data path (s_load_dwordx8) and feed v_xor / v_bcnt as SGPR
// operands -- no LDS traffic, 32 bytes of reference per 64 (or more) distances.  16 integer VALU lane-ops per pair
// (8 XOR + 8 popcount-accumulate) + 3 for selection: VALU-bound, not HBM-bound and not a contraction (no MFMA).
#pragma once
#include "common.h"

namespace orbfe {

constexpr uint32_t kKeyShift = 22;
constexpr uint32_t kKeyIdxMask = (1u << kKeyShift) - 1;
constexpr uint32_t kKeyNone = (256u << kKeyShift) | kKeyIdxMask;   // "no candidate": distance 256, index -1
constexpr int kKnnChunkMax = 1 << kKeyShift;

__device__ __forceinline__ uint32_t med3u(uint32_t a, uint32_t b, uint32_t c) {
    return max(min(a, b), min(max(a, b), c));   // folds to v_med3_u32
}

// v_bcnt_u32_b32 D = popcount(S0) + S1: the accumulate operand makes the 8-dword distance 8 XOR + 8 BCNT with no separate
// adds.  hipcc re-associates `d += __popc(x)` into bcnt(x, 0) + v_add3 trees, so the chain is spelled out.
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) {
    uint32_t d;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(acc));
    return d;
}

__device__ __forceinline__ uint32_t hamming256(const uint32_t (&q)[8], const uint32_t *__restrict__ r) {
    // two 16-byte loads: with a wave-uniform address these become s_load_dwordx4/x8 and the XORs take SGPR operands
    const uint4 a = reinterpret_cast<const uint4 *>(r)[0], b = reinterpret_cast<const uint4 *>(r)[1];
    uint32_t d = bcnt_acc(q[0] ^ a.x, 0);
    d = bcnt_acc(q[1] ^ a.y, d);
    d = bcnt_acc(q[2] ^ a.z, d);
    d = bcnt_acc(q[3] ^ a.w, d);
    d = bcnt_acc(q[4] ^ b.x, d);
    d = bcnt_acc(q[5] ^ b.y, d);
    d = bcnt_acc(q[6] ^ b.z, d);
    d = bcnt_acc(q[7] ^ b.w, d);
    return d;
}

// partial[chunk][nqPad] <- (best key, second key) of queries vs references [chunk*chunkLen, (chunk+1)*chunkLen)
//
// FILTER: the caller knows, per query, a distance T (seed[qi]) that at least two references of the WHOLE scan do not
// exceed (the second-best distance over a sample of the database, knn2 pre-pass).  A reference with d > T can then be
// neither the best nor the second-best of the whole scan, and inside a chunk a reference changes (best, second) only if
// d < the chunk's running second distance (equal distance = later index = larger key).  So the selection -- three
// half-rate VALU ops per pair -- is needed only where d < thr = min(T + 1, running second distance), which for a large
// database is a few pairs in 10^4: the four distances of a trip are reduced with v_min3 / v_min, compared once, and the
// wave branches around the selection unless some lane needs it.  Chunks may then report "none" (256) for a place that a
// reference above T would have taken; the ordered fold of the chunks is unaffected because the final two places are held
// by references <= T.
template <bool FILTER>
__global__ __launch_bounds__(256) void knn2_scan_kernel(const uint32_t *__restrict__ q, int nq, const uint32_t *__restrict__ r,
                                                        int nr, int chunkLen, uint2 *__restrict__ partial, int nqPad,
                                                        const int *__restrict__ seed) {
    const int qi = blockIdx.x * 256 + threadIdx.x;
    uint32_t qv[8];
    {
        const uint4 *qp = reinterpret_cast<const uint4 *>(q + (size_t)min(qi, nq - 1) * 8);
        const uint4 a = qp[0], b = qp[1];
        qv[0] = a.x; qv[1] = a.y; qv[2] = a.z; qv[3] = a.w;
        qv[4] = b.x; qv[5] = b.y; qv[6] = b.z; qv[7] = b.w;
    }
    const int j0 = blockIdx.y * chunkLen;
    const int n = min(chunkLen, nr - j0);
    const uint32_t *__restrict__ rp = r + (size_t)j0 * 8;
    uint32_t best = kKeyNone, second = kKeyNone;
    uint32_t thr = 257;
    if (FILTER) thr = (uint32_t)min(max(seed[min(qi, nq - 1)], 0), 256) + 1u;
    // four references per trip (32 SGPRs of s_load in flight); the inline-asm popcount chain keeps hipcc from unrolling a
    // runtime-trip-count loop by itself
    const int n4 = n & ~3;
    for (int j = 0; j < n4; j += 4) {
        const uint32_t *__restrict__ p4 = rp + (size_t)j * 8;
        const uint32_t d0 = hamming256(qv, p4), d1 = hamming256(qv, p4 + 8), d2 = hamming256(qv, p4 + 16), d3 = hamming256(qv, p4 + 24);
        if (FILTER) {
            const uint32_t m = min(min(min(d0, d1), d2), d3);   // v_min3_u32 + v_min_u32
            if (__ballot(m < thr) == 0) continue;                 // wave-uniform
        }
        const uint32_t k0 = (d0 << kKeyShift) | (uint32_t)j, k1 = (d1 << kKeyShift) | (uint32_t)(j + 1),
                       k2 = (d2 << kKeyShift) | (uint32_t)(j + 2), k3 = (d3 << kKeyShift) | (uint32_t)(j + 3);
        if (FILTER) {
            // lanes whose distances are all >= thr keep their state: a key at distance T + 1 or more must not enter (it would
            // be harmless in this chunk but is not what the unfiltered scan of a later, emptier lane would report) -- keep it
            // simple and exact: apply the rule only to distances below thr
            const uint32_t e0 = d0 < thr ? k0 : kKeyNone, e1 = d1 < thr ? k1 : kKeyNone, e2 = d2 < thr ? k2 : kKeyNone,
                           e3 = d3 < thr ? k3 : kKeyNone;
            second = med3u(best, e0, second); best = min(best, e0);
            second = med3u(best, e1, second); best = min(best, e1);
            second = med3u(best, e2, second); best = min(best, e2);
            second = med3u(best, e3, second); best = min(best, e3);
            thr = min(thr, second >> kKeyShift);
        } else {
            second = med3u(best, k0, second); best = min(best, k0);
            second = med3u(best, k1, second); best = min(best, k1);
            second = med3u(best, k2, second); best = min(best, k2);
            second = med3u(best, k3, second); best = min(best, k3);
        }
    }
    for (int j = n4; j < n; j++) {
        const uint32_t d = hamming256(qv, rp + (size_t)j * 8);
        const uint32_t key = (!FILTER || d < thr) ? ((d << kKeyShift) | (uint32_t)j) : kKeyNone;
        second = med3u(best, key, second);
        best = min(best, key);
        if (FILTER) thr = min(thr, second >> kKeyShift);
    }
    if (qi < nq) partial[(size_t)blockIdx.y * nqPad + qi] = make_uint2(best, second);
}

// fold the chunks in ascending order (= ascending reference index) into the final (best_dist, best_idx, second_dist).
// 64 queries per workgroup; the chunk range is cut into 4 consecutive groups folded by 4 waves in parallel, then the four
// group results are folded in order (the update rule is associative over ordered segments).
__device__ __forceinline__ void knn2_fold(uint32_t &bd, int &bi, uint32_t &sd, uint32_t d1, int i1, uint32_t d2) {
    // a later segment contributes the ordered pair (d1 <= d2): apply the reference's update rule to both in scan order
    if (d1 < bd) {
        sd = min(bd, d2);
        bd = d1;
        bi = i1;
    } else {
        sd = min(sd, d1);
    }
}

// Stage 1 (grid = query blocks of 64 x chunk groups): the group's chunks are cut into 4 consecutive runs folded by the 4
// waves, the four runs are folded in order, and the group's (best_dist, best_idx, second_dist) goes to `groups`.  Stage 2
// folds the groups of a query in order.  With one stage (nGroups == 1) the result goes straight to the outputs.
struct KnnTriple { uint32_t bd; int bi; uint32_t sd; uint32_t pad; };

__global__ __launch_bounds__(256) void knn2_merge_kernel(const uint2 *__restrict__ partial, int nChunks, int nqPad, int nq,
                                                         int chunkLen, int idxBase, int chunksPerGroup,
                                                         KnnTriple *__restrict__ groups, int *__restrict__ bestDist,
                                                         int *__restrict__ bestIdx, int *__restrict__ secondDist) {
    __shared__ uint32_t sBd[4][64], sSd[4][64];
    __shared__ int sBi[4][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int qi = blockIdx.x * 64 + lane;
    const int g0 = blockIdx.y * chunksPerGroup, g1 = min(g0 + chunksPerGroup, nChunks);
    const int per = (g1 - g0 + 3) / 4;
    const int cBeg = g0 + grp * per, cEnd = min(cBeg + per, g1);
    uint32_t bd = 256, sd = 256;
    int bi = -1;
    if (qi < nq)
        for (int c = cBeg; c < cEnd; c++) {
            const uint2 p = partial[(size_t)c * nqPad + qi];
            knn2_fold(bd, bi, sd, p.x >> kKeyShift, idxBase + c * chunkLen + (int)(p.x & kKeyIdxMask), p.y >> kKeyShift);
        }
    sBd[grp][lane] = bd; sSd[grp][lane] = sd; sBi[grp][lane] = bi;
    __syncthreads();
    if (grp == 0 && qi < nq) {
#pragma unroll
        for (int gidx = 1; gidx < 4; gidx++) knn2_fold(bd, bi, sd, sBd[gidx][lane], sBi[gidx][lane], sSd[gidx][lane]);
        if (groups) {
            KnnTriple t;
            t.bd = bd; t.bi = bi; t.sd = sd; t.pad = 0;
            groups[(size_t)blockIdx.y * nqPad + qi] = t;
        } else {
            bestDist[qi] = (int)bd;
            bestIdx[qi] = bi;
            secondDist[qi] = (int)sd;
        }
    }
}

__global__ __launch_bounds__(64) void knn2_merge_groups_kernel(const KnnTriple *__restrict__ groups, int nGroups, int nqPad, int nq,
                                                               int *__restrict__ bestDist, int *__restrict__ bestIdx,
                                                               int *__restrict__ secondDist) {
    const int qi = blockIdx.x * 64 + threadIdx.x;
    if (qi >= nq) return;
    uint32_t bd = 256, sd = 256;
    int bi = -1;
    for (int g = 0; g < nGroups; g++) {
        const KnnTriple t = groups[(size_t)g * nqPad + qi];
        knn2_fold(bd, bi, sd, t.bd, t.bi, t.sd);
    }
    bestDist[qi] = (int)bd;
    bestIdx[qi] = bi;
    secondDist[qi] = (int)sd;
}

// candidate-list argmin: one query per lane, candidates scanned in list order
__global__ __launch_bounds__(256) void match_csr_kernel(const uint32_t *__restrict__ q, int nq, const uint32_t *__restrict__ r,
                                                        int nr, const int *__restrict__ off, const int *__restrict__ idx,
                                                        int *__restrict__ bestDist, int *__restrict__ bestIdx,
                                                        int *__restrict__ secondDist) {
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) return;
    uint32_t qv[8];
    {
        const uint4 *qp = reinterpret_cast<const uint4 *>(q + (size_t)qi * 8);
        const uint4 a = qp[0], b = qp[1];
        qv[0] = a.x; qv[1] = a.y; qv[2] = a.z; qv[3] = a.w;
        qv[4] = b.x; qv[5] = b.y; qv[6] = b.z; qv[7] = b.w;
    }
    const int c0 = off[qi], c1 = off[qi + 1];
    uint32_t bd = 256, sd = 256;
    int bi = -1;
    for (int c = c0; c < c1; c++) {
        const int j = idx[c];
        if (j < 0 || j >= nr) continue;   // malformed list entry: skipped rather than read out of bounds
        const uint4 *rp = reinterpret_cast<const uint4 *>(r + (size_t)j * 8);
        const uint4 a = rp[0], b = rp[1];
        const uint32_t d = __popc(qv[0] ^ a.x) + __popc(qv[1] ^ a.y) + __popc(qv[2] ^ a.z) + __popc(qv[3] ^ a.w) +
                           __popc(qv[4] ^ b.x) + __popc(qv[5] ^ b.y) + __popc(qv[6] ^ b.z) + __popc(qv[7] ^ b.w);
        if (d < bd) { sd = bd; bd = d; bi = j; }
        else if (d < sd) sd = d;
    }
    bestDist[qi] = (int)bd;
    bestIdx[qi] = bi;
    secondDist[qi] = (int)sd;
}

// ---- stereo row band -----------------------------------------------------------------------------------------------
struct StereoRight {   // per right keypoint: row band, octave, u  (src/Frame.cc:483-493)
    int minr, maxr, octave;
    float u;
};

__global__ __launch_bounds__(256) void stereo_prep_kernel(const KeyPointOut *__restrict__ kr, int nr, StereoRight *__restrict__ out,
                                                          float s0, float s1, float s2, float s3, float s4, float s5, float s6,
                                                          float s7, float s8, float s9, float s10, float s11, float s12,
                                                          float s13, float s14, float s15) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nr) return;
    const float sf[16] = {s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15};
    const KeyPointOut k = kr[i];
    float scale = sf[0];
#pragma unroll
    for (int l = 1; l < 16; l++) scale = (k.octave == l) ? sf[l] : scale;
    const float rr = __fmul_rn(2.0f, scale);
    StereoRight o;
    o.maxr = (int)ceilf(__fadd_rn(k.y, rr));
    o.minr = (int)floorf(__fsub_rn(k.y, rr));
    o.octave = k.octave;
    o.u = k.x;
    out[i] = o;
}

// best[iL] must be pre-set to (100 << 22) (TH_HIGH, index 0); blocks covering different right chunks fold with atomicMin
__global__ __launch_bounds__(256) void stereo_rowband_kernel(const KeyPointOut *__restrict__ kl, const uint32_t *__restrict__ dl,
                                                             int nl, const StereoRight *__restrict__ sr,
                                                             const uint32_t *__restrict__ dr, int nr, int chunkLen,
                                                             float minD, float maxD, int nrows, uint32_t *__restrict__ best) {
    const int iL = blockIdx.x * 256 + threadIdx.x;
    const bool live = iL < nl;
    const KeyPointOut k = kl[min(iL, nl - 1)];
    uint32_t qv[8];
    {
        const uint4 *qp = reinterpret_cast<const uint4 *>(dl + (size_t)min(iL, nl - 1) * 8);
        const uint4 a = qp[0], b = qp[1];
        qv[0] = a.x; qv[1] = a.y; qv[2] = a.z; qv[3] = a.w;
        qv[4] = b.x; qv[5] = b.y; qv[6] = b.z; qv[7] = b.w;
    }
    const int row = (int)k.y;                 // vRowIndices[vL]: float -> index truncation (src/Frame.cc:511)
    const float minU = __fsub_rn(k.x, maxD), maxU = __fsub_rn(k.x, minD);
    const bool ok = live && row >= 0 && row < nrows && !(maxU < 0);
    const int j0 = blockIdx.y * chunkLen, j1 = min(j0 + chunkLen, nr);
    uint32_t bk = 100u << kKeyShift;
    for (int j = j0; j < j1; j++) {
        const StereoRight s = sr[j];          // uniform: scalar loads
        const bool cand = ok && row >= s.minr && row <= s.maxr && s.octave >= k.octave - 1 && s.octave <= k.octave + 1 &&
                          s.u >= minU && s.u <= maxU;
        if (cand) {
            const uint32_t d = hamming256(qv, dr + (size_t)j * 8);
            bk = min(bk, (d << kKeyShift) | (uint32_t)j);
        }
    }
    if (live && bk < (100u << kKeyShift)) atomicMin(&best[iL], bk);
}

__global__ __launch_bounds__(256) void stereo_finish_kernel(const uint32_t *__restrict__ best, int nl, int *__restrict__ bestDist,
                                                            int *__restrict__ bestIdx) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nl) return;
    const uint32_t b = best[i];
    bestDist[i] = (int)(b >> kKeyShift);
    bestIdx[i] = (int)(b & kKeyIdxMask);
}

__global__ __launch_bounds__(256) void fill_u32_kernel(uint32_t *p, int n, uint32_t v) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---- stereo: SAD sub-pixel refinement and median gate (N3) ----------------------------------------------------------
// Replaces the second half of Frame::ComputeStereoMatches (reference src/Frame.cc:551-639).  One wavefront per left
// keypoint: lanes hold the 121 pixels of the 11x11 patch (two per lane), the eleven centre-subtracted L1 distances are
// integer wave sums (the reference accumulates the same small integers in floating point, exactly), the parabola and
// the disparity are float32 with individually rounded operations.  sad[iL] = best SAD of an accepted match, -1 otherwise.
struct StereoRefineArgs {
    const KeyPointOut *kl; int nl;
    const KeyPointOut *kr; int nr;
    const uint32_t *best;            // (distance << 22 | right index) from the row-band search
    Geom g;
    ImgSrc left, right;
    int leftFrame, rightFrame;
    float invScale[kMaxLevels], scale[kMaxLevels];
    float mbf, minD, maxD;
    float *uRight, *depth;
    int *sad;
};

__device__ __forceinline__ int stereo_wave_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);   // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);   // row_mirror
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}

__global__ __launch_bounds__(256) void stereo_refine_kernel(StereoRefineArgs a) {
    const int lane = threadIdx.x & 63;
    const int iL = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (iL >= a.nl) return;
    if (lane == 0) { a.uRight[iL] = -1.0f; a.depth[iL] = -1.0f; a.sad[iL] = -1; }
    const uint32_t bk = a.best[iL];
    const int hd = (int)(bk >> kKeyShift), iR = (int)(bk & kKeyIdxMask);
    if (!(hd < 75) || iR >= a.nr) return;                       // thOrbDist = (TH_HIGH + TH_LOW) / 2  (:471, :552)
    const KeyPointOut kpL = a.kl[iL];
    const int oct = min(max(kpL.octave, 0), a.g.nlevels - 1);
    const float uL = kpL.x, uR0 = a.kr[iR].x;
    const float sfi = a.invScale[oct];
    const float scaleduL = roundf(__fmul_rn(kpL.x, sfi)), scaledvL = roundf(__fmul_rn(kpL.y, sfi)),
                scaleduR0 = roundf(__fmul_rn(uR0, sfi));         // :557-559
    const int w = 5, L = 5;
    const LevelGeom &G = a.g.lv[oct];
    const float iniu = scaleduR0 + (float)(L - w), endu = scaleduR0 + (float)(L + w + 1);
    if (iniu < 0 || endu >= (float)G.w) return;                  // :573-576
    const int cy = (int)scaledvL, cxl = (int)scaleduL, cxr0 = (int)scaleduR0;
    // the reference would throw on a window outside the level; keypoints of this library never get there
    if (cy - w < 0 || cy + w >= G.h || cxl - w < 0 || cxl + w >= G.w || cxr0 - L - w < 0) return;
    int pl, pr;
    const uint8_t *imgL = level_ptr(a.left, a.g, a.leftFrame, oct, &pl);
    const uint8_t *imgR = level_ptr(a.right, a.g, a.rightFrame, oct, &pr);
    // patch positions p = lane and lane + 64 (< 121): (dy, dx) = (p / 11 - 5, p % 11 - 5)
    int dy[2], dx[2], aL[2];
    bool in[2];
    const int cL = imgL[(size_t)cy * pl + cxl];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int p = lane + 64 * h;
        in[h] = p < 121;
        const int pp = in[h] ? p : 0;
        dy[h] = pp / 11 - w;
        dx[h] = pp - (pp / 11) * 11 - w;
        aL[h] = (int)imgL[(size_t)(cy + dy[h]) * pl + cxl + dx[h]] - cL;
    }
    int bestDist = 0x7fffffff, bestincR = 0;
    float vDists[2 * 5 + 1];
#pragma unroll
    for (int incR = -L; incR <= L; incR++) {
        const int cxr = cxr0 + incR;
        const int cR = imgR[(size_t)cy * pr + cxr];
        int sum = 0;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int b = (int)imgR[(size_t)(cy + dy[h]) * pr + cxr + dx[h]] - cR;
            const int d = aL[h] - b;
            sum += in[h] ? (d < 0 ? -d : d) : 0;
        }
        const float dist = (float)stereo_wave_sum(sum);          // cv::norm(IL, IR, NORM_L1)  (:584)
        if (dist < (float)bestDist) { bestDist = (int)dist; bestincR = incR; }
        vDists[L + incR] = dist;
    }
    if (bestincR == -L || bestincR == L) return;                 // :594
    float dist1 = 0, dist2 = 0, dist3 = 0;
#pragma unroll
    for (int k = 1; k < 2 * 5; k++)
        if (k == L + bestincR) { dist1 = vDists[k - 1]; dist2 = vDists[k]; dist3 = vDists[k + 1]; }
    const float deltaR = __fdiv_rn(__fsub_rn(dist1, dist3),
                                   __fmul_rn(2.0f, __fsub_rn(__fadd_rn(dist1, dist3), __fmul_rn(2.0f, dist2))));   // :602
    if (deltaR < -1 || deltaR > 1) return;
    float bestuR = __fmul_rn(a.scale[oct], __fadd_rn(__fadd_rn(scaleduR0, (float)bestincR), deltaR));            // :608
    float disparity = __fsub_rn(uL, bestuR);
    if (disparity >= a.minD && disparity < a.maxD) {
        if (disparity <= 0) {
            disparity = (float)0.01;
            bestuR = (float)((double)uL - 0.01);                 // double expression in the reference (:616-617)
        }
        if (lane == 0) {
            a.depth[iL] = __fdiv_rn(a.mbf, disparity);
            a.uRight[iL] = bestuR;
            a.sad[iL] = bestDist;
        }
    }
}

// median gate (:626-639): the accepted matches whose SAD is >= 1.5 * 1.4 * median are dropped.  median = element n/2 of
// the ascending SADs (a SAD is < 2^16: 121 pixels x 510), found with two 256-bin histograms.  One workgroup.
__global__ __launch_bounds__(1024) void stereo_gate_kernel(const int *__restrict__ sad, int nl, float *__restrict__ uRight,
                                                           float *__restrict__ depth) {
    __shared__ int hist[256];
    __shared__ int sel[4];   // chosen high byte, rank inside it, chosen low byte, count
    const int tid = threadIdx.x;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < nl; i += 1024) {
        const int s = sad[i];
        if (s >= 0) atomicAdd(&hist[min(s >> 8, 255)], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int n = 0;
        for (int b = 0; b < 256; b++) n += hist[b];
        int k = n / 2, b = 0;
        while (b < 255 && k >= hist[b]) { k -= hist[b]; b++; }
        sel[0] = b; sel[1] = k; sel[3] = n;
    }
    __syncthreads();
    const int n = sel[3], hb = sel[0];
    if (n == 0) return;       // the reference indexes an empty vector here; nothing to gate
    __syncthreads();
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < nl; i += 1024) {
        const int s = sad[i];
        if (s >= 0 && min(s >> 8, 255) == hb) atomicAdd(&hist[s & 255], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int k = sel[1], b = 0;
        while (b < 255 && k >= hist[b]) { k -= hist[b]; b++; }
        sel[2] = b;
    }
    __syncthreads();
    const float median = (float)((hb << 8) | sel[2]);
    const float thDist = 1.5f * 1.4f * median;
    for (int i = tid; i < nl; i += 1024) {
        const int s = sad[i];
        if (s >= 0 && !((float)s < thDist)) { uRight[i] = -1.0f; depth[i] = -1.0f; }
    }
}

// ---- DBoW2 vocabulary-tree descent (N2) -----------------------------------------------------------------------------
// Replaces the per-feature loop of TemplatedVocabulary<FORB::TDescriptor, FORB>::transform (reference
// Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1218-1259, distance = FORB::distance, FORB.cpp:81-101).  One feature per
// lane: its 256 bits stay in 8 VGPRs while the lane walks from the root to a leaf; at every node the children's
// descriptors are gathered (two 16-byte loads each), v_xor + v_bcnt, strict '<' so the first minimum wins as in the
// reference.  Node ids grow from parent to child (checked on the host), so every walk ends.
struct BowArgs {
    const int32_t *childOff;   // [nNodes + 1]
    const int32_t *childIdx;
    const uint4 *nodeDesc;     // [nNodes][2]
    int nNodes, L, levelsup;
    const uint4 *desc;         // [n][2]
    int n;
    int32_t *leaf, *nid;
};

__global__ __launch_bounds__(256) void bow_descend_kernel(BowArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const uint4 q0 = a.desc[2 * (size_t)i], q1 = a.desc[2 * (size_t)i + 1];
    const int nidLevel = a.L - a.levelsup;
    int nid = 0, node = 0, level = 0;
    for (int guard = 0; guard < 64; guard++) {
        const int c0 = a.childOff[node], c1 = a.childOff[node + 1];
        if (c0 == c1) break;   // leaf
        ++level;
        int best = 0, bestD = 0x7fffffff;
        for (int c = c0; c < c1; c++) {
            const int id = a.childIdx[c];
            const uint4 d0 = a.nodeDesc[2 * (size_t)id], d1 = a.nodeDesc[2 * (size_t)id + 1];
            const int d = __popc(q0.x ^ d0.x) + __popc(q0.y ^ d0.y) + __popc(q0.z ^ d0.z) + __popc(q0.w ^ d0.w) +
                          __popc(q1.x ^ d1.x) + __popc(q1.y ^ d1.y) + __popc(q1.z ^ d1.z) + __popc(q1.w ^ d1.w);
            if (d < bestD) { bestD = d; best = id; }
        }
        node = best;
        if (level == nidLevel) nid = node;
    }
    a.leaf[i] = node;
    a.nid[i] = nid;
}

// ---- MapPoint::ComputeDistinctiveDescriptors, batched (N4) ---------------------------------------------------------
// Replaces src/MapPoint.cc:273-302 for many map points at once.  One wavefront per map point; lane i owns descriptor i
// (N <= 64: the common case, a point is seen from a few dozen keyframes), the N x N distance matrix goes to LDS as
// bytes pairs, the element of rank (int)(0.5*(N-1)) of row i (self distance 0 included) is found by bisection on the
// value range 0..256 (9 counting passes over the row), then a wave-wide arg-min with the lowest index winning.  Points
// with more than 64 descriptors walk the rows in blocks of 64 lanes and recompute distances instead of storing them.
__global__ __launch_bounds__(64) void distinctive_kernel(const uint4 *__restrict__ desc, const int32_t *__restrict__ off, int nPoints,
                                                         int32_t *__restrict__ best) {
    __shared__ uint16_t sD[64][64 + 2];
    const int p = blockIdx.x, lane = threadIdx.x;
    if (p >= nPoints) return;
    const int o0 = off[p], N = off[p + 1] - o0;
    if (N <= 0) { if (lane == 0) best[p] = -1; return; }
    const uint4 *d = desc + 2 * (size_t)o0;
    const int r = (int)(0.5 * (N - 1));
    unsigned bestKey = 0xFFFFFFFFu;   // (median << 16 | index): the minimum is the least median, lowest index first
    for (int i0 = 0; i0 < N; i0 += 64) {
        const int i = i0 + lane;
        const bool live = i < N;
        const uint4 a0 = d[2 * (size_t)min(i, N - 1)], a1 = d[2 * (size_t)min(i, N - 1) + 1];
        auto dist_to = [&](int j) {
            const uint4 b0 = d[2 * (size_t)j], b1 = d[2 * (size_t)j + 1];   // wave-uniform j: scalar loads
            return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
                   __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
        };
        const bool small = N <= 64;
        if (small)
            for (int j = 0; j < N; j++) sD[lane][j] = (uint16_t)dist_to(j);   // row i = lane; d(i,i) = 0 by construction
        // bisection: the smallest v with #{j : d_ij <= v} >= r + 1
        int lo = 0, hi = 256;
        while (lo < hi) {   // wave-uniform bounds are not needed: every lane runs 9 steps at most
            const int mid = (lo + hi) >> 1;
            int cnt = 0;
            if (small) for (int j = 0; j < N; j++) cnt += sD[lane][j] <= mid ? 1 : 0;
            else for (int j = 0; j < N; j++) cnt += dist_to(j) <= mid ? 1 : 0;
            if (cnt >= r + 1) hi = mid; else lo = mid + 1;
        }
        const unsigned key = live ? (((unsigned)lo << 16) | (unsigned)i) : 0xFFFFFFFFu;
        bestKey = min(bestKey, key);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bestKey = min(bestKey, (unsigned)__shfl_xor((int)bestKey, o));
    if (lane == 0) best[p] = (int)(bestKey & 0xFFFFu);
}

// ---- windowed search on the frame grid (rest of N4) -----------------------------------------------------------------
// Replaces Frame::GetFeaturesInArea (reference src/Frame.cc:327-380) plus the candidate loop of
// ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th) (src/ORBmatcher.cc:71-115) for a batch of
// projected map points.  The grid (64 x 48 cells, cell lists in keypoint order as Frame::AssignFeaturesToGrid builds them)
// lives on the device as a CSR; one query per lane walks its cells in the reference's order (ix outer, iy inner, list
// order inside a cell) and keeps best / second with the reference's strict '<' updates, so ties resolve identically.
struct AreaQuery {       // == orbfe_area_query
    float x, y, r;       // mTrackProjX, mTrackProjY, radius
    int minLevel, maxLevel;
    float projXR, erMax; // right-coordinate gate (:87-92), off when erMax < 0
    int pad;
};
struct AreaResult {      // == orbfe_area_result
    int bestDist, bestIdx, bestLevel, bestDist2, bestLevel2, nCand, bestIdx2, pad1;
};
struct AreaArgs {
    const int32_t *cellOff;      // [64*48 + 1], cell id = ix * 48 + iy
    const int32_t *cellIdx;
    const float *kx, *ky;
    const int32_t *koct;
    const uint4 *kdesc;          // [n][2]
    const float *uRight;         // may be null
    const uint8_t *blocked;      // may be null
    const int32_t *distLimit;    // may be null: skip idx when distLimit[idx] <= dist (SearchForInitialization, src/ORBmatcher.cc:444)
    const float *invSigma2;      // may be null: mvInvLevelSigma2, switches on the reprojection gate of Fuse (:914-938)
    double chi2Mono, chi2Stereo; // 5.99 / 7.8
    float minX, minY, wInv, hInv;
    const AreaQuery *q;
    const uint4 *qdesc;          // [nq][2]
    int nq;
    AreaResult *res;
};

__global__ __launch_bounds__(64) void search_area_kernel(AreaArgs a) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= a.nq) return;
    const AreaQuery q = a.q[i];
    AreaResult out;
    out.bestDist = 256; out.bestIdx = -1; out.bestLevel = -1; out.bestDist2 = 256; out.bestLevel2 = -1; out.nCand = 0;
    out.bestIdx2 = -1; out.pad1 = 0;
    // :332-346, float operations in the reference's order
    const int nMinCellX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(q.x, a.minX), q.r), a.wInv)));
    const int nMaxCellX = min(64 - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(q.x, a.minX), q.r), a.wInv)));
    const int nMinCellY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(q.y, a.minY), q.r), a.hInv)));
    const int nMaxCellY = min(48 - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(q.y, a.minY), q.r), a.hInv)));
    if (!(nMinCellX >= 64 || nMaxCellX < 0 || nMinCellY >= 48 || nMaxCellY < 0)) {
        const uint4 q0 = a.qdesc[2 * (size_t)i], q1 = a.qdesc[2 * (size_t)i + 1];
        const bool bCheckLevels = q.minLevel > 0 || q.maxLevel >= 0;
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
            for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
                const int c0 = a.cellOff[ix * 48 + iy], c1 = a.cellOff[ix * 48 + iy + 1];
                for (int c = c0; c < c1; c++) {
                    const int idx = a.cellIdx[c];
                    const int oct = a.koct[idx];
                    if (bCheckLevels) {
                        if (oct < q.minLevel) continue;
                        if (q.maxLevel >= 0 && oct > q.maxLevel) continue;
                    }
                    const float distx = __fsub_rn(a.kx[idx], q.x), disty = __fsub_rn(a.ky[idx], q.y);
                    if (!(fabsf(distx) < q.r && fabsf(disty) < q.r)) continue;
                    out.nCand++;                                        // an element of vIndices
                    if (a.blocked && a.blocked[idx]) continue;          // src/ORBmatcher.cc:83-85
                    if (q.erMax >= 0 && a.uRight) {                     // :87-92
                        const float ur = a.uRight[idx];
                        if (ur > 0 && fabsf(__fsub_rn(q.projXR, ur)) > q.erMax) continue;
                    }
                    if (a.invSigma2) {                                  // Fuse, :914-938: u = q.x, v = q.y, ur = q.projXR
                        const float ex = __fsub_rn(q.x, a.kx[idx]), ey = __fsub_rn(q.y, a.ky[idx]);
                        float e2 = __fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey));
                        const float kpr = a.uRight ? a.uRight[idx] : -1.0f;
                        double lim = a.chi2Mono;
                        if (kpr >= 0) {
                            const float er = __fsub_rn(q.projXR, kpr);
                            e2 = __fadd_rn(e2, __fmul_rn(er, er));
                            lim = a.chi2Stereo;
                        }
                        if ((double)__fmul_rn(e2, a.invSigma2[oct]) > lim) continue;
                    }
                    const uint4 d0 = a.kdesc[2 * (size_t)idx], d1 = a.kdesc[2 * (size_t)idx + 1];
                    const int dist = __popc(q0.x ^ d0.x) + __popc(q0.y ^ d0.y) + __popc(q0.z ^ d0.z) + __popc(q0.w ^ d0.w) +
                                     __popc(q1.x ^ d1.x) + __popc(q1.y ^ d1.y) + __popc(q1.z ^ d1.z) + __popc(q1.w ^ d1.w);
                    if (a.distLimit && a.distLimit[idx] <= dist) continue;   // :444
                    if (dist < out.bestDist) {                          // :100-113
                        out.bestDist2 = out.bestDist; out.bestDist = dist;
                        out.bestLevel2 = out.bestLevel; out.bestLevel = oct;
                        out.bestIdx2 = out.bestIdx; out.bestIdx = idx;
                    } else if (dist < out.bestDist2) {
                        out.bestLevel2 = oct;
                        out.bestDist2 = dist;
                        out.bestIdx2 = idx;
                    }
                }
            }
    }
    a.res[i] = out;
}


// ---------------------------------------------------------------------------------------------------------------------
// SearchByBoW (reference src/ORBmatcher.cc:159-288 for KeyFrame / Frame, :522-655 for KeyFrame / KeyFrame): inside a
// vocabulary node shared by both feature vectors, every side-1 feature with a good MapPoint looks for its best and
// second-best side-2 feature of that node, skipping side-2 features already matched earlier in the same call (:209,
// :576).  A feature sits in exactly one node, so that dependency never leaves the node: one wavefront per shared node
// walks the node's side-1 features in list order, the lanes split the side-2 features, and the matched set lives in an
// LDS bitmask -- the sequential loop's result, association for association, without any replay.
//   key = dist << 16 | position in the node's side-2 list: the smallest key is the first minimum of the list scan (strict
//   '<'), the second smallest key carries the second-smallest distance of the multiset (the reference's bestDist2).
struct BowMatchArgs {
    const uint4 *desc1;          // [n1][2]
    const uint8_t *valid1;       // [n1] or null: MapPoint present and not bad (:191-197, :558-562)
    const uint4 *desc2;          // [n2][2]
    const uint8_t *blocked2;     // [n2] or null: side-2 features that never take part (:574-580)
    const int32_t *off1, *idx1;  // per shared node: side-1 feature indices (FeatureVector order)
    const int32_t *off2, *idx2;  // per shared node: side-2 feature indices
    int nNodes;
    int maxDist;                 // accept bestDist1 <= maxDist (TH_LOW for :228, TH_LOW - 1 for the strict test at :598)
    float nnRatio;
    int32_t *match1;             // [n1], preset to -1: the side-2 feature a side-1 feature was matched to
};

constexpr int kBowMaxNodeFeatures = 65536;   // side-2 features of one node (LDS bitmask; 16-bit list positions)

__global__ __launch_bounds__(64) void bow_match_kernel(BowMatchArgs a) {
    __shared__ uint32_t sTaken[kBowMaxNodeFeatures / 32];
    const int node = blockIdx.x, lane = threadIdx.x;
    const int b1 = a.off1[node], e1 = a.off1[node + 1], b2 = a.off2[node], e2 = a.off2[node + 1];
    const int n2 = e2 - b2;
    if (n2 <= 0 || e1 <= b1) return;
    for (int wI = lane; wI * 32 < n2; wI += 64) {          // side-2 features excluded from the start
        uint32_t bits = 0;
        if (a.blocked2)
            for (int k = 0; k < 32 && wI * 32 + k < n2; k++) bits |= (a.blocked2[a.idx2[b2 + wI * 32 + k]] ? 1u : 0u) << k;
        sTaken[wI] = bits;
    }
    __syncthreads();
    // the first 64 side-2 descriptors stay in registers (most nodes hold fewer)
    uint4 c0 = make_uint4(0, 0, 0, 0), c1 = c0;
    if (lane < n2) {
        const int j = a.idx2[b2 + lane];
        c0 = a.desc2[2 * (size_t)j]; c1 = a.desc2[2 * (size_t)j + 1];
    }
    for (int i = b1; i < e1; i++) {                        // the reference's loop over vIndicesKF, in order
        const int q = a.idx1[i];
        if (a.valid1 && !a.valid1[q]) continue;
        const uint4 q0 = a.desc1[2 * (size_t)q], q1 = a.desc1[2 * (size_t)q + 1];
        uint32_t best = (256u << 16) | 0xFFFFu, second = best;
        for (int base = 0; base < n2; base += 64) {
            const int pos = base + lane;
            if (pos < n2 && !((sTaken[pos >> 5] >> (pos & 31)) & 1u)) {
                uint4 d0 = c0, d1 = c1;
                if (base) {
                    const int j = a.idx2[b2 + pos];
                    d0 = a.desc2[2 * (size_t)j]; d1 = a.desc2[2 * (size_t)j + 1];
                }
                const uint32_t dist = __popc(q0.x ^ d0.x) + __popc(q0.y ^ d0.y) + __popc(q0.z ^ d0.z) + __popc(q0.w ^ d0.w) +
                                      __popc(q1.x ^ d1.x) + __popc(q1.y ^ d1.y) + __popc(q1.z ^ d1.z) + __popc(q1.w ^ d1.w);
                const uint32_t key = (dist << 16) | (uint32_t)pos;
                second = min(second, max(best, key));
                best = min(best, key);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {                 // two smallest keys of the wave
            const uint32_t ob = (uint32_t)__shfl_xor((int)best, o), os = (uint32_t)__shfl_xor((int)second, o);
            second = min(min(second, os), max(best, ob));
            best = min(best, ob);
        }
        const int bestDist1 = (int)(best >> 16), bestDist2 = (int)(second >> 16), bestPos = (int)(best & 0xFFFFu);
        if (bestDist1 <= a.maxDist && (float)bestDist1 < __fmul_rn(a.nnRatio, (float)bestDist2)) {   // :228-230 / :598-600
            if (lane == 0) {
                sTaken[bestPos >> 5] |= 1u << (bestPos & 31);
                a.match1[q] = a.idx2[b2 + bestPos];
            }
        }
        __syncthreads();                                   // one wave per workgroup: orders the LDS update before the next scan
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// SearchForTriangulation (reference src/ORBmatcher.cc:657-823): inside a shared vocabulary node every side-1 feature
// without a MapPoint looks for the side-2 feature without a MapPoint that has the smallest distance <= TH_LOW among those
// that pass the epipole test (:743-749) and CheckDistEpipolarLine (:140-157).  The reference accepts a candidate when
// `dist <= bestDist` (:738), so of several candidates at the least distance the LAST in list order wins; vbMatched2 is
// never set (:677, :725), so the loop has no dependency between side-1 features.  One wavefront per node, lanes split the
// side-2 features; key = dist << 16 | (0xFFFF - position): the smallest key is that candidate.
struct TriArgs {
    const uint4 *desc1; const uint8_t *hasMp1; const float *uRight1; const float2 *xy1;
    const uint4 *desc2; const uint8_t *hasMp2; const float *uRight2; const float2 *xy2; const int32_t *oct2;
    const int32_t *off1, *idx1, *off2, *idx2;
    float F12[9];                // row-major
    float ex, ey;                // epipole in the second image (:664-670)
    float scale[kMaxLevels], sigma2[kMaxLevels];
    int onlyStereo;
    int32_t *match1;             // [n1], preset to -1
};

__global__ __launch_bounds__(64) void triangulation_match_kernel(TriArgs a) {
    const int node = blockIdx.x, lane = threadIdx.x;
    const int b1 = a.off1[node], e1 = a.off1[node + 1], b2 = a.off2[node], e2 = a.off2[node + 1];
    const int n2 = e2 - b2;
    if (n2 <= 0 || e1 <= b1) return;
    for (int i = b1; i < e1; i++) {
        const int q = a.idx1[i];
        if (a.hasMp1[q]) continue;                                           // :699-703
        const bool bStereo1 = a.uRight1[q] >= 0;
        if (a.onlyStereo && !bStereo1) continue;                             // :707-709
        const uint4 q0 = a.desc1[2 * (size_t)q], q1 = a.desc1[2 * (size_t)q + 1];
        const float2 p1 = a.xy1[q];
        // l = x1' F12 (:143-145), every product and sum rounded on its own
        const float la = __fadd_rn(__fadd_rn(__fmul_rn(p1.x, a.F12[0]), __fmul_rn(p1.y, a.F12[3])), a.F12[6]);
        const float lb = __fadd_rn(__fadd_rn(__fmul_rn(p1.x, a.F12[1]), __fmul_rn(p1.y, a.F12[4])), a.F12[7]);
        const float lc = __fadd_rn(__fadd_rn(__fmul_rn(p1.x, a.F12[2]), __fmul_rn(p1.y, a.F12[5])), a.F12[8]);
        const float den = __fadd_rn(__fmul_rn(la, la), __fmul_rn(lb, lb));
        uint32_t best = 0xFFFFFFFFu;
        for (int base = 0; base < n2; base += 64) {
            const int pos = base + lane;
            if (pos >= n2) continue;
            const int j = a.idx2[b2 + pos];
            if (a.hasMp2[j]) continue;                                       // :725
            const bool bStereo2 = a.uRight2[j] >= 0;
            if (a.onlyStereo && !bStereo2) continue;                         // :730-732
            const uint4 d0 = a.desc2[2 * (size_t)j], d1 = a.desc2[2 * (size_t)j + 1];
            const uint32_t dist = __popc(q0.x ^ d0.x) + __popc(q0.y ^ d0.y) + __popc(q0.z ^ d0.z) + __popc(q0.w ^ d0.w) +
                                  __popc(q1.x ^ d1.x) + __popc(q1.y ^ d1.y) + __popc(q1.z ^ d1.z) + __popc(q1.w ^ d1.w);
            if (dist > 50u) continue;                                        // TH_LOW, :738
            const float2 p2 = a.xy2[j];
            const int o2 = a.oct2[j];
            if (!bStereo1 && !bStereo2) {                                    // :743-749
                const float dx = __fsub_rn(a.ex, p2.x), dy = __fsub_rn(a.ey, p2.y);
                if (__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)) < __fmul_rn(100.0f, a.scale[o2])) continue;
            }
            if (den == 0) continue;                                          // :151-152
            const float num = __fadd_rn(__fadd_rn(__fmul_rn(la, p2.x), __fmul_rn(lb, p2.y)), lc);
            const float dsqr = __fdiv_rn(__fmul_rn(num, num), den);
            if (!((double)dsqr < 3.84 * (double)a.sigma2[o2])) continue;     // :156
            best = min(best, (dist << 16) | (uint32_t)(0xFFFF - pos));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, o));
        if (lane == 0 && best != 0xFFFFFFFFu) a.match1[q] = a.idx2[b2 + (0xFFFF - (int)(best & 0xFFFFu))];
    }
}

}  // namespace orbfe
