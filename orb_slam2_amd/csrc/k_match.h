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
__global__ __launch_bounds__(256) void knn2_scan_kernel(const uint32_t *__restrict__ q, int nq, const uint32_t *__restrict__ r,
                                                        int nr, int chunkLen, uint2 *__restrict__ partial, int nqPad) {
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
    // four references per trip (32 SGPRs of s_load in flight); the inline-asm popcount chain keeps hipcc from unrolling a
    // runtime-trip-count loop by itself
    const int n4 = n & ~3;
    for (int j = 0; j < n4; j += 4) {
        const uint32_t *__restrict__ p4 = rp + (size_t)j * 8;
        const uint32_t d0 = hamming256(qv, p4), d1 = hamming256(qv, p4 + 8), d2 = hamming256(qv, p4 + 16), d3 = hamming256(qv, p4 + 24);
        const uint32_t k0 = (d0 << kKeyShift) | (uint32_t)j, k1 = (d1 << kKeyShift) | (uint32_t)(j + 1),
                       k2 = (d2 << kKeyShift) | (uint32_t)(j + 2), k3 = (d3 << kKeyShift) | (uint32_t)(j + 3);
        second = med3u(best, k0, second); best = min(best, k0);
        second = med3u(best, k1, second); best = min(best, k1);
        second = med3u(best, k2, second); best = min(best, k2);
        second = med3u(best, k3, second); best = min(best, k3);
    }
    for (int j = n4; j < n; j++) {
        const uint32_t d = hamming256(qv, rp + (size_t)j * 8);
        const uint32_t key = (d << kKeyShift) | (uint32_t)j;
        second = med3u(best, key, second);
        best = min(best, key);
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

__global__ __launch_bounds__(256) void knn2_merge_kernel(const uint2 *__restrict__ partial, int nChunks, int nqPad, int nq,
                                                         int chunkLen, int idxBase, int *__restrict__ bestDist,
                                                         int *__restrict__ bestIdx, int *__restrict__ secondDist) {
    __shared__ uint32_t sBd[4][64], sSd[4][64];
    __shared__ int sBi[4][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int qi = blockIdx.x * 64 + lane;
    const int per = (nChunks + 3) / 4;
    const int cBeg = grp * per, cEnd = min(cBeg + per, nChunks);
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
        bestDist[qi] = (int)bd;
        bestIdx[qi] = bi;
        secondDist[qi] = (int)sd;
    }
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

}  // namespace orbfe
