// k_quadtree.h -- N1: DistributeOctTree on the GPU, one workgroup per (frame, level) (gfx950).
//
// Replaces ORBextractor::DistributeOctTree / ExtractorNode::DivideNode (reference src/ORBextractor.cc:481-763).  The
// reference splits std::list nodes one at a time; what its result depends on is (a) which nodes are split in which
// pass and (b) the final list order.  Both are reproduced in pass form (the same formulation as quadtree_host.h):
//   breadth pass (:606-665)   every node with > 1 point is split;
//                             new list = reverse(children in creation order) ++ (single-point nodes, old order)
//   size-ordered pass (:676-737) expandable nodes ranked by (point count desc, creation index desc) are split until the
//                             list holds N nodes; new list = reverse(children) ++ (old list minus the split parents)
// A node is a slot in the current list.  Points never move: each candidate carries its node's slot.  With up to
// 256 * kQtRegPts candidates (the normal case) every thread keeps its candidates -- coordinates and (slot, response) --
// in registers for the whole kernel and HBM is read exactly once; larger levels fall back to the slot stored in
// Cand::pad, in place in HBM (every thread only ever re-reads the entries it wrote itself).  Per pass: LDS-atomic child histograms,
// block-wide prefix sums for the new list order, relabel.  "Keep the best response per node, first wins" (:744-760) is an
// LDS ds_max_u64 on (response << 32 | ~order) where order is the candidate's position in the reference's emission order
// (cell row, cell column, y, x), recomputed from its coordinates -- so the unordered, atomically packed candidate array
// of the FAST kernel needs no sorting.
// Ties between equal-size nodes in the size-ordered pass: the reference compares heap addresses (:684); as in the
// oracle they are ordered by creation index.
#pragma once
#include "common.h"

namespace orbfe {

struct QuadtreeArgs {
    Geom g;
    Cand *cand;              // [F][candFrameCap]; pad field is scratch
    const int *levelCount;   // [F][kMaxLevels]
    SelKp *sel;              // [F][selFrameCap]
    int *selCount;           // [F][kMaxLevels]
    int *status;             // [1]: bit 0 = candidate overflow, bit 1 = node table overflow
};

template <int MAXN>
struct QtShared {
    uint32_t boxLo[2][MAXN];        // x0 | y0 << 16
    uint32_t boxHi[2][MAXN];        // x1 | y1 << 16
    int cnt[2][MAXN];
    int child[4 * MAXN];            // child point counts of the slots being split (atomic); later the ds_max_u64 keys
    uint16_t cidx[4 * MAXN];        // creation index of each non-empty child
    uint16_t stay[MAXN];            // new slot offset of the nodes that are not split
    uint16_t rankOf[MAXN];          // size-ordered pass: rank of an expandable slot (0xFFFF otherwise)
    uint16_t slotByRank[MAXN];
    int gain[MAXN];                 // size-ordered pass: nodes added by splitting rank r (inclusive prefix after scan)
    int scanTmp[8];
    int vars[8];
};

// exclusive prefix sum over LDS ints v[0..m) (in place), returns the total; all 256 threads must call
__device__ __forceinline__ int block_scan_excl(int *v, int m, int *scanTmp) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int per = (m + 255) >> 8;
    const int b = tid * per, e = min(b + per, m);
    int s = 0;
    for (int i = b; i < e; i++) s += v[i];
    int inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) scanTmp[wv] = inc;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int t = scanTmp[k];
        if (k < wv) base += t;
        total += t;
    }
    int run = base + inc - s;
    for (int i = b; i < e; i++) {
        const int t = v[i];
        v[i] = run;
        run += t;
    }
    __syncthreads();
    return total;
}

__device__ __forceinline__ int qt_quadrant(uint32_t lo, uint32_t hi, int x, int y, int *mx, int *my) {
    const int x0 = lo & 0xffff, y0 = lo >> 16, x1 = hi & 0xffff, y1 = hi >> 16;
    *mx = x0 + ((x1 - x0 + 1) >> 1);   // UL.x + ceil((UR.x-UL.x)/2)   (:483)
    *my = y0 + ((y1 - y0 + 1) >> 1);
    return (x < *mx ? 0 : 1) + (y < *my ? 0 : 2);   // n1,n2,n3,n4 (:515-525)
}

__device__ __forceinline__ void qt_child_box(uint32_t lo, uint32_t hi, int q, uint32_t *clo, uint32_t *chi) {
    const int x0 = lo & 0xffff, y0 = lo >> 16, x1 = hi & 0xffff, y1 = hi >> 16;
    const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);
    const int cx0 = (q & 1) ? mx : x0, cx1 = (q & 1) ? x1 : mx;
    const int cy0 = (q & 2) ? my : y0, cy1 = (q & 2) ? y1 : my;
    *clo = (uint32_t)cx0 | ((uint32_t)cy0 << 16);
    *chi = (uint32_t)cx1 | ((uint32_t)cy1 << 16);
}

#ifdef ORBFE_QT_TIMING
__device__ unsigned long long g_qtTimes[256];
#define QT_MARK(i) do { if (threadIdx.x == 0 && lvl == 0 && f == 0) g_qtTimes[(i)] = wall_clock64(); } while (0)
#else
#define QT_MARK(i) do {} while (0)
#endif

constexpr int kQtRegPts = 24;   // candidates per thread held in registers (levels with more use the HBM path)

template <int MAXN, bool REG>
__device__ __forceinline__ void quadtree_run(const QuadtreeArgs &a, QtShared<MAXN> &S, int lvl, int f, int n) {
    const int tid = threadIdx.x;
    const LevelGeom &L = a.g.lv[lvl];
    const int N = L.quota;
    int *selCount = a.selCount + f * kMaxLevels + lvl;
    QT_MARK(0);
    Cand *pts = a.cand + (size_t)f * a.g.candFrameCap + L.candOff;
    const int height = (L.h - kEdge + 3) - kMinBorder;
    // the thread's candidates p = tid + 256 k:  xy = x | y << 16,  sr = slot | response << 16
    uint32_t xy[REG ? kQtRegPts : 1], sr[REG ? kQtRegPts : 1];
    if (REG) {
#pragma unroll
        for (int k = 0; k < kQtRegPts; k++) {
            // un-predicated loads (clamped index) so that all of them are in flight together; candidates past n become
            // slot 0 at (0,0), which keeps the un-predicated table reads of the sweeps legal
            const int p = tid + 256 * k;
            const Cand c = pts[min(p, n - 1)];
            xy[k] = p < n ? ((uint32_t)c.x | ((uint32_t)c.y << 16)) : 0u;
            sr[k] = p < n ? ((uint32_t)c.resp << 16) : 0u;
        }
    }
    // Register path: candidates are visited in chunks of 4.  `gather` does the chunk's LDS table reads un-predicated
    // (they are issued together, one wait), `apply` then does the predicated part.  HBM path: one candidate at a time.
    auto sweep2 = [&](auto &&gather, auto &&apply) {
        if (REG) {
#pragma unroll
            for (int c0 = 0; c0 < kQtRegPts; c0 += 4) {
                if (256 * c0 < n) {   // workgroup-uniform
                    decltype(gather(0, 0, 0)) G[4];
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        G[j] = gather((int)(xy[c0 + j] & 0xffffu), (int)(xy[c0 + j] >> 16), (int)(sr[c0 + j] & 0xffffu));
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int k = c0 + j;
                        const bool valid = tid + 256 * k < n;
                        int slot = (int)(sr[k] & 0xffffu);
                        apply(valid, (int)(xy[k] & 0xffffu), (int)(xy[k] >> 16), slot, G[j]);
                        sr[k] = (sr[k] & 0xffff0000u) | (valid ? (uint32_t)slot : 0u);
                    }
                }
            }
        } else {
            for (int p = tid; p < n; p += 256) {
                const Cand c = pts[p];
                int slot = c.pad;
                const auto G = gather((int)c.x, (int)c.y, slot);
                apply(true, (int)c.x, (int)c.y, slot, G);
                if (slot != (int)c.pad) pts[p].pad = (uint16_t)slot;
            }
        }
    };
    // visit every candidate of this thread: fn(x, y, slot) may update slot
    auto sweep = [&](auto &&fn) {
        if (REG) {
#pragma unroll
            for (int k = 0; k < kQtRegPts; k++) {
                if (tid + 256 * k < n) {
                    int slot = (int)(sr[k] & 0xffffu);
                    fn((int)(xy[k] & 0xffffu), (int)(xy[k] >> 16), slot);
                    sr[k] = (sr[k] & 0xffff0000u) | (uint32_t)slot;
                }
            }
        } else {
            for (int p = tid; p < n; p += 256) {
                const Cand c = pts[p];
                int slot = c.pad;
                fn((int)c.x, (int)c.y, slot);
                if (slot != (int)c.pad) pts[p].pad = (uint16_t)slot;
            }
        }
    };

    // ---- roots (:543-585): root r = (int)(x / hX); empty roots are dropped, list order = ascending r
    const int nIni = L.nIni;
    for (int i = tid; i < nIni; i += 256) S.child[i] = 0;
    __syncthreads();
    if (REG) {
        // root of every candidate into its slot field, then one wave-wide count per root (a few roots, thousands of
        // candidates: per-candidate LDS atomics would all hit the same few words)
#pragma unroll
        for (int k = 0; k < kQtRegPts; k++) {
            const int r = min((int)__fdiv_rn((float)(xy[k] & 0xffffu), L.hX), nIni - 1);
            sr[k] = (sr[k] & 0xffff0000u) | (uint32_t)r;
        }
        for (int r = 0; r < nIni; r++) {
            int c = 0;
#pragma unroll
            for (int k = 0; k < kQtRegPts; k++) c += (tid + 256 * k < n && (int)(sr[k] & 0xffffu) == r) ? 1 : 0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
            if ((tid & 63) == 0 && c) atomicAdd(&S.child[r], c);
        }
    } else {
        sweep([&](int x, int, int &) {
            int r = (int)__fdiv_rn((float)x, L.hX);
            r = min(r, nIni - 1);
            atomicAdd(&S.child[r], 1);
        });
    }
    __syncthreads();
    if (tid == 0) {
        int slot = 0;
        for (int r = 0; r < nIni; r++) {
            const int c = S.child[r];
            S.stay[r] = (uint16_t)slot;
            if (c == 0) continue;
            const int x0 = (int)__fmul_rn(L.hX, (float)r), x1 = (int)__fmul_rn(L.hX, (float)(r + 1));
            S.boxLo[0][slot] = (uint32_t)x0;
            S.boxHi[0][slot] = (uint32_t)x1 | ((uint32_t)height << 16);
            S.cnt[0][slot] = c;
            slot++;
        }
        S.vars[0] = slot;
    }
    __syncthreads();
    sweep([&](int x, int, int &slot) {
        int r = (int)__fdiv_rn((float)x, L.hX);
        r = min(r, nIni - 1);
        slot = S.stay[r];
    });
    int cur = 0;
    int Lc = S.vars[0];
    __syncthreads();

    bool sized = false;   // false: breadth passes, true: size-ordered passes
    QT_MARK(1);
    for (int iter = 0; iter < 64; iter++) {
        QT_MARK(8 + iter * 8);
        const uint32_t *bLo = S.boxLo[cur], *bHi = S.boxHi[cur];
        const int *cnt = S.cnt[cur];
        // ---- child histograms of every expandable slot
        for (int i = tid; i < 4 * Lc; i += 256) S.child[i] = 0;
        __syncthreads();
        struct HistG { uint32_t lo, hi; int c; };
        sweep2([&](int, int, int s) { return HistG{bLo[s], bHi[s], cnt[s]}; },
               [&](bool valid, int x, int y, int &s, const HistG &G) {
                   if (valid && G.c > 1) {
                       int mx, my;
                       const int q = qt_quadrant(G.lo, G.hi, x, y, &mx, &my);
                       atomicAdd(&S.child[4 * s + q], 1);
                   }
               });
        __syncthreads();

        QT_MARK(8 + iter * 8 + 1);
        int rStar = 0x7fffffff;   // size-ordered: last rank that is split
        if (sized) {
            // ---- rank expandable slots by (count desc, slot asc) == (count, creation index) walked from the back
            // key = (count << 11 | 2047 - slot) for expandable slots, 0 otherwise: rank = number of larger keys
            int *keys = reinterpret_cast<int *>(S.cidx);   // 2*MAXN ints, not in use before the creation-order step
            const int Lc4 = (Lc + 3) & ~3;
            for (int s = tid; s < Lc4; s += 256) {
                const int c = s < Lc ? cnt[s] : 0;
                keys[s] = c > 1 ? ((c << 11) | (2047 - s)) : 0;
            }
            __syncthreads();
            for (int s = tid; s < Lc; s += 256) {
                int rk = 0xFFFF;
                const int key = keys[s];
                if (key) {
                    rk = 0;
#pragma unroll 4
                    for (int t = 0; t < Lc4; t += 4) {
                        const int4 v = *reinterpret_cast<const int4 *>(keys + t);
                        rk += (v.x > key) + (v.y > key) + (v.z > key) + (v.w > key);
                    }
                    S.slotByRank[rk] = (uint16_t)s;
                    const int ne = (S.child[4 * s] > 0) + (S.child[4 * s + 1] > 0) + (S.child[4 * s + 2] > 0) + (S.child[4 * s + 3] > 0);
                    S.gain[rk] = ne - 1;
                }
                S.rankOf[s] = (uint16_t)rk;
            }
            __syncthreads();   // keys (aliasing cidx) are dead from here
            if (tid == 0) S.vars[1] = 0x7fffffff;
            __syncthreads();
            int E = 0;   // number of expandable slots: they occupy ranks 0..E-1
            for (int s = tid; s < Lc; s += 256) E += cnt[s] > 1;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) E += __shfl_xor(E, o);
            if ((tid & 63) == 0) S.scanTmp[4 + (tid >> 6)] = E;
            __syncthreads();
            E = S.scanTmp[4] + S.scanTmp[5] + S.scanTmp[6] + S.scanTmp[7];
            block_scan_excl(S.gain, E, S.scanTmp);   // gain[r] = nodes added by ranks < r
            for (int r = tid; r < E; r += 256) {
                const int s = S.slotByRank[r];
                const int ne = (S.child[4 * s] > 0) + (S.child[4 * s + 1] > 0) + (S.child[4 * s + 2] > 0) + (S.child[4 * s + 3] > 0);
                if (Lc + S.gain[r] + ne - 1 >= N) atomicMin(&S.vars[1], r);   // list reaches N after splitting rank r (:730)
            }
            __syncthreads();
            rStar = min(S.vars[1], E - 1);
            __syncthreads();
        }

        // ---- creation order of the children.  breadth: parents in list order; size-ordered: parents in rank order
        //      cidx is indexed by 4*slot+q in both cases; the scan runs over the creation sequence.
        //      Use S.gain as the scan scratch is not possible (needed above), so scan in S.child's shadow: cidx via ints
        //      in a temporary view of `stay`/... -> keep it simple: scan over an int array aliased on the next box buffer.
        QT_MARK(8 + iter * 8 + 2);
        int *scanBuf = reinterpret_cast<int *>(S.boxLo[cur ^ 1]);   // 4*MAXN bytes = MAXN ints: not enough for 4*Lc flags
        (void)scanBuf;
        // flags are packed per parent: nonempty-children count (0..4) per creation position, then expanded per child
        int *perParent = S.cnt[cur ^ 1];   // MAXN ints, free until the new generation is written
        const int nPar = sized ? (rStar + 1) : Lc;
        for (int i = tid; i < nPar; i += 256) {
            const int s = sized ? (int)S.slotByRank[i] : i;
            int ne = 0;
            if (cnt[s] > 1)
                ne = (S.child[4 * s] > 0) + (S.child[4 * s + 1] > 0) + (S.child[4 * s + 2] > 0) + (S.child[4 * s + 3] > 0);
            perParent[i] = ne;
        }
        __syncthreads();
        const int C = block_scan_excl(perParent, nPar, S.scanTmp);
        int nExp = 0;
        for (int i = tid; i < nPar; i += 256) {
            const int s = sized ? (int)S.slotByRank[i] : i;
            if (cnt[s] > 1) {
                int k = perParent[i];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int c = S.child[4 * s + q];
                    if (c > 0) S.cidx[4 * s + q] = (uint16_t)k++;
                    nExp += c > 1;
                }
            }
        }
        QT_MARK(8 + iter * 8 + 3);
        // ---- nodes that stay: single-point nodes (breadth) / everything not split (size-ordered), old order
        int *stayFlag = S.gain;   // free again
        for (int s = tid; s < Lc; s += 256) {
            const bool split = sized ? (S.rankOf[s] <= rStar) : (cnt[s] > 1);
            stayFlag[s] = split ? 0 : 1;
        }
        __syncthreads();
        const int K = block_scan_excl(stayFlag, Lc, S.scanTmp);
        const int newL = C + K;
        if (newL > MAXN) {   // cannot happen for quota+3 <= MAXN; refuse rather than corrupt LDS
            if (tid == 0) { atomicOr(a.status, 2); *selCount = 0; }
            return;
        }
        // count expandable children (breadth pass decision :673)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nExp += __shfl_xor(nExp, o);
        if ((tid & 63) == 0) S.scanTmp[4 + (tid >> 6)] = nExp;
        QT_MARK(8 + iter * 8 + 4);
        // ---- relabel the points
        struct RelG { uint32_t lo, hi; int key, stay; };
        sweep2([&](int, int, int s) { return RelG{bLo[s], bHi[s], sized ? (int)S.rankOf[s] : cnt[s], stayFlag[s]}; },
               [&](bool, int x, int y, int &s, const RelG &G) {
                   const bool split = sized ? (G.key <= rStar) : (G.key > 1);
                   int mx, my;
                   const int q = qt_quadrant(G.lo, G.hi, x, y, &mx, &my);
                   const int ci = S.cidx[4 * s + q];   // stale for nodes that are not split: read but not used
                   s = split ? C - 1 - ci : C + G.stay;
               });
        QT_MARK(8 + iter * 8 + 5);
        // ---- write the next generation
        uint32_t *nLo = S.boxLo[cur ^ 1], *nHi = S.boxHi[cur ^ 1];
        __syncthreads();   // perParent (aliases cnt[cur^1]) fully consumed before it is overwritten
        int *nCnt = S.cnt[cur ^ 1];
        for (int s = tid; s < Lc; s += 256) {
            const bool split = sized ? (S.rankOf[s] <= rStar) : (cnt[s] > 1);
            if (split) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int c = S.child[4 * s + q];
                    if (c > 0) {
                        const int ns = C - 1 - S.cidx[4 * s + q];
                        uint32_t lo, hi;
                        qt_child_box(bLo[s], bHi[s], q, &lo, &hi);
                        nLo[ns] = lo;
                        nHi[ns] = hi;
                        nCnt[ns] = c;
                    }
                }
            } else {
                const int ns = C + stayFlag[s];
                nLo[ns] = bLo[s];
                nHi[ns] = bHi[s];
                nCnt[ns] = cnt[s];
            }
        }
        __syncthreads();
        nExp = S.scanTmp[4] + S.scanTmp[5] + S.scanTmp[6] + S.scanTmp[7];
        const int prevL = Lc;
        Lc = newL;
        cur ^= 1;
        __syncthreads();
        QT_MARK(8 + iter * 8 + 6);
#ifdef ORBFE_QT_TIMING
        if (threadIdx.x == 0 && lvl == 0 && f == 0) { g_qtTimes[8 + iter * 8 + 7] = ((unsigned long long)Lc << 32) | (sized ? 1u : 0u); g_qtTimes[2] = iter + 1; }
#endif
        // ---- stop / switch rule (:669-673, :734)
        if (Lc >= N || Lc == prevL) break;
        if (!sized && Lc + 3 * nExp > N) sized = true;
    }

    QT_MARK(3);
    // ---- best response per node, first in reference order wins (:744-760)
    unsigned long long *best = reinterpret_cast<unsigned long long *>(S.child);
    for (int i = tid; i < Lc; i += 256) best[i] = 0ull;
    __syncthreads();
    // a / b == mulhi(a, ceil(2^32 / b)) for a * b < 2^32 (coordinates < 65536, cells of 2..64 px)
    const uint32_t rcpH = (uint32_t)(((1ull << 32) + L.hCell - 1) / L.hCell), rcpW = (uint32_t)(((1ull << 32) + L.wCell - 1) / L.wCell);
    auto vote = [&](int x, int y, int slot, uint32_t resp) {
        // reference emission order: cell row, cell column, then row-major inside the cell (src/ORBextractor.cc:789-829)
        const int ci = (int)__umulhi((uint32_t)(y - 3), rcpH), cj = (int)__umulhi((uint32_t)(x - 3), rcpW);
        const int yin = y - 3 - ci * L.hCell, xin = x - 3 - cj * L.wCell;
        const uint32_t order = ((uint32_t)(ci * L.nCols + cj) << 12) | ((uint32_t)yin << 6) | (uint32_t)xin;
        const unsigned long long key = ((unsigned long long)(resp + 1u) << 32) | (unsigned long long)(0xFFFFFFFFu - order);
        atomicMax(&best[slot], key);
    };
    if (REG) {
#pragma unroll
        for (int k = 0; k < kQtRegPts; k++)
            if (tid + 256 * k < n) vote((int)(xy[k] & 0xffffu), (int)(xy[k] >> 16), (int)(sr[k] & 0xffffu), sr[k] >> 16);
    } else {
        for (int p = tid; p < n; p += 256) {
            const Cand c = pts[p];
            vote(c.x, c.y, c.pad, c.resp);
        }
    }
    __syncthreads();
    SelKp *out = a.sel + (size_t)f * a.g.selFrameCap + L.selOff;
    for (int i = tid; i < Lc && i < L.selCap; i += 256) {
        const unsigned long long key = best[i];
        const uint32_t order = 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull);
        const int cell = order >> 12, yin = (order >> 6) & 63, xin = order & 63;
        const int ci = cell / L.nCols, cj = cell - ci * L.nCols;
        SelKp k;
        k.x = (uint16_t)(cj * L.wCell + 3 + xin + kMinBorder);   // += minBorderX (:843)
        k.y = (uint16_t)(ci * L.hCell + 3 + yin + kMinBorder);
        k.resp = (uint16_t)((key >> 32) - 1);
        k.level = (uint16_t)lvl;
        out[i] = k;
    }
    if (tid == 0) *selCount = min(Lc, L.selCap);
    QT_MARK(4);
#ifdef ORBFE_QT_TIMING
    if (threadIdx.x == 0 && lvl == 0 && f == 0) g_qtTimes[5] = n;
#endif
}

// 4 workgroups per CU (128 registers, 7 of them spilled) instead of the 3 that the natural 145 registers allow: the trees
// are latency-bound (a chain of barrier-separated LDS steps), so residency is what counts: 0.213 -> 0.176 ms per 512
// frames; a 5th (96 registers, 41 spilled) is slower again.
template <int MAXN>
__global__ __launch_bounds__(256, 4) void quadtree_kernel(QuadtreeArgs a) {
    extern __shared__ __align__(16) uint8_t qt_lds_raw[];
    QtShared<MAXN> &S = *reinterpret_cast<QtShared<MAXN> *>(qt_lds_raw);
    const int tid = threadIdx.x;
    // Workgroups are dealt to the 8 XCDs round-robin in launch order (x fastest).  With (level, frame) = (x, y) and 8 levels
    // every level-0 tree -- the longest -- would land on XCD 0 and every level-7 tree on XCD 7.  The launch index is
    // therefore read level-major: all level-0 trees first, spread over the XCDs, then level 1, ... (longest first, which
    // also packs the tail).
    const int id = blockIdx.y * gridDim.x + blockIdx.x, nF = gridDim.y;
    const int lvl = id / nF, f = id - lvl * nF;
    const LevelGeom &L = a.g.lv[lvl];
    int n = a.levelCount[f * kMaxLevels + lvl];
    if (n > L.candCap) {
        if (tid == 0) atomicOr(a.status, 1);
        n = L.candCap;
    }
    if (n == 0) {
        if (tid == 0) a.selCount[f * kMaxLevels + lvl] = 0;
        return;
    }
    if (n <= 256 * kQtRegPts)
        quadtree_run<MAXN, true>(a, S, lvl, f, n);
    else
        quadtree_run<MAXN, false>(a, S, lvl, f, n);
}

}  // namespace orbfe
