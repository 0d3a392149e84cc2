// quadtree_host.h -- host DistributeOctTree in pass form (flat arrays, no std::list).
//
// Replaces ORBextractor::DistributeOctTree / ExtractorNode::DivideNode (reference src/ORBextractor.cc:481-763).
// The reference keeps a std::list<ExtractorNode>, splits nodes one at a time and push_front()s the children; the result
// depends only on (a) which nodes get split in which pass and (b) the final list order.  Both are reproduced here
// without a list:
//   * breadth pass (:606-665): every node with more than one point is split; the new list is
//       reverse(children in creation order) ++ (single-point nodes in their old order);
//   * size-ordered pass (:676-737): the expandable nodes of the previous pass are sorted by (point count, creation
//     index) and split from the back until the list holds N nodes; new list =
//       reverse(children in creation order) ++ (old list minus the split parents).
// Tie-break between equal-size nodes: the reference sorts pair<int, ExtractorNode*>, i.e. by heap address, which is not
// a function of the input; this implementation (and the oracle) orders ties by creation index.
// Points stay in one index array, stably partitioned by node, so "first maximum response wins" (:752) sees the points
// of a node in the reference's order.
#pragma once
#include <algorithm>
#include <cstdint>
#include <utility>
#include <vector>

#include "common.h"

namespace orbfe {

struct QtNode {
    int x0, y0, x1, y1;   // UL = (x0,y0), BR = (x1,y1)
    int begin, end;       // range in the point index array
    bool alive;
};

class QuadtreeHost {
public:
    // cand: n candidates in reference order; returns kept candidates in list order
    void run(const Cand *cand, int n, int width, int height, int N, int nIni, float hX, std::vector<Cand> &out) {
        out.clear();
        idx_.resize(n);
        tmp_.resize(n);
        nodes_.clear();
        order_.clear();
        // roots (:552-570)
        std::vector<int> rootCount(nIni, 0);
        for (int i = 0; i < n; i++) rootCount[root_of(cand[i].x, hX, nIni)]++;
        std::vector<int> rootBegin(nIni + 1, 0);
        for (int r = 0; r < nIni; r++) rootBegin[r + 1] = rootBegin[r] + rootCount[r];
        {
            std::vector<int> fill(rootBegin.begin(), rootBegin.end() - 1);
            for (int i = 0; i < n; i++) idx_[fill[root_of(cand[i].x, hX, nIni)]++] = i;
        }
        for (int r = 0; r < nIni; r++) {
            if (rootCount[r] == 0) continue;   // :581-582
            QtNode q;
            q.x0 = (int)(hX * static_cast<float>(r));
            q.x1 = (int)(hX * static_cast<float>(r + 1));
            q.y0 = 0;
            q.y1 = height;
            q.begin = rootBegin[r];
            q.end = rootBegin[r + 1];
            q.alive = true;
            order_.push_back((int)nodes_.size());
            nodes_.push_back(q);
        }
        (void)width;
        bool finish = false;
        std::vector<int> created, expandable, next;
        while (!finish) {
            const int prevSize = (int)order_.size();
            created.clear();
            expandable.clear();
            next.clear();
            for (int id : order_) {
                if (count(id) == 1) { next.push_back(id); continue; }
                divide(cand, id, created, expandable);
            }
            rebuild(created, next);
            const int size = (int)order_.size();
            if (size >= N || size == prevSize) {
                finish = true;
            } else if (size + 3 * (int)expandable.size() > N) {
                while (!finish) {
                    const int prev2 = (int)order_.size();
                    // (size, creation index) ascending, walked from the back (:684-685)
                    std::vector<std::pair<int, int>> byrank(expandable.size());
                    for (size_t k = 0; k < expandable.size(); k++) byrank[k] = std::make_pair(count(expandable[k]), (int)k);
                    std::sort(byrank.begin(), byrank.end());
                    std::vector<int> prevExp;
                    prevExp.swap(expandable);
                    created.clear();
                    int cur = prev2;
                    for (int k = (int)byrank.size() - 1; k >= 0; k--) {
                        const int id = prevExp[byrank[k].second];
                        const size_t before = created.size();
                        divide(cand, id, created, expandable);
                        nodes_[id].alive = false;
                        cur += (int)(created.size() - before) - 1;
                        if (cur >= N) break;
                    }
                    next.clear();
                    for (int id : order_)
                        if (nodes_[id].alive) next.push_back(id);
                    rebuild(created, next);
                    if ((int)order_.size() >= N || (int)order_.size() == prev2) finish = true;
                }
            }
        }
        // best response per node, first maximum wins (:744-760)
        out.reserve(order_.size());
        for (int id : order_) {
            const QtNode &q = nodes_[id];
            int best = idx_[q.begin];
            for (int k = q.begin + 1; k < q.end; k++)
                if (cand[idx_[k]].resp > cand[best].resp) best = idx_[k];
            out.push_back(cand[best]);
        }
    }

private:
    static int root_of(uint16_t x, float hX, int nIni) {
        int r = (int)((float)x / hX);   // vpIniNodes[kp.pt.x/hX] (:569)
        return r < nIni ? r : nIni - 1;
    }
    int count(int id) const { return nodes_[id].end - nodes_[id].begin; }

    // DivideNode (:481-537) + the four push_front blocks (:621-660): children appended to `created` in n1..n4 order
    void divide(const Cand *cand, int id, std::vector<int> &created, std::vector<int> &expandable) {
        const QtNode p = nodes_[id];
        const int mx = p.x0 + ((p.x1 - p.x0 + 1) >> 1);   // UL.x + ceil((UR.x-UL.x)/2)
        const int my = p.y0 + ((p.y1 - p.y0 + 1) >> 1);
        int cnt[4] = {0, 0, 0, 0};
        for (int k = p.begin; k < p.end; k++) cnt[quadrant(cand[idx_[k]], mx, my)]++;
        int start[4], fill[4];
        start[0] = p.begin;
        for (int c = 1; c < 4; c++) start[c] = start[c - 1] + cnt[c - 1];
        for (int c = 0; c < 4; c++) fill[c] = start[c];
        for (int k = p.begin; k < p.end; k++) tmp_[fill[quadrant(cand[idx_[k]], mx, my)]++] = idx_[k];
        std::copy(tmp_.begin() + p.begin, tmp_.begin() + p.end, idx_.begin() + p.begin);
        const int bx[4][2] = {{p.x0, mx}, {mx, p.x1}, {p.x0, mx}, {mx, p.x1}};
        const int by[4][2] = {{p.y0, my}, {p.y0, my}, {my, p.y1}, {my, p.y1}};
        for (int c = 0; c < 4; c++) {
            if (cnt[c] == 0) continue;
            QtNode q;
            q.x0 = bx[c][0]; q.x1 = bx[c][1];
            q.y0 = by[c][0]; q.y1 = by[c][1];
            q.begin = start[c];
            q.end = start[c] + cnt[c];
            q.alive = true;
            const int nid = (int)nodes_.size();
            nodes_.push_back(q);
            created.push_back(nid);
            if (cnt[c] > 1) expandable.push_back(nid);
        }
    }
    static int quadrant(const Cand &c, int mx, int my) {
        // n1: x<mx,y<my; n2: x>=mx,y<my; n3: x<mx,y>=my; n4: rest (:515-525)
        return (c.x < mx ? 0 : 1) + (c.y < my ? 0 : 2);
    }
    void rebuild(const std::vector<int> &created, const std::vector<int> &kept) {
        order_.clear();
        order_.insert(order_.end(), created.rbegin(), created.rend());
        order_.insert(order_.end(), kept.begin(), kept.end());
    }

    std::vector<int> idx_, tmp_, order_;
    std::vector<QtNode> nodes_;
};

}  // namespace orbfe
