// ORBVocabulary_orbfe.h -- DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>::transform on liborbfe.
//
// Stands beside the reference's include/ORBVocabulary.h (typedef of the DBoW2 template, Thirdparty/DBoW2/DBoW2/
// TemplatedVocabulary.h).  What moves to the GPU is the per-feature tree descent (:1218-1259, ~60 Hamming distances per
// descriptor down the 10-ary, 6-level ORBvoc tree); the accumulation into BowVector / FeatureVector (:1127-1200,
// BowVector.cpp:34-84, FeatureVector.cpp:29-43) is the same std::map arithmetic, done here on the host in feature order
// so that the doubles come out bit-identical.  Frame::ComputeBoW (src/Frame.cc:395-402) becomes
//     mpORBvocabulary->transform(mDescriptors, mBowVec, mFeatVec, 4);      // orbfe::Vocabulary
// BowVector / FeatureVector are template parameters: any std::map<unsigned, double> / std::map<unsigned,
// std::vector<unsigned>> subclass, i.e. DBoW2's own classes, fits.
#pragma once
#include <cmath>
#include <cstdint>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "orbfe.h"
#include "orbfe_context.h"
#include "orbfe_cv.h"

namespace orbfe {

class Vocabulary {
public:
    enum Weighting { TF_IDF = 0, TF = 1, IDF = 2, BINARY = 3 };                       // DBoW2::WeightingType
    enum Scoring { L1_NORM = 0, L2_NORM = 1, CHI_SQUARE = 2, KL = 3, BHATTACHARYYA = 4, DOT_PRODUCT = 5 };   // ScoringType

    // device < 0: the adapters' default device (environment ORBFE_DEVICE, else 0).  transform() runs on the calling thread's
    // matcher context, so Frame::ComputeBoW (Tracking thread) and KeyFrame::ComputeBoW (LocalMapping thread) may use one
    // vocabulary object at the same time, as they do with the reference's.
    explicit Vocabulary(int device = -1) : device_(device < 0 ? default_device() : device) {}
    ~Vocabulary() { orbfe_vocab_destroy(v_); }
    Vocabulary(const Vocabulary &) = delete;
    Vocabulary &operator=(const Vocabulary &) = delete;

    // The tree as flat arrays (the layout orbfe_vocab_create takes) plus, per node, word id (-1 for inner nodes) and weight.
    int set(int k, int L, int scoring, int weighting, const std::vector<int32_t> &childOff, const std::vector<int32_t> &childIdx,
            const std::vector<uint8_t> &nodeDesc, const std::vector<int32_t> &wordId, const std::vector<double> &weight) {
        k_ = k; L_ = L; scoring_ = scoring; weighting_ = weighting;
        wordId_ = wordId; weight_ = weight;
        orbfe_vocab_destroy(v_);
        v_ = nullptr;
        return orbfe_vocab_create(device_, (int)childOff.size() - 1, childOff.data(), childIdx.data(), nodeDesc.data(), L, &v_);
    }

    // TemplatedVocabulary::loadFromTextFile (:1338-1420, the ORB-SLAM2 text format: "k L scoring weighting", then one line
    // per node "parent isLeaf d0 .. d31 weight"; node ids in file order starting at 1, word ids in leaf order)
    bool loadFromTextFile(const std::string &filename) {
        std::ifstream f(filename.c_str());
        if (!f.is_open()) return false;
        std::string s;
        std::getline(f, s);
        std::stringstream ss(s);
        int k = 0, L = 0, n1 = -1, n2 = -1;
        ss >> k >> L >> n1 >> n2;
        if (k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) return false;
        std::vector<int32_t> parent(1, -1), wordId(1, -1);
        std::vector<uint8_t> desc(32, 0);
        std::vector<double> weight(1, 0.0);
        int words = 0;
        while (std::getline(f, s)) {
            std::stringstream sn(s);
            int pid, leaf;
            if (!(sn >> pid >> leaf)) continue;
            if (pid < 0 || pid >= (int)parent.size()) return false;
            parent.push_back(pid);
            for (int i = 0; i < 32; i++) { int b = 0; sn >> b; desc.push_back((uint8_t)b); }
            double w = 0;
            sn >> w;
            weight.push_back(w);
            wordId.push_back(leaf > 0 ? words++ : -1);
        }
        const int n = (int)parent.size();
        std::vector<int32_t> off(n + 1, 0), idx(n > 1 ? n - 1 : 0), fill(n, 0);
        for (int i = 1; i < n; i++) off[parent[i] + 1]++;
        for (int i = 0; i < n; i++) off[i + 1] += off[i];
        for (int i = 1; i < n; i++) idx[off[parent[i]] + fill[parent[i]]++] = i;   // children in file order
        return set(k, L, n1, n2, off, idx, desc, wordId, weight) == ORBFE_OK;
    }

    bool empty() const { return v_ == nullptr; }

    // transform(features, v, fv, levelsup), :1127-1200.  `descriptors` is the N x 32 CV_8U matrix Frame keeps
    // (mDescriptors); the reference first splits it into a vector of rows (Converter::toDescriptorVector).
    template <class BowVector, class FeatureVector>
    int transform(const cv::Mat &descriptors, BowVector &v, FeatureVector &fv, int levelsup) {
        v.clear();
        fv.clear();
        if (empty()) return ORBFE_OK;
        const int n = descriptors.rows;
        std::vector<int32_t> leaf(n), nid(n);
        if (n > 0) {
            int rc = ORBFE_OK;
            orbfe_matcher *m = thread_matcher(&rc);
            if (!m) return rc;
            rc = orbfe_bow_transform_host(m, v_, descriptors.data, descriptors.step, n, levelsup, leaf.data(), nid.data());
            if (rc != ORBFE_OK) return rc;
        }
        const bool tf = weighting_ == TF_IDF || weighting_ == TF;
        for (int i = 0; i < n; i++) {
            const unsigned id = (unsigned)wordId_[leaf[i]];
            const double w = weight_[leaf[i]];
            if (w > 0) {   // not stopped
                typename BowVector::iterator vit = v.lower_bound(id);
                if (vit != v.end() && !(v.key_comp()(id, vit->first))) {
                    if (tf) vit->second += w;                                   // addWeight; addIfNotExist leaves it
                } else {
                    v.insert(vit, typename BowVector::value_type(id, w));
                }
                typename FeatureVector::iterator fit = fv.lower_bound((unsigned)nid[i]);
                if (fit == fv.end() || fit->first != (unsigned)nid[i])
                    fit = fv.insert(fit, typename FeatureVector::value_type((unsigned)nid[i], std::vector<unsigned int>()));
                fit->second.push_back((unsigned)i);
            }
        }
        // L1 / L2 scoring objects normalise, the others do not (ScoringObject.h:74-90)
        const int norm = scoring_ == L1_NORM || scoring_ == CHI_SQUARE || scoring_ == KL || scoring_ == BHATTACHARYYA ? 1
                         : scoring_ == L2_NORM ? 2 : 0;
        if (tf && !v.empty() && norm == 0) {
            const double nd = (double)v.size();
            for (typename BowVector::iterator it = v.begin(); it != v.end(); ++it) it->second /= nd;
        }
        if (norm != 0) {
            double s = 0.0;
            if (norm == 1) { for (typename BowVector::iterator it = v.begin(); it != v.end(); ++it) s += std::fabs(it->second); }
            else { for (typename BowVector::iterator it = v.begin(); it != v.end(); ++it) s += it->second * it->second; s = std::sqrt(s); }
            if (s > 0.0) for (typename BowVector::iterator it = v.begin(); it != v.end(); ++it) it->second /= s;
        }
        return ORBFE_OK;
    }

private:
    int device_;
    orbfe_vocab *v_ = nullptr;
    int k_ = 0, L_ = 0, scoring_ = 0, weighting_ = 0;
    std::vector<int32_t> wordId_;
    std::vector<double> weight_;
};

}  // namespace orbfe
