// ORBmatcher_orbfe.h -- the Hamming part of ORB_SLAM2::ORBmatcher on liborbfe (reference include/ORBmatcher.h:41-106).
//
// The reference's ORBmatcher.h keeps its class declaration (its Search* methods walk Frame / KeyFrame / MapPoint
// pointers and stay host code); what moves to liborbfe is the distance arithmetic underneath them:
//   * ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1647-1663): define it as `return orbfe_DescriptorDistance(a, b);`
//   * the row-band Hamming search of Frame::ComputeStereoMatches (src/Frame.cc:504-549): orbfe::StereoRowBand below
//   * Frame::ComputeStereoMatches as a whole (src/Frame.cc:466-640), SAD refinement and median gate included:
//     orbfe::StereoMatcher below -- no host copy of mvImagePyramid is read
//   * batched best/second-best for the BoW / projection searches: orbfe_match_knn2 / orbfe_match_csr in orbfe.h.
// INTEGRATION.md shows the two-line patches.
#pragma once
#include <hip/hip_runtime_api.h>

#include <vector>

#include "orbfe.h"
#include "orbfe_cv.h"

// body for ORBmatcher::DescriptorDistance(const cv::Mat &a, const cv::Mat &b)
static inline int orbfe_DescriptorDistance(const cv::Mat &a, const cv::Mat &b) { return orbfe_hamming(a.ptr(), b.ptr()); }

namespace orbfe {

// Device-side replacement for the loop at src/Frame.cc:504-549.  Inputs are the members Frame already holds
// (mvKeys, mDescriptors, mvKeysRight, mDescriptorsRight, mvScaleFactors, image rows, minD = 0, maxD = mbf/mb);
// outputs are bestDist / bestIdxR per left keypoint, exactly the values the reference's loop leaves in its locals.
class StereoRowBand {
public:
    explicit StereoRowBand(int device = 0) : m_(nullptr) { orbfe_matcher_create(device, &m_); }
    ~StereoRowBand() {
        for (void *p : bufs_) if (p) (void)hipFree(p);
        orbfe_matcher_destroy(m_);
    }
    int run(const std::vector<cv::KeyPoint> &kl, const cv::Mat &dl, const std::vector<cv::KeyPoint> &kr, const cv::Mat &dr,
            const std::vector<float> &scaleFactors, int nRows, float minD, float maxD, std::vector<int> &bestDist,
            std::vector<int> &bestIdxR) {
        const int nl = (int)kl.size(), nr = (int)kr.size();
        bestDist.assign(nl, 100);
        bestIdxR.assign(nl, 0);
        if (nl == 0) return ORBFE_OK;
        const size_t need[6] = {(size_t)nl * 28, (size_t)nl * 32, (size_t)(nr ? nr : 1) * 28, (size_t)(nr ? nr : 1) * 32,
                                (size_t)nl * 4, (size_t)nl * 4};
        for (int i = 0; i < 6; i++)
            if (need[i] > size_[i]) {
                if (bufs_[i]) (void)hipFree(bufs_[i]);
                if (hipMalloc(&bufs_[i], need[i]) != hipSuccess) return ORBFE_ERR_DEVICE;
                size_[i] = need[i];
            }
        (void)hipMemcpy(bufs_[0], kl.data(), need[0], hipMemcpyHostToDevice);
        for (int i = 0; i < nl; i++) (void)hipMemcpy((char *)bufs_[1] + (size_t)i * 32, dl.ptr(i), 32, hipMemcpyHostToDevice);
        if (nr) {
            (void)hipMemcpy(bufs_[2], kr.data(), (size_t)nr * 28, hipMemcpyHostToDevice);
            for (int i = 0; i < nr; i++) (void)hipMemcpy((char *)bufs_[3] + (size_t)i * 32, dr.ptr(i), 32, hipMemcpyHostToDevice);
        }
        int rc = orbfe_stereo_rowband(m_, (const orbfe_keypoint *)bufs_[0], (const uint8_t *)bufs_[1], nl,
                                      (const orbfe_keypoint *)bufs_[2], (const uint8_t *)bufs_[3], nr, scaleFactors.data(),
                                      (int)scaleFactors.size(), nRows, minD, maxD, (int32_t *)bufs_[4], (int32_t *)bufs_[5], 1);
        if (rc != ORBFE_OK) return rc;
        (void)hipMemcpy(bestDist.data(), bufs_[4], (size_t)nl * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(bestIdxR.data(), bufs_[5], (size_t)nl * 4, hipMemcpyDeviceToHost);
        return ORBFE_OK;
    }

private:
    orbfe_matcher *m_;
    void *bufs_[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t size_[6] = {0, 0, 0, 0, 0, 0};
};

// Frame::ComputeStereoMatches (src/Frame.cc:466-640) on the device.  `left` / `right` are the liborbfe handles of the
// two extractors that have just processed the stereo pair (ORBextractor::orbfeHandle()); keypoints and descriptors are
// the Frame members; outputs are Frame::mvuRight and Frame::mvDepth.
class StereoMatcher {
public:
    explicit StereoMatcher(int device = 0) : m_(nullptr) { orbfe_matcher_create(device, &m_); }
    ~StereoMatcher() {
        for (void *p : bufs_) if (p) (void)hipFree(p);
        orbfe_matcher_destroy(m_);
    }
    int ComputeStereoMatches(orbfe_extractor *left, orbfe_extractor *right, const std::vector<cv::KeyPoint> &kl, const cv::Mat &dl,
                             const std::vector<cv::KeyPoint> &kr, const cv::Mat &dr, float mbf, float mb,
                             std::vector<float> &mvuRight, std::vector<float> &mvDepth) {
        const int nl = (int)kl.size(), nr = (int)kr.size();
        mvuRight.assign(nl, -1.0f);   // :468-469
        mvDepth.assign(nl, -1.0f);
        if (nl == 0) return ORBFE_OK;
        const size_t need[6] = {(size_t)nl * 28, (size_t)nl * 32, (size_t)(nr ? nr : 1) * 28, (size_t)(nr ? nr : 1) * 32,
                                (size_t)nl * 4, (size_t)nl * 4};
        for (int i = 0; i < 6; i++)
            if (need[i] > size_[i]) {
                if (bufs_[i]) (void)hipFree(bufs_[i]);
                bufs_[i] = nullptr;
                size_[i] = 0;
                if (hipMalloc(&bufs_[i], need[i]) != hipSuccess) return ORBFE_ERR_DEVICE;
                size_[i] = need[i];
            }
        (void)hipMemcpy(bufs_[0], kl.data(), need[0], hipMemcpyHostToDevice);
        (void)hipMemcpy2D(bufs_[1], 32, dl.data, dl.step, 32, nl, hipMemcpyHostToDevice);
        if (nr) {
            (void)hipMemcpy(bufs_[2], kr.data(), (size_t)nr * 28, hipMemcpyHostToDevice);
            (void)hipMemcpy2D(bufs_[3], 32, dr.data, dr.step, 32, nr, hipMemcpyHostToDevice);
        }
        int rc = orbfe_stereo_match(m_, left, 0, right, 0, (const orbfe_keypoint *)bufs_[0], (const uint8_t *)bufs_[1], nl,
                                    (const orbfe_keypoint *)bufs_[2], (const uint8_t *)bufs_[3], nr, mbf, mb, (float *)bufs_[4],
                                    (float *)bufs_[5], 1);
        if (rc != ORBFE_OK) return rc;
        (void)hipMemcpy(mvuRight.data(), bufs_[4], (size_t)nl * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(mvDepth.data(), bufs_[5], (size_t)nl * 4, hipMemcpyDeviceToHost);
        return ORBFE_OK;
    }

private:
    orbfe_matcher *m_;
    void *bufs_[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t size_[6] = {0, 0, 0, 0, 0, 0};
};

}  // namespace orbfe
