// ORBextractor_orbfe.cc -- member definitions of ORB_SLAM2::ORBextractor (include/ORBextractor.h of this repository) on
// top of the liborbfe C-ABI.  Compile this INSTEAD of the reference's src/ORBextractor.cc and link liborbfe.so.
//
// Error behaviour mirrors the reference (src/ORBextractor.cc:1046-1050): an empty image returns with the outputs
// untouched, a non-8UC1 image trips an assert, and since the reference has no error channel any liborbfe failure (no
// GPU, unsupported size) prints the C-ABI message and aborts.
#include <cassert>
#include <cstdio>
#include <cstdlib>

#include "ORBextractor.h"

namespace ORB_SLAM2 {

static void orbfe_die(const char *what) {
    std::fprintf(stderr, "ORBextractor(orbfe): %s: %s\n", what, orbfe_last_error());
    std::abort();
}

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST),
      handle_(nullptr) {
    orbfe_params p;
    p.nfeatures = _nfeatures;
    p.scale_factor = _scaleFactor;
    p.nlevels = _nlevels;
    p.ini_th_fast = _iniThFAST;
    p.min_th_fast = _minThFAST;
    int device = 0;
    if (const char *d = std::getenv("ORBFE_DEVICE")) device = std::atoi(d);
    if (orbfe_extractor_create(&p, device, &handle_) != ORBFE_OK) orbfe_die("create");
    mvScaleFactor.resize(nlevels);
    mvInvScaleFactor.resize(nlevels);
    mvLevelSigma2.resize(nlevels);
    mvInvLevelSigma2.resize(nlevels);
    mnFeaturesPerLevel.resize(nlevels);
    umax.resize(16);
    if (orbfe_get_tables(handle_, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(),
                         mvInvLevelSigma2.data(), mnFeaturesPerLevel.data(), umax.data()) != ORBFE_OK)
        orbfe_die("tables");
    mvImagePyramid.resize(nlevels);
    pyramidStore_.resize(nlevels);
    if (const char *mir = std::getenv("ORBFE_PYRAMID_MIRROR")) mirrorPyramid_ = std::atoi(mir) != 0;
}

ORBextractor::~ORBextractor() { orbfe_extractor_destroy(handle_); }

void ORBextractor::operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint> &_keypoints,
                              cv::OutputArray _descriptors) {
    if (_image.empty()) return;
    cv::Mat image = _image.getMat();
    assert(image.type() == CV_8UC1);

    static_assert(sizeof(cv::KeyPoint) == sizeof(orbfe_keypoint), "cv::KeyPoint must be 28 bytes");
    int cap = nfeatures + 4 * nlevels + 64, n = 0;
    std::vector<orbfe_keypoint> kps(cap);
    std::vector<unsigned char> desc((size_t)cap * 32);
    int rc = orbfe_extract(handle_, image.data, image.cols, image.rows, (int)image.step, kps.data(), cap, desc.data(), &n);
    if (rc == ORBFE_ERR_CAPACITY && n > cap) {
        cap = n;
        kps.resize(cap);
        desc.resize((size_t)cap * 32);
        rc = orbfe_extract(handle_, image.data, image.cols, image.rows, (int)image.step, kps.data(), cap, desc.data(), &n);
    }
    if (rc != ORBFE_OK) orbfe_die("extract");

    if (n == 0) {
        _descriptors.release();
    } else {
        _descriptors.create(n, 32, CV_8U);
        cv::Mat d = _descriptors.getMat();
        for (int i = 0; i < n; i++) std::memcpy(d.ptr(i), &desc[(size_t)i * 32], 32);
    }
    _keypoints.clear();
    _keypoints.resize(n);
    if (n) std::memcpy((void *)_keypoints.data(), kps.data(), (size_t)n * sizeof(orbfe_keypoint));

    // mvImagePyramid: host copies with the reference's 19-px frame; the Mat is the ROI inside it (:1113-1115).  Only
    // Frame::ComputeStereoMatches reads them (src/Frame.cc:473,563,575,580): with its body replaced by orbfe::StereoMatcher
    // (INTEGRATION.md section 3), or in a monocular / RGB-D system, ORBFE_PYRAMID_MIRROR=0 saves the copy.
    if (!mirrorPyramid_) return;
    std::vector<unsigned char *> dst(nlevels);
    std::vector<size_t> bytes(nlevels);
    std::vector<int> lw(nlevels), lh(nlevels);
    for (int l = 0; l < nlevels; l++) {
        if (orbfe_pyramid_level(handle_, 0, l, nullptr, 0, &lw[l], &lh[l]) != ORBFE_OK) orbfe_die("pyramid size");
        pyramidStore_[l].resize((size_t)(lw[l] + 38) * (lh[l] + 38));
        dst[l] = pyramidStore_[l].data();
        bytes[l] = pyramidStore_[l].size();
    }
    if (orbfe_pyramid_mirror(handle_, 0, dst.data(), bytes.data(), nlevels) != ORBFE_OK) orbfe_die("pyramid copy");
    for (int l = 0; l < nlevels; l++)
        mvImagePyramid[l] = cv::Mat(lh[l], lw[l], CV_8UC1, pyramidStore_[l].data() + (size_t)19 * (lw[l] + 38) + 19, (size_t)(lw[l] + 38));
}

}  // namespace ORB_SLAM2
