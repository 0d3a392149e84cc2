// ORBextractor.h -- drop-in replacement for the reference's include/ORBextractor.h (abhineet123/ORB_SLAM2).
//
// Same namespace, class name, constructor, operator() signature, getters and the public mvImagePyramid member
// (reference include/ORBextractor.h:45-111), so src/Frame.cc, src/Tracking.cc and the rest of ORB-SLAM2 compile and link
// unchanged against it.  The work is done by liborbfe.so (hand-written HIP for gfx950) through the C-ABI in orbfe.h;
// adapters/ORBextractor_orbfe.cc holds the member definitions.  ExtractorNode / DistributeOctTree / ComputePyramid are
// no longer members: they were protected or internal in the reference and nothing outside ORBextractor.cc used them.
#ifndef ORBEXTRACTOR_H
#define ORBEXTRACTOR_H

#include <vector>

#include "orbfe.h"
#include "orbfe_cv.h"

namespace ORB_SLAM2 {

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
    ~ORBextractor();
    ORBextractor(const ORBextractor &) = delete;
    ORBextractor &operator=(const ORBextractor &) = delete;

    // Compute the ORB features and descriptors on an image.  Mask is ignored, as in the reference.
    void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint> &keypoints,
                    cv::OutputArray descriptors);

    int inline GetLevels() { return nlevels; }
    float inline GetScaleFactor() { return (float)scaleFactor; }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    // ROI views (19-px BORDER_REFLECT_101 frame around each) into host copies of the device pyramid; refreshed by every
    // operator() call exactly as the reference re-allocates them (src/ORBextractor.cc:1113-1115), unless the environment
    // says ORBFE_PYRAMID_MIRROR=0 (only Frame::ComputeStereoMatches reads them, src/Frame.cc:473,563,575,580).
    std::vector<cv::Mat> mvImagePyramid;

    // Not in the reference: the liborbfe handle behind this object, for the device-side stereo matcher
    // (orbfe::StereoMatcher in ORBmatcher_orbfe.h reads the pyramids of the last call in place).
    orbfe_extractor *orbfeHandle() const { return handle_; }

protected:
    int nfeatures;
    double scaleFactor;
    int nlevels;
    int iniThFAST;
    int minThFAST;
    std::vector<int> mnFeaturesPerLevel;
    std::vector<int> umax;
    std::vector<float> mvScaleFactor;
    std::vector<float> mvInvScaleFactor;
    std::vector<float> mvLevelSigma2;
    std::vector<float> mvInvLevelSigma2;

private:
    orbfe_extractor *handle_;
    std::vector<std::vector<unsigned char> > pyramidStore_;
    bool mirrorPyramid_ = true;   // environment ORBFE_PYRAMID_MIRROR=0: mvImagePyramid is left empty (nothing reads it)
};

}  // namespace ORB_SLAM2

#endif
