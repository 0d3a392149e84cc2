// The expressions the reference's callers apply to an ORBextractor (src/Frame.cc:69-75, :247-253, :473, :563-580;
// src/Tracking.cc:120-126), written against THIS repository's include/ORBextractor.h: if the class loses a member or changes a
// signature this file stops compiling.  Syntax check only (tools/refcheck/Makefile).
#include "ORBextractor.h"

namespace {
void frame_constructor_uses(ORB_SLAM2::ORBextractor *ex, const cv::Mat &im) {
    int mnScaleLevels = ex->GetLevels();
    float mfScaleFactor = ex->GetScaleFactor();
    std::vector<float> a = ex->GetScaleFactors(), b = ex->GetInverseScaleFactors(), c = ex->GetScaleSigmaSquares(),
                       d = ex->GetInverseScaleSigmaSquares();
    std::vector<cv::KeyPoint> mvKeys;
    cv::Mat mDescriptors;
    (*ex)(im, cv::Mat(), mvKeys, mDescriptors);                      // Frame::ExtractORB
    const int nRows = ex->mvImagePyramid[0].rows;                    // Frame::ComputeStereoMatches
    const cv::Mat &level = ex->mvImagePyramid[mnScaleLevels - 1];
    (void)mfScaleFactor; (void)nRows; (void)level;
}
ORB_SLAM2::ORBextractor *tracking_constructor_uses(int nFeatures, float fScaleFactor, int nLevels, int fIniThFAST, int fMinThFAST) {
    return new ORB_SLAM2::ORBextractor(2 * nFeatures, fScaleFactor, nLevels, fIniThFAST, fMinThFAST);
}
}  // namespace
