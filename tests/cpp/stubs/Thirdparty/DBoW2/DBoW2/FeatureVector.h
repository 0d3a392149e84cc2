// Stand-in for the reference's Thirdparty/DBoW2/DBoW2/FeatureVector.h (a std::map from vocabulary node id to the indices
// of the features that fell into that node), for compiling adapters/ORBmatcher_orbfe.cc in an image without DBoW2's
// dependencies.  Test infrastructure; a build inside ORB-SLAM2 picks up the real header.
#pragma once
#include <map>
#include <vector>

namespace DBoW2 {

typedef unsigned int NodeId;

class FeatureVector : public std::map<NodeId, std::vector<unsigned int> > {
public:
    void addFeature(NodeId id, unsigned int i_feature) { (*this)[id].push_back(i_feature); }
};

}  // namespace DBoW2
