// ORACLE - TEST INFRASTRUCTURE ONLY (oracle/_ref, second library: libse2lam_ref_map.so).  Pre-included (-include) in front of
// every reference source file of the MAP build, where src/Map.cpp, src/KeyFrame.cpp and src/MapPoint.cpp are compiled with
// their own headers (KeyFrame.h, MapPoint.h, Map.h are the reference's).  What is cut off here is what lies behind them and
// off the hot path: the threads (Track, LocalMapper, GlobalMapper - Map.cpp only calls four of their static functions, which
// get bodies in ref_map_driver.cpp) and the DBoW2 vocabulary template (KeyFrame::ComputeBoW calls transform()).
#pragma once
#include <climits>
#include <map>
#include <memory>
#include <set>
#include <vector>

#include <opencv2/core/core.hpp>
#include <g2o_shim.hpp>

#include "Config.h"
#include "Thirdparty/DBoW2/DBoW2/BowVector.h"
#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"

using namespace std;   // Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:34 does this at global scope, and KeyFrame.h relies on it

#define ORBVOCABULARY_H
#define TRACK_H
#define LOCALMAPPER_H
#define GLOBALMAPPER_H

namespace se2lam {

class ORBVocabulary {   // include/se2lam/ORBVocabulary.h: DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>
public:
    void transform(const std::vector<cv::Mat>&, DBoW2::BowVector&, DBoW2::FeatureVector&, int) const {}
};

class KeyFrame;
class MapPoint;
struct SE3Constraint;

class Track {           // include/se2lam/Track.h:34-36
public:
    static void calcOdoConstraintCam(const Se2& dOdo, cv::Mat& cTc, g2o::Matrix6d& Info_se3);
    static void calcSE3toXYZInfo(cv::Point3f xyz1, const cv::Mat& Tcw1, const cv::Mat& Tcw2, Eigen::Matrix3d& info1, Eigen::Matrix3d& info2);
};
class LocalMapper {};
class GlobalMapper {    // include/se2lam/GlobalMapper.h:51,70
public:
    static int CreateFeatEdge(std::shared_ptr<KeyFrame> from, std::shared_ptr<KeyFrame> to, SE3Constraint& out);
    static std::set<std::shared_ptr<KeyFrame>> GetAllConnectedKFs_nLayers(const std::shared_ptr<KeyFrame> kf, int numLayers = 10,
                                                                          std::set<std::shared_ptr<KeyFrame>> selected = std::set<std::shared_ptr<KeyFrame>>());
};

}  // namespace se2lam
