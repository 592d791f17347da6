// ORACLE - TEST INFRASTRUCTURE ONLY (oracle/_ref).  Pre-included (-include) in front of every reference source file.
//
// src/ORBmatcher.cpp, src/Frame.cpp and include/se2lam/optimizer.h reach KeyFrame.h, MapPoint.h and Track.h through their
// includes; those headers need the rest of the map data model, which is not on the hot path (converter.h is the reference's
// own: it compiles against g2o_shim.hpp).  This file
// defines their include guards, so the reference's own files are skipped where they are included, and puts in their
// place the few members the two sources actually read - PtrKeyFrame / PtrMapPoint, KeyFrame : Frame with the observation
// queries and the DBoW2 feature vector, MapPoint with position, main descriptor and main octave.  Frame.h, Config.h,
// ORBextractor.h, ORBmatcher.h, cvutil.h and DBoW2's BowVector.h / FeatureVector.h are the reference's own files.
#pragma once
#include <map>
#include <memory>
#include <set>
#include <vector>

#include <opencv2/core/core.hpp>

#include "Frame.h"                                    // /root/reference/include/se2lam/Frame.h (with Config.h, ORBextractor.h)
#include "Thirdparty/DBoW2/DBoW2/BowVector.h"         // /root/reference/Thirdparty/DBoW2/DBoW2
#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"

#define MAPPOINT_H
#define KEYFRAME_H
#define TRACK_H

namespace se2lam {

using std::shared_ptr;

class KeyFrame;
typedef std::shared_ptr<KeyFrame> PtrKeyFrame;

// the members ORBmatcher.cpp reads (include/se2lam/MapPoint.h:29-43)
class MapPoint {
public:
    bool isGoodPrl() { return mbGoodParallax; }
    bool isNull() { return mbNull; }
    cv::Point3f getPos() { return mPos; }
    cv::Mat mMainDescriptor;
    int mMainOctave = 0;
    // set by the driver
    cv::Point3f mPos;
    bool mbGoodParallax = true;
    bool mbNull = false;
};
typedef std::shared_ptr<MapPoint> PtrMapPoint;

// the members ORBmatcher.cpp reads (include/se2lam/KeyFrame.h:42-128); everything else comes from Frame
class KeyFrame : public Frame {
public:
    bool isNull() { return mbNull; }
    bool hasObservation(const PtrMapPoint& pMP) { return mObservations.count(pMP) != 0; }
    bool hasObservation(int idx) { return mDualObservations.count(idx) != 0; }
    DBoW2::FeatureVector GetFeatureVector() { return mFeatVec; }
    std::vector<PtrMapPoint> GetMapPointMatches() {   // src/KeyFrame.cpp:266-276: the observed map point of every key point, or null
        std::vector<PtrMapPoint> ret(N, nullptr);
        for (auto& kv : mDualObservations) ret[kv.first] = kv.second;
        return ret;
    }
    DBoW2::FeatureVector mFeatVec;
    // set by the driver
    std::map<PtrMapPoint, int> mObservations;
    std::map<int, PtrMapPoint> mDualObservations;
    bool mbNull = false;
};

}  // namespace se2lam
