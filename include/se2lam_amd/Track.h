// Front-end geometry of se2lam::Track on the device - header-only mirror over the C ABI (include/se2gpu.h).
//   Track::removeOutliers  /root/reference/src/Track.cpp:308-344   (cv::findFundamentalMat RANSAC mask, < 10 inliers => none)
//   Track::doTriangulate   /root/reference/src/Track.cpp:378-419   (cvu::triangulate, Config::acceptDepth, cvu::checkParallax)
// The reference loops over the matches on the host, one 4x4 SVD each; here all matches of the frame pair go through one
// kernel.  Map-point bookkeeping (mLocalMPs[i] = mpKF->mViewMPs[i] for features that already have an observation)
// stays with the caller, exactly where the reference has it.
#pragma once
#include <cstdint>
#include <vector>

#include "../se2gpu.h"
#include "types.h"

namespace se2lam_amd {

struct TriangulationResult {
    std::vector<Point3f> localMPs;      // mLocalMPs[i] for the features triangulated now (zero elsewhere)
    std::vector<uint8_t> goodPrl;       // mvbGoodPrl
    int nGoodPrl = 0;                   // mnGoodPrl
    int nTrackedOld = 0;                // return value of Track::doTriangulate
};

// keyPointsUnRef / keyPointsUnCur: undistorted key points of the key frame and of the current frame; matchIdx =
// Track::mMatchIdx (set to -1 where the depth gate rejects the point); hasObservation[i] = mpKF->hasObservation(i);
// PrjMtrxEye = Config::PrjMtrxEye, P = Config::Kcam * mFrame.Tcr.rowRange(0,3) (3x4 row-major float);
// Ocam = translation of cvu::inv(mFrame.Tcr).
inline TriangulationResult doTriangulate(const std::vector<KeyPoint>& keyPointsUnRef,
                                         const std::vector<KeyPoint>& keyPointsUnCur, std::vector<int>& matchIdx,
                                         const std::vector<uint8_t>& hasObservation, const float PrjMtrxEye[12],
                                         const float P[12], const float Ocam[3], float lowerDepth, float upperDepth,
                                         int minDegree = 2) {
    TriangulationResult r;
    const int n = (int)keyPointsUnRef.size();
    r.localMPs.assign(n, Point3f{0, 0, 0});
    r.goodPrl.assign(n, 0);
    if (n == 0) return r;
    check(se2gpu_triangulate(n, reinterpret_cast<const se2gpu_keypoint*>(keyPointsUnRef.data()),
                             reinterpret_cast<const se2gpu_keypoint*>(keyPointsUnCur.data()),
                             (int)keyPointsUnCur.size(), matchIdx.data(),
                             hasObservation.empty() ? nullptr : hasObservation.data(), PrjMtrxEye, P, Ocam, lowerDepth,
                             upperDepth, minDegree, reinterpret_cast<float*>(r.localMPs.data()), r.goodPrl.data(),
                             &r.nGoodPrl, &r.nTrackedOld),
          "se2gpu_triangulate");
    return r;
}

// Device workspace of the tracking thread; removeOutliers has the reference's signature and semantics
// (Track.cpp:134: nMatched = removeOutliers(mRefFrame.keyPointsUn, mFrame.keyPointsUn, mMatchIdx)).
class TrackGeometry {
public:
    TrackGeometry() { check(se2gpu_track_create(&h_), "se2gpu_track_create"); }
    ~TrackGeometry() { se2gpu_track_destroy(h_); }
    TrackGeometry(const TrackGeometry&) = delete;
    TrackGeometry& operator=(const TrackGeometry&) = delete;

    int removeOutliers(const std::vector<KeyPoint>& kp1, const std::vector<KeyPoint>& kp2, std::vector<int>& matches) {
        int nInlier = 0;
        check(se2gpu_track_remove_outliers(h_, reinterpret_cast<const se2gpu_keypoint*>(kp1.data()), (int)kp1.size(),
                                           reinterpret_cast<const se2gpu_keypoint*>(kp2.data()), (int)kp2.size(),
                                           matches.data(), &nInlier),
              "se2gpu_track_remove_outliers");
        return nInlier;
    }

    // Track::doTriangulate on this workspace (no allocation per call); arguments as the free function above
    TriangulationResult doTriangulate(const std::vector<KeyPoint>& keyPointsUnRef,
                                      const std::vector<KeyPoint>& keyPointsUnCur, std::vector<int>& matchIdx,
                                      const std::vector<uint8_t>& hasObservation, const float PrjMtrxEye[12],
                                      const float P[12], const float Ocam[3], float lowerDepth, float upperDepth,
                                      int minDegree = 2) {
        TriangulationResult r;
        const int n = (int)keyPointsUnRef.size();
        r.localMPs.assign(n, Point3f{0, 0, 0});
        r.goodPrl.assign(n, 0);
        if (n == 0) return r;
        check(se2gpu_track_triangulate(h_, n, reinterpret_cast<const se2gpu_keypoint*>(keyPointsUnRef.data()),
                                       reinterpret_cast<const se2gpu_keypoint*>(keyPointsUnCur.data()),
                                       (int)keyPointsUnCur.size(), matchIdx.data(),
                                       hasObservation.empty() ? nullptr : hasObservation.data(), PrjMtrxEye, P, Ocam,
                                       lowerDepth, upperDepth, minDegree, reinterpret_cast<float*>(r.localMPs.data()),
                                       r.goodPrl.data(), &r.nGoodPrl, &r.nTrackedOld),
              "se2gpu_track_triangulate");
        return r;
    }

    // cv::findFundamentalMat(pt1, pt2, mask): pt = (x, y) pairs
    int findFundamentalMask(const std::vector<float>& pt1, const std::vector<float>& pt2, std::vector<uint8_t>& mask) {
        const int n = (int)(pt1.size() / 2);
        mask.assign(n, 0);
        int nInlier = 0;
        check(se2gpu_track_fundamental_mask(h_, pt1.data(), pt2.data(), n, mask.data(), &nInlier),
              "se2gpu_track_fundamental_mask");
        return nInlier;
    }

private:
    se2gpu_track* h_ = nullptr;
};

}  // namespace se2lam_amd
