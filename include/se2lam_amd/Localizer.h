// Pose-only bundle adjustment of se2lam::Localizer on the device - header-only mirror over the C ABI (include/se2gpu.h).
//   Localizer::DoLocalBA           /root/reference/src/Localizer.cpp:233-302
//   addPlaneMotionSE3Expmap        /root/reference/src/optimizer.cpp:236-314   (+ EdgeSE3ExpmapPrior :159-197)
// The reference builds a g2o graph on the heap for every call (one VertexSE3Expmap, the observed map points as fixed
// vertices, one EdgeProjectXYZ2UV each, the plane-motion prior) and runs optimize(30); here the observations go down
// as three flat arrays and the whole optimisation is one kernel launch.  Collecting the observations of mpKFCurr
// (Localizer.cpp:257-281: good-parallax map points, keyPointsUn[ftrIdx].pt, mvInvLevelSigma2[octave]) and writing the
// pose back (:295-296) stay with the caller.
#pragma once
#include <cstring>
#include <vector>

#include "../se2gpu.h"
#include "types.h"

namespace se2lam_amd {

struct PlaneMotionPrior {
    SE3Quat measurement;     // EdgeSE3ExpmapPrior::setMeasurement
    double information[36];  // row-major 6x6, vector order (rotation, translation)
};

// addPlaneMotionSE3Expmap(opt, pose, vId, extPara): pose = Tcw of the vertex, extPara = Config::bTc;
// infoXrot / infoYrot / infoZ = Config::PLANEMOTION_XROT_INFO / YROT_INFO / Z_INFO
inline PlaneMotionPrior planeMotionSE3Expmap(const SE3Quat& pose, const SE3Quat& bTc, double infoXrot, double infoYrot,
                                             double infoZ) {
    double a[12], b[12], m[12];
    std::memcpy(a, pose.R, sizeof(pose.R)); std::memcpy(a + 9, pose.t, sizeof(pose.t));
    std::memcpy(b, bTc.R, sizeof(bTc.R)); std::memcpy(b + 9, bTc.t, sizeof(bTc.t));
    PlaneMotionPrior p;
    check(se2gpu_plane_motion_prior(a, b, infoXrot, infoYrot, infoZ, m, p.information), "se2gpu_plane_motion_prior");
    std::memcpy(p.measurement.R, m, sizeof(p.measurement.R));
    std::memcpy(p.measurement.t, m + 9, sizeof(p.measurement.t));
    return p;
}

class LocalizerBA {
public:
    LocalizerBA() { check(se2gpu_track_create(&h_), "se2gpu_track_create"); }
    ~LocalizerBA() { se2gpu_track_destroy(h_); }
    LocalizerBA(const LocalizerBA&) = delete;
    LocalizerBA& operator=(const LocalizerBA&) = delete;

    // Localizer::DoLocalBA: Tcw = toSE3Quat(mpKFCurr->getPose()); mapPoints[i] = toVector3d(pMP->getPos()),
    // uv[i] = keyPointsUn[ftrIdx].pt, invSigma2[i] = mvInvLevelSigma2[octave] of the i-th good-parallax observation;
    // f, cx, cy from Config::Kcam (addCamPara uses K(0,0) for both axes); thHuber = Config::TH_HUBER.
    // Returns estimateVertexSE3Expmap after optimize(iterations).
    SE3Quat DoLocalBA(const SE3Quat& Tcw, const SE3Quat& bTc, const std::vector<Vector3D>& mapPoints,
                      const std::vector<Vector2D>& uv, const std::vector<double>& invSigma2, double f, double cx, double cy,
                      double thHuber, double infoXrot, double infoYrot, double infoZ, int iterations = 30,
                      se2gpu_ba_stats* stats = nullptr) {
        const PlaneMotionPrior prior = planeMotionSE3Expmap(Tcw, bTc, infoXrot, infoYrot, infoZ);
        double a[12], m[12], out[12];
        std::memcpy(a, Tcw.R, sizeof(Tcw.R)); std::memcpy(a + 9, Tcw.t, sizeof(Tcw.t));
        std::memcpy(m, prior.measurement.R, sizeof(prior.measurement.R));
        std::memcpy(m + 9, prior.measurement.t, sizeof(prior.measurement.t));
        const int n = (int)mapPoints.size();
        if (uv.size() != mapPoints.size() || invSigma2.size() != mapPoints.size())
            throw std::runtime_error("DoLocalBA: observation arrays differ in length");
        check(se2gpu_track_pose_ba(h_, a, m, prior.information, n, n ? mapPoints[0].v : nullptr, n ? uv[0].v : nullptr,
                                   invSigma2.data(), f, cx, cy, thHuber, iterations, out, stats),
              "se2gpu_track_pose_ba");
        SE3Quat r;
        std::memcpy(r.R, out, sizeof(r.R));
        std::memcpy(r.t, out + 9, sizeof(r.t));
        return r;
    }

private:
    se2gpu_track* h_ = nullptr;
};

}  // namespace se2lam_amd
