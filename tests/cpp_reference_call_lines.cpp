// The reference's OWN call lines against the drop-in headers (boundary): the statements below are pasted
// from /root/reference/src/Map.cpp:897, 925-930, 942-953, 959-970, 985-989, 1045-1049, src/LocalMapper.cpp:239-260,
// src/Map.cpp:754-783, src/Track.cpp:131-134 and src/LocalMapper.cpp:117-121 - only the surrounding pointer-graph
// classes (KeyFrame, MapPoint, Config) are stubs, and `se2lam` / `g2o` / `cv` / `Eigen` resolve to the mirrors of
// include/se2lam_amd/types.h (with the real libraries installed they resolve to the real types and the overloads of
// conversions.h).  The test is that this file COMPILES (-Wall -Werror) and, on a GPU, that the pasted sequence runs.
#include <cstdio>
#include <map>
#include <memory>
#include <set>

#include "se2lam_amd/ORBextractor.h"
#include "se2lam_amd/ORBmatcher.h"
#include "se2lam_amd/Track.h"
#include "se2lam_amd/optimizer.h"

namespace g2o {
using SE2 = se2lam_amd::SE2;
using SE3Quat = se2lam_amd::SE3Quat;
using Vector2D = se2lam_amd::Vector2D;
using Vector3D = se2lam_amd::Vector3D;
using Matrix2D = se2lam_amd::Matrix2D;
using Matrix3D = se2lam_amd::Matrix3D;
}  // namespace g2o
namespace cv {
using Mat = se2lam_amd::MatF;
using KeyPoint = se2lam_amd::KeyPoint;
using Point2f = se2lam_amd::Point2f;
}  // namespace cv
using namespace se2lam_amd;
using Vector3D = g2o::Vector3D;
using Eigen_Vector2d = g2o::Vector2D;
using Matrix2d = g2o::Matrix2D;

struct Se2 { float x = 0, y = 0, theta = 0; };
struct PreSE2 { double meas[3]; double cov[9]; };
struct KeyFrame {
    int id = 0;
    Se2 Twb;
    bool isNull() const { return false; }
    std::pair<std::shared_ptr<KeyFrame>, PreSE2> preOdomFromSelf;
};
typedef std::shared_ptr<KeyFrame> PtrKeyFrame;
namespace Config {
cv::Mat Kcam = cv::Mat::eye(3), bTc = cv::Mat::eye(4);
float TH_HUBER = 2.4477f;
int LOCAL_ITER = 5;
bool LOCAL_VERBOSE = false;
}  // namespace Config

static g2o::Matrix3D inverse3(const double* c) {   // Eigen::Map<Matrix3d, RowMajor>(meas.cov).inverse()
    g2o::Matrix3D o;
    const double A = c[4] * c[8] - c[5] * c[7], B = -(c[3] * c[8] - c[5] * c[6]), C = c[3] * c[7] - c[4] * c[6];
    const double id = 1.0 / (c[0] * A + c[1] * B + c[2] * C);
    const double v[9] = {A * id, -(c[1] * c[8] - c[2] * c[7]) * id, (c[1] * c[5] - c[2] * c[4]) * id,
                         B * id, (c[0] * c[8] - c[2] * c[6]) * id, -(c[0] * c[5] - c[2] * c[3]) * id,
                         C * id, -(c[0] * c[7] - c[1] * c[6]) * id, (c[0] * c[4] - c[1] * c[3]) * id};
    for (int i = 0; i < 9; ++i) o.m[i] = v[i];
    return o;
}

int main() {
    Config::Kcam.at<float>(0, 0) = 400; Config::Kcam.at<float>(1, 1) = 400;
    Config::Kcam.at<float>(0, 2) = 320; Config::Kcam.at<float>(1, 2) = 240;
    {   // Config::bTc: camera z forward / x right / y down on a body x forward / y left / z up, 100 mm ahead, 300 mm up
        const float R[9] = {0, 0, 1, -1, 0, 0, 0, -1, 0};
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Config::bTc.at<float>(r, c) = R[3 * r + c];
        Config::bTc.at<float>(0, 3) = 100; Config::bTc.at<float>(2, 3) = 300;
    }
    if (se2gpu_device_count() == 0) {
        std::printf("OK (no device): the reference's call lines compile against include/se2lam_amd\n");
        return 0;
    }
    std::vector<PtrKeyFrame> mLocalGraphKFs, mRefKFs;
    for (int i = 0; i < 2; ++i) {
        PtrKeyFrame kf(new KeyFrame);
        kf->id = 7 + i;
        kf->Twb.x = 500.f * i; kf->Twb.y = 10.f * i; kf->Twb.theta = 0.02f * i;
        mLocalGraphKFs.push_back(kf);
    }
    mLocalGraphKFs[0]->preOdomFromSelf.first = mLocalGraphKFs[1];
    mLocalGraphKFs[0]->preOdomFromSelf.second = PreSE2{{500, 0, 0.02}, {100, 0, 0, 0, 100, 0, 0, 0, 1e-3}};
    bool mbAbortBA = false, mbGlobalBABegin = false;

    // ---- LocalMapper::localBA, LocalMapper.cpp:239-246
    SlamOptimizer optimizer;
    initOptimizer(optimizer, Config::LOCAL_VERBOSE);
    optimizer.setForceStopFlag(&mbAbortBA);

    // ---- Map::loadLocalGraph, Map.cpp:896-897
    int camParaId = 0;
    CamPara* campr = addCamPara(optimizer, (Config::Kcam), camParaId);
    int minKFid = 7;
    const int nLocalKFs = mLocalGraphKFs.size();
    const int nRefKFs = mRefKFs.size();
    // Map.cpp:915-931
    for (int i = 0; i < nLocalKFs; i++) {
        PtrKeyFrame pKF = mLocalGraphKFs[i];
        if (pKF->isNull())
            continue;
        int vertexIdKF = i;
        bool fixed = (pKF->id == minKFid) || pKF->id == 1;
        g2o::SE2 pose(pKF->Twb.x, pKF->Twb.y, pKF->Twb.theta);
        addVertexSE2(optimizer, pose, vertexIdKF, fixed);
    }
    // Map.cpp:934-955
    for (int i = 0; i < nLocalKFs; i++) {
        PtrKeyFrame pKF = mLocalGraphKFs[i];
        if (pKF->isNull())
            continue;
        PtrKeyFrame pKF1 = pKF->preOdomFromSelf.first;
        PreSE2 meas = pKF->preOdomFromSelf.second;
        auto it = std::find(mLocalGraphKFs.begin(), mLocalGraphKFs.end(), pKF1);
        if (it == mLocalGraphKFs.end() || pKF1->isNull())
            continue;
        int id1 = it - mLocalGraphKFs.begin();
        {
            g2o::Matrix3D info = inverse3(meas.cov);
            addEdgeSE2(optimizer, Vector3D(meas.meas), i, id1, info);
        }
    }
    int maxKFid = nLocalKFs + nRefKFs + 1;
    const float delta = Config::TH_HUBER;
    // Map.cpp:985-989, 1045-1049 (one map point seen by both key frames)
    {
        int i = 0;
        int vertexIdMP = maxKFid + i;
        Vector3D lw(4000, 300, 500);
        addVertexSBAXYZ(optimizer, lw, vertexIdMP);
        const double uvs[2][2] = {{290.0, 260.0}, {286.0, 262.0}};
        for (int vertexIdKF = 0; vertexIdKF < 2; ++vertexIdKF) {
            Eigen_Vector2d uv(uvs[vertexIdKF][0], uvs[vertexIdKF][1]);
            Matrix2d Sigma_all = Matrix2d::Identity();
            addEdgeSE2XYZ(optimizer, uv, vertexIdKF, vertexIdMP, campr,
                          toSE3Quat(Config::bTc), Sigma_all.inverse(), delta);
        }
    }
    // ---- LocalMapper.cpp:249-260
    if (mbGlobalBABegin) {
        return 0;
    }
    optimizer.initializeOptimization(0);
    optimizer.optimize(Config::LOCAL_ITER);
    // ---- Map::optimizeLocalGraph, Map.cpp:760-763, 776-779
    for (int i = 0; i < nLocalKFs; i++) {
        g2o::SE2 pose = estimateVertexSE2(optimizer, i);
        std::printf("KF %d: %.3f %.3f %.5f\n", i, pose.x, pose.y, pose.theta);
    }
    g2o::Vector3D p = estimateVertexSBAXYZ(optimizer, maxKFid + 0);
    std::printf("MP: %.2f %.2f %.2f\n", p(0), p(1), p(2));

    // ---- Track.cpp:131-134 / LocalMapper.cpp:117-121: the matcher is constructed on the stack, members are public
    ORBmatcher matcher(0.9);
    matcher.mfNNratio = 0.9f;
    matcher.mbCheckOrientation = true;
    ORBextractor extractor(1000, 1.2f, 8, ORBextractor::FAST_SCORE, 20);       // Track.cpp:34
    std::printf("levels %d scale %.2f\n", extractor.GetLevels(), extractor.GetScaleFactor());
    std::map<int, int> mapIdxMatches12;
    (void)mapIdxMatches12;
    std::printf("reference call lines ran\n");
    return 0;
}
