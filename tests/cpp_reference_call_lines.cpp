// The reference's OWN call lines against the drop-in headers (boundary): the statements below are pasted
// from /root/reference/src/Map.cpp:897, 925-930, 942-953, 959-970, 985-989, 1045-1049, src/LocalMapper.cpp:239-260,
// src/Map.cpp:754-783, src/Track.cpp:131-134 and src/LocalMapper.cpp:117-121 - only the surrounding pointer-graph
// classes (KeyFrame, MapPoint, Config) are stubs, and `se2lam` / `g2o` / `cv` / `Eigen` resolve to the mirrors of
// include/se2lam_amd/types.h (with the real libraries installed they resolve to the real types and the overloads of
// conversions.h).  The test is that this file COMPILES (-Wall -Werror) and, on a GPU, that the pasted sequence runs.
#include <cmath>
#include <cstdio>
#include <map>
#include <memory>
#include <set>

#include "se2lam_amd/ORBextractor.h"
#include "se2lam_amd/ORBmatcher.h"
#include "se2lam_amd/Track.h"
#include "se2lam_amd/optimizer.h"
#include "se2lam_amd/sparsifier.h"
#include "se2lam_amd/Map.h"

namespace g2o {
using SE2 = se2lam_amd::SE2;
using SE3Quat = se2lam_amd::SE3Quat;
using Vector2D = se2lam_amd::Vector2D;
using Vector3D = se2lam_amd::Vector3D;
using Matrix2D = se2lam_amd::Matrix2D;
using Matrix3D = se2lam_amd::Matrix3D;
using EdgeSE3 = se2lam_amd::EdgeSE3;
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
using se2lam_amd::PreSE2;   // Frame.h:20-24
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

    // ---- LocalMapper::localBA, LocalMapper.cpp:239-246 (the solver construction lines as they stand in the reference)
    SlamOptimizer optimizer;
    SlamLinearSolver* linearSolver = new SlamLinearSolver();
    SlamBlockSolver* blockSolver = new SlamBlockSolver(linearSolver);
    SlamAlgorithm* solver = new SlamAlgorithm(blockSolver);
    optimizer.setAlgorithm(solver);
    optimizer.setVerbose(Config::LOCAL_VERBOSE);
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
        VertexSE2* v = addVertexSE2(optimizer, pose, vertexIdKF, fixed);   // g2o::VertexSE2* in the reference (optimizer.h:104)
        if (v->id() != vertexIdKF) return 1;
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
    // REJECT_IF_LARGE_LAMBDA, LocalMapper.cpp:285-292
    if (solver->currentLambda() > 100.0) {
        std::printf("-- DEBUG LM: current lambda too large %f , reject optimized result\n", solver->currentLambda());
        return 1;
    }
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
    {   // ORBmatcher.cpp:349-377: the rotation-consistency lines around ComputeThreeMaxima, a public member (ORBmatcher.h:57)
        const int HISTO_LENGTH = ORBmatcher::HISTO_LENGTH;
        std::vector<int> rotHist[HISTO_LENGTH];
        for (int i = 0; i < 40; ++i) rotHist[3].push_back(i);
        for (int i = 0; i < 12; ++i) rotHist[17].push_back(i);
        for (int i = 0; i < 3; ++i) rotHist[29].push_back(i);
        int ind1 = -1;
        int ind2 = -1;
        int ind3 = -1;
        matcher.ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        std::printf("ComputeThreeMaxima: %d %d %d\n", ind1, ind2, ind3);
        if (ind1 != 3 || ind2 != 17 || ind3 != -1) return 1;
    }
    // ---- LocalMapper::removeOutlierChi2 (LocalMapper.cpp:172-214) on Map::loadLocalGraph(optimizer, vpEdgesAll, vnAllIdx)
    //      (Map.cpp:414-566): two key frames on the plane, one map point, the prior and the projection edges
    {
        SlamOptimizer optimizer;
        initOptimizer(optimizer);
        std::vector<std::vector<EdgeProjectXYZ2UV*> > vpEdgesAll;
        std::vector<std::vector<int> > vnAllIdx;
        int camParaId = 0;
        addCamPara(optimizer, Config::Kcam, camParaId);
        cv::Mat Tcw0 = cv::Mat::eye(4), Tcw1 = cv::Mat::eye(4);
        {   // Tcw = Tbc^-1 for the body at the origin / 500 mm further along x
            const float Rcb[9] = {0, -1, 0, 0, 0, -1, 1, 0, 0};
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { Tcw0.at<float>(r, c) = Rcb[3 * r + c]; Tcw1.at<float>(r, c) = Rcb[3 * r + c]; }
            Tcw0.at<float>(1, 3) = 300; Tcw0.at<float>(2, 3) = -100;
            Tcw1.at<float>(1, 3) = 300; Tcw1.at<float>(2, 3) = -600;
        }
        for (int vertexIdKF = 0; vertexIdKF < 2; ++vertexIdKF) {
            bool fixed = vertexIdKF == 0;
            const cv::Mat& pose = vertexIdKF ? Tcw1 : Tcw0;
            addVertexSE3Expmap(optimizer, toSE3Quat(pose), vertexIdKF, fixed);                       // Map.cpp:444
            addPlaneMotionSE3Expmap(optimizer, toSE3Quat(pose), vertexIdKF, Config::bTc);            // Map.cpp:445
        }
        Matrix6d odoInfo;
        for (int i = 0; i < 6; ++i) odoInfo(i, i) = i < 3 ? 1e-2 : 1e3;                             // [trans rot] order, as stored
        SE3Quat Tc1c0;   // T_c1_c0 = Tcw1 * Tcw0^-1: pure translation along the optical axis
        Tc1c0.t[2] = -500;
        addEdgeSE3Expmap(optimizer, Tc1c0, 0, 1, odoInfo);                                           // Map.cpp:467
        int vertexIdMP = 2 + 1;
        addVertexSBAXYZ(optimizer, Vector3D(4000, 300, 500), vertexIdMP);                            // Map.cpp:500
        std::vector<EdgeProjectXYZ2UV*> vpEdges;
        std::vector<int> vnIdx;
        const float delta = Config::TH_HUBER;
        const double uvs[2][2] = {{290.0, 260.0}, {286.0, 262.0}};
        for (int vertexIdKF = 0; vertexIdKF < 2; ++vertexIdKF) {
            Eigen_Vector2d uv(uvs[vertexIdKF][0], uvs[vertexIdKF][1]);
            Matrix2d info = Matrix2d::Identity();
            EdgeProjectXYZ2UV* ei = addEdgeXYZ2UV(optimizer, uv, vertexIdMP, vertexIdKF, camParaId, info, delta);   // Map.cpp:548
            ei->setLevel(0);
            optimizer.addEdge(ei);
            vpEdges.push_back(ei);
            vnIdx.push_back(vertexIdKF);
        }
        vpEdgesAll.push_back(vpEdges);
        vnAllIdx.push_back(vnIdx);
        const float chi2 = 25;                                                                        // LocalMapper.cpp:186
        optimizer.initializeOptimization(0);
        optimizer.optimize(10);
        const int nAllMP = vpEdgesAll.size();
        std::vector<std::vector<int> > vnOutlierIdxAll;
        for (int i = 0; i < nAllMP; i++) {
            std::vector<int> vnOutlierIdx;
            for (int j = 0, jend = vpEdgesAll[i].size(); j < jend; j++) {
                EdgeProjectXYZ2UV* eij = vpEdgesAll[i][j];
                if (eij->level() > 0)
                    continue;
                eij->computeError();
                bool chi2Bad = eij->chi2() > chi2;
                int idKF = vnAllIdx[i][j];
                if (chi2Bad) {
                    eij->setLevel(1);
                    vnOutlierIdx.push_back(idKF);
                }
            }
            vnOutlierIdxAll.push_back(vnOutlierIdx);
        }
        SE3Quat T1 = estimateVertexSE3Expmap(optimizer, 1);
        std::printf("removeOutlierChi2 lines: %zu outliers, KF1 t = %.2f %.2f %.2f\n", vnOutlierIdxAll[0].size(), T1.t[0], T1.t[1], T1.t[2]);
        const double chi2_first = vpEdgesAll[0][1]->chi2();
        // g2o's second initializeOptimization() (re-optimising the level-0 edges after setLevel) is refused, not silently run
        // over all edges
        bool refused = false;
        try { optimizer.initializeOptimization(0); } catch (const std::runtime_error&) { refused = true; }
        if (!refused) return 1;
        // ---- the optimizer reused across clear(): handles start over with the new graph - the chi2 of the rebuilt
        // graph's edge 0 is its own, not a stale entry of the first graph's edge list
        optimizer.clear();
        addCamPara(optimizer, Config::Kcam, camParaId);
        // Localizer::DoLocalBA, Localizer.cpp:235-240: the solver lines once more, on the reused optimizer
        SlamLinearSolver* linearSolver = new SlamLinearSolver();
        SlamBlockSolver* blockSolver = new SlamBlockSolver(linearSolver);
        SlamAlgorithm* solver = new SlamAlgorithm(blockSolver);
        optimizer.setAlgorithm(solver);
        optimizer.setVerbose(false);
        for (int vertexIdKF = 0; vertexIdKF < 2; ++vertexIdKF) {
            const cv::Mat& pose = vertexIdKF ? Tcw1 : Tcw0;
            addVertexSE3Expmap(optimizer, toSE3Quat(pose), vertexIdKF, vertexIdKF == 0);
            addPlaneMotionSE3Expmap(optimizer, toSE3Quat(pose), vertexIdKF, Config::bTc);
        }
        addEdgeSE3Expmap(optimizer, Tc1c0, 0, 1, odoInfo);
        addVertexSBAXYZ(optimizer, Vector3D(4000, 300, 500), vertexIdMP);
        Eigen_Vector2d uvb(286.0 + 40.0, 262.0);   // a gross outlier as the ONLY second observation
        EdgeProjectXYZ2UV* e0 = addEdgeXYZ2UV(optimizer, Eigen_Vector2d(290.0, 260.0), vertexIdMP, 0, camParaId, Matrix2d::Identity(), delta);
        EdgeProjectXYZ2UV* e1 = addEdgeXYZ2UV(optimizer, uvb, vertexIdMP, 1, camParaId, Matrix2d::Identity(), delta);
        if (e0->index != 0 || e1->index != 1) return 1;      // numbering restarted with the graph
        optimizer.initializeOptimization(0);
        optimizer.optimize(10);
        std::printf("reused optimizer: edge chi2 %.4f %.4f (first graph: %.4f)\n", e0->chi2(), e1->chi2(), chi2_first);
        if (!(e0->chi2() >= 0.0) || !(e1->chi2() >= 0.0) || !std::isfinite(e1->chi2())) return 1;
    }
    // ---- GlobalMapper::GlobalBA (GlobalMapper.cpp:340-412, 517-524): VertexSE3 + plane-motion prior + EdgeSE3
    {
        SlamOptimizer optimizer;
        //initOptimizer(optimizer);
        SlamLinearSolver* linearSolver = new SlamLinearSolver();                                     // GlobalMapper.cpp:341-344
        SlamBlockSolver* blockSolver = new SlamBlockSolver(linearSolver);
        SlamAlgorithm* solver = new SlamAlgorithm(blockSolver);
        optimizer.setAlgorithm(solver);
        int SE3OffsetParaId = 0;
        addParaSE3Offset(optimizer, Isometry3D(), SE3OffsetParaId);
        cv::Mat T_w_c0 = cv::Mat::eye(4), T_w_c1 = cv::Mat::eye(4);
        {
            const float Rbc[9] = {0, 0, 1, -1, 0, 0, 0, -1, 0};
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { T_w_c0.at<float>(r, c) = Rbc[3 * r + c]; T_w_c1.at<float>(r, c) = Rbc[3 * r + c]; }
            T_w_c0.at<float>(0, 3) = 100; T_w_c0.at<float>(2, 3) = 300;
            T_w_c1.at<float>(0, 3) = 630; T_w_c1.at<float>(1, 3) = 8; T_w_c1.at<float>(2, 3) = 304;   // drifted by 30 / 8 / 4 mm
        }
        std::vector<EdgeSE3*> vpEdgePlane;
        addVertexSE3PlaneMotion(optimizer, toSE3Quat(T_w_c0), 0, Config::bTc, SE3OffsetParaId, true);
        addVertexSE3PlaneMotion(optimizer, toSE3Quat(T_w_c1), 1, Config::bTc, SE3OffsetParaId, false);
        Matrix6d info;
        for (int i = 0; i < 6; ++i) info(i, i) = i < 3 ? 1e-1 : 1e4;
        Isometry3D meas;
        meas.t[2] = 500;                                                   // T_c0_c1 of the true motion: 500 mm along the optical axis
        EdgeSE3* pEdgeOdoTmp = addEdgeSE3(optimizer, meas, 0, 1, info);    // GlobalMapper.cpp:386
        optimizer.initializeOptimization();
        optimizer.optimize(5);
        Isometry3D Twc = estimateVertexSE3(optimizer, 1);
        std::printf("GlobalBA lines: KF1 at %.2f %.2f %.2f, odometry edge chi2 %.4f\n", Twc.t[0], Twc.t[1], Twc.t[2], pEdgeOdoTmp->chi2());
        if (!(std::fabs(Twc.t[0] - 600.0) < 15.0)) return 1;
    }
    // ---- GlobalMapper::GlobalBA with PRE_REJECT_FTR_OUTLIER (GlobalMapper.cpp:421-483): the feature edges whose chi2 exceeds the
    // threshold go to level 1 and the SAME optimizer is initialised and optimised again over the others.
    // Four key frames on a line 500 mm apart, odometry edges between neighbours, consistent feature edges 0-1, 1-2, 2-3, 0-2, 1-3
    // and a feature edge 0-3 that claims 1.8 m instead of 1.5 m: the loop must reject exactly that one.
    {
        SlamOptimizer optimizer;
        SlamLinearSolver* linearSolver = new SlamLinearSolver();
        SlamBlockSolver* blockSolver = new SlamBlockSolver(linearSolver);
        SlamAlgorithm* solver = new SlamAlgorithm(blockSolver);
        optimizer.setAlgorithm(solver);
        int SE3OffsetParaId = 0;
        addParaSE3Offset(optimizer, Isometry3D(), SE3OffsetParaId);
        const float Rbc[9] = {0, 0, 1, -1, 0, 0, 0, -1, 0};
        for (int k = 0; k < 4; ++k) {
            cv::Mat T = cv::Mat::eye(4);
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) T.at<float>(r, c) = Rbc[3 * r + c];
            T.at<float>(0, 3) = 100 + 500.0f * k + (k ? 12.0f : 0.0f); T.at<float>(1, 3) = k ? -5.0f : 0.0f; T.at<float>(2, 3) = 300;
            addVertexSE3PlaneMotion(optimizer, toSE3Quat(T), k, Config::bTc, SE3OffsetParaId, k == 0);
        }
        Matrix6d info;
        for (int i = 0; i < 6; ++i) info(i, i) = i < 3 ? 1e-3 : 1e4;      // sigma = 32 mm: nine consistent edges outvote the wrong one
        auto along = [](double mm) { Isometry3D m; m.t[2] = mm; return m; };
        std::vector<g2o::EdgeSE3*> vpEdgeOdo, vpEdgeFeat;
        for (int k = 0; k < 3; ++k) vpEdgeOdo.push_back(addEdgeSE3(optimizer, along(500), k, k + 1, info));
        for (int k = 0; k < 3; ++k) vpEdgeFeat.push_back(addEdgeSE3(optimizer, along(500), k, k + 1, info));
        vpEdgeFeat.push_back(addEdgeSE3(optimizer, along(1000), 0, 2, info));
        vpEdgeFeat.push_back(addEdgeSE3(optimizer, along(1000), 1, 3, info));
        vpEdgeFeat.push_back(addEdgeSE3(optimizer, along(1800), 0, 3, info));
        const double threshFeatEdgeChi2 = 30.0;
        int rounds = 0;
        while (true) {
            optimizer.initializeOptimization();                            // GlobalMapper.cpp:448-450
            optimizer.optimize(15);
            ++rounds;
            // Reject outliers in feature edges                               GlobalMapper.cpp:459-483
            bool bFindOutlier = false;
            std::vector<g2o::EdgeSE3*> vpEdgeFeatGood;
            for (auto iter = vpEdgeFeat.begin(); iter != vpEdgeFeat.end(); iter++) {
                g2o::EdgeSE3* pEdge = *iter;
                double chi2 = pEdge->chi2();
                if (chi2 > threshFeatEdgeChi2) {
                    pEdge->setLevel(1);
                    bFindOutlier = true;
                }
                else {
                    vpEdgeFeatGood.push_back(pEdge);
                }
            }
            vpEdgeFeat.swap(vpEdgeFeatGood);

            if (!bFindOutlier) {
                break;
            }
            if (rounds > 4) return 1;
        }
        Isometry3D T3 = estimateVertexSE3(optimizer, 3);
        std::printf("GlobalBA outlier loop: %d rounds, %zu feature edges kept, KF3 at %.2f %.2f %.2f\n", rounds, vpEdgeFeat.size(), T3.t[0], T3.t[1], T3.t[2]);
        if (rounds != 2 || vpEdgeFeat.size() != 5 || !(std::fabs(T3.t[0] - 1600.0) < 3.0)) return 1;
    }
    // ---- GlobalMapper::CreateFeatEdge (GlobalMapper.cpp:795-837): the measurement vector and the Sparsifier call
    {
        std::vector<SE3Quat> vSe3KFs(2);
        vSe3KFs[1].t[0] = 400;
        std::vector<Vector3D> vPt3MPs;
        std::vector<MeasSE3XYZ> vMeasSE3XYZ;
        int count = 0;
        for (int k = 0; k < 12; ++k) {
            Vector3D Pw(-800.0 + 170.0 * k, 300.0 * ((k % 3) - 1), 3000.0 + 211.0 * ((k * 5) % 7));
            vPt3MPs.push_back(Pw);
            MeasSE3XYZ Meas1;
            Meas1.idKF = 0;
            Meas1.idMP = count;
            Meas1.z = Pw;
            for (int i = 0; i < 3; ++i) Meas1.info(i, i) = i < 2 ? 1e-2 : 1e-4;
            MeasSE3XYZ Meas2;
            Meas2.idKF = 1;
            Meas2.idMP = count;
            Meas2.z = Vector3D(Pw(0) - 400, Pw(1), Pw(2));
            Meas2.info = Meas1.info;
            vMeasSE3XYZ.push_back(Meas1);
            vMeasSE3XYZ.push_back(Meas2);
            count++;
        }
        SE3Quat meas_out;
        Matrix6d info_out;
        Sparsifier::DoMarginalizeSE3XYZ(vSe3KFs, vPt3MPs, vMeasSE3XYZ, meas_out, info_out);
        std::printf("CreateFeatEdge lines: z.t = %.1f %.1f %.1f, info(0,0) = %.4g\n", meas_out.t[0], meas_out.t[1], meas_out.t[2], info_out(0, 0));
        if (!(std::fabs(meas_out.t[0] - 400.0) < 1e-6) || !(info_out(0, 0) > 0)) return 1;
    }
    // ---- Map::updateLocalGraph + Map::loadLocalGraph(SlamOptimizer&) + LocalMapper::localBA through the flat views
    //      (Map.cpp:285-331, 891-1022; LocalMapper.cpp:232-262): three key frames in a row, nine map points
    {
        MapView map;
        for (int k = 0; k < 3; ++k) map.addKeyFrame(k);
        map.addCovisibility(0, 1); map.addCovisibility(1, 0); map.addCovisibility(1, 2); map.addCovisibility(2, 1);
        for (int m = 0; m < 9; ++m) {
            map.addMapPoint(m);
            for (int k = 0; k < 3; ++k) map.addObservation(k, m);
        }
        std::vector<int32_t> localKFs, refKFs, localMPs;
        map.updateLocalGraph(2, localKFs, refKFs, localMPs);
        if (localKFs.size() != 3 || !refKFs.empty() || localMPs.size() != 9) return 1;

        LocalGraph graph(Config::Kcam, Config::bTc, Config::TH_HUBER);
        const float fx = Config::Kcam.at<float>(0, 0), cx = Config::Kcam.at<float>(0, 2), cy = Config::Kcam.at<float>(1, 2);
        for (int k = 0; k < 3; ++k) {
            cv::Mat Tcw = cv::Mat::eye(4);
            const float Rcb[9] = {0, -1, 0, 0, 0, -1, 1, 0, 0};
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Tcw.at<float>(r, c) = Rcb[3 * r + c];
            graph.addLocalKF(k + 1, Se2f{500.f * k + (k == 2 ? 20.f : 0.f), 0.f, 0.f}, Tcw);   // the last one starts 20 mm off
        }
        for (int k = 0; k < 2; ++k) {
            PreSE2 pre;
            resetPreSE2(pre);
            pre.meas[0] = 500;
            pre.cov[0] = pre.cov[4] = 100; pre.cov[8] = 1e-4;
            graph.setOdometry(k, k + 1, pre);
        }
        for (int m = 0; m < 9; ++m) {
            const float X = 4000.f + 300.f * (m % 3), Y = -600.f + 600.f * (m / 3), Z = 200.f + 100.f * (m % 4);
            graph.addMapPoint(X, Y, Z);
            for (int k = 0; k < 3; ++k) {
                // camera at the body origin looking along +x (bTc rotation of the test Config, zero lever arm is not assumed:
                // the measurement is generated with the same Config::bTc the graph evaluates)
                const float bx = X - 500.f * k, by = Y, bz = Z;
                const float tx = Config::bTc.at<float>(0, 3), ty = Config::bTc.at<float>(1, 3), tz = Config::bTc.at<float>(2, 3);
                float pc[3];
                for (int r = 0; r < 3; ++r)
                    pc[r] = Config::bTc.at<float>(0, r) * (bx - tx) + Config::bTc.at<float>(1, r) * (by - ty) + Config::bTc.at<float>(2, r) * (bz - tz);
                graph.addObservation(k, fx * pc[0] / pc[2] + cx, fx * pc[1] / pc[2] + cy, pc[0], pc[1], pc[2], 1.f);
            }
        }
        SlamOptimizer optimizer;
        initOptimizer(optimizer);
        graph.load(optimizer);
        optimizer.initializeOptimization(0);
        optimizer.optimize(Config::LOCAL_ITER);
        SE2 last = estimateVertexSE2(optimizer, 2);
        Vector3D mp0 = estimateVertexSBAXYZ(optimizer, graph.vertexIdMP(0));
        std::printf("loadLocalGraph lines: KF3 x = %.3f (started at 1020, truth 1000), MP0 x = %.2f\n", last.x, mp0(0));
        // forward motion towards points 4 m ahead barely constrains x and g2o's lambda_0 = 1e-5 * max diag(H) is far
        // above the odometry stiffness, so Config::LOCAL_ITER damped steps only start the walk back to 1000
        if (!(last.x < 1019.5 && last.x > 999.0)) return 1;
    }
    std::printf("reference call lines ran\n");
    return 0;
}
