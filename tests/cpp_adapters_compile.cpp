// Compile-and-link check of the C++ adapters (include/se2lam_amd/*.h) against libse2gpu.so.  Without a GPU it only
// checks that construction fails loudly with SE2GPU_ERR_NO_DEVICE; on the GPU box it runs a tiny localBA-shaped
// call sequence through the reference's names.
#include <cmath>
#include <cstdio>
#include <cstring>

#include "se2lam_amd/ORBextractor.h"
#include "se2lam_amd/ORBmatcher.h"
#include "se2lam_amd/optimizer.h"
#include "se2lam_amd/preintegration.h"
#include "se2lam_amd/Track.h"
#include "se2lam_amd/Localizer.h"

using namespace se2lam_amd;

// Track::updateFramePose pre-integration over a fixed odometry sequence; printed for tests/test_capi.py, which
// recomputes it with numpy
static void preintegration_demo() {
    PreSE2 p;
    resetPreSE2(p);
    Se2f last{100.f, -20.f, 0.3f};
    for (int k = 1; k <= 12; ++k) {
        Se2f now{100.f + 35.f * k + 3.f * (k % 3), -20.f + 4.f * k - 2.f * (k % 2), 0.3f + 0.021f * k};
        updatePreSE2(p, se2Minus(now, last), 2.0, 2.0, 0.002);
        last = now;
    }
    Matrix3D info;
    const bool ok = preSE2Information(p, info);
    std::printf("PRESE2 %.17g %.17g %.17g", p.meas[0], p.meas[1], p.meas[2]);
    for (int i = 0; i < 9; ++i) std::printf(" %.17g", p.cov[i]);
    for (int i = 0; i < 9; ++i) std::printf(" %.17g", ok ? info.m[i] : 0.0);
    std::printf("\n");
}

int main() {
    preintegration_demo();
    if (se2gpu_device_count() == 0) {
        try {
            SlamOptimizer opt;
            std::printf("FAIL: construction succeeded without a device\n");
            return 1;
        } catch (const std::exception& e) {
            std::printf("OK (no device): %s\n", e.what());
            return 0;
        }
    }
    // LocalMapper::localBA shape: two poses, one landmark, two observations, one odometry edge
    SlamOptimizer optimizer;
    bool abortBA = false;
    initOptimizer(optimizer, false);
    optimizer.setForceStopFlag(&abortBA);
    MatF Kcam = MatF::eye(3);
    Kcam.at<float>(0, 0) = 400.f; Kcam.at<float>(1, 1) = 400.f; Kcam.at<float>(0, 2) = 320.f; Kcam.at<float>(1, 2) = 240.f;
    CamPara* campr = addCamPara(optimizer, Kcam, 0);
    SE3Quat Tbc;
    const double R[9] = {0, 0, 1, -1, 0, 0, 0, -1, 0};
    std::memcpy(Tbc.R, R, sizeof(R));
    Tbc.t[0] = 100; Tbc.t[2] = 300;
    addVertexSE2(optimizer, SE2{0, 0, 0}, 0, true);
    addVertexSE2(optimizer, SE2{500, 10, 0.02}, 1, false);
    Matrix3D oinfo{{1e-2, 0, 0, 0, 1e-2, 0, 0, 0, 1e3}};
    addEdgeSE2(optimizer, Vector3D(500, 0, 0), 0, 1, oinfo);
    addVertexSBAXYZ(optimizer, Vector3D(4000, 300, 500), 3);
    Matrix2D info = Matrix2D::Identity();
    addEdgeSE2XYZ(optimizer, Vector2D(290.0, 260.0), 0, 3, campr, Tbc, info, 2.4477);
    addEdgeSE2XYZ(optimizer, Vector2D(286.0, 262.0), 1, 3, campr, Tbc, info, 2.4477);
    optimizer.initializeOptimization(0);
    const double chi0 = optimizer.activeRobustChi2();
    const int it = optimizer.optimize(5);
    const double chi1 = optimizer.activeRobustChi2();
    SE2 p1 = estimateVertexSE2(optimizer, 1);
    Vector3D l = estimateVertexSBAXYZ(optimizer, 3);
    std::printf("localBA adapters: %d iterations, chi2 %.6f -> %.6f, pose1 (%.3f %.3f %.5f), lm (%.2f %.2f %.2f)\n", it, chi0,
                chi1, p1.x, p1.y, p1.theta, l.v[0], l.v[1], l.v[2]);
    if (!(chi1 <= chi0)) return 1;
    ORBextractor extractor(1000, 1.2f, 8, ORBextractor::FAST_SCORE, 20);
    std::vector<uint8_t> img(640 * 480);
    for (size_t i = 0; i < img.size(); ++i) img[i] = (uint8_t)(((i % 640) / 16 + (i / 640) / 16) % 2 ? 200 : 40);
    Mat8U im; im.rows = 480; im.cols = 640; im.step = 640; im.data = img.data();
    std::vector<KeyPoint> kps; Mat8U desc;
    extractor(im, Mat8U(), kps, desc);
    std::printf("ORBextractor adapter: %zu keypoints, levels %d, scale %.2f\n", kps.size(), extractor.GetLevels(),
                extractor.GetScaleFactor());
    ORBmatcher matcher(0.9f);
    FrameView f; f.keyPointsUn = kps.data(); f.descriptors = desc.data; f.N = (int)kps.size();
    std::vector<Point2f> prev(kps.size());
    for (size_t i = 0; i < kps.size(); ++i) prev[i] = kps[i].pt;
    std::vector<int> m12;
    const int nm = matcher.MatchByWindow(f, f, prev, 20, m12);
    std::printf("ORBmatcher adapter: %d self-matches of %zu\n", nm, kps.size());
    TrackGeometry track;
    // forward motion with a different depth per point: kp2 = c + (kp - c) * s_i, epipole at c => every match is exact
    std::vector<KeyPoint> kps2 = kps;
    for (size_t i = 0; i < kps2.size(); ++i) {
        const float s = 1.f + 0.05f * (float)(i % 7 + 1) / 7.f;
        kps2[i].pt.x = 320.f + (kps[i].pt.x - 320.f) * s;
        kps2[i].pt.y = 240.f + (kps[i].pt.y - 240.f) * s;
    }
    const int nInlier = track.removeOutliers(kps, kps2, m12);
    std::printf("Track::removeOutliers adapter: %d inliers of %d matches\n", nInlier, nm);
    // Localizer::DoLocalBA shape: 60 map points in front of a planar pose, exact observations, pose started 30 mm off
    LocalizerBA localizer;
    SE3Quat Tcw0;   // camera looking along world x: Tcw = Tbc^-1 for the body at the origin
    const double Rcb[9] = {0, -1, 0, 0, 0, -1, 1, 0, 0};
    std::memcpy(Tcw0.R, Rcb, sizeof(Rcb));
    Tcw0.t[0] = 0; Tcw0.t[1] = 300; Tcw0.t[2] = -100;
    std::vector<Vector3D> mps; std::vector<Vector2D> obs; std::vector<double> w;
    for (int i = 0; i < 60; ++i) {
        const double X = 3000 + 40 * i, Y = -900 + 31 * i, Z = 100 + 7 * (i % 9);
        const double xc = Rcb[0] * X + Rcb[1] * Y + Rcb[2] * Z + Tcw0.t[0], yc = Rcb[3] * X + Rcb[4] * Y + Rcb[5] * Z + Tcw0.t[1];
        const double zc = Rcb[6] * X + Rcb[7] * Y + Rcb[8] * Z + Tcw0.t[2];
        mps.push_back(Vector3D(X, Y, Z));
        obs.push_back(Vector2D(400 * xc / zc + 320, 400 * yc / zc + 240));
        w.push_back(1.0);
    }
    SE3Quat start = Tcw0;
    start.t[0] += 30;
    se2gpu_ba_stats pst;
    const SE3Quat opt = localizer.DoLocalBA(start, Tbc, mps, obs, w, 400, 320, 240, 2.4477, 1e6, 1e6, 1, 30, &pst);
    std::printf("Localizer::DoLocalBA adapter: chi2 %.3f -> %.6f in %d iterations, t = (%.2f %.2f %.2f)\n", pst.chi2_init,
                pst.chi2_final, pst.iterations, opt.t[0], opt.t[1], opt.t[2]);
    const bool pose_ok = std::fabs(opt.t[0] - Tcw0.t[0]) < 5.0 && pst.chi2_final < 1e-3 * pst.chi2_init;   // the weak planar prior keeps ~2 mm
    return kps.empty() || nInlier != nm || !pose_ok ? 1 : 0;
}
