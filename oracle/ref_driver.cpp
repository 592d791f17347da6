// ORACLE - TEST INFRASTRUCTURE ONLY (oracle/_ref).  Never linked, imported or called by the product path.
//
// C entry points over the REFERENCE's own classes, compiled from /root/reference where the sources lie (oracle/Makefile,
// target `ref`; nothing of the reference is copied into this repository):
//   se2lam::ORBextractor::operator()      src/ORBextractor.cpp:727-788 (with ComputePyramid, ComputeKeyPoints, IC_Angle,
//                                         computeOrbDescriptor, HarrisResponses and the constructor's tables)
//   se2lam::Frame::Frame / PosInGrid / GetFeaturesInArea      src/Frame.cpp:19-83, 209-286
//   se2lam::ORBmatcher::MatchByWindow / MatchByProjection / SearchByBoW / ComputeThreeMaxima / DescriptorDistance
//                                         src/ORBmatcher.cpp:64-454
//   cvu::camprjc / se3map / triangulate / checkParallax       src/cvutil.cpp
//   g2o::EdgeSE2XYZ::computeError / linearizeOplus, SE2ToSE3, SE3ToSE2, d_inv_d_se2      src/EdgeSE2XYZ.cpp:16-106
//   g2o::PreEdgeSE2::computeError / linearizeOplus             include/se2lam/EdgeSE2XYZ.h:62-102
//   se2lam::addCamPara / addVertexSE2 / addVertexSBAXYZ / addEdgeSE2 / addEdgeSE2XYZ / addVertexSE3Expmap / addEdgeSE3Expmap /
//   addPlaneMotionSE3Expmap / addVertexSE3PlaneMotion / EdgeSE3ExpmapPrior / Jl / invJl / invJJl / verifyInfo and
//   toSE3Quat / toIsometry3D / toCvMat                         src/optimizer.cpp, src/converter.cpp (whole files)
//   Sparsifier::DoMarginalizeSE3XYZ / HessianSE3XYZ / JacobianSE3XYZ / InfoSE3 / JacobianSE3      src/sparsifier.cpp (whole file)
// against oracle/_shim (a stand-in for the OpenCV / ROS headers, and stubs of KeyFrame / MapPoint with the members the
// matcher reads).  What this library pins is the se2lam-owned logic; the OpenCV arithmetic underneath is the shim's.
// The signatures mirror oracle/orb_ref.cpp and oracle/match_ref.cpp so that tests call either through the same wrapper.
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <vector>

#include "EdgeSE2XYZ.h"
#include "ORBextractor.h"
#include "ORBmatcher.h"
#include "converter.h"
#include "cvutil.h"
#include "optimizer.h"
#include "sparsifier.h"

using namespace se2lam;

extern "C" {
struct ref_orb_params {
    int32_t nfeatures;
    float scale_factor;
    int32_t nlevels, fast_th, score_type;
};
struct ref_keypoint {  // cv::KeyPoint layout
    float x, y, size, angle, response;
    int32_t octave, class_id;
};
struct ref_bounds { float min_x, min_y, max_x, max_y; };
}

namespace {

std::vector<cv::KeyPoint> to_cv(const ref_keypoint* k, int n) {
    std::vector<cv::KeyPoint> v(n);
    for (int i = 0; i < n; ++i) v[i] = cv::KeyPoint(k[i].x, k[i].y, k[i].size, k[i].angle, k[i].response, k[i].octave, k[i].class_id);
    return v;
}
void from_cv(const std::vector<cv::KeyPoint>& v, ref_keypoint* k) {
    for (size_t i = 0; i < v.size(); ++i)
        k[i] = ref_keypoint{v[i].pt.x, v[i].pt.y, v[i].size, v[i].angle, v[i].response, v[i].octave, v[i].class_id};
}

// A Frame with given key points: what Frame::Frame (src/Frame.cpp:19-83) leaves behind, without the image.  The statics are
// the ones its first call computes (computeBoundUn with D = 0: the image rectangle; :38-44), the grid loop is :64-76 with the
// reference's own PosInGrid.
void set_bounds(const ref_bounds& b) {
    Frame::minXUn = b.min_x; Frame::minYUn = b.min_y; Frame::maxXUn = b.max_x; Frame::maxYUn = b.max_y;
    Frame::mfGridElementWidthInv = static_cast<float>(FRAME_GRID_COLS) / (Frame::maxXUn - Frame::minXUn);
    Frame::mfGridElementHeightInv = static_cast<float>(FRAME_GRID_ROWS) / (Frame::maxYUn - Frame::minYUn);
    Frame::mbInitialComputations = false;
}
void fill_frame(Frame& f, const ref_keypoint* kps, const uint8_t* desc, int n) {
    f.keyPoints = to_cv(kps, n);
    f.keyPointsUn = f.keyPoints;
    f.N = n;
    f.descriptors = cv::Mat(n, 32, CV_8UC1);
    for (int i = 0; i < n; ++i) std::memcpy(f.descriptors.ptr(i), desc + 32 * (size_t)i, 32);
    for (size_t i = 0; i < f.keyPointsUn.size(); i++) {
        cv::KeyPoint& kp = f.keyPointsUn[i];
        int gx, gy;
        if (f.PosInGrid(kp, gx, gy)) f.mGrid[gx][gy].push_back(i);
    }
}

DBoW2::FeatureVector to_fv(const int32_t* nodes, const int32_t* ptr, const int32_t* idx, int nn) {
    DBoW2::FeatureVector fv;
    for (int k = 0; k < nn; ++k)
        for (int t = ptr[k]; t < ptr[k + 1]; ++t) fv.addFeature((DBoW2::NodeId)nodes[k], (unsigned)idx[t]);
    return fv;
}

}  // namespace

extern "C" {

int ref_orb_extract(const ref_orb_params* p, const uint8_t* img, int rows, int cols, int step, ref_keypoint* kps, uint8_t* desc,
                    int cap, int* n_out) {
    ORBextractor ex(p->nfeatures, p->scale_factor, p->nlevels, p->score_type, p->fast_th);
    cv::Mat im(rows, cols, CV_8UC1, (void*)img, (size_t)step);
    std::vector<cv::KeyPoint> v;
    cv::Mat d;
    try {
        ex(im, cv::Mat(), v, d);
    } catch (const std::exception& e) {   // where OpenCV would raise cv::Exception (a cell rectangle outside its level, ...)
        *n_out = 0;
        return -2;
    }
    *n_out = (int)v.size();
    if ((int)v.size() > cap) return -1;
    from_cv(v, kps);
    for (size_t i = 0; i < v.size(); ++i) std::memcpy(desc + 32 * i, d.ptr((int)i), 32);
    return 0;
}

int ref_hamming(const uint8_t* a, const uint8_t* b) {
    cv::Mat A(1, 32, CV_8UC1, (void*)a), B(1, 32, CV_8UC1, (void*)b);
    return ORBmatcher::DescriptorDistance(A, B);
}

void ref_three_maxima(const int32_t* counts, int L, int32_t* out3) {
    std::vector<std::vector<int>> histo(L);
    for (int i = 0; i < L; ++i) histo[i].assign(counts[i], 0);
    int i1 = -1, i2 = -1, i3 = -1;
    ORBmatcher m;
    m.ComputeThreeMaxima(histo.data(), L, i1, i2, i3);
    out3[0] = i1; out3[1] = i2; out3[2] = i3;
}

int ref_features_in_area(const ref_bounds* b, const ref_keypoint* kps, int n, float x, float y, float r, int min_level, int max_level,
                         int32_t* out, int cap) {
    set_bounds(*b);
    Frame f;
    std::vector<uint8_t> zero(32 * (size_t)(n > 0 ? n : 1), 0);
    fill_frame(f, kps, zero.data(), n);
    const std::vector<size_t> v = f.GetFeaturesInArea(x, y, r, min_level, max_level);
    if ((int)v.size() > cap) return -1;
    for (size_t i = 0; i < v.size(); ++i) out[i] = (int32_t)v[i];
    return (int)v.size();
}

int ref_match_window(const ref_bounds* b, const ref_keypoint* kps1, const uint8_t* desc1, int n1, const ref_keypoint* kps2,
                     const uint8_t* desc2, int n2, float* prev_xy, int win, int level_offset, int min_level, int max_level,
                     float nnratio, int32_t* m12) {
    set_bounds(*b);
    Frame f1, f2;
    fill_frame(f1, kps1, desc1, n1);
    fill_frame(f2, kps2, desc2, n2);
    std::vector<cv::Point2f> prev(n1);
    for (int i = 0; i < n1; ++i) prev[i] = cv::Point2f(prev_xy[2 * i], prev_xy[2 * i + 1]);
    std::vector<int> matches;
    ORBmatcher matcher(nnratio);
    const int nm = matcher.MatchByWindow(f1, f2, prev, win, matches, level_offset, min_level, max_level);
    for (int i = 0; i < n1; ++i) { m12[i] = matches[i]; prev_xy[2 * i] = prev[i].x; prev_xy[2 * i + 1] = prev[i].y; }
    return nm;
}

// mp_skip bits: 1 = isNull, 2 = bad parallax, 4 = already observed by the key frame (ORBmatcher.cpp:392-395)
int ref_match_projection(const ref_bounds* b, const float* mp_pos, const uint8_t* mp_desc, const int32_t* mp_octave,
                         const uint8_t* mp_skip, int m, const float* Tcw12, float fx, float fy, float cx, float cy,
                         const ref_keypoint* kps, const uint8_t* desc, const uint8_t* kf_observed, int n, int win, int level_offset,
                         float nnratio, int32_t* out) {
    set_bounds(*b);
    Config::Kcam = (cv::Mat_<float>(3, 3) << fx, 0, cx, 0, fy, cy, 0, 0, 1);
    PtrKeyFrame kf = std::make_shared<KeyFrame>();
    fill_frame(*kf, kps, desc, n);
    kf->Tcw = cv::Mat::eye(4, 4, CV_32FC1);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) kf->Tcw.at<float>(r, c) = Tcw12[4 * r + c];
    PtrMapPoint dummy = std::make_shared<MapPoint>();
    for (int i = 0; i < n; ++i) if (kf_observed[i]) kf->mDualObservations[i] = dummy;
    std::vector<PtrMapPoint> mps(m);
    std::vector<cv::Mat> keep(m);
    for (int i = 0; i < m; ++i) {
        mps[i] = std::make_shared<MapPoint>();
        mps[i]->mPos = cv::Point3f(mp_pos[3 * i], mp_pos[3 * i + 1], mp_pos[3 * i + 2]);
        mps[i]->mMainOctave = mp_octave[i];
        mps[i]->mMainDescriptor = cv::Mat(1, 32, CV_8UC1);
        std::memcpy(mps[i]->mMainDescriptor.ptr(0), mp_desc + 32 * (size_t)i, 32);
        // the oracle's mp_skip is one flag for "the reference continues at :392-395": spread it over the three tests in turn
        const int s = mp_skip[i] ? 1 + (i % 3) : 0;
        mps[i]->mbNull = s == 1;
        mps[i]->mbGoodParallax = s != 2;
        if (s == 3) kf->mObservations[mps[i]] = -1;
    }
    std::vector<int> idx;
    ORBmatcher matcher(nnratio);
    const int nm = matcher.MatchByProjection(kf, mps, win, level_offset, idx);
    for (int i = 0; i < n; ++i) out[i] = idx[i];
    return nm;
}

int ref_search_by_bow(const ref_keypoint* kps1, const uint8_t* desc1, int n1, const int32_t* nodes1, const int32_t* ptr1,
                      const int32_t* idx1, int nn1, const uint8_t* has_mp1, const ref_keypoint* kps2, const uint8_t* desc2, int n2,
                      const int32_t* nodes2, const int32_t* ptr2, const int32_t* idx2, int nn2, const uint8_t* has_mp2, int mp_only,
                      float nnratio, int check_ori, int32_t* m12) {
    ref_bounds b{0.f, 0.f, 640.f, 480.f};
    set_bounds(b);
    PtrKeyFrame k1 = std::make_shared<KeyFrame>(), k2 = std::make_shared<KeyFrame>();
    fill_frame(*k1, kps1, desc1, n1);
    fill_frame(*k2, kps2, desc2, n2);
    k1->mFeatVec = to_fv(nodes1, ptr1, idx1, nn1);
    k2->mFeatVec = to_fv(nodes2, ptr2, idx2, nn2);
    for (int i = 0; i < n1; ++i) if (has_mp1[i]) k1->mDualObservations[i] = std::make_shared<MapPoint>();
    for (int i = 0; i < n2; ++i) if (has_mp2[i]) k2->mDualObservations[i] = std::make_shared<MapPoint>();
    std::map<int, int> matches;
    ORBmatcher matcher(nnratio, check_ori != 0);
    const int nm = matcher.SearchByBoW(k1, k2, matches, mp_only != 0);
    for (int i = 0; i < n1; ++i) m12[i] = -1;
    for (auto& kv : matches) m12[kv.first] = kv.second;
    return nm;
}

// Track's per-frame front end exactly as the reference runs it: Frame::Frame (undistort with D = 0, ORBextractor, grid) on both
// images, then MatchByWindow with vbPrevMatched = the first frame's key points (Track::resetLocalTrack, src/Track.cpp:194)
int ref_track_two_frames(const ref_orb_params* p, const uint8_t* img1, const uint8_t* img2, int rows, int cols, float fx, float fy,
                         float cx, float cy, int win, float nnratio, ref_keypoint* kps1, uint8_t* desc1, int* n1, ref_keypoint* kps2,
                         uint8_t* desc2, int* n2, int cap, int32_t* m12, float* prev_xy) {
    Config::Kcam = (cv::Mat_<float>(3, 3) << fx, 0, cx, 0, fy, cy, 0, 0, 1);
    Config::Dcam = cv::Mat::zeros(4, 1, CV_32FC1);
    Frame::mbInitialComputations = true;
    ORBextractor ex(p->nfeatures, p->scale_factor, p->nlevels, p->score_type, p->fast_th);
    cv::Mat im1(rows, cols, CV_8UC1, (void*)img1), im2(rows, cols, CV_8UC1, (void*)img2);
    Frame f1(im1, Se2(0, 0, 0), &ex, Config::Kcam, Config::Dcam), f2(im2, Se2(0, 0, 0), &ex, Config::Kcam, Config::Dcam);
    *n1 = f1.N; *n2 = f2.N;
    if (f1.N > cap || f2.N > cap) return -1;
    from_cv(f1.keyPoints, kps1);
    from_cv(f2.keyPoints, kps2);
    for (int i = 0; i < f1.N; ++i) std::memcpy(desc1 + 32 * (size_t)i, f1.descriptors.ptr(i), 32);
    for (int i = 0; i < f2.N; ++i) std::memcpy(desc2 + 32 * (size_t)i, f2.descriptors.ptr(i), 32);
    std::vector<cv::Point2f> prev(f1.N);
    for (int i = 0; i < f1.N; ++i) prev[i] = f1.keyPoints[i].pt;
    std::vector<int> matches;
    ORBmatcher matcher(nnratio);
    const int nm = matcher.MatchByWindow(f1, f2, prev, win, matches);
    for (int i = 0; i < f1.N; ++i) { m12[i] = matches[i]; prev_xy[2 * i] = prev[i].x; prev_xy[2 * i + 1] = prev[i].y; }
    return nm;
}

// cvu::triangulate + the depth / parallax tests around it (src/cvutil.cpp:46-98); P 3x4 row-major float
void ref_triangulate_point(const float* pt1, const float* pt2, const float* P1, const float* P2, float* out3) {
    cv::Mat A(3, 4, CV_32FC1, (void*)P1), B(3, 4, CV_32FC1, (void*)P2);
    const cv::Point3f x = cvu::triangulate(cv::Point2f(pt1[0], pt1[1]), cv::Point2f(pt2[0], pt2[1]), A, B);
    out3[0] = x.x; out3[1] = x.y; out3[2] = x.z;
}
int ref_check_parallax(const float* o1, const float* o2, const float* p, int min_degree) {
    return cvu::checkParallax(cv::Point3f(o1[0], o1[1], o1[2]), cv::Point3f(o2[0], o2[1], o2[2]), cv::Point3f(p[0], p[1], p[2]), min_degree) ? 1 : 0;
}
void ref_se2_compose(const float* a, const float* b, int minus, float* out3) {   // Se2::operator+ / operator- (src/Config.cpp:200-223)
    const Se2 A(a[0], a[1], a[2]), B(b[0], b[1], b[2]);
    const Se2 r = minus ? (A - B) : (A + B);
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.theta;
}

// One EdgeSE2XYZ through the reference's own computeError() / linearizeOplus() (src/EdgeSE2XYZ.cpp:61-106): camera f, cx, cy;
// extrinsic Tbc = (Rbc row-major, tbc) as optimizer.cpp:199-215 hands it over (a rotation matrix turned into the SE3Quat's
// quaternion); pose (x, y, theta); landmark; measurement.  e[2], Jp[2x3] = d e / d (x, y, theta), Jl[2x3] = d e / d landmark.
void ref_edge_se2xyz(double f, double cx, double cy, const double* Rbc, const double* tbc, const double* pose, const double* lw,
                     const double* uv, double* e, double* Jp, double* Jl) {
    g2o::CameraParameters cam(f, g2o::Vector2D(cx, cy), 1.0);
    Eigen::Matrix3d R;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R(r, c) = Rbc[3 * r + c];
    g2o::SE3Quat Tbc;
    Tbc.setRotation(Eigen::Quaterniond(R));
    Tbc.setTranslation(g2o::Vector3D(tbc[0], tbc[1], tbc[2]));
    g2o::VertexSE2 v1;
    v1.setEstimate(g2o::SE2(pose[0], pose[1], pose[2]));
    g2o::VertexSBAPointXYZ v2;
    v2.setEstimate(g2o::Vector3D(lw[0], lw[1], lw[2]));
    g2o::EdgeSE2XYZ edge;
    edge.setVertex(0, &v1);
    edge.setVertex(1, &v2);
    edge.setMeasurement(g2o::Vector2D(uv[0], uv[1]));
    edge.setCameraParameter(&cam);
    edge.setExtParameter(Tbc);
    edge.computeError();
    edge.linearizeOplus();
    for (int i = 0; i < 2; ++i) e[i] = edge.error()[i];
    for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) { Jp[3 * r + c] = edge.jacobianOplusXi()(r, c); Jl[3 * r + c] = edge.jacobianOplusXj()(r, c); }
}
void ref_edge_pre_se2(const double* pi, const double* pj, const double* z, double* e, double* Ji, double* Jj) {
    g2o::VertexSE2 v1, v2;
    v1.setEstimate(g2o::SE2(pi[0], pi[1], pi[2]));
    v2.setEstimate(g2o::SE2(pj[0], pj[1], pj[2]));
    g2o::PreEdgeSE2 edge;
    edge.setVertex(0, &v1);
    edge.setVertex(1, &v2);
    edge.setMeasurement(g2o::Vector3D(z[0], z[1], z[2]));
    edge.computeError();
    edge.linearizeOplus();
    for (int i = 0; i < 3; ++i) e[i] = edge.error()[i];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { Ji[3 * r + c] = edge.jacobianOplusXi()(r, c); Jj[3 * r + c] = edge.jacobianOplusXj()(r, c); }
}
// SE3ToSE2(SE2ToSE3(x, y, theta)) and d_inv_d_se2 (src/EdgeSE2XYZ.cpp:16-39)
void ref_se2_se3_round_trip(const double* pose, double* out3, double* dinv9) {
    const g2o::SE2 p(pose[0], pose[1], pose[2]);
    const g2o::SE2 q = g2o::SE3ToSE2(g2o::SE2ToSE3(p));
    const g2o::Vector3D v = q.toVector();
    out3[0] = v[0]; out3[1] = v[1]; out3[2] = v[2];
    const Eigen::Matrix3d d = g2o::d_inv_d_se2(p);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) dinv9[3 * r + c] = d(r, c);
}

}  // extern "C"

// ---- src/optimizer.cpp: the graph construction surface, against the recording SparseOptimizer of oracle/_shim/g2o_shim.hpp
namespace {
cv::Mat mat4f(const double* pose12) {   // (R row-major, t) -> the 4x4 CV_32F that Config::bTc / KeyFrame::Tcw are
    cv::Mat T = cv::Mat::eye(4, 4, CV_32FC1);
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) T.at<float>(r, c) = (float)pose12[3 * r + c];
        T.at<float>(r, 3) = (float)pose12[9 + r];
    }
    return T;
}
g2o::SE3Quat quat12(const double* pose12) {
    Eigen::Matrix3d R;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R(r, c) = pose12[3 * r + c];
    return g2o::SE3Quat(R, Eigen::Vector3d(pose12[9], pose12[10], pose12[11]));
}
void put12(const Eigen::Matrix3d& R, const Eigen::Vector3d& t, double* out12) {
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) out12[3 * r + c] = R(r, c); out12[9 + r] = t[r]; }
}
template <typename M> void put36(const M& m, double* out36) { for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) out36[6 * r + c] = m(r, c); }
void set_plane_config(double xrot, double yrot, double z) {
    Config::PLANEMOTION_XROT_INFO = (float)xrot;
    Config::PLANEMOTION_YROT_INFO = (float)yrot;
    Config::PLANEMOTION_Z_INFO = (float)z;
}
}  // namespace

extern "C" {

// addVertexSE3Expmap + addPlaneMotionSE3Expmap (src/optimizer.cpp:226-314) -> measurement (R, t) and information of the
// EdgeSE3ExpmapPrior it attaches; Tbc goes in as the CV_32F matrix Config::bTc is.  Returns the number of edges in the graph.
int ref_plane_motion_expmap(const double* Tcw12, const double* Tbc12, double xrot, double yrot, double z, double* meas12, double* info36) {
    set_plane_config(xrot, yrot, z);
    SlamOptimizer opt;
    initOptimizer(opt);
    const g2o::SE3Quat pose = quat12(Tcw12);
    addVertexSE3Expmap(opt, pose, 0, false);
    EdgeSE3ExpmapPrior* e = addPlaneMotionSE3Expmap(opt, pose, 0, mat4f(Tbc12));
    put12(e->measurement().rotation().toRotationMatrix(), e->measurement().translation(), meas12);
    put36(e->information(), info36);
    return (int)opt.edges().size() * (e->vertices()[0] == opt.vertex(0) ? 1 : -1);
}
// addVertexSE3PlaneMotion (src/optimizer.cpp:336-470) -> measurement and information of the g2o::EdgeSE3Prior, and the
// SE3-offset parameter id the edge was given
int ref_plane_motion_iso3(const double* Twc12, const double* Tbc12, double xrot, double yrot, double z, double* meas12, double* info36) {
    set_plane_config(xrot, yrot, z);
    SlamOptimizer opt;
    initOptimizer(opt);
    addParaSE3Offset(opt, g2o::Isometry3D(), 7);
    g2o::EdgeSE3Prior* e = addVertexSE3PlaneMotion(opt, (g2o::Isometry3D)quat12(Twc12), 3, mat4f(Tbc12), 7, false);
    put12(e->measurement().linear(), e->measurement().translation(), meas12);
    put36(e->information(), info36);
    return (opt.vertex(3) && e->vertices()[0] == opt.vertex(3)) ? e->parameterId(0) : -1;
}
// EdgeSE3ExpmapPrior::computeError / linearizeOplus (src/optimizer.cpp:159-191): err (rotation, translation), J 6x6
void ref_prior_expmap_edge(const double* meas12, const double* est12, double* err6, double* J36) {
    g2o::VertexSE3Expmap v;
    v.setEstimate(quat12(est12));
    EdgeSE3ExpmapPrior e;
    e.vertices()[0] = &v;
    e.setMeasurement(quat12(meas12));
    e.computeError();
    e.linearizeOplus();
    for (int i = 0; i < 6; ++i) err6[i] = e.error()[i];
    put36(e.jacobianOplusXi(), J36);
}
// addEdgeSE3Expmap (src/optimizer.cpp:482-500): the (translation, rotation) -> (rotation, translation) reordering of the information
int ref_edge_se3expmap_info(const double* info36_tr, double* info36_rt) {
    SlamOptimizer opt;
    addVertexSE3Expmap(opt, g2o::SE3Quat(), 0, true);
    addVertexSE3Expmap(opt, g2o::SE3Quat(), 1, false);
    g2o::Matrix6d info;
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) info(r, c) = info36_tr[6 * r + c];
    if (!verifyInfo(info)) return -1;
    addEdgeSE3Expmap(opt, g2o::SE3Quat(), 0, 1, info);
    const g2o::EdgeSE3Expmap* e = static_cast<const g2o::EdgeSE3Expmap*>(opt.edges().at(0));
    put36(e->information(), info36_rt);
    return 0;
}
// Jl / invJl (src/optimizer.cpp:64-93) and invJJl (:109-157)
void ref_so3_jacobians(const double* v3, double* Jl9, double* invJl9) {
    const g2o::Vector3D v(v3[0], v3[1], v3[2]);
    const g2o::Matrix3D a = Jl(v), b = invJl(v);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { Jl9[3 * r + c] = a(r, c); invJl9[3 * r + c] = b(r, c); }
}
void ref_inv_jjl(const double* v6, double* out36) {
    g2o::Vector6d v;
    for (int i = 0; i < 6; ++i) v[i] = v6[i];
    put36(invJJl(v), out36);
}
// converter.cpp: toSE3Quat(cv::Mat) -> toCvMat(SE3Quat) and toIsometry3D(cv::Mat) -> toCvMat(Isometry3D) round trips of a CV_32F pose
void ref_converter_round_trip(const double* pose12, float* via_quat16, float* via_iso16) {
    const cv::Mat T = mat4f(pose12);
    const cv::Mat a = toCvMat(toSE3Quat(T)), b = toCvMat(toIsometry3D(T));
    for (int i = 0; i < 16; ++i) { via_quat16[i] = a.at<float>(i / 4, i % 4); via_iso16[i] = b.at<float>(i / 4, i % 4); }
}

// One SE(2)-XYZ window built with the reference's own calls, in the order of Map::loadLocalGraph's consumer
// (src/LocalMapper.cpp:355-424 / src/GlobalMapper.cpp:328-388): addCamPara(K), addVertexSE2 per key frame, addEdgeSE2 per
// odometry pair, addVertexSBAXYZ per landmark (ids after the key frames), addEdgeSE2XYZ per observation with Tbc =
// toSE3Quat(Config::bTc) and Huber delta.  Every edge then runs its own computeError(); chi_e / chi_o = e' Omega e per edge,
// returns sum rho(chi) as g2o's activeRobustChi2 does.  K and bTc go in as CV_32F like Config::Kcam / Config::bTc.
double ref_window_chi2(int P, const double* poses, const uint8_t* fixed, int L, const double* lms, int E, const int32_t* e_kf,
                       const int32_t* e_lm, const double* e_uv, const double* e_info3, int O, const int32_t* o_i, const int32_t* o_j,
                       const double* o_meas, const double* o_info9, const float* K9, const double* Tbc12, double huber,
                       double* chi_e, double* chi_o, int32_t* counts3) {
    SlamOptimizer opt;
    initOptimizer(opt);
    cv::Mat K(3, 3, CV_32FC1);
    for (int i = 0; i < 9; ++i) K.at<float>(i / 3, i % 3) = K9[i];
    CamPara* cam = addCamPara(opt, K, 0);
    const g2o::SE3Quat Tbc = toSE3Quat(mat4f(Tbc12));
    for (int i = 0; i < P; ++i) addVertexSE2(opt, g2o::SE2(poses[3 * i], poses[3 * i + 1], poses[3 * i + 2]), i, fixed[i] != 0);
    std::vector<g2o::PreEdgeSE2*> odo;
    for (int k = 0; k < O; ++k) {
        g2o::Matrix3D info;
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) info(r, c) = o_info9[9 * k + 3 * r + c];
        odo.push_back(addEdgeSE2(opt, g2o::Vector3D(o_meas[3 * k], o_meas[3 * k + 1], o_meas[3 * k + 2]), o_i[k], o_j[k], info));
    }
    for (int l = 0; l < L; ++l) addVertexSBAXYZ(opt, Eigen::Vector3d(lms[3 * l], lms[3 * l + 1], lms[3 * l + 2]), P + l, true, false);
    std::vector<g2o::EdgeSE2XYZ*> obs;
    for (int k = 0; k < E; ++k) {
        g2o::Matrix2D info;
        info(0, 0) = e_info3[3 * k]; info(0, 1) = info(1, 0) = e_info3[3 * k + 1]; info(1, 1) = e_info3[3 * k + 2];
        obs.push_back(addEdgeSE2XYZ(opt, g2o::Vector2D(e_uv[2 * k], e_uv[2 * k + 1]), e_kf[k], P + e_lm[k], cam, Tbc, info, huber));
    }
    double total = 0;
    for (int k = 0; k < O; ++k) { odo[k]->computeError(); chi_o[k] = odo[k]->chi2(); total += chi_o[k]; }
    for (int k = 0; k < E; ++k) {
        obs[k]->computeError();
        chi_e[k] = obs[k]->chi2();
        total += obs[k]->robustKernel()->rho(chi_e[k]);
    }
    int nfixed = 0, nmarg = 0;
    for (int i = 0; i < P; ++i) nfixed += static_cast<g2o::VertexSE2*>(opt.vertex(i))->fixed();
    for (int l = 0; l < L; ++l) nmarg += static_cast<g2o::VertexSBAPointXYZ*>(opt.vertex(P + l))->marginalized();
    counts3[0] = (int32_t)opt.edges().size(); counts3[1] = nfixed; counts3[2] = nmarg;
    return total;
}

}  // extern "C"

// ---- src/sparsifier.cpp
extern "C" {
// Sparsifier::DoMarginalizeSE3XYZ (src/sparsifier.cpp:105-177): kf = two poses T_w_c as (R row-major, t), N map points, M
// measurements (key frame 0 / 1 - others are ignored by the reference -, map point id, 3x3 information) -> z_out (R, t), info_out 6x6
void ref_sparsify(const double* kf24, int N, const double* mp, int M, const int32_t* m_kf, const int32_t* m_mp, const double* m_info9,
                  double* z12, double* info36) {
    std::vector<g2o::SE3Quat, Eigen::aligned_allocator<g2o::SE3Quat>> vKF{quat12(kf24), quat12(kf24 + 12)};
    std::vector<g2o::Vector3D, Eigen::aligned_allocator<g2o::Vector3D>> vMP;
    for (int i = 0; i < N; ++i) vMP.push_back(g2o::Vector3D(mp[3 * i], mp[3 * i + 1], mp[3 * i + 2]));
    std::vector<MeasSE3XYZ, Eigen::aligned_allocator<MeasSE3XYZ>> vMeas((size_t)M);
    for (int k = 0; k < M; ++k) {
        vMeas[k].idKF = m_kf[k];
        vMeas[k].idMP = m_mp[k];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) vMeas[k].info(r, c) = m_info9[9 * k + 3 * r + c];
    }
    g2o::SE3Quat z;
    g2o::Matrix6d info;
    Sparsifier::DoMarginalizeSE3XYZ(vKF, vMP, vMeas, z, info);
    put12(z.rotation().toRotationMatrix(), z.translation(), z12);
    put36(info, info36);
}
// Sparsifier::HessianSE3XYZ (:95-102) with its forward-difference Jacobian (:59-93): J 3x9, H 9x9
void ref_sparsify_hessian(const double* kf12, const double* mp3, const double* info9, double* J27, double* H81) {
    const g2o::SE3Quat KF = quat12(kf12);
    const g2o::Vector3D MP(mp3[0], mp3[1], mp3[2]);
    g2o::Matrix3D info;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) info(r, c) = info9[3 * r + c];
    Eigen::Matrix<double, 3, 9> J;
    Eigen::Matrix<double, 9, 9> H;
    Sparsifier::JacobianSE3XYZ(KF, MP, J);
    Sparsifier::HessianSE3XYZ(KF, MP, info, H);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 9; ++c) J27[9 * r + c] = J(r, c);
    for (int r = 0; r < 9; ++r) for (int c = 0; c < 9; ++c) H81[9 * r + c] = H(r, c);
}
// Sparsifier::InfoSE3 (:219-275) on a given 12x12 marginal Hessian
void ref_sparsify_info_se3(const double* kf24, const double* H144, double* I36) {
    Eigen::Matrix<double, 12, 12> H;
    for (int r = 0; r < 12; ++r) for (int c = 0; c < 12; ++c) H(r, c) = H144[12 * r + c];
    Eigen::Matrix<double, 6, 6> I;
    Sparsifier::InfoSE3(quat12(kf24), quat12(kf24 + 12), H, I);
    put36(I, I36);
}
}  // extern "C"
