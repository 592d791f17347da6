// ORACLE - TEST INFRASTRUCTURE ONLY (oracle/_ref/libse2lam_ref_map.so).  Never linked, imported or called by the product path.
//
// C entry points over the REFERENCE's own thread classes, compiled from /root/reference where the sources lie (oracle/Makefile,
// target `ref`; nothing of the reference is copied into this repository):
//   se2lam::Track::doTriangulate                 src/Track.cpp:373-415     (with cvu::triangulate / checkParallax, Config::acceptDepth)
//   se2lam::Track::updateFramePose               src/Track.cpp:162-188     (frame pose from the odometry, the SE(2) pre-integration)
//   se2lam::Localizer::DoLocalBA                 src/Localizer.cpp:233-302 (the pose-only graph, handed over at optimize())
//   se2lam::GlobalMapper::GlobalBA               src/GlobalMapper.cpp:328-535 (the pose graph of all key frames, handed over at optimize();
//                                                then the write-back of key-frame poses and map-point positions)
// The members these functions work on are private in the reference's headers; this translation unit - and only this one -
// reads the headers with `private` / `protected` spelled `public` (access specifiers do not change the object layout with
// this compiler, and no reference source is touched).  g2o's optimize() is the stand-in's: it hands the graph, as the
// reference built it, to the hook registered here and returns without moving anything.
#include <cstdint>
#include <cstring>
#include <deque>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <stdexcept>
#include <thread>
#include <vector>

#include <opencv2/core/core.hpp>
#include <g2o_shim.hpp>

#define private public
#define protected public
#include "KeyFrame.h"
#include "MapPoint.h"
#include "Map.h"
#include "Track.h"
#include "Localizer.h"
#include "LocalMapper.h"
#include "GlobalMapper.h"
#undef private
#undef protected
#include "converter.h"
#include "cvutil.h"
#include "optimizer.h"
#include "ref_map_state.h"

using namespace se2lam;

namespace {
cv::Mat mat_of(const float* v, int rows, int cols) {
    cv::Mat m(rows, cols, CV_32FC1);
    for (int r = 0; r < rows; ++r) for (int c = 0; c < cols; ++c) m.at<float>(r, c) = v[r * cols + c];
    return m;
}
void set_pyramid(Frame& f) {   // what Frame::Frame copies from the extractor (src/Frame.cpp:54-66)
    f.mnScaleLevels = Config::MaxLevel;
    f.mfScaleFactor = Config::ScaleFactor;
    f.mvScaleFactors.assign(f.mnScaleLevels, 1.0f);
    f.mvLevelSigma2.assign(f.mnScaleLevels, 1.0f);
    f.mvInvLevelSigma2.assign(f.mnScaleLevels, 1.0f);
    for (int i = 1; i < f.mnScaleLevels; ++i) {
        f.mvScaleFactors[i] = f.mvScaleFactors[i - 1] * f.mfScaleFactor;
        f.mvLevelSigma2[i] = f.mvScaleFactors[i] * f.mvScaleFactors[i];
    }
    for (int i = 0; i < f.mnScaleLevels; ++i) f.mvInvLevelSigma2[i] = 1.0f / f.mvLevelSigma2[i];
}
struct ref_keypoint {  // cv::KeyPoint layout
    float x, y, size, angle, response;
    int32_t octave, class_id;
};
std::vector<cv::KeyPoint> kps_of(const ref_keypoint* k, int n) {
    std::vector<cv::KeyPoint> v((size_t)n);
    for (int i = 0; i < n; ++i) v[i] = cv::KeyPoint(k[i].x, k[i].y, k[i].size, k[i].angle, k[i].response, k[i].octave, k[i].class_id);
    return v;
}
void base_config(const float* K9, int max_features, int max_level, float scale_factor) {
    Config::Kcam = mat_of(K9, 3, 3);
    Config::fxCam = K9[0];
    Config::fyCam = K9[4];
    Config::PrjMtrxEye = Config::Kcam * cv::Mat::eye(3, 4, CV_32FC1);       // Config::readConfig (src/Config.cpp:123)
    Config::MaxFtrNumber = max_features;
    Config::MaxLevel = max_level;
    Config::ScaleFactor = scale_factor;
    Config::FPS = 30;
}
}  // namespace

extern "C" {

// Track::doTriangulate: the reference key frame's key points (n_ref) with has_obs[i] = it already observes a map point there
// (whose camera-frame position is view_mp[i]), the current frame's key points, match_idx, Tcr of the current frame.
// Returns nTrackedOld; pos = mLocalMPs, good = mvbGoodPrl, match_idx updated, n_good = mnGoodPrl.  frame_gap = mFrame.id - mpKF->id
// (below nMinFrames = 8 the function returns 0 at once).
int ref_track_triangulate(const float* K9, float lower_depth, float upper_depth, int n_ref, const ref_keypoint* kps_ref, const uint8_t* has_obs,
                          const float* view_mp, int n_cur, const ref_keypoint* kps_cur, int32_t* match_idx, const float* Tcr16, int frame_gap,
                          float* pos, uint8_t* good, int32_t* n_good) {
    base_config(K9, std::max(n_ref, 1), 8, 1.2f);
    Config::LOWER_DEPTH = lower_depth;
    Config::UPPER_DEPTH = upper_depth;
    Track t;
    Frame ref;
    ref.N = n_ref;
    ref.keyPoints = ref.keyPointsUn = kps_of(kps_ref, n_ref);
    ref.descriptors = cv::Mat::zeros(std::max(n_ref, 1), 32, CV_8UC1);
    set_pyramid(ref);
    ref.id = 0;
    ref.Tcw = cv::Mat::eye(4, 4, CV_32FC1);
    ref.Tcr = cv::Mat::eye(4, 4, CV_32FC1);
    t.mRefFrame = ref;
    t.mpKF = std::make_shared<KeyFrame>(ref);
    t.mpKF->id = 0;
    PtrMapPoint seen = std::make_shared<MapPoint>(cv::Point3f(0, 0, 1), true);    // hasObservation(i) only asks whether the index is taken
    for (int i = 0; i < n_ref; ++i)
        if (has_obs[i]) {
            t.mpKF->mViewMPs[i] = cv::Point3f(view_mp[3 * i], view_mp[3 * i + 1], view_mp[3 * i + 2]);
            t.mpKF->mDualObservations[i] = seen;
        }
    Frame cur;
    cur.N = n_cur;
    cur.keyPoints = cur.keyPointsUn = kps_of(kps_cur, n_cur);
    set_pyramid(cur);
    cur.id = frame_gap;
    cur.Tcr = mat_of(Tcr16, 4, 4);
    cur.Tcw = cur.Tcr.clone();
    t.mFrame = cur;
    t.mMatchIdx.assign(match_idx, match_idx + n_ref);
    t.mLocalMPs.assign((size_t)std::max(n_ref, 1), cv::Point3f(-1, -1, -1));
    const int n_old = t.doTriangulate();
    for (int i = 0; i < n_ref; ++i) {
        match_idx[i] = t.mMatchIdx[i];
        pos[3 * i] = t.mLocalMPs[i].x; pos[3 * i + 1] = t.mLocalMPs[i].y; pos[3 * i + 2] = t.mLocalMPs[i].z;
        good[i] = (i < (int)t.mvbGoodPrl.size() && t.mvbGoodPrl[i]) ? 1 : 0;
    }
    *n_good = t.mnGoodPrl;
    return n_old;
}

// Track::updateFramePose: the key frame's odometry / body pose / camera pose, the last and the current odometry reading, the
// pre-integrated measurement and covariance so far (in / out).  Out: Trb, Twb of the frame (3 floats each), Tcr, Tcw (16 floats each).
void ref_track_update_frame_pose(const float* bTc16, const float* noise3, const float* kf_odom3, const float* kf_twb3, const float* last_odom3,
                                 const float* odom3, double* meas3, double* cov9, float* trb3, float* twb3, float* Tcr16, float* Tcw16) {
    const float K[9] = {400, 0, 320, 0, 400, 240, 0, 0, 1};
    base_config(K, 1, 8, 1.2f);
    Config::bTc = mat_of(bTc16, 4, 4);
    Config::cTb = cvu::inv(Config::bTc);
    Config::ODO_X_NOISE = noise3[0];
    Config::ODO_Y_NOISE = noise3[1];
    Config::ODO_T_NOISE = noise3[2];
    Track t;
    Frame f;
    f.N = 0;
    set_pyramid(f);
    f.Tcw = cv::Mat::eye(4, 4, CV_32FC1);
    f.Tcr = cv::Mat::eye(4, 4, CV_32FC1);
    t.mpKF = std::make_shared<KeyFrame>(f);
    t.mpKF->odom = Se2(kf_odom3[0], kf_odom3[1], kf_odom3[2]);
    t.mpKF->setPose(Se2(kf_twb3[0], kf_twb3[1], kf_twb3[2]));
    t.mFrame = f;
    t.mFrame.odom = Se2(odom3[0], odom3[1], odom3[2]);
    t.lastOdom = Se2(last_odom3[0], last_odom3[1], last_odom3[2]);
    std::memcpy(t.preSE2.meas, meas3, sizeof(t.preSE2.meas));
    std::memcpy(t.preSE2.cov, cov9, sizeof(t.preSE2.cov));
    t.updateFramePose();
    std::memcpy(meas3, t.preSE2.meas, sizeof(t.preSE2.meas));
    std::memcpy(cov9, t.preSE2.cov, sizeof(t.preSE2.cov));
    trb3[0] = t.mFrame.Trb.x; trb3[1] = t.mFrame.Trb.y; trb3[2] = t.mFrame.Trb.theta;
    twb3[0] = t.mFrame.Twb.x; twb3[1] = t.mFrame.Twb.y; twb3[2] = t.mFrame.Twb.theta;
    for (int i = 0; i < 16; ++i) { Tcr16[i] = t.mFrame.Tcr.at<float>(i / 4, i % 4); Tcw16[i] = t.mFrame.Tcw.at<float>(i / 4, i % 4); }
}

// Localizer::DoLocalBA: the current key frame (pose Tcw, n key points with octaves, the n map points they observe: world
// position and whether the point has good parallax - the others are skipped, src/Localizer.cpp:262).  The graph reaches the
// hook at optimizer.optimize(30): out = {number of vertices, fixed vertices, projection edges, prior edges, iterations asked};
// prior measurement (R, t) / information, per kept point its chi2 (order of the reference's std::set: by address - returned
// with the point's index), returns sum rho(chi2) at the start.
double ref_localizer_do_local_ba(const float* K9, const float* bTc16, float th_huber, float xrot_info, float yrot_info, float z_info, int id_kf,
                                 const float* Tcw16, int n, const ref_keypoint* kps, const float* mp_pos, const uint8_t* mp_good, int32_t* out5,
                                 double* prior_meas12, double* prior_info36, int32_t* e_point, double* e_uv, double* e_w, double* e_chi2, double* e_delta) {
    base_config(K9, std::max(n, 1), 8, 1.2f);
    Config::bTc = mat_of(bTc16, 4, 4);
    Config::cTb = cvu::inv(Config::bTc);
    Config::TH_HUBER = th_huber;
    Config::PLANEMOTION_XROT_INFO = xrot_info;
    Config::PLANEMOTION_YROT_INFO = yrot_info;
    Config::PLANEMOTION_Z_INFO = z_info;
    MapPoint::mNextId = 0;
    Frame f;
    f.N = n;
    f.keyPoints = f.keyPointsUn = kps_of(kps, n);
    f.descriptors = cv::Mat::zeros(std::max(n, 1), 32, CV_8UC1);
    set_pyramid(f);
    f.Tcw = mat_of(Tcw16, 4, 4);
    f.Tcr = cv::Mat::eye(4, 4, CV_32FC1);
    Localizer loc;
    loc.mpKFCurr = std::make_shared<KeyFrame>(f);
    loc.mpKFCurr->mIdKF = id_kf;
    std::vector<PtrMapPoint> mps;
    std::map<const g2o::HyperGraph::Vertex*, int> index_of;
    for (int i = 0; i < n; ++i) {
        PtrMapPoint mp = std::make_shared<MapPoint>(cv::Point3f(mp_pos[3 * i], mp_pos[3 * i + 1], mp_pos[3 * i + 2]), mp_good[i] != 0);
        // (the camera-frame position only feeds MapPoint's viewing-distance bookkeeping)
        loc.mpKFCurr->setViewMP(cvu::se3map(f.Tcw, mp->getPos()), i, Eigen::Matrix3d::Identity());
        loc.mpKFCurr->addObservation(mp, i);
        mp->mObservations[loc.mpKFCurr] = i;         // MapPoint::addObservation without its parallax update (a point may be handed over as "not good")
        mps.push_back(mp);
    }
    double total = -1;
    std::memset(out5, 0, 5 * sizeof(int32_t));
    g2o::SparseOptimizer::optimizeHook() = [&](g2o::SparseOptimizer& opt, int iterations) {
        total = 0;
        out5[0] = (int32_t)opt.vertices().size();
        for (const auto& kv : opt.vertices()) out5[1] += kv.second->fixed() ? 1 : 0;
        out5[4] = iterations;
        std::map<int, int> point_of_vertex;          // vertex id = maxKFid + pMP->mId, mId = 1 .. n in creation order
        for (int i = 0; i < n; ++i) point_of_vertex[id_kf + mps[i]->mId] = i;
        for (g2o::OptimizableGraph::Edge* e : opt.edges()) {
            e->computeError();
            const double c = e->chi2();
            total += e->robustKernel() ? e->robustKernel()->rho(c) : c;
            if (EdgeSE3ExpmapPrior* p = dynamic_cast<EdgeSE3ExpmapPrior*>(e)) {
                const Eigen::Matrix3d R = p->measurement().rotation().toRotationMatrix();
                for (int r = 0; r < 3; ++r) { for (int cc = 0; cc < 3; ++cc) prior_meas12[3 * r + cc] = R(r, cc); prior_meas12[9 + r] = p->measurement().translation()[r]; }
                for (int r = 0; r < 6; ++r) for (int cc = 0; cc < 6; ++cc) prior_info36[6 * r + cc] = p->information()(r, cc);
                ++out5[3];
            } else if (g2o::EdgeProjectXYZ2UV* x = dynamic_cast<g2o::EdgeProjectXYZ2UV*>(e)) {
                const int k = out5[2]++;
                e_point[k] = point_of_vertex.at(x->vertices()[0]->id());
                e_uv[2 * k] = x->measurement()[0]; e_uv[2 * k + 1] = x->measurement()[1];
                e_w[k] = x->information()(0, 0);
                e_chi2[k] = c;
                e_delta[k] = x->robustKernel()->delta();
            }
        }
    };
    loc.DoLocalBA();
    g2o::SparseOptimizer::optimizeHook() = nullptr;
    return total;
}

}  // extern "C"

extern "C" {
// GlobalMapper::GlobalBA on the map of ref_map_* (key frame poses, mOdoMeasureFrom, mFtrMeasureFrom).  At optimize(GLOBAL_ITER):
//   vertices (id order): id = mIdKF, estimate T_w_c (R, t), fixed
//   EdgeSE3Prior (insertion order): vertex id, measurement, information (translation, rotation), chi2
//   EdgeSE3 (insertion order: odometry edges, then feature edges): the two vertex ids, measurement, information, chi2
// counts4 = {vertices, priors, edges, iterations asked}; returns sum chi2 at the start (no robust kernels in this graph).
// Afterwards the reference has written the (unchanged) estimates back into the key frames and re-anchored the map points.
double ref_global_ba(void* h, int global_iter, int cap_v, int32_t* v_id, double* v_est12, uint8_t* v_fixed, int cap_p, int32_t* p_id, double* p_meas12,
                     double* p_info36, double* p_chi2, int cap_e, int32_t* e_ids, double* e_meas12, double* e_info36, double* e_chi2, int32_t* counts4) {
    RefMap* m = static_cast<RefMap*>(h);
    Config::GLOBAL_ITER = global_iter;
    Config::GLOBAL_VERBOSE = false;
    LocalMapper lm;
    GlobalMapper gm;
    gm.mpMap = &m->map;
    gm.mpLocalMapper = &lm;
    double total = -1;
    std::memset(counts4, 0, 4 * sizeof(int32_t));
    auto put_iso = [](const g2o::Isometry3D& T, double* o) {
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) o[3 * r + c] = T.linear()(r, c); o[9 + r] = T.translation()[r]; }
    };
    g2o::SparseOptimizer::optimizeHook() = [&](g2o::SparseOptimizer& opt, int iterations) {
        total = 0;
        counts4[3] = iterations;
        for (const auto& kv : opt.vertices()) {
            const int k = counts4[0]++;
            if (k >= cap_v) continue;
            v_id[k] = kv.first;
            put_iso(static_cast<const g2o::VertexSE3*>(kv.second)->estimate(), v_est12 + 12 * k);
            v_fixed[k] = kv.second->fixed() ? 1 : 0;
        }
        for (g2o::OptimizableGraph::Edge* e : opt.edges()) {
            e->computeError();
            const double c = e->chi2();
            total += c;
            if (g2o::EdgeSE3Prior* p = dynamic_cast<g2o::EdgeSE3Prior*>(e)) {
                const int k = counts4[1]++;
                if (k >= cap_p) continue;
                p_id[k] = p->vertices()[0]->id();
                put_iso(p->measurement(), p_meas12 + 12 * k);
                for (int r = 0; r < 6; ++r) for (int cc = 0; cc < 6; ++cc) p_info36[36 * k + 6 * r + cc] = p->information()(r, cc);
                p_chi2[k] = c;
            } else if (g2o::EdgeSE3* x = dynamic_cast<g2o::EdgeSE3*>(e)) {
                const int k = counts4[2]++;
                if (k >= cap_e) continue;
                e_ids[2 * k] = x->vertices()[0]->id(); e_ids[2 * k + 1] = x->vertices()[1]->id();
                put_iso(x->measurement(), e_meas12 + 12 * k);
                for (int r = 0; r < 6; ++r) for (int cc = 0; cc < 6; ++cc) e_info36[36 * k + 6 * r + cc] = x->information()(r, cc);
                e_chi2[k] = c;
            }
        }
    };
    gm.GlobalBA();
    g2o::SparseOptimizer::optimizeHook() = nullptr;
    if (counts4[0] > cap_v || counts4[1] > cap_p || counts4[2] > cap_e) return -1.0;
    return total;
}
}  // extern "C"
