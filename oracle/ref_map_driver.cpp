// ORACLE - TEST INFRASTRUCTURE ONLY (oracle/_ref/libse2lam_ref_map.so).  Never linked, imported or called by the product path.
//
// C entry points over the REFERENCE's own map data model, compiled from /root/reference where the sources lie (oracle/Makefile,
// target `ref`; nothing of the reference is copied into this repository):
//   se2lam::Map::insertKF / insertMP / setCurrentKF / updateLocalGraph / loadLocalGraph(SlamOptimizer&)      src/Map.cpp:35-139, 285-331, 891-1053
//   se2lam::Map::loadLocalGraph(SlamOptimizer&, vpEdgesAll, vnAllIdx) - the SE3-expmap local graph          src/Map.cpp:414-566
//   se2lam::KeyFrame (constructor from a Frame, setPose(Se2), setViewMP, addObservation, addCovisibleKF,
//                     getAllObsMPs, getAllCovisibleKFs, preOdomFromSelf)                                     src/KeyFrame.cpp
//   se2lam::MapPoint (constructor, addObservation with updateMainKFandDescriptor, getObservations, getOctave,
//                     getFtrIdx)                                                                            src/MapPoint.cpp
// with src/optimizer.cpp's add* calls recording into the SparseOptimizer of oracle/_shim/g2o_shim.hpp.  KeyFrame.h,
// MapPoint.h, Map.h are the reference's own headers here (the front-end library libse2lam_ref.so stubs them); what is cut
// off is listed in oracle/_shim/se2lam_stubs_map.h.  The driver builds a map from flat arrays - the same arrays the product's
// CSR map view (se2gpu_map_update_local_graph) and POD loader (se2gpu_ba_load_local_graph) take - runs the reference's
// functions and hands back what they selected / put into the graph.
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <vector>

#include "Map.h"
#include "MapPoint.h"
#include "KeyFrame.h"
#include "converter.h"
#include "cvutil.h"
#include "optimizer.h"

using namespace se2lam;

#include "ref_map_state.h"

namespace {
cv::Mat mat_of(const float* v, int rows, int cols) {
    cv::Mat m(rows, cols, CV_32FC1);
    for (int r = 0; r < rows; ++r) for (int c = 0; c < cols; ++c) m.at<float>(r, c) = v[r * cols + c];
    return m;
}
}  // namespace

extern "C" {

// Config as Config::readConfig leaves it for this path: Kcam, bTc (cTb = its inverse), fxCam, TH_HUBER, the plane-motion
// informations, the pyramid (KeyFrame's level sigmas)
void* ref_map_create(const float* K9, const float* bTc16, float th_huber, float xrot_info, float yrot_info, float z_info, int max_level,
                     float scale_factor) {
    Config::Kcam = mat_of(K9, 3, 3);
    Config::fxCam = K9[0];
    Config::fyCam = K9[4];
    Config::bTc = mat_of(bTc16, 4, 4);
    Config::cTb = cvu::inv(Config::bTc);
    Config::TH_HUBER = th_huber;
    Config::PLANEMOTION_XROT_INFO = xrot_info;
    Config::PLANEMOTION_YROT_INFO = yrot_info;
    Config::PLANEMOTION_Z_INFO = z_info;
    Config::MaxLevel = max_level;
    Config::ScaleFactor = scale_factor;
    KeyFrame::mNextIdKF = 0;
    MapPoint::mNextId = 0;
    return new RefMap;
}
void ref_map_destroy(void* h) { delete static_cast<RefMap*>(h); }

// a key frame as Track hands it to the map: a Frame with nkp key points (position, octave), zero descriptors, the level
// sigmas of the extractor, then KeyFrame(frame); mIdKF / Frame::id as given; pose from the body pose (KeyFrame::setPose(Se2):
// Tcw = cTb * Twb^-1); mViewMPs[idx] = the map point in camera coordinates.  Returns the key frame's index.
int ref_map_add_kf(void* h, int id_kf, int frame_id, const float* twb3, int nkp, const float* kp_xy, const int32_t* kp_octave, const float* view_lc) {
    RefMap* m = static_cast<RefMap*>(h);
    Frame f;
    f.N = nkp;
    f.keyPoints.resize(nkp);
    for (int i = 0; i < nkp; ++i) {
        f.keyPoints[i].pt = cv::Point2f(kp_xy[2 * i], kp_xy[2 * i + 1]);
        f.keyPoints[i].octave = kp_octave[i];
    }
    f.keyPointsUn = f.keyPoints;
    f.descriptors = cv::Mat::zeros(std::max(nkp, 1), 32, CV_8UC1);
    f.mnScaleLevels = Config::MaxLevel;
    f.mfScaleFactor = Config::ScaleFactor;
    f.mvScaleFactors.resize(f.mnScaleLevels);
    f.mvLevelSigma2.resize(f.mnScaleLevels);
    f.mvInvLevelSigma2.resize(f.mnScaleLevels);
    f.mvScaleFactors[0] = 1.0f;                                       // ORBextractor::ORBextractor (src/ORBextractor.cpp:411-426) / Frame::Frame (:54-66)
    f.mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < f.mnScaleLevels; ++i) {
        f.mvScaleFactors[i] = f.mvScaleFactors[i - 1] * f.mfScaleFactor;
        f.mvLevelSigma2[i] = f.mvScaleFactors[i] * f.mvScaleFactors[i];
    }
    for (int i = 0; i < f.mnScaleLevels; ++i) f.mvInvLevelSigma2[i] = 1.0f / f.mvLevelSigma2[i];
    f.id = frame_id;
    f.Tcw = cv::Mat::eye(4, 4, CV_32FC1);
    f.Tcr = cv::Mat::eye(4, 4, CV_32FC1);
    PtrKeyFrame kf = std::make_shared<KeyFrame>(f);
    kf->mIdKF = id_kf;
    kf->setPose(Se2(twb3[0], twb3[1], twb3[2]));
    for (int i = 0; i < nkp; ++i) kf->setViewMP(cv::Point3f(view_lc[3 * i], view_lc[3 * i + 1], view_lc[3 * i + 2]), i, Eigen::Matrix3d::Identity());
    m->map.insertKF(kf);
    m->kfs.push_back(kf);
    return (int)m->kfs.size() - 1;
}
int ref_map_add_mp(void* h, int id_mp, const float* pos3) {
    RefMap* m = static_cast<RefMap*>(h);
    PtrMapPoint mp = std::make_shared<MapPoint>(cv::Point3f(pos3[0], pos3[1], pos3[2]), true);
    mp->mId = id_mp;
    m->map.insertMP(mp);
    m->mps.push_back(mp);
    return (int)m->mps.size() - 1;
}
// LocalMapper::addNewKF's pairing (src/LocalMapper.cpp): the key frame observes the map point at key point ftr, and vice versa
void ref_map_observe(void* h, int kf, int mp, int ftr) {
    RefMap* m = static_cast<RefMap*>(h);
    m->kfs[kf]->addObservation(m->mps[mp], ftr);
    m->mps[mp]->addObservation(m->kfs[kf], ftr);
}
void ref_map_covisible(void* h, int a, int b) { static_cast<RefMap*>(h)->kfs[a]->addCovisibleKF(static_cast<RefMap*>(h)->kfs[b]); }
void ref_map_set_odo(void* h, int from, int to, const double* meas3, const double* cov9) {   // preOdomFromSelf (Track.cpp:179-188)
    RefMap* m = static_cast<RefMap*>(h);
    PreSE2 p;
    std::memcpy(p.meas, meas3, sizeof(p.meas));
    std::memcpy(p.cov, cov9, sizeof(p.cov));
    m->kfs[from]->preOdomFromSelf = std::make_pair(m->kfs[to], p);
}
// KeyFrame::addFtrMeasureFrom / addFtrMeasureTo (src/KeyFrame.cpp:215-229): the feature constraint GlobalMapper::CreateFeatEdge leaves
void ref_map_add_ftr_measure(void* h, int from, int to, const float* measure16, const float* info36) {
    RefMap* m = static_cast<RefMap*>(h);
    m->kfs[from]->addFtrMeasureFrom(m->kfs[to], mat_of(measure16, 4, 4), mat_of(info36, 6, 6));
    m->kfs[to]->addFtrMeasureTo(m->kfs[from], mat_of(measure16, 4, 4), mat_of(info36, 6, 6));
}
void ref_map_mp_pos(void* h, int mp, float* pos3) {
    const cv::Point3f p = static_cast<RefMap*>(h)->mps[mp]->getPos();
    pos3[0] = p.x; pos3[1] = p.y; pos3[2] = p.z;
}
// the camera pose the reference derived from the body pose (CV_32F, row-major 4x4)
void ref_map_kf_pose(void* h, int kf, float* Tcw16) {
    const cv::Mat T = static_cast<RefMap*>(h)->kfs[kf]->getPose();
    for (int i = 0; i < 16; ++i) Tcw16[i] = T.at<float>(i / 4, i % 4);
}

// Map::setCurrentKF + Map::updateLocalGraph (src/Map.cpp:285-331) -> KeyFrame::mIdKF of the local and the reference key
// frames, MapPoint::mId of the local map points, each in the order of the reference's vectors.  counts3 = {local, ref, points}.
int ref_map_update_local_graph(void* h, int current_kf, int32_t* local_kf, int32_t* ref_kf, int32_t* local_mp, int cap_kf, int cap_mp, int32_t* counts3) {
    RefMap* m = static_cast<RefMap*>(h);
    m->map.setCurrentKF(m->kfs[current_kf]);
    m->map.updateLocalGraph();
    const std::vector<PtrKeyFrame> lk = m->map.getLocalKFs(), rk = m->map.getRefKFs();
    const std::vector<PtrMapPoint> lm = m->map.getLocalMPs();
    counts3[0] = (int32_t)lk.size(); counts3[1] = (int32_t)rk.size(); counts3[2] = (int32_t)lm.size();
    if ((int)lk.size() > cap_kf || (int)rk.size() > cap_kf || (int)lm.size() > cap_mp) return -1;
    for (size_t i = 0; i < lk.size(); ++i) local_kf[i] = lk[i]->mIdKF;
    for (size_t i = 0; i < rk.size(); ++i) ref_kf[i] = rk[i]->mIdKF;
    for (size_t i = 0; i < lm.size(); ++i) local_mp[i] = lm[i]->mId;
    return 0;
}

// Map::loadLocalGraph(SlamOptimizer&) (src/Map.cpp:891-1053) on the local graph of the last updateLocalGraph, into the
// recording optimizer; what it holds afterwards:
//   vertices (in id order): id, kind (0 = VertexSE2, 1 = VertexSBAPointXYZ), estimate (3), fixed, marginalized
//   PreEdgeSE2 (insertion order): the two vertex ids, measurement (3), information (9), chi2 at the start
//   EdgeSE2XYZ (insertion order): key-frame vertex id, map-point vertex id, measurement (2), information (4), Huber delta, chi2
// counts3 = {vertices, odometry edges, observation edges}; returns sum rho(chi2) as g2o's activeRobustChi2, < 0 when a cap is too small.
double ref_map_load_local_graph(void* h, int cap_v, int32_t* v_id, int32_t* v_kind, double* v_est, uint8_t* v_flags, int cap_o, int32_t* o_ids,
                                double* o_meas, double* o_info, double* o_chi2, int cap_e, int32_t* e_ids, double* e_uv, double* e_info,
                                double* e_delta, double* e_chi2, int32_t* counts3) {
    RefMap* m = static_cast<RefMap*>(h);
    SlamOptimizer opt;
    initOptimizer(opt);
    m->map.loadLocalGraph(opt);
    int nv = 0, no = 0, ne = 0;
    for (const auto& kv : opt.vertices()) {
        if (nv < cap_v) {
            v_id[nv] = kv.first;
            if (const g2o::VertexSE2* v = dynamic_cast<const g2o::VertexSE2*>(kv.second)) {
                v_kind[nv] = 0;
                const g2o::Vector3D e = v->estimate().toVector();
                for (int i = 0; i < 3; ++i) v_est[3 * nv + i] = e[i];
            } else if (const g2o::VertexSBAPointXYZ* p = dynamic_cast<const g2o::VertexSBAPointXYZ*>(kv.second)) {
                v_kind[nv] = 1;
                for (int i = 0; i < 3; ++i) v_est[3 * nv + i] = p->estimate()[i];
            } else {
                v_kind[nv] = -1;
            }
            v_flags[nv] = (uint8_t)((kv.second->fixed() ? 1 : 0) | (kv.second->marginalized() ? 2 : 0));
        }
        ++nv;
    }
    double total = 0;
    for (g2o::OptimizableGraph::Edge* e : opt.edges()) {
        e->computeError();
        if (g2o::EdgeSE2XYZ* x = dynamic_cast<g2o::EdgeSE2XYZ*>(e)) {
            const double c = x->chi2();
            total += x->robustKernel() ? x->robustKernel()->rho(c) : c;
            if (ne < cap_e) {
                e_ids[2 * ne] = x->vertices()[0]->id(); e_ids[2 * ne + 1] = x->vertices()[1]->id();
                e_uv[2 * ne] = x->measurement()[0]; e_uv[2 * ne + 1] = x->measurement()[1];
                for (int r = 0; r < 2; ++r) for (int cc = 0; cc < 2; ++cc) e_info[4 * ne + 2 * r + cc] = x->information()(r, cc);
                e_delta[ne] = x->robustKernel() ? x->robustKernel()->delta() : 0.0;
                e_chi2[ne] = c;
            }
            ++ne;
        } else if (g2o::PreEdgeSE2* p = dynamic_cast<g2o::PreEdgeSE2*>(e)) {
            const double c = p->chi2();
            total += c;
            if (no < cap_o) {
                o_ids[2 * no] = p->vertices()[0]->id(); o_ids[2 * no + 1] = p->vertices()[1]->id();
                for (int i = 0; i < 3; ++i) o_meas[3 * no + i] = p->measurement()[i];
                for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) o_info[9 * no + 3 * r + cc] = p->information()(r, cc);
                o_chi2[no] = c;
            }
            ++no;
        }
    }
    counts3[0] = nv; counts3[1] = no; counts3[2] = ne;
    if (nv > cap_v || no > cap_o || ne > cap_e) return -1.0;
    return total;
}

}  // extern "C"

// ---- the SE3-expmap variant
namespace {
void put12(const g2o::SE3Quat& T, double* o) {
    const Eigen::Matrix3d R = T.rotation().toRotationMatrix();
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) o[3 * r + c] = R(r, c); o[9 + r] = T.translation()[r]; }
}
}  // namespace
extern "C" {
// KeyFrame::setOdoMeasureFrom (src/KeyFrame.cpp:231-233): the odometry constraint FROM key frame `kf` to `to` as Track leaves it
// (measure: 4x4 CV_32F, info: 6x6 CV_32F in (translation, rotation) order)
void ref_map_set_odo_se3(void* h, int kf, int to, const float* measure16, const float* info36) {
    RefMap* m = static_cast<RefMap*>(h);
    m->kfs[kf]->setOdoMeasureFrom(m->kfs[to], mat_of(measure16, 4, 4), mat_of(info36, 6, 6));
}
// Map::loadLocalGraph(optimizer, vpEdgesAll, vnAllIdx) (src/Map.cpp:414-566) on the local graph of the last updateLocalGraph:
//   vertices (id order): id, kind (2 = VertexSE3Expmap with estimate (R, t) in v_est12, 1 = VertexSBAPointXYZ in its first 3), flags
//   EdgeSE3ExpmapPrior: vertex id, measurement (R, t), information (36), chi2          [se2lam's own edge]
//   EdgeSE3Expmap: the two vertex ids, measurement, information as the edge holds it, chi2
//   EdgeProjectXYZ2UV: map-point vertex, key-frame vertex, uv, information (4), Huber delta, level, chi2
// counts5 = {vertices, priors, odometry edges, projection edges, sum of vpEdgesAll[i].size()}; returns sum rho(chi2).
double ref_map_load_local_graph_se3(void* h, int cap_v, int32_t* v_id, int32_t* v_kind, double* v_est12, uint8_t* v_flags, int cap_p, int32_t* p_id,
                                    double* p_meas12, double* p_info36, double* p_chi2, int cap_o, int32_t* o_ids, double* o_meas12, double* o_info36,
                                    double* o_chi2, int cap_e, int32_t* e_ids, double* e_uv, double* e_info4, double* e_delta, int32_t* e_level,
                                    double* e_chi2, int32_t* counts5) {
    RefMap* m = static_cast<RefMap*>(h);
    SlamOptimizer opt;
    initOptimizer(opt);
    std::vector<std::vector<g2o::EdgeProjectXYZ2UV*>> vpEdgesAll;
    std::vector<std::vector<int>> vnAllIdx;
    m->map.loadLocalGraph(opt, vpEdgesAll, vnAllIdx);
    int nv = 0, np = 0, no = 0, ne = 0, nall = 0;
    for (const auto& v : vpEdgesAll) nall += (int)v.size();
    for (const auto& kv : opt.vertices()) {
        if (nv < cap_v) {
            v_id[nv] = kv.first;
            for (int i = 0; i < 12; ++i) v_est12[12 * nv + i] = 0;
            if (const g2o::VertexSE3Expmap* v = dynamic_cast<const g2o::VertexSE3Expmap*>(kv.second)) {
                v_kind[nv] = 2;
                put12(v->estimate(), v_est12 + 12 * nv);
            } else if (const g2o::VertexSBAPointXYZ* p = dynamic_cast<const g2o::VertexSBAPointXYZ*>(kv.second)) {
                v_kind[nv] = 1;
                for (int i = 0; i < 3; ++i) v_est12[12 * nv + i] = p->estimate()[i];
            } else {
                v_kind[nv] = -1;
            }
            v_flags[nv] = (uint8_t)((kv.second->fixed() ? 1 : 0) | (kv.second->marginalized() ? 2 : 0));
        }
        ++nv;
    }
    double total = 0;
    for (g2o::OptimizableGraph::Edge* e : opt.edges()) {
        e->computeError();
        const double c = e->chi2();
        total += e->robustKernel() ? e->robustKernel()->rho(c) : c;
        if (EdgeSE3ExpmapPrior* x = dynamic_cast<EdgeSE3ExpmapPrior*>(e)) {
            if (np < cap_p) {
                p_id[np] = x->vertices()[0]->id();
                put12(x->measurement(), p_meas12 + 12 * np);
                for (int r = 0; r < 6; ++r) for (int cc = 0; cc < 6; ++cc) p_info36[36 * np + 6 * r + cc] = x->information()(r, cc);
                p_chi2[np] = c;
            }
            ++np;
        } else if (g2o::EdgeSE3Expmap* x = dynamic_cast<g2o::EdgeSE3Expmap*>(e)) {
            if (no < cap_o) {
                o_ids[2 * no] = x->vertices()[0]->id(); o_ids[2 * no + 1] = x->vertices()[1]->id();
                put12(x->measurement(), o_meas12 + 12 * no);
                for (int r = 0; r < 6; ++r) for (int cc = 0; cc < 6; ++cc) o_info36[36 * no + 6 * r + cc] = x->information()(r, cc);
                o_chi2[no] = c;
            }
            ++no;
        } else if (g2o::EdgeProjectXYZ2UV* x = dynamic_cast<g2o::EdgeProjectXYZ2UV*>(e)) {
            if (ne < cap_e) {
                e_ids[2 * ne] = x->vertices()[0]->id(); e_ids[2 * ne + 1] = x->vertices()[1]->id();
                e_uv[2 * ne] = x->measurement()[0]; e_uv[2 * ne + 1] = x->measurement()[1];
                for (int r = 0; r < 2; ++r) for (int cc = 0; cc < 2; ++cc) e_info4[4 * ne + 2 * r + cc] = x->information()(r, cc);
                e_delta[ne] = x->robustKernel() ? x->robustKernel()->delta() : 0.0;
                e_level[ne] = x->level();
                e_chi2[ne] = c;
            }
            ++ne;
        }
    }
    counts5[0] = nv; counts5[1] = np; counts5[2] = no; counts5[3] = ne; counts5[4] = nall;
    if (nv > cap_v || np > cap_p || no > cap_o || ne > cap_e) return -1.0;
    return total;
}
}  // extern "C"
