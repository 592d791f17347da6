// Drop-in mirror of the reference's optimiser call surface over libse2gpu:
//   /root/reference/include/se2lam/optimizer.h:30-34  typedefs (SlamOptimizer = g2o::SparseOptimizer with
//        OptimizationAlgorithmLevenberg + BlockSolverX + a dense pose solve in place of LinearSolverCholmod)
//   :78 initOptimizer  :85 addCamPara  :91 addVertexSBAXYZ  :100 addEdgeSE2XYZ  :104 addVertexSE2
//   :107 estimateVertexSE2  :109 addEdgeSE2  :141 estimateVertexSBAXYZ
// and the g2o::SparseOptimizer methods LocalMapper::localBA / Map::loadLocalGraph call
//   (/root/reference/src/LocalMapper.cpp:239-260, src/Map.cpp:891-1053).
// Signatures follow the reference's: tests/cpp_reference_call_lines.cpp pastes the call lines of Map.cpp:897, 928-929,
// 951, 966-970, 989, 1048-1049, LocalMapper.cpp:239-260, GlobalMapper.cpp:340-345 and Localizer.cpp:235-240 against this
// header - including the solver construction (SlamLinearSolver / SlamBlockSolver / SlamAlgorithm + setAlgorithm, tag
// types here: the dense pose solve and the Levenberg controller live in the library).  The free functions that return
// g2o vertex / edge pointers in the reference return address-stable handles owned by the optimizer (VertexSE2*,
// EdgeSE2XYZ*, PreEdgeSE2*); addCamPara returns a stable CamPara* as the reference does (Map.cpp:897).  With g2o / Eigen /
// OpenCV headers present, overloads taking the real types are compiled in (conversions.h).
#pragma once
#include <deque>
#include <stdexcept>
#include <vector>

#include "types.h"

namespace se2lam_amd {

struct CamPara {   // g2o::CameraParameters (focal_length, principle_point, baseline 0) + its parameter id
    double focal_length = 0;
    double principle_point[2] = {0, 0};
    int id = 0;
};

class SlamOptimizer;

// What the reference keeps of an EdgeProjectXYZ2UV* / EdgeSE3* after adding it (LocalMapper::removeOutlierChi2,
// LocalMapper.cpp:199-214; GlobalMapper::GlobalBA, GlobalMapper.cpp:415-476): computeError(), chi2(), setLevel(), level().
// chi2() of all edges comes from ONE device pass after optimize() (se2gpu_ba_edge_chi2), cached by the optimizer.
// setLevel(): for the EdgeSE3 edges of a pose graph it is g2o's - initializeOptimization() hands the levels to the library and a
// SECOND initializeOptimization on the same optimizer re-optimises over the level-0 edges from the current estimates
// (GlobalMapper.cpp:421-483, se2gpu_ba_set_edge_level).  For the landmark models it is bookkeeping for the caller's own loops
// (level() > 0 -> skip, LocalMapper.cpp:199-214): there an edge moved to another level makes initializeOptimization throw,
// and a second initializeOptimization is refused by the library - never a silent run over all edges.
struct EdgeHandle {
    SlamOptimizer* opt = nullptr;
    int index = -1;      // position among the edges whose chi2 the library reports; -1: an edge without per-edge chi2
    int level_ = 0;
    int pushed_level_ = 0;   // the level the library was last told (edges start at 0 there)
    void computeError() {}
    double chi2() const;
    void setLevel(int l);
    int level() const { return level_; }
};
typedef EdgeHandle EdgeProjectXYZ2UV;
typedef EdgeHandle EdgeSE3;
typedef EdgeHandle EdgeSE2XYZ;    // what addEdgeSE2XYZ returns (optimizer.h:100)
typedef EdgeHandle PreEdgeSE2;    // what addEdgeSE2 returns (optimizer.h:109)

struct VertexSE2 {                // what addVertexSE2 returns (optimizer.h:104): id() and estimate() of the g2o vertex
    SlamOptimizer* opt = nullptr;
    int id_ = -1;
    int id() const { return id_; }
    SE2 estimate() const;
};

// The reference builds its solver stack by hand in front of every optimisation (LocalMapper.cpp:240-243,
// GlobalMapper.cpp:341-344, Localizer.cpp:236-239, optimizer.cpp:200-203):
//     SlamLinearSolver* linearSolver = new SlamLinearSolver();
//     SlamBlockSolver*  blockSolver  = new SlamBlockSolver(linearSolver);
//     SlamAlgorithm*    solver       = new SlamAlgorithm(blockSolver);
//     optimizer.setAlgorithm(solver);
// Here the Schur complement, the dense pose solve and the Levenberg policy are fixed parts of libse2gpu, so the three
// types are tags with g2o's ownership chain (the optimizer deletes the algorithm, which deletes the block solver, which
// deletes the linear solver) - the lines above compile and run unchanged and leak nothing.
struct SlamLinearSolver {};
struct SlamBlockSolver {
    explicit SlamBlockSolver(SlamLinearSolver* ls) : ls_(ls) {}
    ~SlamBlockSolver() { delete ls_; }
    SlamBlockSolver(const SlamBlockSolver&) = delete;
    SlamBlockSolver& operator=(const SlamBlockSolver&) = delete;
private:
    SlamLinearSolver* ls_;
};
struct SlamAlgorithm {
    explicit SlamAlgorithm(SlamBlockSolver* bs) : bs_(bs) {}
    ~SlamAlgorithm() { delete bs_; }
    SlamAlgorithm(const SlamAlgorithm&) = delete;
    SlamAlgorithm& operator=(const SlamAlgorithm&) = delete;
    double currentLambda() const;                                         // REJECT_IF_LARGE_LAMBDA, LocalMapper.cpp:285-292
private:
    friend class SlamOptimizer;
    SlamBlockSolver* bs_;
    const SlamOptimizer* opt_ = nullptr;
};

class SlamOptimizer {  // g2o::SparseOptimizer (the subset the hot path uses)
public:
    SlamOptimizer() { check(se2gpu_ba_create(&h_), "SlamOptimizer"); }
    ~SlamOptimizer() { se2gpu_ba_destroy(h_); delete algorithm_; }
    SlamOptimizer(const SlamOptimizer&) = delete;
    SlamOptimizer& operator=(const SlamOptimizer&) = delete;

    void setAlgorithm(SlamAlgorithm* a) {                                 // LocalMapper.cpp:243: takes ownership, as g2o does
        if (a != algorithm_) delete algorithm_;
        algorithm_ = a;
        if (a) a->opt_ = this;
    }
    void setVerbose(bool v) { verbose_ = v; }
    void setForceStopFlag(bool* flag) { stop_ = flag; }                   // LocalMapper.cpp:246
    bool initializeOptimization(int level = 0) {                          // LocalMapper.cpp:259
        // g2o optimises over the edges of ONE level.  The pose graph's EdgeSE3 levels go to the library (which leaves the
        // others out); anywhere else an edge moved to another level would be optimised over all the same - refuse instead
        if (level != 0) throw std::runtime_error("initializeOptimization: only level 0 can be optimised");
        for (const EdgeHandle& e : plain_edges_) if (e.level_ != 0) throw std::runtime_error("initializeOptimization: an edge without a library index has been moved to another level (setLevel)");
        if (levels_touched_)
            for (EdgeHandle& e : edges_) {
                if (e.level_ == e.pushed_level_) continue;   // nothing to tell the library (setLevel(1) ... setLevel(0) on a landmark graph: no call at all)
                if (se2gpu_ba_set_edge_level(h_, e.index, e.level_) != SE2GPU_OK)
                    throw std::runtime_error("initializeOptimization: an edge has been moved to another level (setLevel); only the pose graph's EdgeSE3 have levels on the device");
                e.pushed_level_ = e.level_;
            }
        check(se2gpu_ba_initialize(h_), "initializeOptimization");
        edge_chi2_.clear();
        return true;
    }
    int optimize(int iterations) {                                        // LocalMapper.cpp:260
        static_assert(sizeof(bool) == 1, "the stop flag is polled as a byte");
        check(se2gpu_ba_optimize(h_, iterations, SE2GPU_BA_LM, reinterpret_cast<const volatile uint8_t*>(stop_),
                                 verbose_ ? 1 : 0, &stats_), "optimize");
        edge_chi2_.clear();
        return stats_.iterations;
    }
    void addEdge(EdgeHandle*) {}                                          // Map.cpp:551: the library added it already
    EdgeHandle* newEdge() { edges_.emplace_back(); edges_.back().opt = this; edges_.back().index = (int)edges_.size() - 1; return &edges_.back(); }
    EdgeHandle* newPlainEdge() { plain_edges_.emplace_back(); plain_edges_.back().opt = this; return &plain_edges_.back(); }
    VertexSE2* newVertexSE2(int id) { vertices_.emplace_back(); vertices_.back().opt = this; vertices_.back().id_ = id; return &vertices_.back(); }
    double edgeChi2(int index) {
        if (index < 0) throw std::runtime_error("chi2(): the library reports per-edge chi2 for EdgeProjectXYZ2UV / EdgeSE3 only");
        if (edge_chi2_.empty()) {
            edge_chi2_.assign(edges_.size(), 0.0);
            check(se2gpu_ba_edge_chi2(h_, edge_chi2_.data(), (int)edge_chi2_.size()), "EdgeProjectXYZ2UV::chi2");
        }
        return edge_chi2_.at(index);
    }
    // g2o::SparseOptimizer::clear(): vertices and edges are gone, every vertex / edge handle handed out so far is invalid.
    // Parameters survive it, as in g2o (OptimizableGraph::clear() keeps _parameters; clearParameters() drops them): a CamPara*
    // from addCamPara stays valid across clear() and is re-registered with the library's handle, which forgets its camera.
    void clear() {
        check(se2gpu_ba_clear(h_), "clear");
        edges_.clear(); plain_edges_.clear(); vertices_.clear(); edge_chi2_.clear(); levels_touched_ = false;
        for (const CamPara& c : cams_) check(se2gpu_ba_add_cam(h_, c.focal_length, c.principle_point[0], c.principle_point[1]), "clear: camera");
    }
    void clearParameters() { cams_.clear(); }
    double activeRobustChi2() { return se2gpu_ba_chi2(h_); }
    double currentLambda() const { return stats_.lambda_final; }          // REJECT_IF_LARGE_LAMBDA, LocalMapper.cpp:285-292
    const se2gpu_ba_stats& stats() const { return stats_; }
    se2gpu_ba* handle() { return h_; }
    CamPara* newCamPara() { cams_.emplace_back(); return &cams_.back(); }   // owned by the optimizer, address-stable

    void noteLevelChange() { levels_touched_ = true; }

private:
    bool levels_touched_ = false;
    std::deque<CamPara> cams_;
    std::deque<EdgeHandle> edges_, plain_edges_;
    std::deque<VertexSE2> vertices_;
    SlamAlgorithm* algorithm_ = nullptr;
    std::vector<double> edge_chi2_;
    se2gpu_ba* h_ = nullptr;
    bool* stop_ = nullptr;
    bool verbose_ = false;
    se2gpu_ba_stats stats_{};
};

inline double EdgeHandle::chi2() const { return opt->edgeChi2(index); }
inline void EdgeHandle::setLevel(int l) { level_ = l; if (opt) opt->noteLevelChange(); }
inline double SlamAlgorithm::currentLambda() const { return opt_ ? opt_->currentLambda() : 0.0; }

inline void initOptimizer(SlamOptimizer& opt, bool verbose = false) { opt.setVerbose(verbose); }

// CamPara* addCamPara(SlamOptimizer&, const cv::Mat& K, int id) (optimizer.h:85, optimizer.cpp:207-215): K is the 3x3
// CV_32F camera matrix, read as K(0,0), K(0,2), K(1,2) - any matrix with at<float>(r, c) (MatF, cv::Mat) fits.
template <typename MatT>
inline CamPara* addCamPara(SlamOptimizer& opt, const MatT& K, int id) {
    const double f = (double)K.template at<float>(0, 0);
    const double cx = (double)K.template at<float>(0, 2), cy = (double)K.template at<float>(1, 2);
    check(se2gpu_ba_add_cam(opt.handle(), f, cx, cy), "addCamPara");
    CamPara* c = opt.newCamPara();
    c->focal_length = f; c->principle_point[0] = cx; c->principle_point[1] = cy; c->id = id;
    return c;
}

inline VertexSE2* addVertexSE2(SlamOptimizer& opt, const SE2& pose, int id, bool fixed = false) {
    check(se2gpu_ba_add_vertex_se2(opt.handle(), id, pose.x, pose.y, pose.theta, fixed), "addVertexSE2");
    return opt.newVertexSE2(id);
}

inline void addVertexSBAXYZ(SlamOptimizer& opt, const Vector3D& xyz, int id, bool marginal = true, bool fixed = false) {
    check(se2gpu_ba_add_vertex_xyz(opt.handle(), id, xyz.v, marginal, fixed), "addVertexSBAXYZ");
}

// campara and _Tbc are graph-wide in the reference (one CamPara, Config::bTc on every edge, Map.cpp:1048-1049)
inline EdgeSE2XYZ* addEdgeSE2XYZ(SlamOptimizer& opt, const Vector2D& meas, int id0, int id1, CamPara* /*campara*/,
                                 const SE3Quat& _Tbc, const Matrix2D& info, double thHuber) {
    check(se2gpu_ba_set_Tbc(opt.handle(), _Tbc.R, _Tbc.t), "addEdgeSE2XYZ(Tbc)");
    check(se2gpu_ba_add_edge_se2xyz(opt.handle(), id0, id1, meas.v, info.m, thHuber), "addEdgeSE2XYZ");
    return opt.newPlainEdge();
}

inline PreEdgeSE2* addEdgeSE2(SlamOptimizer& opt, const Vector3D& meas, int id0, int id1, const Matrix3D& info) {
    check(se2gpu_ba_add_edge_se2(opt.handle(), id0, id1, meas.v, info.m), "addEdgeSE2");
    return opt.newPlainEdge();
}

// Map::loadLocalGraph's per-observation information Sigma_all.inverse() (src/Map.cpp:1024-1049) for a whole window in
// one device pass: lc = KeyFrame::mViewMPs[ftrIdx], lw = MapPoint::getPos(), Rcw / twb_xy per key frame, sigma2 =
// mvLevelSigma2[octave]; info_out[k] is ready for addEdgeSE2XYZ.
inline void computeEdgeInformation(int E, const float* lc, const float* lw, const int32_t* e_kf, const float* sigma2, int P,
                                   const float* Rcw, const float* twb_xy, float fx, float xrotInfo, float zInfo,
                                   std::vector<Matrix2D>& info_out) {
    info_out.resize(E);
    check(se2gpu_ba_edge_information(E, lc, lw, e_kf, sigma2, P, Rcw, twb_xy, fx, xrotInfo, zInfo,
                                     E ? info_out[0].m : nullptr), "computeEdgeInformation");
}

#if defined(SE2LAM_AMD_HAVE_G2O) && defined(SE2LAM_AMD_HAVE_EIGEN)
// the reference's own argument types, converted explicitly (conversions.h)
inline VertexSE2* addVertexSE2(SlamOptimizer& opt, const g2o::SE2& pose, int id, bool fixed = false) { return addVertexSE2(opt, mirror(pose), id, fixed); }
inline void addVertexSBAXYZ(SlamOptimizer& opt, const Eigen::Vector3d& xyz, int id, bool marginal = true, bool fixed = false) {
    addVertexSBAXYZ(opt, mirror(xyz), id, marginal, fixed);
}
inline EdgeSE2XYZ* addEdgeSE2XYZ(SlamOptimizer& opt, const Eigen::Vector2d& meas, int id0, int id1, CamPara* campara,
                                 const g2o::SE3Quat& _Tbc, const Eigen::Matrix2d& info, double thHuber) {
    return addEdgeSE2XYZ(opt, mirror(meas), id0, id1, campara, mirror(_Tbc), mirror(info), thHuber);
}
inline PreEdgeSE2* addEdgeSE2(SlamOptimizer& opt, const Eigen::Vector3d& meas, int id0, int id1, const Eigen::Matrix3d& info) {
    return addEdgeSE2(opt, mirror(meas), id0, id1, mirror(info));
}
#endif

// ---- SE3-expmap graphs: Map::loadLocalGraph(optimizer, vpEdgesAll, vnAllIdx) (Map.cpp:414-566) --------------------------
inline void pose12Of(const SE3Quat& q, double p[12]) {
    for (int i = 0; i < 9; ++i) p[i] = q.R[i];
    for (int i = 0; i < 3; ++i) p[9 + i] = q.t[i];
}
inline SE3Quat se3QuatOf(const double p[12]) {
    SE3Quat q;
    for (int i = 0; i < 9; ++i) q.R[i] = p[i];
    for (int i = 0; i < 3; ++i) q.t[i] = p[9 + i];
    return q;
}
inline void addVertexSE3Expmap(SlamOptimizer& opt, const SE3Quat& pose, int id, bool fixed = false) {   // optimizer.h:88
    double p[12];
    pose12Of(pose, p);
    check(se2gpu_ba_add_vertex_se3(opt.handle(), id, p, fixed), "addVertexSE3Expmap");
}
// EdgeSE3ExpmapPrior* addPlaneMotionSE3Expmap(opt, pose, vId, extPara) (optimizer.h:82): extPara = Config::bTc (4x4 CV_32F);
// the Config::PLANEMOTION_* weights are arguments here (defaults of src/Config.cpp:46-48)
template <typename MatT>
inline void addPlaneMotionSE3Expmap(SlamOptimizer& opt, const SE3Quat& pose, int vId, const MatT& extPara,
                                    double xrotInfo = 1e6, double yrotInfo = 1e6, double zInfo = 1) {
    double p[12], b[12], meas[12], info[36];
    pose12Of(pose, p);
    pose12Of(toSE3Quat(extPara), b);
    check(se2gpu_plane_motion_prior(p, b, xrotInfo, yrotInfo, zInfo, meas, info), "addPlaneMotionSE3Expmap");
    check(se2gpu_ba_add_prior_se3(opt.handle(), vId, meas, info), "addPlaneMotionSE3Expmap");
}
// addEdgeSE3Expmap (optimizer.h:94, optimizer.cpp:482-500): "The input info is [trans rot] order, but EdgeSE3Expmap
// requires [rot trans]" - the four 3x3 blocks are swapped exactly as the reference swaps them
inline void addEdgeSE3Expmap(SlamOptimizer& opt, const SE3Quat& measure, int id0, int id1, const Matrix6d& info) {
    Matrix6d n;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            n(r, c) = info(3 + r, 3 + c);
            n(3 + r, c) = info(r, 3 + c);
            n(r, 3 + c) = info(3 + r, c);
            n(3 + r, 3 + c) = info(r, c);
        }
    double p[12];
    pose12Of(measure, p);
    check(se2gpu_ba_add_edge_se3(opt.handle(), id0, id1, p, n.m), "addEdgeSE3Expmap");
}
// EdgeProjectXYZ2UV* addEdgeXYZ2UV(opt, measure, id0 (map point), id1 (key frame), paraId, info, thHuber) (optimizer.h:97)
inline EdgeProjectXYZ2UV* addEdgeXYZ2UV(SlamOptimizer& opt, const Vector2D& measure, int id0, int id1, int /*paraId*/,
                                        const Matrix2D& info, double thHuber) {
    check(se2gpu_ba_add_edge_xyz2uv(opt.handle(), id0, id1, measure.v, info.m[0], thHuber), "addEdgeXYZ2UV");
    return opt.newEdge();
}
inline SE3Quat estimateVertexSE3Expmap(SlamOptimizer& opt, int id) {                                      // optimizer.h:138
    double p[12];
    check(se2gpu_ba_get_se3(opt.handle(), id, p), "estimateVertexSE3Expmap");
    return se3QuatOf(p);
}

// ---- pose graphs: GlobalMapper::GlobalBA (GlobalMapper.cpp:328-535); Isometry3D is carried as rotation + translation ----
typedef SE3Quat Isometry3D;
inline void addParaSE3Offset(SlamOptimizer&, const Isometry3D&, int) {}       // optimizer.h:117: the identity offset GlobalBA uses
inline void addVertexSE3(SlamOptimizer& opt, const Isometry3D& pose, int id, bool fixed = false) {          // optimizer.h:120
    double p[12];
    pose12Of(pose, p);
    check(se2gpu_ba_add_vertex_iso3(opt.handle(), id, p, fixed), "addVertexSE3");
}
template <typename MatT>
inline void addVertexSE3PlaneMotion(SlamOptimizer& opt, const Isometry3D& pose, int id, const MatT& extPara,
                                    int /*paraSE3OffsetId*/, bool fixed = false, double xrotInfo = 1e6,
                                    double yrotInfo = 1e6, double zInfo = 1) {                              // optimizer.h:123
    addVertexSE3(opt, pose, id, fixed);
    double p[12], b[12], meas[12], info[36];
    pose12Of(pose, p);
    pose12Of(toSE3Quat(extPara), b);
    check(se2gpu_plane_motion_prior_iso3(p, b, xrotInfo, yrotInfo, zInfo, meas, info), "addVertexSE3PlaneMotion");
    check(se2gpu_ba_add_prior_se3(opt.handle(), id, meas, info), "addVertexSE3PlaneMotion");
}
inline EdgeSE3* addEdgeSE3(SlamOptimizer& opt, const Isometry3D& measure, int id0, int id1, const Matrix6d& info) {   // :129
    double p[12];
    pose12Of(measure, p);
    check(se2gpu_ba_add_edge_se3(opt.handle(), id0, id1, p, info.m), "addEdgeSE3");
    return opt.newEdge();
}
inline Isometry3D estimateVertexSE3(SlamOptimizer& opt, int id) { return estimateVertexSE3Expmap(opt, id); }   // optimizer.h:135

inline SE2 estimateVertexSE2(SlamOptimizer& opt, int id) {
    double v[3];
    check(se2gpu_ba_get_se2(opt.handle(), id, v), "estimateVertexSE2");
    SE2 s; s.x = v[0]; s.y = v[1]; s.theta = v[2];
    return s;
}

inline SE2 VertexSE2::estimate() const { return estimateVertexSE2(*opt, id_); }

inline Vector3D estimateVertexSBAXYZ(SlamOptimizer& opt, int id) {
    Vector3D v;
    check(se2gpu_ba_get_xyz(opt.handle(), id, v.v), "estimateVertexSBAXYZ");
    return v;
}

}  // namespace se2lam_amd
