// Drop-in mirror of the reference's optimiser call surface over libse2gpu:
//   /root/reference/include/se2lam/optimizer.h:30-34  typedefs (SlamOptimizer = g2o::SparseOptimizer with
//        OptimizationAlgorithmLevenberg + BlockSolverX + a dense pose solve in place of LinearSolverCholmod)
//   :78 initOptimizer  :85 addCamPara  :91 addVertexSBAXYZ  :100 addEdgeSE2XYZ  :104 addVertexSE2
//   :107 estimateVertexSE2  :109 addEdgeSE2  :141 estimateVertexSBAXYZ
// and the g2o::SparseOptimizer methods LocalMapper::localBA / Map::loadLocalGraph call
//   (/root/reference/src/LocalMapper.cpp:239-260, src/Map.cpp:891-1053).
// With these in scope, Map::loadLocalGraph(SlamOptimizer&) and LocalMapper::localBA compile unchanged apart from the
// POD value types of types.h.  Unlike g2o the free functions return void: the raw
// vertex / edge pointers g2o returns are never used by the reference's hot path.
#pragma once
#include "types.h"

namespace se2lam_amd {

struct CamPara { double focal_length = 0; double principle_point[2] = {0, 0}; };  // g2o::CameraParameters

class SlamOptimizer {  // g2o::SparseOptimizer (the subset the hot path uses)
public:
    SlamOptimizer() { check(se2gpu_ba_create(&h_), "SlamOptimizer"); }
    ~SlamOptimizer() { se2gpu_ba_destroy(h_); }
    SlamOptimizer(const SlamOptimizer&) = delete;
    SlamOptimizer& operator=(const SlamOptimizer&) = delete;

    void setVerbose(bool v) { verbose_ = v; }
    void setForceStopFlag(bool* flag) { stop_ = flag; }                   // LocalMapper.cpp:246
    bool initializeOptimization(int level = 0) {                          // LocalMapper.cpp:259
        (void)level;
        check(se2gpu_ba_initialize(h_), "initializeOptimization");
        return true;
    }
    int optimize(int iterations) {                                        // LocalMapper.cpp:260
        static_assert(sizeof(bool) == 1, "the stop flag is polled as a byte");
        check(se2gpu_ba_optimize(h_, iterations, SE2GPU_BA_LM, reinterpret_cast<const volatile uint8_t*>(stop_),
                                 verbose_ ? 1 : 0, &stats_), "optimize");
        return stats_.iterations;
    }
    void clear() { check(se2gpu_ba_clear(h_), "clear"); }
    void clearParameters() {}
    double activeRobustChi2() { return se2gpu_ba_chi2(h_); }
    double currentLambda() const { return stats_.lambda_final; }          // REJECT_IF_LARGE_LAMBDA, LocalMapper.cpp:285-292
    const se2gpu_ba_stats& stats() const { return stats_; }
    se2gpu_ba* handle() { return h_; }

private:
    se2gpu_ba* h_ = nullptr;
    bool* stop_ = nullptr;
    bool verbose_ = false;
    se2gpu_ba_stats stats_{};
};

inline void initOptimizer(SlamOptimizer& opt, bool verbose = false) { opt.setVerbose(verbose); }

// K = (fx, cx, cy) as the reference reads K(0,0), K(0,2), K(1,2) (optimizer.cpp:207-215)
inline CamPara addCamPara(SlamOptimizer& opt, float fx, float cx, float cy, int id) {
    (void)id;
    check(se2gpu_ba_add_cam(opt.handle(), fx, cx, cy), "addCamPara");
    CamPara c; c.focal_length = fx; c.principle_point[0] = cx; c.principle_point[1] = cy;
    return c;
}

inline void addVertexSE2(SlamOptimizer& opt, const SE2& pose, int id, bool fixed = false) {
    check(se2gpu_ba_add_vertex_se2(opt.handle(), id, pose.x, pose.y, pose.theta, fixed), "addVertexSE2");
}

inline void addVertexSBAXYZ(SlamOptimizer& opt, const Vector3D& xyz, int id, bool marginal = true, bool fixed = false) {
    check(se2gpu_ba_add_vertex_xyz(opt.handle(), id, xyz.v, marginal, fixed), "addVertexSBAXYZ");
}

// campara and _Tbc are graph-wide in the reference (one CamPara, Config::bTc on every edge, Map.cpp:1048-1049)
inline void addEdgeSE2XYZ(SlamOptimizer& opt, const Vector2D& meas, int id0, int id1, const CamPara* /*campara*/,
                          const SE3Quat& _Tbc, const Matrix2D& info, double thHuber) {
    check(se2gpu_ba_set_Tbc(opt.handle(), _Tbc.R, _Tbc.t), "addEdgeSE2XYZ(Tbc)");
    check(se2gpu_ba_add_edge_se2xyz(opt.handle(), id0, id1, meas.v, info.m, thHuber), "addEdgeSE2XYZ");
}

inline void addEdgeSE2(SlamOptimizer& opt, const Vector3D& meas, int id0, int id1, const Matrix3D& info) {
    check(se2gpu_ba_add_edge_se2(opt.handle(), id0, id1, meas.v, info.m), "addEdgeSE2");
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

inline SE2 estimateVertexSE2(SlamOptimizer& opt, int id) {
    double v[3];
    check(se2gpu_ba_get_se2(opt.handle(), id, v), "estimateVertexSE2");
    SE2 s; s.x = v[0]; s.y = v[1]; s.theta = v[2];
    return s;
}

inline Vector3D estimateVertexSBAXYZ(SlamOptimizer& opt, int id) {
    Vector3D v;
    check(se2gpu_ba_get_xyz(opt.handle(), id, v.v), "estimateVertexSBAXYZ");
    return v;
}

}  // namespace se2lam_amd
