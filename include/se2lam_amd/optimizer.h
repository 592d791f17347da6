// Drop-in mirror of the reference's optimiser call surface over libse2gpu:
//   /root/reference/include/se2lam/optimizer.h:30-34  typedefs (SlamOptimizer = g2o::SparseOptimizer with
//        OptimizationAlgorithmLevenberg + BlockSolverX + a dense pose solve in place of LinearSolverCholmod)
//   :78 initOptimizer  :85 addCamPara  :91 addVertexSBAXYZ  :100 addEdgeSE2XYZ  :104 addVertexSE2
//   :107 estimateVertexSE2  :109 addEdgeSE2  :141 estimateVertexSBAXYZ
// and the g2o::SparseOptimizer methods LocalMapper::localBA / Map::loadLocalGraph call
//   (/root/reference/src/LocalMapper.cpp:239-260, src/Map.cpp:891-1053).
// Signatures follow the reference's: tests/cpp_reference_call_lines.cpp pastes the call lines of Map.cpp:897, 928-929,
// 951, 966-970, 989, 1048-1049 and LocalMapper.cpp:239-260 against this header.  The free functions that return raw
// g2o vertex / edge pointers in the reference (never used by its hot path) return void here; addCamPara returns a
// stable CamPara* as the reference does (Map.cpp:897).  With g2o / Eigen / OpenCV headers present, overloads taking
// the real types are compiled in (conversions.h).
#pragma once
#include <deque>

#include "types.h"

namespace se2lam_amd {

struct CamPara {   // g2o::CameraParameters (focal_length, principle_point, baseline 0) + its parameter id
    double focal_length = 0;
    double principle_point[2] = {0, 0};
    int id = 0;
};

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
    CamPara* newCamPara() { cams_.emplace_back(); return &cams_.back(); }   // owned by the optimizer, address-stable

private:
    std::deque<CamPara> cams_;
    se2gpu_ba* h_ = nullptr;
    bool* stop_ = nullptr;
    bool verbose_ = false;
    se2gpu_ba_stats stats_{};
};

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

inline void addVertexSE2(SlamOptimizer& opt, const SE2& pose, int id, bool fixed = false) {
    check(se2gpu_ba_add_vertex_se2(opt.handle(), id, pose.x, pose.y, pose.theta, fixed), "addVertexSE2");
}

inline void addVertexSBAXYZ(SlamOptimizer& opt, const Vector3D& xyz, int id, bool marginal = true, bool fixed = false) {
    check(se2gpu_ba_add_vertex_xyz(opt.handle(), id, xyz.v, marginal, fixed), "addVertexSBAXYZ");
}

// campara and _Tbc are graph-wide in the reference (one CamPara, Config::bTc on every edge, Map.cpp:1048-1049)
inline void addEdgeSE2XYZ(SlamOptimizer& opt, const Vector2D& meas, int id0, int id1, CamPara* /*campara*/,
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

#if defined(SE2LAM_AMD_HAVE_G2O) && defined(SE2LAM_AMD_HAVE_EIGEN)
// the reference's own argument types, converted explicitly (conversions.h)
inline void addVertexSE2(SlamOptimizer& opt, const g2o::SE2& pose, int id, bool fixed = false) { addVertexSE2(opt, mirror(pose), id, fixed); }
inline void addVertexSBAXYZ(SlamOptimizer& opt, const Eigen::Vector3d& xyz, int id, bool marginal = true, bool fixed = false) {
    addVertexSBAXYZ(opt, mirror(xyz), id, marginal, fixed);
}
inline void addEdgeSE2XYZ(SlamOptimizer& opt, const Eigen::Vector2d& meas, int id0, int id1, CamPara* campara,
                          const g2o::SE3Quat& _Tbc, const Eigen::Matrix2d& info, double thHuber) {
    addEdgeSE2XYZ(opt, mirror(meas), id0, id1, campara, mirror(_Tbc), mirror(info), thHuber);
}
inline void addEdgeSE2(SlamOptimizer& opt, const Eigen::Vector3d& meas, int id0, int id1, const Eigen::Matrix3d& info) {
    addEdgeSE2(opt, mirror(meas), id0, id1, mirror(info));
}
#endif

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
