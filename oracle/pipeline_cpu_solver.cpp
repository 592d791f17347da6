// ORACLE - TEST INFRASTRUCTURE ONLY (oracle/_ref/libse2lam_pipeline_cpu.so).  Never linked, imported or called by the product path.
//
// The CPU build of the reference's pipeline (`make -C oracle pipeline`): all of the reference's sources, its own ORBextractor.cpp and
// ORBmatcher.cpp included, compiled where they lie over oracle/_shim.  Two things the image lacks are third-party libraries,
// not se2lam code, and are supplied from the oracle's restatements here:
//   g2o::SparseOptimizer::optimize()  -> ba_ref_optimize   (oracle/ba_ref.cpp: g2o 20160424's Levenberg policy over the Schur complement)
//   cv::findFundamentalMat            -> match_ref_fundamental_mask (oracle/match_ref.cpp: OpenCV 3.2's FM_RANSAC)
// The drop-in build (tests/dropin/g2o_forward.cpp) has libse2gpu in both places; tests/test_dropin_pipeline.py compares the two runs.
#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <vector>

#include <opencv2/core/core.hpp>
#include <g2o_shim.hpp>

#define private public      // EdgeSE2XYZ keeps its extrinsic (Tbc) private; no reference file is touched
#include "EdgeSE2XYZ.h"
#undef private

extern "C" {
struct ba_ref_problem {   // oracle/ba_ref.cpp
    int32_t P, L, E, O;
    const double* poses; const uint8_t* fixed; const double* lms;
    const int32_t* e_kf; const int32_t* e_lm; const double* e_uv; const double* e_info;
    const int32_t* o_i; const int32_t* o_j; const double* o_meas; const double* o_info;
    double fx, cx, cy;
    double Rbc[9];
    double tbc[3];
    double huber;
};
struct ba_ref_stats {
    int32_t iterations, trials, terminated;
    double chi2_init, chi2_final, lambda_final;
    double chi2_hist[64], lambda_hist[64];
    int32_t trials_hist[64];
    int32_t n_rho;
    double rho_log[256];
};
int ba_ref_optimize(const ba_ref_problem* p, int iters, int mode, const volatile uint8_t* stop_flag, double* poses_out, double* lms_out,
                    ba_ref_stats* stats);
int match_ref_fundamental_mask(const float* m1, const float* m2, int n, uint8_t* mask);
}

namespace {
double g_last[10];

bool solve(g2o::SparseOptimizer& opt, int iterations) {
    std::map<int, int> pose_of, lm_of;
    std::vector<double> poses, lms;
    std::vector<uint8_t> fixed;
    std::vector<g2o::VertexSE2*> vp;
    std::vector<g2o::VertexSBAPointXYZ*> vl;
    for (const auto& kv : opt.vertices()) {
        if (g2o::VertexSE2* v = dynamic_cast<g2o::VertexSE2*>(kv.second)) {
            pose_of[kv.first] = (int)vp.size();
            vp.push_back(v);
            const g2o::Vector3D e = v->estimate().toVector();
            poses.insert(poses.end(), {e[0], e[1], e[2]});
            fixed.push_back(v->fixed() ? 1 : 0);
        } else if (g2o::VertexSBAPointXYZ* x = dynamic_cast<g2o::VertexSBAPointXYZ*>(kv.second)) {
            if (x->fixed() || !x->marginalized()) throw std::runtime_error("pipeline (CPU): map points are free and marginalised in the local window");
            lm_of[kv.first] = (int)vl.size();
            vl.push_back(x);
            lms.insert(lms.end(), {x->estimate()[0], x->estimate()[1], x->estimate()[2]});
        } else {
            return false;
        }
    }
    ba_ref_problem p{};
    std::vector<int32_t> e_kf, e_lm, o_i, o_j;
    std::vector<double> e_uv, e_info, o_meas, o_info;
    bool have_edge = false;
    for (g2o::OptimizableGraph::Edge* e : opt.edges()) {
        if (e->level() != opt.level()) continue;
        if (g2o::EdgeSE2XYZ* x = dynamic_cast<g2o::EdgeSE2XYZ*>(e)) {
            e_kf.push_back(pose_of.at(x->vertices()[0]->id()));
            e_lm.push_back(lm_of.at(x->vertices()[1]->id()));
            e_uv.insert(e_uv.end(), {x->measurement()[0], x->measurement()[1]});
            e_info.insert(e_info.end(), {x->information()(0, 0), x->information()(0, 1), x->information()(1, 1)});
            const double delta = x->robustKernel() ? x->robustKernel()->delta() : 0.0;
            const Eigen::Matrix3d R = x->Tbc.rotation().toRotationMatrix();
            if (!have_edge) {
                have_edge = true;
                p.huber = delta;
                p.fx = x->cam->focal_length; p.cx = x->cam->principle_point[0]; p.cy = x->cam->principle_point[1];
                for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) p.Rbc[3 * r + c] = R(r, c); p.tbc[r] = x->Tbc.translation()[r]; }
            } else if (delta != p.huber) {
                throw std::runtime_error("pipeline (CPU): one Huber width per window (Config::TH_HUBER)");
            }
        } else if (g2o::PreEdgeSE2* o = dynamic_cast<g2o::PreEdgeSE2*>(e)) {
            o_i.push_back(pose_of.at(o->vertices()[0]->id()));
            o_j.push_back(pose_of.at(o->vertices()[1]->id()));
            for (int i = 0; i < 3; ++i) o_meas.push_back(o->measurement()[i]);
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) o_info.push_back(o->information()(r, c));
        } else {
            return false;
        }
    }
    p.P = (int)vp.size(); p.L = (int)vl.size(); p.E = (int)e_kf.size(); p.O = (int)o_i.size();
    p.poses = poses.data(); p.fixed = fixed.data(); p.lms = lms.data();
    p.e_kf = e_kf.data(); p.e_lm = e_lm.data(); p.e_uv = e_uv.data(); p.e_info = e_info.data();
    p.o_i = o_i.data(); p.o_j = o_j.data(); p.o_meas = o_meas.data(); p.o_info = o_info.data();
    std::vector<double> po(poses.size()), lo(lms.size());
    ba_ref_stats st;
    std::memset(&st, 0, sizeof(st));
    static_assert(sizeof(bool) == 1, "the force-stop flag is polled as a byte");
    ba_ref_optimize(&p, iterations, 0, reinterpret_cast<const volatile uint8_t*>(opt.forceStopFlag()), po.data(), lo.data(), &st);
    for (size_t i = 0; i < vp.size(); ++i)
        if (!vp[i]->fixed()) vp[i]->setEstimate(g2o::SE2(po[3 * i], po[3 * i + 1], po[3 * i + 2]));
    for (size_t i = 0; i < vl.size(); ++i) vl[i]->setEstimate(g2o::Vector3D(lo[3 * i], lo[3 * i + 1], lo[3 * i + 2]));
    const double rec[10] = {(double)p.P, (double)p.L, (double)p.E, (double)p.O, st.chi2_init, st.chi2_final, (double)st.iterations,
                            (double)st.trials, st.lambda_final, 0.0};
    std::memcpy(g_last, rec, sizeof(rec));
    return true;
}
}  // namespace

extern "C" {
const char* pipeline_kind(void) { return "reference-cpu"; }
void pipeline_install_hooks(void) {
    g2o::SparseOptimizer::optimizeHook() = [](g2o::SparseOptimizer& opt, int iterations) { solve(opt, iterations); };
    cv::shim_fundamental_hook() = [](const float* a, const float* b, int n, unsigned char* mask) { return match_ref_fundamental_mask(a, b, n, mask); };
}
void pipeline_last_ba(double out[10]) { std::memcpy(out, g_last, sizeof(g_last)); }
}
