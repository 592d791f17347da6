// DROP-IN BINDING - what `g2o::SparseOptimizer::optimize()` becomes in the drop-in build of the
// reference: the graph that the reference's own, unmodified Map::loadLocalGraph / src/optimizer.cpp built (VertexSE2,
// VertexSBAPointXYZ, EdgeSE2XYZ with its information, Huber width, camera and extrinsic, PreEdgeSE2) is replayed through the
// mirror's free functions (include/se2lam_amd/optimizer.h, the overloads that take the g2o / Eigen types: conversions.h) into
// libse2gpu, optimised there (se2gpu_ba_initialize / se2gpu_ba_optimize with the caller's force-stop flag, LocalMapper.cpp:246,
// 259-260), and the estimates are written back into the g2o vertices, where Map::optimizeLocalGraph (src/Map.cpp:754-783) reads
// them.  Also: cv::findFundamentalMat of Track::removeOutliers (src/Track.cpp:326) -> se2gpu_track_fundamental_mask.
// Nothing of oracle/ is compiled into, linked to or called from this file.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <chrono>

#include <opencv2/core/core.hpp>
#include <g2o_shim.hpp>

#define private public      // EdgeSE2XYZ keeps its extrinsic (Tbc) private; no reference file is touched
#include "EdgeSE2XYZ.h"
#undef private

#include "se2lam_amd/optimizer.h"

#if !defined(SE2LAM_AMD_HAVE_G2O) || !defined(SE2LAM_AMD_HAVE_EIGEN) || !defined(SE2LAM_AMD_HAVE_OPENCV)
#error "conversions.h must see the OpenCV / Eigen / g2o headers in this build"
#endif

namespace {
struct LastBA { double v[10]; };
LastBA g_last{};
double g_ms[4] = {0, 0, 0, 0};   // the last forward(): replay in, initialize, optimize, write back (ms) - a probe's aid (SE2_DROPIN_TIMES)
inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
se2gpu_track* g_track = nullptr;

// returns false when the graph is not the SE(2)-XYZ local window (other optimisations of the reference are not driven here)
bool forward(g2o::SparseOptimizer& opt, int iterations) {
    namespace amd = se2lam_amd;
    for (const auto& kv : opt.vertices())
        if (!dynamic_cast<const g2o::VertexSE2*>(kv.second) && !dynamic_cast<const g2o::VertexSBAPointXYZ*>(kv.second)) return false;
    const double t0 = now_ms();
    amd::SlamOptimizer dev;   // se2gpu_ba_create: the handle comes from the library's pool (LocalMapper.cpp:239 builds one per localBA)
    dev.setVerbose(opt.verbose());
    dev.setForceStopFlag(opt.forceStopFlag());
    amd::CamPara* cam = nullptr;
    for (const g2o::Parameter* p : opt.parameters())
        if (const g2o::CameraParameters* c = dynamic_cast<const g2o::CameraParameters*>(p)) {
            cv::Mat K = cv::Mat::eye(3, 3, CV_32FC1);   // addCamPara takes the camera matrix (optimizer.cpp:207-215 reads K(0,0), K(0,2), K(1,2))
            K.at<float>(0, 0) = (float)c->focal_length; K.at<float>(1, 1) = (float)c->focal_length;
            K.at<float>(0, 2) = (float)c->principle_point[0]; K.at<float>(1, 2) = (float)c->principle_point[1];
            cam = amd::addCamPara(dev, K, c->id());
        }
    int P = 0, L = 0, E = 0, O = 0;
    for (const auto& kv : opt.vertices()) {
        if (const g2o::VertexSE2* v = dynamic_cast<const g2o::VertexSE2*>(kv.second)) {
            amd::addVertexSE2(dev, v->estimate(), kv.first, v->fixed());
            ++P;
        } else {
            const g2o::VertexSBAPointXYZ* x = static_cast<const g2o::VertexSBAPointXYZ*>(kv.second);
            amd::addVertexSBAXYZ(dev, x->estimate(), kv.first, x->marginalized(), x->fixed());
            ++L;
        }
    }
    for (g2o::OptimizableGraph::Edge* e : opt.edges()) {
        if (e->level() != opt.level()) continue;   // g2o optimises over the edges of the level handed to initializeOptimization
        if (g2o::EdgeSE2XYZ* x = dynamic_cast<g2o::EdgeSE2XYZ*>(e)) {
            if (!cam) throw std::runtime_error("drop-in optimize(): an EdgeSE2XYZ without camera parameters in the graph");
            const double delta = x->robustKernel() ? x->robustKernel()->delta() : 0.0;
            amd::addEdgeSE2XYZ(dev, x->measurement(), x->vertices()[0]->id(), x->vertices()[1]->id(), cam, x->Tbc, x->information(), delta);
            ++E;
        } else if (g2o::PreEdgeSE2* o = dynamic_cast<g2o::PreEdgeSE2*>(e)) {
            amd::addEdgeSE2(dev, o->measurement(), o->vertices()[0]->id(), o->vertices()[1]->id(), o->information());
            ++O;
        } else {
            return false;
        }
    }
    const double t1 = now_ms();
    dev.initializeOptimization(0);
    const double t2 = now_ms();
    const int done = dev.optimize(iterations);
    const double t3 = now_ms();
    for (const auto& kv : opt.vertices()) {
        if (kv.second->fixed()) continue;
        if (g2o::VertexSE2* v = dynamic_cast<g2o::VertexSE2*>(kv.second)) v->setEstimate(amd::toG2o(amd::estimateVertexSE2(dev, kv.first)));
        else static_cast<g2o::VertexSBAPointXYZ*>(kv.second)->setEstimate(amd::toEigen(amd::estimateVertexSBAXYZ(dev, kv.first)));
    }
    g_ms[0] = t1 - t0; g_ms[1] = t2 - t1; g_ms[2] = t3 - t2; g_ms[3] = now_ms() - t3;
    if (std::getenv("SE2_DROPIN_TIMES"))
        std::fprintf(stderr, "drop-in optimize(%d): replay in %.3f ms, initialize %.3f ms, optimize %.3f ms, write back %.3f ms (P %d L %d E %d)\n",
                     iterations, g_ms[0], g_ms[1], g_ms[2], g_ms[3], P, L, E);
    const se2gpu_ba_stats& s = dev.stats();
    const double rec[10] = {(double)P, (double)L, (double)E, (double)O, s.chi2_init, s.chi2_final, (double)done, (double)s.trials,
                            s.lambda_final, (double)s.stopped};
    std::memcpy(g_last.v, rec, sizeof(rec));
    return true;
}
}  // namespace

extern "C" {
const char* pipeline_kind(void) { return "dropin-gpu"; }
void pipeline_install_hooks(void) {
    g2o::SparseOptimizer::optimizeHook() = [](g2o::SparseOptimizer& opt, int iterations) { forward(opt, iterations); };
    if (!g_track) se2lam_amd::check(se2gpu_track_create(&g_track), "se2gpu_track_create");
    cv::shim_fundamental_hook() = [](const float* a, const float* b, int n, unsigned char* mask) {
        int inl = 0;
        se2lam_amd::check(se2gpu_track_fundamental_mask(g_track, a, b, n, mask, &inl), "se2gpu_track_fundamental_mask");
        if (const char* dump = std::getenv("SE2_DROPIN_DUMP")) {   // debugging aid: the call, for a replay through the other implementation
            static int call = 0;
            char name[512];
            std::snprintf(name, sizeof(name), "%s/fundamental_%d.bin", dump, call++);
            if (FILE* f = std::fopen(name, "wb")) {
                std::fwrite(&n, 4, 1, f);
                std::fwrite(a, 4, (size_t)2 * n, f);
                std::fwrite(b, 4, (size_t)2 * n, f);
                std::fwrite(mask, 1, (size_t)n, f);
                std::fclose(f);
            }
        }
        return inl;
    };
}
void pipeline_last_ba(double out[10]) { std::memcpy(out, g_last.v, sizeof(g_last.v)); }
}
