// Localizer::DoLocalBA (/root/reference/src/Localizer.cpp:233-302) - SURVEY.md 8(f).2: pose-only bundle adjustment of
// the current key frame against the FIXED map points it observes.
//   graph   one VertexSE3Expmap (Tcw), one EdgeProjectXYZ2UV per observation (information invSigma2 * I, Huber with
//           delta = Config::TH_HUBER), one EdgeSE3ExpmapPrior built by addPlaneMotionSE3Expmap
//           (/root/reference/src/optimizer.cpp:236-314, 159-197); Levenberg-Marquardt, optimize(30)
//   [3P g2o 20160424]  EdgeProjectXYZ2UV::computeError / linearizeOplus and VertexSE3Expmap::oplusImpl
//           (types_six_dof_expmap), SE3Quat::exp / log / adj (se3quat.h), OptimizationAlgorithmLevenberg - the same
//           policy as the SE(2) bundle adjustment in ba.hip
// The system has six unknowns, so the whole optimize(30) is ONE launch of one workgroup: the threads share the edges
// (normal equations and robust chi2 by wave shuffles + LDS), thread 0 adds the prior, solves the damped 6x6 system
// (LL^T), applies exp(update) * estimate and runs the accept / reject logic; the key frame's pose and the statistics
// come back in one download.  Rotations are matrices, passed through a unit quaternion after every product and
// exponential the way SE3Quat renormalises its rotation.
#include "track_ws.h"
#include "se3_math.h"

#include <cmath>
#include <limits>

namespace se2gpu {
namespace {

struct PoseParams {
    Se3 est0, prior;
    double info[36];     // information of the prior, row-major, vector order (rotation, translation)
    double f, cx, cy, delta;
    int n, iters;
};

constexpr int kAcc = 28;   // 21 upper entries of H, 6 of b, chi2

// sum of v[0..cnt) over the workgroup, result in s_tot (valid for every thread after the call)
__device__ inline void block_sum(double* v, int cnt, double (*s_part)[kAcc], double* s_tot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = 0; k < cnt; ++k) {
        double x = v[k];
        for (int s = 1; s < 64; s <<= 1) x += __shfl_xor(x, s);
        if (lane == 0) s_part[wave][k] = x;
    }
    __syncthreads();
    if (threadIdx.x < cnt) s_tot[threadIdx.x] = s_part[0][threadIdx.x] + s_part[1][threadIdx.x] + s_part[2][threadIdx.x] + s_part[3][threadIdx.x];
    __syncthreads();
}

// EdgeProjectXYZ2UV at pose T for edge i: robust chi2 contribution, optionally the normal equations
template <bool LIN>
__device__ inline void edge_terms(const PoseParams& P, const Se3& T, const double* __restrict__ xyz,
                                  const double* __restrict__ uv, const double* __restrict__ w, int i, double* acc) {
    const double X = xyz[3 * i], Y = xyz[3 * i + 1], Z = xyz[3 * i + 2];
    const double x = T.R[0] * X + T.R[1] * Y + T.R[2] * Z + T.t[0];
    const double y = T.R[3] * X + T.R[4] * Y + T.R[5] * Z + T.t[1];
    const double z = T.R[6] * X + T.R[7] * Y + T.R[8] * Z + T.t[2];
    const double e0 = uv[2 * i] - (x / z * P.f + P.cx), e1 = uv[2 * i + 1] - (y / z * P.f + P.cy);
    const double wi = w[i];
    const double e2 = wi * (e0 * e0 + e1 * e1);
    double rho0, rho1;
    if (e2 <= P.delta * P.delta) { rho0 = e2; rho1 = 1; }
    else { const double sq = sqrt(e2); rho0 = 2 * sq * P.delta - P.delta * P.delta; rho1 = P.delta / sq; }
    acc[27] += rho0;
    if (!LIN) return;
    const double z2 = z * z, f = P.f;
    const double J0[6] = {x * y / z2 * f, -(1 + (x * x / z2)) * f, y / z * f, -1. / z * f, 0, x / z2 * f};
    const double J1[6] = {(1 + y * y / z2) * f, -x * y / z2 * f, -x / z * f, 0, -1. / z * f, y / z2 * f};
    const double ww = rho1 * wi;
    int k = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        acc[21 + a] += -ww * (J0[a] * e0 + J1[a] * e1);
#pragma unroll
        for (int c = a; c < 6; ++c) acc[k++] += ww * (J0[a] * J0[c] + J1[a] * J1[c]);
    }
}

// EdgeSE3ExpmapPrior: error = log(measurement * estimate^-1), Jacobian -I, no robust kernel
__device__ inline double prior_terms(const PoseParams& P, const Se3& T, double* H, double* b) {
    double ep[6];
    se3_log(se3_mul(P.prior, se3_inv(T)), ep);
    double chi = 0;
    #pragma unroll
    for (int a = 0; a < 6; ++a) {
        double s = 0;
        #pragma unroll
        for (int c = 0; c < 6; ++c) s += P.info[6 * a + c] * ep[c];
        chi += ep[a] * s;
        if (H) {
            b[a] += s;
            #pragma unroll
            for (int c = 0; c < 6; ++c) H[6 * a + c] += P.info[6 * a + c];
        }
    }
    return chi;
}

__device__ inline bool solve6(const double* H, double lambda, const double* b, double* x) {
    double L[36];
    #pragma unroll
    for (int i = 0; i < 6; ++i)
        #pragma unroll
        for (int j = 0; j < 6; ++j) L[6 * i + j] = H[6 * i + j] + (i == j ? lambda : 0.0);
    #pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = L[6 * j + j];
        #pragma unroll
        for (int k = 0; k < j; ++k) d -= L[6 * j + k] * L[6 * j + k];
        if (!(d > 0)) return false;
        d = sqrt(d);
        L[6 * j + j] = d;
        #pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double v = L[6 * i + j];
            #pragma unroll
            for (int k = 0; k < j; ++k) v -= L[6 * i + k] * L[6 * j + k];
            L[6 * i + j] = v / d;
        }
    }
    double y[6];
    #pragma unroll
    for (int i = 0; i < 6; ++i) {
        double v = b[i];
        #pragma unroll
        for (int k = 0; k < i; ++k) v -= L[6 * i + k] * y[k];
        y[i] = v / L[6 * i + i];
    }
    #pragma unroll
    for (int i = 5; i >= 0; --i) {
        double v = y[i];
        #pragma unroll
        for (int k = i + 1; k < 6; ++k) v -= L[6 * k + i] * x[k];
        x[i] = v / L[6 * i + i];
    }
    return true;
}

__global__ __launch_bounds__(256) void k_pose_ba(PoseParams P, const double* __restrict__ xyz, const double* __restrict__ uv,
                                                  const double* __restrict__ w, double* __restrict__ pose_out,
                                                  se2gpu_ba_stats* __restrict__ stats) {
    __shared__ Se3 s_est, s_trial;
    __shared__ double s_part[4][kAcc], s_tot[kAcc];
    __shared__ double s_pH[36], s_pb[6], s_pchi;   // prior terms, computed by a lane of the last wave beside the edge pass
    __shared__ int s_go, s_ok;
    const int tid = threadIdx.x;
    constexpr int kPriorThread = 192;
    if (tid == 0) s_est = P.est0;
    __syncthreads();
    double acc[kAcc];
    // thread 0 only
    double H[36], b[6], x[6], lambda = 0, ni = 2, currentChi = 0, rho = 0;
    int qmax = 0, trials = 0;
    {
        acc[27] = 0;
        const Se3 T = s_est;
        for (int i = tid; i < P.n; i += 256) edge_terms<false>(P, T, xyz, uv, w, i, acc);
        if (tid == kPriorThread) s_pchi = prior_terms(P, T, nullptr, nullptr);
        block_sum(acc + 27, 1, s_part, s_tot);
        if (tid == 0) {
            const double c0 = s_tot[0] + s_pchi;
            stats->chi2_init = c0;
            stats->chi2_final = c0;
            stats->iterations = 0;
            stats->terminated = 0;
            stats->stopped = 0;
        }
    }
    for (int it = 0; it < P.iters; ++it) {
        {   // linearise at the estimate
            #pragma unroll
            for (int k = 0; k < kAcc; ++k) acc[k] = 0;
            const Se3 T = s_est;
            for (int i = tid; i < P.n; i += 256) edge_terms<true>(P, T, xyz, uv, w, i, acc);
            if (tid == kPriorThread) {
                #pragma unroll
                for (int a = 0; a < 36; ++a) s_pH[a] = 0;
                #pragma unroll
                for (int a = 0; a < 6; ++a) s_pb[a] = 0;
                s_pchi = prior_terms(P, T, s_pH, s_pb);
            }
            block_sum(acc, kAcc, s_part, s_tot);
            if (tid == 0) {
                int k = 0;
                #pragma unroll
                for (int a = 0; a < 6; ++a) {
                    b[a] = s_tot[21 + a] + s_pb[a];
                    #pragma unroll
                    for (int c = a; c < 6; ++c) { H[6 * a + c] = s_tot[k]; H[6 * c + a] = s_tot[k]; ++k; }
                }
                #pragma unroll
                for (int a = 0; a < 36; ++a) H[a] += s_pH[a];
                currentChi = s_tot[27] + s_pchi;
                if (it == 0) {   // computeLambdaInit: tau * max |diag(H)|
                    double maxd = 0;
                    #pragma unroll
                    for (int r = 0; r < 6; ++r) maxd = fmax(fabs(H[7 * r]), maxd);
                    lambda = 1e-5 * maxd;
                    ni = 2;
                }
                rho = 0;
                qmax = 0;
            }
        }
        bool ok2 = true;
        do {
            if (tid == 0) {
                ok2 = solve6(H, lambda, b, x);
                if (!ok2)
                    #pragma unroll
                    for (int r = 0; r < 6; ++r) x[r] = 0;
                s_trial = se3_mul(se3_exp(x), s_est);
            }
            __syncthreads();
            acc[27] = 0;
            const Se3 T = s_trial;
            for (int i = tid; i < P.n; i += 256) edge_terms<false>(P, T, xyz, uv, w, i, acc);
            if (tid == kPriorThread) s_pchi = prior_terms(P, T, nullptr, nullptr);
            block_sum(acc + 27, 1, s_part, s_tot);
            if (tid == 0) {
                double tempChi = s_tot[0] + s_pchi;
                if (!ok2) tempChi = 1.7976931348623157e308;
                ++trials;
                ++qmax;
                rho = currentChi - tempChi;
                double scale = 1e-3;
                #pragma unroll
                for (int r = 0; r < 6; ++r) scale += x[r] * (lambda * x[r] + b[r]);
                rho /= scale;
                if (rho > 0 && isfinite(tempChi)) {
                    const double q = 2 * rho - 1;
                    double alpha = 1. - q * q * q;
                    alpha = fmin(alpha, 2. / 3.);
                    lambda *= fmax(1. / 3., alpha);
                    ni = 2;
                    currentChi = tempChi;
                    s_est = T;
                } else {
                    lambda *= ni;
                    ni *= 2;
                }
                s_go = rho < 0 && qmax < 10;
            }
            __syncthreads();
        } while (s_go);
        if (tid == 0) {
            if (it < 64) { stats->chi2_hist[it] = currentChi; stats->lambda_hist[it] = lambda; stats->trials_hist[it] = qmax; }
            stats->iterations = it + 1;
            stats->chi2_final = currentChi;
            const bool term = qmax == 10 || rho == 0;
            if (term) stats->terminated = 1;
            s_ok = !term;
        }
        __syncthreads();
        if (!s_ok) break;
    }
    if (tid == 0) {
        stats->trials = trials;
        stats->lambda_final = lambda;
        #pragma unroll
        for (int i = 0; i < 9; ++i) pose_out[i] = s_est.R[i];
        #pragma unroll
        for (int i = 0; i < 3; ++i) pose_out[9 + i] = s_est.t[i];
    }
}

Se3 se3_from(const double* p) {
    Se3 T;
    std::memcpy(T.R, p, 9 * sizeof(double));
    std::memcpy(T.t, p + 9, 3 * sizeof(double));
    return T;
}

}  // namespace
}  // namespace se2gpu

using namespace se2gpu;

// addPlaneMotionSE3Expmap (optimizer.cpp:236-314, the branch that is compiled): graph construction on the host.
extern "C" int se2gpu_plane_motion_prior(const double* Tcw12, const double* Tbc12, double xrot_info, double yrot_info,
                                         double z_info, double* meas12, double* info36) {
    SE2_REQUIRE(Tcw12 && Tbc12 && meas12 && info36, SE2GPU_ERR_INVALID, "plane_motion_prior: NULL argument");
    const Se3 Tcw = se3_from(Tcw12), Tbc = se3_from(Tbc12);
    Se3 Tbw = se3_mul(Tbc, Tcw);
    // Log_Rbw = angle * axis of Rbw as Eigen::AngleAxisd takes it from the QUATERNION (angle = 2 atan2(|q_xyz|, w)); only its
    // z component survives (:271-274), and the height is dropped (:276-278).  (Until round 4 this went through the logarithm
    // of the rotation matrix, which loses ten digits near a yaw of pi: 6e-7 mm in the measurement against the compiled
    // reference, found by tools/fuzz_ref_backend.py.)
    double q[4];
    quat_of(Tbw.R, q);
    const double nv = std::sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double yaw = nv > 0 ? 2 * std::atan2(nv, q[0]) * q[3] / nv : 0.0;
    const double c = std::cos(yaw), sn = std::sin(yaw);
    const double Rz[9] = {c, -sn, 0, sn, c, 0, 0, 0, 1};
    std::memcpy(Tbw.R, Rz, sizeof(Rz));
    Tbw.t[2] = 0;
    const Se3 meas = se3_mul(se3_inv(Tbc), Tbw);
    std::memcpy(meas12, meas.R, 9 * sizeof(double));
    std::memcpy(meas12 + 9, meas.t, 3 * sizeof(double));
    // Info_cw = adj(Tbc)^T diag(xrot, yrot, 1e-4, 1e-4, 1e-4, z) adj(Tbc);  SE3Quat::adj = [R 0; skew(t) R, R]
    double A[36] = {0}, sk[9], sR[9];
    skew3(Tbc.t, sk);
    mat3_mul(sk, Tbc.R, sR);
        for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
            A[6 * i + j] = Tbc.R[3 * i + j];
            A[6 * (i + 3) + (j + 3)] = Tbc.R[3 * i + j];
            A[6 * (i + 3) + j] = sR[3 * i + j];
        }
    const double D[6] = {xrot_info, yrot_info, 1e-4, 1e-4, 1e-4, z_info};
        for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 6; ++j) {
            double s = 0;
                        for (int k = 0; k < 6; ++k) s += A[6 * k + i] * D[k] * A[6 * k + j];
            info36[6 * i + j] = s;
        }
        for (int i = 0; i < 6; ++i)   // "make sure the info matrix is symmetric" (:296-298)
                for (int j = 0; j < i; ++j) info36[6 * i + j] = info36[6 * j + i];
    return SE2GPU_OK;
}

extern "C" int se2gpu_track_pose_ba(se2gpu_track* h, const double* Tcw12, const double* prior_meas12,
                                    const double* prior_info36, int n, const double* xyz, const double* uv,
                                    const double* inv_sigma2, double f, double cx, double cy, double huber_delta, int iters,
                                    double* Tcw_out12, se2gpu_ba_stats* stats) {
    SE2_REQUIRE(h && Tcw12 && prior_meas12 && prior_info36 && Tcw_out12, SE2GPU_ERR_INVALID, "pose_ba: NULL argument");
    SE2_REQUIRE(n >= 0 && iters >= 0, SE2GPU_ERR_INVALID, "pose_ba: negative size");
    SE2_REQUIRE(n == 0 || (xyz && uv && inv_sigma2), SE2GPU_ERR_INVALID, "pose_ba: NULL buffer");
    PoseParams P;
    P.est0 = se3_from(Tcw12);
    P.prior = se3_from(prior_meas12);
    std::memcpy(P.info, prior_info36, sizeof(P.info));
    P.f = f; P.cx = cx; P.cy = cy; P.delta = huber_delta;
    P.n = n; P.iters = iters;
    // one upload [xyz | uv | w], one download [pose (12) | stats]
    const size_t nn = (size_t)std::max(n, 1);
    const size_t in_b = nn * 6 * sizeof(double);
    const size_t out_b = 12 * sizeof(double) + sizeof(se2gpu_ba_stats);
    SE2_CHECK(h->h_in.reserve(in_b));
    SE2_CHECK(h->d_in.reserve(in_b));
    SE2_CHECK(h->h_out.reserve(out_b));
    SE2_CHECK(h->d_out.reserve(out_b));
    double* hi = (double*)h->h_in.p;
    if (n) {
        std::memcpy(hi, xyz, (size_t)n * 3 * sizeof(double));
        std::memcpy(hi + 3 * nn, uv, (size_t)n * 2 * sizeof(double));
        std::memcpy(hi + 5 * nn, inv_sigma2, (size_t)n * sizeof(double));
    }
    hipStream_t st = h->stream;
    SE2_HIP(hipMemcpyAsync(h->d_in.p, h->h_in.p, in_b, hipMemcpyHostToDevice, st));
    SE2_HIP(hipMemsetAsync(h->d_out.p, 0, out_b, st));
    const double* di = (const double*)h->d_in.p;
    hipLaunchKernelGGL(k_pose_ba, dim3(1), dim3(256), 0, st, P, di, di + 3 * nn, di + 5 * nn, (double*)h->d_out.p,
                       (se2gpu_ba_stats*)(h->d_out.p + 12 * sizeof(double)));
    SE2_HIP(hipGetLastError());
    SE2_HIP(hipMemcpyAsync(h->h_out.p, h->d_out.p, out_b, hipMemcpyDeviceToHost, st));
    SE2_HIP(hipStreamSynchronize(st));
    std::memcpy(Tcw_out12, h->h_out.p, 12 * sizeof(double));
    if (stats) std::memcpy(stats, h->h_out.p + 12 * sizeof(double), sizeof(se2gpu_ba_stats));
    return SE2GPU_OK;
}
