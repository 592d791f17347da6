// Device-side pieces of the SE(2)-XYZ bundle adjustment that more than one translation unit needs (csrc/ba.hip: the
// multi-launch solver; csrc/ba_window.hip: the one-workgroup-per-window solver): the controller block of the device-side
// Levenberg-Marquardt policy, the edge arithmetic of /root/reference/src/EdgeSE2XYZ.cpp:61-106 and
// include/se2lam/EdgeSE2XYZ.h:62-102, the 3 x 3 landmark factor.  Everything is inline / internal.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

#include "../../include/se2gpu.h"

namespace se2gpu {
namespace badev {

constexpr int kGroup = 8;       // lanes cooperating on one landmark
constexpr int kBlock = 256;
constexpr double kPi = 3.14159265358979323846;

struct CamDev {
    double fx, cx, cy;
    double Rcb[9];
    double tcb[3];
    double huber;
};

// Levenberg-Marquardt controller state, resident on the device: every kernel of an LM trial reads it (which buffer holds
// the estimate, the damping, whether this trial is a retry on the same linearisation, whether the run is over) and the
// last kernel of the trial (k_finalize / k_lm_decide) advances it with g2o's policy.  The host only enqueues trial
// "slots" and reads the block back once per optimize() call - no host round trip between trials.
constexpr int kMailSeq = 4;   // mailbox word of the device-side slot counter (BaCtl::seq)
struct BaCtl {
    double lambda, ni, current_chi, rho;
    double chi2_init, chi2_final;
    int it, qmax, trials, iters;
    int done, terminated, stopped, retry;
    int sel;          // 0: the estimate lives in the "a" buffers (trial state in "b"), 1: the other way round
    int mode, error, pad;
    double chi2_hist[64], lambda_hist[64];
    int trials_hist[64];
    // survive k_ctl_init (and, like sel, say something about the handle rather than about one run):
    double seq;       // trial slots finished so far; posted next to the block (mail[kMailSeq]) - the host mirrors the count
    unsigned epoch;   // dense solves so far = the value the tile flags of k_chol_tiles are compared with
    unsigned pad2;
};

__host__ __device__ inline double normalize_theta(double theta) {
    if (theta >= -kPi && theta < kPi) return theta;
    double multiplier = floor(theta / (2 * kPi));
    theta = theta - multiplier * 2 * kPi;
    if (theta >= kPi) theta -= 2 * kPi;
    if (theta < -kPi) theta += 2 * kPi;
    return theta;
}

// Residual (and optionally the 2x3 pose / 2x3 landmark Jacobians) of one EdgeSE2XYZ.
// lc = Rcb Rz(-theta) (lw - [x,y,0]) + tcb ; e = f (X/Z, Y/Z) + c - z     (EdgeSE2XYZ.cpp:61-106)
template <bool JAC>
__device__ inline void se2xyz(const CamDev& cam, double px, double py, double pth, double lx, double ly, double lz,
                              double u, double v, double& e0, double& e1, double* Jp, double* Jl) {
    double s, c;
    sincos(pth, &s, &c);
    const double dx = lx - px, dy = ly - py;
    double R[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        R[i * 3 + 0] = cam.Rcb[i * 3 + 0] * c - cam.Rcb[i * 3 + 1] * s;
        R[i * 3 + 1] = cam.Rcb[i * 3 + 0] * s + cam.Rcb[i * 3 + 1] * c;
        R[i * 3 + 2] = cam.Rcb[i * 3 + 2];
    }
    const double X = R[0] * dx + R[1] * dy + R[2] * lz + cam.tcb[0];
    const double Y = R[3] * dx + R[4] * dy + R[5] * lz + cam.tcb[1];
    const double Z = R[6] * dx + R[7] * dy + R[8] * lz + cam.tcb[2];
    const double zi = 1.0 / Z;
    e0 = cam.fx * X * zi + cam.cx - u;
    e1 = cam.fx * Y * zi + cam.cy - v;
    if (JAC) {
        const double zi2 = zi * zi;
        const double j00 = cam.fx * zi, j02 = -cam.fx * X * zi2, j12 = -cam.fx * Y * zi2;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            Jl[k] = j00 * R[k] + j02 * R[6 + k];
            Jl[3 + k] = j00 * R[3 + k] + j12 * R[6 + k];
        }
        Jp[0] = -Jl[0]; Jp[1] = -Jl[1]; Jp[2] = Jl[0] * dy - Jl[1] * dx;
        Jp[3] = -Jl[3]; Jp[4] = -Jl[4]; Jp[5] = Jl[3] * dy - Jl[4] * dx;
    }
}

__device__ inline void huber(double e2, double delta, double& rho0, double& rho1) {  // RobustKernelHuber
    const double dsqr = delta * delta;
    if (e2 <= dsqr) {
        rho0 = e2; rho1 = 1.0;
    } else {
        const double sq = sqrt(e2);
        rho0 = 2 * sq * delta - dsqr;
        rho1 = delta / sq;
    }
}

// Butterfly sums on DPP moves (no trip through the LDS crossbar, no wait): the partner of the first two steps is lane ^ 1, lane ^ 2;
// from then on every lane of a quad (of eight, of sixteen) holds the same partial sum, so ANY lane of the other half is the partner
// of lane ^ 4 (lane ^ 8) - the mirrors 7 - lane and 15 - lane are such lanes.  Bit for bit what the shuffles by xor gave.
template <int CTRL>
__device__ __forceinline__ double dpp_partner(double v) {
    const long long b = __double_as_longlong(v);
    int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ inline double group_sum(double v) {  // sum over an aligned group of kGroup = 8 lanes
    static_assert(kGroup == 8, "three butterfly steps");
    v += dpp_partner<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_partner<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_partner<0x141>(v);   // row_half_mirror
    return v;
}

__device__ inline double wave_sum(double v) {
    v += dpp_partner<0xB1>(v);
    v += dpp_partner<0x4E>(v);
    v += dpp_partner<0x141>(v);
    v += dpp_partner<0x140>(v);   // row_mirror
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// A = G^-1 for M = h + lambda I = G G^T
__device__ inline void chol_inv3(const double h[6], double lambda, double a[6]) {
    const double m00 = h[0] + lambda, m10 = h[1], m20 = h[2], m11 = h[3] + lambda, m21 = h[4], m22 = h[5] + lambda;
    // M is positive definite whenever lambda > 0; in Gauss-Newton mode (lambda = 0) a landmark without parallax makes it
    // singular to rounding, and a pivot that comes out at -1e-17 must give a huge finite step (as a cofactor inverse, and
    // g2o's, would), not a NaN: pivots are floored 30 orders of magnitude below the block's trace
    const double floor_ = 1e-30 * (m00 + m11 + m22);
    const double a00 = 1.0 / sqrt(fmax(m00, floor_));
    const double g10 = m10 * a00, g20 = m20 * a00;
    const double a11 = 1.0 / sqrt(fmax(m11 - g10 * g10, floor_));
    const double g21 = (m21 - g20 * g10) * a11;
    const double a22 = 1.0 / sqrt(fmax(m22 - g20 * g20 - g21 * g21, floor_));
    const double a10 = -(a11 * g10) * a00;             // rows of A G = I
    const double a21 = -(a22 * g21) * a11;
    const double a20 = -(a21 * g10 + a22 * g20) * a00;
    a[0] = a00; a[1] = a10; a[2] = a11; a[3] = a20; a[4] = a21; a[5] = a22;
}

__device__ inline void pre_se2(const double* pi, const double* pj, const double* z, double e[3], double A[9],
                               double B[9]) {
    double s, c;
    sincos(pi[2], &s, &c);
    const double rx = pj[0] - pi[0], ry = pj[1] - pi[1];
    e[0] = c * rx + s * ry - z[0];
    e[1] = -s * rx + c * ry - z[1];
    e[2] = pj[2] - pi[2] - z[2];
    const double qx = -ry, qy = rx;
#pragma unroll
    for (int i = 0; i < 9; ++i) { A[i] = 0; B[i] = 0; }
    A[0] = -c; A[1] = -s; A[3] = s; A[4] = -c;
    A[2] = -(c * qx + s * qy);
    A[5] = -(-s * qx + c * qy);
    A[8] = -1;
    B[0] = c; B[1] = s; B[3] = -s; B[4] = c; B[8] = 1;
}

// One step of g2o's OptimizationAlgorithmLevenberg::solve / OptimizationAlgorithmGaussNewton on the controller block, run
// by ONE thread after the (all-reduced) scalars of a trial are known:  sc = {chi2 of the trial state, computeScale()
// denominator, factorisation flag}.  Mirrors, statement for statement, the host loop this replaces
// (rho = (chi - chi_trial) / (scale + 1e-3); accept: lambda *= max(1/3, min(1 - (2 rho - 1)^3, 2/3)), ni = 2, the trial
// state becomes the estimate (sel flips = discardTop); reject: lambda *= ni, ni *= 2 (pop); at most 10 trials per
// iteration; Terminate when all 10 failed or rho == 0).  `stop` is the caller's force-stop flag, mirrored by the host
// into mapped memory (SparseOptimizer::setForceStopFlag).
__device__ inline void lm_advance(BaCtl* c, const double* sc, bool stopped) {
    double tempChi = sc[0];
    const double scale_in = sc[1], fail = sc[2];
    if (fail >= 1e5) { c->error = 1; c->done = 1; return; }   // a dataflow spin of k_chol_tiles timed out
    if (fail > 0.0) tempChi = 1.7976931348623157e308;          // factorisation failed: the step is rejected
    c->trials += 1;
    const int qmax = c->qmax + 1;
    c->qmax = qmax;
    double rho;
    if (c->mode == SE2GPU_BA_GN) {
        c->sel ^= 1;
        c->current_chi = tempChi;
        rho = 1;
    } else {
        rho = (c->current_chi - tempChi) / (scale_in + 1e-3);
        if (rho > 0 && tempChi < 1.7976931348623157e308 && tempChi == tempChi) {
            const double t = 2 * rho - 1;
            double alpha = 1. - t * t * t;
            alpha = fmin(alpha, 2. / 3.);
            c->lambda *= fmax(1. / 3., alpha);
            c->ni = 2;
            c->current_chi = tempChi;
            c->sel ^= 1;
        } else {
            c->lambda *= c->ni;
            c->ni *= 2;
        }
        if (rho < 0 && qmax < 10 && !stopped) {   // do { ... } while (rho < 0 && qmax < 10 && !terminate())
            c->rho = rho;
            c->retry = 1;
            return;
        }
    }
    c->rho = rho;
    const int it = c->it;
    if (it < 64) { c->chi2_hist[it] = c->current_chi; c->lambda_hist[it] = c->lambda; c->trials_hist[it] = qmax; }
    c->it = it + 1;
    c->chi2_final = c->current_chi;
    c->qmax = 0;
    c->retry = 0;
    if (c->mode == SE2GPU_BA_LM && (qmax == 10 || rho == 0)) { c->terminated = 1; c->done = 1; }
    if (stopped) { c->stopped = 1; c->done = 1; }
    if (c->it >= c->iters) c->done = 1;
}

}  // namespace badev
}  // namespace se2gpu
