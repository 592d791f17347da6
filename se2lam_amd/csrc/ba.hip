// libse2gpu - SE(2)-XYZ local bundle adjustment on gfx950 (MI355X).
//
// Replaces the g2o::SparseOptimizer that LocalMapper::localBA builds and runs
//   /root/reference/src/LocalMapper.cpp:232-302      (driver: LM(BlockSolverX(Cholmod)), optimize(LOCAL_ITER))
//   /root/reference/src/Map.cpp:891-1053              (graph: VertexSE2 / VertexSBAPointXYZ / EdgeSE2XYZ / PreEdgeSE2)
//   /root/reference/src/EdgeSE2XYZ.cpp:61-106         (per-edge residual + 2x3 / 2x3 Jacobians)
//   /root/reference/include/se2lam/EdgeSE2XYZ.h:62-102 (PreEdgeSE2)
// with FP64 HIP kernels.  Design:
//   * SoA arena in HBM; observation edges are sorted by landmark (CSR) once per graph.
//   * k_linearize      8 lanes per landmark: residual, Jacobians, Huber weight, per-landmark Hll / bl via an in-group
//                      shuffle reduction (no atomics), then the whitened per-edge records W_e = Hpl_e G^-T and
//                      Dg_e = {Hpp_e - W_e W_e^T, bp_e, W_e zeta} with Hll + lambda I = G G^T (a retry runs it again).
//   * k_reduce2        the reduced system S|b_s in one launch from a PRECOMPUTED contributor plan (output-stationary:
//                      deterministic, atomic-free); k_pose_reduce / k_maxdiag only for lambda_0.
//   * k_chol_tiles     dense pose solve: LDL^T of the augmented (3P)^2 system as one dataflow launch over 32x32 tiles
//                      (k_chol_step: one launch per block column, second implementation); k_chol_apply: x = R y.
//   * k_update, k_finalize  back-substitution, oplus into a trial state, robust chi^2, the LM gain denominator; the
//                      scalars reach the host LM controller through a mapped mailbox.
//   * the LM controller mirrors g2o's OptimizationAlgorithmLevenberg (lambda policy, <= 10 trials, Terminate rule)
//     and polls the caller's stop flag between trials (SparseOptimizer::setForceStopFlag).
// Multi-GPU: landmarks are sharded; S|bs is summed over ranks by RCCL (se2gpu_comm_*) or the caller's callback.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <functional>
#include <limits>
#include <mutex>
#include <thread>
#include <tuple>
#include <memory>
#include <unordered_map>

#include "common.h"
#include "se3_math.h"
#include "ba_device.h"
#include "ba_window.h"

using namespace se2gpu;
using namespace se2gpu::badev;

namespace {


// ---------------------------------------------------------------------------------------------
// k_linearize: per landmark group (8 lanes per landmark, a lane takes every 8th edge of the landmark).
//
// Per-edge records in "whitened landmark coordinates" (round 4).  With M_l = Hll_l + lambda I = G G^T (3x3 Cholesky) and
// A = G^-1 (lower triangular), every product the Schur complement needs factors through W_e = Hpl_e A^T:
//     Y_e Hpl_j^T = Hpl_e M^-1 Hpl_j^T = W_e W_j^T,      Hpl_e z_l = W_e zeta_l  with  zeta_l = A bl_l,
//     x_l = z_l - sum_e Y_e^T dp_e = A^T (zeta_l - sum_e W_e^T dp_e).
// So ONE 72-byte operand per edge (W_e) replaces the two (Hpl_e and Y_e = Hpl_e M^-1) the pair products used to gather, and
// the un-reduced pose blocks Hpp_e / bp_e never reach memory on the common path: the first edge of a lane stays in
// registers between the landmark sum and the second pass (a landmark has ~6 observations, a lane almost never a second edge).
//   per edge:      W_e (3x3 row-major, 9)   and   Dg_e = { sym(Hpp_e - W_e W_e^T) (6), bp_e (3), W_e zeta_l (3) }   (12)
//   per landmark:  Hll (6 sym: xx xy xz yy yz zz), bl (3), A (6: a00 a10 a11 a20 a21 a22), zeta (3)
// 168 bytes written per edge instead of 312.  A rejected LM trial keeps its estimate, so its retry simply runs this kernel
// again with the new lambda (same inputs, same code: the linearisation comes out identical to the bit) - there is no
// lambda-only kernel and nothing lambda-independent has to be kept per edge.
// <FUSED = false> is the opening pass of an optimize() call only: lambda_0 = 1e-5 max diag H needs the diagonals first, so
// it writes Hll, bl and the un-reduced pose terms Hpp_e (6 sym), bp_e (3) for k_lambda0 / k_pose_reduce and no records.
// ---------------------------------------------------------------------------------------------

// one edge's whitened record from its un-reduced blocks: hh = Hpl_e (9), hp = Hpp_e (6 sym), bpe = bp_e (3)
__device__ inline void write_edge_record(double* __restrict__ w_out, double* __restrict__ dg, const double* __restrict__ hh,
                                         const double* __restrict__ hp, const double* __restrict__ bpe,
                                         const double a[6], const double zeta[3]) {
    double w[9];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double h0 = hh[r * 3], h1 = hh[r * 3 + 1], h2 = hh[r * 3 + 2];
        w[r * 3 + 0] = h0 * a[0];
        w[r * 3 + 1] = h0 * a[1] + h1 * a[2];
        w[r * 3 + 2] = h0 * a[3] + h1 * a[4] + h2 * a[5];
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) w_out[i] = w[i];
    dg[0] = hp[0] - (w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    dg[1] = hp[1] - (w[0] * w[3] + w[1] * w[4] + w[2] * w[5]);
    dg[2] = hp[2] - (w[0] * w[6] + w[1] * w[7] + w[2] * w[8]);
    dg[3] = hp[3] - (w[3] * w[3] + w[4] * w[4] + w[5] * w[5]);
    dg[4] = hp[4] - (w[3] * w[6] + w[4] * w[7] + w[5] * w[8]);
    dg[5] = hp[5] - (w[6] * w[6] + w[7] * w[7] + w[8] * w[8]);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        dg[6 + r] = bpe[r];
        dg[9 + r] = w[r * 3] * zeta[0] + w[r * 3 + 1] * zeta[1] + w[r * 3 + 2] * zeta[2];
    }
}

template <bool FUSED>
__device__ __forceinline__ void d_linearize(const unsigned bx, CamDev cam, int L, const int* __restrict__ lm_ptr,
                                                       const int* __restrict__ e_kf, const double* __restrict__ e_uv,
                                                       const double* __restrict__ e_info,
                                                       const double* __restrict__ poses,
                                                       const uint8_t* __restrict__ fixed,
                                                       const double* __restrict__ lms, double* W,
                                                       double* Hpp_e, double* bp_e,
                                                       double* __restrict__ Hll, double* __restrict__ bl, double lambda,
                                                       double* __restrict__ Ainv, double* __restrict__ zeta,
                                                       double* __restrict__ Dg,
                                                       const BaCtl* __restrict__ ctl, const double* __restrict__ poses_b,
                                                       const double* __restrict__ lms_b) {
    if (ctl) {   // device-side LM: nothing to do after the run has ended; a retry runs like any other trial (the estimate
                 // did not move, only lambda did)
        if (ctl->done) return;
        if (ctl->sel) { poses = poses_b; lms = lms_b; }
        lambda = ctl->lambda;
    }
    const int gid = bx * kBlock + threadIdx.x;
    const int l = gid / kGroup, sub = gid % kGroup;
    double hll[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
    // the lane's first edge waits for the second pass in LDS (18 doubles per lane, column = thread: conflict-free, no barrier -
    // a lane reads back its own words); in registers they cost the kernel a wave of occupancy (186 VGPRs)
    __shared__ double keep[18][kBlock];
    int beg = 0, end = 0;
    if (l < L) {
        const double lx = lms[3 * l], ly = lms[3 * l + 1], lz = lms[3 * l + 2];
        beg = lm_ptr[l];
        end = lm_ptr[l + 1];
        for (int e = beg + sub; e < end; e += kGroup) {
            const int kf = e_kf[e];
            double e0, e1, Jp[6], Jl[6];
            se2xyz<true>(cam, poses[3 * kf], poses[3 * kf + 1], poses[3 * kf + 2], lx, ly, lz, e_uv[2 * e],
                         e_uv[2 * e + 1], e0, e1, Jp, Jl);
            const double w0 = e_info[3 * e], w1 = e_info[3 * e + 1], w2 = e_info[3 * e + 2];
            const double we0 = w0 * e0 + w1 * e1, we1 = w1 * e0 + w2 * e1;
            double r0, r1;
            huber(e0 * we0 + e1 * we1, cam.huber, r0, r1);
            const double W0 = r1 * w0, W1 = r1 * w1, W2 = r1 * w2;  // weightedOmega
            const double or0 = -r1 * we0, or1 = -r1 * we1;          // omega_r
            double WJl[6], WJp[6];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                WJl[c] = W0 * Jl[c] + W1 * Jl[3 + c];
                WJl[3 + c] = W1 * Jl[c] + W2 * Jl[3 + c];
                WJp[c] = W0 * Jp[c] + W1 * Jp[3 + c];
                WJp[3 + c] = W1 * Jp[c] + W2 * Jp[3 + c];
            }
            hll[0] += Jl[0] * WJl[0] + Jl[3] * WJl[3];
            hll[1] += Jl[0] * WJl[1] + Jl[3] * WJl[4];
            hll[2] += Jl[0] * WJl[2] + Jl[3] * WJl[5];
            hll[3] += Jl[1] * WJl[1] + Jl[4] * WJl[4];
            hll[4] += Jl[1] * WJl[2] + Jl[4] * WJl[5];
            hll[5] += Jl[2] * WJl[2] + Jl[5] * WJl[5];
#pragma unroll
            for (int r = 0; r < 3; ++r) b[r] += Jl[r] * or0 + Jl[3 + r] * or1;
            const bool fr = !fixed[kf];
            double hpl[9], hpp[6], bpe[3];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) hpl[r * 3 + c] = fr ? Jp[r] * WJl[c] + Jp[3 + r] * WJl[3 + c] : 0.0;
            hpp[0] = fr ? Jp[0] * WJp[0] + Jp[3] * WJp[3] : 0.0;
            hpp[1] = fr ? Jp[0] * WJp[1] + Jp[3] * WJp[4] : 0.0;
            hpp[2] = fr ? Jp[0] * WJp[2] + Jp[3] * WJp[5] : 0.0;
            hpp[3] = fr ? Jp[1] * WJp[1] + Jp[4] * WJp[4] : 0.0;
            hpp[4] = fr ? Jp[1] * WJp[2] + Jp[4] * WJp[5] : 0.0;
            hpp[5] = fr ? Jp[2] * WJp[2] + Jp[5] * WJp[5] : 0.0;
#pragma unroll
            for (int r = 0; r < 3; ++r) bpe[r] = fr ? Jp[r] * or0 + Jp[3 + r] * or1 : 0.0;
            if (FUSED && e == beg + sub) {
#pragma unroll
                for (int i = 0; i < 9; ++i) keep[i][threadIdx.x] = hpl[i];
#pragma unroll
                for (int i = 0; i < 6; ++i) keep[9 + i][threadIdx.x] = hpp[i];
#pragma unroll
                for (int i = 0; i < 3; ++i) keep[15 + i][threadIdx.x] = bpe[i];
            } else {
                // un-fused pass: the pose terms of every edge (lambda_0); fused pass: a lane's further edges wait in memory
                // (landmarks with more than 8 observations) - the W slot holds the raw block until the second pass
                if (FUSED) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) W[(size_t)e * 9 + i] = hpl[i];
                }
#pragma unroll
                for (int i = 0; i < 6; ++i) Hpp_e[(size_t)e * 6 + i] = hpp[i];
#pragma unroll
                for (int i = 0; i < 3; ++i) bp_e[(size_t)e * 3 + i] = bpe[i];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) hll[i] = group_sum(hll[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) b[i] = group_sum(b[i]);
    if (l < L && sub == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) Hll[(size_t)l * 6 + i] = hll[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) bl[(size_t)l * 3 + i] = b[i];
    }
    if (FUSED && l < L) {
        double a[6], zt[3];
        chol_inv3(hll, lambda, a);
        zt[0] = a[0] * b[0];
        zt[1] = a[1] * b[0] + a[2] * b[1];
        zt[2] = a[3] * b[0] + a[4] * b[1] + a[5] * b[2];
        if (sub == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) Ainv[(size_t)l * 6 + i] = a[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) zeta[(size_t)l * 3 + i] = zt[i];
        }
        if (beg + sub < end) {
            double k_hpl[9], k_hpp[6], k_bp[3];
#pragma unroll
            for (int i = 0; i < 9; ++i) k_hpl[i] = keep[i][threadIdx.x];
#pragma unroll
            for (int i = 0; i < 6; ++i) k_hpp[i] = keep[9 + i][threadIdx.x];
#pragma unroll
            for (int i = 0; i < 3; ++i) k_bp[i] = keep[15 + i][threadIdx.x];
            write_edge_record(W + (size_t)(beg + sub) * 9, Dg + (size_t)(beg + sub) * 12, k_hpl, k_hpp, k_bp, a, zt);
        }
        for (int e = beg + sub + kGroup; e < end; e += kGroup) {   // written by this same lane above
            double hh[9], hp[6], bpe[3];
#pragma unroll
            for (int i = 0; i < 9; ++i) hh[i] = W[(size_t)e * 9 + i];
#pragma unroll
            for (int i = 0; i < 6; ++i) hp[i] = Hpp_e[(size_t)e * 6 + i];
#pragma unroll
            for (int i = 0; i < 3; ++i) bpe[i] = bp_e[(size_t)e * 3 + i];
            write_edge_record(W + (size_t)e * 9, Dg + (size_t)e * 12, hh, hp, bpe, a, zt);
        }
    }
}
template <bool FUSED>
__global__ __launch_bounds__(kBlock) void k_linearize(CamDev cam, int L, const int* __restrict__ lm_ptr,
                                                       const int* __restrict__ e_kf, const double* __restrict__ e_uv,
                                                       const double* __restrict__ e_info,
                                                       const double* __restrict__ poses,
                                                       const uint8_t* __restrict__ fixed,
                                                       const double* __restrict__ lms, double* W,
                                                       double* Hpp_e, double* bp_e,
                                                       double* __restrict__ Hll, double* __restrict__ bl, double lambda,
                                                       double* __restrict__ Ainv, double* __restrict__ zeta,
                                                       double* __restrict__ Dg,
                                                       const BaCtl* __restrict__ ctl, const double* __restrict__ poses_b,
                                                       const double* __restrict__ lms_b) {
    d_linearize<FUSED>(blockIdx.x, cam, L, lm_ptr, e_kf, e_uv, e_info, poses, fixed, lms, W, Hpp_e, bp_e, Hll, bl, lambda, Ainv, zeta, Dg, ctl, poses_b, lms_b);
}

// ---------------------------------------------------------------------------------------------
// k_odometry: one thread per PreEdgeSE2 (EdgeSE2XYZ.h:62-102, no robust kernel): blocks Oii, Ojj, Oij (3x3
// row-major) and gradients obi, obj.  Fixed vertices get zero blocks (constructQuadraticForm skips them).
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ void d_odometry(const unsigned bx, int O, const int* __restrict__ o_i, const int* __restrict__ o_j,
                           const double* __restrict__ o_meas, const double* __restrict__ o_info,
                           const double* __restrict__ poses, const uint8_t* __restrict__ fixed,
                           double* __restrict__ Oii, double* __restrict__ Ojj, double* __restrict__ Oij,
                           double* __restrict__ obi, double* __restrict__ obj, const BaCtl* __restrict__ ctl,
                           const double* __restrict__ poses_b) {
    if (ctl && ctl->sel) poses = poses_b;   // (poses = the "a" buffer then: the controller says which holds the estimate)
    const int k = bx * blockDim.x + threadIdx.x;
    if (k >= O) return;
    const int i = o_i[k], j = o_j[k];
    double e[3], A[9], B[9];
    pre_se2(poses + 3 * i, poses + 3 * j, o_meas + 3 * k, e, A, B);
    const double* W = o_info + 9 * k;
    double omr[3], WA[9], WB[9];
    for (int r = 0; r < 3; ++r) {
        omr[r] = -(W[r * 3] * e[0] + W[r * 3 + 1] * e[1] + W[r * 3 + 2] * e[2]);
        for (int c = 0; c < 3; ++c) {
            WA[r * 3 + c] = W[r * 3] * A[c] + W[r * 3 + 1] * A[3 + c] + W[r * 3 + 2] * A[6 + c];
            WB[r * 3 + c] = W[r * 3] * B[c] + W[r * 3 + 1] * B[3 + c] + W[r * 3 + 2] * B[6 + c];
        }
    }
    const bool fi = !fixed[i], fj = !fixed[j];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) {
            const double aa = A[r] * WA[c] + A[3 + r] * WA[3 + c] + A[6 + r] * WA[6 + c];
            const double ab = A[r] * WB[c] + A[3 + r] * WB[3 + c] + A[6 + r] * WB[6 + c];
            const double bb = B[r] * WB[c] + B[3 + r] * WB[3 + c] + B[6 + r] * WB[6 + c];
            Oii[k * 9 + r * 3 + c] = fi ? aa : 0.0;
            Ojj[k * 9 + r * 3 + c] = fj ? bb : 0.0;
            Oij[k * 9 + r * 3 + c] = (fi && fj) ? ab : 0.0;
        }
        obi[k * 3 + r] = fi ? A[r] * omr[0] + A[3 + r] * omr[1] + A[6 + r] * omr[2] : 0.0;
        obj[k * 3 + r] = fj ? B[r] * omr[0] + B[3 + r] * omr[1] + B[6 + r] * omr[2] : 0.0;
    }
}
__global__ void k_odometry(int O, const int* __restrict__ o_i, const int* __restrict__ o_j,
                           const double* __restrict__ o_meas, const double* __restrict__ o_info,
                           const double* __restrict__ poses, const uint8_t* __restrict__ fixed,
                           double* __restrict__ Oii, double* __restrict__ Ojj, double* __restrict__ Oij,
                           double* __restrict__ obi, double* __restrict__ obj, const BaCtl* __restrict__ ctl,
                           const double* __restrict__ poses_b) {
    d_odometry(blockIdx.x, O, o_i, o_j, o_meas, o_info, poses, fixed, Oii, Ojj, Oij, obi, obj, ctl, poses_b);
}

// ---------------------------------------------------------------------------------------------
// k_pose_reduce: one wave per pose.  Hpp (3x3 row-major, full) and bp (3) = sum over the pose's observation edges
// (CSR pose_ptr / pose_edges) + its odometry edges (CSR podo_ptr / podo_item: item = 2*k + (pose is j ? 1 : 0)).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double d_pose_reduce_max(const unsigned bx, int P, const int* __restrict__ pose_ptr,
                                                         const int* __restrict__ pose_edges,
                                                         const double* __restrict__ Hpp_e,
                                                         const double* __restrict__ bp_e,
                                                         const int* __restrict__ podo_ptr,
                                                         const int* __restrict__ podo_item,
                                                         const double* __restrict__ Oii, const double* __restrict__ Ojj,
                                                         const double* __restrict__ obi, const double* __restrict__ obj,
                                                         double* __restrict__ Hpp, double* __restrict__ bp) {
    const int p = bx * (kBlock / 64) + threadIdx.x / 64;
    const int lane = threadIdx.x & 63;
    if (p >= P) return 0.0;
    double h[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
    for (int t = pose_ptr[p] + lane; t < pose_ptr[p + 1]; t += 64) {
        const int e = pose_edges[t];
#pragma unroll
        for (int i = 0; i < 6; ++i) h[i] += Hpp_e[(size_t)e * 6 + i];
#pragma unroll
        for (int i = 0; i < 3; ++i) b[i] += bp_e[(size_t)e * 3 + i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) h[i] = wave_sum(h[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) b[i] = wave_sum(b[i]);
    if (lane == 0) {
        double H[9] = {h[0], h[1], h[2], h[1], h[3], h[4], h[2], h[4], h[5]};
        for (int t = podo_ptr[p]; t < podo_ptr[p + 1]; ++t) {
            const int k = podo_item[t] >> 1, isj = podo_item[t] & 1;
            const double* M = (isj ? Ojj : Oii) + (size_t)k * 9;
            const double* g = (isj ? obj : obi) + (size_t)k * 3;
            for (int i = 0; i < 9; ++i) H[i] += M[i];
            for (int i = 0; i < 3; ++i) b[i] += g[i];
        }
        for (int i = 0; i < 9; ++i) Hpp[(size_t)p * 9 + i] = H[i];
        for (int i = 0; i < 3; ++i) bp[(size_t)p * 3 + i] = b[i];
        return fmax(fabs(H[0]), fmax(fabs(H[4]), fabs(H[8])));   // lane 0: the block's largest diagonal entry (lambda_0)
    }
    return 0.0;
}
__device__ __forceinline__ void d_pose_reduce(const unsigned bx, int P, const int* __restrict__ pose_ptr,
                                              const int* __restrict__ pose_edges, const double* __restrict__ Hpp_e,
                                              const double* __restrict__ bp_e, const int* __restrict__ podo_ptr,
                                              const int* __restrict__ podo_item, const double* __restrict__ Oii,
                                              const double* __restrict__ Ojj, const double* __restrict__ obi,
                                              const double* __restrict__ obj, double* __restrict__ Hpp, double* __restrict__ bp) {
    (void)d_pose_reduce_max(bx, P, pose_ptr, pose_edges, Hpp_e, bp_e, podo_ptr, podo_item, Oii, Ojj, obi, obj, Hpp, bp);
}
__global__ __launch_bounds__(kBlock) void k_pose_reduce(int P, const int* __restrict__ pose_ptr,
                                                         const int* __restrict__ pose_edges,
                                                         const double* __restrict__ Hpp_e,
                                                         const double* __restrict__ bp_e,
                                                         const int* __restrict__ podo_ptr,
                                                         const int* __restrict__ podo_item,
                                                         const double* __restrict__ Oii, const double* __restrict__ Ojj,
                                                         const double* __restrict__ obi, const double* __restrict__ obj,
                                                         double* __restrict__ Hpp, double* __restrict__ bp) {
    d_pose_reduce(blockIdx.x, P, pose_ptr, pose_edges, Hpp_e, bp_e, podo_ptr, podo_item, Oii, Ojj, obi, obj, Hpp, bp);
}

// k_pose_reduce and k_maxdiag in one launch (single GPU, first trial of an optimize()): the pose blocks by the first
// workgroups, the landmark diagonals by the others, every workgroup folds its maximum into one word with an atomic
// maximum on the bit pattern (non-negative doubles order like their bits; a maximum is exact whatever the order), and the
// workgroup that arrives last sets lambda_0 = 1e-5 * max.  One launch and 13 us less in front of the first Schur step.
constexpr int kL0PerBlock = 8 * kBlock;   // landmarks per workgroup of the landmark part
__global__ __launch_bounds__(kBlock) void k_lambda0(int P, const int* __restrict__ pose_ptr, const int* __restrict__ pose_edges,
                                                     const double* __restrict__ Hpp_e, const double* __restrict__ bp_e,
                                                     const int* __restrict__ podo_ptr, const int* __restrict__ podo_item,
                                                     const double* __restrict__ Oii, const double* __restrict__ Ojj,
                                                     const double* __restrict__ obi, const double* __restrict__ obj,
                                                     double* __restrict__ Hpp, double* __restrict__ bp,
                                                     const uint8_t* __restrict__ fixed, int L, const double* __restrict__ Hll,
                                                     unsigned long long* __restrict__ acc, double* __restrict__ out,
                                                     BaCtl* __restrict__ ctl) {
    __shared__ double sm[kBlock / 64];
    const int nP = (P + kBlock / 64 - 1) / (kBlock / 64), nL = (L + kL0PerBlock - 1) / kL0PerBlock;
    double m = 0.0;
    if ((int)blockIdx.x < nP) {
        const double dm = d_pose_reduce_max(blockIdx.x, P, pose_ptr, pose_edges, Hpp_e, bp_e, podo_ptr, podo_item, Oii, Ojj, obi, obj, Hpp, bp);
        const int p = blockIdx.x * (kBlock / 64) + threadIdx.x / 64;
        if ((threadIdx.x & 63) == 0 && p < P && !fixed[p]) m = dm;
    } else {
        const int i0 = ((int)blockIdx.x - nP) * kL0PerBlock + (int)threadIdx.x;
        double v[8][3];
#pragma unroll
        for (int u = 0; u < 8; ++u) {   // eight landmarks' diagonals in flight per thread
            const size_t i = (size_t)min(i0 + u * kBlock, L - 1);
            v[u][0] = Hll[i * 6 + 0]; v[u][1] = Hll[i * 6 + 3]; v[u][2] = Hll[i * 6 + 5];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i0 + u * kBlock < L) m = fmax(m, fmax(fabs(v[u][0]), fmax(fabs(v[u][1]), fabs(v[u][2]))));
    }
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) m = fmax(m, __shfl_xor(m, s));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) m = fmax(m, sm[w]);
        if (m > 0.0) atomicMax(acc, (unsigned long long)__double_as_longlong(m));
        __threadfence();
        const unsigned long long arrived = atomicAdd(acc + 1, 1ull);
        if (arrived == (unsigned long long)(nP + nL) - 1ull) {   // the last workgroup: everybody's maximum is in
            __threadfence();
            const double mm = __longlong_as_double((long long)atomicExch(acc, 0ull));   // (and the words are clear for the next run)
            atomicExch(acc + 1, 0ull);
            out[0] = mm;
            ctl->lambda = 1e-5 * mm;
            ctl->ni = 2;
        }
    }
}

// max |diag| over a strided array (computeLambdaInit); single block, deterministic.
// max |diag H| over the landmark blocks and the free poses' blocks (computeLambdaInit).  The pose diagonals come either as
// `dpp` values per pose (Hpp_diag3) or, with stride9 set, straight from the 3 x 3 row-major blocks of k_pose_reduce; with
// `ctl` the kernel also sets lambda_0 = 1e-5 * max itself (single GPU: no k_extract_diag / k_set_lambda launches).
__device__ __forceinline__ void d_maxdiag(const unsigned bx, int L, const double* __restrict__ Hll, int P, const double* __restrict__ Hpp_diag3,
                          const uint8_t* __restrict__ fixed, double* __restrict__ out, int dpp, int stride9,
                          BaCtl* __restrict__ ctl) {
    __shared__ double sm[16];
    double m = 0;
    for (int i0 = threadIdx.x; i0 < L; i0 += 4 * blockDim.x) {   // four landmarks' diagonals in flight per thread
        double v[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t i = (size_t)min(i0 + u * (int)blockDim.x, L - 1);
            v[u][0] = Hll[i * 6 + 0]; v[u][1] = Hll[i * 6 + 3]; v[u][2] = Hll[i * 6 + 5];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) m = fmax(m, fmax(fabs(v[u][0]), fmax(fabs(v[u][1]), fabs(v[u][2]))));
    }
    for (int i = threadIdx.x; i < P; i += blockDim.x)
        if (!fixed[i])
            for (int r = 0; r < dpp; ++r)
                m = fmax(m, fabs(stride9 ? Hpp_diag3[(size_t)i * 9 + r * 4] : Hpp_diag3[(size_t)i * dpp + r]));
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) m = fmax(m, __shfl_xor(m, s));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmax(m, sm[w]);
        out[0] = m;
        if (ctl) {
            ctl->lambda = 1e-5 * m;
            ctl->ni = 2;
        }
    }
}
__global__ void k_maxdiag(int L, const double* __restrict__ Hll, int P, const double* __restrict__ Hpp_diag3,
                          const uint8_t* __restrict__ fixed, double* __restrict__ out, int dpp, int stride9,
                          BaCtl* __restrict__ ctl) {
    d_maxdiag(blockIdx.x, L, Hll, P, Hpp_diag3, fixed, out, dpp, stride9, ctl);
}

__global__ void k_extract_diag(int P, const double* __restrict__ Hpp, double* __restrict__ d3) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P * 3) d3[i] = Hpp[(size_t)(i / 3) * 9 + (i % 3) * 4];
}

// (Hll + lambda I)^-1 as 6 symmetric entries (the SE3-expmap model's k3_schur_lm)
__device__ inline void inv_sym3(const double h[6], double lambda, double d[6]) {
    const double a = h[0] + lambda, b = h[1], c = h[2], e = h[3] + lambda, f = h[4], i = h[5] + lambda;
    const double A = e * i - f * f, B = -(b * i - f * c), C = b * f - e * c;
    const double id = 1.0 / (a * A + b * B + c * C);
    d[0] = A * id; d[1] = B * id; d[2] = C * id;
    d[3] = (a * i - c * c) * id; d[4] = -(a * f - c * b) * id; d[5] = (a * e - b * b) * id;
}

// ---------------------------------------------------------------------------------------------
// k_reduce2: the reduced system in ONE launch, output-stationary, atomic-free.
//   blocks [0, nb_off)        : off-diagonal (a < b) blocks, one THREAD per entry (9 consecutive lanes per block):
//                               S_ab(r,c) = - sum_{(i,j) in pairs(a,b)} W_i(r,:) . W_j(c,:)  (+ the PreEdgeSE2 block)
//   blocks [nb_off, nb_off+nb_diag) : one WAVE per pose: S_aa = Hpp_a + lambda I - sum_e W_e W_e^T, b_s, b_p
//                               (absorbs k_pose_reduce, k_odometry and k_reduce_odo; odometry terms are recomputed
//                               on the fly from the poses: <= 2 edges per pose)
//   last wave of the diagonal part clears the padding of the augmented matrix.
// ---------------------------------------------------------------------------------------------
__device__ inline void odo_terms(const double* poses, const uint8_t* fixed, const int* o_i, const int* o_j,
                                 const double* o_meas, const double* o_info, int k, double e[3], double A[9], double B[9],
                                 double WA[9], double WB[9], double omr[3]) {
    const int i = o_i[k], j = o_j[k];
    pre_se2(poses + 3 * i, poses + 3 * j, o_meas + 3 * k, e, A, B);
    const double* W = o_info + 9 * k;
    for (int r = 0; r < 3; ++r) {
        omr[r] = -(W[r * 3] * e[0] + W[r * 3 + 1] * e[1] + W[r * 3 + 2] * e[2]);
        for (int c = 0; c < 3; ++c) {
            WA[r * 3 + c] = W[r * 3] * A[c] + W[r * 3 + 1] * A[3 + c] + W[r * 3 + 2] * A[6 + c];
            WB[r * 3 + c] = W[r * 3] * B[c] + W[r * 3 + 1] * B[3 + c] + W[r * 3 + 2] * B[6 + c];
        }
    }
}

constexpr int kGrpPerWG = 28;   // 9-lane groups per 256-thread workgroup (252 lanes used)
constexpr int kChunk = 16;      // contributor pairs per group (d_reduce2 keeps the indices of two rounds of 8 per lane)

// Off-diagonal part: workgroup w owns the groups [w*28, w*28+28) of the host-built plan.  A group = 9 lanes (one per
// entry of a 3x3 block) accumulating one chunk of <= kChunk contributor pairs of ONE reduced-system block; blocks
// with several chunks are packed into the same workgroup and combined in LDS in chunk order (deterministic).
// grp = {block index or -1, first pair, last pair, (first group of the block in this WG) | (number of groups << 8)}
__device__ __forceinline__ void d_reduce2(const unsigned bx, int P, int ld, int nwg_off, double lambda, int root,
                                                     const int4* __restrict__ grp, const int* __restrict__ blk_a,
                                                     const int* __restrict__ blk_b, const int* __restrict__ pair_i,
                                                     const int* __restrict__ pair_j, const int* __restrict__ blk_odo,
                                                     const double* __restrict__ W,
                                                     const double* __restrict__ Dg,
                                                     const uint8_t* __restrict__ fixed, const int* __restrict__ pose_ptr,
                                                     const int* __restrict__ pose_edges, const int* __restrict__ podo_ptr,
                                                     const int* __restrict__ podo_item, const int* __restrict__ o_i,
                                                     const int* __restrict__ o_j, const double* __restrict__ o_meas,
                                                     const double* __restrict__ o_info, const double* __restrict__ poses,
                                                     double* __restrict__ S, double* __restrict__ bp,
                                                     const BaCtl* __restrict__ ctl, const double* __restrict__ poses_b,
                                                     unsigned* __restrict__ epoch, const int* __restrict__ pose_off, int nsys) {
    // pose p's three unknowns live in the columns pose_off[p] .. + 2 of the system (the solver's fill-reducing order, with
    // identity padding between its partitions: solve_plan_build); nullptr = natural order, 3 p
    const int n = pose_off ? nsys : 3 * P;
    __shared__ double part[kGrpPerWG][9];
    __shared__ double dpart[kBlock / 64][12];
    // grid = [P + 1 diagonal workgroups | off-diagonal workgroups, rounded up to a multiple of the 8 XCDs].  The diagonal
    // ones have the longest dependent chain (edge list -> edge rows -> landmark z, then the PreEdgeSE2 terms) and must
    // not queue behind the others for a CU slot, so they come FIRST.
    const int ndiag = (P + 1 + 7) & ~7;
    if ((int)bx >= ndiag) {
        const int bid = (int)bx - ndiag;
        const int nwg_pad = (nwg_off + 7) & ~7;
        // XCD-aware order: workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2).  The plan is ordered
        // by pose block row, so XCD x takes the x-th CONTIGUOUS eighth of it: the W rows of its pose range are
        // fetched into one L2 instead of all eight (measured: 178 MB of HBM traffic per launch for 51 MB of operands).
        const int wg = (bid & 7) * (nwg_pad >> 3) + (bid >> 3);
        if (wg >= nwg_off) return;
        // 7 groups of 9 lanes per wave (lane 63 idles), 4 waves: the 9 lanes of a group share their loads through
        // ds_bpermute, so a group must not straddle two waves
        const int lane = threadIdx.x & 63, gw = lane / 9, en = lane - 9 * gw;
        const int g = gw < 7 ? (int)(threadIdx.x >> 6) * 7 + gw : kGrpPerWG;
        // A workgroup is one chain of dependent loads (controller -> group -> pair indices -> edge blocks -> block descriptor
        // -> fixed flags / columns -> store) around a few dozen multiply-adds; with ~9 us per wave and 16 waves per CU the
        // chain, not a bandwidth, is what a window batch pays for (counters, round 4).  So every link is requested as early
        // as its address is known: the group descriptor beside the controller block, the block descriptor and the indices of
        // ALL pairs of the chunk beside each other, the fixed flags and columns beside the edge blocks.
        int4 d = make_int4(-1, 0, 0, 0);
        if (g < kGrpPerWG) d = grp[(size_t)wg * kGrpPerWG + g];
        if (ctl) {
            if (ctl->done) return;
            if (ctl->sel) poses = poses_b;
        }
        const int r = en / 3, c = en - 3 * r;
        double acc0 = 0, acc1 = 0;
        int a = 0, b = 0, od = -1, ca = 0, cb = 0;
        bool free_ab = false;
        if (d.x >= 0) {
            // lane t < 8 of the group holds the indices of pairs t and 8 + t of sixteen pairs (a chunk is kChunk = 16 pairs unless
            // its block has more than 28 chunks' worth); slots past the chunk are clamped to its last pair and weighted 0 (no
            // branches inside a round)
            const int gb = 9 * gw, t8 = min(en, 7);
            int q0 = min(d.y + t8, d.z - 1), q1 = min(d.y + 8 + t8, d.z - 1);
            int my_i0 = 0, my_j0 = 0, my_i1 = 0, my_j1 = 0;
            if (d.z > d.y) { my_i0 = pair_i[q0]; my_j0 = pair_j[q0]; my_i1 = pair_i[q1]; my_j1 = pair_j[q1]; }
            a = blk_a[d.x];
            b = blk_b[d.x];
            od = blk_odo[d.x];
            free_ab = !fixed[a] && !fixed[b];      // (a block may have no pairs at all - an odometry edge only)
            ca = pose_off ? pose_off[a] : 3 * a;
            cb = pose_off ? pose_off[b] : 3 * b;
            // The group loads every word ONCE (counters: 21 L1 accesses per load instruction with one row per lane): lane t
            // the t-th word of both 72-byte blocks of every pair (16 load instructions per round of 8 pairs instead of 48),
            // and the row of W_i / row of W_j an entry needs comes from the neighbours' registers (ds_bpermute).
            // Same products, same order of the sums as one row per lane.
            for (int q = d.y; q < d.z; q += 8) {
                const bool second = ((q - d.y) & 8) != 0;
                if (!second && q != d.y) {   // a long chunk: the indices of its next sixteen pairs
                    q0 = min(q + t8, d.z - 1); q1 = min(q + 8 + t8, d.z - 1);
                    my_i0 = pair_i[q0]; my_j0 = pair_j[q0]; my_i1 = pair_i[q1]; my_j1 = pair_j[q1];
                }
                const int my_i = second ? my_i1 : my_i0, my_j = second ? my_j1 : my_j0;
                double wi[8], wj[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int ia = __shfl(my_i, gb + u), ib = __shfl(my_j, gb + u);
                    wi[u] = W[(size_t)ia * 9 + en];
                    wj[u] = W[(size_t)ib * 9 + en];
                }
#pragma unroll
                for (int u = 0; u < 8; u += 2) {
                    const double w0 = (q + u < d.z) ? 1.0 : 0.0, w1 = (q + u + 1 < d.z) ? 1.0 : 0.0;
                    double y[2][3], hh[2][3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        y[0][k] = __shfl(wi[u], gb + 3 * r + k);
                        hh[0][k] = __shfl(wj[u], gb + 3 * c + k);
                        y[1][k] = __shfl(wi[u + 1], gb + 3 * r + k);
                        hh[1][k] = __shfl(wj[u + 1], gb + 3 * c + k);
                    }
                    acc0 += w0 * (y[0][0] * hh[0][0] + y[0][1] * hh[0][1] + y[0][2] * hh[0][2]);
                    acc1 += w1 * (y[1][0] * hh[1][0] + y[1][1] * hh[1][1] + y[1][2] * hh[1][2]);
                }
            }
            part[g][en] = acc0 + acc1;
        }
        __syncthreads();
        if (d.x >= 0 && (d.w & 0xff) == g) {
            const int ng = d.w >> 8;
            double tot = 0;
            for (int t = 0; t < ng; ++t) tot += part[g + t][en];
            double out = 0.0;
            if (free_ab) {
                out = -tot;
                if (od >= 0) {  // PreEdgeSE2 between a and b: A^T W B (transposed when the edge runs b -> a)
                    double e[3], A[9], B[9], WA[9], WB[9], omr[3];
                    odo_terms(poses, fixed, o_i, o_j, o_meas, o_info, od >> 1, e, A, B, WA, WB, omr);
                    // (no dynamic index into the private arrays: that would put them - and a scratch segment - into memory)
                    const int rr = (od & 1) ? c : r, cc = (od & 1) ? r : c;
                    const double a0 = rr == 0 ? A[0] : (rr == 1 ? A[1] : A[2]), a1 = rr == 0 ? A[3] : (rr == 1 ? A[4] : A[5]),
                                 a2 = rr == 0 ? A[6] : (rr == 1 ? A[7] : A[8]);
                    const double b0 = cc == 0 ? WB[0] : (cc == 1 ? WB[1] : WB[2]), b1 = cc == 0 ? WB[3] : (cc == 1 ? WB[4] : WB[5]),
                                 b2 = cc == 0 ? WB[6] : (cc == 1 ? WB[7] : WB[8]);
                    out += a0 * b0 + a1 * b1 + a2 * b2;
                }
            }
            S[(size_t)(ca + r) * ld + cb + c] = out;
            S[(size_t)(cb + c) * ld + ca + r] = out;
        }
        return;
    }
    if (ctl) {
        if (ctl->done) return;
        if (ctl->sel) poses = poses_b;
        lambda = ctl->lambda;
    }
    // ---- diagonal part: one workgroup per pose (+ one that clears the padding)
    const int p = (int)bx;
    if (p > P) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double* __restrict__ bs = S + (size_t)n * ld;
    if (p == P) {
        for (size_t t = (size_t)n * ld + n + threadIdx.x; t < (size_t)ld * ld; t += kBlock) S[t] = 0.0;
        if (threadIdx.x == 0) {
            S[(size_t)ld * ld + 2] = 0.0;  // factorisation flag of the solve that follows
            *epoch += 1u;                   // ... and its epoch (the tile flags of k_chol_tiles are compared with it)
        }
        return;
    }
    double acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // S diag (6 sym), bp (3), g (3)
    const bool fa = fixed[p];
    __shared__ double odoc[8][12];  // PreEdgeSE2 contributions of this pose, one lane each (sin/cos: kept off thread 0)
    const int no = fa ? 0 : podo_ptr[p + 1] - podo_ptr[p];
    if ((int)threadIdx.x >= kBlock - 8 && (int)threadIdx.x - (kBlock - 8) < min(no, 8)) {
        const int t = podo_ptr[p] + (int)threadIdx.x - (kBlock - 8);
        const int k = podo_item[t] >> 1, isj = podo_item[t] & 1;
        double e[3], A[9], B[9], WA[9], WB[9], omr[3];
        odo_terms(poses, fixed, o_i, o_j, o_meas, o_info, k, e, A, B, WA, WB, omr);
        double J[9], WJ[9];   // (element-wise selects: a pointer to one of two private arrays would put both into scratch memory)
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9) { J[t9] = isj ? B[t9] : A[t9]; WJ[t9] = isj ? WB[t9] : WA[t9]; }
        double* o = odoc[threadIdx.x - (kBlock - 8)];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c) o[r * 3 + c] = J[r] * WJ[c] + J[3 + r] * WJ[3 + c] + J[6 + r] * WJ[6 + c];
            o[9 + r] = J[r] * omr[0] + J[3 + r] * omr[1] + J[6 + r] * omr[2];
        }
    }
    if (!fa) {
        // gather of the per-edge diagonal records (write_diag_record): edge list -> one 96-byte record per edge; the
        // indices of up to four edges per thread are fetched first, then all the records
        const int e0 = pose_ptr[p], ne = pose_ptr[p + 1] - e0;
        for (int base = 0; base < ne; base += 4 * kBlock) {
            int ee[4];
            double wgt[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = base + u * kBlock + (int)threadIdx.x;
                wgt[u] = t < ne ? 1.0 : 0.0;
                ee[u] = pose_edges[e0 + min(t, ne - 1)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (base + u * kBlock >= ne) break;  // uniform
                const double2* rec = reinterpret_cast<const double2*>(Dg + (size_t)ee[u] * 12);
                const double w = wgt[u];
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const double2 v = rec[i];
                    acc[2 * i] += w * v.x;
                    acc[2 * i + 1] += w * v.y;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = wave_sum(acc[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 12; ++i) dpart[wv][i] = acc[i];
    }
    __syncthreads();
    if (threadIdx.x < 12) {  // entries 0..8 of the 3x3 block (row-major), 9..11 of the right-hand sides
        const int i = threadIdx.x;
        const int sym[9] = {0, 1, 2, 1, 3, 4, 2, 4, 5};
        const int src = i < 9 ? sym[i] : i - 3;  // H entry, or bp component (slots 6..8)
        double v = dpart[0][src] + dpart[1][src] + dpart[2][src] + dpart[3][src];
        for (int t = 0; t < min(no, 8); ++t) v += odoc[t][i];
        for (int t = podo_ptr[p] + 8; t < podo_ptr[p] + no; ++t) {  // more than 8 PreEdgeSE2 at one pose: serial tail
            const int k = podo_item[t] >> 1, isj = podo_item[t] & 1;
            double e[3], A[9], B[9], WA[9], WB[9], omr[3];
            odo_terms(poses, fixed, o_i, o_j, o_meas, o_info, k, e, A, B, WA, WB, omr);
            // (selects, no pointers or dynamic indices into the private arrays: they would live in scratch memory, and the
            // kernel would need a scratch segment for a tail that practically never runs)
            const int r = i < 9 ? i / 3 : i - 9, c = i < 9 ? i - 3 * (i / 3) : 0;
            double jr[3], wc[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const double j0 = isj ? B[3 * q] : A[3 * q], j1 = isj ? B[3 * q + 1] : A[3 * q + 1], j2 = isj ? B[3 * q + 2] : A[3 * q + 2];
                const double w0 = isj ? WB[3 * q] : WA[3 * q], w1 = isj ? WB[3 * q + 1] : WA[3 * q + 1], w2 = isj ? WB[3 * q + 2] : WA[3 * q + 2];
                jr[q] = r == 0 ? j0 : (r == 1 ? j1 : j2);
                wc[q] = i < 9 ? (c == 0 ? w0 : (c == 1 ? w1 : w2)) : omr[q];
            }
            v += jr[0] * wc[0] + jr[1] * wc[1] + jr[2] * wc[2];
        }
        if (i < 9) {
            const int r = i / 3, c = i - 3 * r;
            double out;
            if (fa) out = (r == c && root) ? 1.0 : 0.0;
            else out = v + ((r == c && root) ? lambda : 0.0);
            const int cp = pose_off ? pose_off[p] : 3 * p;
            S[(size_t)(cp + r) * ld + cp + c] = out;
        } else {
            const int r = i - 9;
            const double gz = dpart[0][9 + r] + dpart[1][9 + r] + dpart[2][9 + r] + dpart[3][9 + r];
            bs[(pose_off ? pose_off[p] : 3 * p) + r] = fa ? 0.0 : v - gz;
            bp[(size_t)p * 3 + r] = fa ? 0.0 : v;
        }
    }
}
__global__ __launch_bounds__(kBlock) void k_reduce2(int P, int ld, int nwg_off, double lambda, int root,
                                                     const int4* __restrict__ grp, const int* __restrict__ blk_a,
                                                     const int* __restrict__ blk_b, const int* __restrict__ pair_i,
                                                     const int* __restrict__ pair_j, const int* __restrict__ blk_odo,
                                                     const double* __restrict__ W,
                                                     const double* __restrict__ Dg,
                                                     const uint8_t* __restrict__ fixed, const int* __restrict__ pose_ptr,
                                                     const int* __restrict__ pose_edges, const int* __restrict__ podo_ptr,
                                                     const int* __restrict__ podo_item, const int* __restrict__ o_i,
                                                     const int* __restrict__ o_j, const double* __restrict__ o_meas,
                                                     const double* __restrict__ o_info, const double* __restrict__ poses,
                                                     double* __restrict__ S, double* __restrict__ bp,
                                                     const BaCtl* __restrict__ ctl, const double* __restrict__ poses_b,
                                                     unsigned* __restrict__ epoch, const int* __restrict__ pose_off, int nsys) {
    d_reduce2(blockIdx.x, P, ld, nwg_off, lambda, root, grp, blk_a, blk_b, pair_i, pair_j, blk_odo, W, Dg, fixed, pose_ptr, pose_edges, podo_ptr, podo_item, o_i, o_j, o_meas, o_info, poses, S, bp, ctl, poses_b, epoch, pose_off, nsys);
}

// odometry pose-pose blocks: S_ij += Oij, S_ji += Oij^T.  One thread per (edge, entry).
__global__ void k_reduce_odo(int O, int ld, const int* __restrict__ o_i, const int* __restrict__ o_j,
                             const double* __restrict__ Oij, double* __restrict__ S) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= O * 9) return;
    const int k = t / 9, r = (t % 9) / 3, c = t % 3;
    const int i = o_i[k], j = o_j[k];
    const double v = Oij[t];
    if (v == 0.0) return;
    atomicAdd(&S[(size_t)(3 * i + r) * ld + 3 * j + c], v);
    atomicAdd(&S[(size_t)(3 * j + c) * ld + 3 * i + r], v);
}

// ---------------------------------------------------------------------------------------------
// Dense pose solve on the device: blocked right-looking LL^T (FP64) of the augmented matrix
//     [ S  ]   rows 0..n-1 : reduced system (lower triangle is used)
//     [ b' ]   row  n      : right-hand side     -> the elimination maps it to y' = (L^-1 b)'
//     [ I  ]   (kept in R) : identity            -> the elimination maps it to R  = L^-T
// so neither triangular solve has a sequential phase: x = R y is one GEMV at the end.
// One launch per 32-wide block column (k_chol_step); replaces CHOLMOD on the (3P)^2 system.
// ---------------------------------------------------------------------------------------------
constexpr int kNB = 32;
constexpr int kSlabs = 4;
constexpr int kSlabDoubles = 2 * kNB * 8;   // a published slab: M then MR, each 4 column pairs x 32 rows x 2 doubles
// publish buffer of the dataflow solve: tile (kind, i, j) -> 4 slabs; kind 0 = L tiles below the diagonal, 1 = R = L^-T tiles
__host__ __device__ inline size_t pub_tile(int kind, int i, int j, int nt, int nbc) {
    return ((size_t)(kind * nt + i) * nbc + j) * (kSlabs * kSlabDoubles);
}
// the solver's work buffer: R of the column-launch solver (ld^2), or the publish buffer of the dataflow solve (nbc <= nt
// block columns) followed by y_un (ld)
inline size_t chol_work_doubles(int ld) {
    const size_t nt = (size_t)ld / kNB;
    return std::max((size_t)ld * ld, 2 * nt * nt * kSlabs * kSlabDoubles + (size_t)ld);
}   // k_chol_tiles: a tile is handed on in the four 8-column slabs its waves eliminate

__device__ inline double bcast_lane(double v, int lane) {  // lane must be a compile-time / wave-uniform constant
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ inline double fast_rcp(double d) {  // v_rcp_f64 + 2 Newton steps (full FP64 accuracy)
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}
__device__ inline double fast_rsqrt(double d) {  // v_rsq_f64 + 2 Newton steps
    double y = __builtin_amdgcn_rsq(d);
    y = y * fma(-0.5 * d * y, y, 1.5);
    y = y * fma(-0.5 * d * y, y, 1.5);
    return y;
}

// Step k of the factorisation, ONE launch per 32-wide block column: the trailing update with panel k-1 and the
// elimination of panel k are fused, so a solve is nbc launches instead of 2 nbc (kernel boundaries, not flops, bound
// this 600-column problem).  Grid (nbc - k, nt): x = block column j = k + x, y = tile row (A rows k.., then R rows).
//   * every workgroup first applies panel k-1 to its tile:   T(i,j) -= L(i,k-1) L(j,k-1)^T          (256 threads)
//     (R(r,j) is first touched at step r+1, where its previous value is known to be zero);
//   * workgroups of columns j > k write T back and are done;
//   * workgroups of column k also form the updated diagonal tile D = A(k,k) - L(k,k-1) L(k,k-1)^T (every one of
//     them, redundantly - nothing is exchanged between workgroups) and wave 0 eliminates the stacked 64x32 matrix
//     [D; T] held entirely in registers (lane = row, 32 doubles per lane): lanes 0..31 = D, lanes 32..63 = T, the
//     identity for the diagonal workgroup (-> R(k,k)), or the R(r,k) tile.
//         M[r][c] -= M[r][j] * M[c][j] / M[j][j]   (c > j),      L[r][j] = M[r][j] / sqrt(M[j][j]) at the end.
//     The loop is software pipelined: column j first updates column j+1 and starts the reciprocal of the NEXT
//     pivot, then does the bulk rank-1 update of columns j+2.. while that chain is in flight.  M[c][j] of the bulk
//     comes from a 64-entry LDS column (uniform-address ds_read = broadcast); only the pivot path uses v_readlane.
//     The loop has no branches (one basic block): columns past n are replaced by a decoupled block.
// L(k,k) itself is never read again and is not written (so no workgroup writes what another one reads).
__global__ __launch_bounds__(256) void k_chol_step(double* __restrict__ A, double* __restrict__ R, int ld, int n, int nt,
                                                    int k, double* __restrict__ fail, const BaCtl* __restrict__ ctl) {
    if (ctl && ctl->done) return;
    __shared__ __attribute__((aligned(16))) double Ti[kNB][kNB + 2];  // L(i,k-1), then the updated tile T
    __shared__ __attribute__((aligned(16))) double Tj[kNB][kNB + 2];  // L(j,k-1), then the updated diagonal D
    __shared__ __attribute__((aligned(16))) double colA[64];  // two column buffers: column j+1 can be staged while
    __shared__ __attribute__((aligned(16))) double colB[64];  // the bulk update of column j is still reading
    const int tid = threadIdx.x;
    const int j = k + (int)blockIdx.x;
    const int nS = nt - k;
    const bool isR = (int)blockIdx.y >= nS;
    const int i = isR ? (int)blockIdx.y - nS : k + (int)blockIdx.y;  // tile row (of A, or of R)
    if (!isR && j > i) return;
    const bool elim = (j == k);
    const bool isDiag = elim && !isR && i == k;
    double* own = isR ? R : A;
    const int r = tid / 8, cc = (tid % 8) * 4;
    double* ownp = own + (size_t)(kNB * i + r) * ld + kNB * j + cc;
    double2 t0, t1, d0 = make_double2(0, 0), d1 = make_double2(0, 0);
    if (k > 0) {
        const int cp = kNB * (k - 1);
        const bool first = isR && i == k - 1;  // R(k-1, j): previous value is zero, never read
        const double* src = own + (size_t)(kNB * i + r) * ld + cp + cc;
        const double* srj = A + (size_t)(kNB * j + r) * ld + cp + cc;
        // issue every global load up front
        const double2 li0 = *reinterpret_cast<const double2*>(src), li1 = *reinterpret_cast<const double2*>(src + 2);
        const double2 lj0 = *reinterpret_cast<const double2*>(srj), lj1 = *reinterpret_cast<const double2*>(srj + 2);
        t0 = make_double2(0, 0);
        t1 = make_double2(0, 0);
        if (!first) {
            t0 = *reinterpret_cast<const double2*>(ownp);
            t1 = *reinterpret_cast<const double2*>(ownp + 2);
        }
        if (elim) {
            const double* dp = A + (size_t)(kNB * k + r) * ld + kNB * k + cc;
            d0 = *reinterpret_cast<const double2*>(dp);
            d1 = *reinterpret_cast<const double2*>(dp + 2);
        }
        Ti[r][cc] = li0.x; Ti[r][cc + 1] = li0.y; Ti[r][cc + 2] = li1.x; Ti[r][cc + 3] = li1.y;
        Tj[r][cc] = lj0.x; Tj[r][cc + 1] = lj0.y; Tj[r][cc + 2] = lj1.x; Tj[r][cc + 3] = lj1.y;
        __syncthreads();
        double acc[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
        double dcc[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
#pragma unroll
        for (int m = 0; m < kNB; m += 2) {
            const double a0 = Ti[r][m], a1 = Ti[r][m + 1];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[q][0] = fma(a0, Tj[cc + q][m], acc[q][0]);
                acc[q][1] = fma(a1, Tj[cc + q][m + 1], acc[q][1]);
            }
        }
        if (elim) {  // D -= L(k,k-1) L(k,k-1)^T  (Tj holds L(k,k-1))
#pragma unroll
            for (int m = 0; m < kNB; m += 2) {
                const double a0 = Tj[r][m], a1 = Tj[r][m + 1];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    dcc[q][0] = fma(a0, Tj[cc + q][m], dcc[q][0]);
                    dcc[q][1] = fma(a1, Tj[cc + q][m + 1], dcc[q][1]);
                }
            }
        }
        t0 = make_double2(t0.x - (acc[0][0] + acc[0][1]), t0.y - (acc[1][0] + acc[1][1]));
        t1 = make_double2(t1.x - (acc[2][0] + acc[2][1]), t1.y - (acc[3][0] + acc[3][1]));
        if (!elim) {
            *reinterpret_cast<double2*>(ownp) = t0;
            *reinterpret_cast<double2*>(ownp + 2) = t1;
            return;
        }
        d0 = make_double2(d0.x - (dcc[0][0] + dcc[0][1]), d0.y - (dcc[1][0] + dcc[1][1]));
        d1 = make_double2(d1.x - (dcc[2][0] + dcc[2][1]), d1.y - (dcc[3][0] + dcc[3][1]));
        __syncthreads();  // everyone is done reading L(i,k-1), L(k,k-1)
    } else {
        // k == 0: every column-0 tile is eliminated as loaded (the grid has one column)
        t0 = *reinterpret_cast<const double2*>(ownp);
        t1 = *reinterpret_cast<const double2*>(ownp + 2);
        const double* dp = A + (size_t)r * ld + cc;
        d0 = *reinterpret_cast<const double2*>(dp);
        d1 = *reinterpret_cast<const double2*>(dp + 2);
        if (!elim) return;  // (cannot happen: j == k == 0)
    }
    // stage D and T for wave 0 (row per lane)
    Ti[r][cc] = t0.x; Ti[r][cc + 1] = t0.y; Ti[r][cc + 2] = t1.x; Ti[r][cc + 3] = t1.y;
    Tj[r][cc] = d0.x; Tj[r][cc + 1] = d0.y; Tj[r][cc + 2] = d1.x; Tj[r][cc + 3] = d1.y;
    __syncthreads();
    if (tid >= 64) return;
    const int lane = tid;
    const int c0 = kNB * k;
    const int ncol = min(kNB, n - c0);
    const int rr = lane & 31;
    double m[kNB];
    {
        const double* lp = (lane < kNB) ? &Tj[rr][0] : &Ti[rr][0];
        const bool ident = isDiag && lane >= kNB;
#pragma unroll
        for (int c = 0; c < kNB; c += 2) {
            const double2 v = *reinterpret_cast<const double2*>(lp + c);
            m[c] = ident ? (c == rr ? 1.0 : 0.0) : v.x;
            m[c + 1] = ident ? (c + 1 == rr ? 1.0 : 0.0) : v.y;
        }
        // columns past n (last panel only) become a harmless decoupled block: huge diagonal, zero elsewhere
#pragma unroll
        for (int c = 0; c < kNB; ++c)
            if (c >= ncol) m[c] = (lane < kNB && c == rr) ? 1e300 : 0.0;
    }
    double inv = fast_rcp(bcast_lane(m[0], 0));
#pragma unroll
    for (int jj = 0; jj < kNB; ++jj) {
        const double mr = m[jj] * inv;
        if (jj + 1 < kNB) {
            m[jj + 1] = fma(-mr, bcast_lane(m[jj], jj + 1), m[jj + 1]);
            inv = fast_rcp(bcast_lane(m[jj + 1], jj + 1));  // next pivot: in flight during the bulk update
        }
        if (jj + 2 < kNB) {
            double* col = (jj & 1) ? colB : colA;
            col[lane] = m[jj];
#pragma unroll
            for (int c = jj + 2; c < kNB; ++c) m[c] = fma(-mr, col[c], m[c]);
        }
    }
    bool bad = false;
    double out[kNB];
#pragma unroll
    for (int c = 0; c < kNB; ++c) {
        const double d = bcast_lane(m[c], c);
        const bool pos = (d > 0.0) & (d < __builtin_inf());  // false for NaN; bitwise: no short-circuit branches
        bad |= (c < ncol) & !pos;
        out[c] = m[c] * fast_rsqrt(((c < ncol) & pos) ? d : 1.0);
    }
    // lanes 32..63 own the output rows: L(i,k), R(r,k) or R(k,k).  Of the diagonal tile only rows >= n (the rhs
    // row when it lives in the last diagonal tile) are results: y = L^-1 b.
    double* rowp;
    if (lane < kNB) rowp = A + (size_t)(c0 + rr) * ld + c0;
    else if (isDiag) rowp = R + (size_t)(c0 + rr) * ld + c0;
    else rowp = own + (size_t)(kNB * i + rr) * ld + c0;
    if (ncol == kNB) {
        if (lane >= kNB) {
            double2* wp2 = reinterpret_cast<double2*>(rowp);
#pragma unroll
            for (int c = 0; c < kNB; c += 2) wp2[c / 2] = make_double2(out[c], out[c + 1]);
        }
    } else {
#pragma unroll
        for (int c = 0; c < kNB; ++c)
            if (c < ncol && (lane >= kNB || (isDiag && rr >= ncol))) rowp[c] = out[c];
    }
    if (bad && isDiag && lane == 0) fail[0] = 1.0;
}

// ---------------------------------------------------------------------------------------------
// k_chol_tiles: the whole factorisation in ONE launch - a dataflow over 32x32 tiles (default path).
// The 600-column problem is bound by kernel boundaries and dependent memory latency, not by flops, so each tile (i,j)
// of the lower triangle of A, and each tile (r,j) of R = "L^-T", gets its own persistent workgroup that
//   1. keeps its tile T and a private copy of the diagonal tile D(j,j) in registers and subtracts the contribution of
//      block column m = 0..j-1 as soon as that column is published (left-looking; only m = j-1 is on the critical path),
//   2. eliminates the stacked [D; T] exactly like k_chol_step (wave 0, lane = row),
//   3. publishes its result and raises flag(i,j) (agent-scope release; consumers poll with a bounded spin).
// Tasks are ordered by block column, dependencies point to lower task numbers only, and a workgroup DRAWS its task number
// from a counter when it starts (not its block index): whatever it waits for is already running, whatever order the
// workgroups of the grid are dispatched in.  (A grid of at most half the device's resident workgroups - one window's solve -
// skips the counter: all of it is dispatched in any case.)  A spin that exceeds 2 s would report a failure instead of hanging (reached
// only by the fault-injection test).  The factorisation is kept in LDL^T form, which takes the 32 reciprocal square roots off the critical
// path: with M = the unnormalised elimination result and MR = M * diag(1/pivot),
//      L L^T = MR M^T,    x = R y = MR_R y_un      (no square root anywhere)
// both M (in place in A / R) and MR (in AM / RM) are published.  Flags carry the solve's epoch: no clearing needed.
// ---------------------------------------------------------------------------------------------
// Tile hand-off between workgroups that may sit behind different L2s (one per XCD): agent-scope ("sc1") vector loads
// and write-through stores, so that neither side has to invalidate / write back a whole L2 (which is what an
// acquire / release fence costs, and what every other workgroup on that XCD then pays for).  The flag protocol
// orders them: data stores -> s_waitcnt vmcnt(0) -> flag store;  flag seen -> barrier -> data loads.
typedef double d2_t __attribute__((ext_vector_type(2)));
typedef double d4_t __attribute__((ext_vector_type(4)));
__device__ inline d2_t load_agent(const double* p) {
    d2_t v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// (s_nop: a VMEM store of more than 64 bits reads its data a few cycles after issue, and a VALU write to those VGPRs in
// the next instruction would change what is stored.  The compiler's hazard recogniser does not see through inline asm, so
// the wait states are part of the instruction here - found with a temporary in the upper half of the operand, see the
// express-copy experiment in docs/history/DESIGN_rounds_1-5.md 4.1.1.)
__device__ inline void store_agent(double* p, d2_t v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
#define SE2_WAIT_VM6(a, b, c, d, e, f) \
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : : "memory")

// `lazy`: the tile is not the one this task needs next on the critical path (it belongs to a block column further back
// than the previous one) - poll every ~1.3 us instead of every ~50 ns.  Every waiting task of a solve polls the same few
// flag words with agent-scope loads that go to memory; with 361 tasks (200 key frames) or 64 windows x 35 tasks doing so
// flat out, the polls queue in front of the publishers' stores (measured: a batch of windows solved side by side scaled at
// 3.6 us per window instead of overlapping).
// a flag poll that stays in flight: the caller waits for it (s_waitcnt vmcnt(0) on the result) when it has nothing better to do
__device__ inline unsigned poll_agent(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ inline bool spin_until(const unsigned* f, unsigned epoch, bool lazy = false) {
    if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch) return true;
    const long long t0 = wall_clock64();  // 100 MHz
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
        if (lazy) __builtin_amdgcn_s_sleep(48);
        else __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > 200000000ll) return false;
    }
    return true;
}

// VERIFY (SE2GPU_BA_CHOL_VERIFY=1, the soak tool's mode): every published half-slab (M or MR: 8 columns x 32 rows) travels with
// the XOR of its 256 bit patterns, and every consumer checks what it loaded against it.  A hand-off that delivered a stale
// or torn payload - the flag seen before the payload stores had landed, a line served from a cache that should have been
// bypassed - shows up as a record {epoch, consumer task, kind / tile row / column of the tile, slab, part, consumer XCC,
// got, want} in `vfy` instead of as a last-bit difference in some chi^2 a few hundred operations later.
//   vfy[0] = mismatches, vfy[1] = half-slabs checked, records of 8 words from vfy[8] (the first 64), checksums from vfy[520]:
//   [(kind nt + i) nbc + j][slab][part]
constexpr int kVfyRecords = 64, kVfyChk = 8 + 8 * kVfyRecords;
__device__ inline unsigned long long wave_xor64(unsigned long long v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v ^= (unsigned long long)__shfl_xor((long long)v, m);
    return v;
}
__device__ inline unsigned long long bits4(d2_t a, d2_t b) {
    return (unsigned long long)__double_as_longlong(a.x) ^ (unsigned long long)__double_as_longlong(a.y) ^
           (unsigned long long)__double_as_longlong(b.x) ^ (unsigned long long)__double_as_longlong(b.y);
}
template <bool VERIFY>
__device__ __forceinline__ void d_chol_tiles(const unsigned bx_dispatch, const double* __restrict__ A, double* __restrict__ PUB,
                                                     double* __restrict__ YU, int ld, int n,
                                                     int nbc, const int4* __restrict__ tasks, const int* __restrict__ deps,
                                                     const int* __restrict__ col_src,
                                                     unsigned* __restrict__ flagA, unsigned* __restrict__ flagR,
                                                     const unsigned* __restrict__ epoch_ptr, double* __restrict__ fail,
                                                     long long* __restrict__ dbg, const BaCtl* __restrict__ ctl,
                                                     double* __restrict__ xout, unsigned long long* __restrict__ vfy,
                                                     unsigned long long* __restrict__ head, int ntask) {
    if (ctl && ctl->done) return;   // uniform over the grid: nobody waits for a tile that will not be published
    // Which task a workgroup runs is not its block index but the next number of a counter it draws when it STARTS (round 6): the
    // tasks are listed in topological order, so whatever a task waits for has been drawn by a workgroup that is already running -
    // no assumption about the order in which the hardware dispatches the workgroups of a grid is left (HIP promises none), and the
    // spin time-out below is unreachable by construction (it stays as the guard of the fault-injection test).  Every launch of a
    // handle draws exactly `ntask` numbers (all workgroups pass here or none does); whoever draws the last one puts the counter
    // back to zero for the next launch, which cannot start before this one has ended.
    // head == nullptr: the host has checked that the WHOLE grid fits the device at once (ba_chol_grid_fits) - then every workgroup
    // is dispatched whatever the others wait for, the block index can be the task number, and the first task of the chain does not
    // pay the counter's round trip (2-3 us of a 96 us solve).
    __shared__ unsigned task_s;
    if (head) {
        if (threadIdx.x == 0) {
            const unsigned c = __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(head), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (c == (unsigned)ntask - 1u) __hip_atomic_store(reinterpret_cast<unsigned*>(head), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            task_s = c;
        }
        __syncthreads();
    }
    const unsigned bx = head ? task_s : bx_dispatch;
    const unsigned epoch = *epoch_ptr;   // moved on by the kernel that built this system (k_reduce2 / k3_reduce2)
    const int nt = ld / kNB;
    // LDS: 36 KB per task (was 59), so that more tasks share a CU when a batch of windows is solved side by side
    // (k_batched).  The multiplier columns of the elimination re-use the operand tiles of the update phase: rows 0..15
    // (waves 0, 1) lie over Tc, which nobody reads after the last tile product; rows 16..31 (waves 2, 3) reach into Ta,
    // which every wave reads once more - its share of the finished tile - and reports in `loaded_s`.
    constexpr int kTile = kNB * (kNB + 2);
    __shared__ __attribute__((aligned(16))) double LD[3 * kTile + kNB * kNB + 8 * kNB];
    // order in LDS: Tb | Tc | Ta | COLV | dummy row.  The 32 multiplier columns (32 x 64 doubles, contiguous: immediate
    // offsets in the pivot loop) start where Tc starts and run on into Ta
    double (*Tb)[kNB + 2] = reinterpret_cast<double (*)[kNB + 2]>(LD);              // M(j,m),  then the finished diagonal D
    double (*Tc)[kNB + 2] = reinterpret_cast<double (*)[kNB + 2]>(LD + kTile);      // MR(j,m)
    double (*Ta)[kNB + 2] = reinterpret_cast<double (*)[kNB + 2]>(LD + 2 * kTile);  // MR(i,m), then the finished tile T
    double (*COLV)[kNB] = reinterpret_cast<double (*)[kNB]>(LD + 3 * kTile);        // an eliminated column's values in the D rows
    double* const DUMMY = LD + 3 * kTile + kNB * kNB;                               // where the T rows' lanes write instead
    double (*MRC)[64] = reinterpret_cast<double (*)[64]>(LD + kTile);               // multiplier column of every eliminated column
    static_assert(16 * 64 <= kTile && 32 * 64 <= 2 * kTile, "rows 0..15 of MRC lie inside Tc, all of it inside Tc | Ta");
    __shared__ int ok_s, ready_s, loaded_s, slab_s[kSlabs];
    __shared__ int deps_s[64];   // this task's dependency list (at most 64 tile rows): fetched once, not one global load per column
    const int tid = threadIdx.x;
    if (tid == 0) { ready_s = 0; loaded_s = 0; ok_s = 1; }
    if (tid < kSlabs) slab_s[tid] = 0;
    long long* stamp = dbg ? dbg + (size_t)bx * 16 : nullptr;  // SE2GPU_BA_CHOL_TRACE=1: 100 MHz stamps
    if (stamp && tid == 0) stamp[0] = wall_clock64();
    const int4 tk = tasks[bx];   // {tile row | kind << 16, block column, first, one past the last entry of its dependency list}
    if (tid < tk.w - tk.z && tid < 64) deps_s[tid] = deps[tk.z + tid];
    __syncthreads();
    if ((tk.x >> 16) == 2) {
        // ---- x = R y for the 32 rows of tile row r (what k_chol_apply did as a kernel of its own): the last tasks of the
        // list.  x(r) = sum_{j >= r} MR_R(r,j) y_un(j); the terms are taken as their tiles are published, so that only the last
        // block column is left when the factorisation ends.  y_un(j) = row n of A in the columns of tile j: written by the
        // task of tile (n / 32, j) - the diagonal task when the rhs row lives in the last diagonal tile.
        const int r = tk.x & 0xffff;
        const int it = n / kNB;                       // tile row of the rhs row
        const int row = tid / 8, c4 = (tid % 8) * 4;  // 8 lanes per row, 4 columns each
        double acc = 0.0;
        for (int dq = tk.z; dq < tk.w; ++dq) {   // the block columns j >= r with a non-zero R(r, j), ascending
            const int j = deps_s[dq - tk.z];
            if (tid == 0) {
                const bool lazy = dq + 1 < tk.w;   // only the last term is waited for in earnest
                bool ok = true;
                for (int sl = 0; sl < kSlabs; ++sl) {
                    ok = ok && spin_until(flagR + ((size_t)r * nbc + j) * kSlabs + sl, epoch, lazy);
                    ok = ok && spin_until((it == j ? flagR : flagA) + ((size_t)it * nbc + j) * kSlabs + sl, epoch, lazy);
                }
                ok_s = ok ? 1 : 0;
            }
            __syncthreads();
            if (!ok_s) {
                if (tid == 0) fail[0] = 1e6;
                return;
            }
            const int c0 = kNB * j + c4;
            const double* rt = PUB + pub_tile(1, r, j, nt, nbc) + (c4 >> 3) * kSlabDoubles + kSlabDoubles / 2;   // MR_R(r, j), the slab of these 4 columns
            const int p0 = (c4 & 7) >> 1;
            d2_t m0 = load_agent(rt + (p0 * kNB + row) * 2), m1 = load_agent(rt + ((p0 + 1) * kNB + row) * 2);
            d2_t y0 = load_agent(YU + c0), y1 = load_agent(YU + c0 + 2);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(m0), "+v"(m1), "+v"(y0), "+v"(y1) : : "memory");
            // (columns past n of the last tile: R is zero there by construction of the padded elimination, y is not read)
            if (c0 + 0 < n) acc += m0.x * y0.x;
            if (c0 + 1 < n) acc += m0.y * y0.y;
            if (c0 + 2 < n) acc += m1.x * y1.x;
            if (c0 + 3 < n) acc += m1.y * y1.y;
            __syncthreads();   // ok_s is rewritten in the next round
        }
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        acc += __shfl_xor(acc, 4);
        if ((tid & 7) == 0 && kNB * r + row < n) {
            const int dst = col_src ? col_src[kNB * r + row] : kNB * r + row;   // system column -> 3 * pose + component (< 0: padding)
            if (dst >= 0) xout[dst] = acc;
        }
        return;
    }
    const bool isR = (tk.x >> 16) != 0;
    const int i = tk.x & 0xffff, j = tk.y;
    const bool isDiag = !isR && i == j;
    // The 32x32x32 tile products run on the matrix cores: v_mfma_f64_16x16x4_f64, wave w owns the 16x16 quadrant
    // (w >> 1, w & 1) of T and of D.  Layouts (tools/mfma_probe.hip): A[i][k]: lane = i + 16 k; B[k][j]: lane = j + 16 k;
    // D[i][j]: lane = j + 16 (i % 4), register = i / 4.
    const int wv = tid >> 6, ln = tid & 63;
    const int qi = 16 * (wv >> 1), qj = 16 * (wv & 1);
    const int orow = qi + (ln >> 4), ocol = qj + (ln & 15);   // output element v: (orow + 4 v, ocol)
    const int arow = ln & 15, acol = ln >> 4;                 // operand element of k-chunk ks: (arow, 4 ks + acol)
    d4_t T0 = {0, 0, 0, 0}, D0, accT = {0, 0, 0, 0}, accD = {0, 0, 0, 0};
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        D0[v] = A[(size_t)(kNB * j + orow + 4 * v) * ld + kNB * j + ocol];
        if (!isR && !isDiag) T0[v] = A[(size_t)(kNB * i + orow + 4 * v) * ld + kNB * j + ocol];
    }
    // the block columns m < j with a non-zero L(j, m), ascending (all of them for a dense system); bit 15 of an entry: the
    // task's own tile row has a non-zero tile in that column too (L(i, m) / R(r, m)), otherwise only D is updated
    // A tile travels in the four 8-column slabs its producer's waves finish in (wave w eliminates columns 8 w .. 8 w + 7 and
    // publishes them with a flag of their own while the later waves still work).  Wave w of this task fetches slab w of
    // the operand tiles and stages it in LDS; every wave multiplies the slabs in ascending order as they appear there
    // (the k-chunks in the order of the one-piece version: bit-identical sums).  A wave must not sit in a poll loop for
    // its own slab while earlier slabs wait to be multiplied - the wave of the LAST slab would then do all eight
    // k-chunks after the producer's last pivot -, so fetching and multiplying are two halves of one loop: the flag poll
    // is issued, a ready slab is multiplied while the poll is in flight, then the poll is looked at.  What is left between
    // the producer's last pivot and this task's elimination is the last slab's flag, its loads and two k-chunks.
    const int sr = ln >> 1, sc = 8 * wv + 4 * (ln & 1);   // staging: lane = (row, half of the slab's 8 columns)
    for (int dq = tk.z; dq < tk.w; ++dq) {
        const int dep = deps_s[dq - tk.z];
        const int m = dep & 0x7fff;
        const bool hasT = (dep >> 15) != 0;
        __syncthreads();  // everyone is done with the LDS tiles of the previous column
        if (!ok_s) {
            if (tid == 0) fail[0] = 1e6;
            return;
        }
        const bool lazy = dq + 1 < tk.w;       // not the block column this task's elimination waits for
        const unsigned* f0 = flagA + ((size_t)j * nbc + m) * kSlabs + wv;
        const unsigned* f1 = hasT ? (isR ? flagR : flagA) + ((size_t)i * nbc + m) * kSlabs + wv : f0;
        const int need = dq - tk.z + 1;
        bool staged = false;
        int mult = 0;          // slabs multiplied so far
        long long t0 = 0;
        while (mult < kSlabs) {
            unsigned v0 = 0, v1 = 0;
            if (!staged) { v0 = poll_agent(f0); v1 = poll_agent(f1); }   // in flight during the products below
            bool progress = false;
            if (__hip_atomic_load(&slab_s[mult], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= need) {
                // the slab's two k-chunks: all six operand reads in flight before the first product
                const int k0 = 8 * mult + acol;
                const double b0 = Tb[qj + arow][k0], b1 = Tb[qj + arow][k0 + 4];
                const double c0 = Tc[qi + arow][k0], c1 = Tc[qi + arow][k0 + 4];
                const double a0 = Ta[qi + arow][k0], a1 = Ta[qi + arow][k0 + 4];
                __builtin_amdgcn_sched_barrier(0);
                accD = __builtin_amdgcn_mfma_f64_16x16x4f64(c0, b0, accD, 0, 0, 0);
                if (hasT) accT = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, accT, 0, 0, 0);
                accD = __builtin_amdgcn_mfma_f64_16x16x4f64(c1, b1, accD, 0, 0, 0);
                if (hasT) accT = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, accT, 0, 0, 0);
                ++mult;
                progress = true;
            }
            if (!staged) {
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(v0), "+v"(v1) : : "memory");
                bool here = __builtin_amdgcn_readfirstlane(v0) == epoch && __builtin_amdgcn_readfirstlane(v1) == epoch;
                if (!here && !progress) {   // 2 s without the flag: report, and stage whatever is there so that nobody waits for this slab
                    if (t0 == 0) t0 = wall_clock64();
                    else if (wall_clock64() - t0 > 200000000ll) {
                        if (ln == 0) ok_s = 0;
                        here = true;
                    }
                }
                if (here) {
                    if (stamp && tid == 192) stamp[1] = wall_clock64();
                    // slab wv of L(j, m) - M and MR - and of this task's own tile row in that column (MR); every load instruction
                    // covers two runs of 512 contiguous bytes (column pair, 32 rows)
                    const double* tj = PUB + pub_tile(0, j, m, nt, nbc) + wv * kSlabDoubles + ((2 * (ln & 1)) * kNB + sr) * 2;
                    const double* ti = hasT ? PUB + pub_tile(isR ? 1 : 0, i, m, nt, nbc) + wv * kSlabDoubles + ((2 * (ln & 1)) * kNB + sr) * 2 : tj;
                    d2_t mj0 = load_agent(tj), mj1 = load_agent(tj + 2 * kNB);
                    d2_t rj0 = load_agent(tj + kSlabDoubles / 2), rj1 = load_agent(tj + kSlabDoubles / 2 + 2 * kNB);
                    d2_t ri0 = load_agent(ti + kSlabDoubles / 2), ri1 = load_agent(ti + kSlabDoubles / 2 + 2 * kNB);
                    SE2_WAIT_VM6(mj0, mj1, rj0, rj1, ri0, ri1);
                    if constexpr (VERIFY) {
                        const unsigned long long got[3] = {wave_xor64(bits4(mj0, mj1)), wave_xor64(bits4(rj0, rj1)), wave_xor64(bits4(ri0, ri1))};
                        const size_t tjx = ((size_t)(0 * nt + j) * nbc + m) * kSlabs + wv;
                        const size_t tix = ((size_t)((hasT && isR ? 1 : 0) * nt + (hasT ? i : j)) * nbc + m) * kSlabs + wv;
                        const unsigned long long* want_p[3] = {vfy + kVfyChk + 2 * tjx, vfy + kVfyChk + 2 * tjx + 1, vfy + kVfyChk + 2 * tix + 1};
                        if (ln == 0) {
                            for (int part = 0; part < 3; ++part) {
                                const unsigned long long want = __hip_atomic_load(want_p[part], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                atomicAdd(vfy + 1, 1ull);
                                if (want != got[part]) {
                                    const unsigned long long k = atomicAdd(vfy, 1ull);
                                    if (k < (unsigned long long)kVfyRecords) {
                                        unsigned long long* r = vfy + 8 + 8 * k;
                                        unsigned xcc;
                                        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                                        r[0] = epoch; r[1] = bx;
                                        r[2] = ((unsigned long long)(part == 2 ? (isR ? 1 : 0) : 0) << 32) | ((unsigned long long)(part == 2 ? i : j) << 16) | (unsigned)m;
                                        r[3] = ((unsigned long long)wv << 8) | (unsigned)part;
                                        r[4] = xcc & 0xf; r[5] = got[part]; r[6] = want; r[7] = (unsigned long long)wall_clock64();
                                    }
                                }
                            }
                        }
                    }
                    Ta[sr][sc] = ri0.x; Ta[sr][sc + 1] = ri0.y; Ta[sr][sc + 2] = ri1.x; Ta[sr][sc + 3] = ri1.y;
                    Tb[sr][sc] = mj0.x; Tb[sr][sc + 1] = mj0.y; Tb[sr][sc + 2] = mj1.x; Tb[sr][sc + 3] = mj1.y;
                    Tc[sr][sc] = rj0.x; Tc[sr][sc + 1] = rj0.y; Tc[sr][sc + 2] = rj1.x; Tc[sr][sc + 3] = rj1.y;
                    // (LDS operations of a wave execute in issue order: the count is behind the writes)
                    asm volatile("" ::: "memory");
                    if (ln == 0) __hip_atomic_fetch_add(&slab_s[wv], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (stamp && tid == 192) stamp[7] = wall_clock64();
                    staged = true;
                    progress = true;
                }
            }
            if (!progress) {
                if (staged) __builtin_amdgcn_s_sleep(1);
                else if (lazy) __builtin_amdgcn_s_sleep(48);
                else __builtin_amdgcn_s_sleep(2);
            }
        }
    }
    __syncthreads();
    if (!ok_s) {
        if (tid == 0) fail[0] = 1e6;
        return;
    }
    if (stamp && tid == 0) stamp[2] = wall_clock64();
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        Ta[orow + 4 * v][ocol] = T0[v] - accT[v];
        Tb[orow + 4 * v][ocol] = D0[v] - accD[v];
    }
    __syncthreads();
    // ---- eliminate the stacked [D; T] (lane = row): the 32 columns are split over the 4 waves, 8 each.  Wave w first
    // applies the rank-1 updates of the 8 w columns owned by the waves before it, as their owners publish them in LDS
    // (multiplier column MRC[jj][row], pivot-row values COLV[jj][c] and a progress counter), then eliminates its own 8
    // columns (pivot path through v_readlane, reciprocal of the next pivot in flight during the update).  The chain
    // is still 32 pivots long, but the 465 rank-1 column updates that bounded the single-wave version are spread over
    // four SIMDs and the early waves publish their part of the tile while the later ones still work.
    const int lane = ln;
    const int w = __builtin_amdgcn_readfirstlane(wv);
    const int cb = 8 * w;
    const int c0 = kNB * j;
    const int ncol = min(kNB, n - c0);
    const int rr = lane & 31;
    double m[8], mrs[8];
    {
        const double* lp = (lane < kNB) ? &Tb[rr][cb] : &Ta[rr][cb];
        const bool ident = isDiag && lane >= kNB;
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
            const double2 v = *reinterpret_cast<const double2*>(lp + q);
            m[q] = ident ? (cb + q == rr ? 1.0 : 0.0) : v.x;
            m[q + 1] = ident ? (cb + q + 1 == rr ? 1.0 : 0.0) : v.y;
        }
        if (ncol < kNB) {  // last panel only: columns past n become a decoupled block (huge diagonal, zero elsewhere)
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (cb + q >= ncol) m[q] = (lane < kNB && cb + q == rr) ? 1e300 : 0.0;
        }
        // this wave has read its share of Ta / Tb (LDS operations of a wave execute in issue order: the reads above are
        // ahead of this add) - waves 2 and 3 wait for all four before their multiplier columns overwrite Ta
        if (lane == 0) __hip_atomic_fetch_add(&loaded_s, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    long long clk0 = 0;
    if (stamp) clk0 = clock64();
    if (stamp && tid == 0) stamp[3] = wall_clock64();
    // published columns are consumed four at a time (cb is a multiple of 8): one poll and one LDS latency per batch, all
    // 20 reads in flight before the first FMA needs one - a consumer that polled per column could not keep up with
    // the ~200 clk per column of the producing wave
    for (int done = 0; done < cb; done += 4) {
        while (__hip_atomic_load(&ready_s, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < done + 4)
            __builtin_amdgcn_s_sleep(1);
        double mrv[4], cv[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            mrv[u] = MRC[done + u][lane];
#pragma unroll
            for (int q = 0; q < 8; ++q) cv[u][q] = COLV[done + u][cb + q];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int q = 0; q < 8; ++q) m[q] = fma(-mrv[u], cv[u][q], m[q]);
        }
    }
    // Every instruction of this wave is issued in order at ~8 clk, so the cost of a pivot column is its instruction
    // count: one Newton step on v_rcp_f64 (relative error 2e-15, used consistently for M and MR, i.e. a 2e-15 relative
    // perturbation of the pivots of an LDL^T whose rounding errors are larger), pivot test folded into one running
    // minimum, LDS rows addressed with immediate offsets.
    if (w >= 2)
        while (__hip_atomic_load(&loaded_s, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 4) __builtin_amdgcn_s_sleep(1);
    double* mrc = &MRC[cb][lane];
    double* colv = lane < kNB ? &COLV[cb][lane] : DUMMY + (lane - kNB);   // (same row stride: immediate offsets below)
    double pmin = 1e300, mr_prev = 0.0, rvb[2][8];   // rvb: double buffer (static indices: the loop is unrolled)
    // The multiplier comes straight from the reciprocal seed x0 and the Newton residual e = 1 - piv x0,
    //     mr = m x0 (1 + e)        (= m / piv up to e^2 ~ 2e-15, like one Newton step on x0 first)
    // which takes the refined reciprocal - one dependent FP64 operation of 44 clk - off the pivot chain
    // (tools/fp64_issue_probe.hip; the refined-reciprocal variant measured the same and was removed in round 4).
    double piv = bcast_lane(m[0], cb);
    double x0 = __builtin_amdgcn_rcp(piv), e = fma(-piv, x0, 1.0);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int jj = cb + q;
        pmin = fmin(pmin, piv);  // (ignores NaN: caught below)
        const double mr0 = m[q] * x0;
        const double mr = fma(mr0, e, mr0);
        mrs[q] = mr;
        mrc[q * 64] = mr;        // later waves need this column (the last wave's copy is never read)
        colv[q * kNB] = m[q];
        // LDS operations of one wave execute in issue order, so the counter needs no s_waitcnt in front of it (that
        // wait would sit on the pivot chain) - only the compiler has to keep the three writes in this order
        asm volatile("" ::: "memory");
        __hip_atomic_store(&ready_s, jj + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
        // This column's pivot-row values for the wave's remaining columns come back from the LDS row just written
        // (uniform-address reads = broadcast, in order behind the write) and are applied ONE column later, when they
        // have long arrived: 2 v_readlane per element would otherwise be 40 % of the instructions of this section.
#pragma unroll
        for (int q2 = q + 2; q2 < 8; ++q2) rvb[q & 1][q2] = COLV[jj][cb + q2];
        if (q > 0) {
#pragma unroll
            for (int q2 = q + 1; q2 < 8; ++q2) m[q2] = fma(-mr_prev, rvb[(q & 1) ^ 1][q2], m[q2]);
        }
        if (q + 1 < 8) {
            m[q + 1] = fma(-mr, bcast_lane(m[q], jj + 1), m[q + 1]);
            piv = bcast_lane(m[q + 1], jj + 1);
            // next pivot: in flight during the rest of the update
            x0 = __builtin_amdgcn_rcp(piv);
            e = fma(-piv, x0, 1.0);
        }
        mr_prev = mr;
    }
    // pivots must be positive and finite (pad pivots are ~1e300)
    // (the pivot row's own multiplier is pivot * inv: NaN exactly when this or an earlier pivot was NaN)
    const double chk = bcast_lane(mrs[7], cb + 7);
    const bool bad = !(pmin > 0.0) | !(pmin < __builtin_inf()) | !(chk == chk);
    if (stamp && tid == 224) stamp[4] = wall_clock64() + (long long)(m[7] == 1.2345e-300);
    if (stamp && tid == 224) stamp[6] = clock64() - clk0;
    // lanes 32..63 own the output rows (diagonal task: R(j,j) from the identity).  The tile is not written back into the
    // matrix but into the publish buffer, slab by slab: slab w = this wave's 8 columns of M, then of MR, each as four
    // column pairs x 32 rows x 16 bytes - one store instruction fills 512 contiguous bytes (four whole lines).  In the
    // row-major matrix the same instruction touched 32 lines with 16 bytes each; partial-line write-through stores take
    // longer to land, and the hand-off is a store latency, a flag latency and a load latency (tools/xcd_probe.hip: 1.7-1.9 us
    // per hand-off with the strided pattern, 1.2-1.3 us with whole lines).
    if (lane >= kNB) {
        double* pb = PUB + pub_tile(isR || isDiag ? 1 : 0, i, j, nt, nbc) + w * kSlabDoubles + rr * 2;
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
            // write-through stores for every wave: no L2 write-back (release fence) anywhere in the solve - a fence walks
            // the whole L2 of its XCD, and with 64 windows' tasks fencing side by side the L2s did little else
            store_agent(pb + q * kNB, d2_t{m[q], m[q + 1]});
            store_agent(pb + kSlabDoubles / 2 + q * kNB, d2_t{mrs[q], mrs[q + 1]});
        }
    }
    if constexpr (VERIFY) {   // the checksums of this slab's two halves, in front of the flag like the payload
        unsigned long long xm = 0, xr = 0;
        if (lane >= kNB) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                xm ^= (unsigned long long)__double_as_longlong(m[q]);
                xr ^= (unsigned long long)__double_as_longlong(mrs[q]);
            }
        }
        xm = wave_xor64(xm);
        xr = wave_xor64(xr);
        if (lane == 0) {
            const size_t tx = ((size_t)((isR || isDiag ? 1 : 0) * nt + i) * nbc + j) * kSlabs + w;
            __hip_atomic_store(vfy + kVfyChk + 2 * tx, xm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(vfy + kVfyChk + 2 * tx + 1, xr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // y_un = the rhs row after the elimination of this block column: a T row of the task in its tile row, or a row of the
    // diagonal tile itself when the system does not end on a tile boundary; read by the x tasks of this launch
    if (!isR && i == n / kNB && lane == (isDiag ? 0 : kNB) + n % kNB) {
#pragma unroll
        for (int q = 0; q < 8; q += 2) store_agent(YU + c0 + cb + q, d2_t{m[q], m[q + 1]});
    }
    if (bad && isDiag && lane == 0) fail[0] = 1.0;
    // slab w is this wave's alone: its write-through stores have landed -> its flag, no workgroup barrier in between
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0)
        __hip_atomic_store((isR || isDiag ? flagR : flagA) + ((size_t)i * nbc + j) * kSlabs + w, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (stamp && lane == 0) stamp[12 + w] = wall_clock64();
    if (stamp && tid == 192) stamp[5] = wall_clock64();
}
template <bool VERIFY>
__global__ __launch_bounds__(256) void k_chol_tiles(const double* __restrict__ A, double* __restrict__ PUB,
                                                     double* __restrict__ YU, int ld, int n,
                                                     int nbc, const int4* __restrict__ tasks, const int* __restrict__ deps,
                                                     const int* __restrict__ col_src,
                                                     unsigned* __restrict__ flagA, unsigned* __restrict__ flagR,
                                                     const unsigned* __restrict__ epoch_ptr, double* __restrict__ fail,
                                                     long long* __restrict__ dbg, const BaCtl* __restrict__ ctl,
                                                     double* __restrict__ xout, unsigned long long* __restrict__ vfy,
                                                     unsigned long long* __restrict__ head, int ntask) {
    d_chol_tiles<VERIFY>(blockIdx.x, A, PUB, YU, ld, n, nbc, tasks, deps, col_src, flagA, flagR, epoch_ptr, fail, dbg, ctl, xout, vfy, head, ntask);
}

// x = R y  (R = L^-T upper triangular, y = augmented row n of A).  One wave per row.
__global__ __launch_bounds__(256) void k_chol_apply(const double* __restrict__ A, const double* __restrict__ R, int ld,
                                                     int n, double* __restrict__ x, const BaCtl* __restrict__ ctl,
                                                     const int* __restrict__ col_src) {
    if (ctl && ctl->done) return;
    const int r = blockIdx.x * 4 + threadIdx.x / 64;
    const int lane = threadIdx.x & 63;
    if (r >= n) return;
    const double* y = A + (size_t)n * ld;
    // R(r, c) is defined for c >= kNB * (r / kNB); inside the diagonal tile only c >= r is non-zero
    double acc = 0.0;
    for (int c = r + lane; c < n; c += 64) acc += R[(size_t)r * ld + c] * y[c];
    acc = wave_sum(acc);
    const int dst = col_src ? col_src[r] : r;   // permuted system: column -> 3 * pose + component, < 0 = padding
    if (lane == 0 && dst >= 0) x[dst] = acc;
}

// The end of a trial without a kernel boundary (single GPU): k_update gets ONE extra workgroup, launched last, that does
// what k_finalize does - oplus of the poses, odometry chi^2 at the trial poses, the pose part of computeScale() - while
// the landmark workgroups run, then waits until their partials have all arrived (an agent-scope counter; it was
// dispatched after every one of them, so the wait cannot deadlock), sums them in a fixed order and advances the
// Levenberg-Marquardt controller.  k_finalize stays for the sharded (multi-GPU) runs, whose scalars go through an
// all-reduce between the two steps.
struct FinArgs {
    int enabled, nblk, P, O, root, step, decide, notify;
    const uint8_t* fixed;
    const double* xp;
    const double* bp;
    const double* poses;      // "a" buffer (the controller's sel bit says which one holds the estimate)
    double* poses_trial;      // "b" buffer
    const int* o_i;
    const int* o_j;
    const double* o_meas;
    const double* o_info;
    double* out;              // {chi2_trial, scale, factorisation flag, 0}
    volatile double* mail;
    double seq;
    BaCtl* ctl;
    const volatile int* stop;
    unsigned* counter;        // landmark workgroups that have published their partials (reset by the finisher)
};
__device__ void finish_trial(const FinArgs& fin, const double* part, double lambda);

// ---------------------------------------------------------------------------------------------
// k_update: per landmark group: back-substitute x_l = A^T (zeta_l - sum_e W_e^T x_p[kf(e)]) (the whitened records of
// k_linearize: = z_l - sum_e Y_e^T x_p), trial landmark = lw + x_l,
// robust chi^2 of the landmark's edges at the trial state, and the landmark part of computeScale().
// With xp == nullptr it evaluates chi^2 at the current state (x = 0).  Per-block partials -> part[2*blockIdx].
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void d_update(const unsigned bx, CamDev cam, int L, double lambda, const int* __restrict__ lm_ptr,
                                                    const int* __restrict__ e_kf, const double* __restrict__ e_uv,
                                                    const double* __restrict__ e_info,
                                                    const double* __restrict__ poses,
                                                    const uint8_t* __restrict__ fixed, const double* __restrict__ lms,
                                                    const double* __restrict__ xp, const double* __restrict__ zeta,
                                                    const double* __restrict__ W, const double* __restrict__ bl,
                                                    double* __restrict__ lms_trial, double* __restrict__ part,
                                                    const BaCtl* __restrict__ ctl, const double* __restrict__ poses_b,
                                                    FinArgs fin, const double* __restrict__ Ainv) {
    if (fin.enabled && (int)bx == fin.nblk) {
        finish_trial(fin, part, lambda);
        return;
    }
    // The kernel is a chain of dependent loads (controller / CSR bounds -> edge records -> pose gathers) around very
    // little arithmetic, so everything that does not depend on the previous link is requested together: the CSR bounds
    // beside the controller block, and for the first two edges of a lane (a landmark has 6 observations on average, a
    // lane takes every 8th) ALL operands of both passes - W_e and x_p for the back-substitution, the pose, measurement and
    // information for the robust chi^2 - before the first use.  Lanes with more edges take the rest in the old two-pass form.
    const int gid = bx * kBlock + threadIdx.x;
    const int l = gid / kGroup, sub = gid % kGroup;
    int beg = 0, end = 0;
    if (l < L) {
        beg = lm_ptr[l];
        end = lm_ptr[l + 1];
    }
    if (ctl) {   // (poses, lms) / lms_trial are the "a" / "b" buffers: the controller says which holds the estimate
        if (ctl->done) return;
        if (ctl->sel) {
            const double* t = lms; lms = lms_trial; lms_trial = const_cast<double*>(t);
            poses = poses_b;
        }
        lambda = ctl->lambda;
    }
    __shared__ double sm[2][kBlock / 64];
    double chi = 0, scale = 0;
    double x[3] = {0, 0, 0};
    const bool step = xp != nullptr;
    constexpr int KS = 2;
    int ekf[KS];
    bool ev[KS];
    double ey[KS][9], ep[KS][3], epose[KS][3], euv[KS][2], ew[KS][3];
    bool efix[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k) {
        const int e = beg + sub + k * kGroup;
        ev[k] = e < end;
        const int ec = ev[k] ? e : max(end - 1, 0);   // (a landmark without edges loads edge 0 of the array: harmless)
        const bool has = l < L && end > beg;
        ekf[k] = has ? e_kf[ec] : 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) ey[k][i] = (has && step) ? W[(size_t)ec * 9 + i] : 0.0;
        euv[k][0] = has ? e_uv[2 * (size_t)ec] : 0.0;
        euv[k][1] = has ? e_uv[2 * (size_t)ec + 1] : 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) ew[k][i] = has ? e_info[3 * (size_t)ec + i] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < KS; ++k) {
        const int kf = ekf[k];
        epose[k][0] = poses[3 * kf]; epose[k][1] = poses[3 * kf + 1]; epose[k][2] = poses[3 * kf + 2];
        efix[k] = fixed[kf] != 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) ep[k][i] = step ? xp[3 * kf + i] : 0.0;
    }
    double lw0[3] = {0, 0, 0}, zz[3] = {0, 0, 0}, blv[3] = {0, 0, 0}, av[6] = {0, 0, 0, 0, 0, 0};
    if (l < L) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            lw0[c] = lms[(size_t)l * 3 + c];
            if (step) { zz[c] = zeta[(size_t)l * 3 + c]; blv[c] = bl[(size_t)l * 3 + c]; }
        }
        if (step) {
#pragma unroll
            for (int c = 0; c < 6; ++c) av[c] = Ainv[(size_t)l * 6 + c];
        }
    }
    if (l < L && step) {
#pragma unroll
        for (int k = 0; k < KS; ++k)
            if (ev[k]) {
#pragma unroll
                for (int c = 0; c < 3; ++c) x[c] -= ey[k][c] * ep[k][0] + ey[k][3 + c] * ep[k][1] + ey[k][6 + c] * ep[k][2];
            }
        for (int e = beg + sub + KS * kGroup; e < end; e += kGroup) {
            const int kf = e_kf[e];
            const double* y = W + (size_t)e * 9;
            const double p0 = xp[3 * kf], p1 = xp[3 * kf + 1], p2 = xp[3 * kf + 2];
#pragma unroll
            for (int c = 0; c < 3; ++c) x[c] -= y[c] * p0 + y[3 + c] * p1 + y[6 + c] * p2;
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) x[c] = group_sum(x[c]);
    if (l < L && step) {   // x holds -sum_e W_e^T dp_e:  x_l = A^T (zeta + x)
        const double t0 = zz[0] + x[0], t1 = zz[1] + x[1], t2 = zz[2] + x[2];
        x[0] = av[0] * t0 + av[1] * t1 + av[3] * t2;
        x[1] = av[2] * t1 + av[4] * t2;
        x[2] = av[5] * t2;
    }
    if (l < L) {
        double lw[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) lw[c] = lw0[c] + x[c];
        if (sub == 0 && step) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                lms_trial[(size_t)l * 3 + c] = lw[c];
                scale += x[c] * (lambda * x[c] + blv[c]);
            }
        }
        auto edge_chi = [&](double px, double py, double pth, bool fx, const double* dp, double u, double v, double w0,
                            double w1, double w2) {
            if (step && !fx) {
                px += dp[0];
                py += dp[1];
                pth = normalize_theta(pth + dp[2]);
            }
            double e0, e1;
            se2xyz<false>(cam, px, py, pth, lw[0], lw[1], lw[2], u, v, e0, e1, nullptr, nullptr);
            double r0, r1;
            huber(e0 * (w0 * e0 + w1 * e1) + e1 * (w1 * e0 + w2 * e1), cam.huber, r0, r1);
            chi += r0;
        };
#pragma unroll
        for (int k = 0; k < KS; ++k)
            if (ev[k]) edge_chi(epose[k][0], epose[k][1], epose[k][2], efix[k], ep[k], euv[k][0], euv[k][1], ew[k][0], ew[k][1], ew[k][2]);
        for (int e = beg + sub + KS * kGroup; e < end; e += kGroup) {
            const int kf = e_kf[e];
            double dp[3] = {0, 0, 0};
            if (step) { dp[0] = xp[3 * kf]; dp[1] = xp[3 * kf + 1]; dp[2] = xp[3 * kf + 2]; }
            edge_chi(poses[3 * kf], poses[3 * kf + 1], poses[3 * kf + 2], fixed[kf] != 0, dp, e_uv[2 * e], e_uv[2 * e + 1],
                     e_info[3 * e], e_info[3 * e + 1], e_info[3 * e + 2]);
        }
    }
    chi = wave_sum(chi);
    scale = wave_sum(scale);
    const int w = threadIdx.x / 64;
    if ((threadIdx.x & 63) == 0) { sm[0][w] = chi; sm[1][w] = scale; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double c = 0, s = 0;
        for (int i = 0; i < kBlock / 64; ++i) { c += sm[0][i]; s += sm[1][i]; }
        if (fin.enabled) {   // write-through stores, landed before the counter moves (the finisher may sit behind another L2)
            store_agent(part + 2 * bx, d2_t{c, s});
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(fin.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            part[2 * bx] = c;
            part[2 * bx + 1] = s;
        }
    }
}
__global__ __launch_bounds__(kBlock) void k_update(CamDev cam, int L, double lambda, const int* __restrict__ lm_ptr,
                                                    const int* __restrict__ e_kf, const double* __restrict__ e_uv,
                                                    const double* __restrict__ e_info,
                                                    const double* __restrict__ poses,
                                                    const uint8_t* __restrict__ fixed, const double* __restrict__ lms,
                                                    const double* __restrict__ xp, const double* __restrict__ zeta,
                                                    const double* __restrict__ W, const double* __restrict__ bl,
                                                    double* __restrict__ lms_trial, double* __restrict__ part,
                                                    const BaCtl* __restrict__ ctl, const double* __restrict__ poses_b,
                                                    FinArgs fin, const double* __restrict__ Ainv) {
    d_update(blockIdx.x, cam, L, lambda, lm_ptr, e_kf, e_uv, e_info, poses, fixed, lms, xp, zeta, W, bl, lms_trial, part, ctl, poses_b, fin, Ainv);
}


// a dataflow solve timed out and the host switched the handle to the per-column launches: the trial is simply redone
__global__ void k_ctl_clear_error(BaCtl* __restrict__ ctl) {
    ctl->error = 0;
    ctl->done = 0;
}

// The controller block goes to the host through a mapped, coherent mailbox (no stream synchronise + D2H copy):
// payload first, system-scope fence, sequence number last.
__device__ inline void post_ctl(const BaCtl* c, volatile double* mail, double seq, int tid, int nthr) {
    constexpr int kWords = (int)(sizeof(BaCtl) / 8);
    const double* src = reinterpret_cast<const double*>(c);
    for (int i = tid; i < kWords; i += nthr) mail[8 + i] = src[i];
}

// One trial slot has finished (or was skipped because the run is over): the device-side slot counter moves on and, when
// asked for, the controller block goes to the host mailbox with the counter as its sequence number.  All threads of the
// (single) deciding workgroup call it; `post` is uniform.
// Only the evaluation of a trial STEP ends a slot: the first slot of a run also evaluates the starting state (step == 0),
// which neither counts nor posts (if that evaluation ends the run - stop flag, zero iterations - the step evaluation of
// the same slot finds `done` and answers).
__device__ inline void end_slot(BaCtl* ctl, volatile double* mail, bool post, int tid, int nthr, bool step) {
    if (!step) return;
    if (tid == 0) ctl->seq += 1.0;
    __syncthreads();
    if (post && mail) {
        __threadfence();
        post_ctl(ctl, mail, 0.0, tid, nthr);
        __threadfence_system();
        __syncthreads();
        if (tid == 0) mail[kMailSeq] = ctl->seq;
    }
}

// the finisher workgroup of k_update (kBlock threads): see FinArgs
__device__ void finish_trial(const FinArgs& fin, const double* part, double lambda) {
    __shared__ double fsm[2][kBlock / 64];
    __shared__ double fsp[3 * 1024];  // trial poses staged for the odometry pass when P <= 1024
    __shared__ int fpost;
    BaCtl* ctl = fin.ctl;
    const volatile double* cmail = fin.mail;
    volatile double* mail = fin.mail;
    (void)cmail;
    const double* poses = fin.poses;
    double* poses_trial = fin.poses_trial;
    const bool step = fin.step != 0;
    int stopped = 0;
    double failflag = 0;
    if (ctl) {
        if (ctl->done) {   // the run is over: only answer a pending notification
            end_slot(ctl, mail, fin.notify != 0, threadIdx.x, blockDim.x, fin.step != 0);
            return;
        }
        if (ctl->sel) { const double* t = poses; poses = poses_trial; poses_trial = const_cast<double*>(t); }
        lambda = ctl->lambda;
    }
    if (threadIdx.x == 0) {   // the two slow reads of the decision (mapped host memory; the solver's flag) start now
        stopped = (fin.stop && *fin.stop) ? 1 : 0;
        failflag = step ? fin.out[2] : 0.0;
    }
    double chi = 0, scale = 0;
    for (int p = threadIdx.x; p < fin.P; p += blockDim.x) {
        double x = poses[3 * p], y = poses[3 * p + 1], th = poses[3 * p + 2];
        if (step && !fin.fixed[p]) {
            const double d0 = fin.xp[3 * p], d1 = fin.xp[3 * p + 1], d2 = fin.xp[3 * p + 2];
            x += d0; y += d1; th = normalize_theta(th + d2);
            const double* bp = fin.bp;
            if (fin.root) scale += d0 * (lambda * d0 + bp[3 * p]) + d1 * (lambda * d1 + bp[3 * p + 1]) + d2 * (lambda * d2 + bp[3 * p + 2]);
            else scale += d0 * bp[3 * p] + d1 * bp[3 * p + 1] + d2 * bp[3 * p + 2];
        }
        if (step) { poses_trial[3 * p] = x; poses_trial[3 * p + 1] = y; poses_trial[3 * p + 2] = th; }
        if (p < 1024) { fsp[3 * p] = x; fsp[3 * p + 1] = y; fsp[3 * p + 2] = th; }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < fin.O; k += blockDim.x) {
        const int i = fin.o_i[k], j = fin.o_j[k];
        double pi[3], pj[3];
        for (int c = 0; c < 3; ++c) {
            pi[c] = (i < 1024) ? fsp[3 * i + c] : (step ? poses_trial[3 * i + c] : poses[3 * i + c]);
            pj[c] = (j < 1024) ? fsp[3 * j + c] : (step ? poses_trial[3 * j + c] : poses[3 * j + c]);
        }
        double e[3], A[9], B[9];
        pre_se2(pi, pj, fin.o_meas + 3 * k, e, A, B);
        const double* W = fin.o_info + 9 * k;
        for (int r = 0; r < 3; ++r) chi += e[r] * (W[r * 3] * e[0] + W[r * 3 + 1] * e[1] + W[r * 3 + 2] * e[2]);
    }
    // the landmark workgroups' partials
    if (threadIdx.x == 0) {
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(fin.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)fin.nblk) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 200000000ll) break;   // 2 s: cannot happen (in-order dispatch); never hang
        }
        __hip_atomic_store(fin.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    for (int i0 = threadIdx.x; i0 < fin.nblk; i0 += 4 * blockDim.x) {   // four loads in flight per thread and round
        d2_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = load_agent(part + 2 * min(i0 + u * (int)blockDim.x, fin.nblk - 1));
        // (the asm loads are invisible to the compiler's wait counts)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : : "memory");
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * (int)blockDim.x < fin.nblk) { chi += v[u].x; scale += v[u].y; }
    }
    chi = wave_sum(chi);
    scale = wave_sum(scale);
    if ((threadIdx.x & 63) == 0) {
        fsm[0][threadIdx.x >> 6] = chi;
        fsm[1][threadIdx.x >> 6] = scale;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {   // fixed order: deterministic
            fsm[0][0] += fsm[0][w];
            fsm[1][0] += fsm[1][w];
        }
        double* out = fin.out;
        out[0] = fsm[0][0]; out[1] = fsm[1][0]; out[3] = 0;
        if (!step) out[2] = 0;
        fpost = 0;
        if (ctl && fin.decide) {
            if (!step) {   // evaluation of the starting state
                ctl->current_chi = ctl->chi2_init = ctl->chi2_final = fsm[0][0];
                if (stopped) { ctl->stopped = 1; ctl->done = 1; }
                if (ctl->iters <= 0) ctl->done = 1;
            } else {
                const double sc[3] = {fsm[0][0], fsm[1][0], failflag};
                lm_advance(ctl, sc, stopped != 0);
            }
            fpost = (ctl->done || fin.notify) && mail;
        } else if (mail && !ctl) {  // synchronous callers (se2gpu_ba_chi2, the host controller): the three scalars
            mail[0] = fsm[0][0];
            mail[1] = fsm[1][0];
            mail[2] = step ? failflag : 0.0;
            __threadfence_system();
            mail[3] = fin.seq;
        }
    }
    if (ctl && fin.decide) {
        __syncthreads();
        end_slot(ctl, mail, fpost != 0, threadIdx.x, blockDim.x, step);
    }
}

// k_finalize: single block.  Sums the k_update partials, applies oplus to the poses (VertexSE2::oplusImpl:
// additive x,y; theta = normalize_theta(theta + dtheta)), adds the odometry chi^2 at the trial poses and the pose
// part of computeScale().  out[0] = chi2_trial, out[1] = scale (local to this rank), out[2] = factorisation flag (already
// there).  Single GPU with a controller: also advances the LM state (decide != 0) and posts it when the run has ended
// or the host asked for it (notify).  step == 0 evaluates the current state (x = 0) and seeds the controller's chi^2.
__global__ void k_finalize(int nparts, const double* __restrict__ part, int P, double lambda,
                           const double* __restrict__ poses, const uint8_t* __restrict__ fixed,
                           const double* __restrict__ xp, const double* __restrict__ bp, double* __restrict__ poses_trial,
                           int O, const int* __restrict__ o_i, const int* __restrict__ o_j,
                           const double* __restrict__ o_meas, const double* __restrict__ o_info, int root,
                           double* __restrict__ out, volatile double* __restrict__ mail, double seq,
                           BaCtl* __restrict__ ctl, int step_arg, int decide, int notify,
                           const volatile int* __restrict__ stop) {
    __shared__ double sm[2][16];
    __shared__ double sp[3 * 1024];  // trial poses staged for the odometry pass when P <= 1024
    __shared__ int post_s;
    bool step = xp != nullptr;
    if (ctl) {
        step = step_arg != 0;
        if (ctl->done) {   // the run is over: only answer a pending notification (sharded runs: k_lm_decide does)
            if (decide) end_slot(ctl, mail, notify != 0, threadIdx.x, blockDim.x, step);
            return;
        }
        if (ctl->sel) { const double* t = poses; poses = poses_trial; poses_trial = const_cast<double*>(t); }
        lambda = ctl->lambda;
    }
    double chi = 0, scale = 0;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) {
        chi += part[2 * i];
        scale += part[2 * i + 1];
    }
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
        double x = poses[3 * p], y = poses[3 * p + 1], th = poses[3 * p + 2];
        if (step && !fixed[p]) {
            const double d0 = xp[3 * p], d1 = xp[3 * p + 1], d2 = xp[3 * p + 2];
            x += d0; y += d1; th = normalize_theta(th + d2);
            // computeScale(): sum_j x_j (lambda x_j + b_j).  bp is this rank's LOCAL pose gradient (its landmark shard's
            // edges, + the odometry edges on the root): every rank contributes x.b_local to the all-reduced sum, the
            // lambda x.x term enters exactly once (root).
            if (root) scale += d0 * (lambda * d0 + bp[3 * p]) + d1 * (lambda * d1 + bp[3 * p + 1]) + d2 * (lambda * d2 + bp[3 * p + 2]);
            else scale += d0 * bp[3 * p] + d1 * bp[3 * p + 1] + d2 * bp[3 * p + 2];
        }
        if (step) { poses_trial[3 * p] = x; poses_trial[3 * p + 1] = y; poses_trial[3 * p + 2] = th; }
        if (p < 1024) { sp[3 * p] = x; sp[3 * p + 1] = y; sp[3 * p + 2] = th; }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < O; k += blockDim.x) {
        const int i = o_i[k], j = o_j[k];
        double pi[3], pj[3];
        for (int c = 0; c < 3; ++c) {
            pi[c] = (i < 1024) ? sp[3 * i + c] : (step ? poses_trial[3 * i + c] : poses[3 * i + c]);
            pj[c] = (j < 1024) ? sp[3 * j + c] : (step ? poses_trial[3 * j + c] : poses[3 * j + c]);
        }
        double e[3], A[9], B[9];
        pre_se2(pi, pj, o_meas + 3 * k, e, A, B);
        const double* W = o_info + 9 * k;
        for (int r = 0; r < 3; ++r) chi += e[r] * (W[r * 3] * e[0] + W[r * 3 + 1] * e[1] + W[r * 3 + 2] * e[2]);
    }
    chi = wave_sum(chi);
    scale = wave_sum(scale);
    if ((threadIdx.x & 63) == 0) {
        sm[0][threadIdx.x >> 6] = chi;
        sm[1][threadIdx.x >> 6] = scale;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {   // fixed order: deterministic
            sm[0][0] += sm[0][w];
            sm[1][0] += sm[1][w];
        }
        out[0] = sm[0][0]; out[1] = sm[1][0]; out[3] = 0;
        if (!step) out[2] = 0;
        post_s = 0;
        if (ctl && decide) {
            if (!step) {   // evaluation of the starting state
                ctl->current_chi = ctl->chi2_init = ctl->chi2_final = sm[0][0];
                if (stop && *stop) { ctl->stopped = 1; ctl->done = 1; }
                if (ctl->iters <= 0) ctl->done = 1;
            } else {
                const double sc[3] = {sm[0][0], sm[1][0], out[2]};
                lm_advance(ctl, sc, stop && *stop);
            }
            post_s = (ctl->done || notify) && mail;
        } else if (mail && !ctl) {  // synchronous callers (se2gpu_ba_chi2): the three scalars
            mail[0] = sm[0][0];
            mail[1] = sm[1][0];
            mail[2] = step ? out[2] : 0.0;
            __threadfence_system();
            mail[3] = seq;
        }
    }
    if (ctl && decide) {
        __syncthreads();
        end_slot(ctl, mail, post_s != 0, threadIdx.x, blockDim.x, step != 0);
    }
}

// Sharded (multi-GPU) runs: k_finalize leaves this rank's partial scalars in the fused buffer, the all-reduce sums them,
// and this kernel takes the LM decision - identically on every rank, since the summed scalars are identical.
__global__ void k_lm_decide(BaCtl* __restrict__ ctl, const double* __restrict__ sc, int step, int notify,
                            volatile double* __restrict__ mail, double seq, const volatile int* __restrict__ stop) {
    __shared__ int post_s;
    if (threadIdx.x == 0) {
        if (!ctl->done) {
            if (!step) {
                ctl->current_chi = ctl->chi2_init = ctl->chi2_final = sc[0];
                if (stop && *stop) { ctl->stopped = 1; ctl->done = 1; }
                if (ctl->iters <= 0) ctl->done = 1;
            } else {
                lm_advance(ctl, sc, stop && *stop);
            }
        }
        post_s = (ctl->done || notify) && mail;
    }
    __syncthreads();
    end_slot(ctl, mail, post_s != 0, threadIdx.x, blockDim.x, step != 0);
}

// start of an optimize() call: fresh controller block
// start of an optimize(): everything is reset but the handle's own counters and the sel bit (which of the two estimate
// buffers is current: it follows from the runs before, the host only mirrors it) - nothing here changes from one run to
// the next, so the whole optimize() can be replayed as a hipGraph
__device__ __forceinline__ void d_ctl_init(const unsigned bx, BaCtl* __restrict__ ctl, int iters, int mode) {
    constexpr int kWords = (int)(offsetof(BaCtl, seq) / 8);
    __shared__ int sel_s;
    if (threadIdx.x == 0) sel_s = ctl->sel;
    __syncthreads();
    double* w = reinterpret_cast<double*>(ctl);
    for (int i = threadIdx.x; i < kWords; i += blockDim.x) w[i] = 0.0;
    __syncthreads();
    if (threadIdx.x == 0) {
        ctl->ni = 2;
        ctl->sel = sel_s;
        ctl->iters = iters;
        ctl->mode = mode;
    }
}
__global__ void k_ctl_init(BaCtl* __restrict__ ctl, int iters, int mode) {
    d_ctl_init(blockIdx.x, ctl, iters, mode);
}

// lambda_0 = 1e-5 * max |diag H| (computeLambdaInit); src holds the maximum (one entry), or one slot per rank
__global__ void k_set_lambda(BaCtl* __restrict__ ctl, const double* __restrict__ src, int count) {
    double m = 0;
    for (int i = 0; i < count; ++i) m = fmax(m, src[i]);
    ctl->lambda = 1e-5 * m;
    ctl->ni = 2;
}
__global__ void k_fill_slots(double* __restrict__ dst, const double* __restrict__ maxd, int rank, int world) {
    const int r = threadIdx.x;
    if (r < world) dst[r] = (r == rank) ? maxd[0] : 0.0;
}

// =============================================================================================
// SE3-expmap model: the marginalising local bundle adjustment of
//   Map::loadLocalGraph(optimizer, vpEdgesAll, vnAllIdx)      /root/reference/src/Map.cpp:414-566
//   EdgeSE3ExpmapPrior / addPlaneMotionSE3Expmap               /root/reference/src/optimizer.cpp:159-197, 236-314
//   LocalMapper::removeOutlierChi2                             /root/reference/src/LocalMapper.cpp:172-230
// on the same machinery as the SE(2) model (landmark-sorted edges, device-built contributor plan, dataflow pose solve,
// device-side LM controller): only the per-edge / per-pose arithmetic differs.  Poses are Tcw as 12 doubles (R row-major,
// t); a pose has D = 6 unknowns (update = (omega, upsilon), estimate <- exp(update) * estimate); reduced-system blocks
// are 6x6 (36-lane groups, 7 per workgroup); per-edge blocks: Hpl 6x3, Hpp_e sym 6x6 (21), bp_e 6, Y 6x3 and the
// diagonal record Dg = {sym(Hpp_e - Y Hpl^T) (21), bp_e (6), Hpl z (6)} (33).
// [3P g2o 20160424] VertexSE3Expmap, EdgeProjectXYZ2UV, EdgeSE3Expmap, SE3Quat - restated as in oracle/ba3_ref.cpp.
// =============================================================================================
constexpr int kGrpPerWG3 = 7;
struct Cam3 { double f, cx, cy, huber; };
__device__ __host__ inline int sym6(int r, int c) { return r * 6 - r * (r - 1) / 2 + (c - r); }   // r <= c

template <bool JAC>
__device__ inline void proj3(const Cam3& cam, const double* __restrict__ T, double X0, double X1, double X2, double u, double v,
                             double& e0, double& e1, double* Jp, double* Jl) {
    const double x = T[0] * X0 + T[1] * X1 + T[2] * X2 + T[9];
    const double y = T[3] * X0 + T[4] * X1 + T[5] * X2 + T[10];
    const double z = T[6] * X0 + T[7] * X1 + T[8] * X2 + T[11];
    const double zi = 1.0 / z;
    e0 = u - (x * zi * cam.f + cam.cx);
    e1 = v - (y * zi * cam.f + cam.cy);
    if (JAC) {
        const double f = cam.f, zi2 = zi * zi;
        const double t02 = -x * zi * f, t12 = -y * zi * f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            Jl[c] = -zi * (f * T[c] + t02 * T[6 + c]);
            Jl[3 + c] = -zi * (f * T[3 + c] + t12 * T[6 + c]);
        }
        Jp[0] = x * y * zi2 * f; Jp[1] = -(1 + (x * x * zi2)) * f; Jp[2] = y * zi * f; Jp[3] = -zi * f; Jp[4] = 0; Jp[5] = x * zi2 * f;
        Jp[6] = (1 + y * y * zi2) * f; Jp[7] = -x * y * zi2 * f; Jp[8] = -x * zi * f; Jp[9] = 0; Jp[10] = -zi * f; Jp[11] = y * zi2 * f;
    }
}

// Dinv-dependent part of one edge: Y = Hpl Dinv and the diagonal record
__device__ inline void schur_edge3(const double* __restrict__ hh, const double d[6], const double* __restrict__ hp,
                                   const double* __restrict__ bpe, double z0, double z1, double z2,
                                   double* __restrict__ y, double* __restrict__ dg) {
    double yy[18];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const double b0 = hh[3 * r], b1 = hh[3 * r + 1], b2 = hh[3 * r + 2];
        yy[3 * r] = b0 * d[0] + b1 * d[1] + b2 * d[2];
        yy[3 * r + 1] = b0 * d[1] + b1 * d[3] + b2 * d[4];
        yy[3 * r + 2] = b0 * d[2] + b1 * d[4] + b2 * d[5];
    }
#pragma unroll
    for (int i = 0; i < 18; ++i) y[i] = yy[i];
    int k = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = r; c < 6; ++c, ++k)
            dg[k] = hp[k] - (yy[3 * r] * hh[3 * c] + yy[3 * r + 1] * hh[3 * c + 1] + yy[3 * r + 2] * hh[3 * c + 2]);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        dg[21 + r] = bpe[r];
        dg[27 + r] = hh[3 * r] * z0 + hh[3 * r + 1] * z1 + hh[3 * r + 2] * z2;
    }
}

template <bool FUSED>
__global__ __launch_bounds__(kBlock) void k3_linearize(Cam3 cam, int L, const int* __restrict__ lm_ptr,
                                                        const int* __restrict__ e_kf, const double* __restrict__ e_uv,
                                                        const double* __restrict__ e_info, const double* __restrict__ poses,
                                                        const uint8_t* __restrict__ fixed, const double* __restrict__ lms,
                                                        double* Hpl, double* __restrict__ Hpp_e, double* __restrict__ bp_e,
                                                        double* __restrict__ Hll, double* __restrict__ bl, double lambda,
                                                        double* __restrict__ Dinv, double* __restrict__ z,
                                                        double* __restrict__ Y, double* __restrict__ Dg,
                                                        const BaCtl* __restrict__ ctl, const double* __restrict__ poses_b,
                                                        const double* __restrict__ lms_b) {
    if (ctl) {
        if (ctl->done | ctl->retry) return;
        if (ctl->sel) { poses = poses_b; lms = lms_b; }
        lambda = ctl->lambda;
    }
    const int gid = blockIdx.x * kBlock + threadIdx.x;
    const int l = gid / kGroup, sub = gid % kGroup;
    double hll[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
    if (l < L) {
        const double X0 = lms[3 * (size_t)l], X1 = lms[3 * (size_t)l + 1], X2 = lms[3 * (size_t)l + 2];
        for (int e = lm_ptr[l] + sub; e < lm_ptr[l + 1]; e += kGroup) {
            const int kf = e_kf[e];
            double e0, e1, Jp[12], Jl[6];
            proj3<true>(cam, poses + 12 * (size_t)kf, X0, X1, X2, e_uv[2 * (size_t)e], e_uv[2 * (size_t)e + 1], e0, e1, Jp, Jl);
            const double w = e_info[3 * (size_t)e];   // information = w I
            double r0, r1;
            huber(w * (e0 * e0 + e1 * e1), cam.huber, r0, r1);
            const double W = r1 * w, o0 = -W * e0, o1 = -W * e1;
            hll[0] += W * (Jl[0] * Jl[0] + Jl[3] * Jl[3]);
            hll[1] += W * (Jl[0] * Jl[1] + Jl[3] * Jl[4]);
            hll[2] += W * (Jl[0] * Jl[2] + Jl[3] * Jl[5]);
            hll[3] += W * (Jl[1] * Jl[1] + Jl[4] * Jl[4]);
            hll[4] += W * (Jl[1] * Jl[2] + Jl[4] * Jl[5]);
            hll[5] += W * (Jl[2] * Jl[2] + Jl[5] * Jl[5]);
#pragma unroll
            for (int r = 0; r < 3; ++r) b[r] += Jl[r] * o0 + Jl[3 + r] * o1;
            const bool fr = !fixed[kf];
            double* hpl = Hpl + (size_t)e * 18;
            double* hpp = Hpp_e + (size_t)e * 21;
            int k = 0;
#pragma unroll
            for (int r = 0; r < 6; ++r) {
#pragma unroll
                for (int c = 0; c < 3; ++c) hpl[3 * r + c] = fr ? W * (Jp[r] * Jl[c] + Jp[6 + r] * Jl[3 + c]) : 0.0;
#pragma unroll
                for (int c = r; c < 6; ++c, ++k) hpp[k] = fr ? W * (Jp[r] * Jp[c] + Jp[6 + r] * Jp[6 + c]) : 0.0;
                bp_e[(size_t)e * 6 + r] = fr ? Jp[r] * o0 + Jp[6 + r] * o1 : 0.0;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) hll[i] = group_sum(hll[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) b[i] = group_sum(b[i]);
    if (l < L && sub == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) Hll[(size_t)l * 6 + i] = hll[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) bl[(size_t)l * 3 + i] = b[i];
    }
    if (FUSED && l < L) {
        double d[6];
        inv_sym3(hll, lambda, d);
        const double z0 = d[0] * b[0] + d[1] * b[1] + d[2] * b[2];
        const double z1 = d[1] * b[0] + d[3] * b[1] + d[4] * b[2];
        const double z2 = d[2] * b[0] + d[4] * b[1] + d[5] * b[2];
        if (sub == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) Dinv[(size_t)l * 6 + i] = d[i];
            z[(size_t)l * 3] = z0; z[(size_t)l * 3 + 1] = z1; z[(size_t)l * 3 + 2] = z2;
        }
        for (int e = lm_ptr[l] + sub; e < lm_ptr[l + 1]; e += kGroup)
            schur_edge3(Hpl + (size_t)e * 18, d, Hpp_e + (size_t)e * 21, bp_e + (size_t)e * 6, z0, z1, z2, Y + (size_t)e * 18,
                        Dg + (size_t)e * 33);
    }
}

__global__ __launch_bounds__(kBlock) void k3_schur_lm(int L, double lambda, const int* __restrict__ lm_ptr,
                                                       const double* __restrict__ Hll, const double* __restrict__ bl,
                                                       const double* __restrict__ Hpl, const double* __restrict__ Hpp_e,
                                                       const double* __restrict__ bp_e, double* __restrict__ Dinv,
                                                       double* __restrict__ z, double* __restrict__ Y, double* __restrict__ Dg,
                                                       const BaCtl* __restrict__ ctl, int force) {
    if (ctl) {
        if (ctl->done || !(force | ctl->retry)) return;
        lambda = ctl->lambda;
    }
    const int gid = blockIdx.x * kBlock + threadIdx.x;
    const int l = gid / kGroup, sub = gid % kGroup;
    if (l >= L) return;
    double h[6], d[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) h[i] = Hll[(size_t)l * 6 + i];
    inv_sym3(h, lambda, d);
    const double b0 = bl[(size_t)l * 3], b1 = bl[(size_t)l * 3 + 1], b2 = bl[(size_t)l * 3 + 2];
    const double z0 = d[0] * b0 + d[1] * b1 + d[2] * b2, z1 = d[1] * b0 + d[3] * b1 + d[4] * b2, z2 = d[2] * b0 + d[4] * b1 + d[5] * b2;
    if (sub == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) Dinv[(size_t)l * 6 + i] = d[i];
        z[(size_t)l * 3] = z0; z[(size_t)l * 3 + 1] = z1; z[(size_t)l * 3 + 2] = z2;
    }
    for (int e = lm_ptr[l] + sub; e < lm_ptr[l + 1]; e += kGroup)
        schur_edge3(Hpl + (size_t)e * 18, d, Hpp_e + (size_t)e * 21, bp_e + (size_t)e * 6, z0, z1, z2, Y + (size_t)e * 18,
                    Dg + (size_t)e * 33);
}

// Per-pose and per-odometry-edge terms that depend on the estimate only (not on lambda): the prior gradient Omega e of
// EdgeSE3ExpmapPrior (e = log(M T^-1), Jacobian -I: H += Omega, b += Omega e) and the blocks of every EdgeSE3Expmap
// (e = log(T_j^-1 C T_i), J_i = adj(T_j^-1 C), J_j = -adj(T_i^-1 C^-1)): Oii, Ojj, Oij (6x6), obi, obj (6).
// One thread per pose, then one thread per odometry edge.
__global__ __launch_bounds__(64) void k3_terms(int P, int O, const double* __restrict__ poses, const uint8_t* __restrict__ fixed,
                         const uint8_t* __restrict__ prior_has, const double* __restrict__ prior_meas,
                         const double* __restrict__ prior_info, double* __restrict__ pb, const int* __restrict__ o_i,
                         const int* __restrict__ o_j, const double* __restrict__ o_meas, const double* __restrict__ o_info,
                         double* __restrict__ Oii, double* __restrict__ Ojj, double* __restrict__ Oij,
                         double* __restrict__ obi, double* __restrict__ obj, const BaCtl* __restrict__ ctl,
                         const double* __restrict__ poses_b) {
    if (ctl) {
        if (ctl->done | ctl->retry) return;
        if (ctl->sel) poses = poses_b;
    }
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < P) {
        double g[6] = {0, 0, 0, 0, 0, 0};
        if (prior_has[t] && !fixed[t]) {
            double e[6];
            se3_log(se3_mul(se3_load(prior_meas + 12 * (size_t)t), se3_inv(se3_load(poses + 12 * (size_t)t))), e);
            const double* W = prior_info + 36 * (size_t)t;
            for (int r = 0; r < 6; ++r)
                for (int c = 0; c < 6; ++c) g[r] += W[6 * r + c] * e[c];
        }
        for (int r = 0; r < 6; ++r) pb[6 * (size_t)t + r] = g[r];
        return;
    }
    const int k = t - P;
    if (k >= O) return;
    const int i = o_i[k], j = o_j[k];
    const Se3 Ti = se3_load(poses + 12 * (size_t)i), Tj = se3_load(poses + 12 * (size_t)j), C = se3_load(o_meas + 12 * (size_t)k);
    const Se3 TjC = se3_mul(se3_inv(Tj), C);
    double e[6], Ji[36], Jj[36], We[6];
    se3_log(se3_mul(TjC, Ti), e);
    se3_adj(TjC, Ji);
    se3_adj(se3_mul(se3_inv(Ti), se3_inv(C)), Jj);
    for (int q = 0; q < 36; ++q) Jj[q] = -Jj[q];
    const double* W = o_info + 36 * (size_t)k;
    for (int r = 0; r < 6; ++r) {
        We[r] = 0;
        for (int c = 0; c < 6; ++c) We[r] += W[6 * r + c] * e[c];
    }
    const bool fi = !fixed[i], fj = !fixed[j];
    for (int r = 0; r < 6; ++r) {
        for (int c = 0; c < 6; ++c) {
            double ii = 0, jj = 0, ij = 0;
            for (int a = 0; a < 6; ++a) {
                double wi = 0, wj = 0;
                for (int q = 0; q < 6; ++q) { wi += W[6 * a + q] * Ji[6 * q + c]; wj += W[6 * a + q] * Jj[6 * q + c]; }
                ii += Ji[6 * a + r] * wi;
                jj += Jj[6 * a + r] * wj;
                ij += Ji[6 * a + r] * wj;
            }
            Oii[36 * (size_t)k + 6 * r + c] = fi ? ii : 0.0;
            Ojj[36 * (size_t)k + 6 * r + c] = fj ? jj : 0.0;
            Oij[36 * (size_t)k + 6 * r + c] = (fi && fj) ? ij : 0.0;
        }
        double bi = 0, bj = 0;
        for (int q = 0; q < 6; ++q) { bi += Ji[6 * q + r] * We[q]; bj += Jj[6 * q + r] * We[q]; }
        obi[6 * (size_t)k + r] = fi ? -bi : 0.0;
        obj[6 * (size_t)k + r] = fj ? -bj : 0.0;
    }
}

// The reduced system of the SE3 model in one launch (structure of k_reduce2): off-diagonal 6x6 blocks from the
// contributor plan (36 lanes per 16-pair chunk, 7 chunks per workgroup), one workgroup per pose for the diagonal
// block, b_s and b_p (gather of the 33-double records + prior + odometry terms of k3_terms).
__global__ __launch_bounds__(kBlock) void k3_reduce2(int P, int ld, int nwg_off, double lambda, const int4* __restrict__ grp,
                                                      const int* __restrict__ blk_a, const int* __restrict__ blk_b,
                                                      const int* __restrict__ pair_i, const int* __restrict__ pair_j,
                                                      const int* __restrict__ blk_odo, const double* __restrict__ Y,
                                                      const double* __restrict__ Hpl, const double* __restrict__ Dg,
                                                      const uint8_t* __restrict__ fixed, const int* __restrict__ pose_ptr,
                                                      const int* __restrict__ pose_edges, const int* __restrict__ podo_ptr,
                                                      const int* __restrict__ podo_item, const uint8_t* __restrict__ prior_has,
                                                      const double* __restrict__ prior_info, const double* __restrict__ pb,
                                                      const double* __restrict__ Oii, const double* __restrict__ Ojj,
                                                      const double* __restrict__ Oij, const double* __restrict__ obi,
                                                      const double* __restrict__ obj, double* __restrict__ S,
                                                      double* __restrict__ bp, const BaCtl* __restrict__ ctl,
                                                      unsigned* __restrict__ epoch) {
    if (ctl) {
        if (ctl->done) return;
        lambda = ctl->lambda;
    }
    const int n = 6 * P;
    __shared__ double part[kGrpPerWG3][36];
    __shared__ double dpart[kBlock / 64][33];
    const int ndiag = (P + 1 + 7) & ~7;
    if ((int)blockIdx.x >= ndiag) {
        const int bid = (int)blockIdx.x - ndiag;
        const int nwg_pad = (nwg_off + 7) & ~7;
        const int wg = (bid & 7) * (nwg_pad >> 3) + (bid >> 3);   // contiguous eighths of the plan per XCD (see k_reduce2)
        if (wg >= nwg_off) return;
        const int g = threadIdx.x / 36, en = threadIdx.x - 36 * g;
        int4 d = make_int4(-1, 0, 0, 0);
        if (g < kGrpPerWG3) d = grp[(size_t)wg * kGrpPerWG3 + g];
        const int r = en / 6, c = en - 6 * r;
        double acc = 0;
        if (d.x >= 0) {
            for (int q = d.y; q < d.z; q += 4) {
                int ia[4], ib[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int qq = min(q + u, d.z - 1);
                    ia[u] = pair_i[qq];
                    ib[u] = pair_j[qq];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double* y = Y + (size_t)ia[u] * 18 + 3 * r;
                    const double* hh = Hpl + (size_t)ib[u] * 18 + 3 * c;
                    const double w = (q + u < d.z) ? 1.0 : 0.0;
                    acc += w * (y[0] * hh[0] + y[1] * hh[1] + y[2] * hh[2]);
                }
            }
            part[g][en] = acc;
        }
        __syncthreads();
        if (d.x >= 0 && (d.w & 0xff) == g) {
            const int ng = d.w >> 8;
            double tot = 0;
            for (int t = 0; t < ng; ++t) tot += part[g + t][en];
            const int a = blk_a[d.x], b = blk_b[d.x];
            double out = 0.0;
            if (!fixed[a] && !fixed[b]) {
                out = -tot;
                const int od = blk_odo[d.x];
                if (od >= 0) out += (od & 1) ? Oij[36 * (size_t)(od >> 1) + 6 * c + r] : Oij[36 * (size_t)(od >> 1) + 6 * r + c];
            }
            S[(size_t)(6 * a + r) * ld + 6 * b + c] = out;
            S[(size_t)(6 * b + c) * ld + 6 * a + r] = out;
        }
        return;
    }
    const int p = (int)blockIdx.x;
    if (p > P) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double* __restrict__ bs = S + (size_t)n * ld;
    if (p == P) {
        for (size_t t = (size_t)n * ld + n + threadIdx.x; t < (size_t)ld * ld; t += kBlock) S[t] = 0.0;
        if (threadIdx.x == 0) {
            S[(size_t)ld * ld + 2] = 0.0;
            *epoch += 1u;   // see k_reduce2
        }
        return;
    }
    const bool fa = fixed[p];
    double acc[33];
#pragma unroll
    for (int i = 0; i < 33; ++i) acc[i] = 0;
    if (!fa) {
        const int e0 = pose_ptr[p], ne = pose_ptr[p + 1] - e0;
        for (int t = threadIdx.x; t < ne; t += kBlock) {
            const double* rec = Dg + (size_t)pose_edges[e0 + t] * 33;
#pragma unroll
            for (int i = 0; i < 33; ++i) acc[i] += rec[i];
        }
    }
#pragma unroll
    for (int i = 0; i < 33; ++i) acc[i] = wave_sum(acc[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 33; ++i) dpart[wv][i] = acc[i];
    }
    __syncthreads();
    if (threadIdx.x < 42) {   // 36 entries of the block (row-major), then 6 right-hand-side components
        const int i = threadIdx.x;
        if (i < 36) {
            const int r = i / 6, c = i - 6 * r;
            double out;
            if (fa) {
                out = (r == c) ? 1.0 : 0.0;
            } else {
                const int k = r <= c ? sym6(r, c) : sym6(c, r);
                double v = dpart[0][k] + dpart[1][k] + dpart[2][k] + dpart[3][k];
                if (prior_has[p]) v += prior_info[36 * (size_t)p + i];
                for (int t = podo_ptr[p]; t < podo_ptr[p + 1]; ++t) {
                    const int ko = podo_item[t] >> 1;
                    v += ((podo_item[t] & 1) ? Ojj : Oii)[36 * (size_t)ko + i];
                }
                out = v + (r == c ? lambda : 0.0);
            }
            S[(size_t)(6 * p + r) * ld + 6 * p + c] = out;
        } else {
            const int r = i - 36;
            double v = dpart[0][21 + r] + dpart[1][21 + r] + dpart[2][21 + r] + dpart[3][21 + r];
            const double gz = dpart[0][27 + r] + dpart[1][27 + r] + dpart[2][27 + r] + dpart[3][27 + r];
            if (!fa) {
                v += pb[6 * (size_t)p + r];
                for (int t = podo_ptr[p]; t < podo_ptr[p + 1]; ++t) {
                    const int ko = podo_item[t] >> 1;
                    v += ((podo_item[t] & 1) ? obj : obi)[6 * (size_t)ko + r];
                }
            }
            bs[6 * p + r] = fa ? 0.0 : v - gz;
            bp[6 * (size_t)p + r] = fa ? 0.0 : v;
        }
    }
}

// diagonal of the un-reduced pose blocks (computeLambdaInit): one wave per pose
__global__ __launch_bounds__(kBlock) void k3_pose_diag(int P, const int* __restrict__ pose_ptr, const int* __restrict__ pose_edges,
                                                        const double* __restrict__ Hpp_e, const uint8_t* __restrict__ fixed,
                                                        const uint8_t* __restrict__ prior_has, const double* __restrict__ prior_info,
                                                        const int* __restrict__ podo_ptr, const int* __restrict__ podo_item,
                                                        const double* __restrict__ Oii, const double* __restrict__ Ojj,
                                                        double* __restrict__ diag) {
    const int p = blockIdx.x * (kBlock / 64) + threadIdx.x / 64;
    const int lane = threadIdx.x & 63;
    if (p >= P) return;
    double d[6] = {0, 0, 0, 0, 0, 0};
    for (int t = pose_ptr[p] + lane; t < pose_ptr[p + 1]; t += 64) {
        const double* hp = Hpp_e + (size_t)pose_edges[t] * 21;
#pragma unroll
        for (int r = 0; r < 6; ++r) d[r] += hp[sym6(r, r)];
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) d[r] = wave_sum(d[r]);
    if (lane == 0) {
        for (int r = 0; r < 6; ++r) {
            double v = d[r];
            if (!fixed[p]) {
                if (prior_has[p]) v += prior_info[36 * (size_t)p + 7 * r];
                for (int t = podo_ptr[p]; t < podo_ptr[p + 1]; ++t)
                    v += ((podo_item[t] & 1) ? Ojj : Oii)[36 * (size_t)(podo_item[t] >> 1) + 7 * r];
            }
            diag[6 * (size_t)p + r] = v;
        }
    }
}

// trial poses: exp(update) * estimate (VertexSE3Expmap::oplusImpl); estimate copied for fixed poses / without a step
__global__ void k3_oplus(int P, const double* __restrict__ poses, const uint8_t* __restrict__ fixed,
                         const double* __restrict__ xp, double* __restrict__ poses_trial, const BaCtl* __restrict__ ctl) {
    if (ctl) {
        if (ctl->done) return;
        if (ctl->sel) { const double* t = poses; poses = poses_trial; poses_trial = const_cast<double*>(t); }
    }
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    Se3 T = se3_load(poses + 12 * (size_t)p);
    if (!fixed[p]) {
        double u[6];
        for (int r = 0; r < 6; ++r) u[r] = xp[6 * (size_t)p + r];
        T = se3_mul(se3_exp(u), T);
    }
    se3_store(T, poses_trial + 12 * (size_t)p);
}

// back-substitution, trial landmarks, robust chi^2 of the landmark's edges at the trial state (poses from k3_oplus)
__global__ __launch_bounds__(kBlock) void k3_update(Cam3 cam, int L, double lambda, const int* __restrict__ lm_ptr,
                                                     const int* __restrict__ e_kf, const double* __restrict__ e_uv,
                                                     const double* __restrict__ e_info, const double* __restrict__ poses_a,
                                                     const double* __restrict__ poses_b, const double* __restrict__ lms,
                                                     const double* __restrict__ xp, const double* __restrict__ z,
                                                     const double* __restrict__ Y, const double* __restrict__ bl,
                                                     double* __restrict__ lms_trial, double* __restrict__ part,
                                                     const BaCtl* __restrict__ ctl, int step, double* __restrict__ edge_chi2) {
    // poses_a / poses_b: "a" / "b" pose buffers.  With a step the TRIAL poses are read (the buffer that does not hold the
    // estimate), without one the estimate.
    const double* poses = poses_a;
    if (ctl) {
        if (ctl->done && !edge_chi2) return;
        const bool est_b = ctl->sel != 0;
        poses = (est_b != (step != 0)) ? poses_b : poses_a;
        if (est_b) { const double* t = lms; lms = lms_trial; lms_trial = const_cast<double*>(t); }
        lambda = ctl->lambda;
    }
    __shared__ double sm[2][kBlock / 64];
    const int gid = blockIdx.x * kBlock + threadIdx.x;
    const int l = gid / kGroup, sub = gid % kGroup;
    double chi = 0, scale = 0, x[3] = {0, 0, 0};
    int beg = 0, end = 0;
    if (l < L) {
        beg = lm_ptr[l];
        end = lm_ptr[l + 1];
        if (step)
            for (int e = beg + sub; e < end; e += kGroup) {
                const double* y = Y + (size_t)e * 18;
                const double* q = xp + 6 * (size_t)e_kf[e];
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int r = 0; r < 6; ++r) x[c] -= y[3 * r + c] * q[r];
            }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) x[c] = group_sum(x[c]);
    if (l < L) {
        double lw[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (step) x[c] += z[(size_t)l * 3 + c];
            lw[c] = lms[(size_t)l * 3 + c] + x[c];
        }
        if (sub == 0 && step) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                lms_trial[(size_t)l * 3 + c] = lw[c];
                scale += x[c] * (lambda * x[c] + bl[(size_t)l * 3 + c]);
            }
        }
        for (int e = beg + sub; e < end; e += kGroup) {
            double e0, e1, r0, r1;
            proj3<false>(cam, poses + 12 * (size_t)e_kf[e], lw[0], lw[1], lw[2], e_uv[2 * (size_t)e], e_uv[2 * (size_t)e + 1], e0, e1,
                         nullptr, nullptr);
            const double c2 = e_info[3 * (size_t)e] * (e0 * e0 + e1 * e1);
            if (edge_chi2) edge_chi2[e] = c2;
            huber(c2, cam.huber, r0, r1);
            chi += r0;
        }
    }
    chi = wave_sum(chi);
    scale = wave_sum(scale);
    const int w = threadIdx.x / 64;
    if ((threadIdx.x & 63) == 0) { sm[0][w] = chi; sm[1][w] = scale; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double c = 0, s2 = 0;
        for (int i = 0; i < kBlock / 64; ++i) { c += sm[0][i]; s2 += sm[1][i]; }
        part[2 * blockIdx.x] = c;
        part[2 * blockIdx.x + 1] = s2;
    }
}

// single block: sums the k3_update partials, adds the prior and odometry chi^2 at the trial poses and the pose part of
// computeScale(), advances the LM controller (structure of k_finalize)
__global__ void k3_finalize(int nparts, const double* __restrict__ part, int P, const double* __restrict__ poses_a,
                            const double* __restrict__ poses_b, const uint8_t* __restrict__ fixed, const double* __restrict__ xp,
                            const double* __restrict__ bp, const uint8_t* __restrict__ prior_has,
                            const double* __restrict__ prior_meas, const double* __restrict__ prior_info, int O,
                            const int* __restrict__ o_i, const int* __restrict__ o_j, const double* __restrict__ o_meas,
                            const double* __restrict__ o_info, double* __restrict__ out, volatile double* __restrict__ mail,
                            double seq, BaCtl* __restrict__ ctl, int step, int notify, const volatile int* __restrict__ stop) {
    __shared__ double sm[2][16];
    __shared__ int post_s;
    if (ctl->done) {
        end_slot(ctl, mail, notify != 0, threadIdx.x, blockDim.x, step != 0);
        return;
    }
    const bool est_b = ctl->sel != 0;
    const double* poses = (est_b != (step != 0)) ? poses_b : poses_a;   // trial poses with a step, the estimate without
    const double lambda = ctl->lambda;
    double chi = 0, scale = 0;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) {
        chi += part[2 * i];
        scale += part[2 * i + 1];
    }
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
        if (step && !fixed[p])
            for (int r = 0; r < 6; ++r) scale += xp[6 * (size_t)p + r] * (lambda * xp[6 * (size_t)p + r] + bp[6 * (size_t)p + r]);
        if (prior_has[p]) {
            double e[6];
            se3_log(se3_mul(se3_load(prior_meas + 12 * (size_t)p), se3_inv(se3_load(poses + 12 * (size_t)p))), e);
            const double* W = prior_info + 36 * (size_t)p;
            for (int r = 0; r < 6; ++r) {
                double v = 0;
                for (int c = 0; c < 6; ++c) v += W[6 * r + c] * e[c];
                chi += e[r] * v;
            }
        }
    }
    for (int k = threadIdx.x; k < O; k += blockDim.x) {
        const Se3 Ti = se3_load(poses + 12 * (size_t)o_i[k]), Tj = se3_load(poses + 12 * (size_t)o_j[k]);
        double e[6];
        se3_log(se3_mul(se3_mul(se3_inv(Tj), se3_load(o_meas + 12 * (size_t)k)), Ti), e);
        const double* W = o_info + 36 * (size_t)k;
        for (int r = 0; r < 6; ++r) {
            double v = 0;
            for (int c = 0; c < 6; ++c) v += W[6 * r + c] * e[c];
            chi += e[r] * v;
        }
    }
    chi = wave_sum(chi);
    scale = wave_sum(scale);
    if ((threadIdx.x & 63) == 0) {
        sm[0][threadIdx.x >> 6] = chi;
        sm[1][threadIdx.x >> 6] = scale;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {
            sm[0][0] += sm[0][w];
            sm[1][0] += sm[1][w];
        }
        out[0] = sm[0][0]; out[1] = sm[1][0]; out[3] = 0;
        if (!step) {
            out[2] = 0;
            ctl->current_chi = ctl->chi2_init = ctl->chi2_final = sm[0][0];
            if (stop && *stop) { ctl->stopped = 1; ctl->done = 1; }
            if (ctl->iters <= 0) ctl->done = 1;
        } else {
            const double sc[3] = {sm[0][0], sm[1][0], out[2]};
            lm_advance(ctl, sc, stop && *stop);
        }
        post_s = (ctl->done || notify) && mail;
    }
    __syncthreads();
    end_slot(ctl, mail, post_s != 0, threadIdx.x, blockDim.x, step != 0);
}

// =============================================================================================
// Pose-graph model: GlobalMapper::GlobalBA (/root/reference/src/GlobalMapper.cpp:328-535) -
// g2o::VertexSE3 (T_w_c), one EdgeSE3Prior per key frame (addVertexSE3PlaneMotion, src/optimizer.cpp:336-470), EdgeSE3
// odometry and feature edges; no landmarks, so the "reduced" system is the whole system and the SE3 model's
// k3_reduce2 / dataflow solve / LM controller run unchanged on the blocks this kernel family produces.
// Several edges may join the same two key frames (odometry + feature edge): the edges are grouped into PAIR SLOTS
// (canonical orientation a < b), one thread sums a slot's edges into H_aa, H_bb, H_ab, b_a, b_b.
// [3P g2o 20160424 types/slam3d] restated as in oracle/pg_ref.cpp (se3_math.h: to_mqt / from_mqt / mqt_jac_*).
// =============================================================================================
__device__ inline void jtwj6(const double* Ja, const double* W, const double* Jc, double* out, double sgn) {   // out += Ja' W Jc
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) {
            double v = 0;
            for (int q = 0; q < 6; ++q) {
                double wj = 0;
                for (int u = 0; u < 6; ++u) wj += W[6 * q + u] * Jc[6 * u + c];
                v += Ja[6 * q + r] * wj;
            }
            out[6 * r + c] += sgn * v;
        }
}
__device__ inline void jtwe6(const double* Ja, const double* W, const double* e, double* out) {   // out -= Ja' W e
    double We[6];
    for (int r = 0; r < 6; ++r) { We[r] = 0; for (int c = 0; c < 6; ++c) We[r] += W[6 * r + c] * e[c]; }
    for (int r = 0; r < 6; ++r) { double v = 0; for (int q = 0; q < 6; ++q) v += Ja[6 * q + r] * We[q]; out[r] -= v; }
}

__global__ __launch_bounds__(64) void k4_terms(int P, int nslot, const double* __restrict__ poses, const uint8_t* __restrict__ fixed,
                         const uint8_t* __restrict__ prior_has, const double* __restrict__ prior_meas,
                         const double* __restrict__ prior_info, double* __restrict__ ph, double* __restrict__ pb,
                         const int* __restrict__ slot_a, const int* __restrict__ slot_ptr, const int* __restrict__ e_i,
                         const int* __restrict__ e_j, const double* __restrict__ e_meas, const double* __restrict__ e_info,
                         double* __restrict__ Oii, double* __restrict__ Ojj, double* __restrict__ Oij,
                         double* __restrict__ obi, double* __restrict__ obj, const BaCtl* __restrict__ ctl,
                         const double* __restrict__ poses_b) {
    if (ctl) {
        if (ctl->done | ctl->retry) return;
        if (ctl->sel) poses = poses_b;
    }
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < P) {
        double H[36], g[6];
        for (int i = 0; i < 36; ++i) H[i] = 0;
        for (int i = 0; i < 6; ++i) g[i] = 0;
        if (prior_has[t] && !fixed[t]) {
            const Se3 E = iso_mul(se3_inv(se3_load(prior_meas + 12 * (size_t)t)), se3_load(poses + 12 * (size_t)t));
            double e[6], J[36];
            to_mqt(E, e);
            mqt_jac_right(E, J);
            jtwj6(J, prior_info + 36 * (size_t)t, J, H, 1.0);
            jtwe6(J, prior_info + 36 * (size_t)t, e, g);
        }
        for (int i = 0; i < 36; ++i) ph[36 * (size_t)t + i] = H[i];
        for (int i = 0; i < 6; ++i) pb[6 * (size_t)t + i] = g[i];
        return;
    }
    const int sl = t - P;
    if (sl >= nslot) return;
    const int a = slot_a[sl];
    double Haa[36], Hbb[36], Hab[36], ba[6], bb[6];
    for (int i = 0; i < 36; ++i) Haa[i] = Hbb[i] = Hab[i] = 0;
    for (int i = 0; i < 6; ++i) ba[i] = bb[i] = 0;
    for (int k = slot_ptr[sl]; k < slot_ptr[sl + 1]; ++k) {
        const int i = e_i[k], j = e_j[k];
        const Se3 A = se3_inv(se3_load(e_meas + 12 * (size_t)k));
        const Se3 B = iso_mul(se3_inv(se3_load(poses + 12 * (size_t)i)), se3_load(poses + 12 * (size_t)j));
        const Se3 E = iso_mul(A, B);
        double e[6], Ji[36], Jj[36];
        to_mqt(E, e);
        mqt_jac_left_inv(A, B, Ji);
        mqt_jac_right(E, Jj);
        const double* W = e_info + 36 * (size_t)k;
        const bool fi = !fixed[i], fj = !fixed[j];
        // vertex 0 (i) is the slot's `a` or its `b`
        double* Hi = i == a ? Haa : Hbb;
        double* Hj = i == a ? Hbb : Haa;
        double* bi = i == a ? ba : bb;
        double* bj = i == a ? bb : ba;
        if (fi) { jtwj6(Ji, W, Ji, Hi, 1.0); jtwe6(Ji, W, e, bi); }
        if (fj) { jtwj6(Jj, W, Jj, Hj, 1.0); jtwe6(Jj, W, e, bj); }
        if (fi && fj) {
            if (i == a) jtwj6(Ji, W, Jj, Hab, 1.0);   // H_ab = J_a' W J_b
            else jtwj6(Jj, W, Ji, Hab, 1.0);
        }
    }
    for (int q = 0; q < 36; ++q) { Oii[36 * (size_t)sl + q] = Haa[q]; Ojj[36 * (size_t)sl + q] = Hbb[q]; Oij[36 * (size_t)sl + q] = Hab[q]; }
    for (int q = 0; q < 6; ++q) { obi[6 * (size_t)sl + q] = ba[q]; obj[6 * (size_t)sl + q] = bb[q]; }
}

// trial poses: estimate * fromVectorMQT(update) (VertexSE3::oplusImpl)
__global__ void k4_oplus(int P, const double* __restrict__ poses, const uint8_t* __restrict__ fixed,
                         const double* __restrict__ xp, double* __restrict__ poses_trial, const BaCtl* __restrict__ ctl) {
    if (ctl) {
        if (ctl->done) return;
        if (ctl->sel) { const double* t = poses; poses = poses_trial; poses_trial = const_cast<double*>(t); }
    }
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    Se3 T = se3_load(poses + 12 * (size_t)p);
    if (!fixed[p]) {
        double u[6];
        for (int r = 0; r < 6; ++r) u[r] = xp[6 * (size_t)p + r];
        T = iso_mul(T, from_mqt(u));
    }
    se3_store(T, poses_trial + 12 * (size_t)p);
}

__device__ inline double quad6(const double* W, const double* e) {
    double s = 0;
    for (int r = 0; r < 6; ++r) {
        double v = 0;
        for (int c = 0; c < 6; ++c) v += W[6 * r + c] * e[c];
        s += e[r] * v;
    }
    return s;
}

// chi^2 of the priors and of every EdgeSE3 at the trial poses (the estimate without a step), pose part of computeScale(),
// LM decision.  edge_chi2 (nullable): chi2() per edge, in slot order.
__global__ void k4_finalize(int P, int nedge, const double* __restrict__ poses_a, const double* __restrict__ poses_b,
                            const uint8_t* __restrict__ fixed, const double* __restrict__ xp, const double* __restrict__ bp,
                            const uint8_t* __restrict__ prior_has, const double* __restrict__ prior_meas,
                            const double* __restrict__ prior_info, const int* __restrict__ e_i, const int* __restrict__ e_j,
                            const double* __restrict__ e_meas, const double* __restrict__ e_info, double* __restrict__ out,
                            volatile double* __restrict__ mail, double seq, BaCtl* __restrict__ ctl, int step, int notify,
                            const volatile int* __restrict__ stop, double* __restrict__ edge_chi2) {
    __shared__ double sm[2][16];
    __shared__ int post_s;
    const double* poses = poses_a;
    double lambda = 0;
    if (ctl) {
        if (ctl->done) {
            end_slot(ctl, mail, notify != 0, threadIdx.x, blockDim.x, step != 0);
            return;
        }
        poses = ((ctl->sel != 0) != (step != 0)) ? poses_b : poses_a;
        lambda = ctl->lambda;
    }
    double chi = 0, scale = 0;
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
        if (step && !fixed[p])
            for (int r = 0; r < 6; ++r) scale += xp[6 * (size_t)p + r] * (lambda * xp[6 * (size_t)p + r] + bp[6 * (size_t)p + r]);
        if (prior_has[p]) {
            double e[6];
            to_mqt(iso_mul(se3_inv(se3_load(prior_meas + 12 * (size_t)p)), se3_load(poses + 12 * (size_t)p)), e);
            chi += quad6(prior_info + 36 * (size_t)p, e);
        }
    }
    for (int k = threadIdx.x; k < nedge; k += blockDim.x) {
        double e[6];
        to_mqt(iso_mul(se3_inv(se3_load(e_meas + 12 * (size_t)k)),
                       iso_mul(se3_inv(se3_load(poses + 12 * (size_t)e_i[k])), se3_load(poses + 12 * (size_t)e_j[k]))), e);
        const double c2 = quad6(e_info + 36 * (size_t)k, e);
        if (edge_chi2) edge_chi2[k] = c2;
        chi += c2;
    }
    if (!ctl) return;
    chi = wave_sum(chi);
    scale = wave_sum(scale);
    if ((threadIdx.x & 63) == 0) {
        sm[0][threadIdx.x >> 6] = chi;
        sm[1][threadIdx.x >> 6] = scale;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {
            sm[0][0] += sm[0][w];
            sm[1][0] += sm[1][w];
        }
        out[0] = sm[0][0]; out[1] = sm[1][0]; out[3] = 0;
        if (!step) {
            out[2] = 0;
            ctl->current_chi = ctl->chi2_init = ctl->chi2_final = sm[0][0];
            if (stop && *stop) { ctl->stopped = 1; ctl->done = 1; }
            if (ctl->iters <= 0) ctl->done = 1;
        } else {
            const double sc[3] = {sm[0][0], sm[1][0], out[2]};
            lm_advance(ctl, sc, stop && *stop);
        }
        post_s = (ctl->done || notify) && mail;
    }
    __syncthreads();
    end_slot(ctl, mail, post_s != 0, threadIdx.x, blockDim.x, step != 0);
}

// ---------------------------------------------------------------------------------------------
// Host: dense SPD solve of the reduced pose system (replaces CHOLMOD on the (3P)^2 matrix).
// Blocked right-looking LL^T on the lower triangle, row-major, FP64.
// ---------------------------------------------------------------------------------------------
bool host_cholesky_solve(double* A, int n, double* x) {
    constexpr int NB = 48;
    for (int k0 = 0; k0 < n; k0 += NB) {
        const int kb = std::min(NB, n - k0);
        // factor the diagonal block
        for (int j = k0; j < k0 + kb; ++j) {
            double* Aj = A + (size_t)j * n;
            double d = Aj[j];
            for (int k = k0; k < j; ++k) d -= Aj[k] * Aj[k];
            if (!(d > 0.0) || !std::isfinite(d)) return false;
            d = std::sqrt(d);
            Aj[j] = d;
            const double id = 1.0 / d;
            for (int i = j + 1; i < k0 + kb; ++i) {
                double* Ai = A + (size_t)i * n;
                double s = Ai[j];
                for (int k = k0; k < j; ++k) s -= Ai[k] * Aj[k];
                Ai[j] = s * id;
            }
        }
        // panel: rows below the block, solve X * L_kk^T = A_ik
        for (int i = k0 + kb; i < n; ++i) {
            double* Ai = A + (size_t)i * n;
            for (int j = k0; j < k0 + kb; ++j) {
                const double* Aj = A + (size_t)j * n;
                double s = Ai[j];
                for (int k = k0; k < j; ++k) s -= Ai[k] * Aj[k];
                Ai[j] = s / Aj[j];
            }
        }
        // trailing update (lower triangle): A_ij -= sum_k L_ik L_jk
        for (int i = k0 + kb; i < n; ++i) {
            double* Ai = A + (size_t)i * n;
            const double* Li = Ai + k0;
            for (int j = k0 + kb; j <= i; ++j) {
                const double* Lj = A + (size_t)j * n + k0;
                double s = 0;

                for (int k = 0; k < kb; ++k) s += Li[k] * Lj[k];
                Ai[j] -= s;
            }
        }
    }
    for (int i = 0; i < n; ++i) {
        const double* Li = A + (size_t)i * n;
        double s = x[i];
        for (int k = 0; k < i; ++k) s -= Li[k] * x[k];
        x[i] = s / Li[i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = x[i];
        for (int k = i + 1; k < n; ++k) s -= A[(size_t)k * n + i] * x[k];
        x[i] = s / A[(size_t)i * n + i];
    }
    return true;
}

struct EdgeOdo {
    int i, j;
    double meas[3];
    double info[9];
};

}  // namespace

// Map::loadLocalGraph's per-observation information (src/Map.cpp:1024-1049), one thread per edge
__global__ void k_edge_information(int E, const float* __restrict__ lc_, const float* __restrict__ lw_,
                                   const int* __restrict__ e_kf, const float* __restrict__ sigma2,
                                   const float* __restrict__ Rcw_, const float* __restrict__ twb, float fx,
                                   float s_rot, float s_z, double* __restrict__ out, int sym3) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= E) return;
    const double lc0 = lc_[3 * k], lc1 = lc_[3 * k + 1], lc2 = lc_[3 * k + 2];
    const int kf = e_kf[k];
    double R[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = Rcw_[9 * kf + i];
    const double zi = 1. / lc2, zi2 = zi * zi;
    const double j00 = fx * zi, j02 = -fx * lc0 * zi2, j12 = -fx * lc1 * zi2;
    double A[6];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        A[c] = j00 * R[c] + j02 * R[6 + c];
        A[3 + c] = j00 * R[3 + c] + j12 * R[6 + c];
    }
    const double d0 = (double)lw_[3 * k] - (double)twb[2 * kf], d1 = (double)lw_[3 * k + 1] - (double)twb[2 * kf + 1];
    const double d2 = (double)lw_[3 * k + 2];
    // (A * skew(d))[:, 0] = A[:,1]*d2 - A[:,2]*d1 ; [:, 1] = -A[:,0]*d2 + A[:,2]*d0
    const double r00 = A[1] * d2 - A[2] * d1, r01 = -A[0] * d2 + A[2] * d0;
    const double r10 = A[4] * d2 - A[5] * d1, r11 = -A[3] * d2 + A[5] * d0;
    const double z0 = -A[2], z1 = -A[5];
    const double s2 = sigma2[k];
    const double S00 = s_rot * (r00 * r00 + r01 * r01) + s_z * z0 * z0 + s2;
    const double S01 = s_rot * (r00 * r10 + r01 * r11) + s_z * z0 * z1;
    const double S11 = s_rot * (r10 * r10 + r11 * r11) + s_z * z1 * z1 + s2;
    const double id = 1.0 / (S00 * S11 - S01 * S01);
    if (sym3) {   // (xx, xy, yy): the layout of the graph's e_info
        out[3 * (size_t)k] = S11 * id; out[3 * (size_t)k + 1] = -S01 * id; out[3 * (size_t)k + 2] = S00 * id;
    } else {
        out[4 * k] = S11 * id; out[4 * k + 1] = -S01 * id; out[4 * k + 2] = -S01 * id; out[4 * k + 3] = S00 * id;
    }
}

// ---------------------------------------------------------------------------------------------
// handle
// ---------------------------------------------------------------------------------------------
// vertex id -> slot.  g2o ids are small dense integers (Map.cpp:925-985 numbers them 0 .. nKF + nMP); a direct table keeps
// the bulk load of a 20 k-landmark window out of a hash map.  Ids outside [0, kDenseIds) fall back to the map.
struct IdTable {
    static constexpr unsigned kDenseIds = 1u << 24;
    std::vector<int> dense;
    std::unordered_map<int, int> sparse;
    int find(int id) const {
        if ((unsigned)id < dense.size()) return dense[(unsigned)id];
        if ((unsigned)id < kDenseIds) return -1;
        const auto it = sparse.find(id);
        return it == sparse.end() ? -1 : it->second;
    }
    void set(int id, int slot) {
        if ((unsigned)id < kDenseIds) {
            if ((unsigned)id >= dense.size()) dense.resize(std::max<size_t>((size_t)id + 1, 2 * dense.size()), -1);
            dense[(unsigned)id] = slot;
        } else {
            sparse[id] = slot;
        }
    }
    void clear() { dense.clear(); sparse.clear(); }
    void reserve(size_t n) { dense.reserve(n); }
};

constexpr int kMailDoubles = 256, kMailStop = 250;
static_assert(8 + sizeof(BaCtl) / 8 <= kMailStop, "controller block does not fit the mailbox");

// how many handle streams have been destroyed so far: an event that was last recorded on a stream which no longer exists must not be
// handed to hipEventSynchronize any more (the runtime looks at the stream: "operation not permitted on an event last recorded in a
// capturing stream", or worse) - se2gpu_ba_reset_estimates_batch keeps such events across calls and compares this count
inline std::atomic<unsigned long>& streams_destroyed() { static std::atomic<unsigned long> n{0}; return n; }

struct se2gpu_ba {
    hipStream_t own_stream = nullptr, stream = nullptr;
    LaunchProfile prof;
    // graph under construction (host)
    CamDev cam{};
    bool have_cam = false, have_tbc = false;
    double Rbc[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, tbc[3] = {0, 0, 0};
    IdTable pose_of_id, lm_of_id;
    std::vector<int> pose_ids, lm_ids;
    std::vector<double> h_poses, h_lms;
    std::vector<uint8_t> h_fixed;
    std::vector<int> he_kf, he_lm;            // observation edges, SoA in insertion order (indices, not ids)
    std::vector<double> he_uv, he_info;       // 2 / 3 (xx, xy, yy) per edge
    double huber_delta = 0;
    bool huber_mixed = false;
    // pose model: 0 = SE(2) (VertexSE2 / EdgeSE2XYZ / PreEdgeSE2), 1 = SE3-expmap (VertexSE3Expmap / EdgeProjectXYZ2UV /
    // EdgeSE3ExpmapPrior / EdgeSE3Expmap).  D = unknowns per pose, ps = doubles a pose is stored in.
    int model = 0, D = 3, ps = 3;
    Cam3 cam3{};
    std::vector<uint8_t> h_prior_has;
    std::vector<double> h_prior_meas, h_prior_info;      // 12 / 36 per pose
    struct Odo3 { int i, j; double meas[12], info[36]; int level = 0; };   // level: g2o's OptimizableGraph::Edge::setLevel (pose graph only)
    std::vector<Odo3> odo3;
    DevBuf<uint8_t> prior_has;
    DevBuf<double> prior_meas, prior_info, pb, edge_chi2;
    std::vector<int> edge_perm;                          // sorted position -> insertion index (empty = identity)
    // model 2 (pose graph: VertexSE3 / EdgeSE3 / EdgeSE3Prior): the EdgeSE3 edges grouped into pair slots
    int pg_edges = 0;
    std::vector<int> pg_perm;                            // slot-ordered position -> insertion index
    DevBuf<int> slot_a, slot_ptr, pe_i, pe_j;
    DevBuf<double> pe_meas, pe_info, ph;
    // se2gpu_ba_load: the caller's edge arrays, borrowed until initialize (copied once, straight into the pinned arena)
    int bulk_E = 0;
    const int32_t *bulk_kf = nullptr, *bulk_lm = nullptr;
    const double *bulk_uv = nullptr, *bulk_info = nullptr;
    // se2gpu_ba_load_local_graph: raw inputs of the per-observation information (Map.cpp:1024-1049), evaluated on the
    // device straight into e_info after the upload
    bool lg_active = false;
    std::vector<float> lg_lc, lg_lw, lg_sigma2, lg_Rcw, lg_twb;
    float lg_fx = 0, lg_srot = 0, lg_sz = 0;
    DevBuf<float> d_lg_lc, d_lg_lw, d_lg_sigma2, d_lg_Rcw, d_lg_twb;
    std::vector<EdgeOdo> odo;
    bool initialized = false;
    unsigned long init_serial = 0;   // unique per initialize(): what a cached batch plan (ba_lockstep) was built for
    // se2gpu_ba_reset_estimates_batch wrote this handle's estimate on ANOTHER stream: the event to wait for before the next
    // operation on the handle's own stream (ba_join), and the stream it was recorded on
    hipEvent_t join_event = nullptr;
    hipStream_t join_stream = nullptr;
    bool own_pending = false;      // se2gpu_ba_reset_estimates enqueued copies on the handle's own stream that nobody has waited for
    int P = 0, L = 0, E = 0, O = 0, nblk = 0, nparts = 0;
    int ld = 0;              // leading dimension = padded order of the augmented reduced system
    bool host_solve = false; // SE2GPU_BA_HOST_SOLVE=1: factorise on the host instead (north-star wording)
    // device SoA
    DevBuf<double> poses0, lms0, poses_a, poses_b, lms_a, lms_b;
    double *poses = nullptr, *poses_t = nullptr, *lms = nullptr, *lms_t = nullptr;
    DevBuf<uint8_t> fixed;
    DevBuf<int> lm_ptr, e_kf, e_lm, pose_ptr, pose_edges, podo_ptr, podo_item, o_i, o_j;
    DevBuf<int> blk_a, blk_b, blk_ptr, pair_i, pair_j, blk_odo;
    DevBuf<int4> grp;
    int nwg_off = 0, grp_cap_wg = 0;
    DevBuf<uint8_t> garena;        // every uploaded graph array lives in this one allocation (one H2D copy)
    DevBuf<int> plan_np, plan_base, plan_key0, plan_key1, plan_idx, plan_hist, plan_offs, plan_out, plan_tsum, plan_toff;  // device plan scratch
    DevBuf<int2> plan_st0, plan_st1;
    bool odo_fallback = false;
    DevBuf<double> e_uv, e_info, o_meas, o_info;
    DevBuf<double> Hpl, Hpp_e, bp_e, Hll, bl, Dinv, z, Y, Dg, Hpp, bp, Oii, Ojj, Oij, obi, obj;
    DevBuf<double> red_own, xp, part, scal, diag3, Rinv;
    DevBuf<double> red_packed;    // sharded runs: the lower-triangular tiles of [S; b^T], what the all-reduce ships
    DevBuf<int4> chol_tasks;      // k_chol_tiles: {tile row | kind << 16, block column, dependency list [first, last)}, by column
    DevBuf<int> chol_deps;        // the dependency lists: block column | (own tile row non-zero there) << 15
    DevBuf<int> pose_off;         // fill-reducing order of the pose solve: first system column of pose p (nullptr = 3 p)
    DevBuf<int> col_src;          // system column -> 3 * pose + component, -1 = padding (nullptr = identity)
    std::vector<int> h_pose_off;  // host copy (debug_reduced_system gathers S back into pose order)
    DevBuf<uint8_t> plan_nz;      // per block of the upper triangle: structurally non-zero (k_plan_odo, k_plan_pairs2)
    DevBuf<double> plan_nzd;      // the same as doubles: what a sharded run's all-reduce can merge
    PinBuf<uint8_t> h_plan_nz;
    DevBuf<uint8_t> solver_arena; // chol_tasks | chol_deps | pose_off | col_src
    int nsys = 0;                 // order of the (padded) system the solver factorises; D * P in natural order
    int solve_depth = 0;          // block columns on the longest dependency chain of the plan (debug)
    DevBuf<unsigned> chol_flags;  // [2][nt][nbc][kSlabs] epochs
    DevBuf<unsigned long long> chol_head;   // k_chol_tiles: task numbers drawn so far (a workgroup's task = this count mod the task count)
    DevBuf<unsigned long long> chol_vfy;   // SE2GPU_BA_CHOL_VERIFY=1: mismatch records + half-slab checksums (d_chol_tiles<true>)
    DevBuf<int> plan_place;       // k_plan_tab -> k_plan_tabscan -> k_plan_place: the 256 runs' transfer tables, then their start states
    DevBuf<unsigned> fin_counter; // k_update: landmark workgroups that have published their partials (FinArgs)
    DevBuf<unsigned long long> l0_acc;   // k_lambda0: {max |diag H| as bits, workgroups arrived}; left at zero by every run
    int chol_ntask = 0;
    // optimize(n) as ONE hipGraph launch: captured the second time the same (iterations, mode) is asked of an initialised
    // handle (a one-shot localBA never pays for the capture), replayed from then on.  Nothing in the slot sequence changes
    // between runs: the slot counter, the solver's epoch and the estimate's buffer bit live on the device (BaCtl).
    // A small cache keyed by (iterations, mode): optimize(n) -> chi2 (= a run of zero iterations for the SE3 models) ->
    // optimize(n) keeps replaying, and a caller that alternates two iteration counts does not re-capture every time.
    struct GraphSlot { int iters = -1, mode = -1; hipGraphExec_t exec = nullptr; unsigned long stamp = 0; };
    static constexpr int kGraphSlots = 3;
    GraphSlot graphs[kGraphSlots];
    unsigned long graph_clock = 0;
    void drop_graphs() {
        for (auto& g : graphs) {
            if (g.exec) (void)hipGraphExecDestroy(g.exec);
            g = GraphSlot{};
        }
    }
    double dev_seq = 0;           // host mirror of BaCtl::seq
    DevBuf<long long> chol_trace; // SE2GPU_BA_CHOL_TRACE=1: per-task stamps of the last solve -> stderr (debug_solve)
    bool chol_steps = false;      // SE2GPU_BA_CHOL=steps: one launch per block column (k_chol_step) instead
    bool chol_faulted = false;    // SE2GPU_BA_CHOL_FAULT=1: the injected fault has been spent
    bool chol_fallback = false;   // a dataflow solve timed out once: this handle stays with k_chol_step
    double* red = nullptr;  // [augmented (ld x ld): rows 0..n-1 = S, row n = bs | 4 scalars]
    PinBuf<double> h_red, h_x, h_scal;
    // mapped + coherent host mailbox (kMailDoubles): [0..2] = {chi2, scale, fail} of a synchronous evaluation, [3] = sequence
    // number (written last), [8 ..] = the controller block posted by k_finalize / k_lm_decide, [kMailStop] = the force-stop
    // word the device reads (written by the host while it waits)
    double* h_mail = nullptr;
    double* d_mail = nullptr;
    volatile int* h_stop = nullptr;
    int* d_stop = nullptr;
    uint64_t mail_seq = 0;
    DevBuf<BaCtl> ctl;             // device-side LM controller
    int run_mode = SE2GPU_BA_LM, run_iters = 0, run_enqueued = 0;
    bool run_sync = false, run_active = false;
    double run_seq = 0;
    // multi-GPU
    se2gpu_allreduce_fn allreduce = nullptr;
    void* ar_user = nullptr;
    se2gpu_comm* comm = nullptr;   // native RCCL path (se2gpu_ba_set_comm)
    void* ar_buffer = nullptr;
    int root = 1, rank = 0, world = 1;
    int device = 0;                // device the buffers live on (handles are pooled per device, see se2gpu_ba_destroy)
    PinBuf<uint8_t> h_stage;       // pinned arena the graph arrays pass through on their way to the device
    PinBuf<uint8_t> h_stage_solver;   // the solve plan's lists (built while the arena above may still be read by the copy engine)
    // host copy of the estimates: Map::optimizeLocalGraph asks for every vertex separately (Map.cpp:754-783), one
    // download serves all of those calls until the estimates change again
    PinBuf<double> est;            // [poses 3P | landmarks 3L]
    bool est_valid = false;

    // the two big edge arrays (measurements, information: 40 of the 48 bytes of an edge) travel on a copy stream beside
    // the plan kernels, which only need the indices.  The stream is shared by all handles of a device (ba_copy_stream): a
    // stream per handle would take hardware queues away from the handles' own streams when many windows run at once
    hipEvent_t ev_copy0 = nullptr, ev_copy1 = nullptr, ev_pat = nullptr;

    ~se2gpu_ba() {
        if (ev_copy0) (void)hipEventDestroy(ev_copy0);
        if (ev_copy1) (void)hipEventDestroy(ev_copy1);
        if (ev_pat) (void)hipEventDestroy(ev_pat);
        drop_graphs();
        if (own_stream) {
            (void)hipStreamDestroy(own_stream);
            streams_destroyed().fetch_add(1, std::memory_order_relaxed);
        }
        if (h_mail) (void)hipHostFree(h_mail);
    }
};

namespace {

// =============================================================================================
// Graph plan on the device: everything initializeOptimization used to build on the host - the
// landmark / pose CSR lists and the contributor plan of k_reduce2 - from the raw edge arrays, in a handful of small
// launches.  The result is element-for-element what the host builder (ba_plan_host, kept as the reference
// implementation and fallback: SE2GPU_BA_PLAN=host, or more than 1024 poses) produces, so the LM results are bit-identical
// (tests/test_ba_gpu.py::test_device_plan_equals_host_plan).
//   k_plan_init      lm_ptr by binary search in the landmark-sorted edge list + pairs per landmark | identity permutation |
//                    blk_a / blk_b / blk_odo | cleared group descriptors         (one launch, segment after segment)
//   k_scan_i32       exclusive scan (one workgroup)
//   k_plan_pairs2    (block key, edge s, edge t) of every contributor pair, in (landmark, s, t) order | tail keys
//   k_radix_*        stable LSD radix sort (hist / scan / scatter), <= 10 bits per pass: pairs by block key, edges by pose
//   k_lower_bounds   CSR pointers of a sorted key list (pose_ptr); k_plan_bounds: blk_ptr | split of the (s, t) pairs
//   k_plan_pack(2)   first-fit-in-order packing of the 16-pair chunks into workgroups of 28 groups: the sequential rule
//                    of the host builder evaluated as a scan over transfer functions on the 28 fill states
// =============================================================================================
__device__ __host__ inline int blk_index_of(int P, int a, int b) { return a * P - a * (a - 1) / 2 + (b - a); }

// exclusive scan of n ints by ONE workgroup of 1024 threads (contiguous chunk per thread); out[n] = total.  A chunk of up
// to 32 values stays in registers between the two passes (all its loads in flight at once - the version that read the
// chunk twice with dependent loads and ran a 10-step Hillis-Steele scan behind 20 barriers took 24 us for 20,000 values),
// the 1024 chunk sums are scanned by wave shuffles and one pass over the 16 wave totals.  (`initialize` as a whole did not move
// measurably with it: 0.43-0.45 ms at 200 key frames either way.)
__global__ __launch_bounds__(1024) void k_scan_i32(const int* __restrict__ in, int* __restrict__ out, int n) {
    __shared__ int wtot[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int per = (n + 1023) / 1024;
    const int b = min(t * per, n), e = min(b + per, n);
    constexpr int kKeep = 32;
    int v[kKeep];
    int s = 0;
    if (per <= kKeep) {
#pragma unroll
        for (int u = 0; u < kKeep; ++u) v[u] = (u < per && b + u < e) ? in[b + u] : 0;
#pragma unroll
        for (int u = 0; u < kKeep; ++u) s += v[u];
    } else {
        for (int i = b; i < e; ++i) s += in[i];
    }
    int inc = s;                                  // inclusive scan of the chunk sums inside the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wtot[w];
    int run = base + inc - s;                     // exclusive prefix of this thread's chunk
    if (per <= kKeep) {
#pragma unroll
        for (int u = 0; u < kKeep; ++u)
            if (u < per && b + u < e) { out[b + u] = run; run += v[u]; }
    } else {
        for (int i = b; i < e; ++i) {
            const int x = in[i];
            out[i] = run;
            run += x;
        }
    }
    if (t == 1023) out[n] = base + inc;
}

// Large scans (the digit histograms of a radix pass: bins x workgroups counters) in three coalesced launches: every
// workgroup scans a tile of 2048 counters and leaves the tile total, k_scan_i32 scans the totals, k_scan_tile_add adds
// them back.  (One workgroup walking 10^6 counters with a stride of 10^3 per thread took 0.5 ms per pass.)
constexpr int kScanTile = 2048;
__global__ __launch_bounds__(256) void k_scan_tiles(const int* __restrict__ in, int* __restrict__ out, int n,
                                                     int* __restrict__ tile_sum) {
    __shared__ int tile[kScanTile + kScanTile / 32];   // padded: thread t walks 8 consecutive entries
    __shared__ int wsum[256];
    const int base = blockIdx.x * kScanTile, t = threadIdx.x;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int i = u * 256 + t;
        tile[i + i / 32] = base + i < n ? in[base + i] : 0;
    }
    __syncthreads();
    int v[8], s = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int i = 8 * t + u; v[u] = tile[i + i / 32]; s += v[u]; }
    wsum[t] = s;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int x = t >= off ? wsum[t - off] : 0;
        __syncthreads();
        wsum[t] += x;
        __syncthreads();
    }
    int run = t ? wsum[t - 1] : 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int i = 8 * t + u; tile[i + i / 32] = run; run += v[u]; }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int i = u * 256 + t;
        if (base + i < n) out[base + i] = tile[i + i / 32];
    }
    if (t == 255) tile_sum[blockIdx.x] = wsum[255];
}
__global__ void k_scan_tile_add(int* __restrict__ out, int n, const int* __restrict__ tile_off) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] += tile_off[i / kScanTile];
}

// keys past the real pair count (known only on the device: *count) sort behind every block
constexpr int kRadixItems = 256;   // items per workgroup of a radix pass

__global__ __launch_bounds__(kRadixItems) void k_radix_hist(const int* __restrict__ key, int n, int shift, int nbins,
                                                             int nblk, int* __restrict__ hist) {
    __shared__ int h[1024];
    for (int i = threadIdx.x; i < nbins; i += kRadixItems) h[i] = 0;
    __syncthreads();
    const int i = blockIdx.x * kRadixItems + threadIdx.x;
    if (i < n) atomicAdd(&h[(key[i] >> shift) & (nbins - 1)], 1);
    __syncthreads();
    for (int d = threadIdx.x; d < nbins; d += kRadixItems) hist[(size_t)d * nblk + blockIdx.x] = h[d];
}

// stable: an item's position = start of its digit's run for this workgroup + the number of EARLIER items of the
// workgroup with the same digit (brute force over the <= 255 predecessors staged in LDS)
template <typename V>
__global__ __launch_bounds__(kRadixItems) void k_radix_scatter(const int* __restrict__ key_in, const V* __restrict__ val_in,
                                                                int n, int shift, int nbins, int nblk,
                                                                const int* __restrict__ offs, int* __restrict__ key_out,
                                                                V* __restrict__ val_out) {
    __shared__ int dg[kRadixItems];
    const int i = blockIdx.x * kRadixItems + threadIdx.x;
    int k = 0, d = -1;
    if (i < n) {
        k = key_in[i];
        d = (k >> shift) & (nbins - 1);
    }
    dg[threadIdx.x] = d;
    __syncthreads();
    if (i >= n) return;
    int rank = 0;
    for (int t = 0; t < (int)threadIdx.x; ++t) rank += dg[t] == d ? 1 : 0;
    const int pos = offs[(size_t)d * nblk + blockIdx.x] + rank;
    key_out[pos] = k;
    val_out[pos] = val_in[i];
}

// out[q] = first position in the sorted key list with key >= q, for q = 0 .. nq (out has nq + 1 entries)
__global__ void k_lower_bounds(const int* __restrict__ key, int n, int nq, int* __restrict__ out) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q > nq) return;
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (key[mid] < q) lo = mid + 1; else hi = mid;
    }
    out[q] = lo;
}

// PreEdgeSE2 pose-pose blocks: at most one per (a, b) block goes through the plan (the host checked: no self loops, no
// duplicates; otherwise the edges take the k_odometry / k_reduce_odo fallback and this kernel is not launched)
__global__ void k_plan_odo(int P, int O, const int* __restrict__ o_i, const int* __restrict__ o_j, int* __restrict__ blk_odo,
                           uint8_t* __restrict__ nz) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= O) return;
    const int i = o_i[k], j = o_j[k];
    const int q = blk_index_of(P, min(i, j), max(i, j));
    blk_odo[q] = 2 * k + (i > j ? 1 : 0);
    if (nz) nz[q] = 1;
}

// Packing of the off-diagonal blocks' 16-pair chunks ("groups") into workgroups of kGrpPerWG groups, exactly as the
// sequential host loop does it (a block never straddles a workgroup; a block that does not fit closes the current
// workgroup), but as a scan: the blocks are cut into 256 runs; a run's effect on the fill state `used` (0..27) is a
// function {used -> (used', workgroups closed)}, tabulated by one thread for all 28 start states; one thread composes
// the 256 tables; every thread then replays its run from its now known start and writes the group descriptors.
// out_n[0] = number of workgroups.
__device__ inline void pack_step(int cnt, int gpw, int& used, int& wg, int& first, int& ng, int& chunk) {
    chunk = kChunk;
    ng = max(1, (cnt + chunk - 1) / chunk);
    if (ng > gpw) { chunk = (cnt + gpw - 1) / gpw; ng = (cnt + chunk - 1) / chunk; }
    if (used + ng > gpw) { ++wg; used = 0; }   // flush(): pad the current workgroup
    first = used;
    used += ng;
}
__global__ __launch_bounds__(256) void k_plan_pack(int nblk, const int* __restrict__ blk_a, const int* __restrict__ blk_b,
                                                    const int* __restrict__ blk_ptr, int4* __restrict__ grp,
                                                    int grp_cap_wg, int* __restrict__ out_n, int gpw) {
    __shared__ int tab_used[256][kGrpPerWG], tab_wg[256][kGrpPerWG];
    __shared__ int start_used[256], start_wg[256];
    const int t = threadIdx.x;
    const int per = (nblk + 255) / 256;
    const int b0 = min(t * per, nblk), b1 = min(b0 + per, nblk);
    {   // the run's transfer function for all 28 start states in ONE pass over its blocks
        int used[kGrpPerWG], wgs[kGrpPerWG];
#pragma unroll
        for (int u = 0; u < kGrpPerWG; ++u) { used[u] = u; wgs[u] = 0; }
        for (int kb = b0; kb < b1; ++kb) {
            if (blk_a[kb] == blk_b[kb]) continue;
            const int cnt = blk_ptr[kb + 1] - blk_ptr[kb];
            int ng = max(1, (cnt + kChunk - 1) / kChunk);
            if (ng > gpw) { const int chunk = (cnt + gpw - 1) / gpw; ng = (cnt + chunk - 1) / chunk; }
#pragma unroll
            for (int u = 0; u < kGrpPerWG; ++u) {   // (states >= gpw are never reached)
                if (used[u] + ng > gpw) { ++wgs[u]; used[u] = 0; }
                used[u] += ng;
                if (used[u] == gpw) { used[u] = 0; ++wgs[u]; }
            }
        }
#pragma unroll
        for (int u = 0; u < kGrpPerWG; ++u) { tab_used[t][u] = used[u]; tab_wg[t][u] = wgs[u]; }
    }
    __syncthreads();
    if (t == 0) {
        int used = 0, wg = 0;
        for (int r = 0; r < 256; ++r) {
            start_used[r] = used;
            start_wg[r] = wg;
            wg += tab_wg[r][used];
            used = tab_used[r][used];
        }
        out_n[0] = wg + (used ? 1 : 0);
    }
    __syncthreads();
    int used = start_used[t], wg = start_wg[t];
    for (int kb = b0; kb < b1; ++kb) {
        if (blk_a[kb] == blk_b[kb]) continue;
        const int q0 = blk_ptr[kb], q1 = blk_ptr[kb + 1];
        int first, ng, chunk;
        pack_step(q1 - q0, gpw, used, wg, first, ng, chunk);
        if (wg < grp_cap_wg)
            for (int g = 0; g < ng; ++g) {
                const int a0 = q0 + g * chunk, a1 = min(q1, a0 + chunk);
                grp[(size_t)wg * gpw + first + g] = make_int4(kb, a0, max(a0, a1), first | (ng << 8));
            }
        if (used == gpw) { used = 0; ++wg; }
    }
}
// The same packing for graphs of up to kPackMaxBlk blocks (~440 key frames; the kernel above stays for the rest), spread
// over the machine - as ONE workgroup (k_plan_pack2, rounds 3-4) it was the longest kernel of `initialize`: 100 us at 200
// key frames, most of it twenty dependent rounds of global loads and 632 table steps per lane:
//   k_plan_tab     one workgroup per run of `per` blocks (256 runs): groups per block into LDS (coalesced loads), then
//                  the run's transfer function with LANE = START STATE - {used -> (used', workgroups closed)};
//   k_plan_tabscan one workgroup: the composition of the 256 tables as a parallel (Hillis-Steele) scan in LDS, 8 steps
//                  -> every run's start state, and the number of workgroups;
//   k_plan_place   one workgroup per run: one lane walks the run from its start state (workgroup, first group of every
//                  block, in LDS), then all lanes write the group descriptors.
constexpr int kPackMaxBlk = 96 * 1024, kPackRuns = 256, kPackPerMax = kPackMaxBlk / kPackRuns;
__device__ inline int pack_groups_of(int cnt, int gpw) {
    int ng = max(1, (cnt + kChunk - 1) / kChunk);
    if (ng > gpw) { const int chunk = (cnt + gpw - 1) / gpw; ng = (cnt + chunk - 1) / chunk; }
    return ng;
}
__global__ __launch_bounds__(128) void k_plan_tab(int nblk, const int* __restrict__ blk_a, const int* __restrict__ blk_b,
                                                   const int* __restrict__ blk_ptr, int* __restrict__ tab, int gpw) {
    __shared__ unsigned char ng8[kPackPerMax];
    const int r = blockIdx.x, t = threadIdx.x;
    const int per = (nblk + kPackRuns - 1) / kPackRuns;
    const int b0 = min(r * per, nblk), b1 = min(b0 + per, nblk);
    for (int x = t; x < b1 - b0; x += 128) {
        const int kb = b0 + x;
        ng8[x] = (unsigned char)(blk_a[kb] != blk_b[kb] ? pack_groups_of(blk_ptr[kb + 1] - blk_ptr[kb], gpw) : 0);   // 0 = diagonal block, skipped
    }
    __syncthreads();
    if (t >= 32) return;
    int used = t, wgs = 0;
    for (int x = 0; x < b1 - b0; ++x) {
        const int ng = ng8[x];
        if (ng == 0) continue;   // (uniform over the lanes)
        const bool fl = used + ng > gpw;
        used = (fl ? 0 : used) + ng;
        wgs += fl;
        const bool ex = used == gpw;
        used = ex ? 0 : used;
        wgs += ex;
    }
    if (t < kGrpPerWG) tab[r * kGrpPerWG + t] = used | (wgs << 8);
}
__global__ __launch_bounds__(1024) void k_plan_tabscan(const int* __restrict__ tab, int* __restrict__ start, int* __restrict__ out_n) {
    __shared__ int tab0[kPackRuns * kGrpPerWG], tab1[kPackRuns * kGrpPerWG];
    const int t = threadIdx.x;
    for (int e = t; e < kPackRuns * kGrpPerWG; e += 1024) tab0[e] = tab[e];
    __syncthreads();
    int* src = tab0;
    int* dst = tab1;
    for (int d = 1; d < kPackRuns; d <<= 1) {   // inclusive scan of the tables under composition (earlier run first)
        for (int e = t; e < kPackRuns * kGrpPerWG; e += 1024) {
            const int r = e / kGrpPerWG, u = e - r * kGrpPerWG;
            int v = src[e];
            if (r >= d) {
                const int first = src[(r - d) * kGrpPerWG + u];          // the earlier runs, from state u
                const int second = src[r * kGrpPerWG + (first & 0xff)];  // this run (and those already folded in) from there
                v = (second & 0xff) | (((first >> 8) + (second >> 8)) << 8);
            }
            dst[e] = v;
        }
        __syncthreads();
        int* tmp = src; src = dst; dst = tmp;
    }
    if (t < kPackRuns) start[t] = t ? src[(t - 1) * kGrpPerWG] : 0;   // used | workgroup << 8 in front of run t (from state 0)
    if (t == 0) {
        const int v = src[(kPackRuns - 1) * kGrpPerWG];
        out_n[0] = (v >> 8) + ((v & 0xff) ? 1 : 0);
    }
}
__global__ __launch_bounds__(128) void k_plan_place(int nblk, const int* __restrict__ blk_a, const int* __restrict__ blk_b,
                                                     const int* __restrict__ blk_ptr, const int* __restrict__ start,
                                                     int4* __restrict__ grp, int grp_cap_wg, int gpw) {
    __shared__ unsigned char ng8[kPackPerMax];
    __shared__ int place[kPackPerMax];
    const int r = blockIdx.x, t = threadIdx.x;
    const int per = (nblk + kPackRuns - 1) / kPackRuns;
    const int b0 = min(r * per, nblk), b1 = min(b0 + per, nblk);
    for (int x = t; x < b1 - b0; x += 128) {
        const int kb = b0 + x;
        ng8[x] = (unsigned char)(blk_a[kb] != blk_b[kb] ? pack_groups_of(blk_ptr[kb + 1] - blk_ptr[kb], gpw) : 0);
    }
    __syncthreads();
    if (t == 0) {
        const int v = start[r];
        int used = v & 0xff, wg = v >> 8;
        for (int x = 0; x < b1 - b0; ++x) {
            const int ng = ng8[x];
            if (ng == 0) { place[x] = -1; continue; }
            if (used + ng > gpw) { ++wg; used = 0; }
            place[x] = (wg << 8) | used;
            used += ng;
            if (used == gpw) { used = 0; ++wg; }
        }
    }
    __syncthreads();
    for (int x = t; x < b1 - b0; x += 128) {
        const int pl = place[x];
        if (pl < 0) continue;
        const int wg = pl >> 8, first = pl & 0xff;
        if (wg >= grp_cap_wg) continue;
        const int kb = b0 + x;
        const int q0 = blk_ptr[kb], q1 = blk_ptr[kb + 1];
        const int cnt = q1 - q0;
        int chunk = kChunk;
        int ng = max(1, (cnt + chunk - 1) / chunk);
        if (ng > gpw) { chunk = (cnt + gpw - 1) / gpw; ng = (cnt + chunk - 1) / chunk; }
        for (int g = 0; g < ng; ++g) {
            const int a0 = q0 + g * chunk, a1 = min(q1, a0 + chunk);
            grp[(size_t)wg * gpw + first + g] = make_int4(kb, a0, max(a0, a1), first | (ng << 8));
        }
    }
}
// Independent steps of the plan as ONE launch each (a launch boundary costs about as much as these kernels run: 26 launches
// were 90 us of host time and as much again on the device for a 50-key-frame window).  Items of the merged kernels are
// laid out segment after segment, every segment starting on a workgroup boundary.
//   k_plan_init   = landmark CSR | identity permutation | block tables | cleared descriptors   (need nothing but the graph)
//   k_plan_pairs2 = pair enumeration | tail keys                                             (after the scan of the pair counts)
//   k_plan_bounds = lower bounds (blk_ptr) | split of the sorted (s, t) pairs                (after the pairs are sorted)
__global__ __launch_bounds__(256) void k_plan_init(int L, int E, int P, const int* __restrict__ e_lm, const int* __restrict__ e_kf,
                                                    const uint8_t* __restrict__ fixed, int* __restrict__ lm_ptr,
                                                    int* __restrict__ npair, int* __restrict__ idx, int* __restrict__ blk_a,
                                                    int* __restrict__ blk_b, int* __restrict__ blk_odo, size_t ngrp,
                                                    int4* __restrict__ grp, int b1, int b2, int b3, uint8_t* __restrict__ nz) {
    const int bx = blockIdx.x;
    if (bx < b1) {   // landmark CSR + pairs per landmark
        const int l = bx * 256 + threadIdx.x;
        if (l > L) return;
        int lo = 0, hi = E;   // first edge with e_lm >= l
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (e_lm[mid] < l) lo = mid + 1; else hi = mid;
        }
        lm_ptr[l] = lo;
        if (l == L) return;
        int end = lo;
        while (end < E && e_lm[end] == l) ++end;
        int np = 0;
        for (int s = lo; s < end; ++s) {
            const int a = e_kf[s];
            if (fixed[a]) continue;
            for (int t = s + 1; t < end; ++t) {
                const int b = e_kf[t];
                np += (!fixed[b] && a != b) ? 1 : 0;
            }
        }
        npair[l] = np;
    } else if (bx < b2) {   // edge indices 0 .. E-1 (values of the sort by key frame)
        const int i = (bx - b1) * 256 + threadIdx.x;
        if (i < E) idx[i] = i;
    } else if (bx < b3) {   // blk_a / blk_b / blk_odo
        const int i = (bx - b2) * 256 + threadIdx.x;
        const int a = i / P, b = i - a * P;
        if (a >= P || b < a) return;
        const int q = blk_index_of(P, a, b);
        blk_a[q] = a;
        blk_b[q] = b;
        blk_odo[q] = -1;
        if (nz) nz[q] = 0;   // the block pattern for the solver's order: set by k_plan_odo and k_plan_pairs2
    } else {   // empty group descriptors
        const size_t i = (size_t)(bx - b3) * 256 + threadIdx.x;
        if (i < ngrp) grp[i] = make_int4(-1, 0, 0, 0);
    }
}
__global__ __launch_bounds__(256) void k_plan_pairs2(int L, int P, const int* __restrict__ lm_ptr, const int* __restrict__ e_kf,
                                                      const uint8_t* __restrict__ fixed, const int* __restrict__ pair_base,
                                                      int* __restrict__ key, int2* __restrict__ st, int cap, int big, int b1,
                                                      uint8_t* __restrict__ nz) {
    if ((int)blockIdx.x < b1) {
        const int l = blockIdx.x * 256 + threadIdx.x;
        if (l >= L) return;
        int o = pair_base[l];
        const int beg = lm_ptr[l], end = lm_ptr[l + 1];
        for (int s = beg; s < end; ++s) {
            const int a = e_kf[s];
            if (fixed[a]) continue;
            for (int t = s + 1; t < end; ++t) {
                const int b = e_kf[t];
                if (fixed[b] || a == b) continue;
                const int q = a < b ? blk_index_of(P, a, b) : blk_index_of(P, b, a);
                key[o] = q;
                st[o] = a < b ? make_int2(s, t) : make_int2(t, s);
                if (nz) nz[q] = 1;   // (the same byte from many lanes: any of them may win)
                ++o;
            }
        }
    } else {   // keys past the real pair count (pair_base[L], known only on the device) sort behind every block
        const int i = (blockIdx.x - b1) * 256 + threadIdx.x;
        if (i < cap && i >= pair_base[L]) key[i] = big;
    }
}
__global__ __launch_bounds__(256) void k_plan_bounds(const int* __restrict__ key, int n, int nq, int* __restrict__ out,
                                                      const int2* __restrict__ st, int* __restrict__ pi, int* __restrict__ pj,
                                                      int b1) {
    if ((int)blockIdx.x < b1) {
        const int q = blockIdx.x * 256 + threadIdx.x;
        if (q > nq) return;
        int lo = 0, hi = n;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (key[mid] < q) lo = mid + 1; else hi = mid;
        }
        out[q] = lo;
    } else {
        const int i = (blockIdx.x - b1) * 256 + threadIdx.x;
        if (i < n) { pi[i] = st[i].x; pj[i] = st[i].y; }
    }
}
inline dim3 grid1(size_t n, int block) { return dim3((unsigned)std::max<size_t>((n + block - 1) / block, 1)); }

// (Which blocks of the reduced system are structurally non-zero - a contributor pair or an odometry edge - is what the
// solver's fill-reducing order is chosen from, solve_plan_choose: k_plan_odo and k_plan_pairs2 mark them on the way, so that
// the pattern can travel to the host while the pairs are still being sorted.)
// identity on the diagonal of the padding columns of a permuted system (set once: neither k_reduce2 nor the in-place
// factorisation ever writes anything but zero into padding rows / columns)
// block pattern <-> doubles (dir 0: widen, 1: narrow "any rank has it")
__global__ void k_pattern_widen(int n, uint8_t* __restrict__ nz, double* __restrict__ d, int dir) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (dir == 0) d[i] = nz[i] ? 1.0 : 0.0;
    else nz[i] = d[i] > 0.5 ? 1 : 0;
}
__global__ void k_solver_pads(double* __restrict__ S, int ld, int nsys, const int* __restrict__ col_src) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < nsys && col_src[c] < 0) S[(size_t)c * ld + c] = 1.0;
}

// The host builder of the contributor plan: the reference implementation of the device kernels above and the fallback
// for graphs they do not take (more than 1024 poses; SE2GPU_BA_PLAN=host).  Edges are already sorted by landmark.
struct HostPlan {
    std::vector<int> lm_ptr, pose_ptr, pose_edges, blk_a, blk_b, blk_odo, pair_i, pair_j;
    std::vector<uint8_t> blk_nz;   // per block of the upper triangle: any contributor pair or an odometry edge (solve_plan_choose)
    std::vector<int4> grp;
    int nwg_off = 0;
};
void ba_plan_host(int P, int L, int E, const int* e_kf, const int* e_lm, const uint8_t* fx, int O, const int* o_i,
                  const int* o_j, bool odo_ok, HostPlan& pl, int gpw = kGrpPerWG) {
    pl.lm_ptr.assign(L + 1, 0);
    for (int s = 0; s < E; ++s) pl.lm_ptr[e_lm[s] + 1]++;
    for (int l = 0; l < L; ++l) pl.lm_ptr[l + 1] += pl.lm_ptr[l];
    const std::vector<int>& lm_ptr = pl.lm_ptr;
    // --- pose -> edges CSR
    pl.pose_ptr.assign(P + 1, 0);
    pl.pose_edges.resize(E);
    for (int s = 0; s < E; ++s) pl.pose_ptr[e_kf[s] + 1]++;
    for (int p = 0; p < P; ++p) pl.pose_ptr[p + 1] += pl.pose_ptr[p];
    {
        std::vector<int> f(pl.pose_ptr.begin(), pl.pose_ptr.end() - 1);
        for (int s = 0; s < E; ++s) pl.pose_edges[f[e_kf[s]]++] = s;
    }
    // --- reduced-system block plan: upper-triangular (a <= b) blocks, contributor pairs per block
    const int nblk = P * (P + 1) / 2;
    std::vector<int> blk_ptr(nblk + 1, 0);
    pl.blk_a.resize(nblk);
    pl.blk_b.resize(nblk);
    for (int a = 0; a < P; ++a)
        for (int b = a; b < P; ++b) {
            pl.blk_a[blk_index_of(P, a, b)] = a;
            pl.blk_b[blk_index_of(P, a, b)] = b;
        }
    // every unordered pair of observations of a landmark by two different free poses contributes to one block; the pair
    // is stored with the lower pose first.  Pass 1 records the block of each pair (in landmark order), pass 2 scatters.
    std::vector<int> pkey;
    std::vector<int2> pst;
    pkey.reserve(4 * (size_t)E);
    pst.reserve(4 * (size_t)E);
    for (int l = 0; l < L; ++l)
        for (int s = lm_ptr[l]; s < lm_ptr[l + 1]; ++s) {
            const int a = e_kf[s];
            if (fx[a]) continue;
            for (int t = s + 1; t < lm_ptr[l + 1]; ++t) {
                const int b = e_kf[t];
                if (fx[b] || a == b) continue;
                const int q = a < b ? blk_index_of(P, a, b) : blk_index_of(P, b, a);
                pkey.push_back(q);
                pst.push_back(a < b ? make_int2(s, t) : make_int2(t, s));
                blk_ptr[q + 1]++;
            }
        }
    for (int k = 0; k < nblk; ++k) blk_ptr[k + 1] += blk_ptr[k];
    const size_t npairs = (size_t)blk_ptr[nblk];
    pl.pair_i.resize(npairs);
    pl.pair_j.resize(npairs);
    {
        std::vector<int> f(blk_ptr.begin(), blk_ptr.end() - 1);
        for (size_t k = 0; k < pkey.size(); ++k) {
            const int q = f[pkey[k]]++;
            pl.pair_i[q] = pst[k].x;
            pl.pair_j[q] = pst[k].y;
        }
    }
    pl.blk_odo.assign(nblk, -1);
    if (odo_ok)
        for (int k = 0; k < O; ++k) {
            const int i = o_i[k], j = o_j[k];
            pl.blk_odo[blk_index_of(P, std::min(i, j), std::max(i, j))] = 2 * k + (i > j ? 1 : 0);
        }
    pl.blk_nz.resize(nblk);
    for (int q = 0; q < nblk; ++q) pl.blk_nz[q] = (blk_ptr[q + 1] > blk_ptr[q] || pl.blk_odo[q] >= 0) ? 1 : 0;
    for (int k = 0; k < O; ++k) pl.blk_nz[blk_index_of(P, std::min(o_i[k], o_j[k]), std::max(o_i[k], o_j[k]))] = 1;
    // pack 16-pair chunks of the off-diagonal blocks into workgroups of 28 nine-lane groups
    std::vector<int4>& grp = pl.grp;
    grp.clear();
    int used = 0;  // groups used in the current workgroup
    auto flush = [&]() {
        while (used % gpw) { grp.push_back(make_int4(-1, 0, 0, 0)); ++used; }
        used = 0;
    };
    for (int kb = 0; kb < nblk; ++kb) {
        if (pl.blk_a[kb] == pl.blk_b[kb]) continue;
        const int q0 = blk_ptr[kb], q1 = blk_ptr[kb + 1];
        int chunk = kChunk;
        int ng = std::max(1, (q1 - q0 + chunk - 1) / chunk);
        if (ng > gpw) { chunk = (q1 - q0 + gpw - 1) / gpw; ng = (q1 - q0 + chunk - 1) / chunk; }
        if (used + ng > gpw) flush();
        const int first = used;
        for (int t = 0; t < ng; ++t) {
            const int a0 = q0 + t * chunk, a1 = std::min(q1, a0 + chunk);
            grp.push_back(make_int4(kb, a0, std::max(a0, a1), first | (ng << 8)));
            ++used;
        }
        if (used == gpw) used = 0;
    }
    flush();
    if (grp.empty()) grp.assign(gpw, make_int4(-1, 0, 0, 0));
    pl.nwg_off = (int)(grp.size() / gpw);
}

// =============================================================================================
// Plan of the dense pose solve (round 3): which tiles exist, what every tile task waits for, and - for the SE(2) model on
// one GPU - a fill-reducing ORDER of the poses.  The reference hands the reduced system to CHOLMOD, whose ordering
// exploits that key frames far apart share no landmarks; the dataflow solve gets the same from a nested-dissection
// order: the tile tasks of two interiors that no landmark couples depend on disjoint block columns and run side by side,
// so the chain of block columns (7 us each) is the separator's plus ONE interior's instead of all of them (200 key frames
// on a loop: 19 -> 15).  Everything here is host code on the P x P block pattern; correctness never depends on the
// heuristics that propose an order - the tile structure of whatever order wins is derived from the pattern itself by a
// symbolic factorisation (fill included), and an order only wins by a shorter chain.
//   layout     partitions of poses; a partition starts on a tile boundary (identity padding behind it)
//   tile L     symbolic Cholesky on the 32-wide tiles of the permuted pattern; the rhs row is a tile row of its own
//   tile R     R = L^-T restricted to what x = R y needs: R(r,r), R(r,j) <- any m in [r, j): R(r,m) and L(j,m)
//   tasks      per block column: its L tiles (diagonal first), its R tiles; the x tasks last.  A task lists the block
//              columns m < j with L(j,m) != 0 (bit 15: its own tile row is non-zero in m as well)
// =============================================================================================
struct SolvePlan {
    int nsys = 0, nbc = 0, nt = 0, depth = 0, ntile = 0;
    bool permuted = false;
    std::vector<int> pose_off, col_src;
    std::vector<int4> tasks;
    std::vector<int> deps;
};

// parts: the poses of every partition in order; pad: partitions (and therefore the rhs row) start on tile boundaries;
// pat: nullptr (dense) or a SYMMETRIC P x P byte pattern
void solve_plan_build(int P, int D, const uint8_t* pat, const std::vector<std::vector<int>>& parts, bool pad, SolvePlan& sp, int nb = kNB) {
    sp.pose_off.assign(P, 0);
    int c = 0;
    for (const auto& part : parts) {
        for (int p : part) { sp.pose_off[p] = c; c += D; }
        if (pad) c = (c + nb - 1) / nb * nb;
    }
    sp.nsys = c;
    sp.permuted = pad;
    sp.col_src.assign(sp.nsys, -1);
    for (int p = 0; p < P; ++p)
        for (int k = 0; k < D; ++k) sp.col_src[sp.pose_off[p] + k] = D * p + k;
    const int nbc = (sp.nsys + nb - 1) / nb;
    const int ld = ((sp.nsys + 1 + nb - 1) / nb) * nb, nt = ld / nb;
    const int it = sp.nsys / nb;                     // tile row of the rhs row (== nbc when the system is padded)
    sp.nbc = nbc;
    sp.nt = nt;
    // tile pattern of the lower triangle (rows 0 .. nt-1: the rhs tile row is dense)
    std::vector<uint8_t> Lt((size_t)nt * nbc, 0);
    auto L = [&](int i, int j) -> uint8_t& { return Lt[(size_t)i * nbc + j]; };
    for (int j = 0; j < nbc; ++j) L(j, j) = 1;
    // pose a touches tile rows t0[a] .. t1[a]; the tiles its neighbours touch as one 64-bit mask per pose (nt <= 64 is what
    // the dataflow solve takes, larger systems go to k_chol_step and need no tile pattern): P^2 branch-free ORs instead of
    // P^2 / 2 tests with up to four stores each - this loop was most of the 0.18 ms the plan cost at 200 key frames
    std::vector<int> t0(P), t1(P);
    for (int p = 0; p < P; ++p) { t0[p] = sp.pose_off[p] / nb; t1[p] = (sp.pose_off[p] + D - 1) / nb; }
    if (nt <= 64) {
        std::vector<unsigned long long> tm(P), rowmask(nt, 0ull);
        for (int p = 0; p < P; ++p) tm[p] = (t1[p] >= 63 ? ~0ull : (1ull << (t1[p] + 1)) - 1) & ~((1ull << t0[p]) - 1);
        for (int a = 0; a < P; ++a) {
            unsigned long long m = tm[a];
            if (pat) {
                const uint8_t* row = pat + (size_t)a * P;
                for (int b = 0; b < P; ++b) m |= tm[b] & (0ull - (unsigned long long)(row[b] != 0));   // (pat is symmetric: solve_plan_choose)
            } else {
                m = ~0ull;
            }
            for (int ta = t0[a]; ta <= t1[a]; ++ta) rowmask[ta] |= m;
        }
        for (int i = 0; i < nbc; ++i)
            for (int j = 0; j <= i; ++j)
                if (((rowmask[i] >> j) | (rowmask[j] >> i)) & 1ull) L(i, j) = 1;
    } else {
        for (int a = 0; a < P; ++a)
            for (int b = 0; b <= a; ++b) {
                if (pat && !pat[(size_t)a * P + b]) continue;
                for (int ta = t0[a]; ta <= t1[a]; ++ta)
                    for (int tb = t0[b]; tb <= t1[b]; ++tb) {
                        if (ta >= tb) L(ta, tb) = 1; else L(tb, ta) = 1;
                    }
            }
    }
    for (int j = 0; j < nbc; ++j) L(it, j) = 1;        // y = L^-1 b
    for (int j = 0; j < nbc; ++j)                      // symbolic factorisation: fill between the rows of a column
        for (int i = j + 1; i < nt; ++i) {
            if (!L(i, j)) continue;
            for (int k = i; k < nt; ++k)
                if (L(k, j) && k < nt && i < nbc) L(k, i) = 1;
        }
    std::vector<uint8_t> Rt((size_t)nbc * nbc, 0);
    auto R = [&](int r, int j) -> uint8_t& { return Rt[(size_t)r * nbc + j]; };
    for (int r = 0; r < nbc; ++r) {
        R(r, r) = 1;
        for (int j = r + 1; j < nbc; ++j)
            for (int m = r; m < j; ++m)
                if (R(r, m) && L(j, m)) { R(r, j) = 1; break; }
    }
    sp.tasks.clear();
    sp.deps.clear();
    sp.ntile = 0;
    std::vector<int> depth(nbc, 0);
    sp.depth = 0;
    for (int j = 0; j < nbc; ++j) {
        int dj = 0;
        for (int m = 0; m < j; ++m)
            if (L(j, m)) dj = std::max(dj, depth[m]);
        depth[j] = dj + 1;
        sp.depth = std::max(sp.depth, depth[j]);
        for (int i = j; i < nt; ++i) {
            if (!L(i, j)) continue;
            const int first = (int)sp.deps.size();
            for (int m = 0; m < j; ++m)
                if (L(j, m)) sp.deps.push_back(m | ((i != j && L(i, m)) ? 0x8000 : 0));
            sp.tasks.push_back(make_int4(i, j, first, (int)sp.deps.size()));
            ++sp.ntile;
        }
        for (int r = 0; r < j; ++r) {
            if (!R(r, j)) continue;
            const int first = (int)sp.deps.size();
            for (int m = 0; m < j; ++m)
                if (L(j, m)) sp.deps.push_back(m | ((m >= r && R(r, m)) ? 0x8000 : 0));
            sp.tasks.push_back(make_int4(r | (1 << 16), j, first, (int)sp.deps.size()));
            ++sp.ntile;
        }
    }
    for (int r = 0; r < nbc; ++r) {   // x = R y, one task per tile row
        const int first = (int)sp.deps.size();
        for (int j = r; j < nbc; ++j)
            if (R(r, j)) sp.deps.push_back(j);
        sp.tasks.push_back(make_int4(r | (2 << 16), 0, first, (int)sp.deps.size()));
    }
}

// candidate orders for a pattern whose natural order is a band - open or closed to a ring (key frames in sequence, a loop
// closure at most between the ends) - and the choice by chain length
void solve_plan_choose(int P, int D, const uint8_t* pat_in, bool allow_nd, SolvePlan& best, bool symmetric = false, int nb = kNB) {
    std::vector<int> all(P);
    for (int p = 0; p < P; ++p) all[p] = p;
    std::vector<uint8_t> sym;
    const uint8_t* pat = pat_in;
    if (pat_in && !symmetric) {   // a caller's pattern may be one-sided (the debug entry point); initialize hands over a symmetric one
        sym.assign((size_t)P * P, 0);
        for (int a = 0; a < P; ++a)
            for (int b = 0; b < P; ++b) sym[(size_t)a * P + b] = (uint8_t)((pat_in[(size_t)a * P + b] | pat_in[(size_t)b * P + a]) != 0);
        pat = sym.data();
    }
    solve_plan_build(P, D, pat, {all}, false, best, nb);
    if (!allow_nd || !pat || D * P < 4 * nb) return;
    // widths of the band: which distances |a - b| occur at all (one OR per entry, no branch), then the largest linear and
    // cyclic distance among them
    int w_lin = 0, w_cyc = 0;
    {
        std::vector<uint8_t> dist(P, 0);
        for (int a = 1; a < P; ++a) {
            const uint8_t* row = pat + (size_t)a * P;
            uint8_t* dd = dist.data() + a;       // dd[-b] = distance a - b
            for (int b = 0; b < a; ++b) dd[-b] |= row[b];
        }
        for (int d = 1; d < P; ++d)
            if (dist[d]) { w_lin = std::max(w_lin, d); w_cyc = std::max(w_cyc, std::min(d, P - d)); }
    }
    const int min_interior = (nb + D - 1) / D;       // an interior below one tile of poses is not worth a partition
    std::function<void(int, int, int, int, std::vector<std::vector<int>>&)> linear = [&](int lo, int hi, int w, int levels,
                                                                                         std::vector<std::vector<int>>& out) {
        const int n = hi - lo;
        if (n <= 0) return;
        if (levels == 0 || n < 2 * min_interior + w) {
            std::vector<int> v;
            for (int p = lo; p < hi; ++p) v.push_back(p);
            out.push_back(v);
            return;
        }
        const int s0 = lo + (n - w) / 2, s1 = s0 + w;
        linear(lo, s0, w, levels - 1, out);
        linear(s1, hi, w, levels - 1, out);
        std::vector<int> sep;
        for (int p = s0; p < s1; ++p) sep.push_back(p);
        out.push_back(sep);
    };
    // the levels of a dissection that cannot split any further give the same partitions again: every distinct candidate is
    // built once
    std::vector<std::vector<std::vector<int>>> seen;
    auto consider = [&](const std::vector<std::vector<int>>& parts) {
        for (const auto& sd : seen)
            if (sd == parts) return;
        seen.push_back(parts);
        SolvePlan cand;
        solve_plan_build(P, D, pat, parts, true, cand, nb);
        // a shorter chain of block columns wins; the natural order keeps ties (no padding, fewer tiles)
        if (cand.depth < best.depth && cand.nt <= 64) best = std::move(cand);
    };
    if (w_lin >= 1 && 3 * w_lin <= P)                 // open band: separators of the band's width, up to three levels
        for (int lv = 1; lv <= 3; ++lv) {
            std::vector<std::vector<int>> parts;
            linear(0, P, w_lin, lv, parts);
            if (parts.size() > 1) consider(parts);
        }
    if (w_cyc < w_lin && w_cyc >= 1 && 2 * w_cyc + 2 * min_interior <= P) {   // ring: two separators, joined into the last partition
        const int h = (P + 1) / 2;
        auto ring = [&](int w, int sw, int lv) {   // separators of sw >= w poses each; inner levels use the band's width
            std::vector<std::vector<int>> parts;
            linear(sw, h, w, lv, parts);
            linear(h + sw, P, w, lv, parts);
            std::vector<int> sep;
            for (int p = 0; p < sw; ++p) sep.push_back(p);
            for (int p = h; p < std::min(h + sw, P); ++p) sep.push_back(p);
            parts.push_back(sep);
            consider(parts);
        };
        const int w = w_cyc;
        for (int lv = 0; lv <= 2; ++lv) ring(w, w, lv);
        // partitions are padded to tile boundaries, so the chain is ceil(arc / tile) + ceil(separators / tile) block columns: a few
        // poses moved from the arcs into the separators can take a tile off the arcs without adding one to the separators
        // (200 key frames, band 43: arcs 57 -> 52 poses = 6 -> 5 tiles, separators 86 -> 96 poses = 9 tiles either way)
        auto tiles = [&](int poses) { return (D * poses + nb - 1) / nb; };
        int sw_best = w, est_best = tiles(std::max(h - w, P - h - w)) + tiles(2 * w);
        for (int sw = w + 1; sw <= w + nb && 2 * sw + 2 * min_interior <= P; ++sw) {
            const int est = tiles(std::max(h - sw, P - h - sw)) + tiles(2 * sw);
            if (est < est_best) { est_best = est; sw_best = sw; }
        }
        if (sw_best != w) ring(w, sw_best, 0);
    }
}

// one copy stream per device, shared by all handles, never destroyed
int ba_copy_stream(int device, hipStream_t* out) {
    static std::mutex mu;
    static hipStream_t streams[64] = {};
    std::lock_guard<std::mutex> lk(mu);
    SE2_REQUIRE(device >= 0 && device < 64, SE2GPU_ERR_INVALID, "device %d", device);
    if (!streams[device]) SE2_HIP(hipStreamCreateWithFlags(&streams[device], hipStreamNonBlocking));
    *out = streams[device];
    return SE2GPU_OK;
}

int ba_allreduce(se2gpu_ba* h, double* ptr, size_t count);

int ba_upload_graph(se2gpu_ba* h) {
    static const bool trace = [] { const char* e = getenv("SE2GPU_BA_INIT_TRACE"); return e && e[0] == '1'; }();
    const auto t_begin = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (trace)
            std::fprintf(stderr, "[ba init] %-28s %8.1f us\n", what,
                         std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count());
    };
    const int P = (int)h->pose_ids.size(), L = (int)h->lm_ids.size();
    const int E = h->bulk_E ? h->bulk_E : (int)h->he_kf.size(), O_in = h->model ? (int)h->odo3.size() : (int)h->odo.size();
    const int D = h->D, ps = h->ps, gpw = h->model ? kGrpPerWG3 : kGrpPerWG;
    const int DS = D * (D + 1) / 2;   // entries of a symmetric pose block
    SE2_REQUIRE(P > 0, SE2GPU_ERR_STATE, "initialize: no pose vertices");
    SE2_REQUIRE(!h->model || (h->world == 1 && !h->allreduce), SE2GPU_ERR_STATE, "the SE3 model is single-GPU");
    SE2_REQUIRE(h->have_cam, SE2GPU_ERR_STATE, "initialize: add_cam was not called");
    SE2_REQUIRE(!h->huber_mixed, SE2GPU_ERR_INVALID, "all EdgeSE2XYZ must share one Huber delta (Map.cpp:977)");
    h->P = P; h->L = L; h->E = E; h->O = O_in;
    h->cam.huber = E ? h->huber_delta : 0.0;
    h->cam3 = Cam3{h->cam.fx, h->cam.cx, h->cam.cy, h->cam.huber};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) h->cam.Rcb[i * 3 + j] = h->Rbc[j * 3 + i];
    for (int i = 0; i < 3; ++i)
        h->cam.tcb[i] = -(h->cam.Rcb[i * 3] * h->tbc[0] + h->cam.Rcb[i * 3 + 1] * h->tbc[1] + h->cam.Rcb[i * 3 + 2] * h->tbc[2]);

    // --- observation edges sorted by landmark.  Map::loadLocalGraph adds them map point by map point, i.e. already in
    // order: one pass checks that (and sums the pairs per landmark for the buffer sizes); otherwise a stable counting sort.
    const int* e_kf = h->bulk_E ? h->bulk_kf : h->he_kf.data();
    const int* e_lm = h->bulk_E ? h->bulk_lm : h->he_lm.data();
    const double* e_uv = h->bulk_E ? h->bulk_uv : h->he_uv.data();
    const double* e_info = h->bulk_E ? h->bulk_info : h->he_info.data();
    std::vector<int> s_kf, s_lm;
    std::vector<double> s_uv, s_info;
    bool sorted = true, in_range = true;
    size_t npairs_max = 0;   // pairs of observations per landmark (an upper bound of the plan's pairs: fixed poses drop out)
    {   // index range (borrowed bulk arrays are validated here), order, pairs per landmark: three branch-free loops (the
        // first two vectorise; the third carries only the start of the current run.  Measured and dropped in round 4: one
        // fused pass over both arrays, and the third loop as four independent chains - both slower with clang's code)
        unsigned bad = 0;
        for (int k = 0; k < E; ++k) bad |= (unsigned)((unsigned)e_kf[k] >= (unsigned)P) | (unsigned)((unsigned)e_lm[k] >= (unsigned)L);
        in_range = !bad;
        int unsorted = 0;
        for (int k = 1; k < E; ++k) unsorted |= e_lm[k] < e_lm[k - 1];
        sorted = !unsorted;
    }
    SE2_REQUIRE(in_range, SE2GPU_ERR_INVALID, "an edge references a vertex out of range");
    h->edge_perm.clear();
    if (!sorted) {
        std::vector<int> ptr(L + 1, 0);
        for (int k = 0; k < E; ++k) ptr[e_lm[k] + 1]++;
        for (int l = 0; l < L; ++l) ptr[l + 1] += ptr[l];
        s_kf.resize(E); s_lm.resize(E); s_uv.resize(2 * (size_t)E); s_info.resize(3 * (size_t)E);
        h->edge_perm.assign(E, 0);
        for (int k = 0; k < E; ++k) {
            const int t = ptr[e_lm[k]]++;
            h->edge_perm[t] = k;
            s_kf[t] = e_kf[k]; s_lm[t] = e_lm[k];
            s_uv[2 * (size_t)t] = e_uv[2 * (size_t)k]; s_uv[2 * (size_t)t + 1] = e_uv[2 * (size_t)k + 1];
            for (int c = 0; c < 3; ++c) s_info[3 * (size_t)t + c] = e_info[3 * (size_t)k + c];
        }
        e_kf = s_kf.data(); e_lm = s_lm.data(); e_uv = s_uv.data(); e_info = s_info.data();
    }
    auto count_pairs = [&]() {   // pairs = sum over the edges (sorted by landmark) of their position inside their landmark's run
        size_t np = 0;
        int run_start = 0;
        for (int k = 1; k < E; ++k) {
            run_start = e_lm[k] != e_lm[k - 1] ? k : run_start;
            np += (size_t)(k - run_start);
        }
        return np;
    };
    npairs_max = count_pairs();   // (on the sorted order in either case)
    SE2_REQUIRE(npairs_max < (size_t)1 << 30, SE2GPU_ERR_CAPACITY, "the contributor plan would hold %zu pairs", npairs_max);
    // --- odometry (tiny: host)
    // pose graph: the EdgeSE3 edges of one unordered key-frame pair form a slot (a < b); the "odometry" arrays of the
    // plan are the slots
    std::vector<int> pg_slot_a, pg_slot_b, pg_slot_ptr, pg_i, pg_j;
    std::vector<double> pg_meas, pg_info;
    int O = O_in;
    if (h->model == 2) {
        // (g2o's initializeOptimization(0) takes the edges of level 0: an EdgeSE3 moved to another level by
        // se2gpu_ba_set_edge_level - GlobalMapper.cpp:437,466 - stays in the handle and out of the device graph)
        std::vector<int> order;
        for (int k = 0; k < (int)h->odo3.size(); ++k)
            if (h->odo3[k].level == 0) order.push_back(k);
        const int NE = (int)order.size();
        auto key = [&](int k) { return std::make_pair(std::min(h->odo3[k].i, h->odo3[k].j), std::max(h->odo3[k].i, h->odo3[k].j)); };
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return key(x) < key(y); });
        h->pg_perm = order;
        h->pg_edges = NE;
        pg_i.resize(NE); pg_j.resize(NE); pg_meas.resize(12 * (size_t)NE); pg_info.resize(36 * (size_t)NE);
        for (int t = 0; t < NE; ++t) {
            const auto& e = h->odo3[order[t]];
            pg_i[t] = e.i; pg_j[t] = e.j;
            std::memcpy(&pg_meas[12 * (size_t)t], e.meas, 96);
            std::memcpy(&pg_info[36 * (size_t)t], e.info, 288);
            if (t == 0 || key(order[t]) != key(order[t - 1])) {
                pg_slot_a.push_back(key(order[t]).first);
                pg_slot_b.push_back(key(order[t]).second);
                pg_slot_ptr.push_back(t);
            }
        }
        pg_slot_ptr.push_back(NE);
        O = (int)pg_slot_a.size();
        h->O = O;
    }
    std::vector<int> o_i(O), o_j(O), podo_ptr(P + 1, 0), podo_item(2 * (size_t)O);
    const int ms = h->model ? 12 : 3, is = h->model ? 36 : 9;   // doubles per measurement / information
    std::vector<double> o_meas(h->model == 2 ? 0 : (size_t)ms * O), o_info(h->model == 2 ? 0 : (size_t)is * O);
    for (int k = 0; k < O; ++k) {
        if (h->model == 2) {
            o_i[k] = pg_slot_a[k]; o_j[k] = pg_slot_b[k];
        } else if (h->model) {
            o_i[k] = h->odo3[k].i; o_j[k] = h->odo3[k].j;
            std::memcpy(&o_meas[12 * (size_t)k], h->odo3[k].meas, 96);
            std::memcpy(&o_info[36 * (size_t)k], h->odo3[k].info, 288);
        } else {
            o_i[k] = h->odo[k].i; o_j[k] = h->odo[k].j;
            std::memcpy(&o_meas[3 * k], h->odo[k].meas, 24);
            std::memcpy(&o_info[9 * k], h->odo[k].info, 72);
        }
        podo_ptr[o_i[k] + 1]++;
        podo_ptr[o_j[k] + 1]++;
    }
    for (int p = 0; p < P; ++p) podo_ptr[p + 1] += podo_ptr[p];
    {
        std::vector<int> f(podo_ptr.begin(), podo_ptr.end() - 1);
        for (int k = 0; k < O; ++k) {
            podo_item[f[o_i[k]]++] = 2 * k;
            podo_item[f[o_j[k]]++] = 2 * k + 1;
        }
    }
    // PreEdgeSE2 pose-pose blocks: at most one per (a, b) block goes through the plan; self loops / duplicates fall back
    h->odo_fallback = false;
    {
        std::vector<std::pair<int, int>> seen(O);
        for (int k = 0; k < O; ++k) {
            if (o_i[k] == o_j[k]) h->odo_fallback = true;
            seen[k] = {std::min(o_i[k], o_j[k]), std::max(o_i[k], o_j[k])};
        }
        std::sort(seen.begin(), seen.end());
        for (int k = 1; k < O; ++k)
            if (seen[k] == seen[k - 1]) h->odo_fallback = true;
    }
    const int nblk = P * (P + 1) / 2;
    h->nblk = nblk;
    const char* plan_env = getenv("SE2GPU_BA_PLAN");
    const bool device_plan = P <= 1024 && !(plan_env && std::strcmp(plan_env, "host") == 0);
    lap("edge check + odometry");
    HostPlan pl;
    if (!device_plan) {
        ba_plan_host(P, L, E, e_kf, e_lm, h->h_fixed.data(), O, o_i.data(), o_j.data(), !h->odo_fallback, pl, gpw);
        h->nwg_off = pl.nwg_off;
        lap("host plan");
    }
    hipStream_t st = h->stream;
    const int n = D * P;
    // All uploads go through ONE pinned arena into ONE device arena with ONE copy (a copy from pageable memory is staged
    // synchronously by the runtime, about 20 us each; two dozen separate enqueues were a tenth of a local window's cycle).
    struct Staged { std::function<void(uint8_t*)> bind; const void* src; size_t bytes, off; };
    std::vector<Staged> staged;
    size_t staged_bytes = 0;
    auto stage = [&](auto& buf, const auto* src, size_t count) {
        using T = std::remove_reference_t<decltype(*buf.p)>;
        const size_t bytes = count * sizeof(T);
        auto* bp = &buf;
        staged.push_back(Staged{[bp](uint8_t* base) { bp->alias(reinterpret_cast<T*>(base)); }, src, bytes, staged_bytes});
        staged_bytes += (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255;
    };
    stage(h->poses0, h->h_poses.data(), h->h_poses.size());
    stage(h->lms0, h->h_lms.data(), h->h_lms.size());
    stage(h->fixed, h->h_fixed.data(), h->h_fixed.size());
    stage(h->e_kf, e_kf, (size_t)E);
    stage(h->e_lm, e_lm, (size_t)E);
    stage(h->podo_ptr, podo_ptr.data(), podo_ptr.size());
    stage(h->podo_item, podo_item.data(), podo_item.size());
    stage(h->o_i, o_i.data(), o_i.size());
    stage(h->o_j, o_j.data(), o_j.size());
    stage(h->o_meas, o_meas.data(), o_meas.size());
    stage(h->o_info, o_info.data(), o_info.size());
    if (h->model == 2) {
        stage(h->slot_a, pg_slot_a.data(), pg_slot_a.size());
        stage(h->slot_ptr, pg_slot_ptr.data(), pg_slot_ptr.size());
        stage(h->pe_i, pg_i.data(), pg_i.size());
        stage(h->pe_j, pg_j.data(), pg_j.size());
        stage(h->pe_meas, pg_meas.data(), pg_meas.size());
        stage(h->pe_info, pg_info.data(), pg_info.size());
    }
    if (h->model) {
        stage(h->prior_has, h->h_prior_has.data(), h->h_prior_has.size());
        stage(h->prior_meas, h->h_prior_meas.data(), h->h_prior_meas.size());
        stage(h->prior_info, h->h_prior_info.data(), h->h_prior_info.size());
    }
    if (h->lg_active) {
        SE2_REQUIRE(sorted && h->lg_sigma2.size() == (size_t)E, SE2GPU_ERR_STATE,
                    "load_local_graph must not be mixed with add_edge calls");
        stage(h->d_lg_lc, h->lg_lc.data(), h->lg_lc.size());
        stage(h->d_lg_lw, h->lg_lw.data(), h->lg_lw.size());
        stage(h->d_lg_sigma2, h->lg_sigma2.data(), h->lg_sigma2.size());
        stage(h->d_lg_Rcw, h->lg_Rcw.data(), h->lg_Rcw.size());
        stage(h->d_lg_twb, h->lg_twb.data(), h->lg_twb.size());
    }
    if (!device_plan) {
        stage(h->grp, pl.grp.data(), pl.grp.size());
        stage(h->lm_ptr, pl.lm_ptr.data(), pl.lm_ptr.size());
        stage(h->pose_ptr, pl.pose_ptr.data(), pl.pose_ptr.size());
        stage(h->pose_edges, pl.pose_edges.data(), pl.pose_edges.size());
        stage(h->blk_a, pl.blk_a.data(), pl.blk_a.size());
        stage(h->blk_b, pl.blk_b.data(), pl.blk_b.size());
        stage(h->blk_odo, pl.blk_odo.data(), pl.blk_odo.size());
        stage(h->pair_i, pl.pair_i.data(), pl.pair_i.size());
        stage(h->pair_j, pl.pair_j.data(), pl.pair_j.size());
    }
    SE2_CHECK(h->poses_a.reserve((size_t)ps * P));
    SE2_CHECK(h->poses_b.reserve((size_t)ps * P));
    SE2_CHECK(h->lms_a.reserve(3 * (size_t)L + 1));
    SE2_CHECK(h->lms_b.reserve(3 * (size_t)L + 1));
    SE2_CHECK(h->Hpl.reserve((size_t)D * 3 * E + 1));
    if (h->model) SE2_CHECK(h->Y.reserve((size_t)D * 3 * E + 1));   // (the SE(2) model's whitened records need no Y_e)
    SE2_CHECK(h->Hpp_e.reserve((size_t)DS * E + 1));
    SE2_CHECK(h->bp_e.reserve((size_t)D * E + 1));
    SE2_CHECK(h->Dg.reserve((size_t)(DS + 2 * D) * E + 1));
    SE2_CHECK(h->Hll.reserve(6 * (size_t)L + 1));
    SE2_CHECK(h->bl.reserve(3 * (size_t)L + 1));
    SE2_CHECK(h->Dinv.reserve(6 * (size_t)L + 1));
    SE2_CHECK(h->z.reserve(3 * (size_t)L + 1));
    SE2_CHECK(h->Hpp.reserve((size_t)D * D * P));
    SE2_CHECK(h->bp.reserve((size_t)D * P));
    SE2_CHECK(h->diag3.reserve((size_t)D * P));
    SE2_CHECK(h->Oii.reserve((size_t)D * D * O + 1));
    SE2_CHECK(h->Ojj.reserve((size_t)D * D * O + 1));
    SE2_CHECK(h->Oij.reserve((size_t)D * D * O + 1));
    SE2_CHECK(h->obi.reserve((size_t)D * O + 1));
    SE2_CHECK(h->obj.reserve((size_t)D * O + 1));
    if (h->model) {
        SE2_CHECK(h->pb.reserve(6 * (size_t)P));
        SE2_CHECK(h->edge_chi2.reserve((size_t)std::max(E, h->pg_edges) + 1));
        if (h->model == 2) SE2_CHECK(h->ph.reserve(36 * (size_t)P));
    }
    SE2_CHECK(h->xp.reserve(n));
    h->nparts = (L * kGroup + kBlock - 1) / kBlock;
    SE2_CHECK(h->part.reserve(2 * (size_t)std::max(h->nparts, 1)));
    SE2_CHECK(h->scal.reserve(8 + (size_t)h->world));
    SE2_CHECK(h->ctl.reserve(1));
    h->ld = ((n + 1 + kNB - 1) / kNB) * kNB;
    const size_t nred = (size_t)h->ld * h->ld + 4;
    {
        const char* env = getenv("SE2GPU_BA_HOST_SOLVE");
        h->host_solve = env && env[0] == '1';
    }
    if (h->ar_buffer) {
        h->red = (double*)h->ar_buffer;
    } else {
        SE2_CHECK(h->red_own.reserve(nred));
        h->red = h->red_own.p;
    }
    if (h->host_solve) SE2_CHECK(h->h_red.reserve(nred));
    SE2_CHECK(h->Rinv.reserve(chol_work_doubles(h->ld)));
    // (the tile tasks of the solve are planned after the block pattern is known: ba_setup_solver, below)
    static const bool nd_env = [] { const char* e = getenv("SE2GPU_BA_ND"); return !(e && e[0] == '0'); }();
    // (a sharded run can re-order too: every rank must choose the SAME order, so the ranks' block patterns - each sees its own
    // landmarks' pairs only - are merged by one small all-reduce below; a caller-owned exchange buffer keeps the natural
    // layout it was sized for)
    const bool nd_sharded = h->allreduce && h->world > 1;
    const bool nd_possible = nd_env && h->model == 0 && !h->host_solve && !h->ar_buffer && D * P >= 4 * kNB && P <= 1024 &&
                             (!nd_sharded || device_plan);
    // last in the arena: the measurements and information matrices (copied on their own stream, see below; with a local
    // graph loaded through se2gpu_ba_load_local_graph the information is evaluated on the device and not copied at all)
    const size_t big_off = staged_bytes;
    stage(h->e_uv, e_uv, 2 * (size_t)E);
    const size_t info_off = staged_bytes;
    stage(h->e_info, e_info, 3 * (size_t)E);
    const size_t big_end = h->lg_active ? info_off : staged_bytes;
    lap("reserve");
    SE2_CHECK(h->h_stage.reserve(staged_bytes));
    SE2_CHECK(h->garena.reserve(staged_bytes));
    // The copy into the pinned arena runs ahead of the DMA in pieces of about 1 MB: the engine starts on the first piece
    // while the host is still copying the rest (5.7 MB at 200 key frames: 130 us of memcpy beside the DMA).  The graph's
    // index arrays and estimates go first, on the handle's stream, and the plan kernels are enqueued right behind them; the
    // measurements and information matrices - four fifths of the bytes, needed by nothing before the first linearisation -
    // are copied afterwards on the copy stream, while the device already builds the plan (round 4: the plan chain used to
    // start only when the host had staged everything, 0.11 ms later).
    constexpr size_t kPiece = (size_t)1 << 20;
    size_t sent = 0;   // bytes of the arena already handed to the copy engine
    hipStream_t copy_stream = nullptr;
    SE2_CHECK(ba_copy_stream(h->device, &copy_stream));
    if (!h->ev_copy0) {
        SE2_HIP(hipEventCreateWithFlags(&h->ev_copy0, hipEventDisableTiming));
        SE2_HIP(hipEventCreateWithFlags(&h->ev_copy1, hipEventDisableTiming));
        SE2_HIP(hipEventCreateWithFlags(&h->ev_pat, hipEventDisableTiming));
    }
    SE2_HIP(hipEventRecord(h->ev_copy0, st));                      // (the arena may still be read by earlier work)
    SE2_HIP(hipStreamWaitEvent(copy_stream, h->ev_copy0, 0));
    // From here on DMA out of the pinned arena is in flight on two streams.  The success path waits for all of it at the end;
    // an error return in between must not leave it running (the next load / initialize on the handle writes into the arena
    // without waiting): wait on the way out.
    struct DrainOnError {
        hipStream_t a, b;
        bool armed = true;
        ~DrainOnError() {
            if (armed) { (void)hipStreamSynchronize(b); (void)hipStreamSynchronize(a); }
        }
    } drain{st, copy_stream};
    auto stage_range = [&](size_t lo, size_t hi, hipStream_t cur) -> int {   // the staged arrays with lo <= offset < hi
        auto flush_on = [&](size_t upto) -> int {
            if (upto > sent) {
                SE2_HIP(hipMemcpyAsync(h->garena.p + sent, h->h_stage.p + sent, upto - sent, hipMemcpyHostToDevice, cur));
                sent = upto;
            }
            return SE2GPU_OK;
        };
        for (const Staged& sg : staged) {
            if (sg.off < lo || sg.off >= hi) continue;
            for (size_t done = 0; done < sg.bytes;) {
                const size_t nb = std::min(sg.bytes - done, kPiece);
                std::memcpy(h->h_stage.p + sg.off + done, (const uint8_t*)sg.src + done, nb);
                done += nb;
                if (sg.off + done - sent >= kPiece) SE2_CHECK(flush_on(sg.off + done));
            }
        }
        return flush_on(std::min(hi, staged_bytes));
    };
    for (const Staged& sg : staged) sg.bind(h->garena.p + sg.off);
    SE2_CHECK(stage_range(0, big_off, st));
    lap("indices staged");
    bool pattern_early = false;
    if (h->lg_active && E)
        hipLaunchKernelGGL(k_edge_information, grid1(E, 256), dim3(256), 0, st, E, h->d_lg_lc.p, h->d_lg_lw.p, h->e_kf.p,
                           h->d_lg_sigma2.p, h->d_lg_Rcw.p, h->d_lg_twb.p, h->lg_fx, h->lg_srot, h->lg_sz, h->e_info.p, 1);
    if (device_plan) {
        // ---- the plan, built where it is used
        const int Pb = std::max(1, (int)std::ceil(std::log2((double)std::max(P, 2))));          // bits of a pose index
        const int Qb = std::max(2, (int)std::ceil(std::log2((double)nblk + 2.0)));               // bits of a block key (and of nblk, the tail key)
        const int qlo = (Qb + 1) / 2, qhi = Qb - qlo;                                            // two passes, <= 10 bits each
        SE2_REQUIRE(qlo <= 10 && Pb <= 10, SE2GPU_ERR_CAPACITY, "device plan: %d poses", P);
        const size_t NP = std::max<size_t>(npairs_max, 1);
        const int nblkE = (int)((E + kRadixItems - 1) / kRadixItems), nblkP = (int)((NP + kRadixItems - 1) / kRadixItems);
        SE2_CHECK(h->lm_ptr.reserve((size_t)L + 2));
        SE2_CHECK(h->pose_ptr.reserve((size_t)P + 2));
        SE2_CHECK(h->pose_edges.reserve((size_t)E + 1));
        SE2_CHECK(h->blk_a.reserve(nblk));
        SE2_CHECK(h->blk_b.reserve(nblk));
        SE2_CHECK(h->blk_odo.reserve(nblk));
        SE2_CHECK(h->blk_ptr.reserve((size_t)nblk + 2));
        SE2_CHECK(h->pair_i.reserve(NP));
        SE2_CHECK(h->pair_j.reserve(NP));
        SE2_CHECK(h->plan_np.reserve((size_t)L + 2));
        SE2_CHECK(h->plan_base.reserve((size_t)L + 2));
        SE2_CHECK(h->plan_key0.reserve(std::max<size_t>(NP, (size_t)E + 1)));
        SE2_CHECK(h->plan_key1.reserve(std::max<size_t>(NP, (size_t)E + 1)));
        SE2_CHECK(h->plan_st0.reserve(NP));
        SE2_CHECK(h->plan_st1.reserve(NP));
        SE2_CHECK(h->plan_idx.reserve((size_t)E + 1));
        const size_t nhist = (size_t)1024 * std::max(nblkE, nblkP) + 2;
        SE2_CHECK(h->plan_hist.reserve(nhist));
        SE2_CHECK(h->plan_offs.reserve(nhist));
        SE2_CHECK(h->plan_out.reserve(8));
        SE2_CHECK(h->plan_tsum.reserve(nhist / kScanTile + 4));
        SE2_CHECK(h->plan_toff.reserve(nhist / kScanTile + 4));
        // groups <= pairs / 16 + off-diagonal blocks; two consecutive workgroups hold more than 28 groups together
        const size_t G = npairs_max / kChunk + (size_t)P * (P - 1) / 2 + 1;
        const int cap_wg = (int)std::min<size_t>((size_t)P * (P - 1) / 2 + 1, 2 * G / gpw + 2) + 1;
        SE2_CHECK(h->grp.reserve((size_t)cap_wg * gpw));
        h->grp_cap_wg = cap_wg;
        auto radix = [&](const int* kin, int* kout, auto* vin, auto* vout, size_t cnt, int shift, int bits) -> int {
            using V = std::remove_pointer_t<decltype(vout)>;
            const int nb = (int)((cnt + kRadixItems - 1) / kRadixItems), bins = 1 << bits;
            hipLaunchKernelGGL(k_radix_hist, dim3(std::max(nb, 1)), dim3(kRadixItems), 0, st, kin, (int)cnt, shift, bins, nb,
                               h->plan_hist.p);
            const int cnt_h = bins * nb, ntile = (cnt_h + kScanTile - 1) / kScanTile;
            if (ntile <= 8) {   // up to 16 counters per thread of ONE workgroup: cheaper than three launches
                hipLaunchKernelGGL(k_scan_i32, dim3(1), dim3(1024), 0, st, h->plan_hist.p, h->plan_offs.p, cnt_h);
            } else {
                hipLaunchKernelGGL(k_scan_tiles, dim3(ntile), dim3(256), 0, st, h->plan_hist.p, h->plan_offs.p, cnt_h,
                                   h->plan_tsum.p);
                hipLaunchKernelGGL(k_scan_i32, dim3(1), dim3(1024), 0, st, h->plan_tsum.p, h->plan_toff.p, ntile);
                hipLaunchKernelGGL(k_scan_tile_add, grid1(cnt_h, 256), dim3(256), 0, st, h->plan_offs.p, cnt_h, h->plan_toff.p);
            }
            hipLaunchKernelGGL((k_radix_scatter<V>), dim3(std::max(nb, 1)), dim3(kRadixItems), 0, st, kin, (const V*)vin,
                               (int)cnt, shift, bins, nb, h->plan_offs.p, kout, vout);
            return SE2GPU_OK;
        };
        uint8_t* nzp = nullptr;   // the block pattern for the solver's order, marked on the way (k_plan_odo, k_plan_pairs2)
        if (nd_possible) {
            SE2_CHECK(h->plan_nz.reserve((size_t)nblk));
            SE2_CHECK(h->h_plan_nz.reserve((size_t)nblk));
            nzp = h->plan_nz.p;
        }
        // landmark CSR + pairs per landmark | edge indices | block table | empty group descriptors
        {
            const int g1 = (int)grid1((size_t)L + 1, 256).x, g2 = (int)grid1(E, 256).x, g3 = (int)grid1((size_t)P * P, 256).x;
            const size_t ngrp = (size_t)cap_wg * gpw;
            const int g4 = (int)grid1(ngrp, 256).x;
            hipLaunchKernelGGL(k_plan_init, dim3(g1 + g2 + g3 + g4), dim3(256), 0, st, L, E, P, h->e_lm.p, h->e_kf.p, h->fixed.p,
                               h->lm_ptr.p, h->plan_np.p, h->plan_idx.p, h->blk_a.p, h->blk_b.p, h->blk_odo.p, ngrp, h->grp.p,
                               g1, g1 + g2, g1 + g2 + g3, nzp);
        }
        hipLaunchKernelGGL(k_scan_i32, dim3(1), dim3(1024), 0, st, h->plan_np.p, h->plan_base.p, L);   // 20 per thread
        if (O && !h->odo_fallback)
            hipLaunchKernelGGL(k_plan_odo, grid1(O, 64), dim3(64), 0, st, P, O, h->o_i.p, h->o_j.p, h->blk_odo.p, nzp);
        // pose -> edges CSR: edge indices stably sorted by key frame (one pass)
        if (E) SE2_CHECK(radix(h->e_kf.p, h->plan_key0.p, h->plan_idx.p, h->pose_edges.p, (size_t)E, 0, Pb));
        hipLaunchKernelGGL(k_lower_bounds, grid1((size_t)P + 1, 256), dim3(256), 0, st, h->plan_key0.p, E, P, h->pose_ptr.p);
        // contributor pairs, stably sorted by block key (two passes).  The number of pairs is plan_base[L] - only the device
        // knows it; the sort runs over the host's upper bound with the tail keyed past every block, where it stays (stable)
        // and is never referenced
        {
            const int g1 = (int)grid1(L, 256).x, g2 = (int)grid1(NP, 256).x;
            hipLaunchKernelGGL(k_plan_pairs2, dim3(g1 + g2), dim3(256), 0, st, L, P, h->lm_ptr.p, h->e_kf.p, h->fixed.p,
                               h->plan_base.p, h->plan_key0.p, h->plan_st0.p, (int)NP, nblk, g1, nzp);
        }
        if (nd_possible) {   // the pattern goes home now: the host chooses the solver's order while the device sorts the pairs
            if (nd_sharded) {   // the union of the ranks' patterns: widen to doubles, the run's all-reduce, narrow again
                SE2_CHECK(h->plan_nzd.reserve((size_t)nblk));
                hipLaunchKernelGGL(k_pattern_widen, grid1((size_t)nblk, 256), dim3(256), 0, st, nblk, h->plan_nz.p, h->plan_nzd.p, 0);
                SE2_HIP(hipGetLastError());
                SE2_CHECK(ba_allreduce(h, h->plan_nzd.p, (size_t)nblk));
                hipLaunchKernelGGL(k_pattern_widen, grid1((size_t)nblk, 256), dim3(256), 0, st, nblk, h->plan_nz.p, h->plan_nzd.p, 1);
            }
            SE2_HIP(hipMemcpyAsync(h->h_plan_nz.p, h->plan_nz.p, (size_t)nblk, hipMemcpyDeviceToHost, st));
            SE2_HIP(hipEventRecord(h->ev_pat, st));
            pattern_early = true;
        }
        SE2_CHECK(radix(h->plan_key0.p, h->plan_key1.p, h->plan_st0.p, h->plan_st1.p, NP, 0, qlo));
        SE2_CHECK(radix(h->plan_key1.p, h->plan_key0.p, h->plan_st1.p, h->plan_st0.p, NP, qlo, std::max(qhi, 1)));
        {
            const int g1 = (int)grid1((size_t)nblk + 1, 256).x, g2 = (int)grid1(NP, 256).x;
            hipLaunchKernelGGL(k_plan_bounds, dim3(g1 + g2), dim3(256), 0, st, h->plan_key0.p, (int)NP, nblk, h->blk_ptr.p,
                               h->plan_st0.p, h->pair_i.p, h->pair_j.p, g1);
        }
        if (nblk <= kPackMaxBlk) {
            SE2_CHECK(h->plan_place.reserve((size_t)kPackRuns * (kGrpPerWG + 1)));   // the runs' tables, then their start states
            int* tab = h->plan_place.p;
            int* start = tab + kPackRuns * kGrpPerWG;
            hipLaunchKernelGGL(k_plan_tab, dim3(kPackRuns), dim3(128), 0, st, nblk, h->blk_a.p, h->blk_b.p, h->blk_ptr.p, tab, gpw);
            hipLaunchKernelGGL(k_plan_tabscan, dim3(1), dim3(1024), 0, st, tab, start, h->plan_out.p);
            hipLaunchKernelGGL(k_plan_place, dim3(kPackRuns), dim3(128), 0, st, nblk, h->blk_a.p, h->blk_b.p, h->blk_ptr.p, start,
                               h->grp.p, cap_wg, gpw);
        } else {
            hipLaunchKernelGGL(k_plan_pack, dim3(1), dim3(256), 0, st, nblk, h->blk_a.p, h->blk_b.p, h->blk_ptr.p, h->grp.p,
                               cap_wg, h->plan_out.p, gpw);
        }
        SE2_HIP(hipGetLastError());
        SE2_CHECK(h->h_scal.reserve(8 + (size_t)h->world));
        SE2_HIP(hipMemcpyAsync(h->h_scal.p, h->plan_out.p, sizeof(int), hipMemcpyDeviceToHost, st));
        lap("plan kernels enqueued");
    }
    // measurements and information matrices (with a local graph the information is evaluated on the device: big_end stops in
    // front of it); the handle's stream waits for them before anything later runs
    sent = big_off;
    SE2_CHECK(stage_range(big_off, big_end, copy_stream));
    SE2_HIP(hipEventRecord(h->ev_copy1, copy_stream));
    lap("measurements staged");
    SE2_HIP(hipStreamWaitEvent(st, h->ev_copy1, 0));
    SE2_CHECK(h->fin_counter.reserve(1));
    SE2_HIP(hipMemsetAsync(h->fin_counter.p, 0, sizeof(unsigned), st));
    SE2_CHECK(h->l0_acc.reserve(2));
    SE2_HIP(hipMemsetAsync(h->l0_acc.p, 0, 2 * sizeof(unsigned long long), st));
    // the handle's device-side counters start over with the graph
    SE2_HIP(hipMemsetAsync(h->ctl.p, 0, sizeof(BaCtl), st));
    h->dev_seq = 0;
    h->drop_graphs();
    SE2_CHECK(h->h_x.reserve(n));
    SE2_CHECK(h->h_scal.reserve(8 + (size_t)h->world));
    if (!h->h_mail) {
        SE2_HIP(hipHostMalloc((void**)&h->h_mail, kMailDoubles * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(h->h_mail, 0, kMailDoubles * sizeof(double));
        SE2_HIP(hipHostGetDevicePointer((void**)&h->d_mail, h->h_mail, 0));
        h->h_stop = reinterpret_cast<volatile int*>(h->h_mail + kMailStop);
        h->d_stop = reinterpret_cast<int*>(h->d_mail + kMailStop);
    }
    ((volatile double*)h->h_mail)[kMailSeq] = 0.0;   // the slot counter starts over with the graph (BaCtl::seq was cleared above)
    h->poses = h->poses_a.p; h->poses_t = h->poses_b.p;
    h->lms = h->lms_a.p; h->lms_t = h->lms_b.p;
    SE2_HIP(hipMemcpyAsync(h->poses, h->poses0.p, (size_t)ps * P * 8, hipMemcpyDeviceToDevice, st));
    if (L) SE2_HIP(hipMemcpyAsync(h->lms, h->lms0.p, 3 * (size_t)L * 8, hipMemcpyDeviceToDevice, st));
    // the host waits for the block pattern only (it left the device in front of the pair sorts) and plans the solve while the
    // device finishes the plan and the copy engine the measurements; everything is waited for at the end
    if (pattern_early) SE2_HIP(hipEventSynchronize(h->ev_pat));
    else SE2_HIP(hipStreamSynchronize(st));
    lap(pattern_early ? "pattern" : "synchronise");
    // ---- the dense pose solve: order of the poses, tile tasks, dependency lists (solve_plan_choose), then its buffers
    {
        std::vector<uint8_t> pat;
        if (nd_possible) {
            const uint8_t* nz = device_plan ? h->h_plan_nz.p : pl.blk_nz.data();
            pat.assign((size_t)P * P, 0);
            for (int a = 0; a < P; ++a) {
                pat[(size_t)a * P + a] = 1;
                for (int b = a + 1; b < P; ++b)
                    if (nz[blk_index_of(P, a, b)]) pat[(size_t)a * P + b] = pat[(size_t)b * P + a] = 1;
            }
        }
        SolvePlan sp;
        solve_plan_choose(P, D, pat.empty() ? nullptr : pat.data(), !pat.empty(), sp, true);
        h->nsys = sp.nsys;
        h->solve_depth = sp.depth;
        h->chol_ntask = (int)sp.tasks.size();
        {   // tasks | dependency lists | pose_off | col_src: one copy, through a pinned buffer of their own
            auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
            const size_t b0 = up(sp.tasks.size() * sizeof(int4)), b1 = up(std::max<size_t>(sp.deps.size(), 1) * 4);
            const size_t b2 = sp.permuted ? up(sp.pose_off.size() * 4) : 0, b3 = sp.permuted ? up(sp.col_src.size() * 4) : 0;
            // (at least 1 MB each: the lists of the next window differ in length, and growing a pinned or a device buffer
            // costs more than a whole initialize - seen as +0.25 ms on the first window after se2gpu_ba_reserve)
            SE2_CHECK(h->h_stage_solver.reserve(std::max<size_t>(b0 + b1 + b2 + b3, (size_t)1 << 20)));
            SE2_CHECK(h->solver_arena.reserve(std::max<size_t>(b0 + b1 + b2 + b3, (size_t)1 << 20)));
            uint8_t* hp = h->h_stage_solver.p;
            std::memcpy(hp, sp.tasks.data(), sp.tasks.size() * sizeof(int4));
            std::memcpy(hp + b0, sp.deps.data(), sp.deps.size() * 4);
            if (sp.permuted) {
                std::memcpy(hp + b0 + b1, sp.pose_off.data(), sp.pose_off.size() * 4);
                std::memcpy(hp + b0 + b1 + b2, sp.col_src.data(), sp.col_src.size() * 4);
            }
            SE2_HIP(hipMemcpyAsync(h->solver_arena.p, hp, b0 + b1 + b2 + b3, hipMemcpyHostToDevice, st));
            h->chol_tasks.alias(reinterpret_cast<int4*>(h->solver_arena.p));
            h->chol_deps.alias(reinterpret_cast<int*>(h->solver_arena.p + b0));
            if (sp.permuted) {
                h->pose_off.alias(reinterpret_cast<int*>(h->solver_arena.p + b0 + b1));
                h->col_src.alias(reinterpret_cast<int*>(h->solver_arena.p + b0 + b1 + b2));
            }
        }
        if (sp.permuted) {
            h->h_pose_off = sp.pose_off;
            h->ld = ((sp.nsys + 1 + kNB - 1) / kNB) * kNB;
            const size_t nred2 = (size_t)h->ld * h->ld + 4;
            SE2_CHECK(h->red_own.reserve(nred2));
            h->red = h->red_own.p;
            SE2_CHECK(h->Rinv.reserve(chol_work_doubles(h->ld)));
            SE2_CHECK(h->xp.reserve((size_t)std::max(n, sp.nsys)));
            SE2_HIP(hipMemsetAsync(h->red, 0, nred2 * sizeof(double), st));
            hipLaunchKernelGGL(k_solver_pads, grid1((size_t)sp.nsys, 256), dim3(256), 0, st, h->red, h->ld, sp.nsys, h->col_src.p);
            SE2_HIP(hipGetLastError());
        } else {
            h->h_pose_off.clear();
            h->pose_off.release();
            h->col_src.release();
        }
        const int nt2 = h->ld / kNB, nbc2 = (h->nsys + kNB - 1) / kNB;
        SE2_CHECK(h->chol_flags.reserve(2 * kSlabs * (size_t)nt2 * nbc2));
        SE2_HIP(hipMemsetAsync(h->chol_flags.p, 0, 2 * kSlabs * (size_t)nt2 * nbc2 * sizeof(unsigned), st));   // flags = 0 = "no epoch yet"
        SE2_CHECK(h->chol_head.reserve(1));
        SE2_HIP(hipMemsetAsync(h->chol_head.p, 0, sizeof(unsigned long long), st));
        static const bool verify = [] { const char* e = getenv("SE2GPU_BA_CHOL_VERIFY"); return e && e[0] == '1'; }();
        if (verify) {
            const size_t words = kVfyChk + 2 * 2 * kSlabs * (size_t)nt2 * nbc2;
            SE2_CHECK(h->chol_vfy.reserve(words));
            SE2_HIP(hipMemsetAsync(h->chol_vfy.p, 0, words * sizeof(unsigned long long), st));
        } else {
            h->chol_vfy.release();
        }
        const char* env = getenv("SE2GPU_BA_CHOL");
        h->chol_steps = (env && std::strcmp(env, "steps") == 0) || nt2 > 64 || h->chol_fallback;
        const char* tr = getenv("SE2GPU_BA_CHOL_TRACE");
        if (tr && tr[0] == '1') {
            SE2_CHECK(h->chol_trace.reserve(16 * (size_t)h->chol_ntask));
            SE2_HIP(hipMemsetAsync(h->chol_trace.p, 0, 16 * (size_t)h->chol_ntask * sizeof(long long), st));
        }
        lap("solve plan");
    }
    SE2_HIP(hipStreamSynchronize(st));
    if (device_plan) {
        int nwg = 0;
        std::memcpy(&nwg, h->h_scal.p, sizeof(int));
        SE2_REQUIRE(nwg <= h->grp_cap_wg, SE2GPU_ERR_CAPACITY, "device plan: %d workgroups exceed the bound %d", nwg, h->grp_cap_wg);
        h->nwg_off = std::max(nwg, 1);
    }
    lap("synchronise");
    // the borrow of se2gpu_ba_load ends here: everything has been copied into the pinned arena and uploaded
    h->bulk_E = 0;
    h->bulk_kf = h->bulk_lm = nullptr;
    h->bulk_uv = h->bulk_info = nullptr;
    h->initialized = true;
    h->est_valid = false;
    static std::atomic<unsigned long> serial{0};
    h->init_serial = ++serial;
    drain.armed = false;   // (everything was waited for above)
    return SE2GPU_OK;
}


// Every launch below exists in two forms: with the device-side controller (`c` = h->ctl.p: the kernel picks the
// estimate buffer, lambda and whether it has anything to do from the block; "a"/"b" buffers are passed in fixed order)
// and without (c = nullptr: explicit lambda, h->poses / h->lms are the estimate) for the synchronous entry points
// (se2gpu_ba_chi2, the debug_* introspection).
// work enqueued for this handle on a batch stream (se2gpu_ba_reset_estimates_batch) comes first
inline int ba_join(se2gpu_ba* h) {
    if (h->join_event) {
        if (h->join_stream != h->stream) SE2_HIP(hipStreamWaitEvent(h->stream, h->join_event, 0));
        h->join_event = nullptr;
        h->join_stream = nullptr;
    }
    return SE2GPU_OK;
}
struct ResetItem {
    double* dst_p; const double* src_p; unsigned np;
    double* dst_l; const double* src_l; unsigned nl;
};
__global__ __launch_bounds__(256) void k_reset_one(double* __restrict__ poses, const double* __restrict__ poses0, size_t np,
                                                    double* __restrict__ lms, const double* __restrict__ lms0, size_t nl) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < np) poses[i] = poses0[i];
    if (i < nl) lms[i] = lms0[i];
}
__global__ __launch_bounds__(256) void k_reset_batch(const ResetItem* __restrict__ items) {
    const ResetItem it = items[blockIdx.y];
    const unsigned stride = gridDim.x * 256, t0 = blockIdx.x * 256 + threadIdx.x;
    for (unsigned i = t0; i < it.np; i += stride) it.dst_p[i] = it.src_p[i];
    for (unsigned i = t0; i < it.nl; i += stride) it.dst_l[i] = it.src_l[i];
}

struct Bufs {
    const BaCtl* c;
    double *pa, *pb, *la, *lb;   // estimate / trial when c == nullptr, "a" / "b" otherwise
};
inline Bufs bufs(se2gpu_ba* h, bool ctl) {
    if (ctl) return Bufs{h->ctl.p, h->poses_a.p, h->poses_b.p, h->lms_a.p, h->lms_b.p};
    return Bufs{nullptr, h->poses, h->poses_t, h->lms, h->lms_t};
}

// SE(2) model (whitened records, k_linearize): h->Hpl holds W_e, h->Dinv the inverse Cholesky factors A_l, h->z the zeta_l;
// h->Y is not used.  The SE3-expmap model keeps Hpl / Y / Dinv / z as named.

// linearise at the current state: Hpl, Hpp_e, bp_e, Hll, bl; with fuse_lambda >= 0 (or fused && ctl) also Dinv, z, Y
int ba_linearize(se2gpu_ba* h, double fuse_lambda, bool ctl = false) {
    hipStream_t st = h->stream;
    const Bufs B = bufs(h, ctl);
    if (h->model == 2) {   // pose graph: nothing but the prior and edge blocks
        SE2_LAUNCH(h->prof, st, "k4_terms", k4_terms, grid1((size_t)h->P + h->O, 64), dim3(64), 0, h->P, h->O, B.pa, h->fixed.p,
                   h->prior_has.p, h->prior_meas.p, h->prior_info.p, h->ph.p, h->pb.p, h->slot_a.p, h->slot_ptr.p, h->pe_i.p,
                   h->pe_j.p, h->pe_meas.p, h->pe_info.p, h->Oii.p, h->Ojj.p, h->Oij.p, h->obi.p, h->obj.p, B.c, B.pb);
        SE2_HIP(hipGetLastError());
        return SE2GPU_OK;
    }
    if (h->model) {
        if (fuse_lambda >= 0.0)
            SE2_LAUNCH(h->prof, st, "k3_linearize", (k3_linearize<true>), grid1((size_t)h->L * kGroup, kBlock), dim3(kBlock), 0,
                       h->cam3, h->L, h->lm_ptr.p, h->e_kf.p, h->e_uv.p, h->e_info.p, B.pa, h->fixed.p, B.la, h->Hpl.p,
                       h->Hpp_e.p, h->bp_e.p, h->Hll.p, h->bl.p, fuse_lambda, h->Dinv.p, h->z.p, h->Y.p, h->Dg.p, B.c, B.pb, B.lb);
        else
            SE2_LAUNCH(h->prof, st, "k3_linearize", (k3_linearize<false>), grid1((size_t)h->L * kGroup, kBlock), dim3(kBlock), 0,
                       h->cam3, h->L, h->lm_ptr.p, h->e_kf.p, h->e_uv.p, h->e_info.p, B.pa, h->fixed.p, B.la, h->Hpl.p,
                       h->Hpp_e.p, h->bp_e.p, h->Hll.p, h->bl.p, 0.0, h->Dinv.p, h->z.p, h->Y.p, h->Dg.p, B.c, B.pb, B.lb);
        // the prior gradients and the odometry blocks of this linearisation
        SE2_LAUNCH(h->prof, st, "k3_terms", k3_terms, grid1((size_t)h->P + h->O, 64), dim3(64), 0, h->P, h->O, B.pa, h->fixed.p,
                   h->prior_has.p, h->prior_meas.p, h->prior_info.p, h->pb.p, h->o_i.p, h->o_j.p, h->o_meas.p, h->o_info.p,
                   h->Oii.p, h->Ojj.p, h->Oij.p, h->obi.p, h->obj.p, B.c, B.pb);
        SE2_HIP(hipGetLastError());
        return SE2GPU_OK;
    }
    if (fuse_lambda >= 0.0)
        SE2_LAUNCH(h->prof, st, "k_linearize", (k_linearize<true>), grid1((size_t)h->L * kGroup, kBlock), dim3(kBlock), 0,
                   h->cam, h->L, h->lm_ptr.p, h->e_kf.p, h->e_uv.p, h->e_info.p, B.pa, h->fixed.p, B.la, h->Hpl.p,
                   h->Hpp_e.p, h->bp_e.p, h->Hll.p, h->bl.p, fuse_lambda, h->Dinv.p, h->z.p, h->Dg.p, B.c, B.pb, B.lb);
    else
        SE2_LAUNCH(h->prof, st, "k_linearize0", (k_linearize<false>), grid1((size_t)h->L * kGroup, kBlock), dim3(kBlock), 0,
                   h->cam, h->L, h->lm_ptr.p, h->e_kf.p, h->e_uv.p, h->e_info.p, B.pa, h->fixed.p, B.la, h->Hpl.p,
                   h->Hpp_e.p, h->bp_e.p, h->Hll.p, h->bl.p, 0.0, h->Dinv.p, h->z.p, h->Dg.p, B.c, B.pb, B.lb);
    SE2_HIP(hipGetLastError());
    return SE2GPU_OK;
}

// the estimate's pose buffer as a host-known pointer (only valid between optimize() calls / in synchronous mode)
// un-reduced pose blocks Hpp / bp (only needed for lambda_0 = 1e-5 max diag H and by the odometry fallback)
int ba_pose_blocks(se2gpu_ba* h, const double* poses, bool ctl = false, bool odometry_only = false) {
    hipStream_t st = h->stream;
    if (h->O) {
        // with the device-side controller the estimate's buffer is the controller's to name (a captured graph must not
        // carry the host's idea of it)
        if (ctl)
            SE2_LAUNCH(h->prof, st, "k_odometry", k_odometry, grid1(h->O, 64), dim3(64), 0, h->O, h->o_i.p, h->o_j.p,
                       h->o_meas.p, h->o_info.p, h->poses_a.p, h->fixed.p, h->Oii.p, h->Ojj.p, h->Oij.p, h->obi.p, h->obj.p,
                       (const BaCtl*)h->ctl.p, (const double*)h->poses_b.p);
        else
            SE2_LAUNCH(h->prof, st, "k_odometry", k_odometry, grid1(h->O, 64), dim3(64), 0, h->O, h->o_i.p, h->o_j.p,
                       h->o_meas.p, h->o_info.p, poses, h->fixed.p, h->Oii.p, h->Ojj.p, h->Oij.p, h->obi.p, h->obj.p,
                       (const BaCtl*)nullptr, (const double*)nullptr);
    }
    if (!odometry_only)   // (k_lambda0 does the pose blocks itself)
        SE2_LAUNCH(h->prof, st, "k_pose_reduce", k_pose_reduce, grid1((size_t)h->P * 64, kBlock), dim3(kBlock), 0, h->P,
                   h->pose_ptr.p, h->pose_edges.p, h->Hpp_e.p, h->bp_e.p, h->podo_ptr.p, h->podo_item.p, h->Oii.p,
                   h->Ojj.p, h->obi.p, h->obj.p, h->Hpp.p, h->bp.p);
    SE2_HIP(hipGetLastError());
    return SE2GPU_OK;
}

// reduced system for damping lambda into h->red (local contribution of this rank).
// schur: 0 = the lambda-dependent records are current, 1 = recompute them, 2 = let the controller decide (retry; SE3 model), 3 = force
int ba_reduce(se2gpu_ba* h, double lambda, int schur, bool ctl = false) {
    hipStream_t st = h->stream;
    double* S = h->red;
    const Bufs B = bufs(h, ctl);
    if (h->model) {
        const double* pinfo = h->model == 2 ? h->ph.p : h->prior_info.p;   // Hessian of the prior: J' Omega J (pose graph) / Omega
        if (schur && h->model == 1)
            SE2_LAUNCH(h->prof, st, "k3_schur_lm", k3_schur_lm, grid1((size_t)h->L * kGroup, kBlock), dim3(kBlock), 0, h->L,
                       lambda, h->lm_ptr.p, h->Hll.p, h->bl.p, h->Hpl.p, h->Hpp_e.p, h->bp_e.p, h->Dinv.p, h->z.p, h->Y.p,
                       h->Dg.p, B.c, schur == 2 ? 0 : 1);
        SE2_LAUNCH(h->prof, st, "k3_reduce2", k3_reduce2, dim3(((h->P + 1 + 7) & ~7) + ((h->nwg_off + 7) & ~7)), dim3(kBlock), 0,
                   h->P, h->ld, h->nwg_off, lambda, h->grp.p, h->blk_a.p, h->blk_b.p, h->pair_i.p, h->pair_j.p, h->blk_odo.p,
                   h->Y.p, h->Hpl.p, h->Dg.p, h->fixed.p, h->pose_ptr.p, h->pose_edges.p, h->podo_ptr.p, h->podo_item.p,
                   h->prior_has.p, pinfo, h->pb.p, h->Oii.p, h->Ojj.p, h->Oij.p, h->obi.p, h->obj.p, S, h->bp.p, B.c,
                   &h->ctl.p->epoch);
        SE2_HIP(hipGetLastError());
        return SE2GPU_OK;
    }
    // the per-edge records depend on lambda: a new damping on the same linearisation (first trial after lambda_0, a retry
    // the host knows about) runs the fused linearisation again - same estimate, same code, identical blocks
    if (schur) SE2_CHECK(ba_linearize(h, lambda, ctl));
    SE2_LAUNCH(h->prof, st, "k_reduce2", k_reduce2, dim3(((h->P + 1 + 7) & ~7) + ((h->nwg_off + 7) & ~7)), dim3(kBlock), 0, h->P, h->ld, h->nwg_off,
               lambda, h->root, h->grp.p, h->blk_a.p, h->blk_b.p, h->pair_i.p, h->pair_j.p, h->blk_odo.p,
               h->Hpl.p, h->Dg.p, h->fixed.p, h->pose_ptr.p, h->pose_edges.p,
               h->podo_ptr.p, h->podo_item.p, h->o_i.p, h->o_j.p, h->o_meas.p, h->o_info.p, B.pa, S, h->bp.p, B.c, B.pb,
               &h->ctl.p->epoch, (const int*)h->pose_off.p, h->nsys);
    if (h->O && h->odo_fallback) {
        // PreEdgeSE2 edges the plan cannot carry (self loops, duplicates): blocks from the estimate, added atomically.  Only
        // reachable with a host-known estimate pointer, so such graphs run in synchronous mode (ba_needs_sync).
        SE2_LAUNCH(h->prof, st, "k_odometry", k_odometry, grid1(h->O, 64), dim3(64), 0, h->O, h->o_i.p, h->o_j.p,
                   h->o_meas.p, h->o_info.p, h->poses, h->fixed.p, h->Oii.p, h->Ojj.p, h->Oij.p, h->obi.p, h->obj.p,
                   (const BaCtl*)nullptr, (const double*)nullptr);
        SE2_LAUNCH(h->prof, st, "k_reduce_odo", k_reduce_odo, grid1((size_t)h->O * 9, 256), dim3(256), 0, h->O, h->ld,
                   h->o_i.p, h->o_j.p, h->Oij.p, S);
    }
    SE2_HIP(hipGetLastError());
    return SE2GPU_OK;
}

int ba_allreduce(se2gpu_ba* h, double* ptr, size_t count) {
    if (!h->allreduce) return SE2GPU_OK;
    // in the per-kernel pass (se2gpu_ba_profile) the exchange is timed like a launch: "allreduce_system" is the reduced
    // system [S | b | scalars] of a trial, "allreduce_small" the scalar / diagonal exchanges (SURVEY.md section 8e asks for the
    // all-reduce time of the 1/2/4/8 curve to be broken out)
    h->prof.begin(h->stream);
    const int rc = h->allreduce(ptr, count, (void*)h->stream, h->ar_user);
    h->prof.end(h->stream, count > 4096 ? "allreduce_system" : "allreduce_small");
    SE2_REQUIRE(rc == 0, SE2GPU_ERR_HIP, "all-reduce callback failed with %d", rc);
    return SE2GPU_OK;
}

// The exchange of the sharded run ships only what the dense solver reads: of row r of [S; b^T] the columns up to the end of
// its diagonal tile (the lower-triangular tiles; the rhs row n in full) - 52 % of the rectangle at 200 key frames.  Rows are
// packed back to back: offset(r) = 32 * (32 q (q + 1) / 2 + (r mod 32) (q + 1)),  q = r / 32.
__host__ __device__ inline size_t tri_row_off(int r) {
    const size_t q = (size_t)(r / kNB), rem = (size_t)(r % kNB);
    return kNB * (kNB * q * (q + 1) / 2 + rem * (q + 1));
}
__global__ __launch_bounds__(256) void k_tri_pack(const double* __restrict__ A, int ld, int rows, double* __restrict__ packed,
                                                   int unpack, double* __restrict__ Aout) {
    const int r = blockIdx.x;
    if (r >= rows) return;
    const int len = min(kNB * (r / kNB + 1), ld);
    const size_t off = tri_row_off(r);
    if (!unpack) {
        for (int c = threadIdx.x; c < len; c += 256) packed[off + c] = A[(size_t)r * ld + c];
    } else {
        for (int c = threadIdx.x; c < len; c += 256) Aout[(size_t)r * ld + c] = packed[off + c];
    }
}
int ba_allreduce_system(se2gpu_ba* h) {
    if (!h->allreduce) return SE2GPU_OK;
    const int n = h->nsys, rows = n + 1;   // (nsys > D * P when the poses were re-ordered into padded partitions)
    // a caller-owned exchange buffer (se2gpu_ba_set_allreduce with `buffer`) can only be reduced where it is, and the host
    // solve reads whole rows: the rectangle then
    if (h->ar_buffer || h->host_solve) return ba_allreduce(h, h->red, (size_t)rows * h->ld);
    const size_t count = tri_row_off(rows);   // (slightly above the exact size when the last row's tile is cut by ld)
    SE2_CHECK(h->red_packed.reserve(count));
    SE2_LAUNCH(h->prof, h->stream, "k_tri_pack", k_tri_pack, dim3(rows), dim3(256), 0, h->red, h->ld, rows, h->red_packed.p, 0,
               (double*)nullptr);
    SE2_CHECK(ba_allreduce(h, h->red_packed.p, count));
    SE2_LAUNCH(h->prof, h->stream, "k_tri_pack", k_tri_pack, dim3(rows), dim3(256), 0, (const double*)nullptr, h->ld, rows,
               h->red_packed.p, 1, h->red);
    SE2_HIP(hipGetLastError());
    return SE2GPU_OK;
}

// host side of the mailbox: poll the mapped, coherent buffer until the device has written sequence number `seq`;
// meanwhile the caller's force-stop flag is mirrored into the word the device reads
int ba_wait_mail(se2gpu_ba* h, double seq, const volatile uint8_t* stop_flag = nullptr) {
    volatile double* mb = h->h_mail;
    const auto t0 = std::chrono::steady_clock::now();
    long spins = 0;
    while (mb[3] != seq) {
        __builtin_ia32_pause();
        if (stop_flag && *stop_flag) *h->h_stop = 1;
        if ((++spins & 0xfffff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) {
            SE2_HIP(hipStreamSynchronize(h->stream));  // surfaces a device fault, if that is what happened
            SE2_REQUIRE(mb[3] == seq, SE2GPU_ERR_HIP, "the host mailbox was not written");
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    h->h_scal.p[0] = mb[0]; h->h_scal.p[1] = mb[1]; h->h_scal.p[2] = mb[2];
    return SE2GPU_OK;
}

// multi-GPU: the all-reduced scalars go to the host mailbox too (instead of a stream synchronise + D2H copy per trial)
__global__ void k_post_mail(const double* __restrict__ scal, volatile double* __restrict__ mail, double seq) {
    mail[0] = scal[0];
    mail[1] = scal[1];
    mail[2] = scal[2];
    __threadfence_system();
    mail[3] = seq;
}

// chi2 / scale of a (trial) step, synchronous form (se2gpu_ba_chi2): xp == nullptr evaluates the current state.
// Result in h->h_scal[0..1].
int ba_evaluate(se2gpu_ba* h, const double* xp, double lambda) {
    SE2_CHECK(ba_join(h));
    hipStream_t st = h->stream;
    double* scal = h->red + (size_t)h->ld * h->ld;  // 4 trailing scalars of the fused buffer
    const bool use_mail = h->d_mail && !(h->allreduce && h->world > 1) && !h->comm;
    const double seq = (double)(++h->mail_seq);
    const dim3 ug = grid1((size_t)h->L * kGroup, kBlock);
    if (use_mail) {   // single GPU: the trial ends inside k_update (FinArgs)
        FinArgs fin{1, (int)ug.x, h->P, h->O, h->root, xp ? 1 : 0, 0, 0, h->fixed.p, xp, h->bp.p, h->poses, h->poses_t,
                    h->o_i.p, h->o_j.p, h->o_meas.p, h->o_info.p, scal, h->d_mail, seq, (BaCtl*)nullptr,
                    (const volatile int*)nullptr, h->fin_counter.p};
        SE2_LAUNCH(h->prof, st, "k_update", k_update, dim3(ug.x + 1), dim3(kBlock), 0, h->cam, h->L, lambda, h->lm_ptr.p,
                   h->e_kf.p, h->e_uv.p, h->e_info.p, h->poses, h->fixed.p, h->lms, xp, h->z.p, (const double*)h->Hpl.p, h->bl.p, h->lms_t,
                   h->part.p, (const BaCtl*)nullptr, (const double*)nullptr, fin, (const double*)h->Dinv.p);
        SE2_HIP(hipGetLastError());
        return ba_wait_mail(h, seq);
    }
    SE2_LAUNCH(h->prof, st, "k_update", k_update, ug, dim3(kBlock), 0, h->cam, h->L,
               lambda, h->lm_ptr.p, h->e_kf.p, h->e_uv.p, h->e_info.p, h->poses, h->fixed.p, h->lms, xp, h->z.p,
               (const double*)h->Hpl.p, h->bl.p, h->lms_t, h->part.p, (const BaCtl*)nullptr, (const double*)nullptr, FinArgs{}, (const double*)h->Dinv.p);
    SE2_LAUNCH(h->prof, st, "k_finalize", k_finalize, dim3(1), dim3(1024), 0, h->L ? h->nparts : 0, h->part.p, h->P,
               lambda, h->poses, h->fixed.p, xp, h->bp.p, h->poses_t, h->O, h->o_i.p, h->o_j.p, h->o_meas.p,
               h->o_info.p, h->root, scal, (volatile double*)nullptr, seq, (BaCtl*)nullptr, xp ? 1 : 0, 0, 0,
               (const volatile int*)nullptr);
    SE2_HIP(hipGetLastError());
    SE2_CHECK(ba_allreduce(h, scal, 4));
    if (h->d_mail) {
        hipLaunchKernelGGL(k_post_mail, dim3(1), dim3(1), 0, st, scal, h->d_mail, seq);
        SE2_HIP(hipGetLastError());
        SE2_CHECK(ba_wait_mail(h, seq));
        return SE2GPU_OK;
    }
    SE2_HIP(hipMemcpyAsync(h->h_scal.p, scal, 3 * sizeof(double), hipMemcpyDeviceToHost, st));
    SE2_HIP(hipStreamSynchronize(st));
    return SE2GPU_OK;
}

// dense pose solve: augmented S|bs (device, already all-reduced) -> xp (device).
// `fail` = scalar slot [2] of the fused buffer: set to 1 by the factorisation on a non-positive pivot.
int ba_solve(se2gpu_ba* h, bool ctl = false) {
    hipStream_t st = h->stream;
    const int n = h->nsys;   // D * P, or the padded order of the permuted system (solve_plan_choose)
    const int ld = h->ld;
    double* A = h->red;
    double* fail = h->red + (size_t)ld * ld + 2;
    const BaCtl* c = ctl ? h->ctl.p : nullptr;
    if (h->host_solve) {
        SE2_HIP(hipMemcpyAsync(h->h_red.p, A, (size_t)(n + 1) * ld * sizeof(double), hipMemcpyDeviceToHost, st));
        SE2_HIP(hipStreamSynchronize(st));
        std::vector<double> M((size_t)n * n);
        for (int r = 0; r < n; ++r) std::memcpy(&M[(size_t)r * n], h->h_red.p + (size_t)r * ld, n * sizeof(double));
        std::memcpy(h->h_x.p, h->h_red.p + (size_t)n * ld, n * sizeof(double));
        const bool ok = host_cholesky_solve(M.data(), n, h->h_x.p);
        if (!ok) std::memset(h->h_x.p, 0, n * sizeof(double));
        const double f = ok ? 0.0 : 1.0;
        SE2_HIP(hipMemcpyAsync(h->xp.p, h->h_x.p, n * sizeof(double), hipMemcpyHostToDevice, st));
        SE2_HIP(hipMemcpyAsync(fail, &f, sizeof(double), hipMemcpyHostToDevice, st));
        SE2_HIP(hipStreamSynchronize(st));
        return SE2GPU_OK;
    }
    const int nt = ld / kNB;                 // tile rows of A (incl. the rhs / padding tile row)
    const int nbc = (n + kNB - 1) / kNB;     // block columns to factor
    double* Rm = h->Rinv.p;
    // A grid that fits the device at once - with a factor two to spare - needs no task counter: all of its workgroups get a slot
    // in any order of dispatch (waiting ones hold at most half the slots; whatever else runs on the device ends by itself).
    static const int capacity = [] {
        int per_cu = 0, dev = 0;
        hipDeviceProp_t pr{};
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&k_chol_tiles<false>), 256, 0) != hipSuccess) return 0;
        return per_cu * pr.multiProcessorCount;
    }();
    unsigned long long* const head = (2 * h->chol_ntask <= capacity) ? nullptr : h->chol_head.p;
    if (h->chol_steps) {
        for (int k = 0; k < nbc; ++k)  // update with panel k-1 fused with the elimination of panel k
            SE2_LAUNCH(h->prof, st, "k_chol_step", k_chol_step, dim3(nbc - k, nt), dim3(256), 0, A, Rm, ld, n, nt, k, fail, c);
        SE2_LAUNCH(h->prof, st, "k_chol_apply", k_chol_apply, dim3((n + 3) / 4), dim3(256), 0, A, Rm, ld, n, h->xp.p, c,
                   (const int*)h->col_src.p);
    } else {
        double* YU = Rm + 2 * (size_t)nt * nbc * kSlabs * kSlabDoubles;   // behind the publish buffer
        unsigned* flagA = h->chol_flags.p;
        unsigned* flagR = flagA + kSlabs * (size_t)nt * nbc;
        // SE2GPU_BA_CHOL_FAULT=1 (tests): the first dataflow solve of a handle runs without its first task, so that every
        // other task times out - exercises the fallback to k_chol_step in ba_run_step
        static const bool fault = [] { const char* e = getenv("SE2GPU_BA_CHOL_FAULT"); return e && e[0] == '1'; }();
        const int skip = (fault && !h->chol_faulted && h->chol_ntask > 1) ? 1 : 0;
        h->chol_faulted = true;
        if (h->chol_vfy.p)
            SE2_LAUNCH(h->prof, st, "k_chol_tiles", k_chol_tiles<true>, dim3(h->chol_ntask - skip), dim3(256), 0, A, Rm, YU, ld, n, nbc,
                       h->chol_tasks.p + skip, (const int*)h->chol_deps.p, (const int*)h->col_src.p, flagA, flagR, &h->ctl.p->epoch,
                       fail, h->chol_trace.p, c, h->xp.p, h->chol_vfy.p, head, h->chol_ntask - skip);
        else
            SE2_LAUNCH(h->prof, st, "k_chol_tiles", k_chol_tiles<false>, dim3(h->chol_ntask - skip), dim3(256), 0, A, Rm, YU, ld, n, nbc,
                       h->chol_tasks.p + skip, (const int*)h->chol_deps.p, (const int*)h->col_src.p, flagA, flagR, &h->ctl.p->epoch,
                       fail, h->chol_trace.p, c, h->xp.p, (unsigned long long*)nullptr, head, h->chol_ntask - skip);
    }
    SE2_HIP(hipGetLastError());
    return SE2GPU_OK;
}

// lambda_0 = 1e-5 * max |diag H| into the controller block, device to device (no host round trip).  The estimate is in
// h->poses at this point (first trial of an optimize() call).
int ba_lambda_init(se2gpu_ba* h) {
    hipStream_t st = h->stream;
    if (h->model) {   // diagonal of the pose blocks (observations + prior + odometry), then the common maximum
        SE2_LAUNCH(h->prof, st, "k3_pose_diag", k3_pose_diag, grid1((size_t)h->P * 64, kBlock), dim3(kBlock), 0, h->P,
                   h->pose_ptr.p, h->pose_edges.p, h->Hpp_e.p, h->fixed.p, h->prior_has.p,
                   h->model == 2 ? h->ph.p : h->prior_info.p, h->podo_ptr.p, h->podo_item.p, h->Oii.p, h->Ojj.p, h->diag3.p);
        SE2_LAUNCH(h->prof, st, "k_maxdiag", k_maxdiag, dim3(1), dim3(1024), 0, h->L, h->Hll.p, h->P, h->diag3.p,
                   h->fixed.p, h->scal.p, 6, 0, h->ctl.p);
        SE2_HIP(hipGetLastError());
        return SE2GPU_OK;
    }
    const bool sharded = h->allreduce && h->world > 1;
    const size_t nd = 3 * (size_t)h->P;
    if (!sharded) {   // the pose blocks and the maximum over their and the landmarks' diagonals in one launch, lambda_0 set by it
        SE2_CHECK(ba_pose_blocks(h, h->poses, true, /*odometry_only=*/true));
        const int nP = (h->P + kBlock / 64 - 1) / (kBlock / 64), nL = (h->L + kL0PerBlock - 1) / kL0PerBlock;
        SE2_LAUNCH(h->prof, st, "k_lambda0", k_lambda0, dim3(nP + nL), dim3(kBlock), 0, h->P, h->pose_ptr.p, h->pose_edges.p,
                   h->Hpp_e.p, h->bp_e.p, h->podo_ptr.p, h->podo_item.p, h->Oii.p, h->Ojj.p, h->obi.p, h->obj.p, h->Hpp.p,
                   h->bp.p, h->fixed.p, h->L, h->Hll.p, h->l0_acc.p, h->scal.p, h->ctl.p);
        SE2_HIP(hipGetLastError());
        return SE2GPU_OK;
    }
    SE2_CHECK(ba_pose_blocks(h, h->poses, true));
    SE2_LAUNCH(h->prof, st, "k_extract_diag", k_extract_diag, grid1((size_t)h->P * 3, 256), dim3(256), 0, h->P,
               h->Hpp.p, h->diag3.p);
    if (sharded) {
        // global Hpp diagonal when landmark-sharded.  Every all-reduce goes through the fused buffer `red`
        // (it may alias a caller tensor); at this point of the iteration it holds nothing live.
        SE2_HIP(hipMemcpyAsync(h->red, h->diag3.p, nd * 8, hipMemcpyDeviceToDevice, st));
        SE2_CHECK(ba_allreduce(h, h->red, nd));
        SE2_HIP(hipMemcpyAsync(h->diag3.p, h->red, nd * 8, hipMemcpyDeviceToDevice, st));
    }
    SE2_LAUNCH(h->prof, st, "k_maxdiag", k_maxdiag, dim3(1), dim3(1024), 0, h->L, h->Hll.p, h->P, h->diag3.p,
               h->fixed.p, h->scal.p, 3, 0, (BaCtl*)nullptr);
    if (sharded) {
        // max over ranks through the SUM all-reduce: every rank deposits its local max in its own slot
        SE2_REQUIRE(h->world <= 1024, SE2GPU_ERR_INVALID, "world size %d > 1024", h->world);
        hipLaunchKernelGGL(k_fill_slots, dim3(1), dim3(1024), 0, st, h->red, h->scal.p, h->rank, h->world);
        SE2_CHECK(ba_allreduce(h, h->red, (size_t)h->world));
        hipLaunchKernelGGL(k_set_lambda, dim3(1), dim3(1), 0, st, h->ctl.p, h->red, h->world);
    } else {
        hipLaunchKernelGGL(k_set_lambda, dim3(1), dim3(1), 0, st, h->ctl.p, h->scal.p, 1);
    }
    SE2_HIP(hipGetLastError());
    return SE2GPU_OK;
}

// ---- one LM trial "slot": everything a trial may need, enqueued without knowing what the controller will decide.
// first: the opening trial of an optimize() call (chi^2 of the start, un-fused linearisation, lambda_0, forced Schur).
// know_retry: -1 = unknown (asynchronous mode: the kernels look at the controller), 0 / 1 = the host knows (synchronous
// mode: kernels that would do nothing are not launched, so per-kernel profiles stay clean).
int ba_enqueue_trial(se2gpu_ba* h, bool first, int know_retry, bool notify, double seq) {
    hipStream_t st = h->stream;
    const bool lm = h->run_mode == SE2GPU_BA_LM;
    const bool sharded = h->allreduce != nullptr;
    double* scal = h->red + (size_t)h->ld * h->ld;
    const Bufs B = bufs(h, true);
    auto evaluate = [&](bool step, bool note) -> int {
        if (h->model == 2) {
            if (step)
                SE2_LAUNCH(h->prof, st, "k4_oplus", k4_oplus, grid1(h->P, 64), dim3(64), 0, h->P, B.pa, h->fixed.p, h->xp.p, B.pb, B.c);
            SE2_LAUNCH(h->prof, st, "k4_finalize", k4_finalize, dim3(1), dim3(1024), 0, h->P, h->pg_edges, B.pa, B.pb, h->fixed.p,
                       h->xp.p, h->bp.p, h->prior_has.p, h->prior_meas.p, h->prior_info.p, h->pe_i.p, h->pe_j.p, h->pe_meas.p,
                       h->pe_info.p, scal, h->d_mail, seq, h->ctl.p, step ? 1 : 0, note ? 1 : 0, (const volatile int*)h->d_stop,
                       (double*)nullptr);
            SE2_HIP(hipGetLastError());
            return SE2GPU_OK;
        }
        if (h->model) {
            if (step)
                SE2_LAUNCH(h->prof, st, "k3_oplus", k3_oplus, grid1(h->P, 64), dim3(64), 0, h->P, B.pa, h->fixed.p, h->xp.p, B.pb, B.c);
            SE2_LAUNCH(h->prof, st, "k3_update", k3_update, grid1((size_t)h->L * kGroup, kBlock), dim3(kBlock), 0, h->cam3, h->L,
                       0.0, h->lm_ptr.p, h->e_kf.p, h->e_uv.p, h->e_info.p, B.pa, B.pb, B.la, h->xp.p, h->z.p, h->Y.p, h->bl.p,
                       B.lb, h->part.p, B.c, step ? 1 : 0, (double*)nullptr);
            SE2_LAUNCH(h->prof, st, "k3_finalize", k3_finalize, dim3(1), dim3(1024), 0, h->L ? h->nparts : 0, h->part.p, h->P,
                       B.pa, B.pb, h->fixed.p, h->xp.p, h->bp.p, h->prior_has.p, h->prior_meas.p, h->prior_info.p, h->O,
                       h->o_i.p, h->o_j.p, h->o_meas.p, h->o_info.p, scal, h->d_mail, seq, h->ctl.p, step ? 1 : 0, note ? 1 : 0,
                       (const volatile int*)h->d_stop);
            SE2_HIP(hipGetLastError());
            return SE2GPU_OK;
        }
        const dim3 ug = grid1((size_t)h->L * kGroup, kBlock);
        if (!sharded) {   // the trial ends inside k_update (FinArgs): no kernel boundary before the controller's decision
            FinArgs fin{1, (int)ug.x, h->P, h->O, h->root, step ? 1 : 0, 1, note ? 1 : 0, h->fixed.p, h->xp.p, h->bp.p, B.pa,
                        B.pb, h->o_i.p, h->o_j.p, h->o_meas.p, h->o_info.p, scal, h->d_mail, seq, h->ctl.p,
                        (const volatile int*)h->d_stop, h->fin_counter.p};
            SE2_LAUNCH(h->prof, st, "k_update", k_update, dim3(ug.x + 1), dim3(kBlock), 0, h->cam, h->L, 0.0, h->lm_ptr.p,
                       h->e_kf.p, h->e_uv.p, h->e_info.p, B.pa, h->fixed.p, B.la, step ? h->xp.p : (const double*)nullptr,
                       h->z.p, (const double*)h->Hpl.p, h->bl.p, B.lb, h->part.p, B.c, B.pb, fin, (const double*)h->Dinv.p);
            SE2_HIP(hipGetLastError());
            return SE2GPU_OK;
        }
        SE2_LAUNCH(h->prof, st, "k_update", k_update, ug, dim3(kBlock), 0, h->cam, h->L,
                   0.0, h->lm_ptr.p, h->e_kf.p, h->e_uv.p, h->e_info.p, B.pa, h->fixed.p, B.la,
                   step ? h->xp.p : (const double*)nullptr, h->z.p, (const double*)h->Hpl.p, h->bl.p, B.lb, h->part.p, B.c, B.pb, FinArgs{},
                   (const double*)h->Dinv.p);
        SE2_LAUNCH(h->prof, st, "k_finalize", k_finalize, dim3(1), dim3(1024), 0, h->L ? h->nparts : 0, h->part.p, h->P,
                   0.0, B.pa, h->fixed.p, h->xp.p, h->bp.p, B.pb, h->O, h->o_i.p, h->o_j.p, h->o_meas.p,
                   h->o_info.p, h->root, scal, h->d_mail, seq, h->ctl.p, step ? 1 : 0, sharded ? 0 : 1, note ? 1 : 0,
                   (const volatile int*)h->d_stop);
        if (sharded) {
            SE2_CHECK(ba_allreduce(h, scal, 4));
            hipLaunchKernelGGL(k_lm_decide, dim3(1), dim3(256), 0, st, h->ctl.p, scal, step ? 1 : 0, note ? 1 : 0,
                               h->d_mail, seq, (const volatile int*)h->d_stop);
        }
        SE2_HIP(hipGetLastError());
        return SE2GPU_OK;
    };
    if (first) {
        SE2_CHECK(evaluate(false, false));                  // chi^2 of the starting state -> controller
        if (lm) {
            SE2_CHECK(ba_linearize(h, -1.0, true));         // lambda_0 needs max diag(H) of this linearisation first
            SE2_CHECK(ba_lambda_init(h));
            SE2_CHECK(ba_reduce(h, 0.0, 3, true));
        } else {
            SE2_CHECK(ba_linearize(h, 0.0, true));          // Gauss-Newton: lambda = 0 throughout (controller block is zeroed)
            SE2_CHECK(ba_reduce(h, 0.0, 0, true));
        }
    } else {
        if (know_retry != 1) SE2_CHECK(ba_linearize(h, 0.0, true));
        // (model 0: an undecided slot needs nothing of its own for a retry - k_linearize<FUSED> runs again with the new lambda)
        SE2_CHECK(ba_reduce(h, 0.0, !lm ? 0 : know_retry == 0 ? 0 : know_retry == 1 ? 3 : (h->model ? 2 : 0), true));
    }
    SE2_CHECK(ba_allreduce_system(h));
    SE2_CHECK(ba_solve(h, true));
    SE2_CHECK(evaluate(true, notify));
    return SE2GPU_OK;
}


// =============================================================================================
// Lock-step batches: ONE launch per stage for ALL windows of se2gpu_ba_optimize_batch.
// N windows on N streams are 4 N dependent launches per LM iteration and saturate the command processor at ~157 k
// launches / s (39 k it/s at 64 windows, > 99 % of the chip idle).  Every model-0 kernel body is a __device__ function
// of (block index, arguments); k_batched runs it for a whole batch from per-window argument packs in device memory:
// blockIdx.y = window, blockIdx.x = the window's block.  The windows keep their own controller (BaCtl), mailbox, solver
// flags and buffers - a window whose trial was rejected redoes its lambda part in the next slot while its neighbours
// linearise, a finished window's blocks return at once - so the results are bit-identical to one-by-one runs.  Within a
// window the dispatch order of the 2-D grid (x fastest) is the order the dataflow solve and the finisher of k_update rely on.
// =============================================================================================
template <typename... A>
struct Packed {
    int nblk;
    std::tuple<A...> args;
};
template <auto Body, int BS, typename... A>
__global__ __launch_bounds__(BS) void k_batched(const Packed<A...>* __restrict__ p) {
    const Packed<A...>& x = p[blockIdx.y];
    if ((int)blockIdx.x >= x.nblk) return;
    std::apply([&](const A&... a) { Body(blockIdx.x, a...); }, x.args);
}
// The same with every window pinned to ONE XCD (round 4).  Workgroups are dealt to the 8 XCDs round-robin in dispatch order
// (linear id mod 8), each XCD has its own 4 MB L2, and a kernel that re-reads a window's records (k_reduce2: every W_e is an
// operand of ~5 pairs in different blocks) thrashes them when four windows' blocks pass through every L2 at once (counters:
// 35 % L2 misses, twice the batch's footprint fetched from HBM).  Here XCD x takes the windows x, x + 8, ... one after the
// other: the k-th workgroup of XCD x (k = linear id / 8) is block k mod nb of window 8 (k / nb) + x.  gridDim.x = nb is a
// multiple of 8; windows beyond the last full eight take the plain mapping.  Inside a window the blocks are still
// dispatched in increasing order.
template <auto Body, int BS, typename... A>
__global__ __launch_bounds__(BS) void k_batched_xcd(const Packed<A...>* __restrict__ p) {
    const unsigned nb = gridDim.x, nw8 = gridDim.y & ~7u;
    unsigned w = blockIdx.y, bx = blockIdx.x;
    if (w < nw8) {
        const unsigned lin = w * nb + bx, k = lin >> 3;
        w = 8 * (k / nb) + (lin & 7);
        bx = k % nb;
    }
    const Packed<A...>& x = p[w];
    if ((int)bx >= x.nblk) return;
    std::apply([&](const A&... a) { Body(bx, a...); }, x.args);
}

// the argument packs of one kernel for all windows of a batch; they live in the plan's arena (one upload per plan)
struct BatchArena {
    std::vector<uint8_t> host;
    DevBuf<uint8_t> dev;
    size_t add(const void* src, size_t bytes) {
        const size_t off = (host.size() + 255) & ~(size_t)255;
        host.resize(off + bytes);
        std::memcpy(host.data() + off, src, bytes);
        return off;
    }
};
template <auto Body, int BS>
struct BatchKernel;
template <typename... A, void (*Body)(unsigned, A...), int BS>
struct BatchKernel<Body, BS> {
    using P = Packed<std::remove_cv_t<A>...>;
    std::vector<P> packs;
    size_t off = 0;
    int maxblk = 0;
    void add(int nblk, A... a) {
        packs.push_back(P{nblk, std::tuple<std::remove_cv_t<A>...>(a...)});
        maxblk = std::max(maxblk, nblk);
    }
    void commit(BatchArena& ar) { off = ar.add(packs.data(), packs.size() * sizeof(P)); }
    void launch(const BatchArena& ar, hipStream_t st, int block = BS, size_t shmem = 0) const {
        if (packs.empty() || maxblk <= 0) return;
        hipLaunchKernelGGL((k_batched<Body, BS, std::remove_cv_t<A>...>), dim3((unsigned)maxblk, (unsigned)packs.size()), dim3(block),
                           shmem, st, reinterpret_cast<const P*>(ar.dev.p + off));
    }
    void launch_xcd(const BatchArena& ar, hipStream_t st) const {   // every window on one XCD (k_batched_xcd)
        if (packs.empty() || maxblk <= 0) return;
        // k_batched_xcd's mapping is written for 8 XCDs dealt round-robin (MI355X in SPX mode).  Any other device or
        // partition mode would get correct results from it but windows serialised in groups of eight for nothing: ask.
        static const bool eight = [] {
            int dev = 0, n = 0;
            return hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeNumberOfXccs, dev) == hipSuccess && n == 8;
        }();
        if (!eight) { launch(ar, st); return; }
        hipLaunchKernelGGL((k_batched_xcd<Body, BS, std::remove_cv_t<A>...>), dim3((unsigned)((maxblk + 7) & ~7), (unsigned)packs.size()),
                           dim3(BS), 0, st, reinterpret_cast<const P*>(ar.dev.p + off));
    }
    static const void* kernel() { return (const void*)k_batched<Body, BS, std::remove_cv_t<A>...>; }
};

struct BatchPlan {
    std::vector<se2gpu_ba*> hs;
    std::vector<unsigned long> serials;
    std::vector<hipStream_t> streams;   // every window's stream at build time (se2gpu_ba_set_stream between two batches: rebuild)
    int iters = -1, mode = -1;
    BatchArena arena;
    hipStream_t stream = nullptr;
    BatchKernel<d_ctl_init, 64> ctl_init;
    BatchKernel<d_update, kBlock> eval0, step, step_notify;
    BatchKernel<d_linearize<false>, kBlock> lin0;
    BatchKernel<d_linearize<true>, kBlock> lin;
    BatchKernel<d_odometry, 64> odo;
    BatchKernel<d_pose_reduce, kBlock> pose_reduce;
    BatchKernel<d_maxdiag, 1024> maxdiag;
    BatchKernel<d_reduce2, kBlock> reduce2;
    BatchKernel<d_chol_tiles<false>, 256> chol;
    BatchKernel<d_chol_tiles<true>, 256> chol_v;     // SE2GPU_BA_CHOL_VERIFY=1
    bool verify = false;
    std::vector<hipEvent_t> events;
    ~BatchPlan() {
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
    }
    bool matches(se2gpu_ba** h, int count, int it, int md) const {
        if ((int)hs.size() != count || it != iters || md != mode) return false;
        for (int i = 0; i < count; ++i)
            if (hs[i] != h[i] || serials[i] != h[i]->init_serial || streams[i] != h[i]->stream) return false;
        return true;
    }
};

// Per-call scratch objects that own device memory, pinned memory or events (the plan caches of the lock-step driver, the
// staging of the batched reset) are LEASED from a process-wide pool for the duration of a call instead of living in
// thread_local storage: a host that runs its mapper on short-lived threads would otherwise leave a set behind
// with every thread that exits.  The pool is LIFO - a single calling thread gets its own set back, cached plans included -
// and holds as many sets as calls ever overlapped.  Nothing in it is destroyed at process exit (device memory must not be
// freed after the HIP runtime has shut down), and events handed out from a set (se2gpu_ba::join_event) stay valid for good.
template <typename T>
struct LeasePool {
    std::mutex mu;
    std::vector<T*> idle;
    static LeasePool& get() { static LeasePool* p = new LeasePool; return *p; }
};
template <typename T>
struct Lease {
    T* obj;
    Lease() {
        LeasePool<T>& p = LeasePool<T>::get();
        std::lock_guard<std::mutex> g(p.mu);
        if (p.idle.empty()) obj = new T;
        else { obj = p.idle.back(); p.idle.pop_back(); }
    }
    ~Lease() {
        LeasePool<T>& p = LeasePool<T>::get();
        std::lock_guard<std::mutex> g(p.mu);
        p.idle.push_back(obj);
    }
    Lease(const Lease&) = delete;
    Lease& operator=(const Lease&) = delete;
};
constexpr int kPlanSlots = 4;
struct PlanSet {
    BatchPlan* cache[kPlanSlots] = {};
    unsigned long stamp[kPlanSlots] = {}, clock = 0;
};
struct ResetScratch {
    PinBuf<ResetItem> host;
    DevBuf<ResetItem> dev;
    hipEvent_t ring[64] = {};
    unsigned next = 0;
    unsigned long deaths = 0;   // streams_destroyed() when the ring's last event was recorded
};

// a window the lock-step path can take: SE(2) model, one GPU, device controller, dataflow solve, no per-kernel profile
bool ba_lockstep_ok(const se2gpu_ba* h) {
    return h->initialized && h->model == 0 && !h->allreduce && !h->comm && !h->host_solve && !h->chol_steps && !h->odo_fallback &&
           !h->prof.enabled && h->d_mail && h->L > 0;
}

int ba_build_batch_plan(BatchPlan& bp, se2gpu_ba** hs, int count, int iters, int mode) {
    bp.hs.assign(hs, hs + count);
    bp.serials.resize(count);
    bp.streams.resize(count);
    bp.iters = iters;
    bp.mode = mode;
    bp.stream = hs[0]->stream;
    for (int w = 0; w < count; ++w) {
        se2gpu_ba* h = hs[w];
        bp.serials[w] = h->init_serial;
        bp.streams[w] = h->stream;
        const Bufs B = bufs(h, true);
        double* scal = h->red + (size_t)h->ld * h->ld;
        const dim3 ug = grid1((size_t)h->L * kGroup, kBlock);
        bp.ctl_init.add(1, h->ctl.p, iters, mode);
        auto fin = [&](int step, int notify) {
            return FinArgs{1, (int)ug.x, h->P, h->O, h->root, step, 1, notify, h->fixed.p, h->xp.p, h->bp.p, B.pa, B.pb, h->o_i.p,
                           h->o_j.p, h->o_meas.p, h->o_info.p, scal, h->d_mail, 0.0, h->ctl.p, (const volatile int*)h->d_stop,
                           h->fin_counter.p};
        };
        auto upd = [&](auto& k, int step, int notify) {
            k.add((int)ug.x + 1, h->cam, h->L, 0.0, h->lm_ptr.p, h->e_kf.p, h->e_uv.p, h->e_info.p, B.pa, h->fixed.p, B.la,
                  step ? h->xp.p : (const double*)nullptr, h->z.p, (const double*)h->Hpl.p, h->bl.p, B.lb, h->part.p, B.c, B.pb, fin(step, notify),
                  (const double*)h->Dinv.p);
        };
        upd(bp.eval0, 0, 0);
        upd(bp.step, 1, 0);
        upd(bp.step_notify, 1, 1);
        bp.lin0.add((int)ug.x, h->cam, h->L, h->lm_ptr.p, h->e_kf.p, h->e_uv.p, h->e_info.p, B.pa, h->fixed.p, B.la, h->Hpl.p,
                    h->Hpp_e.p, h->bp_e.p, h->Hll.p, h->bl.p, 0.0, h->Dinv.p, h->z.p, h->Dg.p, B.c, B.pb, B.lb);
        bp.lin.add((int)ug.x, h->cam, h->L, h->lm_ptr.p, h->e_kf.p, h->e_uv.p, h->e_info.p, B.pa, h->fixed.p, B.la, h->Hpl.p,
                   h->Hpp_e.p, h->bp_e.p, h->Hll.p, h->bl.p, 0.0, h->Dinv.p, h->z.p, h->Dg.p, B.c, B.pb, B.lb);
        bp.odo.add(h->O ? (int)grid1(h->O, 64).x : 0, h->O, h->o_i.p, h->o_j.p, h->o_meas.p, h->o_info.p, h->poses_a.p, h->fixed.p,
                   h->Oii.p, h->Ojj.p, h->Oij.p, h->obi.p, h->obj.p, (const BaCtl*)h->ctl.p, (const double*)h->poses_b.p);
        bp.pose_reduce.add((int)grid1((size_t)h->P * 64, kBlock).x, h->P, h->pose_ptr.p, h->pose_edges.p, h->Hpp_e.p, h->bp_e.p,
                           h->podo_ptr.p, h->podo_item.p, h->Oii.p, h->Ojj.p, h->obi.p, h->obj.p, h->Hpp.p, h->bp.p);
        bp.maxdiag.add(1, h->L, h->Hll.p, h->P, h->Hpp.p, h->fixed.p, h->scal.p, 3, 1, h->ctl.p);
        bp.reduce2.add(((h->P + 1 + 7) & ~7) + ((h->nwg_off + 7) & ~7), h->P, h->ld, h->nwg_off, 0.0, h->root, h->grp.p, h->blk_a.p,
                       h->blk_b.p, h->pair_i.p, h->pair_j.p, h->blk_odo.p, h->Hpl.p, h->Dg.p, h->fixed.p, h->pose_ptr.p,
                       h->pose_edges.p, h->podo_ptr.p, h->podo_item.p, h->o_i.p, h->o_j.p, h->o_meas.p, h->o_info.p, B.pa, h->red,
                       h->bp.p, B.c, B.pb, &h->ctl.p->epoch, (const int*)h->pose_off.p, h->nsys);
        {
            const int n = h->nsys, ld = h->ld;
            const int nt = ld / kNB, nbc = (n + kNB - 1) / kNB;
            double* Rm = h->Rinv.p;
            double* YU = Rm + 2 * (size_t)nt * nbc * kSlabs * kSlabDoubles;
            unsigned* flagA = h->chol_flags.p;
            unsigned* flagR = flagA + kSlabs * (size_t)nt * nbc;
            double* fail = h->red + (size_t)ld * ld + 2;
            bp.verify = hs[0]->chol_vfy.p != nullptr;
            if (bp.verify)
                bp.chol_v.add(h->chol_ntask, (const double*)h->red, Rm, YU, ld, n, nbc, h->chol_tasks.p, (const int*)h->chol_deps.p,
                              (const int*)h->col_src.p, flagA, flagR, &h->ctl.p->epoch,
                              fail, h->chol_trace.p, (const BaCtl*)h->ctl.p, h->xp.p, h->chol_vfy.p, h->chol_head.p, h->chol_ntask);
            else
                bp.chol.add(h->chol_ntask, (const double*)h->red, Rm, YU, ld, n, nbc, h->chol_tasks.p, (const int*)h->chol_deps.p,
                            (const int*)h->col_src.p, flagA, flagR, &h->ctl.p->epoch,
                            fail, h->chol_trace.p, (const BaCtl*)h->ctl.p, h->xp.p, (unsigned long long*)nullptr, h->chol_head.p, h->chol_ntask);
        }
    }
    bp.ctl_init.commit(bp.arena); bp.eval0.commit(bp.arena); bp.step.commit(bp.arena); bp.step_notify.commit(bp.arena);
    bp.lin0.commit(bp.arena); bp.lin.commit(bp.arena); bp.odo.commit(bp.arena); bp.pose_reduce.commit(bp.arena);
    bp.maxdiag.commit(bp.arena); bp.reduce2.commit(bp.arena);
    bp.chol.commit(bp.arena); bp.chol_v.commit(bp.arena);
    SE2_CHECK(bp.arena.dev.reserve(bp.arena.host.size()));
    SE2_HIP(hipMemcpyAsync(bp.arena.dev.p, bp.arena.host.data(), bp.arena.host.size(), hipMemcpyHostToDevice, bp.stream));
    while ((int)bp.events.size() < count) {
        hipEvent_t e = nullptr;
        SE2_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        bp.events.push_back(e);
    }
    return SE2GPU_OK;
}

// one trial slot of every window (ba_enqueue_trial, model 0, one GPU, asynchronous controller)
int ba_batch_slot(const BatchPlan& bp, bool first, bool notify) {
    hipStream_t st = bp.stream;
    const BatchArena& ar = bp.arena;
    const bool lm = bp.mode == SE2GPU_BA_LM;
    if (first) {
        bp.eval0.launch(ar, st);
        if (lm) {
            bp.lin0.launch(ar, st);
            bp.odo.launch(ar, st);
            bp.pose_reduce.launch(ar, st);
            bp.maxdiag.launch(ar, st);
            bp.lin.launch(ar, st);     // the records at lambda_0 (the fused pass again: same estimate, identical blocks)
        } else {
            bp.lin.launch(ar, st);
        }
    } else {
        bp.lin.launch(ar, st);
    }
    bp.reduce2.launch_xcd(ar, st);   // every window on one XCD: its W rows stay in that L2 (k_batched_xcd)
    if (bp.verify) bp.chol_v.launch(ar, st);
    else bp.chol.launch(ar, st);
    (notify ? bp.step_notify : bp.step).launch(ar, st);
    SE2_HIP(hipGetLastError());
    return SE2GPU_OK;
}

}  // namespace

static int ba_fetch_estimates(se2gpu_ba* h);

extern "C" {

// The reference constructs a SlamOptimizer on the stack of every localBA call (LocalMapper.cpp:239); a handle's streams,
// mailbox and ~50 device buffers cost milliseconds to create and free, more than the optimisation of a local window.
// Destroyed handles are therefore parked (per device, at most kPoolMax) with their buffers and handed out again by
// se2gpu_ba_create, reset to the state of a new one.  SE2GPU_BA_POOL=0 disables this.
namespace {
constexpr size_t kPoolMax = 4;
static std::mutex g_ba_pool_mu;
static std::vector<se2gpu_ba*> g_ba_pool;
static bool ba_pool_enabled() {
    static const bool on = [] { const char* e = getenv("SE2GPU_BA_POOL"); return !(e && e[0] == '0'); }();
    return on;
}
}  // namespace

int se2gpu_ba_create(se2gpu_ba** out) {
    SE2_REQUIRE(out, SE2GPU_ERR_INVALID, "ba_create: out is NULL");
    SE2_REQUIRE(have_device(), SE2GPU_ERR_NO_DEVICE, "no HIP device visible (libse2gpu has no CPU fallback)");
    int dev = 0;
    SE2_HIP(hipGetDevice(&dev));
    {
        std::lock_guard<std::mutex> lk(g_ba_pool_mu);
        for (size_t i = 0; i < g_ba_pool.size(); ++i)
            if (g_ba_pool[i]->device == dev) {
                *out = g_ba_pool[i];
                g_ba_pool.erase(g_ba_pool.begin() + (ptrdiff_t)i);
                return SE2GPU_OK;
            }
    }
    se2gpu_ba* h = new se2gpu_ba;
    h->device = dev;
    if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        set_error("hipStreamCreate failed");
        return SE2GPU_ERR_HIP;
    }
    h->stream = h->own_stream;
    *out = h;
    return SE2GPU_OK;
}

void se2gpu_ba_destroy(se2gpu_ba* h) {
    if (!h) return;
    if (h->join_event) { (void)hipEventSynchronize(h->join_event); h->join_event = nullptr; h->join_stream = nullptr; }
    if (ba_pool_enabled()) {
        (void)hipStreamSynchronize(h->stream);
        if (h->own_stream != h->stream) (void)hipStreamSynchronize(h->own_stream);
        (void)se2gpu_ba_clear(h);
        h->own_pending = false;
        h->have_tbc = false;
        const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        std::memcpy(h->Rbc, I3, sizeof(I3));
        h->tbc[0] = h->tbc[1] = h->tbc[2] = 0;
        h->stream = h->own_stream;
        h->prof.enabled = false;
        h->prof.reset();
        h->allreduce = nullptr; h->ar_user = nullptr; h->comm = nullptr; h->ar_buffer = nullptr;
        h->red = nullptr;
        h->root = 1; h->rank = 0; h->world = 1;
        std::lock_guard<std::mutex> lk(g_ba_pool_mu);
        if (g_ba_pool.size() < kPoolMax) {
            g_ba_pool.push_back(h);
            return;
        }
    }
    delete h;
}

int se2gpu_ba_clear(se2gpu_ba* h) {
    SE2_REQUIRE(h, SE2GPU_ERR_INVALID, "ba handle is NULL");
    h->pose_of_id.clear(); h->lm_of_id.clear(); h->pose_ids.clear(); h->lm_ids.clear();
    h->h_poses.clear(); h->h_lms.clear(); h->h_fixed.clear(); h->odo.clear();
    h->he_kf.clear(); h->he_lm.clear(); h->he_uv.clear(); h->he_info.clear();
    h->huber_delta = 0; h->huber_mixed = false;
    h->bulk_E = 0; h->bulk_kf = h->bulk_lm = nullptr; h->bulk_uv = h->bulk_info = nullptr;
    h->model = 0; h->D = 3; h->ps = 3;
    h->h_prior_has.clear(); h->h_prior_meas.clear(); h->h_prior_info.clear(); h->odo3.clear(); h->edge_perm.clear();
    h->pg_edges = 0; h->pg_perm.clear();
    h->lg_active = false;
    h->lg_lc.clear(); h->lg_lw.clear(); h->lg_sigma2.clear(); h->lg_Rcw.clear(); h->lg_twb.clear();
    h->have_cam = false;
    h->initialized = false;
    h->est_valid = false;
    return SE2GPU_OK;
}

// Pre-warm: the first initializeOptimization of a process pays for the code objects of ~40 kernels,
// a stream, the mapped mailbox and ~50 device allocations - 3.5 to 5.2 ms against 0.2 to 0.5 ms in steady state, i.e. the
// first localBA after start-up would miss the mapper's budget.  se2gpu_ba_reserve(P, L, E) runs one throw-away window of
// that size (P key frames on a line, L landmarks in front of them, E observations with exact measurements, the P - 1
// odometry edges) through initialize + optimize(1) and parks the handle in the pool: the next se2gpu_ba_create returns it
// with every buffer already large enough.  Call it once at start-up, off the tracking thread (LocalMapper's constructor).
int se2gpu_ba_reserve(int P, int L, int E) {
    SE2_REQUIRE(P >= 2 && L >= 1 && E >= L, SE2GPU_ERR_INVALID, "ba_reserve: need P >= 2, L >= 1, E >= L (got %d, %d, %d)", P, L, E);
    SE2_REQUIRE(ba_pool_enabled(), SE2GPU_ERR_STATE, "ba_reserve: the handle pool is disabled (SE2GPU_BA_POOL=0)");
    E = (int)std::min<long long>(E, (long long)L * P);
    se2gpu_ba* h = nullptr;
    SE2_CHECK(se2gpu_ba_create(&h));
    struct Guard { se2gpu_ba* h; ~Guard() { se2gpu_ba_destroy(h); } } guard{h};
    const double fx = 400, cx = 320, cy = 240;
    SE2_CHECK(se2gpu_ba_add_cam(h, fx, cx, cy));
    for (int i = 0; i < P; ++i) SE2_CHECK(se2gpu_ba_add_vertex_se2(h, i, 100.0 * i, 0.0, 0.0, i == 0));
    const double info3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int i = 0; i + 1 < P; ++i) {
        const double meas[3] = {100.0, 0.0, 0.0};
        SE2_CHECK(se2gpu_ba_add_edge_se2(h, i, i + 1, meas, info3));
    }
    // Tbc = identity: the camera frame is the body frame, so a landmark at height z is at depth z for every pose on the plane
    std::vector<double> lm(3 * (size_t)L);
    for (int l = 0; l < L; ++l) {
        lm[3 * (size_t)l] = 50.0 * P + 37.0 * (l % 41) - 700.0;
        lm[3 * (size_t)l + 1] = 23.0 * (l % 53) - 600.0;
        lm[3 * (size_t)l + 2] = 3000.0 + 11.0 * (l % 97);
        SE2_CHECK(se2gpu_ba_add_vertex_xyz(h, P + l, &lm[3 * (size_t)l], 1, 0));
    }
    const double info2[4] = {1, 0, 0, 1};
    const int base = E / L, extra = E % L;
    for (int l = 0; l < L; ++l) {
        const int k = base + (l < extra ? 1 : 0);
        for (int j = 0; j < k; ++j) {
            const int kf = (int)(((long long)l * 7 + j) % P);
            const double xc = lm[3 * (size_t)l] - 100.0 * kf, yc = lm[3 * (size_t)l + 1], zc = lm[3 * (size_t)l + 2];
            const double uv[2] = {fx * xc / zc + cx, fx * yc / zc + cy};
            SE2_CHECK(se2gpu_ba_add_edge_se2xyz(h, kf, P + l, uv, info2, 2.4477));
        }
    }
    SE2_CHECK(se2gpu_ba_initialize(h));
    se2gpu_ba_stats st;
    SE2_CHECK(se2gpu_ba_optimize(h, 1, SE2GPU_BA_LM, nullptr, 0, &st));
    double xyt[3];
    SE2_CHECK(se2gpu_ba_get_se2(h, 0, xyt));   // the estimate download path (pinned staging) as well
    return SE2GPU_OK;                          // ~Guard parks the handle
}

int se2gpu_ba_set_stream(se2gpu_ba* h, void* s) {
    SE2_REQUIRE(h, SE2GPU_ERR_INVALID, "ba handle is NULL");
    h->stream = s ? (hipStream_t)s : h->own_stream;
    return SE2GPU_OK;
}
void* se2gpu_ba_stream(se2gpu_ba* h) { return h ? (void*)h->stream : nullptr; }

int se2gpu_ba_add_cam(se2gpu_ba* h, double f, double cx, double cy) {
    SE2_REQUIRE(h, SE2GPU_ERR_INVALID, "ba handle is NULL");
    h->cam.fx = f; h->cam.cx = cx; h->cam.cy = cy;
    h->have_cam = true;
    return SE2GPU_OK;
}

int se2gpu_ba_set_Tbc(se2gpu_ba* h, const double R[9], const double t[3]) {
    SE2_REQUIRE(h && R && t, SE2GPU_ERR_INVALID, "set_Tbc: NULL argument");
    std::memcpy(h->Rbc, R, 72);
    std::memcpy(h->tbc, t, 24);
    h->have_tbc = true;
    return SE2GPU_OK;
}

int se2gpu_ba_add_vertex_se2(se2gpu_ba* h, int id, double x, double y, double theta, int fixed) {
    SE2_REQUIRE(h, SE2GPU_ERR_INVALID, "ba handle is NULL");
    SE2_REQUIRE(!h->initialized, SE2GPU_ERR_STATE, "graph is frozen; call se2gpu_ba_clear first");
    SE2_REQUIRE(h->pose_of_id.find(id) < 0 && h->lm_of_id.find(id) < 0, SE2GPU_ERR_INVALID, "duplicate vertex id %d", id);
    SE2_REQUIRE(h->model == 0, SE2GPU_ERR_STATE, "VertexSE2 in a graph of VertexSE3Expmap poses");
    h->pose_of_id.set(id, (int)h->pose_ids.size());
    h->pose_ids.push_back(id);
    h->h_poses.push_back(x); h->h_poses.push_back(y); h->h_poses.push_back(normalize_theta(theta));  // SE2 ctor
    h->h_fixed.push_back(fixed ? 1 : 0);
    return SE2GPU_OK;
}

int se2gpu_ba_add_vertex_xyz(se2gpu_ba* h, int id, const double xyz[3], int marginal, int fixed) {
    SE2_REQUIRE(h && xyz, SE2GPU_ERR_INVALID, "add_vertex_xyz: NULL argument");
    SE2_REQUIRE(!h->initialized, SE2GPU_ERR_STATE, "graph is frozen; call se2gpu_ba_clear first");
    SE2_REQUIRE(marginal && !fixed, SE2GPU_ERR_INVALID,
                "only marginalised, free landmarks are supported (addVertexSBAXYZ defaults, optimizer.h:91)");
    SE2_REQUIRE(h->pose_of_id.find(id) < 0 && h->lm_of_id.find(id) < 0, SE2GPU_ERR_INVALID, "duplicate vertex id %d", id);
    h->lm_of_id.set(id, (int)h->lm_ids.size());
    h->lm_ids.push_back(id);
    h->h_lms.push_back(xyz[0]); h->h_lms.push_back(xyz[1]); h->h_lms.push_back(xyz[2]);
    return SE2GPU_OK;
}

// a bulk load followed by single-edge calls: the borrowed arrays become owned ones first
static void ba_materialize_bulk(se2gpu_ba* h) {
    if (!h->bulk_E) return;
    const size_t E = (size_t)h->bulk_E;
    h->he_kf.insert(h->he_kf.end(), h->bulk_kf, h->bulk_kf + E);
    h->he_lm.insert(h->he_lm.end(), h->bulk_lm, h->bulk_lm + E);
    h->he_uv.insert(h->he_uv.end(), h->bulk_uv, h->bulk_uv + 2 * E);
    h->he_info.insert(h->he_info.end(), h->bulk_info, h->bulk_info + 3 * E);
    h->bulk_E = 0;
}

int se2gpu_ba_add_edge_se2xyz(se2gpu_ba* h, int id_kf, int id_mp, const double uv[2], const double info[4],
                              double huber_delta) {
    SE2_REQUIRE(h && uv && info, SE2GPU_ERR_INVALID, "add_edge_se2xyz: NULL argument");
    SE2_REQUIRE(!h->initialized, SE2GPU_ERR_STATE, "graph is frozen; call se2gpu_ba_clear first");
    const int a = h->pose_of_id.find(id_kf), b = h->lm_of_id.find(id_mp);
    SE2_REQUIRE(a >= 0 && b >= 0, SE2GPU_ERR_INVALID, "add_edge_se2xyz: unknown vertex id (%d, %d)", id_kf, id_mp);
    ba_materialize_bulk(h);
    if (h->he_kf.empty()) h->huber_delta = huber_delta;
    else if (huber_delta != h->huber_delta) h->huber_mixed = true;
    h->he_kf.push_back(a); h->he_lm.push_back(b);
    h->he_uv.push_back(uv[0]); h->he_uv.push_back(uv[1]);
    h->he_info.push_back(info[0]); h->he_info.push_back(0.5 * (info[1] + info[2])); h->he_info.push_back(info[3]);
    return SE2GPU_OK;
}

int se2gpu_ba_add_edge_se2(se2gpu_ba* h, int id0, int id1, const double meas[3], const double info[9]) {
    SE2_REQUIRE(h && meas && info, SE2GPU_ERR_INVALID, "add_edge_se2: NULL argument");
    SE2_REQUIRE(!h->initialized, SE2GPU_ERR_STATE, "graph is frozen; call se2gpu_ba_clear first");
    const int a = h->pose_of_id.find(id0), b = h->pose_of_id.find(id1);
    SE2_REQUIRE(a >= 0 && b >= 0, SE2GPU_ERR_INVALID, "add_edge_se2: unknown pose id (%d, %d)", id0, id1);
    EdgeOdo e;
    e.i = a; e.j = b;
    std::memcpy(e.meas, meas, 24);
    std::memcpy(e.info, info, 72);
    h->odo.push_back(e);
    return SE2GPU_OK;
}

int se2gpu_ba_load(se2gpu_ba* h, int P, int L, int E, int O, const double* poses, const uint8_t* fixed,
                   const double* lms, const int32_t* e_kf, const int32_t* e_lm, const double* e_uv,
                   const double* e_info, const int32_t* o_i, const int32_t* o_j, const double* o_meas,
                   const double* o_info, double huber_delta) {
    SE2_REQUIRE(h, SE2GPU_ERR_INVALID, "ba handle is NULL");
    SE2_REQUIRE(!h->initialized, SE2GPU_ERR_STATE, "graph is frozen; call se2gpu_ba_clear first");
    SE2_REQUIRE(P >= 0 && L >= 0 && E >= 0 && O >= 0, SE2GPU_ERR_INVALID, "negative size");
    for (int p = 0; p < P; ++p)
        SE2_CHECK(se2gpu_ba_add_vertex_se2(h, p, poses[3 * p], poses[3 * p + 1], poses[3 * p + 2], fixed[p]));
    if (h->lm_ids.empty() && h->pose_ids.size() == (size_t)P) {   // fresh handle: the landmark block in one go
        h->lm_ids.resize(L);
        for (int l = 0; l < L; ++l) { h->lm_ids[l] = P + l; h->lm_of_id.set(P + l, l); }
        h->h_lms.assign(lms, lms + 3 * (size_t)L);
    } else {
        for (int l = 0; l < L; ++l) SE2_CHECK(se2gpu_ba_add_vertex_xyz(h, P + l, lms + 3 * (size_t)l, 1, 0));
    }
    if (E) {
        // the bulk arrays ARE the internal layout (indices, (xx, xy, yy)): they are borrowed until se2gpu_ba_initialize,
        // which validates them and copies them once, into the pinned upload arena
        if (h->he_kf.empty() && !h->bulk_E) {
            h->huber_delta = huber_delta;
            h->bulk_E = E; h->bulk_kf = e_kf; h->bulk_lm = e_lm; h->bulk_uv = e_uv; h->bulk_info = e_info;
        } else {
            ba_materialize_bulk(h);
            if (huber_delta != h->huber_delta) h->huber_mixed = true;
            h->he_kf.insert(h->he_kf.end(), e_kf, e_kf + E);
            h->he_lm.insert(h->he_lm.end(), e_lm, e_lm + E);
            h->he_uv.insert(h->he_uv.end(), e_uv, e_uv + 2 * (size_t)E);
            h->he_info.insert(h->he_info.end(), e_info, e_info + 3 * (size_t)E);
        }
    }
    for (int k = 0; k < O; ++k) SE2_CHECK(se2gpu_ba_add_edge_se2(h, o_i[k], o_j[k], o_meas + 3 * k, o_info + 9 * k));
    return SE2GPU_OK;
}

// ---- SE3-expmap graph construction (optimizer.h:82-98): addVertexSE3Expmap, addPlaneMotionSE3Expmap's prior edge,
// addEdgeSE3Expmap, addEdgeXYZ2UV.  pose12 = rotation row-major (9) then translation (3) of Tcw.
int se2gpu_ba_add_vertex_se3(se2gpu_ba* h, int id, const double pose12[12], int fixed) {
    SE2_REQUIRE(h && pose12, SE2GPU_ERR_INVALID, "add_vertex_se3: NULL argument");
    SE2_REQUIRE(!h->initialized, SE2GPU_ERR_STATE, "graph is frozen; call se2gpu_ba_clear first");
    SE2_REQUIRE(h->model == 1 || h->pose_ids.empty(), SE2GPU_ERR_STATE, "VertexSE3Expmap in a graph of VertexSE2 poses");
    SE2_REQUIRE(h->pose_of_id.find(id) < 0 && h->lm_of_id.find(id) < 0, SE2GPU_ERR_INVALID, "duplicate vertex id %d", id);
    h->model = 1; h->D = 6; h->ps = 12;
    h->pose_of_id.set(id, (int)h->pose_ids.size());
    h->pose_ids.push_back(id);
    Se3 T = se3_load(pose12);
    normalize_rotation(T.R);                       // SE3Quat(R, t) normalises its quaternion
    double q[12];
    se3_store(T, q);
    h->h_poses.insert(h->h_poses.end(), q, q + 12);
    h->h_fixed.push_back(fixed ? 1 : 0);
    h->h_prior_has.push_back(0);
    h->h_prior_meas.insert(h->h_prior_meas.end(), q, q + 12);
    h->h_prior_info.insert(h->h_prior_info.end(), 36, 0.0);
    return SE2GPU_OK;
}

// g2o::VertexSE3 (pose = T_w_c as an Isometry3D): the pose-graph model of GlobalMapper::GlobalBA.  A prior added with
// se2gpu_ba_add_prior_se3 is then an EdgeSE3Prior, an edge added with se2gpu_ba_add_edge_se3 an EdgeSE3 - error vectors
// and information matrices in the order (translation, rotation).
int se2gpu_ba_add_vertex_iso3(se2gpu_ba* h, int id, const double pose12[12], int fixed) {
    SE2_REQUIRE(h && pose12, SE2GPU_ERR_INVALID, "add_vertex_iso3: NULL argument");
    SE2_REQUIRE(!h->initialized, SE2GPU_ERR_STATE, "graph is frozen; call se2gpu_ba_clear first");
    SE2_REQUIRE(h->model == 2 || h->pose_ids.empty(), SE2GPU_ERR_STATE, "VertexSE3 in a graph of another pose type");
    SE2_REQUIRE(h->lm_ids.empty(), SE2GPU_ERR_STATE, "the pose graph has no landmarks");
    SE2_REQUIRE(h->pose_of_id.find(id) < 0, SE2GPU_ERR_INVALID, "duplicate vertex id %d", id);
    h->model = 2; h->D = 6; h->ps = 12;
    h->pose_of_id.set(id, (int)h->pose_ids.size());
    h->pose_ids.push_back(id);
    h->h_poses.insert(h->h_poses.end(), pose12, pose12 + 12);
    h->h_fixed.push_back(fixed ? 1 : 0);
    h->h_prior_has.push_back(0);
    h->h_prior_meas.insert(h->h_prior_meas.end(), pose12, pose12 + 12);
    h->h_prior_info.insert(h->h_prior_info.end(), 36, 0.0);
    h->have_cam = true;   // no camera in a pose graph
    return SE2GPU_OK;
}

// addVertexSE3PlaneMotion's prior (src/optimizer.cpp:336-470, the current branch): measurement = T_w_c with the body's
// roll, pitch and height removed; information = AdjTR(Tbc)' diag(1e-4, 1e-4, z, xrot, yrot, 1e-4) AdjTR(Tbc) in the
// order (translation, rotation), AdjTR(T) = [R skew(t) R; 0 R] (:95-104).  Graph construction: host code.
int se2gpu_plane_motion_prior_iso3(const double* Twc12, const double* Tbc12, double xrot_info, double yrot_info,
                                   double z_info, double* meas12, double* info36) {
    SE2_REQUIRE(Twc12 && Tbc12 && meas12 && info36, SE2GPU_ERR_INVALID, "plane_motion_prior_iso3: NULL argument");
    const Se3 Twc = se3_load(Twc12), Tbc = se3_load(Tbc12);
    Se3 Twb = iso_mul(Twc, se3_inv(Tbc));
    double q[4];
    quat_of(Twb.R, q);
    const double nv = std::sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    double yaw = 0;
    if (nv > 0) yaw = 2 * std::atan2(nv, q[0]) * q[3] / nv;    // z component of the rotation vector (Eigen::AngleAxisd)
    const double qz[4] = {std::cos(0.5 * yaw), 0, 0, std::sin(0.5 * yaw)};
    mat_of_quat(qz, Twb.R);
    Twb.t[2] = 0;
    se3_store(iso_mul(Twb, Tbc), meas12);
    double A[36] = {0}, sk[9], sR[9];
    skew3(Tbc.t, sk);
    mat3_mul(sk, Tbc.R, sR);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            A[6 * i + j] = Tbc.R[3 * i + j];
            A[6 * (i + 3) + (j + 3)] = Tbc.R[3 * i + j];
            A[6 * i + (j + 3)] = sR[3 * i + j];
        }
    const double D[6] = {1e-4, 1e-4, z_info, xrot_info, yrot_info, 1e-4};
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double v = 0;
            for (int k = 0; k < 6; ++k) v += A[6 * k + i] * D[k] * A[6 * k + j];
            info36[6 * i + j] = v;
        }
    return SE2GPU_OK;
}

int se2gpu_ba_add_prior_se3(se2gpu_ba* h, int id, const double meas12[12], const double info36[36]) {
    SE2_REQUIRE(h && meas12 && info36, SE2GPU_ERR_INVALID, "add_prior_se3: NULL argument");
    SE2_REQUIRE(!h->initialized, SE2GPU_ERR_STATE, "graph is frozen; call se2gpu_ba_clear first");
    const int a = h->model >= 1 ? h->pose_of_id.find(id) : -1;
    SE2_REQUIRE(a >= 0, SE2GPU_ERR_INVALID, "add_prior_se3: unknown SE3 pose id %d", id);
    SE2_REQUIRE(!h->h_prior_has[a], SE2GPU_ERR_INVALID, "pose %d already has an EdgeSE3ExpmapPrior", id);
    h->h_prior_has[a] = 1;
    std::memcpy(&h->h_prior_meas[12 * (size_t)a], meas12, 96);
    std::memcpy(&h->h_prior_info[36 * (size_t)a], info36, 288);
    return SE2GPU_OK;
}

int se2gpu_ba_add_edge_se3(se2gpu_ba* h, int id0, int id1, const double meas12[12], const double info36[36]) {
    SE2_REQUIRE(h && meas12 && info36, SE2GPU_ERR_INVALID, "add_edge_se3: NULL argument");
    SE2_REQUIRE(!h->initialized, SE2GPU_ERR_STATE, "graph is frozen; call se2gpu_ba_clear first");
    const int a = h->model >= 1 ? h->pose_of_id.find(id0) : -1, b = h->model >= 1 ? h->pose_of_id.find(id1) : -1;
    SE2_REQUIRE(a >= 0 && b >= 0, SE2GPU_ERR_INVALID, "add_edge_se3: unknown SE3 pose id (%d, %d)", id0, id1);
    SE2_REQUIRE(a != b, SE2GPU_ERR_INVALID, "add_edge_se3: self loop on pose %d", id0);
    if (h->model == 1)   // the pose graph groups parallel edges into slots; the expmap model carries one per pair
        for (const auto& o : h->odo3)
            SE2_REQUIRE(!((o.i == a && o.j == b) || (o.i == b && o.j == a)), SE2GPU_ERR_INVALID,
                        "a second EdgeSE3Expmap between poses %d and %d", id0, id1);
    se2gpu_ba::Odo3 e;
    e.i = a; e.j = b;
    std::memcpy(e.meas, meas12, 96);
    std::memcpy(e.info, info36, 288);
    h->odo3.push_back(e);
    return SE2GPU_OK;
}

// addEdgeXYZ2UV(opt, measure, idMP, idKF, paraId, info, thHuber) (optimizer.h:97): information = inv_sigma2 * I
int se2gpu_ba_add_edge_xyz2uv(se2gpu_ba* h, int id_mp, int id_kf, const double uv[2], double inv_sigma2, double huber_delta) {
    SE2_REQUIRE(h && uv, SE2GPU_ERR_INVALID, "add_edge_xyz2uv: NULL argument");
    SE2_REQUIRE(!h->initialized, SE2GPU_ERR_STATE, "graph is frozen; call se2gpu_ba_clear first");
    const int a = h->model == 1 ? h->pose_of_id.find(id_kf) : -1, b = h->lm_of_id.find(id_mp);
    SE2_REQUIRE(a >= 0 && b >= 0, SE2GPU_ERR_INVALID, "add_edge_xyz2uv: unknown vertex id (%d, %d)", id_mp, id_kf);
    if (h->he_kf.empty()) h->huber_delta = huber_delta;
    else if (huber_delta != h->huber_delta) h->huber_mixed = true;
    h->he_kf.push_back(a); h->he_lm.push_back(b);
    h->he_uv.push_back(uv[0]); h->he_uv.push_back(uv[1]);
    h->he_info.push_back(inv_sigma2); h->he_info.push_back(0.0); h->he_info.push_back(inv_sigma2);
    return SE2GPU_OK;
}

int se2gpu_ba_get_se3(se2gpu_ba* h, int id, double pose12[12]) {
    SE2_REQUIRE(h && h->initialized && pose12 && h->model >= 1, SE2GPU_ERR_STATE, "get_se3 needs an initialised SE3 graph");
    const int a = h->pose_of_id.find(id);
    SE2_REQUIRE(a >= 0, SE2GPU_ERR_INVALID, "unknown pose id %d", id);
    SE2_CHECK(ba_fetch_estimates(h));
    std::memcpy(pose12, h->est.p + 12 * (size_t)a, 96);
    return SE2GPU_OK;
}

// EdgeProjectXYZ2UV::chi2() of every projection edge at the current estimate, in the order the edges were added
// (LocalMapper::removeOutlierChi2 compares it with 25, LocalMapper.cpp:199-214)
int se2gpu_ba_edge_chi2(se2gpu_ba* h, double* chi2, int cap) {
    SE2_REQUIRE(h && h->initialized && chi2 && h->model >= 1, SE2GPU_ERR_STATE, "edge_chi2 needs an initialised SE3 graph");
    SE2_CHECK(ba_join(h));
    hipStream_t st = h->stream;
    if (h->model == 2) {   // EdgeSE3::chi2() of every edge of the pose graph, in the order the edges were added
        const int NE = h->pg_edges, NT = (int)h->odo3.size();   // active (level 0 at the last initialize) / all edges
        SE2_REQUIRE(cap >= NT, SE2GPU_ERR_CAPACITY, "edge_chi2: %d edges, room for %d", NT, cap);
        for (int k = 0; k < NT; ++k) chi2[k] = 0.0;   // an edge outside the optimised level reports 0
        if (!NE) return SE2GPU_OK;
        hipLaunchKernelGGL(k4_finalize, dim3(1), dim3(1024), 0, st, h->P, NE, h->poses, h->poses_t, h->fixed.p, h->xp.p, h->bp.p,
                           h->prior_has.p, h->prior_meas.p, h->prior_info.p, h->pe_i.p, h->pe_j.p, h->pe_meas.p, h->pe_info.p,
                           (double*)nullptr, (volatile double*)nullptr, 0.0, (BaCtl*)nullptr, 0, 0, (const volatile int*)nullptr,
                           h->edge_chi2.p);
        SE2_HIP(hipGetLastError());
        std::vector<double> tmp(NE);
        SE2_HIP(hipMemcpyAsync(tmp.data(), h->edge_chi2.p, (size_t)NE * 8, hipMemcpyDeviceToHost, st));
        SE2_HIP(hipStreamSynchronize(st));
        for (int t = 0; t < NE; ++t) chi2[h->pg_perm[t]] = tmp[t];
        return SE2GPU_OK;
    }
    SE2_REQUIRE(cap >= h->E, SE2GPU_ERR_CAPACITY, "edge_chi2: %d edges, room for %d", h->E, cap);
    if (!h->E) return SE2GPU_OK;
    hipLaunchKernelGGL(k3_update, grid1((size_t)h->L * kGroup, kBlock), dim3(kBlock), 0, st, h->cam3, h->L, 0.0, h->lm_ptr.p,
                       h->e_kf.p, h->e_uv.p, h->e_info.p, h->poses, h->poses_t, h->lms, (const double*)nullptr, h->z.p, h->Y.p,
                       h->bl.p, h->lms_t, h->part.p, (const BaCtl*)nullptr, 0, h->edge_chi2.p);
    SE2_HIP(hipGetLastError());
    std::vector<double> tmp(h->E);
    SE2_HIP(hipMemcpyAsync(tmp.data(), h->edge_chi2.p, (size_t)h->E * 8, hipMemcpyDeviceToHost, st));
    SE2_HIP(hipStreamSynchronize(st));
    if (h->edge_perm.empty()) std::memcpy(chi2, tmp.data(), (size_t)h->E * 8);
    else for (int t = 0; t < h->E; ++t) chi2[h->edge_perm[t]] = tmp[t];
    return SE2GPU_OK;
}

// Map::loadLocalGraph (/root/reference/src/Map.cpp:891-1022) through ONE call on a POD view of the local window.
// What the reference does with ~E std::find calls over the key-frame vectors, E heap-allocated edges and E Eigen 2x2
// inversions becomes: the vertex ids of Map.cpp:925/966/985, the fixed rule of :927/:969, cov^-1 of the PreSE2 edges
// (:942-953) on the host (at most one per key frame), and flat copies; the per-observation information (:1024-1049) is
// evaluated on the device, straight into the edge array, by initialize.
int se2gpu_ba_load_local_graph(se2gpu_ba* h, const se2gpu_local_graph* g) {
    SE2_REQUIRE(h && g, SE2GPU_ERR_INVALID, "load_local_graph: NULL argument");
    SE2_REQUIRE(!h->initialized, SE2GPU_ERR_STATE, "graph is frozen; call se2gpu_ba_clear first");
    SE2_REQUIRE(h->pose_ids.empty() && h->lm_ids.empty() && h->he_kf.empty() && h->odo.empty(), SE2GPU_ERR_STATE,
                "load_local_graph needs an empty optimizer");
    const int nL = g->n_local_kf, nR = g->n_ref_kf, nK = nL + nR, N = g->n_mp, M = g->n_obs;
    SE2_REQUIRE(nL > 0 && nR >= 0 && N >= 0 && M >= 0, SE2GPU_ERR_INVALID, "load_local_graph: bad sizes");
    SE2_REQUIRE(g->kf_id && g->kf_Twb && g->kf_Rcw && (N == 0 || g->mp_pos) &&
                (M == 0 || (g->obs_mp && g->obs_kf && g->obs_uv && g->obs_lc && g->obs_sigma2)), SE2GPU_ERR_INVALID,
                "load_local_graph: NULL array");
    SE2_CHECK(se2gpu_ba_add_cam(h, g->fx, g->cx, g->cy));
    SE2_CHECK(se2gpu_ba_set_Tbc(h, g->Rbc, g->tbc));
    // "If no reference KF, the KF with minId should be fixed" (Map.cpp:899-912); KeyFrame id 1 is always fixed (:927)
    int minKFid = -1;
    if (nR == 0) {
        minKFid = g->kf_id[0];
        for (int i = 0; i < nL; ++i) minKFid = std::min(minKFid, g->kf_id[i]);
    }
    for (int i = 0; i < nL; ++i)
        SE2_CHECK(se2gpu_ba_add_vertex_se2(h, i, g->kf_Twb[3 * i], g->kf_Twb[3 * i + 1], g->kf_Twb[3 * i + 2],
                                           (g->kf_id[i] == minKFid) || g->kf_id[i] == 1));
    if (g->odo_to)
        for (int i = 0; i < nL; ++i) {
            const int j = g->odo_to[i];
            if (j < 0) continue;
            SE2_REQUIRE(j < nL && g->odo_meas && g->odo_cov, SE2GPU_ERR_INVALID, "load_local_graph: odo_to[%d] = %d", i, j);
            const double* c = g->odo_cov + 9 * (size_t)i;   // info = cov^-1 (Eigen's closed-form 3x3 inverse: cofactors / det)
            const double A = c[4] * c[8] - c[5] * c[7], B = -(c[3] * c[8] - c[5] * c[6]), C = c[3] * c[7] - c[4] * c[6];
            const double id = 1.0 / (c[0] * A + c[1] * B + c[2] * C);
            const double info[9] = {A * id, -(c[1] * c[8] - c[2] * c[7]) * id, (c[1] * c[5] - c[2] * c[4]) * id,
                                    B * id, (c[0] * c[8] - c[2] * c[6]) * id, -(c[0] * c[5] - c[2] * c[3]) * id,
                                    C * id, -(c[0] * c[7] - c[1] * c[6]) * id, (c[0] * c[4] - c[1] * c[3]) * id};
            SE2_CHECK(se2gpu_ba_add_edge_se2(h, i, j, g->odo_meas + 3 * (size_t)i, info));
        }
    for (int i = 0; i < nR; ++i)
        SE2_CHECK(se2gpu_ba_add_vertex_se2(h, nL + i, g->kf_Twb[3 * (nL + i)], g->kf_Twb[3 * (nL + i) + 1],
                                           g->kf_Twb[3 * (nL + i) + 2], 1));
    const int maxKFid = nL + nR + 1;
    h->lm_of_id.reserve((size_t)maxKFid + N);
    for (int i = 0; i < N; ++i) {
        const double lw[3] = {g->mp_pos[3 * (size_t)i], g->mp_pos[3 * (size_t)i + 1], g->mp_pos[3 * (size_t)i + 2]};
        SE2_CHECK(se2gpu_ba_add_vertex_xyz(h, maxKFid + i, lw, 1, 0));
    }
    h->he_kf.reserve(M); h->he_lm.reserve(M); h->he_uv.reserve(2 * (size_t)M);
    h->lg_lc.reserve(3 * (size_t)M); h->lg_lw.reserve(3 * (size_t)M); h->lg_sigma2.reserve(M);
    int prev = 0;
    for (int k = 0; k < M; ++k) {
        const int mp = g->obs_mp[k], kf = g->obs_kf[k];
        SE2_REQUIRE(mp >= prev && mp < N, SE2GPU_ERR_INVALID, "load_local_graph: obs_mp must be nondecreasing and < n_mp (obs %d)", k);
        prev = mp;
        if (kf < 0) continue;   // the observing key frame is in neither list (Map.cpp:1016-1017)
        SE2_REQUIRE(kf < nK, SE2GPU_ERR_INVALID, "load_local_graph: obs_kf[%d] = %d", k, kf);
        h->he_kf.push_back(kf); h->he_lm.push_back(mp);
        h->he_uv.push_back(g->obs_uv[2 * (size_t)k]); h->he_uv.push_back(g->obs_uv[2 * (size_t)k + 1]);
        for (int c = 0; c < 3; ++c) {
            h->lg_lc.push_back(g->obs_lc[3 * (size_t)k + c]);
            h->lg_lw.push_back(g->mp_pos[3 * (size_t)mp + c]);
        }
        h->lg_sigma2.push_back(g->obs_sigma2[k]);
    }
    h->he_info.assign(3 * h->he_kf.size(), 0.0);
    h->huber_delta = g->huber_delta;
    h->lg_Rcw.assign(g->kf_Rcw, g->kf_Rcw + 9 * (size_t)nK);
    h->lg_twb.resize(2 * (size_t)nK);
    for (int i = 0; i < nK; ++i) { h->lg_twb[2 * i] = g->kf_Twb[3 * i]; h->lg_twb[2 * i + 1] = g->kf_Twb[3 * i + 1]; }
    h->lg_fx = g->fx;
    h->lg_srot = (float)(1. / g->xrot_info);
    h->lg_sz = (float)(1. / g->z_info);
    h->lg_active = true;
    return SE2GPU_OK;
}

int se2gpu_ba_initialize(se2gpu_ba* h) {
    SE2_REQUIRE(h, SE2GPU_ERR_INVALID, "ba handle is NULL");
    // The graph is frozen by the first call (the arrays borrowed by se2gpu_ba_load are released there).  g2o allows a
    // second initializeOptimization(level) over the edges of that level.  The pose graph - where the reference does that:
    // GlobalMapper::GlobalBA moves rejected feature edges to level 1 and initialises again, GlobalMapper.cpp:421-483 - supports
    // it: the vertices keep their CURRENT estimates, as g2o's do, and the device graph is rebuilt from the level-0 edges
    // (se2gpu_ba_set_edge_level).  For the landmark models it stays an error, not a silent re-run over all edges: rebuild the
    // graph after se2gpu_ba_clear, or start over with se2gpu_ba_reset_estimates.
    if (h->initialized && h->model == 2 && !h->bulk_E) {
        SE2_CHECK(ba_fetch_estimates(h));
        const size_t np = 12 * (size_t)h->P;
        SE2_REQUIRE(h->h_poses.size() == np, SE2GPU_ERR_STATE, "initialize: the pose graph's host vertices are gone");
        std::memcpy(h->h_poses.data(), h->est.p, np * sizeof(double));
        h->initialized = false;
        h->est_valid = false;
        return ba_upload_graph(h);
    }
    SE2_REQUIRE(!h->initialized, SE2GPU_ERR_STATE,
                "initialize: the graph is already initialised (se2gpu_ba_clear and rebuild it; re-optimising a subset "
                "of the edges after setLevel is supported for the pose graph only)");
    return ba_upload_graph(h);
}

// g2o::OptimizableGraph::Edge::setLevel for the EdgeSE3 edges of the pose graph (`edge` = position in the order the edges were
// added, the index se2gpu_ba_edge_chi2 reports them under): takes effect at the next se2gpu_ba_initialize.
int se2gpu_ba_set_edge_level(se2gpu_ba* h, int edge, int level) {
    SE2_REQUIRE(h && h->model == 2, SE2GPU_ERR_STATE, "set_edge_level: only the pose graph (VertexSE3 / EdgeSE3) has edge levels");
    SE2_REQUIRE(edge >= 0 && edge < (int)h->odo3.size(), SE2GPU_ERR_INVALID, "set_edge_level: no EdgeSE3 number %d", edge);
    h->odo3[edge].level = level;
    return SE2GPU_OK;
}

int se2gpu_ba_reset_estimates(se2gpu_ba* h) {
    SE2_REQUIRE(h && h->initialized, SE2GPU_ERR_STATE, "reset_estimates before initialize");
    SE2_CHECK(ba_join(h));
    h->est_valid = false;
    // one launch instead of two device-to-device copies (each costs the host about 10 us to enqueue - a caller that
    // re-optimises the same window again and again, like the bench, pays that between every two optimize() calls)
    const size_t np = (size_t)h->ps * h->P, nl = 3 * (size_t)h->L;
    hipLaunchKernelGGL(k_reset_one, grid1(std::max(np, nl), 256), dim3(256), 0, h->stream, h->poses, (const double*)h->poses0.p, np,
                       h->lms, (const double*)h->lms0.p, nl);
    SE2_HIP(hipGetLastError());
    h->own_pending = true;   // (a batch on another stream orders itself behind this launch)
    return SE2GPU_OK;
}

// se2gpu_ba_reset_estimates of `count` windows with ONE launch (on the first window's stream - the stream a lock-step
// se2gpu_ba_optimize_batch of the same windows runs on): 2 x count device-to-device copies enqueued one by one cost a mapper
// that re-optimises many windows more host time than the lock-step optimisation costs the device.  Any later operation on
// one of the windows alone is ordered behind it (ba_join).
int se2gpu_ba_reset_estimates_batch(se2gpu_ba** hs, int count) {
    SE2_REQUIRE(hs && count >= 0, SE2GPU_ERR_INVALID, "reset_estimates_batch: bad argument");
    if (count == 0) return SE2GPU_OK;
    for (int i = 0; i < count; ++i)
        SE2_REQUIRE(hs[i] && hs[i]->initialized && hs[i]->device == hs[0]->device, SE2GPU_ERR_STATE,
                    "reset_estimates_batch: window %d is not initialised (or lives on another device)", i);
    Lease<ResetScratch> lease;   // (process-wide pool, never destroyed: see LeasePool)
    ResetScratch* sc = lease.obj;
    hipStream_t st = hs[0]->stream;
    SE2_CHECK(ba_join(hs[0]));
    // the staging buffer of the previous call may still be read by its copy: the ring's event of that call says when not
    if (sc->next) {
        // (that event sits on the stream of the previous call's first window; if any handle stream has been destroyed since, it
        // may be that one - then the whole device is waited for instead: rare, a batch of windows has just been torn down)
        if (sc->deaths == streams_destroyed().load(std::memory_order_relaxed)) SE2_HIP(hipEventSynchronize(sc->ring[(sc->next - 1) % 64]));
        else SE2_HIP(hipDeviceSynchronize());
    }
    SE2_CHECK(sc->host.reserve((size_t)count));
    SE2_CHECK(sc->dev.reserve((size_t)count));
    unsigned most = 1;
    for (int i = 0; i < count; ++i) {
        se2gpu_ba* h = hs[i];
        h->est_valid = false;
        ResetItem it{h->poses, h->poses0.p, (unsigned)((size_t)h->ps * h->P), h->lms, h->lms0.p, (unsigned)(3 * (size_t)h->L)};
        sc->host.p[i] = it;
        most = std::max(most, std::max(it.np, it.nl));
        // a window whose own stream still has work in flight (an optimize that has not been waited for cannot happen: the
        // optimize entry points return after the controller has posted) is ordered first
        if (i > 0 && h->stream != st && h->own_pending) {
            if (h->join_event) SE2_CHECK(ba_join(h));
            hipEvent_t& e = sc->ring[sc->next++ % 64];
            if (!e) SE2_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            SE2_HIP(hipEventRecord(e, h->stream));
            SE2_HIP(hipStreamWaitEvent(st, e, 0));
        }
        h->own_pending = false;
    }
    SE2_HIP(hipMemcpyAsync(sc->dev.p, sc->host.p, (size_t)count * sizeof(ResetItem), hipMemcpyHostToDevice, st));
    const unsigned gx = std::min(64u, (most + 255) / 256);
    hipLaunchKernelGGL(k_reset_batch, dim3(gx, (unsigned)count), dim3(256), 0, st, (const ResetItem*)sc->dev.p);
    SE2_HIP(hipGetLastError());
    hipEvent_t& e = sc->ring[sc->next++ % 64];
    if (!e) SE2_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    SE2_HIP(hipEventRecord(e, st));
    sc->deaths = streams_destroyed().load(std::memory_order_relaxed);
    for (int i = 0; i < count; ++i) {
        hs[i]->join_event = e;
        hs[i]->join_stream = st;
    }
    return SE2GPU_OK;
}

size_t se2gpu_ba_reduce_buffer_doubles(se2gpu_ba*, int P) {
    const size_t n = 3 * (size_t)P;
    const size_t ld = ((n + 1 + kNB - 1) / kNB) * kNB;
    return ld * ld + 4;
}

size_t se2gpu_ba_exchange_doubles(int P) { return tri_row_off(3 * P + 1); }
size_t se2gpu_ba_exchange_doubles_h(const se2gpu_ba* h) { return (h && h->initialized) ? tri_row_off(h->nsys + 1) : 0; }

int se2gpu_ba_exchange_row(int row, size_t* offset, int* length) {
    SE2_REQUIRE(row >= 0 && offset && length, SE2GPU_ERR_INVALID, "exchange_row: bad argument");
    *offset = tri_row_off(row);
    *length = kNB * (row / kNB + 1);
    return SE2GPU_OK;
}

int se2gpu_ba_set_allreduce(se2gpu_ba* h, se2gpu_allreduce_fn fn, void* user, void* buffer) {
    SE2_REQUIRE(h, SE2GPU_ERR_INVALID, "ba handle is NULL");
    SE2_REQUIRE(!h->initialized, SE2GPU_ERR_STATE, "set_allreduce must precede initialize");
    h->allreduce = fn;
    h->ar_user = user;
    h->ar_buffer = buffer;
    return SE2GPU_OK;
}

static int ba_comm_trampoline(void* dev_ptr, size_t count, void* stream, void* user) {
    return comm_allreduce((se2gpu_comm*)user, dev_ptr, count, stream);
}

int se2gpu_ba_set_comm(se2gpu_ba* h, se2gpu_comm* c) {
    SE2_REQUIRE(h && c, SE2GPU_ERR_INVALID, "ba_set_comm: NULL argument");
    SE2_REQUIRE(!h->initialized, SE2GPU_ERR_STATE, "set_comm must precede initialize");
    h->comm = c;
    h->rank = comm_rank(c);
    h->world = comm_world(c);
    h->root = h->rank == 0 ? 1 : 0;
    h->allreduce = ba_comm_trampoline;
    h->ar_user = c;
    h->ar_buffer = nullptr;
    return SE2GPU_OK;
}

// landmark shard `rank` of `world`; rank 0 owns the odometry edges and the lambda*I / identity terms
int se2gpu_ba_set_shard(se2gpu_ba* h, int rank, int world) {
    SE2_REQUIRE(h, SE2GPU_ERR_INVALID, "ba handle is NULL");
    SE2_REQUIRE(!h->initialized, SE2GPU_ERR_STATE, "set_shard must precede initialize");
    SE2_REQUIRE(world >= 1 && rank >= 0 && rank < world, SE2GPU_ERR_INVALID, "bad rank/world %d/%d", rank, world);
    h->rank = rank;
    h->world = world;
    h->root = rank == 0 ? 1 : 0;
    return SE2GPU_OK;
}

double se2gpu_ba_chi2(se2gpu_ba* h) {
    if (!h || !h->initialized) {
        set_error("chi2 before initialize");
        return -1.0;
    }
    if (h->model) {   // the SE3 kernels are controller-driven: a run of zero iterations evaluates the starting state
        se2gpu_ba_stats st;
        if (se2gpu_ba_optimize(h, 0, SE2GPU_BA_LM, nullptr, 0, &st) != SE2GPU_OK) return -1.0;
        return st.chi2_init;
    }
    if (ba_evaluate(h, nullptr, 0.0) != SE2GPU_OK) return -1.0;
    return h->h_scal.p[0];
}

int se2gpu_ba_debug_reduced_system(se2gpu_ba* h, double lambda, double* S, double* bs) {
    SE2_REQUIRE(h && h->initialized, SE2GPU_ERR_STATE, "debug_reduced_system before initialize");
    SE2_CHECK(ba_join(h));
    const int n = h->D * h->P;
    SE2_CHECK(ba_linearize(h, lambda));
    SE2_CHECK(ba_reduce(h, lambda, false));
    // (nsys > D * P when the poses were re-ordered into padded partitions: the rhs row is row nsys, not D * P)
    SE2_CHECK(ba_allreduce(h, h->red, (size_t)(h->nsys + 1) * h->ld));
    SE2_HIP(hipStreamSynchronize(h->stream));
    if (!h->h_pose_off.empty()) {   // the solver's fill-reducing order: gather the system back into pose order
        const int ns = h->nsys, ld = h->ld, D = h->D;
        std::vector<double> full((size_t)(ns + 1) * ld);
        SE2_HIP(hipMemcpy(full.data(), h->red, full.size() * 8, hipMemcpyDeviceToHost));
        for (int a = 0; a < h->P; ++a)
            for (int r = 0; r < D; ++r) {
                const int ra = h->h_pose_off[a] + r;
                if (bs) bs[D * a + r] = full[(size_t)ns * ld + ra];
                if (S)
                    for (int b = 0; b < h->P; ++b)
                        for (int c = 0; c < D; ++c) S[(size_t)(D * a + r) * n + D * b + c] = full[(size_t)ra * ld + h->h_pose_off[b] + c];
            }
        return SE2GPU_OK;
    }
    if (S)
        SE2_HIP(hipMemcpy2D(S, (size_t)n * 8, h->red, (size_t)h->ld * 8, (size_t)n * 8, n, hipMemcpyDeviceToHost));
    if (bs) SE2_HIP(hipMemcpy(bs, h->red + (size_t)n * h->ld, (size_t)n * 8, hipMemcpyDeviceToHost));
    return SE2GPU_OK;
}

// The plan of the dense pose solve for a P x P block pattern (row-major bytes, != 0 where two poses share a landmark or an
// odometry edge; NULL = dense), without a device: tests/test_solve_plan.py runs the tile algorithm of k_chol_tiles in numpy
// from these lists.  Arrays may be NULL (sizes only); tasks: {tile row | kind << 16, block column, first dep, end dep}.
// (the same with the tile size as an argument: 32 = what k_chol_tiles runs on, 64 = the wide block column of docs/history/DESIGN_rounds_1-5.md 8.1)
int se2gpu_ba_debug_solve_plan_tile(int P, int D, const uint8_t* pattern, int allow_nd, int tile, int* nsys, int* nbc, int* depth, int* ntask,
                                    int* ndep, int32_t* pose_off, int32_t* tasks4, int task_cap, int32_t* deps, int dep_cap) {
    SE2_REQUIRE(P > 0 && (D == 3 || D == 6) && (tile == 32 || tile == 64) && nsys && nbc && depth && ntask && ndep, SE2GPU_ERR_INVALID, "debug_solve_plan_tile: bad argument");
    SolvePlan sp;
    solve_plan_choose(P, D, pattern, allow_nd != 0, sp, false, tile);
    *nsys = sp.nsys; *nbc = sp.nbc; *depth = sp.depth; *ntask = (int)sp.tasks.size(); *ndep = (int)sp.deps.size();
    if (pose_off) std::memcpy(pose_off, sp.pose_off.data(), (size_t)P * 4);
    if (tasks4) {
        SE2_REQUIRE(task_cap >= (int)sp.tasks.size(), SE2GPU_ERR_CAPACITY, "debug_solve_plan_tile: %zu tasks", sp.tasks.size());
        std::memcpy(tasks4, sp.tasks.data(), sp.tasks.size() * sizeof(int4));
    }
    if (deps) {
        SE2_REQUIRE(dep_cap >= (int)sp.deps.size(), SE2GPU_ERR_CAPACITY, "debug_solve_plan_tile: %zu dependency entries", sp.deps.size());
        std::memcpy(deps, sp.deps.data(), sp.deps.size() * 4);
    }
    return SE2GPU_OK;
}

int se2gpu_ba_debug_solve_plan(int P, int D, const uint8_t* pattern, int allow_nd, int* nsys, int* nbc, int* depth, int* ntask,
                               int* ndep, int32_t* pose_off, int32_t* tasks4, int task_cap, int32_t* deps, int dep_cap) {
    SE2_REQUIRE(P > 0 && (D == 3 || D == 6) && nsys && nbc && depth && ntask && ndep, SE2GPU_ERR_INVALID, "debug_solve_plan: bad argument");
    SolvePlan sp;
    solve_plan_choose(P, D, pattern, allow_nd != 0, sp);
    *nsys = sp.nsys; *nbc = sp.nbc; *depth = sp.depth; *ntask = (int)sp.tasks.size(); *ndep = (int)sp.deps.size();
    if (pose_off) std::memcpy(pose_off, sp.pose_off.data(), (size_t)P * 4);
    if (tasks4) {
        SE2_REQUIRE(task_cap >= (int)sp.tasks.size(), SE2GPU_ERR_CAPACITY, "debug_solve_plan: %zu tasks", sp.tasks.size());
        std::memcpy(tasks4, sp.tasks.data(), sp.tasks.size() * sizeof(int4));
    }
    if (deps) {
        SE2_REQUIRE(dep_cap >= (int)sp.deps.size(), SE2GPU_ERR_CAPACITY, "debug_solve_plan: %zu dependency entries", sp.deps.size());
        std::memcpy(deps, sp.deps.data(), sp.deps.size() * 4);
    }
    return SE2GPU_OK;
}

// SE2GPU_BA_CHOL_VERIFY=1: {mismatches, half-slabs checked} and up to `cap` records of 8 words (see d_chol_tiles); returns
// SE2GPU_ERR_STATE when the handle does not run in verify mode
int se2gpu_ba_debug_chol_verify(se2gpu_ba* h, unsigned long long* counts2, unsigned long long* records, int cap) {
    SE2_REQUIRE(h && h->initialized && h->chol_vfy.p, SE2GPU_ERR_STATE, "the handle does not verify its hand-offs (SE2GPU_BA_CHOL_VERIFY=1)");
    SE2_CHECK(ba_join(h));
    SE2_HIP(hipStreamSynchronize(h->stream));
    unsigned long long head[8];
    SE2_HIP(hipMemcpy(head, h->chol_vfy.p, sizeof(head), hipMemcpyDeviceToHost));
    counts2[0] = head[0]; counts2[1] = head[1];
    const int nrec = (int)std::min<unsigned long long>(std::min<unsigned long long>(head[0], kVfyRecords), (unsigned long long)std::max(cap, 0));
    if (nrec && records) SE2_HIP(hipMemcpy(records, h->chol_vfy.p + 8, (size_t)nrec * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return SE2GPU_OK;
}

// idle sets in the two lease pools (plan caches of the lock-step driver, staging of the batched reset): tests check that
// short-lived calling threads do not grow them
int se2gpu_ba_debug_pool_sizes(int out2[2]) {
    SE2_REQUIRE(out2, SE2GPU_ERR_INVALID, "debug_pool_sizes: NULL argument");
    { LeasePool<PlanSet>& p = LeasePool<PlanSet>::get(); std::lock_guard<std::mutex> g(p.mu); out2[0] = (int)p.idle.size(); }
    { LeasePool<ResetScratch>& p = LeasePool<ResetScratch>::get(); std::lock_guard<std::mutex> g(p.mu); out2[1] = (int)p.idle.size(); }
    return SE2GPU_OK;
}

int se2gpu_ba_debug_solver_path(const se2gpu_ba* h) {
    if (!h || !h->initialized) return -1;
    return h->host_solve ? 3 : h->chol_fallback ? 2 : h->chol_steps ? 1 : 0;
}

int se2gpu_ba_debug_solve(se2gpu_ba* h, double lambda, double* x, int* factor_ok) {
    SE2_REQUIRE(h && h->initialized && x, SE2GPU_ERR_STATE, "debug_solve before initialize");
    SE2_CHECK(ba_join(h));
    const int n = h->D * h->P;
    SE2_CHECK(ba_linearize(h, lambda));
    SE2_CHECK(ba_reduce(h, lambda, false));
    SE2_CHECK(ba_allreduce(h, h->red, (size_t)(h->nsys + 1) * h->ld));   // all rows of the (possibly re-ordered, padded) system
    SE2_CHECK(ba_solve(h));
    SE2_HIP(hipStreamSynchronize(h->stream));
    SE2_HIP(hipMemcpy(x, h->xp.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    double f = 0;
    SE2_HIP(hipMemcpy(&f, h->red + (size_t)h->ld * h->ld + 2, 8, hipMemcpyDeviceToHost));
    SE2_REQUIRE(f < 1e5, SE2GPU_ERR_HIP, "k_chol_tiles: a dependency spin timed out (2 s)");
    if (factor_ok) *factor_ok = !(f > 0.0);
    if (h->chol_trace.p && !h->chol_steps) {  // task, tile row, kind, column, last dependency (-1: none), then stamps relative to the first in 10 ns ticks
        std::vector<long long> tr(16 * (size_t)h->chol_ntask);
        std::vector<int4> tk(h->chol_ntask);
        SE2_HIP(hipMemcpy(tr.data(), h->chol_trace.p, tr.size() * sizeof(long long), hipMemcpyDeviceToHost));
        SE2_HIP(hipMemcpy(tk.data(), h->chol_tasks.p, tk.size() * sizeof(int4), hipMemcpyDeviceToHost));
        int ndep = 0;
        for (const int4& t : tk) ndep = std::max(ndep, t.w);
        std::vector<int> dp((size_t)std::max(ndep, 1));
        if (ndep) SE2_HIP(hipMemcpy(dp.data(), h->chol_deps.p, (size_t)ndep * sizeof(int), hipMemcpyDeviceToHost));
        long long t0 = tr[0];
        for (int t = 0; t < h->chol_ntask; ++t) t0 = std::min(t0, tr[16 * (size_t)t]);
        for (int t = 0; t < h->chol_ntask; ++t) {
            std::fprintf(stderr, "choltrace %d %d %d %d %d", t, tk[t].x & 0xffff, tk[t].x >> 16, tk[t].y,
                         tk[t].w > tk[t].z ? dp[tk[t].w - 1] : -1);
            for (int q = 0; q < 6; ++q) std::fprintf(stderr, " %lld", tr[16 * (size_t)t + q] ? tr[16 * (size_t)t + q] - t0 : -1);
            std::fprintf(stderr, " %lld %lld", tr[16 * (size_t)t + 6], tr[16 * (size_t)t + 7] ? tr[16 * (size_t)t + 7] - t0 : -1);
            for (int q = 8; q < 16; ++q) std::fprintf(stderr, " %lld", tr[16 * (size_t)t + q] ? tr[16 * (size_t)t + q] - t0 : -1);
            std::fprintf(stderr, "\n");
        }
    }
    return SE2GPU_OK;
}

// ---- optimize(): the LM / GN loop runs on the device (BaCtl); the host enqueues trial slots and waits for the posted
// controller block.  Three phases so that several handles can be driven at once (se2gpu_ba_optimize_batch):
//   ba_run_begin   resets the controller and enqueues `iters` slots (the common case: no rejected trial)
//   ba_run_step    is the posted block there?  not finished -> enqueue the missing slots; returns 1 when the run is over
//   ba_run_finish  stats, estimate pointers
// Synchronous mode (profiling, verbose, SE2GPU_BA_SYNC=1, host solve, PreEdgeSE2 edges outside the block plan) reads the
// block back after every trial and only launches the kernels that trial needs.
namespace {

bool ba_env_sync() {
    static const bool on = [] { const char* e = getenv("SE2GPU_BA_SYNC"); return e && e[0] == '1'; }();
    return on;
}

const BaCtl* ba_posted(se2gpu_ba* h) { return reinterpret_cast<const BaCtl*>(h->h_mail + 8); }

int ba_run_begin(se2gpu_ba* h, int iters, int mode, const volatile uint8_t* stop_flag, int verbose) {
    SE2_REQUIRE(h && h->initialized, SE2GPU_ERR_STATE, "optimize before initialize");
    SE2_REQUIRE(mode == SE2GPU_BA_LM || mode == SE2GPU_BA_GN, SE2GPU_ERR_INVALID, "unknown mode %d", mode);
    SE2_REQUIRE(h->d_mail, SE2GPU_ERR_STATE, "SE2GPU_BA_MAILBOX=0 is no longer supported: the LM controller posts its state "
                                              "through the mapped mailbox");
    SE2_CHECK(ba_join(h));
    h->est_valid = false;
    h->run_mode = mode;
    h->run_iters = iters;
    h->run_enqueued = 0;
    h->run_sync = ba_env_sync() || h->prof.enabled || verbose || h->odo_fallback || h->host_solve;
    h->run_active = true;
    *h->h_stop = (stop_flag && *stop_flag) ? 1 : 0;
    const int n0 = h->run_sync ? 1 : std::max(iters, 1);
    auto enqueue_all = [&]() -> int {
        hipLaunchKernelGGL(k_ctl_init, dim3(1), dim3(64), 0, h->stream, h->ctl.p, iters, mode);
        SE2_HIP(hipGetLastError());
        for (int k = 0; k < n0; ++k) SE2_CHECK(ba_enqueue_trial(h, k == 0, h->run_sync ? 0 : -1, k == n0 - 1, 0.0));
        return SE2GPU_OK;
    };
    static const bool graphs_on = [] { const char* e = getenv("SE2GPU_BA_GRAPH"); return !(e && e[0] == '0'); }();
    // (a run of zero iterations is an evaluation - se2gpu_ba_chi2 of the SE3 models: one slot, never worth a graph and never
    // allowed to push a real shape out of the cache)
    const bool graphable = graphs_on && !h->run_sync && !h->allreduce && !h->comm && iters > 0;
    se2gpu_ba::GraphSlot* slot = nullptr;
    if (graphable) {
        for (auto& g : h->graphs)
            if (g.iters == iters && g.mode == mode) slot = &g;
        if (!slot) {   // first run of this shape: remember it (least recently used slot), enqueue directly
            slot = &h->graphs[0];
            for (auto& g : h->graphs)
                if (g.stamp < slot->stamp) slot = &g;
            if (slot->exec) (void)hipGraphExecDestroy(slot->exec);
            *slot = se2gpu_ba::GraphSlot{};
            slot->iters = iters;
            slot->mode = mode;
            slot->stamp = ++h->graph_clock;
            SE2_CHECK(enqueue_all());
        } else if (slot->exec) {
            slot->stamp = ++h->graph_clock;
            SE2_HIP(hipGraphLaunch(slot->exec, h->stream));
        } else {
            // second run of this shape: capture it (the capture itself does not execute anything), then launch
            slot->stamp = ++h->graph_clock;
            hipGraph_t g = nullptr;
            SE2_HIP(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
            const int rc = enqueue_all();
            const hipError_t ce = hipStreamEndCapture(h->stream, &g);
            SE2_CHECK(rc);
            SE2_HIP(ce);
            const hipError_t ie = hipGraphInstantiate(&slot->exec, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            SE2_HIP(ie);
            SE2_HIP(hipGraphLaunch(slot->exec, h->stream));
        }
    } else {
        SE2_CHECK(enqueue_all());
    }
    h->dev_seq += n0;
    h->run_seq = h->dev_seq;
    h->run_enqueued = n0;
    return SE2GPU_OK;
}

// returns 1 in *finished when the run is over; wait = block until the pending notification has arrived
int ba_run_step(se2gpu_ba* h, bool wait, const volatile uint8_t* stop_flag, int verbose, int* finished) {
    *finished = 0;
    if (stop_flag && *stop_flag) *h->h_stop = 1;
    if (!wait && ((volatile double*)h->h_mail)[kMailSeq] != h->run_seq) {
        // an earlier slot may already have posted "done" (Terminate, stop flag): its sequence number is lower
        const double got = ((volatile double*)h->h_mail)[kMailSeq];
        if (!(got > h->run_seq - h->run_enqueued && got <= h->run_seq && ba_posted(h)->done)) return SE2GPU_OK;
    } else if (wait) {
        volatile double* mb = h->h_mail;
        const auto t0 = std::chrono::steady_clock::now();
        long spins = 0;
        for (;;) {
            const double got = mb[kMailSeq];
            if (got == h->run_seq) break;
            if (got > h->run_seq - h->run_enqueued && got < h->run_seq) {   // an earlier slot of this run posted: done?
                std::atomic_thread_fence(std::memory_order_acquire);
                if (ba_posted(h)->done) break;
            }
            __builtin_ia32_pause();
            if (stop_flag && *stop_flag) *h->h_stop = 1;
            if ((++spins & 0xfffff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) {
                SE2_HIP(hipStreamSynchronize(h->stream));
                SE2_REQUIRE(mb[kMailSeq] == h->run_seq || ba_posted(h)->done, SE2GPU_ERR_HIP, "the LM controller never reported back");
                break;
            }
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    BaCtl c;
    std::memcpy(&c, (const void*)(h->h_mail + 8), sizeof(BaCtl));
    if (c.error && !h->chol_steps) {
        // A dependency spin of k_chol_tiles timed out (2 s).  Since round 6 the tasks are drawn from a counter in topological
        // order, so this cannot follow from the order of dispatch any more; it is what SE2GPU_BA_CHOL_FAULT=1 injects (a launch
        // without its first task).  The handle then falls back to one launch per block column (k_chol_step) for the rest of its life.  The failed trial changed
        // nothing (lm_advance returns before touching the state), so it is simply redone.
        std::fprintf(stderr, "se2gpu_ba: k_chol_tiles timed out; continuing with k_chol_step\n");
        h->chol_steps = h->chol_fallback = true;
        h->drop_graphs();   // they launch k_chol_tiles
        SE2_HIP(hipStreamSynchronize(h->stream));   // the remaining slots of this run have all exited early
        hipLaunchKernelGGL(k_ctl_clear_error, dim3(1), dim3(1), 0, h->stream, h->ctl.p);
        c.error = 0;
        c.done = 0;
        h->run_enqueued = 0;                          // the posted block of the failed slot is no longer part of this run
        ((volatile double*)h->h_mail)[kMailSeq] = 0.0;
    }
    SE2_REQUIRE(!c.error, SE2GPU_ERR_HIP, "k_chol_step: the factorisation reported a time-out");
    if (verbose)
        fprintf(stderr, "se2gpu_ba: it %d trial %d chi2 %.9g rho %.3g lambda %.6g%s\n", c.it, c.qmax, c.current_chi, c.rho,
                c.lambda, c.retry ? " (rejected)" : "");
    if (h->run_sync)   // the estimate pointer follows the controller (the odometry fallback reads h->poses)
        if ((h->poses == h->poses_b.p) != (c.sel != 0)) { std::swap(h->poses, h->poses_t); std::swap(h->lms, h->lms_t); }
    if (c.done) {
        *finished = 1;
        return SE2GPU_OK;
    }
    const int more = h->run_sync ? 1 : std::max(1, c.iters - c.it);
    for (int k = 0; k < more; ++k) {
        SE2_CHECK(ba_enqueue_trial(h, false, h->run_sync ? (c.retry ? 1 : 0) : -1, k == more - 1, 0.0));
        h->dev_seq += 1;
        h->run_seq = h->dev_seq;
        ++h->run_enqueued;
    }
    return SE2GPU_OK;
}

int ba_run_finish(se2gpu_ba* h, se2gpu_ba_stats* stats) {
    h->own_pending = false;   // (the run was enqueued behind whatever the stream held, and it has reported back)
    BaCtl c;
    std::memcpy(&c, (const void*)(h->h_mail + 8), sizeof(BaCtl));
    h->run_active = false;
    if ((h->poses == h->poses_b.p) != (c.sel != 0)) { std::swap(h->poses, h->poses_t); std::swap(h->lms, h->lms_t); }
    if (stats) {
        se2gpu_ba_stats s;
        std::memset(&s, 0, sizeof(s));
        s.iterations = c.it;
        s.trials = c.trials;
        s.terminated = c.terminated;
        s.stopped = c.stopped;
        s.chi2_init = c.chi2_init;
        s.chi2_final = c.chi2_final;
        s.lambda_final = c.lambda;
        std::memcpy(s.chi2_hist, c.chi2_hist, sizeof(s.chi2_hist));
        std::memcpy(s.lambda_hist, c.lambda_hist, sizeof(s.lambda_hist));
        std::memcpy(s.trials_hist, c.trials_hist, sizeof(s.trials_hist));
        *stats = s;
    }
    return SE2GPU_OK;
}

// ---- optimize() of `count` windows, ONE WORKGROUP PER WINDOW (csrc/ba_window.hip): every window lives in one compute unit's
// LDS for its whole optimize(iters) - no per-edge records, no launches per trial.  Taken for batches of SE2GPU_BA_RESIDENT_MIN
// windows or more (default 96: below that the multi-launch paths, which spread a window over the chip, finish a batch sooner)
// whose windows all fit (SE(2) model, one GPU, at most ~60 free key frames, no landmark with more than 64 observations).
// SE2GPU_BA_RESIDENT=0 switches the path off, =1 takes it for any batch.  *handled = 0: the caller goes on to the other paths.
struct ResidentScratch {
    PinBuf<WindowArgs> host;
    DevBuf<WindowArgs> dev;
    DevBuf<long long> stamps;
};
int ba_resident_threads(const se2gpu_ba* h, size_t* lds) {
    int nfree = 0;
    for (int p = 0; p < h->P; ++p) nfree += h->h_fixed[p] ? 0 : 1;
    for (int t : {512, 256, 128}) {
        const size_t b = ba_window_lds_bytes(h->P, nfree, t);
        if (b) { *lds = b; return t; }
    }
    return 0;
}
bool ba_resident_ok(const se2gpu_ba* h) {
    return h->initialized && h->model == 0 && !h->allreduce && !h->comm && !h->host_solve && !h->prof.enabled && h->d_mail &&
           h->L > 0 && h->P > 0 && (int)h->h_fixed.size() == h->P && h->Hpl.p && h->Hpl.cap * 8 >= (size_t)h->L * 16 + (size_t)h->E * 44 + 16 &&
           h->Dinv.p && h->Dinv.cap >= 6 * (size_t)h->L;
}
int ba_optimize_resident(se2gpu_ba** hs, int count, int iters, int mode, const volatile uint8_t* stop_flag,
                         se2gpu_ba_stats* stats, int* handled) {
    *handled = 0;
    // (read per call, not once: a test - or a mapper - can switch the path between two batches)
    const char* e_on = getenv("SE2GPU_BA_RESIDENT");
    const char* e_min = getenv("SE2GPU_BA_RESIDENT_MIN");
    const int env_on = e_on ? atoi(e_on) : -1, env_min = e_min ? atoi(e_min) : 96;
    if (env_on == 0 || count < 1 || (env_on != 1 && count < env_min) || ba_env_sync() || iters < 0) return SE2GPU_OK;
    if (mode != SE2GPU_BA_LM && mode != SE2GPU_BA_GN) return SE2GPU_OK;
    // Windows are dealt to (at most) three launches by the widest workgroup their reduced system leaves room for in LDS (512, 256 or
    // 128 threads; a 50-key-frame window takes 512, one of 60 takes 256), each launch on the stream of its first window, the heaviest
    // windows first (workgroups start in index order: the long ones must not be the tail).
    struct Item { int i, threads; size_t lds; };
    std::vector<Item> items((size_t)count);
    for (int i = 0; i < count; ++i) {
        if (!ba_resident_ok(hs[i]) || hs[i]->device != hs[0]->device) return SE2GPU_OK;
        size_t b = 0;
        const int t = ba_resident_threads(hs[i], &b);
        if (!t) return SE2GPU_OK;
        // (a window that only fits the 128-thread workgroup - 61 free key frames and up - takes twice as long here as the whole
        // batch takes on the lock-step path: a batch that holds one is left to the other paths unless this one is forced)
        if (t < 256 && env_on != 1) return SE2GPU_OK;
        items[(size_t)i] = Item{i, t, b};
    }
    for (int i = 0; i < count; ++i)
        for (int j = 0; j < i; ++j)
            if (hs[i] == hs[j]) return SE2GPU_OK;
    std::stable_sort(items.begin(), items.end(), [&](const Item& a, const Item& b) {
        if (a.threads != b.threads) return a.threads > b.threads;
        return hs[a.i]->E > hs[b.i]->E;
    });
    static std::mutex launch_mu;   // (hipFuncSetAttribute inside the launcher)
    Lease<ResidentScratch> lease;
    ResidentScratch& rs = *lease.obj;
    SE2_CHECK(rs.host.reserve((size_t)count));
    SE2_CHECK(rs.dev.reserve((size_t)count));
    static const bool trace = [] { const char* e = getenv("SE2GPU_BA_RESIDENT_TRACE"); return e && e[0] == '1'; }();
    if (trace) SE2_CHECK(rs.stamps.reserve(16 * (size_t)count));
    std::vector<hipStream_t> class_streams;
    int threads = 0;
    size_t lds = 0;
    for (int k = 0; k < count; ++k) {
        const int i = items[(size_t)k].i;
        se2gpu_ba* h = hs[i];
        if (k == 0 || items[(size_t)k].threads != items[(size_t)k - 1].threads) class_streams.push_back(h->stream);
        hipStream_t st = class_streams.back();
        h->est_valid = false;
        h->run_mode = mode;
        h->run_iters = iters;
        h->run_enqueued = 1;
        h->run_sync = false;
        h->run_active = true;
        *h->h_stop = (stop_flag && *stop_flag) ? 1 : 0;
        if (h->join_event) {   // a batched reset on another stream ...
            if (h->join_stream != st) SE2_HIP(hipStreamWaitEvent(st, h->join_event, 0));
            h->join_event = nullptr;
            h->join_stream = nullptr;
        }
        if (h->stream != st && h->own_pending) SE2_HIP(hipStreamSynchronize(h->stream));   // ... or a reset still enqueued on the window's own stream
        h->own_pending = false;
        WindowArgs& a = rs.host.p[k];
        a.cam = h->cam;
        a.P = h->P; a.L = h->L; a.E = h->E; a.O = h->O; a.iters = iters; a.mode = mode;
        a.lm_ptr = h->lm_ptr.p; a.e_kf = h->e_kf.p; a.e_uv = h->e_uv.p; a.e_info = h->e_info.p;
        a.poses_a = h->poses_a.p; a.poses_b = h->poses_b.p; a.lms_a = h->lms_a.p; a.lms_b = h->lms_b.p;
        a.fixed = h->fixed.p;
        a.o_i = h->o_i.p; a.o_j = h->o_j.p; a.o_meas = h->o_meas.p; a.o_info = h->o_info.p;
        a.ctl = h->ctl.p;
        a.mail = h->d_mail;
        a.stop = h->d_stop;
        a.desc = reinterpret_cast<int4*>(h->Hpl.p);   // (the multi-launch path's W records: 72 B per edge, idle on this path; the list + the records by class)
        a.ainv = h->Dinv.p;   // (the multi-launch path's A_l: 6 L + 1 doubles, idle on this path)
        a.debug = 0;
        a.stamps = trace ? rs.stamps.p + 16 * (size_t)k : nullptr;
    }
    {
        std::lock_guard<std::mutex> lk(launch_mu);
        size_t cs = 0;
        for (int k0 = 0; k0 < count;) {
            int k1 = k0;
            size_t need = 0;
            while (k1 < count && items[(size_t)k1].threads == items[(size_t)k0].threads) { need = std::max(need, items[(size_t)k1].lds); ++k1; }
            hipStream_t st = class_streams[cs++];
            SE2_HIP(hipMemcpyAsync(rs.dev.p + k0, rs.host.p + k0, (size_t)(k1 - k0) * sizeof(WindowArgs), hipMemcpyHostToDevice, st));
            SE2_CHECK(ba_window_launch(rs.dev.p + k0, k1 - k0, items[(size_t)k0].threads, need, st));
            if (k0 == 0) { threads = items[0].threads; lds = need; }
            k0 = k1;
        }
    }
    for (int i = 0; i < count; ++i) {
        hs[i]->dev_seq += 1;
        hs[i]->run_seq = hs[i]->dev_seq;
    }
    bool refused = false;
    for (int i = 0; i < count; ++i) {
        se2gpu_ba* h = hs[i];
        volatile double* mb = h->h_mail;
        const auto t0 = std::chrono::steady_clock::now();
        long spins = 0;
        while (mb[kMailSeq] != h->run_seq) {
            __builtin_ia32_pause();
            if (stop_flag && *stop_flag)
                for (int j = 0; j < count; ++j) *hs[j]->h_stop = 1;
            if ((++spins & 0xfffff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) {
                for (hipStream_t st : class_streams) SE2_HIP(hipStreamSynchronize(st));
                SE2_REQUIRE(mb[kMailSeq] == h->run_seq, SE2GPU_ERR_HIP, "window %d of the resident batch never reported back", i);
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        refused |= ba_posted(h)->error == 2;
    }
    for (hipStream_t st : class_streams) SE2_HIP(hipStreamSynchronize(st));   // (the argument packs are leased: nothing of this call may be in flight when they go back)
    if (trace) {
        std::vector<long long> hst(16 * (size_t)count);
        SE2_HIP(hipMemcpy(hst.data(), rs.stamps.p, hst.size() * 8, hipMemcpyDeviceToHost));
        const long long* s0 = hst.data();
        std::fprintf(stderr, "se2gpu_ba resident: window 0, last trial: build %.1f us, factorise %.1f us, back-substitute %.1f us, update %.1f us (%d threads, %zu B LDS)\n",
                     (s0[1] - s0[0]) * 0.01, (s0[2] - s0[1]) * 0.01, (s0[3] - s0[2]) * 0.01, (s0[4] - s0[3]) * 0.01, threads, lds);
        std::fprintf(stderr, "se2gpu_ba resident: window 0, prologue: list %.1f us, opening pass (records, chi2, lambda_0) %.1f us\n",
                     (s0[6] - s0[5]) * 0.01, (s0[7] - s0[6]) * 0.01);
    }
    if (refused) {
        // a window holds a landmark the kernel does not take (more than 64 observations):
        // the refused windows have not been touched; the batch is finished by the other paths, window by window
        for (int i = 0; i < count; ++i) {
            se2gpu_ba* h = hs[i];
            if (ba_posted(h)->error != 2) { SE2_CHECK(ba_run_finish(h, stats ? stats + i : nullptr)); continue; }
            hipLaunchKernelGGL(k_ctl_clear_error, dim3(1), dim3(1), 0, h->stream, h->ctl.p);
            SE2_HIP(hipGetLastError());
            SE2_CHECK(se2gpu_ba_optimize(h, iters, mode, stop_flag, 0, stats ? stats + i : nullptr));
        }
        *handled = 1;
        return SE2GPU_OK;
    }
    for (int i = 0; i < count; ++i) SE2_CHECK(ba_run_finish(hs[i], stats ? stats + i : nullptr));
    *handled = 1;
    return SE2GPU_OK;
}

// optimize() of `count` windows in lock step.  *handled = 0 when the windows are not all eligible (the caller falls back
// to one stream per window).
int ba_optimize_lockstep(se2gpu_ba** hs, int count, int iters, int mode, const volatile uint8_t* stop_flag,
                         se2gpu_ba_stats* stats, int* handled) {
    *handled = 0;
    static const bool on = [] { const char* e = getenv("SE2GPU_BA_LOCKSTEP"); return !(e && e[0] == '0'); }();
    static const bool graphs_on = [] { const char* e = getenv("SE2GPU_BA_GRAPH"); return !(e && e[0] == '0'); }();
    (void)graphs_on;
    if (!on || count < 2 || ba_env_sync() || iters < 0) return SE2GPU_OK;
    if (mode != SE2GPU_BA_LM && mode != SE2GPU_BA_GN) return SE2GPU_OK;
    for (int i = 0; i < count; ++i)
        if (!ba_lockstep_ok(hs[i]) || hs[i]->device != hs[0]->device) return SE2GPU_OK;
    for (int i = 0; i < count; ++i)
        for (int j = 0; j < i; ++j)
            if (hs[i] == hs[j]) return SE2GPU_OK;
    *handled = 1;
    // Groups: the windows are dealt to a few groups, each in lock step on a stream of its own, slot by slot in turn - while
    // one group's dataflow solves wait on their chains (latency: the chip is nearly idle) the other groups' linearisation
    // and reduction (bandwidth / issue bound) run beside them.  SE2GPU_BA_BATCH_GROUPS overrides (1 = one stream).
    static const int env_groups = [] { const char* e = getenv("SE2GPU_BA_BATCH_GROUPS"); return e ? atoi(e) : 0; }();
    int G = env_groups > 0 ? env_groups : (count >= 4 ? 2 : 1);   // (two groups pay from four windows on: 4 / 8 / 12 windows +4 / +12 / +9 %)
    if (env_groups <= 0 && count >= 6) {
        // windows of different sizes: a third group.  In lock step a slot lasts as long as its group's slowest window, and
        // a window that rejects trials holds its whole group for the extra slots; with three groups fewer windows wait for
        // any one of them.  Measured on the 64 distinct windows of the bench (30-60 key frames, six starts that reject):
        // 85.1 k -> 102.4 k LM it/s with three groups, 64.7 k with four; uniform batches lose with three (105.8 k -> 99.4 k
        // at 64 windows) and keep two.
        int pmin = hs[0]->P, pmax = hs[0]->P;
        for (int i = 1; i < count; ++i) { pmin = std::min(pmin, hs[i]->P); pmax = std::max(pmax, hs[i]->P); }
        if (4 * (pmax - pmin) > pmax) G = 3;
    }
    G = std::max(1, std::min(G, std::min(count, 4)));
    // the plans of the last batches are kept: a mapper (or the bench) that optimises the same windows again re-uses the
    // argument packs on the device (plain pointers, replaced on a miss; leased from the process-wide pool for this call)
    constexpr int kPlans = kPlanSlots;
    Lease<PlanSet> plan_lease;
    BatchPlan** cache = plan_lease.obj->cache;
    unsigned long* stamp = plan_lease.obj->stamp;
    unsigned long& clock = plan_lease.obj->clock;
    struct Group { BatchPlan* bp; se2gpu_ba** hs; int count; bool finished; int slot; };
    std::vector<Group> groups(G);
    auto acquire = [&](se2gpu_ba** ghs, int gcount, BatchPlan** out, int* slot_out, const std::vector<Group>& taken) -> int {
        for (int k = 0; k < kPlans; ++k)
            if (cache[k] && cache[k]->matches(ghs, gcount, iters, mode)) { stamp[k] = ++clock; *out = cache[k]; *slot_out = k; return SE2GPU_OK; }
        int victim = -1;
        for (int k = 0; k < kPlans; ++k) {
            bool busy = false;
            for (const Group& g : taken) busy |= (g.bp && g.bp == cache[k]);
            if (busy) continue;
            if (victim < 0 || !cache[k] || (cache[victim] && stamp[k] < stamp[victim])) victim = k;
            if (!cache[k]) break;
        }
        SE2_REQUIRE(victim >= 0, SE2GPU_ERR_STATE, "optimize_batch: no plan slot");
        // (no stream call on the evicted plan: its stream is its first window's, and that window may have been destroyed - with
        // its stream - since the plan was used; a stale hipStream_t handed to the runtime is a rare segmentation fault.  The plan
        // is idle anyway: every batch waits for the last slot of all of its windows before it returns.)
        if (cache[victim]) { delete cache[victim]; cache[victim] = nullptr; }
        cache[victim] = new BatchPlan;
        const int rc = ba_build_batch_plan(*cache[victim], ghs, gcount, iters, mode);
        if (rc != SE2GPU_OK) { delete cache[victim]; cache[victim] = nullptr; return rc; }
        stamp[victim] = ++clock;
        *out = cache[victim];
        *slot_out = victim;
        return SE2GPU_OK;
    };
    // windows of unequal size are dealt to the groups by size (largest first): a group's slot then waits for windows of
    // its own size class only
    std::vector<se2gpu_ba*> by_size;
    se2gpu_ba** ghs_all = hs;
    if (G == 3 && env_groups <= 0) {
        by_size.assign(hs, hs + count);
        std::stable_sort(by_size.begin(), by_size.end(), [](const se2gpu_ba* a, const se2gpu_ba* b) { return a->P > b->P; });
        ghs_all = by_size.data();
    }
    for (int g = 0; g < G; ++g) {
        const int b0 = (int)((long long)count * g / G), b1 = (int)((long long)count * (g + 1) / G);
        groups[g] = Group{nullptr, ghs_all + b0, b1 - b0, false, -1};
        SE2_CHECK(acquire(ghs_all + b0, b1 - b0, &groups[g].bp, &groups[g].slot, groups));
    }
    // ---- prologue of every window (ba_run_begin) and the order behind whatever was enqueued for it before
    for (Group& g : groups) {
        hipStream_t st = g.bp->stream;
        for (int i = 0; i < g.count; ++i) {
            se2gpu_ba* h = g.hs[i];
            h->est_valid = false;
            h->run_mode = mode;
            h->run_iters = iters;
            h->run_enqueued = 0;
            h->run_sync = false;
            h->run_active = true;
            *h->h_stop = (stop_flag && *stop_flag) ? 1 : 0;
            if (h->join_event) {   // a batched reset on another stream ...
                if (h->join_stream != st) SE2_HIP(hipStreamWaitEvent(st, h->join_event, 0));
                h->join_event = nullptr;
                h->join_stream = nullptr;
            }
            if (h->stream != st && h->own_pending) {   // ... or copies of se2gpu_ba_reset_estimates on the window's own stream
                SE2_HIP(hipEventRecord(g.bp->events[i], h->stream));
                SE2_HIP(hipStreamWaitEvent(st, g.bp->events[i], 0));
            }
            h->own_pending = false;
        }
        g.bp->ctl_init.launch(g.bp->arena, st);
    }
    const int n0 = std::max(iters, 1);
    for (int k = 0; k < n0; ++k)
        for (Group& g : groups) SE2_CHECK(ba_batch_slot(*g.bp, k == 0, k == n0 - 1));
    for (int i = 0; i < count; ++i) {
        hs[i]->dev_seq += n0;
        hs[i]->run_seq = hs[i]->dev_seq;
        hs[i]->run_enqueued = n0;
    }
    for (int left = G; left > 0;) {
        for (Group& g : groups) {
            if (g.finished) continue;
            hipStream_t st = g.bp->stream;
            // every window answers the notification of the round's last slot (a finished one from end_slot's early path)
            int more = 0;
            for (int i = 0; i < g.count; ++i) {
                se2gpu_ba* h = g.hs[i];
                volatile double* mb = h->h_mail;
                const auto t0 = std::chrono::steady_clock::now();
                long spins = 0;
                while (mb[kMailSeq] != h->run_seq) {
                    __builtin_ia32_pause();
                    if (stop_flag && *stop_flag)
                        for (int j = 0; j < count; ++j) *hs[j]->h_stop = 1;
                    if ((++spins & 0xfffff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) {
                        SE2_HIP(hipStreamSynchronize(st));
                        SE2_REQUIRE(mb[kMailSeq] == h->run_seq, SE2GPU_ERR_HIP, "window %d of the batch never reported back", i);
                    }
                }
                std::atomic_thread_fence(std::memory_order_acquire);
                BaCtl c;
                std::memcpy(&c, (const void*)(h->h_mail + 8), sizeof(BaCtl));
                if (c.error) {
                    // a dataflow solve timed out in this window: its group leaves the lock step and finishes window by window
                    // on the windows' own streams with the per-column solver (ba_run_step does the switch for the failed
                    // one; the others go on from where they stand - all their slots have been consumed)
                    SE2_HIP(hipStreamSynchronize(st));
                    delete cache[g.slot];
                    cache[g.slot] = nullptr;
                    g.bp = nullptr;
                    for (int fin = 0; !fin;) SE2_CHECK(ba_run_step(h, true, stop_flag, 0, &fin));
                    for (int j = 0; j < g.count; ++j)
                        if (j != i)
                            for (int fin = 0; !fin;) SE2_CHECK(ba_run_step(g.hs[j], true, stop_flag, 0, &fin));
                    more = 0;
                    break;
                }
                if (!c.done) more = std::max(more, std::max(1, c.iters - c.it));
            }
            if (!more) { g.finished = true; --left; continue; }
            for (int k = 0; k < more; ++k) SE2_CHECK(ba_batch_slot(*g.bp, false, k == more - 1));
            for (int i = 0; i < g.count; ++i) {
                g.hs[i]->dev_seq += more;
                g.hs[i]->run_seq = g.hs[i]->dev_seq;
                g.hs[i]->run_enqueued += more;
            }
        }
    }
    for (int i = 0; i < count; ++i) SE2_CHECK(ba_run_finish(hs[i], stats ? stats + i : nullptr));
    return SE2GPU_OK;
}

}  // namespace

int se2gpu_ba_optimize(se2gpu_ba* h, int iters, int mode, const volatile uint8_t* stop_flag, int verbose,
                       se2gpu_ba_stats* stats) {
    SE2_CHECK(ba_run_begin(h, iters, mode, stop_flag, verbose));
    for (int fin = 0; !fin;) SE2_CHECK(ba_run_step(h, true, stop_flag, verbose, &fin));
    return ba_run_finish(h, stats);
}

// optimize() of `count` independent windows at once (one handle each, every handle on its own stream): all runs are
// enqueued before the first wait, so the device works on them concurrently - local windows are far too small to fill
// the chip one at a time.  stats may be NULL or an array of `count`.
static int ba_optimize_group(se2gpu_ba** hs, int count, int iters, int mode, const volatile uint8_t* stop_flag,
                             se2gpu_ba_stats* stats) {
    for (int i = 0; i < count; ++i) SE2_CHECK(ba_run_begin(hs[i], iters, mode, stop_flag, 0));
    std::vector<char> fin(count, 0);
    int left = count;
    const auto t0 = std::chrono::steady_clock::now();
    long spins = 0;
    while (left) {
        for (int i = 0; i < count; ++i) {
            if (fin[i]) continue;
            int f = 0;
            SE2_CHECK(ba_run_step(hs[i], hs[i]->run_sync, stop_flag, 0, &f));
            if (f) { fin[i] = 1; --left; }
        }
        __builtin_ia32_pause();
        if ((++spins & 0xfffff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60))
            for (int i = 0; i < count; ++i)
                if (!fin[i]) { int f = 0; SE2_CHECK(ba_run_step(hs[i], true, stop_flag, 0, &f)); if (f) { fin[i] = 1; --left; } }
    }
    for (int i = 0; i < count; ++i) SE2_CHECK(ba_run_finish(hs[i], stats ? stats + i : nullptr));
    return SE2GPU_OK;
}

static thread_local int t_last_batch_path = -1;
int se2gpu_ba_last_batch_path(void) { return t_last_batch_path; }

int se2gpu_ba_optimize_batch(se2gpu_ba** hs, int count, int iters, int mode, const volatile uint8_t* stop_flag,
                             se2gpu_ba_stats* stats) {
    SE2_REQUIRE(hs && count >= 0, SE2GPU_ERR_INVALID, "optimize_batch: bad argument");
    t_last_batch_path = 0;
    for (int i = 0; i < count; ++i) SE2_REQUIRE(hs[i] && hs[i]->initialized, SE2GPU_ERR_STATE, "optimize_batch: handle %d is not initialised", i);
    // A local window is a dozen launches per LM iteration; one host thread enqueues ~0.3 M launches per second, which is
    // what bounds many small windows in flight.  The windows are therefore dealt to a few enqueue threads (each window
    // stays on one thread: a handle is not thread-safe, its stream is its own).  SE2GPU_BA_BATCH_THREADS overrides.
    {   // one workgroup per window for its whole optimize(), when the batch is large enough to fill compute units that way
        int handled = 0;
        const int rc = ba_optimize_resident(hs, count, iters, mode, stop_flag, stats, &handled);
        if (handled) t_last_batch_path = 2;
        if (handled || rc != SE2GPU_OK) return rc;
    }
    {   // one launch per stage for all windows, when every window qualifies (model 0, one GPU, dataflow solve)
        int handled = 0;
        const int rc = ba_optimize_lockstep(hs, count, iters, mode, stop_flag, stats, &handled);
        if (handled) t_last_batch_path = 1;
        if (handled || rc != SE2GPU_OK) return rc;
    }
    // Cross-stream orderings left by se2gpu_ba_reset_estimates_batch are resolved HERE, by the calling thread, before any
    // enqueue thread starts: a window's second optimize() of a shape is captured into a hipGraph, and while one thread
    // captures the stream the batched reset ran on, the runtime refuses another thread's hipStreamWaitEvent on that reset's
    // event ("dependency created on uncaptured work in another stream" - found by tools/soak_fresh.sh, round 4: 91 of 130
    // per-stream batches after a batched reset).
    for (int i = 0; i < count; ++i) SE2_CHECK(ba_join(hs[i]));
    static const int env_threads = [] { const char* e = getenv("SE2GPU_BA_BATCH_THREADS"); return e ? atoi(e) : 0; }();
    int nthr = env_threads > 0 ? env_threads : 8;
    nthr = std::max(1, std::min(nthr, count / 4));
    if (nthr <= 1) return ba_optimize_group(hs, count, iters, mode, stop_flag, stats);
    int dev = 0;
    SE2_HIP(hipGetDevice(&dev));
    std::vector<int> rc(nthr, SE2GPU_OK);
    std::vector<std::string> err(nthr);
    std::vector<std::thread> th;
    for (int t = 0; t < nthr; ++t) {
        const int b0 = (int)((long long)count * t / nthr), b1 = (int)((long long)count * (t + 1) / nthr);
        th.emplace_back([=, &rc, &err]() {
            (void)hipSetDevice(dev);
            rc[t] = ba_optimize_group(hs + b0, b1 - b0, iters, mode, stop_flag, stats ? stats + b0 : nullptr);
            if (rc[t] != SE2GPU_OK) err[t] = se2gpu_last_error();
        });
    }
    for (auto& x : th) x.join();
    for (int t = 0; t < nthr; ++t)
        if (rc[t] != SE2GPU_OK) { set_error("%s", err[t].c_str()); return rc[t]; }
    return SE2GPU_OK;
}

static int ba_fetch_estimates(se2gpu_ba* h) {
    if (h->est_valid) return SE2GPU_OK;
    SE2_CHECK(ba_join(h));
    const size_t np = (size_t)h->ps * h->P, nl = 3 * (size_t)h->L;
    SE2_CHECK(h->est.reserve(np + nl + 1));
    SE2_HIP(hipMemcpyAsync(h->est.p, h->poses, np * 8, hipMemcpyDeviceToHost, h->stream));
    if (nl) SE2_HIP(hipMemcpyAsync(h->est.p + np, h->lms, nl * 8, hipMemcpyDeviceToHost, h->stream));
    SE2_HIP(hipStreamSynchronize(h->stream));
    h->est_valid = true;
    return SE2GPU_OK;
}

int se2gpu_ba_get_all(se2gpu_ba* h, double* poses, double* lms) {
    SE2_REQUIRE(h && h->initialized, SE2GPU_ERR_STATE, "get before initialize");
    SE2_CHECK(ba_fetch_estimates(h));
    if (poses) std::memcpy(poses, h->est.p, (size_t)h->ps * h->P * 8);
    if (lms && h->L) std::memcpy(lms, h->est.p + (size_t)h->ps * h->P, 3 * (size_t)h->L * 8);
    return SE2GPU_OK;
}

int se2gpu_ba_get_se2(se2gpu_ba* h, int id, double xyt[3]) {
    SE2_REQUIRE(h && h->initialized && xyt, SE2GPU_ERR_STATE, "get_se2 before initialize");
    SE2_REQUIRE(h->model == 0, SE2GPU_ERR_STATE, "get_se2 on an SE3 graph (use se2gpu_ba_get_se3)");
    const int a = h->pose_of_id.find(id);
    SE2_REQUIRE(a >= 0, SE2GPU_ERR_INVALID, "unknown pose id %d", id);
    SE2_CHECK(ba_fetch_estimates(h));
    std::memcpy(xyt, h->est.p + 3 * (size_t)a, 24);
    return SE2GPU_OK;
}

int se2gpu_ba_get_xyz(se2gpu_ba* h, int id, double xyz[3]) {
    SE2_REQUIRE(h && h->initialized && xyz, SE2GPU_ERR_STATE, "get_xyz before initialize");
    const int a = h->lm_of_id.find(id);
    SE2_REQUIRE(a >= 0, SE2GPU_ERR_INVALID, "unknown landmark id %d", id);
    SE2_CHECK(ba_fetch_estimates(h));
    std::memcpy(xyz, h->est.p + (size_t)h->ps * h->P + 3 * (size_t)a, 24);
    return SE2GPU_OK;
}

int se2gpu_ba_shard_landmarks(int L, int E, const int32_t* e_kf, const int32_t* e_lm, int world, int32_t* owner) {
    SE2_REQUIRE(L >= 0 && E >= 0 && world >= 1 && owner && (E == 0 || (e_kf && e_lm)), SE2GPU_ERR_INVALID,
                "shard_landmarks: bad argument");
    std::vector<int64_t> first(L, std::numeric_limits<int64_t>::max());
    std::vector<int64_t> deg(L, 0);
    for (int k = 0; k < E; ++k) {
        SE2_REQUIRE(e_lm[k] >= 0 && e_lm[k] < L, SE2GPU_ERR_INVALID, "edge %d: landmark out of range", k);
        first[e_lm[k]] = std::min<int64_t>(first[e_lm[k]], e_kf[k]);
        deg[e_lm[k]]++;
    }
    std::vector<int> order(L);
    for (int l = 0; l < L; ++l) order[l] = l;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return first[a] < first[b]; });
    int64_t total = 0;
    for (int l = 0; l < L; ++l) total += deg[l];
    int64_t csum = 0;
    for (int k = 0; k < L; ++k) {
        const int l = order[k];
        csum += deg[l];
        // first chunk r with bound_r = total*(r+1)/world >= csum
        int r = 0;
        while (r < world - 1 && (total * (int64_t)(r + 1)) / world < csum) ++r;
        owner[l] = r;
    }
    return SE2GPU_OK;
}

int se2gpu_ba_profile(se2gpu_ba* h, int enable) {
    SE2_REQUIRE(h, SE2GPU_ERR_INVALID, "ba handle is NULL");
    h->prof.enabled = enable != 0;
    h->prof.reset();
    return SE2GPU_OK;
}

int se2gpu_ba_profile_get(se2gpu_ba* h, int idx, const char** name, double* ms, int64_t* launches) {
    SE2_REQUIRE(h, SE2GPU_ERR_INVALID, "ba handle is NULL");
    if (idx < 0 || idx >= (int)h->prof.slots.size()) return SE2GPU_ERR_INVALID;
    if (name) *name = h->prof.slots[idx].name;
    if (ms) *ms = h->prof.slots[idx].ms;
    if (launches) *launches = h->prof.slots[idx].launches;
    return SE2GPU_OK;
}

int se2gpu_ba_edge_information(int E, const float* lc, const float* lw, const int32_t* e_kf, const float* sigma2, int P,
                               const float* Rcw, const float* twb_xy, float fx, float xrot_info, float z_info,
                               double* info_out) {
    SE2_REQUIRE(have_device(), SE2GPU_ERR_NO_DEVICE, "no HIP device visible (libse2gpu has no CPU fallback)");
    SE2_REQUIRE(E >= 0 && P >= 0, SE2GPU_ERR_INVALID, "edge_information: negative size");
    if (E == 0) return SE2GPU_OK;
    SE2_REQUIRE(lc && lw && e_kf && sigma2 && Rcw && twb_xy && info_out && P > 0, SE2GPU_ERR_INVALID,
                "edge_information: NULL argument");
    for (int k = 0; k < E; ++k)
        SE2_REQUIRE(e_kf[k] >= 0 && e_kf[k] < P, SE2GPU_ERR_INVALID, "edge %d references key frame %d of %d", k, e_kf[k], P);
    DevBuf<float> d_lc, d_lw, d_s2, d_R, d_t;
    DevBuf<int> d_kf;
    DevBuf<double> d_out;
    hipStream_t st = nullptr;
    SE2_CHECK(d_lc.upload(lc, 3 * (size_t)E, st));
    SE2_CHECK(d_lw.upload(lw, 3 * (size_t)E, st));
    SE2_CHECK(d_kf.upload(e_kf, (size_t)E, st));
    SE2_CHECK(d_s2.upload(sigma2, (size_t)E, st));
    SE2_CHECK(d_R.upload(Rcw, 9 * (size_t)P, st));
    SE2_CHECK(d_t.upload(twb_xy, 2 * (size_t)P, st));
    SE2_CHECK(d_out.reserve(4 * (size_t)E));
    const float s_rot = (float)(1. / xrot_info), s_z = (float)(1. / z_info);
    hipLaunchKernelGGL(k_edge_information, grid1(E, 256), dim3(256), 0, st, E, d_lc.p, d_lw.p, d_kf.p, d_s2.p, d_R.p,
                       d_t.p, fx, s_rot, s_z, d_out.p, 0);
    SE2_HIP(hipGetLastError());
    SE2_HIP(hipMemcpyAsync(info_out, d_out.p, 4 * (size_t)E * sizeof(double), hipMemcpyDeviceToHost, st));
    SE2_HIP(hipStreamSynchronize(st));
    return SE2GPU_OK;
}

}  // extern "C"
