// libse2gpu - SE(2)-XYZ local bundle adjustment, ONE WORKGROUP PER WINDOW (round 6).
//
// Replaces, per window of a batch, what LocalMapper::localBA runs on one g2o::SparseOptimizer:
//   /root/reference/src/LocalMapper.cpp:259-260      optimizer.initializeOptimization(0); optimizer.optimize(Config::LOCAL_ITER)
//   /root/reference/src/EdgeSE2XYZ.cpp:61-106         per-edge residual + 2x3 / 2x3 Jacobians
//   /root/reference/include/se2lam/EdgeSE2XYZ.h:62-102 PreEdgeSE2
//   [3P g2o 20160424]  Huber, Schur complement of the landmarks, dense pose solve, Levenberg policy (restated: oracle/ba_ref.cpp)
//
// Why.  The multi-launch path (csrc/ba.hip) spreads ONE window over the chip: four launches per LM trial, per-edge records
// (W_e, Dg_e: 168 B) written by k_linearize and read back 2.2x over by k_reduce2, tile hand-offs of the dense solve through
// L2 - 15.6x the algorithmic bytes in HBM traffic, and a batch of 32-64 windows in lock step tops out at 110 k LM
// iterations/s.  A 50-key-frame window is small enough to LIVE in one compute unit: the lower triangle of the
// reduced system S (147 unknowns: 87 KB) fits the 160 KiB of LDS.  So here a workgroup owns a window for its whole
// optimize(iters): the poses, S, the right-hand sides and the solution never leave LDS, the LM controller runs in the
// workgroup, and nothing per edge is ever written to memory:
//   BUILD   the edges stream in once (44 B each), 8 lanes per landmark: residual, Jacobians, Huber weight, Hll / bl by a
//           DPP butterfly inside the group, the 3x3 factor A = G^-1 of Hll + lambda I, W_e = Hpl_e A^T in registers; the
//           pose blocks Hpp_e - W_e W_e^T and b_e - W_e zeta go to S / b_s by LDS atomics (ds_add_f64), the pair products
//           W_i W_j^T of a landmark's observations through a per-wave staging strip (each pair once: lane i takes the
//           partners i + 1 .. i + k/2 cyclically)
//   SOLVE   left-looking LL^T on 3x3 blocks in LDS with the right-hand side as an extra row (the forward substitution comes
//           with the factorisation), 8 lanes share a block's dot product; x = L^-T y by one wave, y in registers
//   UPDATE  the edges stream in a second time: the linearisation is RECOMPUTED (flops are free here, bytes are not) for the
//           back-substitution x_l = A^T (zeta - sum_e W_e^T dp_e), the trial landmark goes to the other estimate buffer,
//           robust chi^2 at the trial state, the gain denominator; then g2o's accept / reject on the controller block
// Per LM trial a window reads its edge arrays twice and writes its landmarks once: 2.6 MB where the multi-launch path moves
// 26 MB.  Sums into S are atomic, hence in no fixed order: results agree with the multi-launch path and the oracle to
// rounding (1e-12 relative on the cost), not bit for bit - the parity bar of this path is north_star's 1e-5.
//
// Landmarks with more than 8 observations take 16 lanes, with more than 16 a whole wave; more than 64 is refused
// (BaCtl::error = 2: the caller runs the window on the multi-launch path).
#include "ba_window.h"

using namespace se2gpu;
using namespace se2gpu::badev;

namespace {

constexpr int kStageDoubles = 10;    // per lane in the staging strip: W_e (9) + the column of the edge's pose (1)

// ------------------------------------------------------------------------------------------------------------------
// cross-lane moves without the LDS crossbar: DPP on the two halves of a double
// ------------------------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    const long long b = __double_as_longlong(v);
    int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
constexpr int kDppXor1 = 0xB1;          // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;          // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141;   // lane i <-> 7 - i inside every 8 lanes
constexpr int kDppMirror = 0x140;       // lane i <-> 15 - i inside every 16 lanes

// the sum over an aligned group of G lanes, in every lane of the group
template <int G>
__device__ __forceinline__ double gsum(double v) {
    v += dpp_move<kDppXor1>(v);
    v += dpp_move<kDppXor2>(v);
    v += dpp_move<kDppHalfMirror>(v);
    if (G >= 16) v += dpp_move<kDppMirror>(v);
    if (G >= 64) {
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
    }
    return v;
}
__device__ __forceinline__ double wsum(double v) { return gsum<64>(v); }
__device__ __forceinline__ double wmax(double v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v = fmax(v, __shfl_xor(v, m));
    return v;
}

__device__ __forceinline__ int tri(int r, int c) { return r * (r + 1) / 2 + c; }   // packed lower triangle, r >= c

__device__ __forceinline__ void lds_add(double* p, double v) {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// EdgeSE2XYZ (EdgeSE2XYZ.cpp:61-106) with the pose's sine / cosine at hand
template <bool JAC>
__device__ __forceinline__ void edge_se2xyz(const CamDev& cam, double px, double py, double s, double c, double lx, double ly, double lz,
                                            double u, double v, double& e0, double& e1, double* Jp, double* Jl) {
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

enum { kEval = 0, kDiag = 1, kBuild = 2, kUpdate = 3 };

// what a pass needs of the window, all in LDS except the edge arrays and the landmarks
struct Ctx {
    const WindowArgs* a;
    double* S;          // packed lower triangle of the augmented system, rows 0 .. n-1 = S, row n = b_s
    double* x;          // n: the pose step (scratch of the lambda_0 pass: the diagonal of Hpp)
    const double* cur;  // 3P: the estimate
    const double* scur; // 2P: sin, cos of its headings
    const double* trl;  // 3P: the trial state
    const double* strl; // 2P
    const int* col;     // P: first column of a pose in the system, -1 = fixed
    double* stage;      // this wave's staging strip: 64 lanes x kStageDoubles
    const double* lms;  // L x 3: the estimate's landmarks
    double* lms_trial;  // L x 3: the other buffer
    int n;
    double lambda;
};

// One landmark by an aligned group of G lanes (a lane per observation).  Returns through chi / scale / dmax the lane's
// contributions (to be summed / maximised over the workgroup by the caller).
template <int MODE, int G>
__device__ __forceinline__ void landmark(const Ctx& c, int l, bool valid, int lane, double& chi, double& scale, double& dmax) {
    const WindowArgs& a = *c.a;
    const int sub = lane & (G - 1);
    int beg = 0, k = 0;
    if (valid) {
        beg = a.lm_ptr[l];
        k = a.lm_ptr[l + 1] - beg;
    }
    const bool has = sub < k;
    const int e = has ? beg + sub : 0;
    int kf = 0;
    double u = 0, v = 0, w0 = 0, w1 = 0, w2 = 0, lx = 0, ly = 0, lz = 1;
    if (has) {
        kf = a.e_kf[e];
        u = a.e_uv[2 * (size_t)e]; v = a.e_uv[2 * (size_t)e + 1];
        w0 = a.e_info[3 * (size_t)e]; w1 = a.e_info[3 * (size_t)e + 1]; w2 = a.e_info[3 * (size_t)e + 2];
    }
    if (valid) { lx = c.lms[3 * (size_t)l]; ly = c.lms[3 * (size_t)l + 1]; lz = c.lms[3 * (size_t)l + 2]; }
    const double px = c.cur[3 * kf], py = c.cur[3 * kf + 1], ps = c.scur[2 * kf], pc = c.scur[2 * kf + 1];
    const int c0 = has ? c.col[kf] : -1;
    double e0, e1, Jp[6], Jl[6];
    if (MODE == kEval) {
        edge_se2xyz<false>(a.cam, px, py, ps, pc, lx, ly, lz, u, v, e0, e1, nullptr, nullptr);
        double r0, r1;
        huber(e0 * (w0 * e0 + w1 * e1) + e1 * (w1 * e0 + w2 * e1), a.cam.huber, r0, r1);
        if (has) chi += r0;
        return;
    }
    edge_se2xyz<true>(a.cam, px, py, ps, pc, lx, ly, lz, u, v, e0, e1, Jp, Jl);
    const double we0 = w0 * e0 + w1 * e1, we1 = w1 * e0 + w2 * e1;
    double r0, r1;
    huber(e0 * we0 + e1 * we1, a.cam.huber, r0, r1);
    const double W0 = r1 * w0, W1 = r1 * w1, W2 = r1 * w2;    // weightedOmega
    const double or0 = -r1 * we0, or1 = -r1 * we1;            // omega_r
    double WJl[6];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        WJl[q] = W0 * Jl[q] + W1 * Jl[3 + q];
        WJl[3 + q] = W1 * Jl[q] + W2 * Jl[3 + q];
    }
    double hll[6], b[3];
    hll[0] = Jl[0] * WJl[0] + Jl[3] * WJl[3];
    hll[1] = Jl[0] * WJl[1] + Jl[3] * WJl[4];
    hll[2] = Jl[0] * WJl[2] + Jl[3] * WJl[5];
    hll[3] = Jl[1] * WJl[1] + Jl[4] * WJl[4];
    hll[4] = Jl[1] * WJl[2] + Jl[4] * WJl[5];
    hll[5] = Jl[2] * WJl[2] + Jl[5] * WJl[5];
#pragma unroll
    for (int r = 0; r < 3; ++r) b[r] = Jl[r] * or0 + Jl[3 + r] * or1;
    if (!has) {   // (its arithmetic ran on a made-up edge and may hold infinities: nothing of it may reach the group's sums)
#pragma unroll
        for (int i = 0; i < 6; ++i) hll[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) b[i] = 0.0;
    }
    const bool fr = c0 >= 0;
    if (MODE == kDiag) {
        // lambda_0 = 1e-5 max diag H (computeLambdaInit): the landmark blocks' diagonals here, the free poses' by atomics
        const double h0 = gsum<G>(hll[0]), h3 = gsum<G>(hll[3]), h5 = gsum<G>(hll[5]);
        if (valid) dmax = fmax(dmax, fmax(fabs(h0), fmax(fabs(h3), fabs(h5))));
        if (fr) {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double wj0 = W0 * Jp[r] + W1 * Jp[3 + r], wj1 = W1 * Jp[r] + W2 * Jp[3 + r];
                lds_add(c.x + c0 + r, Jp[r] * wj0 + Jp[3 + r] * wj1);
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) hll[i] = gsum<G>(hll[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) b[i] = gsum<G>(b[i]);
    double A[6], zt[3];
    chol_inv3(hll, c.lambda, A);
    zt[0] = A[0] * b[0];
    zt[1] = A[1] * b[0] + A[2] * b[1];
    zt[2] = A[3] * b[0] + A[4] * b[1] + A[5] * b[2];
    // W_e = Hpl_e A^T, Hpl_e = Jp^T (Omega' Jl)  (zero for a fixed pose: constructQuadraticForm skips it)
    double Wm[9];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double h0 = fr ? Jp[r] * WJl[0] + Jp[3 + r] * WJl[3] : 0.0;
        const double h1 = fr ? Jp[r] * WJl[1] + Jp[3 + r] * WJl[4] : 0.0;
        const double h2 = fr ? Jp[r] * WJl[2] + Jp[3 + r] * WJl[5] : 0.0;
        Wm[r * 3 + 0] = h0 * A[0];
        Wm[r * 3 + 1] = h0 * A[1] + h1 * A[2];
        Wm[r * 3 + 2] = h0 * A[3] + h1 * A[4] + h2 * A[5];
    }
    double bpe[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) bpe[r] = fr ? Jp[r] * or0 + Jp[3 + r] * or1 : 0.0;

    if (MODE == kBuild) {
        if (fr) {
            double WJp[6];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                WJp[q] = W0 * Jp[q] + W1 * Jp[3 + q];
                WJp[3 + q] = W1 * Jp[q] + W2 * Jp[3 + q];
            }
            // the pose's own block: Hpp_e - W_e W_e^T (lower triangle) and its right-hand side b_e - W_e zeta
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int q = 0; q <= r; ++q) {
                    const double hpp = Jp[r] * WJp[q] + Jp[3 + r] * WJp[3 + q];
                    const double ww = Wm[r * 3] * Wm[q * 3] + Wm[r * 3 + 1] * Wm[q * 3 + 1] + Wm[r * 3 + 2] * Wm[q * 3 + 2];
                    lds_add(c.S + tri(c0 + r, c0 + q), hpp - ww);
                }
                lds_add(c.S + tri(c.n, c0 + r), bpe[r] - (Wm[r * 3] * zt[0] + Wm[r * 3 + 1] * zt[1] + Wm[r * 3 + 2] * zt[2]));
            }
        }
        // the pair products of the landmark's observations: every lane puts W_e and its column into the wave's strip, lane i then
        // takes the partners (i + s) mod k, s = 1 .. k / 2 (the pairs at distance k / 2 of an even k only from the lower half)
        double* mine = c.stage + (size_t)lane * kStageDoubles;
#pragma unroll
        for (int i = 0; i < 9; ++i) mine[i] = Wm[i];
        mine[9] = (double)c0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int half = k >> 1;
        int smax = half;
#pragma unroll
        for (int m = G; m < 64; m <<= 1) smax = max(smax, __shfl_xor(smax, m));   // the wave's longest landmark sets the trip count
        const int gbase = lane & ~(G - 1);
        for (int s = 1; s <= smax; ++s) {
            const bool act = has && s <= half && !(2 * s == k && sub >= half);
            int j = sub + s;
            if (j >= k) j -= k;
            const double* his = c.stage + (size_t)(gbase + (act ? j : sub)) * kStageDoubles;
            double Wp[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) Wp[i] = his[i];
            const int cp = (int)his[9];
            if (act && fr && cp >= 0) {
                // block (mine, his) of S loses W_mine W_his^T; it is stored where row > column
                const bool lower = c0 > cp;
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const double pr = Wm[r * 3] * Wp[q * 3] + Wm[r * 3 + 1] * Wp[q * 3 + 1] + Wm[r * 3 + 2] * Wp[q * 3 + 2];
                        lds_add(c.S + (lower ? tri(c0 + r, cp + q) : tri(cp + q, c0 + r)), -pr);
                    }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        return;
    }

    // ---- kUpdate: back-substitution, trial landmark, robust chi^2 at the trial state, the gain denominator
    double dp[3] = {0, 0, 0};
    if (fr) { dp[0] = c.x[c0]; dp[1] = c.x[c0 + 1]; dp[2] = c.x[c0 + 2]; }
    double t[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) t[q] = gsum<G>(Wm[q] * dp[0] + Wm[3 + q] * dp[1] + Wm[6 + q] * dp[2]);   // sum_e W_e^T dp_e
    const double t0 = zt[0] - t[0], t1 = zt[1] - t[1], t2 = zt[2] - t[2];
    double xl[3];
    xl[0] = A[0] * t0 + A[1] * t1 + A[3] * t2;                 // x_l = A^T (zeta - sum_e W_e^T dp_e)
    xl[1] = A[2] * t1 + A[4] * t2;
    xl[2] = A[5] * t2;
    const double nx = lx + xl[0], ny = ly + xl[1], nz = lz + xl[2];
    if (valid && sub == 0) {
        c.lms_trial[3 * (size_t)l] = nx; c.lms_trial[3 * (size_t)l + 1] = ny; c.lms_trial[3 * (size_t)l + 2] = nz;
#pragma unroll
        for (int q = 0; q < 3; ++q) scale += xl[q] * (c.lambda * xl[q] + b[q]);
    }
    if (has) {
        scale += dp[0] * bpe[0] + dp[1] * bpe[1] + dp[2] * bpe[2];   // the landmark edges' share of dp . b_p
        edge_se2xyz<false>(a.cam, c.trl[3 * kf], c.trl[3 * kf + 1], c.strl[2 * kf], c.strl[2 * kf + 1], nx, ny, nz, u, v, e0, e1, nullptr, nullptr);
        double q0, q1;
        huber(e0 * (w0 * e0 + w1 * e1) + e1 * (w1 * e0 + w2 * e1), a.cam.huber, q0, q1);
        chi += q0;
    }
}

// all landmarks of the window: 8 lanes each; those with more observations from the list the prologue made, 16 lanes or a wave each
template <int MODE, int NT>
__device__ __forceinline__ void landmark_pass(const Ctx& c, const int* big, int nbig, double& chi, double& scale, double& dmax) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = c.a->L;
    for (int l0 = 0; l0 < L; l0 += NT / 8) {
        const int l = l0 + tid / 8;
        bool valid = l < L;
        if (valid) valid = c.a->lm_ptr[l + 1] - c.a->lm_ptr[l] <= 8;
        landmark<MODE, 8>(c, l, valid, lane, chi, scale, dmax);
    }
    for (int i0 = 0; i0 < nbig; i0 += NT / 16) {
        const int i = i0 + tid / 16;
        bool valid = i < nbig;
        const int l = valid ? big[i] : 0;
        if (valid) valid = c.a->lm_ptr[l + 1] - c.a->lm_ptr[l] <= 16;
        landmark<MODE, 16>(c, l, valid, lane, chi, scale, dmax);
    }
    for (int i = wave; i < nbig; i += NT / 64) {
        const int l = big[i];
        if (c.a->lm_ptr[l + 1] - c.a->lm_ptr[l] <= 16) continue;   // (uniform over the wave)
        landmark<MODE, 64>(c, l, true, lane, chi, scale, dmax);
    }
}

// PreEdgeSE2 (EdgeSE2XYZ.h:62-102), one thread per edge
template <int MODE>
__device__ __forceinline__ void odometry_edge(const Ctx& c, int k, double& chi, double& scale) {
    const WindowArgs& a = *c.a;
    const int i = a.o_i[k], j = a.o_j[k];
    const double* W = a.o_info + 9 * (size_t)k;
    double e[3], A[9], B[9];
    if (MODE == kUpdate) {   // the gain denominator's share at the estimate's linearisation, chi^2 at the trial state
        pre_se2(c.cur + 3 * i, c.cur + 3 * j, a.o_meas + 3 * (size_t)k, e, A, B);
        const int ci = c.col[i], cj = c.col[j];
        double omr[3];
        for (int r = 0; r < 3; ++r) omr[r] = -(W[r * 3] * e[0] + W[r * 3 + 1] * e[1] + W[r * 3 + 2] * e[2]);
        for (int r = 0; r < 3; ++r) {
            if (ci >= 0) scale += c.x[ci + r] * (A[r] * omr[0] + A[3 + r] * omr[1] + A[6 + r] * omr[2]);
            if (cj >= 0) scale += c.x[cj + r] * (B[r] * omr[0] + B[3 + r] * omr[1] + B[6 + r] * omr[2]);
        }
        pre_se2(c.trl + 3 * i, c.trl + 3 * j, a.o_meas + 3 * (size_t)k, e, A, B);
        for (int r = 0; r < 3; ++r) chi += e[r] * (W[r * 3] * e[0] + W[r * 3 + 1] * e[1] + W[r * 3 + 2] * e[2]);
        return;
    }
    pre_se2(c.cur + 3 * i, c.cur + 3 * j, a.o_meas + 3 * (size_t)k, e, A, B);
    if (MODE == kEval) {
        for (int r = 0; r < 3; ++r) chi += e[r] * (W[r * 3] * e[0] + W[r * 3 + 1] * e[1] + W[r * 3 + 2] * e[2]);
        return;
    }
    const int ci = c.col[i], cj = c.col[j];
    double omr[3], WA[9], WB[9];
    for (int r = 0; r < 3; ++r) {
        omr[r] = -(W[r * 3] * e[0] + W[r * 3 + 1] * e[1] + W[r * 3 + 2] * e[2]);
        for (int q = 0; q < 3; ++q) {
            WA[r * 3 + q] = W[r * 3] * A[q] + W[r * 3 + 1] * A[3 + q] + W[r * 3 + 2] * A[6 + q];
            WB[r * 3 + q] = W[r * 3] * B[q] + W[r * 3 + 1] * B[3 + q] + W[r * 3 + 2] * B[6 + q];
        }
    }
    for (int r = 0; r < 3; ++r) {
        for (int q = 0; q < 3; ++q) {
            const double aa = A[r] * WA[q] + A[3 + r] * WA[3 + q] + A[6 + r] * WA[6 + q];
            const double ab = A[r] * WB[q] + A[3 + r] * WB[3 + q] + A[6 + r] * WB[6 + q];
            const double bb = B[r] * WB[q] + B[3 + r] * WB[3 + q] + B[6 + r] * WB[6 + q];
            if (MODE == kDiag) {
                if (q == r) {
                    if (ci >= 0) lds_add(c.x + ci + r, aa);
                    if (cj >= 0) lds_add(c.x + cj + r, bb);
                }
                continue;
            }
            if (ci >= 0 && q <= r) lds_add(c.S + tri(ci + r, ci + q), aa);
            if (cj >= 0 && q <= r) lds_add(c.S + tri(cj + r, cj + q), bb);
            if (ci >= 0 && cj >= 0) lds_add(c.S + (ci > cj ? tri(ci + r, cj + q) : tri(cj + q, ci + r)), ab);   // H(i r, j q)
        }
        if (MODE == kBuild) {
            if (ci >= 0) lds_add(c.S + tri(c.n, ci + r), A[r] * omr[0] + A[3 + r] * omr[1] + A[6 + r] * omr[2]);
            if (cj >= 0) lds_add(c.S + tri(c.n, cj + r), B[r] * omr[0] + B[3 + r] * omr[1] + B[6 + r] * omr[2]);
        }
    }
}

// workgroup sums of two values and a maximum; every thread gets the results
template <int NT>
__device__ __forceinline__ void wg_reduce(double* red, double& s0, double& s1, double& m) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    s0 = wsum(s0);
    s1 = wsum(s1);
    m = wmax(m);
    __syncthreads();
    if (lane == 0) { red[wave] = s0; red[8 + wave] = s1; red[16 + wave] = m; }
    __syncthreads();
    double a = 0, b = 0, mm = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) { a += red[w]; b += red[8 + w]; mm = fmax(mm, red[16 + w]); }   // fixed order
    s0 = a; s1 = b; m = mm;
}

// LL^T of the augmented system in place (left-looking, 3x3 blocks, the right-hand side as row n): the off-diagonal blocks of
// L overwrite S, the diagonal blocks go to dl (6 per block: l00 l10 l11 l20 l21 l22) with their reciprocals in invd.
// *fail is set when a pivot is not positive (the step is then rejected, as g2o rejects a failed Cholesky).
template <int NT>
__device__ __forceinline__ void factorize(double* S, double* dl, double* invd, int nf, int* fail) {
    const int tid = threadIdx.x, sub = tid & 7, grp = tid >> 3;
    const int n = 3 * nf;
    for (int J = 0; J < nf; ++J) {
        // ---- T(I, J) = S(I, J) - sum_{K < J} L(I, K) L(J, K)^T for every block row I >= J (the last "block" is the row of b_s)
        for (int I = J + grp; I <= nf; I += NT / 8) {
            const int rows = I < nf ? 3 : 1;
            double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int K = sub; K < J; K += 8) {
                double lj[9], li[9];
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int m = 0; m < 3; ++m) lj[q * 3 + m] = S[tri(3 * J + q, 3 * K + m)];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int m = 0; m < 3; ++m) li[r * 3 + m] = r < rows ? S[tri(3 * I + r, 3 * K + m)] : 0.0;
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int q = 0; q < 3; ++q) acc[r * 3 + q] += li[r * 3] * lj[q * 3] + li[r * 3 + 1] * lj[q * 3 + 1] + li[r * 3 + 2] * lj[q * 3 + 2];
            }
#pragma unroll
            for (int i = 0; i < 9; ++i) acc[i] = gsum<8>(acc[i]);
            if (sub == 0) {
                for (int r = 0; r < rows; ++r)
                    for (int q = 0; q < 3; ++q)
                        if (I > J || q <= r) S[tri(3 * I + r, 3 * J + q)] -= acc[r * 3 + q];
            }
        }
        __syncthreads();
        // ---- L(J, J) from T(J, J) by every group's first lane (6 multiplies: cheaper than a hand-off), L(I, J) = T(I, J) L(J, J)^-T
        if (sub == 0) {
            const double t00 = S[tri(3 * J, 3 * J)], t10 = S[tri(3 * J + 1, 3 * J)], t11 = S[tri(3 * J + 1, 3 * J + 1)];
            const double t20 = S[tri(3 * J + 2, 3 * J)], t21 = S[tri(3 * J + 2, 3 * J + 1)], t22 = S[tri(3 * J + 2, 3 * J + 2)];
            bool bad = !(t00 > 0.0);
            const double l00 = sqrt(bad ? 1.0 : t00), i00 = 1.0 / l00;
            const double l10 = t10 * i00, l20 = t20 * i00;
            const double d1 = t11 - l10 * l10;
            bad |= !(d1 > 0.0);
            const double l11 = sqrt(d1 > 0.0 ? d1 : 1.0), i11 = 1.0 / l11;
            const double l21 = (t21 - l20 * l10) * i11;
            const double d2 = t22 - l20 * l20 - l21 * l21;
            bad |= !(d2 > 0.0);
            const double l22 = sqrt(d2 > 0.0 ? d2 : 1.0), i22 = 1.0 / l22;
            for (int I = J + grp; I <= nf; I += NT / 8) {
                if (I == J) {
                    if (bad) *fail = 1;
                    dl[6 * J] = l00; dl[6 * J + 1] = l10; dl[6 * J + 2] = l11; dl[6 * J + 3] = l20; dl[6 * J + 4] = l21; dl[6 * J + 5] = l22;
                    invd[3 * J] = i00; invd[3 * J + 1] = i11; invd[3 * J + 2] = i22;
                    continue;
                }
                const int rows = I < nf ? 3 : 1;
                for (int r = 0; r < rows; ++r) {
                    double* t = S + tri(3 * I + r, 3 * J);
                    const double x0 = t[0] * i00;
                    const double x1 = (t[1] - x0 * l10) * i11;
                    const double x2 = (t[2] - x0 * l20 - x1 * l21) * i22;
                    t[0] = x0; t[1] = x1; t[2] = x2;
                }
            }
        }
        __syncthreads();
    }
    (void)n;
}

// x = L^-T y by ONE wave: y (row n of the factor) in registers, three unknowns per lane; row j of L is read once, x_j leaves by a
// scalar broadcast.  n <= 192.
__device__ __forceinline__ void back_substitute(const double* S, const double* dl, const double* invd, int nf, double* x) {
    const int lane = threadIdx.x & 63;
    const int n = 3 * nf;
    double y[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int i = lane + 64 * s;
        y[s] = i < n ? S[tri(n, i)] : 0.0;
    }
    for (int j = n - 1; j >= 0; --j) {
        const int J = j / 3, r = j - 3 * J;
        const double ys = j >= 128 ? y[2] : (j >= 64 ? y[1] : y[0]);
        const long long bits = __double_as_longlong(ys);
        const int lo = __builtin_amdgcn_readlane((int)(bits & 0xffffffffll), j & 63);
        const int hi = __builtin_amdgcn_readlane((int)(bits >> 32), j & 63);
        const double xj = __longlong_as_double(((long long)hi << 32) | (unsigned int)lo) * invd[j];
        if (lane == 0) x[j] = xj;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int i = lane + 64 * s;
            if (i < j) {
                double lji;
                if (i >= 3 * J) lji = dl[6 * J + (r == 1 ? 1 : 3 + (i - 3 * J))];   // inside the diagonal block: l10 (r = 1) or l20 / l21 (r = 2)
                else lji = S[tri(j, i)];
                y[s] -= lji * xj;
            }
        }
    }
}

template <int NT>
__global__ __launch_bounds__(NT) void k_window_lm(const WindowArgs* __restrict__ all) {
    const WindowArgs& a = all[blockIdx.x];
    extern __shared__ double lds[];
    __shared__ BaCtl ctl;
    __shared__ int s_nf, s_nbig, s_fail, s_err, s_stop;
    __shared__ int big[kWindowBigCap];
    __shared__ double red[24];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = a.P;

    // ---- prologue: the controller block (k_ctl_init's rules), columns of the free poses, the list of wide landmarks
    if (tid == 0) {
        const BaCtl* g = a.ctl;
        const int sel = g->sel;
        const double seq = g->seq;
        const unsigned epoch = g->epoch;
        double* w = reinterpret_cast<double*>(&ctl);
        for (int i = 0; i < (int)(sizeof(BaCtl) / 8); ++i) w[i] = 0.0;
        ctl.ni = 2;
        ctl.sel = sel;
        ctl.iters = a.iters;
        ctl.mode = a.mode;
        ctl.seq = seq;
        ctl.epoch = epoch;
        s_nbig = 0; s_fail = 0; s_err = 0;
        s_stop = (a.stop && *(const volatile int*)a.stop) ? 1 : 0;
    }
    // LDS map (doubles): cur 3P | trl 3P | scur 2P | strl 2P | x n | invd n | dl 2n | stage NT x kStageDoubles | S (n+1)(n+2)/2 ; col P ints first
    int* col = reinterpret_cast<int*>(lds);
    double* base = lds + (P + 1) / 2;
    double* bufA = base;
    double* bufB = bufA + 3 * P;
    double* scA = bufB + 3 * P;
    double* scB = scA + 2 * P;
    __syncthreads();
    if (tid == 0) {
        int cnt = 0;
        for (int p = 0; p < P; ++p) col[p] = a.fixed[p] ? -1 : 3 * cnt++;
        s_nf = cnt;
    }
    {
        const double* src = ctl.sel ? a.poses_b : a.poses_a;
        for (int i = tid; i < 3 * P; i += NT) bufA[i] = src[i];
    }
    for (int l = tid; l < a.L; l += NT) {
        const int k = a.lm_ptr[l + 1] - a.lm_ptr[l];
        if (k > kWindowMaxDegree) s_err = 2;
        else if (k > 8) {
            const int at = atomicAdd(&s_nbig, 1);
            if (at < kWindowBigCap) big[at] = l;
            else s_err = 2;
        }
    }
    __syncthreads();
    const int nf = s_nf, n = 3 * nf;
    double* xs = scB + 2 * P;
    double* invd = xs + n;
    double* dl = invd + n;
    double* stage_all = dl + 2 * n;
    double* S = stage_all + (size_t)NT * kStageDoubles;
    const int ntri = (n + 1) * (n + 2) / 2;
    for (int p = tid; p < P; p += NT) sincos(bufA[3 * p + 2], &scA[2 * p], &scA[2 * p + 1]);
    const int nbig = min(s_nbig, kWindowBigCap);
    const bool refused = s_err != 0;   // a landmark this kernel does not take: nothing is touched, the caller runs the window elsewhere
    __syncthreads();
    if (refused && tid == 0) { ctl.error = 2; ctl.done = 1; }
    __syncthreads();

    Ctx c;
    c.a = &a;
    c.S = S;
    c.x = xs;
    c.cur = bufA; c.scur = scA; c.trl = bufB; c.strl = scB;
    c.col = col;
    c.stage = stage_all + (size_t)wave * 64 * kStageDoubles;
    c.lms = ctl.sel ? a.lms_b : a.lms_a;
    c.lms_trial = ctl.sel ? a.lms_a : a.lms_b;
    c.n = n;
    c.lambda = 0.0;
    double* cur = bufA;
    double* trl = bufB;
    double* scur = scA;
    double* strl = scB;

    // ---- chi^2 of the starting state (computeActiveErrors + activeRobustChi2 in front of the first iteration)
    if (!refused) {
        double chi = 0, sc = 0, dm = 0;
        landmark_pass<kEval, NT>(c, big, nbig, chi, sc, dm);
        for (int k = tid; k < a.O; k += NT) odometry_edge<kEval>(c, k, chi, sc);
        wg_reduce<NT>(red, chi, sc, dm);
        if (tid == 0) {
            ctl.current_chi = ctl.chi2_init = ctl.chi2_final = chi;
            if (s_stop) { ctl.stopped = 1; ctl.done = 1; }
            if (ctl.iters <= 0) ctl.done = 1;
        }
        __syncthreads();
    }
    // ---- lambda_0 = 1e-5 max diag H of the first linearisation (computeLambdaInit); Gauss-Newton keeps lambda = 0
    if (!ctl.done && a.mode == SE2GPU_BA_LM) {
        for (int i = tid; i < n; i += NT) xs[i] = 0.0;
        __syncthreads();
        double chi = 0, sc = 0, dm = 0;
        landmark_pass<kDiag, NT>(c, big, nbig, chi, sc, dm);
        for (int k = tid; k < a.O; k += NT) odometry_edge<kDiag>(c, k, chi, sc);
        __syncthreads();
        for (int i = tid; i < n; i += NT) dm = fmax(dm, fabs(xs[i]));
        wg_reduce<NT>(red, chi, sc, dm);
        if (tid == 0) { ctl.lambda = 1e-5 * dm; ctl.ni = 2; }
        __syncthreads();
    }

    long long* stamps = a.stamps;
    // ---- the trials
    while (!ctl.done) {
        const double lambda = ctl.lambda;
        c.lambda = lambda;
        if (stamps && tid == 0) stamps[0] = wall_clock64();
        for (int i = tid; i < ntri; i += NT) S[i] = 0.0;
        if (tid == 0) s_fail = 0;
        __syncthreads();
        {
            double chi = 0, sc = 0, dm = 0;
            landmark_pass<kBuild, NT>(c, big, nbig, chi, sc, dm);
            for (int k = tid; k < a.O; k += NT) odometry_edge<kBuild>(c, k, chi, sc);
        }
        __syncthreads();
        for (int i = tid; i < n; i += NT) S[tri(i, i)] += lambda;      // setLambda: the damping on the pose diagonal (the landmarks' went into A)
        __syncthreads();
        if (stamps && tid == 0) stamps[1] = wall_clock64();
        factorize<NT>(S, dl, invd, nf, &s_fail);
        if (stamps && tid == 0) stamps[2] = wall_clock64();
        if (wave == 0) back_substitute(S, dl, invd, nf, xs);
        __syncthreads();
        if (stamps && tid == 0) stamps[3] = wall_clock64();
        // ---- oplus into the trial state (VertexSE2::oplusImpl: additive x, y; normalised heading)
        double chi = 0, sc = 0, dm = 0;
        for (int p = tid; p < P; p += NT) {
            double px = cur[3 * p], py = cur[3 * p + 1], th = cur[3 * p + 2];
            const int cp = col[p];
            if (cp >= 0) {
                const double d0 = xs[cp], d1 = xs[cp + 1], d2 = xs[cp + 2];
                px += d0; py += d1; th = normalize_theta(th + d2);
                sc += lambda * (d0 * d0 + d1 * d1 + d2 * d2);           // the damping's share of x^T (lambda x + b)
            }
            trl[3 * p] = px; trl[3 * p + 1] = py; trl[3 * p + 2] = th;
            sincos(th, &strl[2 * p], &strl[2 * p + 1]);
        }
        __syncthreads();
        landmark_pass<kUpdate, NT>(c, big, nbig, chi, sc, dm);
        for (int k = tid; k < a.O; k += NT) odometry_edge<kUpdate>(c, k, chi, sc);
        wg_reduce<NT>(red, chi, sc, dm);
        if (stamps && tid == 0) stamps[4] = wall_clock64();
        if (tid == 0) {
            const int stopped = (a.stop && *(const volatile int*)a.stop) ? 1 : 0;
            const int sel_before = ctl.sel;
            const double v[3] = {chi, sc, s_fail ? 1.0 : 0.0};
            lm_advance(&ctl, v, stopped != 0);
            s_stop = ctl.sel != sel_before;     // (re-used: the trial state became the estimate)
        }
        __syncthreads();
        if (s_stop) {
            double* t = cur; cur = trl; trl = t;
            t = scur; scur = strl; strl = t;
            c.cur = cur; c.scur = scur; c.trl = trl; c.strl = strl;
            const double* tl = c.lms; c.lms = c.lms_trial; c.lms_trial = const_cast<double*>(tl);
        }
        __syncthreads();
    }

    // ---- epilogue: the estimate's poses to the buffer the controller names, the block to the handle and its mailbox
    if (!refused) {
        double* dst = ctl.sel ? a.poses_b : a.poses_a;
        for (int i = tid; i < 3 * P; i += NT) dst[i] = cur[i];
    }
    __syncthreads();
    if (tid == 0) ctl.seq += 1.0;
    __syncthreads();
    {
        constexpr int kWords = (int)(sizeof(BaCtl) / 8);
        const double* src = reinterpret_cast<const double*>(&ctl);
        double* gdst = reinterpret_cast<double*>(a.ctl);
        for (int i = tid; i < kWords; i += NT) gdst[i] = src[i];
        if (a.mail) {
            volatile double* mail = a.mail;
            for (int i = tid; i < kWords; i += NT) mail[8 + i] = src[i];
            __threadfence_system();
            __syncthreads();
            if (tid == 0) mail[kMailSeq] = ctl.seq;
        }
    }
}

}  // namespace

namespace se2gpu {

size_t ba_window_lds_bytes(int P, int nfree, int threads) {
    const size_t n = 3 * (size_t)nfree;
    size_t doubles = (size_t)(P + 1) / 2 + 10 * (size_t)P + 4 * n + (size_t)threads * kStageDoubles + (n + 1) * (n + 2) / 2;
    const size_t bytes = doubles * 8;
    // static LDS of the kernel: the controller block, the list of wide landmarks, the reduction scratch
    const size_t fixed = sizeof(BaCtl) + kWindowBigCap * sizeof(int) + 24 * 8 + 128;
    if (n > 192 || bytes + fixed > 160 * 1024) return 0;
    return bytes;
}

template <int NT>
static int launch_nt(const WindowArgs* d_args, int count, size_t lds_bytes, hipStream_t st) {
    static size_t allowed = 0;   // (grown under the caller's lock: se2gpu_ba_optimize_batch serialises its resident launches)
    if (lds_bytes > allowed) {
        SE2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_window_lm<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        allowed = lds_bytes;
    }
    hipLaunchKernelGGL(k_window_lm<NT>, dim3(count), dim3(NT), lds_bytes, st, d_args);
    SE2_HIP(hipGetLastError());
    return SE2GPU_OK;
}

int ba_window_launch(const WindowArgs* d_args, int count, int threads, size_t lds_bytes, hipStream_t st) {
    if (count <= 0) return SE2GPU_OK;
    if (threads == 512) return launch_nt<512>(d_args, count, lds_bytes, st);
    if (threads == 256) return launch_nt<256>(d_args, count, lds_bytes, st);
    if (threads == 128) return launch_nt<128>(d_args, count, lds_bytes, st);
    set_error("window kernel: 128, 256 or 512 threads");
    return SE2GPU_ERR_INVALID;
}

}  // namespace se2gpu
