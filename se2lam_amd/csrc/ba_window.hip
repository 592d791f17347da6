// libse2gpu - SE(2)-XYZ local bundle adjustment, ONE WORKGROUP PER WINDOW (round 6).
//
// Replaces, per window of a batch, what LocalMapper::localBA runs on one g2o::SparseOptimizer:
//   /root/reference/src/LocalMapper.cpp:259-260      optimizer.initializeOptimization(0); optimizer.optimize(Config::LOCAL_ITER)
//   /root/reference/src/EdgeSE2XYZ.cpp:61-106         per-edge residual + 2x3 / 2x3 Jacobians
//   /root/reference/include/se2lam/EdgeSE2XYZ.h:62-102 PreEdgeSE2
//   [3P g2o 20160424]  Huber, Schur complement of the landmarks, dense pose solve, Levenberg policy (as the multi-launch path restates them)
//
// Why.  The multi-launch path (csrc/ba.hip) spreads ONE window over the chip: four launches per LM trial, per-edge records
// (W_e, Dg_e: 168 B) written by k_linearize and read back 2.2x over by k_reduce2, tile hand-offs of the dense solve through
// L2 - 15.6x the algorithmic bytes in HBM traffic, and a batch of 32-64 windows in lock step tops out at 110 k LM
// iterations/s.  A 50-key-frame window is small enough to LIVE in one compute unit: the lower triangle of the
// reduced system S (147 unknowns: 87 KB) fits the 160 KiB of LDS.  So here a workgroup owns a window for its whole
// optimize(iters): the poses, S, the right-hand sides and the solution never leave LDS, the LM controller runs in the
// workgroup, and nothing per edge and trial is written to memory:
//   OPEN    (once per optimize) the landmarks are listed by observation count, and one pass gives chi^2 of the starting state,
//           the diagonal for lambda_0 and a copy of the observations in the order of that list (whole cache lines from then on)
//   BUILD   the observations stream in once (44 B each), one lane each, 4 / 8 / 16 / 64 lanes per landmark by its count: residual,
//           Jacobians, Huber weight, Hll / bl by a DPP butterfly inside the group, the 3x3 factor A = G^-1 of Hll + lambda I,
//           W_e = Hpl_e A^T in registers; the pose blocks Hpp_e - W_e W_e^T and b_e - W_e zeta go to S / b_s by LDS atomics
//           (ds_add_f64), the pair products W_i W_j^T of a landmark's observations through a per-wave staging strip (each pair
//           once: lane i takes the partners i + 1 .. i + k/2 cyclically); A (48 B per landmark) is left in memory for UPDATE
//   SOLVE   left-looking LL^T on 3x3 blocks in LDS with the right-hand side as an extra row (the forward substitution comes
//           with the factorisation), 8-64 lanes share a block's dot product; x = L^-T y by one wave, y in registers
//   UPDATE  the observations stream in a second time: the Jacobians are RECOMPUTED (nothing per edge is kept) for the
//           back-substitution x_l = A^T (A (bl - sum_e Hlp_e dp_e)), the trial landmark goes to the other estimate buffer,
//           robust chi^2 at the trial state, the gain denominator; then g2o's accept / reject on the controller block
// Per LM trial a window reads its observations twice, writes and reads A once and writes its landmarks once.  Measured (PMC,
// tools/resident_pmc.sh): 4.6 MB per window and LM iteration = 2.8x the algorithmic 1.66 MB, where the multi-launch path moves 26 MB.
// Sums into S are atomic, hence in no fixed order: results agree with the multi-launch path and the oracle to
// rounding (1e-12 relative on the cost), not bit for bit - the parity bar of this path is north_star's 1e-5.
//
// A landmark with more than 64 observations is refused (BaCtl::error = 2: the caller runs the window on the multi-launch path).
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
    // (no "old" operand: every lane reads a lane of its own row, so nothing of the destination survives - with one, the compiler
    // copies the source first and a butterfly step costs five instructions instead of three)
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
constexpr int kDppXor1 = 0xB1;          // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;          // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141;   // lane i <-> 7 - i inside every 8 lanes
constexpr int kDppMirror = 0x140;       // lane i <-> 15 - i inside every 16 lanes

// the sum over an aligned group of G lanes (4, 8, 16, 32 or 64), in every lane of the group
template <int G>
__device__ __forceinline__ double gsum(double v) {
    v += dpp_move<kDppXor1>(v);
    v += dpp_move<kDppXor2>(v);
    if (G >= 8) v += dpp_move<kDppHalfMirror>(v);
    if (G >= 16) v += dpp_move<kDppMirror>(v);
    if (G >= 32) v += __shfl_xor(v, 16);
    if (G >= 64) v += __shfl_xor(v, 32);
    return v;
}
__device__ __forceinline__ double wsum(double v) { return gsum<64>(v); }
__device__ __forceinline__ double wmax(double v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v = fmax(v, __shfl_xor(v, m));
    return v;
}

__device__ __forceinline__ int tri(int r, int c) { return r * (r + 1) / 2 + c; }   // packed lower triangle, r >= c

// (a run-time debug switch lived here until r06q - plain read-modify-write instead of the atomic, the pair loop skipped: it put a
// branch around every one of the kernel's atomics and cut the schedule into as many pieces; the two timings it gave are in DESIGN.md)
__device__ __forceinline__ void lds_add(double* p, double v) {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// 1 / x and 1 / sqrt(x) from the hardware's seeds (v_rcp_f64 / v_rsq_f64: ~26 good bits) and two Newton steps: within an ulp or two of
// the correctly rounded value at 5 / 9 instructions, where the compiler's IEEE division and square root take 12 / 25
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double fast_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double h = 0.5 * x;
    y = y * fma(-h * y, y, 1.5);
    y = y * fma(-h * y, y, 1.5);
    return y;
}

// A = G^-1 for M = h + lambda I = G G^T (badev::chol_inv3 with the reciprocal square roots above; the same pivot floor)
__device__ __forceinline__ void chol3(const double h[6], double lambda, double a[6]) {
    const double m00 = h[0] + lambda, m10 = h[1], m20 = h[2], m11 = h[3] + lambda, m21 = h[4], m22 = h[5] + lambda;
    const double floor_ = fmax(1e-30 * (m00 + m11 + m22), 1e-300);
    const double a00 = fast_rsqrt(fmax(m00, floor_));
    const double g10 = m10 * a00, g20 = m20 * a00;
    const double a11 = fast_rsqrt(fmax(m11 - g10 * g10, floor_));
    const double g21 = (m21 - g20 * g10) * a11;
    const double a22 = fast_rsqrt(fmax(m22 - g20 * g20 - g21 * g21, floor_));
    const double a10 = -(a11 * g10) * a00;
    const double a21 = -(a22 * g21) * a11;
    const double a20 = -(a21 * g10 + a22 * g20) * a00;
    a[0] = a00; a[1] = a10; a[2] = a11; a[3] = a20; a[4] = a21; a[5] = a22;
}

// RobustKernelHuber (badev::huber) without the branch and with the reciprocal square root above: sqrt(e2) = e2 rsqrt(e2)
__device__ __forceinline__ void huber_w(double e2, double delta, double& rho0, double& rho1) {
    const double dsqr = delta * delta;
    const double rs = fast_rsqrt(fmax(e2, 1e-300));
    const bool in = e2 <= dsqr;
    rho0 = in ? e2 : 2.0 * (e2 * rs) * delta - dsqr;
    rho1 = in ? 1.0 : delta * rs;
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
    const double zi = fast_rcp(Z);
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

enum { kEval = 0, kDiag = 1, kUpdate = 3 };

// what a pass needs of the window, all in LDS except the edge arrays, the landmarks and the landmark order
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
    const int4* desc;   // L: {landmark, first record, observations, first edge}: the landmarks class by class (made by the prologue)
    // the observations of the landmarks with at most 16 of them, copied by the prologue in the ORDER OF THAT LIST (one array per field)
    const double2* r_uv;
    const double2* r_w01;
    const double* r_w2;
    const int* r_kf;
    int n;
    double lambda;
};

struct EdgeIn {   // one observation as it comes from memory
    int kf;
    double u, v, w0, w1, w2;
};
__device__ __forceinline__ EdgeIn load_edge(const WindowArgs& a, int e) {
    EdgeIn r;
    r.kf = a.e_kf[e];
    const double2 uv = reinterpret_cast<const double2*>(a.e_uv)[e];
    r.u = uv.x; r.v = uv.y;
    r.w0 = a.e_info[3 * (size_t)e]; r.w1 = a.e_info[3 * (size_t)e + 1]; r.w2 = a.e_info[3 * (size_t)e + 2];
    return r;
}

// ------------------------------------------------------------------------------------------------------------------
// All passes: one lane per OBSERVATION, an aligned group of G lanes per landmark (G = 4, 8, 16 or 64 by the landmark's count; the
// landmarks are visited in ascending order of their counts, so a wave's groups are of a kind and consecutive lanes read consecutive
// edges).
// ------------------------------------------------------------------------------------------------------------------
struct GroupIn {
    int l, beg, k, at;   // landmark, first record, observations, place in the list
    double lx, ly, lz;
    EdgeIn ed;
    bool has;
};
// a landmark's descriptor {landmark, first record, observations, first edge} from the list the prologue made; zero beyond the class's end
__device__ __forceinline__ int4 load_desc(const Ctx& c, int idx, int end) {
    return idx < end ? c.desc[idx] : make_int4(0, 0, 0, 0);
}
// FIRST: the opening pass of an optimize() - the observations still come from the caller's arrays (d.w) and go to the record arrays
// on the way, so that every later pass reads them in the order of the list
template <int G, bool FIRST = false>
__device__ __forceinline__ GroupIn load_group(const Ctx& c, const int4 d, int lane, int at) {
    const WindowArgs& a = *c.a;
    GroupIn g;
    g.l = d.x; g.beg = d.y; g.k = d.z; g.at = at; g.lx = 0; g.ly = 0; g.lz = 1; g.has = false;
    g.ed = EdgeIn{0, 0, 0, 0, 0, 0};
    if (g.k > 0) {
        g.lx = c.lms[3 * (size_t)g.l]; g.ly = c.lms[3 * (size_t)g.l + 1]; g.lz = c.lms[3 * (size_t)g.l + 2];
        const int sub = lane & (G - 1);
        g.has = sub < g.k;
        if (g.has) {
            if (G == 64) {
                g.ed = load_edge(a, g.beg + sub);   // (a wave per landmark: its observations lie together in the caller's arrays as they are)
            } else if (FIRST) {
                g.ed = load_edge(a, d.w + sub);
                const int e = g.beg + sub;
                const_cast<double2*>(c.r_uv)[e] = double2{g.ed.u, g.ed.v};
                const_cast<double2*>(c.r_w01)[e] = double2{g.ed.w0, g.ed.w1};
                const_cast<double*>(c.r_w2)[e] = g.ed.w2;
                const_cast<int*>(c.r_kf)[e] = g.ed.kf;
            } else {
                const int e = g.beg + sub;
                const double2 uv = c.r_uv[e], w01 = c.r_w01[e];
                g.ed.kf = c.r_kf[e];
                g.ed.u = uv.x; g.ed.v = uv.y;
                g.ed.w0 = w01.x; g.ed.w1 = w01.y; g.ed.w2 = c.r_w2[e];
            }
        }
    }
    return g;
}
// EVAL / DIAG / UPDATE of one landmark.  Nothing but sums crosses lanes: for the back-substitution bl - q with
// q = sum_e Hlp_e dp_e, and then   x_l = (Hll + lambda I)^-1 (bl - q) = A^T A (bl - q)   needs neither W_e nor a second look at the
// Jacobians (sum_e W_e^T dp_e = A q: the factor A comes out of the sum).  A itself - 48 bytes per landmark - is what the build pass of
// the same trial computed: it travels through memory (WindowArgs::ainv, list order), which spares this pass the six sums of Hll and
// its factorisation, a third of its instructions.
template <int MODE, int G>
__device__ __forceinline__ void eval_group(const Ctx& c, const GroupIn& g, int lane, double& chi, double& scale, double& dmax) {
    const WindowArgs& a = *c.a;
    const int sub = lane & (G - 1);
    const bool has = g.has;
    const int kf = g.ed.kf;
    const double w0 = g.ed.w0, w1 = g.ed.w1, w2 = g.ed.w2;
    const double px = c.cur[3 * kf], py = c.cur[3 * kf + 1], ps = c.scur[2 * kf], pc = c.scur[2 * kf + 1];
    double e0, e1;
    if (MODE == kEval) {
        edge_se2xyz<false>(a.cam, px, py, ps, pc, g.lx, g.ly, g.lz, g.ed.u, g.ed.v, e0, e1, nullptr, nullptr);
        double r0, r1;
        huber_w(e0 * (w0 * e0 + w1 * e1) + e1 * (w1 * e0 + w2 * e1), a.cam.huber, r0, r1);
        if (has) chi += r0;
        return;
    }
    const int c0 = has ? c.col[kf] : -1;
    // the update pass: A = G^-1 of Hll + lambda I as this trial's build pass left it (requested here, needed after the sums)
    double2 A01 = {1, 0}, A23 = {1, 0}, A45 = {0, 1};
    if (MODE == kUpdate && g.k > 0) {
        const double2* src = reinterpret_cast<const double2*>(a.ainv + 6 * (size_t)g.at);
        A01 = src[0]; A23 = src[1]; A45 = src[2];
    }
    double Jp[6], Jl[6];
    edge_se2xyz<true>(a.cam, px, py, ps, pc, g.lx, g.ly, g.lz, g.ed.u, g.ed.v, e0, e1, Jp, Jl);
    const double we0 = w0 * e0 + w1 * e1, we1 = w1 * e0 + w2 * e1;
    double r0, r1;
    huber_w(e0 * we0 + e1 * we1, a.cam.huber, r0, r1);
    if (MODE == kDiag && has) chi += r0;                       // (the lambda_0 pass is the chi^2 of the starting state as well)
    const double W0 = r1 * w0, W1 = r1 * w1, W2 = r1 * w2;    // weightedOmega
    const double or0 = -r1 * we0, or1 = -r1 * we1;            // omega_r
    double WJl[6];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        WJl[m] = W0 * Jl[m] + W1 * Jl[3 + m];
        WJl[3 + m] = W1 * Jl[m] + W2 * Jl[3 + m];
    }
    double acc[12];   // hll (6: the lambda_0 pass only) | bl (3) | q (3); the update pass sums bl - q as one vector (slots 6..8)
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[i] = 0.0;
    if (MODE == kDiag) {
        acc[0] = Jl[0] * WJl[0] + Jl[3] * WJl[3];
        acc[3] = Jl[1] * WJl[1] + Jl[4] * WJl[4];
        acc[5] = Jl[2] * WJl[2] + Jl[5] * WJl[5];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) acc[6 + r] = Jl[r] * or0 + Jl[3 + r] * or1;
    acc[9] = acc[10] = acc[11] = 0.0;
    if (c0 >= 0) {
        if (MODE == kDiag) {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double wj0 = W0 * Jp[r] + W1 * Jp[3 + r], wj1 = W1 * Jp[r] + W2 * Jp[3 + r];
                lds_add(c.x + c0 + r, Jp[r] * wj0 + Jp[3 + r] * wj1);
            }
        } else {
            const double d0 = c.x[c0], d1 = c.x[c0 + 1], d2 = c.x[c0 + 2];
            const double v0 = Jp[0] * d0 + Jp[1] * d1 + Jp[2] * d2, v1 = Jp[3] * d0 + Jp[4] * d1 + Jp[5] * d2;   // Jp dp
#pragma unroll
            for (int m = 0; m < 3; ++m) acc[9 + m] = WJl[m] * v0 + WJl[3 + m] * v1;    // Hlp_e dp_e = Jl^T Omega' (Jp dp)
            scale += v0 * or0 + v1 * or1;                                              // dp . b_e, the edge's share of dp . b_p
        }
    }
    if (!has) {   // (its arithmetic ran on a made-up edge and may hold infinities: nothing of it may reach the group's sums)
#pragma unroll
        for (int i = 0; i < 12; ++i) acc[i] = 0.0;
    }
    if (MODE == kDiag) {   // lambda_0 = 1e-5 max diag H (computeLambdaInit): the landmark blocks' diagonals
        const double h0 = gsum<G>(acc[0]), h3 = gsum<G>(acc[3]), h5 = gsum<G>(acc[5]);
        if (g.k > 0) dmax = fmax(dmax, fmax(fabs(h0), fmax(fabs(h3), fabs(h5))));
        return;
    }
    const double ble[3] = {acc[6], acc[7], acc[8]};   // this observation's own b_l share: x_l . b_l is summed observation by observation
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[6 + i] = gsum<G>(acc[6 + i] - acc[9 + i]);
    double xl[3] = {0, 0, 0};
    if (g.k > 0) {
        const double A[6] = {A01.x, A01.y, A23.x, A23.y, A45.x, A45.y};
        const double g0 = acc[6], g1 = acc[7], g2 = acc[8];
        const double t0 = A[0] * g0, t1 = A[1] * g0 + A[2] * g1, t2 = A[3] * g0 + A[4] * g1 + A[5] * g2;    // A (bl - q)
        xl[0] = A[0] * t0 + A[1] * t1 + A[3] * t2;                                                           // A^T (...)
        xl[1] = A[2] * t1 + A[4] * t2;
        xl[2] = A[5] * t2;
    }
    const double nxl = g.lx + xl[0], nyl = g.ly + xl[1], nzl = g.lz + xl[2];
    scale += xl[0] * ble[0] + xl[1] * ble[1] + xl[2] * ble[2];   // (zero without an observation)
    if (g.k > 0 && sub == 0) {
        c.lms_trial[3 * (size_t)g.l] = nxl; c.lms_trial[3 * (size_t)g.l + 1] = nyl; c.lms_trial[3 * (size_t)g.l + 2] = nzl;
        scale += c.lambda * (xl[0] * xl[0] + xl[1] * xl[1] + xl[2] * xl[2]);
    }
    if (has) {   // robust chi^2 of the observation at the trial state
        edge_se2xyz<false>(a.cam, c.trl[3 * kf], c.trl[3 * kf + 1], c.strl[2 * kf], c.strl[2 * kf + 1], nxl, nyl, nzl, g.ed.u, g.ed.v, e0, e1, nullptr, nullptr);
        double q0, q1;
        huber_w(e0 * (w0 * e0 + w1 * e1) + e1 * (w1 * e0 + w2 * e1), a.cam.huber, q0, q1);
        chi += q0;
    }
}

// the landmarks [begin, end) of the order, G lanes each; the next group's operands are in flight while this one is worked on
template <int MODE, int G, int NT, bool FIRST = false>
__device__ __forceinline__ void eval_class(const Ctx& c, int begin, int end, double& chi, double& scale, double& dmax) {
    const int tid = threadIdx.x, lane = tid & 63;
    if (begin >= end) return;
    constexpr int kStep = NT / G;
    GroupIn nx = load_group<G, FIRST>(c, load_desc(c, begin + tid / G, end), lane, begin + tid / G);
    int4 d2 = load_desc(c, begin + kStep + tid / G, end);
    for (int i0 = begin; i0 < end; i0 += kStep) {
        const GroupIn g = nx;
        nx = load_group<G, FIRST>(c, d2, lane, i0 + kStep + tid / G);
        d2 = load_desc(c, i0 + 2 * kStep + tid / G, end);
        eval_group<MODE, G>(c, g, lane, chi, scale, dmax);
    }
}
// landmarks without an observation keep their place: their trial position is their position
template <int NT>
__device__ __forceinline__ void copy_unobserved(const Ctx& c, int end) {
    for (int i = threadIdx.x; i < end; i += NT) {
        const int l = c.desc[i].x;
#pragma unroll
        for (int m = 0; m < 3; ++m) c.lms_trial[3 * (size_t)l + m] = c.lms[3 * (size_t)l + m];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// BUILD of one landmark
// ------------------------------------------------------------------------------------------------------------------
template <int G>
__device__ __forceinline__ void build_group(const Ctx& c, const GroupIn& g, int lane) {
    const WindowArgs& a = *c.a;
    const int sub = lane & (G - 1);
    const int k = g.k;
    const bool has = g.has;
    const int kf = g.ed.kf;
    const double w0 = g.ed.w0, w1 = g.ed.w1, w2 = g.ed.w2;
    const double px = c.cur[3 * kf], py = c.cur[3 * kf + 1], ps = c.scur[2 * kf], pc = c.scur[2 * kf + 1];
    const int c0 = has ? c.col[kf] : -1;
    double e0, e1, Jp[6], Jl[6];
    edge_se2xyz<true>(a.cam, px, py, ps, pc, g.lx, g.ly, g.lz, g.ed.u, g.ed.v, e0, e1, Jp, Jl);
    const double we0 = w0 * e0 + w1 * e1, we1 = w1 * e0 + w2 * e1;
    double r0, r1;
    huber_w(e0 * we0 + e1 * we1, a.cam.huber, r0, r1);
    const double W0 = r1 * w0, W1 = r1 * w1, W2 = r1 * w2;    // weightedOmega
    const double or0 = -r1 * we0, or1 = -r1 * we1;            // omega_r
    double WJl[6];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        WJl[m] = W0 * Jl[m] + W1 * Jl[3 + m];
        WJl[3 + m] = W1 * Jl[m] + W2 * Jl[3 + m];
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
#pragma unroll
    for (int i = 0; i < 6; ++i) hll[i] = gsum<G>(hll[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) b[i] = gsum<G>(b[i]);
    double A[6], zt[3];
    chol3(hll, c.lambda, A);
    if (k > 0 && sub == 0) {   // the update pass of this trial takes the factor from here instead of summing Hll and factorising it again
        double2* dst = reinterpret_cast<double2*>(a.ainv + 6 * (size_t)g.at);
        dst[0] = double2{A[0], A[1]}; dst[1] = double2{A[2], A[3]}; dst[2] = double2{A[4], A[5]};
    }
    zt[0] = A[0] * b[0];
    zt[1] = A[1] * b[0] + A[2] * b[1];
    zt[2] = A[3] * b[0] + A[4] * b[1] + A[5] * b[2];
    const bool fr = c0 >= 0;
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
    if (fr) {
        double WJp[6];
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            WJp[m] = W0 * Jp[m] + W1 * Jp[3 + m];
            WJp[3 + m] = W1 * Jp[m] + W2 * Jp[3 + m];
        }
        // the pose's own block: Hpp_e - W_e W_e^T (lower triangle) and its right-hand side b_e - W_e zeta
        int ob[3];   // the block's three rows at its first column: one multiplication, two additions
        ob[0] = tri(c0, c0); ob[1] = ob[0] + c0 + 1; ob[2] = ob[1] + c0 + 2;
        const int nrow = tri(c.n, 0) + c0;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int m = 0; m <= r; ++m) {
                const double hpp = Jp[r] * WJp[m] + Jp[3 + r] * WJp[3 + m];
                const double ww = Wm[r * 3] * Wm[m * 3] + Wm[r * 3 + 1] * Wm[m * 3 + 1] + Wm[r * 3 + 2] * Wm[m * 3 + 2];
                lds_add(c.S + ob[r] + m, hpp - ww);
            }
            const double bpe = Jp[r] * or0 + Jp[3 + r] * or1;
            lds_add(c.S + nrow + r, bpe - (Wm[r * 3] * zt[0] + Wm[r * 3 + 1] * zt[1] + Wm[r * 3 + 2] * zt[2]));
        }
    }
    // the pair products of the landmark's observations: every lane puts W_e and its column into the wave's strip, lane i then takes
    // the partners (i + s) mod k, s = 1 .. k / 2 (the pairs at distance k / 2 of an even k only from the lower half)
    double* mine = c.stage + lane * kStageDoubles;   // (32-bit index arithmetic: LDS)
#pragma unroll
    for (int i = 0; i < 9; ++i) mine[i] = Wm[i];
    mine[9] = (double)c0;
    // (the strip is this wave's alone and a wave's LDS operations execute in the order they were issued: what the other lanes wrote
    // is there when the reads below arrive - only the compiler has to keep the order)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    const int half = k >> 1;
    int smax = half;
#pragma unroll
    for (int m = G; m < 64; m <<= 1) smax = max(smax, __shfl_xor(smax, m));   // the wave's longest landmark sets the trip count
    const int gbase = lane & ~(G - 1);
    // (requesting the partner of step s + 1 before the atomics of step s go out - so that its products run while those drain - was
    // measured and is slower: 406 k against 414 k LM it/s at 256 windows; ten more live registers per lane)
    for (int s = 1; s <= smax; ++s) {
        const bool act = has && s <= half && !(2 * s == k && sub >= half);
        int j = sub + s;
        if (j >= k) j -= k;
        const double* his = c.stage + (gbase + (act ? j : sub)) * kStageDoubles;
        double Wp[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) Wp[i] = his[i];
        const int cp = (int)his[9];
        if (act && fr && cp >= 0) {
            // block (mine, his) of S loses W_mine W_his^T; it is stored where row > column.  (Two observations of one landmark by
            // the SAME key frame - the reference never builds that - land in the pose's own block: P + P^T, lower triangle.)
            // One multiplication for the block's place - the first of its three rows, the others follow by additions - and one
            // select per entry between "my rows, his columns" and the transposed place.
            const bool lower = c0 > cp, same = c0 == cp;
            const int hi = lower ? c0 : cp, lo = lower ? cp : c0;
            int rb[3];
            rb[0] = tri(hi, lo); rb[1] = rb[0] + hi + 1; rb[2] = rb[1] + hi + 2;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    double pr = Wm[r * 3] * Wp[m * 3] + Wm[r * 3 + 1] * Wp[m * 3 + 1] + Wm[r * 3 + 2] * Wp[m * 3 + 2];
                    int at = rb[r] + m;                                   // r == m: the same place either way
                    if (r > m) at = (lower || same) ? rb[r] + m : rb[m] + r;
                    if (r < m) at = lower ? rb[r] + m : rb[m] + r;        // (same: not lower, row m = max(r, m))
                    if (r == m && same) pr *= 2.0;
                    lds_add(c.S + at, -pr);
                }
        }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();   // (the next landmark's strip writes stay behind these reads)
    asm volatile("" ::: "memory");
}

// the landmarks [begin, end) of the list, G lanes each; the next group's operands and the descriptor after that are in flight while
// this one is worked on
template <int G, int NT>
__device__ __forceinline__ void build_class(const Ctx& c, int begin, int end) {
    const int tid = threadIdx.x, lane = tid & 63;
    if (begin >= end) return;
    constexpr int kStep = NT / G;
    GroupIn nx = load_group<G>(c, load_desc(c, begin + tid / G, end), lane, begin + tid / G);
    int4 d2 = load_desc(c, begin + kStep + tid / G, end);
    for (int i0 = begin; i0 < end; i0 += kStep) {
        const GroupIn g = nx;
        nx = load_group<G>(c, d2, lane, i0 + kStep + tid / G);
        d2 = load_desc(c, i0 + 2 * kStep + tid / G, end);
        build_group<G>(c, g, lane);
    }
}

// PreEdgeSE2 (EdgeSE2XYZ.h:62-102), one thread per edge
enum { kOdoEval = 0, kOdoDiag = 1, kOdoBuild = 2, kOdoUpdate = 3 };
template <int MODE>
__device__ __forceinline__ void odometry_edge(const Ctx& c, int k, double& chi, double& scale) {
    const WindowArgs& a = *c.a;
    const int i = a.o_i[k], j = a.o_j[k];
    const double* W = a.o_info + 9 * (size_t)k;
    double e[3], A[9], B[9];
    if (MODE == kOdoUpdate) {   // the gain denominator's share at the estimate's linearisation, chi^2 at the trial state
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
    if (MODE == kOdoEval) {
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
            if (MODE == kOdoDiag) {
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
        if (MODE == kOdoBuild) {
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

// ------------------------------------------------------------------------------------------------------------------
// LL^T of the augmented system in place (left-looking, 3x3 blocks, the right-hand side as row n): the off-diagonal blocks of L
// overwrite S - the diagonal blocks too (over T(J, J), once everybody has it) - with the reciprocals of the diagonal in invd.  *fail is set
// when a pivot is not positive (the step is then rejected, as g2o rejects a failed Cholesky).
// Block column J: T(I, J) = S(I, J) - sum_{K < J} L(I, K) L(J, K)^T for the block rows I >= J, LPB lanes sharing a block's sum over
// K (8 while the column is long, up to 64 near the end, where few rows are left and the sum is longest); then every block's first
// lane factorises T(J, J) for itself (six multiplies: cheaper than a hand-off) and solves its own L(I, J) = T(I, J) L(J, J)^-T.
// ------------------------------------------------------------------------------------------------------------------
template <int NT, int LPB>
__device__ __forceinline__ void factor_column(double* S, double* invd, double* tjj, int nf, int J, int* fail) {
    const int tid = threadIdx.x, sub = tid & (LPB - 1), grp = tid / LPB;
    // (the right-hand side is block row nf: row n and two rows of zeros, so every block row is 3 x 3 and the loops unroll)
    for (int I0 = J; I0 <= nf; I0 += NT / LPB) {
        const int I = I0 + grp;
        const bool live = I <= nf;
        const int Ic = live ? I : J;
        const double* rowI[3] = {S + tri(3 * Ic, 0), S + tri(3 * Ic + 1, 0), S + tri(3 * Ic + 2, 0)};
        const double* rowJ[3] = {S + tri(3 * J, 0), S + tri(3 * J + 1, 0), S + tri(3 * J + 2, 0)};
        double t[9];   // S(I, J), requested before the sum it will be reduced by
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) t[r * 3 + q] = (sub == 0 && (Ic > J || q <= r)) ? rowI[r][3 * J + q] : 0.0;
        double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int K = sub; K < J; K += LPB) {
            double lj[9], li[9];
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int m = 0; m < 3; ++m) lj[q * 3 + m] = rowJ[q][3 * K + m];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int m = 0; m < 3; ++m) li[r * 3 + m] = rowI[r][3 * K + m];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int q = 0; q < 3; ++q) acc[r * 3 + q] += li[r * 3] * lj[q * 3] + li[r * 3 + 1] * lj[q * 3 + 1] + li[r * 3 + 2] * lj[q * 3 + 2];
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) t[i] -= gsum<LPB>(acc[i]);
        // T(J, J) goes to everybody through a six-word strip; the other blocks keep their T in registers across the barrier
        if (sub == 0 && live && I == J) { tjj[0] = t[0]; tjj[1] = t[3]; tjj[2] = t[4]; tjj[3] = t[6]; tjj[4] = t[7]; tjj[5] = t[8]; }
        __syncthreads();
        if (sub == 0 && live) {
            const double t00 = tjj[0], t10 = tjj[1], t11 = tjj[2], t20 = tjj[3], t21 = tjj[4], t22 = tjj[5];
            bool bad = !(t00 > 0.0);
            const double i00 = fast_rsqrt(bad ? 1.0 : t00), l00 = t00 * i00;
            const double l10 = t10 * i00, l20 = t20 * i00;
            const double d1 = t11 - l10 * l10;
            bad |= !(d1 > 0.0);
            const double i11 = fast_rsqrt(d1 > 0.0 ? d1 : 1.0), l11 = d1 * i11;
            const double l21 = (t21 - l20 * l10) * i11;
            const double d2 = t22 - l20 * l20 - l21 * l21;
            bad |= !(d2 > 0.0);
            const double i22 = fast_rsqrt(d2 > 0.0 ? d2 : 1.0), l22 = d2 * i22;
            if (I == J) {
                if (bad) *fail = 1;
                // L(J, J) over T(J, J) in place (everybody reads T(J, J) from the strip tjj, and no later column reads a diagonal block)
                S[tri(3 * J, 3 * J)] = l00;
                S[tri(3 * J + 1, 3 * J)] = l10; S[tri(3 * J + 1, 3 * J + 1)] = l11;
                S[tri(3 * J + 2, 3 * J)] = l20; S[tri(3 * J + 2, 3 * J + 1)] = l21; S[tri(3 * J + 2, 3 * J + 2)] = l22;
                invd[3 * J] = i00; invd[3 * J + 1] = i11; invd[3 * J + 2] = i22;
            } else {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    double* o = S + tri(3 * I + r, 3 * J);
                    const double x0 = t[r * 3] * i00;
                    const double x1 = (t[r * 3 + 1] - x0 * l10) * i11;
                    const double x2 = (t[r * 3 + 2] - x0 * l20 - x1 * l21) * i22;
                    o[0] = x0; o[1] = x1; o[2] = x2;
                }
            }
        }
        __syncthreads();
    }
}
template <int NT>
__device__ __forceinline__ void factorize(double* S, double* invd, double* tjj, int nf, int* fail) {
    for (int J = 0; J < nf; ++J) {
        const int blocks = nf - J + 1;
        if (blocks * 64 <= NT) factor_column<NT, 64>(S, invd, tjj, nf, J, fail);
        else if (blocks * 32 <= NT) factor_column<NT, 32>(S, invd, tjj, nf, J, fail);
        else if (blocks * 16 <= NT) factor_column<NT, 16>(S, invd, tjj, nf, J, fail);
        else factor_column<NT, 8>(S, invd, tjj, nf, J, fail);
    }
}

// x = L^-T y by ONE wave: y (row n of the factor) in registers, three unknowns per lane, a pose block per step from the last to the
// first: the block's three unknowns by scalar broadcasts (readlane) and its own 3x3 triangle, then its three rows of L - requested a
// step ahead - leave every earlier unknown's y.  n <= 192.
__device__ __forceinline__ double lane_value(double v, int src) {
    const long long bits = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(bits & 0xffffffffll), src);
    const int hi = __builtin_amdgcn_readlane((int)(bits >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ void back_substitute(const double* S, const double* invd, int nf, double* x) {
    const int lane = threadIdx.x & 63;
    const int n = 3 * nf;
    double y[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int i = lane + 64 * s;
        y[s] = i < n ? S[tri(n, i)] : 0.0;
    }
    auto load_rows = [&](int J, double (&rw)[3][3], double (&d)[6], double (&iv)[3]) {
        if (J < 0) return;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int i = lane + 64 * s;
                rw[r][s] = i < 3 * J ? S[tri(3 * J + r, i)] : 0.0;
            }
        d[0] = S[tri(3 * J, 3 * J)];
        d[1] = S[tri(3 * J + 1, 3 * J)]; d[2] = S[tri(3 * J + 1, 3 * J + 1)];
        d[3] = S[tri(3 * J + 2, 3 * J)]; d[4] = S[tri(3 * J + 2, 3 * J + 1)]; d[5] = S[tri(3 * J + 2, 3 * J + 2)];
#pragma unroll
        for (int i = 0; i < 3; ++i) iv[i] = invd[3 * J + i];
    };
    double rw[3][3], d[6], iv[3], nrw[3][3] = {}, nd[6] = {}, niv[3] = {};
    load_rows(nf - 1, rw, d, iv);
    for (int J = nf - 1; J >= 0; --J) {
        load_rows(J - 1, nrw, nd, niv);
        double yb[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int j = 3 * J + r;
            const double ys = j >= 128 ? y[2] : (j >= 64 ? y[1] : y[0]);
            yb[r] = lane_value(ys, j & 63);
        }
        const double x2 = yb[2] * iv[2];
        const double x1 = (yb[1] - d[4] * x2) * iv[1];
        const double x0 = (yb[0] - d[1] * x1 - d[3] * x2) * iv[0];
        if (lane == 0) { x[3 * J] = x0; x[3 * J + 1] = x1; x[3 * J + 2] = x2; }
#pragma unroll
        for (int s = 0; s < 3; ++s) y[s] -= rw[0][s] * x0 + rw[1][s] * x1 + rw[2][s] * x2;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) rw[r][s] = nrw[r][s];
#pragma unroll
        for (int i = 0; i < 6; ++i) d[i] = nd[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) iv[i] = niv[i];
    }
}

template <int NT>
__global__ __launch_bounds__(NT) void k_window_lm(const WindowArgs* __restrict__ all) {
    const WindowArgs& a = all[blockIdx.x];
    extern __shared__ double lds[];
    __shared__ BaCtl ctl;
    __shared__ int s_nf, s_fail, s_err, s_stop;
    __shared__ int hist[kWindowMaxDegree + 2], wtot[18][8], rstart[18];
    __shared__ double red[24], tjj[6];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int P = a.P, L = a.L;
    if (a.stamps && tid == 0) a.stamps[5] = wall_clock64();

    // ---- prologue: the controller block (k_ctl_init's rules), columns of the free poses, the landmarks ordered by their counts
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
        s_fail = 0; s_err = 0;
        s_stop = (a.stop && *(const volatile int*)a.stop) ? 1 : 0;
    }
    for (int i = tid; i < kWindowMaxDegree + 2; i += NT) hist[i] = 0;
    // LDS map (doubles): cur 3P | trl 3P | scur 2P | strl 2P | x n | invd n | stage NT x kStageDoubles | S (n+3)(n+4)/2 ; col P ints first
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
    for (int l = tid; l < L; l += NT) {
        const int k = a.lm_ptr[l + 1] - a.lm_ptr[l];
        if (k > kWindowMaxDegree) s_err = 2;
        else atomicAdd(&hist[k], 1);
    }
    __syncthreads();
    if (tid == 0) {   // hist[k] -> first position of the landmarks with k observations, rstart[k] -> their first record
        int at = 0, rat = 0;
        for (int k = 0; k <= kWindowMaxDegree + 1; ++k) {
            const int h = hist[k];
            hist[k] = at;
            if (k < 18) rstart[k] = rat;
            at += h;
            rat += h * k;
        }
    }
    __syncthreads();
    const bool refused = s_err != 0;   // a landmark this kernel does not take: nothing is touched, the caller runs the window elsewhere
    if (!refused) {
        // the list: the landmarks by their number of observations (0, 1, ... 16, more), STABLE inside a count, so that the groups of
        // a wave are alike (the longest landmark of a wave sets the trip count of its pair loop) and consecutive groups read
        // ascending addresses of the edge arrays (a list in arbitrary order inside a count fetched every cache line of the edges
        // about twice: PMC, profiles/r06h).  Rounds of NT landmarks; a landmark's place = its count's start + the members before
        // it (ballot ranks inside the wave, wave totals through LDS, the rounds' totals in registers).
        constexpr int kBuckets = 18;
        int next[kBuckets];
#pragma unroll
        for (int b = 0; b < kBuckets; ++b) next[b] = hist[b];
        const int lane = tid & 63;
        const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
        for (int l0 = 0; l0 < L; l0 += NT) {
            const int l = l0 + tid;
            int beg = 0, k = 0, cls = -1;
            if (l < L) {
                beg = a.lm_ptr[l];
                k = a.lm_ptr[l + 1] - beg;
                cls = min(k, kBuckets - 1);
            }
            int rank = 0;
#pragma unroll
            for (int b = 0; b < kBuckets; ++b) {
                const unsigned long long m = __ballot(cls == b);
                if (cls == b) rank = __popcll(m & below);
                if (lane == 0) wtot[b][wave] = __popcll(m);
            }
            __syncthreads();
            if (cls >= 0) {
                int at = rank;
#pragma unroll
                for (int b = 0; b < kBuckets; ++b)
                    if (cls == b) at += next[b];
                for (int w = 0; w < wave; ++w) at += wtot[cls][w];
                // (all landmarks of a count below 17 have that many records: the place of a landmark's first one follows from its own)
                a.desc[at] = make_int4(l, cls < kBuckets - 1 ? rstart[cls] + (at - hist[cls]) * k : beg, k, beg);
            }
#pragma unroll
            for (int b = 0; b < kBuckets; ++b)
                for (int w = 0; w < NT / 64; ++w) next[b] += wtot[b][w];
            __syncthreads();
        }
    }
    if (a.stamps && tid == 0) a.stamps[6] = wall_clock64();   // (the list is made)
    const int nf = s_nf, n = 3 * nf;
    double* xs = scB + 2 * P;
    double* invd = xs + n;
    double* stage_all = invd + n;
    double* S = stage_all + (size_t)NT * kStageDoubles;
    const int ntri = (n + 3) * (n + 4) / 2;   // rows 0 .. n-1 = S, row n = b_s, two rows of zeros (the factorisation's last block row)
    for (int p = tid; p < P; p += NT) sincos(bufA[3 * p + 2], &scA[2 * p], &scA[2 * p + 1]);
    if (refused && tid == 0) { ctl.error = 2; ctl.done = 1; }
    __syncthreads();
    // classes of the build pass: landmarks with 1-4 observations take 4 lanes, 5-8 take 8, 9-16 take 16, the rest a wave
    const int b1 = hist[1], b5 = hist[5], b9 = hist[9], b17 = hist[17];

    Ctx c;
    c.a = &a;
    c.S = S;
    c.x = xs;
    c.cur = bufA; c.scur = scA; c.trl = bufB; c.strl = scB;
    c.col = col;
    c.stage = stage_all + (size_t)wave * 64 * kStageDoubles;
    c.lms = ctl.sel ? a.lms_b : a.lms_a;
    c.lms_trial = ctl.sel ? a.lms_a : a.lms_b;
    c.desc = a.desc;
    {   // the record arrays lie behind the list: 16 + 16 + 8 + 4 bytes per observation (the caller has checked the room: ba_resident_ok)
        const size_t E = (size_t)a.E;
        c.r_uv = reinterpret_cast<const double2*>(a.desc + L);
        c.r_w01 = c.r_uv + E;
        c.r_w2 = reinterpret_cast<const double*>(c.r_w01 + E);
        c.r_kf = reinterpret_cast<const int*>(c.r_w2 + E);
    }
    c.n = n;
    c.lambda = 0.0;
    double* cur = bufA;
    double* trl = bufB;
    double* scur = scA;
    double* strl = scB;

    // ---- the opening pass: chi^2 of the starting state (computeActiveErrors + activeRobustChi2 in front of the first iteration),
    // for Levenberg-Marquardt together with the diagonal of the first linearisation (lambda_0 = 1e-5 max diag H, computeLambdaInit;
    // Gauss-Newton keeps lambda = 0) - and on the way the observations go into the ORDER OF THE LIST.  The passes visit the landmarks
    // class by class; in the caller's arrays a class's landmarks alternate with the others', and every cache line of the edge arrays
    // came in once PER CLASS that has a landmark in it (PMC, profiles/r06k: 6.6 MB per window and iteration for 3.0 MB of operands).
    // One gapped read here, and every pass of every trial reads whole lines.
    if (!refused) {
        const bool lm = a.mode == SE2GPU_BA_LM;
        for (int i = tid; i < n; i += NT) xs[i] = 0.0;
        __syncthreads();
        double chi = 0, sc = 0, dm = 0;
        if (lm) {
            eval_class<kDiag, 4, NT, true>(c, b1, b5, chi, sc, dm);
            eval_class<kDiag, 8, NT, true>(c, b5, b9, chi, sc, dm);
            eval_class<kDiag, 16, NT, true>(c, b9, b17, chi, sc, dm);
            eval_class<kDiag, 64, NT, true>(c, b17, L, chi, sc, dm);
            for (int k = tid; k < a.O; k += NT) { odometry_edge<kOdoEval>(c, k, chi, sc); odometry_edge<kOdoDiag>(c, k, chi, sc); }
        } else {
            eval_class<kEval, 4, NT, true>(c, b1, b5, chi, sc, dm);
            eval_class<kEval, 8, NT, true>(c, b5, b9, chi, sc, dm);
            eval_class<kEval, 16, NT, true>(c, b9, b17, chi, sc, dm);
            eval_class<kEval, 64, NT, true>(c, b17, L, chi, sc, dm);
            for (int k = tid; k < a.O; k += NT) odometry_edge<kOdoEval>(c, k, chi, sc);
        }
        __syncthreads();   // (the diagonal's atomics have landed; the records are written)
        if (lm)
            for (int i = tid; i < n; i += NT) dm = fmax(dm, fabs(xs[i]));
        wg_reduce<NT>(red, chi, sc, dm);
        if (tid == 0) {
            ctl.current_chi = ctl.chi2_init = ctl.chi2_final = chi;
            if (s_stop) { ctl.stopped = 1; ctl.done = 1; }
            if (ctl.iters <= 0) ctl.done = 1;
            if (lm && !ctl.done) { ctl.lambda = 1e-5 * dm; ctl.ni = 2; }
        }
        __syncthreads();
    }
    long long* stamps = a.stamps;
    if (stamps && tid == 0) stamps[7] = wall_clock64();       // (the opening pass)
    // ---- the trials
    while (!ctl.done) {
        const double lambda = ctl.lambda;
        c.lambda = lambda;
        if (stamps && tid == 0) stamps[0] = wall_clock64();
        for (int i = tid; i < ntri; i += NT) S[i] = 0.0;
        if (tid == 0) s_fail = 0;
        __syncthreads();
        {
            double chi = 0, sc = 0;
            build_class<4, NT>(c, b1, b5);
            build_class<8, NT>(c, b5, b9);
            build_class<16, NT>(c, b9, b17);
            build_class<64, NT>(c, b17, L);
            for (int k = tid; k < a.O; k += NT) odometry_edge<kOdoBuild>(c, k, chi, sc);
        }
        __syncthreads();
        for (int i = tid; i < n; i += NT) S[tri(i, i)] += lambda;      // setLambda: the damping on the pose diagonal (the landmarks' went into A)
        __syncthreads();
        if (stamps && tid == 0) stamps[1] = wall_clock64();
        factorize<NT>(S, invd, tjj, nf, &s_fail);
        if (stamps && tid == 0) stamps[2] = wall_clock64();
        if (wave == 0) back_substitute(S, invd, nf, xs);
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
        copy_unobserved<NT>(c, b1);
        eval_class<kUpdate, 4, NT>(c, b1, b5, chi, sc, dm);
        eval_class<kUpdate, 8, NT>(c, b5, b9, chi, sc, dm);
        eval_class<kUpdate, 16, NT>(c, b9, b17, chi, sc, dm);
        eval_class<kUpdate, 64, NT>(c, b17, L, chi, sc, dm);
        for (int k = tid; k < a.O; k += NT) odometry_edge<kOdoUpdate>(c, k, chi, sc);
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
    size_t doubles = (size_t)(P + 1) / 2 + 10 * (size_t)P + 2 * n + (size_t)threads * kStageDoubles + (n + 3) * (n + 4) / 2;
    const size_t bytes = doubles * 8;
    // static LDS of the kernel: the controller block, the list of wide landmarks, the reduction scratch
    const size_t fixed = sizeof(BaCtl) + (kWindowMaxDegree + 2 + 18 * 8) * sizeof(int) + 30 * 8 + 128;
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
