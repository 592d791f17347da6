// ORACLE - TEST INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg and tests/).  PARITY UNPINNED (see ba_ref.cpp).
//
// ba_ref_optimize_mt: the ALL-CORE CPU baseline of the SE(2)-XYZ bundle adjustment (BASELINE.md §3
// "OpenMP over all host cores").  Same algorithm and LM policy as ba_ref_optimize (oracle/ba_ref.cpp, which follows
//   /root/reference/src/EdgeSE2XYZ.cpp:61-106, include/se2lam/EdgeSE2XYZ.h:62-102, src/LocalMapper.cpp:232-302 and
//   g2o 20160424 OptimizationAlgorithmLevenberg / BlockSolverX), organised the way a tuned CPU code would run it:
//   * linearisation, Schur products, back-substitution and chi^2 are OpenMP loops over landmarks / pose rows
//     (owner-computes: no atomics, deterministic for a fixed thread count),
//   * the reduced (3P)^2 pose system is factorised by LAPACK dpotrf / dpotrs (the caller passes the function
//     pointers, e.g. scipy.linalg.cython_lapack's OpenBLAS) - standing in for the reference's CHOLMOD the way a
//     threaded supernodal factorisation would run on a 41 %-dense system - or by the scalar LL^T when none is given.
// Results agree with ba_ref_optimize to round-off (tests/test_ba_oracle.py); it is NOT the checker, only the timed
// CPU baseline next to the 1-thread port.
#include <omp.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

extern "C" {
struct ba_ref_problem {   // layout of oracle/ba_ref.cpp
    int32_t P, L, E, O;
    const double* poses; const uint8_t* fixed; const double* lms;
    const int32_t* e_kf; const int32_t* e_lm; const double* e_uv; const double* e_info;
    const int32_t* o_i; const int32_t* o_j; const double* o_meas; const double* o_info;
    double fx, cx, cy, Rbc[9], tbc[3], huber;
};
struct ba_ref_stats {
    int32_t iterations, trials, terminated;
    double chi2_init, chi2_final, lambda_final;
    double chi2_hist[64], lambda_hist[64];
    int32_t trials_hist[64];
    int32_t n_rho;
    double rho_log[256];
};
typedef void (*dpotrf_fn)(char* uplo, int* n, double* a, int* lda, int* info);
typedef void (*dpotrs_fn)(char* uplo, int* n, int* nrhs, double* a, int* lda, double* b, int* ldb, int* info);
}

namespace {
const double kPi = 3.14159265358979323846;
inline double normalize_theta(double t) {
    if (t >= -kPi && t < kPi) return t;
    t -= std::floor(t / (2 * kPi)) * 2 * kPi;
    if (t >= kPi) t -= 2 * kPi;
    if (t < -kPi) t += 2 * kPi;
    return t;
}
struct Cam { double fx, cx, cy, Rcb[9], tcb[3], huber; };

// EdgeSE2XYZ error (+ Jacobians): lc = Rcb Rz(-th) (lw - [x, y, 0]) + tcb, e = f (X/Z, Y/Z) + c - uv
inline void edge(const Cam& cam, const double* ps, const double* lw, const double* uv, double e[2], double* Jp, double* Jl) {
    const double c = std::cos(ps[2]), s = std::sin(ps[2]);
    const double dx = lw[0] - ps[0], dy = lw[1] - ps[1];
    double R[9];
    for (int i = 0; i < 3; ++i) {
        R[i * 3] = cam.Rcb[i * 3] * c - cam.Rcb[i * 3 + 1] * s;
        R[i * 3 + 1] = cam.Rcb[i * 3] * s + cam.Rcb[i * 3 + 1] * c;
        R[i * 3 + 2] = cam.Rcb[i * 3 + 2];
    }
    double lc[3];
    for (int i = 0; i < 3; ++i) lc[i] = R[i * 3] * dx + R[i * 3 + 1] * dy + R[i * 3 + 2] * lw[2] + cam.tcb[i];
    const double zi = 1.0 / lc[2];
    e[0] = cam.fx * lc[0] * zi + cam.cx - uv[0];
    e[1] = cam.fx * lc[1] * zi + cam.cy - uv[1];
    if (!Jp) return;
    const double zi2 = zi * zi, j00 = cam.fx * zi, j02 = -cam.fx * lc[0] * zi2, j12 = -cam.fx * lc[1] * zi2;
    for (int k = 0; k < 3; ++k) {
        Jl[k] = j00 * R[k] + j02 * R[6 + k];
        Jl[3 + k] = j00 * R[3 + k] + j12 * R[6 + k];
    }
    Jp[0] = -Jl[0]; Jp[1] = -Jl[1]; Jp[2] = Jl[0] * dy - Jl[1] * dx;
    Jp[3] = -Jl[3]; Jp[4] = -Jl[4]; Jp[5] = Jl[3] * dy - Jl[4] * dx;
}
inline void huber(double e2, double d, double& r0, double& r1) {
    if (e2 <= d * d) { r0 = e2; r1 = 1; } else { const double q = std::sqrt(e2); r0 = 2 * q * d - d * d; r1 = d / q; }
}
inline void pre_se2(const double* pi, const double* pj, const double* z, double e[3], double* A, double* B) {
    const double c = std::cos(pi[2]), s = std::sin(pi[2]), rx = pj[0] - pi[0], ry = pj[1] - pi[1];
    e[0] = c * rx + s * ry - z[0]; e[1] = -s * rx + c * ry - z[1]; e[2] = pj[2] - pi[2] - z[2];
    if (!A) return;
    std::memset(A, 0, 72); std::memset(B, 0, 72);
    A[0] = -c; A[1] = -s; A[3] = s; A[4] = -c; A[2] = -(c * -ry + s * rx); A[5] = -(-s * -ry + c * rx); A[8] = -1;
    B[0] = c; B[1] = s; B[3] = -s; B[4] = c; B[8] = 1;
}

struct Work {
    const ba_ref_problem* p; Cam cam; int n;
    std::vector<int> lm_ptr, lm_e, ps_ptr, ps_e;          // landmark -> edges, pose -> edges
    std::vector<double> Hpl, Hppe, bpe, Hll, bl, Dinv, z, Y;  // per edge 9 / 9 / 3, per landmark 9 / 3 / 9 / 3, per edge 9
    std::vector<double> Hodo, bodo;                       // pose-pose odometry part, dense n*n / n (tiny work: serial)
    std::vector<double> S, bs, bp, xp, xl;
};

double chi2(const Work& w, const double* poses, const double* lms) {
    const ba_ref_problem& p = *w.p;
    double chi = 0;
#pragma omp parallel for reduction(+ : chi) schedule(static)
    for (int k = 0; k < p.E; ++k) {
        double e[2], r0, r1;
        edge(w.cam, poses + 3 * p.e_kf[k], lms + 3 * (size_t)p.e_lm[k], p.e_uv + 2 * (size_t)k, e, nullptr, nullptr);
        const double* q = p.e_info + 3 * (size_t)k;
        huber(e[0] * (q[0] * e[0] + q[1] * e[1]) + e[1] * (q[1] * e[0] + q[2] * e[1]), w.cam.huber, r0, r1);
        chi += r0;
    }
    for (int k = 0; k < p.O; ++k) {
        double e[3];
        pre_se2(poses + 3 * p.o_i[k], poses + 3 * p.o_j[k], p.o_meas + 3 * k, e, nullptr, nullptr);
        const double* W = p.o_info + 9 * k;
        for (int r = 0; r < 3; ++r) chi += e[r] * (W[r * 3] * e[0] + W[r * 3 + 1] * e[1] + W[r * 3 + 2] * e[2]);
    }
    return chi;
}

void linearize(Work& w, const double* poses, const double* lms) {
    const ba_ref_problem& p = *w.p;
    const int n = w.n;
#pragma omp parallel for schedule(dynamic, 64)
    for (int l = 0; l < p.L; ++l) {
        double H[9] = {0}, b[3] = {0};
        for (int t = w.lm_ptr[l]; t < w.lm_ptr[l + 1]; ++t) {
            const int k = w.lm_e[t], kf = p.e_kf[k];
            double e[2], Jp[6], Jl[6], r0, r1;
            edge(w.cam, poses + 3 * kf, lms + 3 * (size_t)l, p.e_uv + 2 * (size_t)k, e, Jp, Jl);
            const double* q = p.e_info + 3 * (size_t)k;
            const double we0 = q[0] * e[0] + q[1] * e[1], we1 = q[1] * e[0] + q[2] * e[1];
            huber(e[0] * we0 + e[1] * we1, w.cam.huber, r0, r1);
            const double W0 = r1 * q[0], W1 = r1 * q[1], W2 = r1 * q[2], o0 = -r1 * we0, o1 = -r1 * we1;
            double WJl[6], WJp[6];
            for (int c = 0; c < 3; ++c) {
                WJl[c] = W0 * Jl[c] + W1 * Jl[3 + c]; WJl[3 + c] = W1 * Jl[c] + W2 * Jl[3 + c];
                WJp[c] = W0 * Jp[c] + W1 * Jp[3 + c]; WJp[3 + c] = W1 * Jp[c] + W2 * Jp[3 + c];
            }
            const bool fr = !p.fixed[kf];
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) {
                    H[r * 3 + c] += Jl[r] * WJl[c] + Jl[3 + r] * WJl[3 + c];
                    w.Hpl[(size_t)k * 9 + r * 3 + c] = fr ? Jp[r] * WJl[c] + Jp[3 + r] * WJl[3 + c] : 0.0;
                    w.Hppe[(size_t)k * 9 + r * 3 + c] = fr ? Jp[r] * WJp[c] + Jp[3 + r] * WJp[3 + c] : 0.0;
                }
                b[r] += Jl[r] * o0 + Jl[3 + r] * o1;
                w.bpe[(size_t)k * 3 + r] = fr ? Jp[r] * o0 + Jp[3 + r] * o1 : 0.0;
            }
        }
        std::memcpy(&w.Hll[(size_t)l * 9], H, 72);
        std::memcpy(&w.bl[(size_t)l * 3], b, 24);
    }
    std::fill(w.Hodo.begin(), w.Hodo.end(), 0.0);
    std::fill(w.bodo.begin(), w.bodo.end(), 0.0);
    for (int k = 0; k < p.O; ++k) {
        const int i = p.o_i[k], j = p.o_j[k];
        double e[3], A[9], B[9], WA[9], WB[9], om[3];
        pre_se2(poses + 3 * i, poses + 3 * j, p.o_meas + 3 * k, e, A, B);
        const double* W = p.o_info + 9 * k;
        for (int r = 0; r < 3; ++r) {
            om[r] = -(W[r * 3] * e[0] + W[r * 3 + 1] * e[1] + W[r * 3 + 2] * e[2]);
            for (int c = 0; c < 3; ++c) {
                WA[r * 3 + c] = W[r * 3] * A[c] + W[r * 3 + 1] * A[3 + c] + W[r * 3 + 2] * A[6 + c];
                WB[r * 3 + c] = W[r * 3] * B[c] + W[r * 3 + 1] * B[3 + c] + W[r * 3 + 2] * B[6 + c];
            }
        }
        const bool fi = !p.fixed[i], fj = !p.fixed[j];
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) {
                const double aa = A[r] * WA[c] + A[3 + r] * WA[3 + c] + A[6 + r] * WA[6 + c];
                const double ab = A[r] * WB[c] + A[3 + r] * WB[3 + c] + A[6 + r] * WB[6 + c];
                const double bb = B[r] * WB[c] + B[3 + r] * WB[3 + c] + B[6 + r] * WB[6 + c];
                if (fi) w.Hodo[(size_t)(3 * i + r) * n + 3 * i + c] += aa;
                if (fj) w.Hodo[(size_t)(3 * j + r) * n + 3 * j + c] += bb;
                if (fi && fj) { w.Hodo[(size_t)(3 * i + r) * n + 3 * j + c] += ab; w.Hodo[(size_t)(3 * j + c) * n + 3 * i + r] += ab; }
            }
            if (fi) w.bodo[3 * i + r] += A[r] * om[0] + A[3 + r] * om[1] + A[6 + r] * om[2];
            if (fj) w.bodo[3 * j + r] += B[r] * om[0] + B[3 + r] * om[1] + B[6 + r] * om[2];
        }
    }
}

inline void inv3(const double* M, double lambda, double* Mi) {
    const double a = M[0] + lambda, b = M[1], c = M[2], d = M[3], e = M[4] + lambda, f = M[5], g = M[6], h = M[7], i = M[8] + lambda;
    const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g, id = 1.0 / (a * A + b * B + c * C);
    Mi[0] = A * id; Mi[1] = -(b * i - c * h) * id; Mi[2] = (b * f - c * e) * id;
    Mi[3] = B * id; Mi[4] = (a * i - c * g) * id; Mi[5] = -(a * f - c * d) * id;
    Mi[6] = C * id; Mi[7] = -(a * h - b * g) * id; Mi[8] = (a * e - b * d) * id;
}

void schur(Work& w, double lambda) {
    const ba_ref_problem& p = *w.p;
    const int n = w.n;
#pragma omp parallel for schedule(dynamic, 64)
    for (int l = 0; l < p.L; ++l) {   // Dinv, z = Dinv bl, Y_e = Hpl_e Dinv
        double* Di = &w.Dinv[(size_t)l * 9];
        inv3(&w.Hll[(size_t)l * 9], lambda, Di);
        const double* b = &w.bl[(size_t)l * 3];
        for (int r = 0; r < 3; ++r) w.z[(size_t)l * 3 + r] = Di[r * 3] * b[0] + Di[r * 3 + 1] * b[1] + Di[r * 3 + 2] * b[2];
        for (int t = w.lm_ptr[l]; t < w.lm_ptr[l + 1]; ++t) {
            const int k = w.lm_e[t];
            const double* B = &w.Hpl[(size_t)k * 9];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c)
                    w.Y[(size_t)k * 9 + r * 3 + c] = B[r * 3] * Di[c] + B[r * 3 + 1] * Di[3 + c] + B[r * 3 + 2] * Di[6 + c];
        }
    }
#pragma omp parallel for schedule(dynamic, 1)
    for (int a = 0; a < p.P; ++a) {   // pose block row a of S and b_s: owner computes
        double* Srow = &w.S[(size_t)3 * a * n];
        std::memcpy(Srow, &w.Hodo[(size_t)3 * a * n], (size_t)3 * n * sizeof(double));
        double b[3] = {w.bodo[3 * a], w.bodo[3 * a + 1], w.bodo[3 * a + 2]}, gz[3] = {0, 0, 0};
        if (p.fixed[a]) {
            std::memset(Srow, 0, (size_t)3 * n * sizeof(double));
            for (int r = 0; r < 3; ++r) { Srow[(size_t)r * n + 3 * a + r] = 1.0; w.bs[3 * a + r] = 0; w.bp[3 * a + r] = 0; }
            continue;
        }
        for (int t = w.ps_ptr[a]; t < w.ps_ptr[a + 1]; ++t) {
            const int k = w.ps_e[t], l = p.e_lm[k];
            const double* Yk = &w.Y[(size_t)k * 9];
            const double* zl = &w.z[(size_t)l * 3];
            for (int r = 0; r < 3; ++r) {
                b[r] += w.bpe[(size_t)k * 3 + r];
                gz[r] += w.Hpl[(size_t)k * 9 + r * 3] * zl[0] + w.Hpl[(size_t)k * 9 + r * 3 + 1] * zl[1] + w.Hpl[(size_t)k * 9 + r * 3 + 2] * zl[2];
                for (int c = 0; c < 3; ++c) Srow[(size_t)r * n + 3 * a + c] += w.Hppe[(size_t)k * 9 + r * 3 + c];
            }
            for (int u = w.lm_ptr[l]; u < w.lm_ptr[l + 1]; ++u) {
                const int k2 = w.lm_e[u], b2 = p.e_kf[k2];
                if (p.fixed[b2]) continue;
                const double* B = &w.Hpl[(size_t)k2 * 9];
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c)
                        Srow[(size_t)r * n + 3 * b2 + c] -= Yk[r * 3] * B[c * 3] + Yk[r * 3 + 1] * B[c * 3 + 1] + Yk[r * 3 + 2] * B[c * 3 + 2];
            }
        }
        for (int r = 0; r < 3; ++r) {
            Srow[(size_t)r * n + 3 * a + r] += lambda;
            w.bp[3 * a + r] = b[r];
            w.bs[3 * a + r] = b[r] - gz[r];
        }
    }
    for (int k = 0; k < p.P; ++k)   // columns of fixed poses
        if (p.fixed[k])
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < n; ++c)
                    if (c != 3 * k + r) w.S[(size_t)c * n + 3 * k + r] = 0;
}

bool scalar_cholesky_solve(std::vector<double>& A, int n, std::vector<double>& x) {
    for (int j = 0; j < n; ++j) {
        double* Aj = &A[(size_t)j * n];
        double d = Aj[j];
        for (int k = 0; k < j; ++k) d -= Aj[k] * Aj[k];
        if (!(d > 0.0)) return false;
        d = std::sqrt(d); Aj[j] = d;
#pragma omp parallel for schedule(static)
        for (int i = j + 1; i < n; ++i) {
            double* Ai = &A[(size_t)i * n];
            double s = Ai[j];
            for (int k = 0; k < j; ++k) s -= Ai[k] * Aj[k];
            Ai[j] = s / d;
        }
    }
    for (int i = 0; i < n; ++i) { double s = x[i]; for (int k = 0; k < i; ++k) s -= A[(size_t)i * n + k] * x[k]; x[i] = s / A[(size_t)i * n + i]; }
    for (int i = n - 1; i >= 0; --i) { double s = x[i]; for (int k = i + 1; k < n; ++k) s -= A[(size_t)k * n + i] * x[k]; x[i] = s / A[(size_t)i * n + i]; }
    return true;
}
}  // namespace

extern "C" {

int ba_ref_mt_threads(void) { return omp_get_max_threads(); }

// threads <= 0: all cores.  dpotrf / dpotrs may be NULL (scalar LL^T with an OpenMP panel loop instead).
int ba_ref_optimize_mt(const ba_ref_problem* pp, int iters, int threads, void* dpotrf_ptr, void* dpotrs_ptr,
                       double* poses_out, double* lms_out, ba_ref_stats* stats) {
    const ba_ref_problem& p = *pp;
    if (threads > 0) omp_set_num_threads(threads);
    Work w;
    w.p = pp;
    w.n = 3 * p.P;
    const int n = w.n;
    w.cam.fx = p.fx; w.cam.cx = p.cx; w.cam.cy = p.cy; w.cam.huber = p.huber;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) w.cam.Rcb[i * 3 + j] = p.Rbc[j * 3 + i];
    for (int i = 0; i < 3; ++i)
        w.cam.tcb[i] = -(w.cam.Rcb[i * 3] * p.tbc[0] + w.cam.Rcb[i * 3 + 1] * p.tbc[1] + w.cam.Rcb[i * 3 + 2] * p.tbc[2]);
    auto csr = [&](const int32_t* key, int nk, std::vector<int>& ptr, std::vector<int>& idx) {
        ptr.assign(nk + 1, 0);
        for (int k = 0; k < p.E; ++k) ptr[key[k] + 1]++;
        for (int i = 0; i < nk; ++i) ptr[i + 1] += ptr[i];
        idx.resize(p.E);
        std::vector<int> f(ptr.begin(), ptr.end() - 1);
        for (int k = 0; k < p.E; ++k) idx[f[key[k]]++] = k;
    };
    csr(p.e_lm, p.L, w.lm_ptr, w.lm_e);
    csr(p.e_kf, p.P, w.ps_ptr, w.ps_e);
    w.Hpl.resize((size_t)p.E * 9); w.Hppe.resize((size_t)p.E * 9); w.bpe.resize((size_t)p.E * 3); w.Y.resize((size_t)p.E * 9);
    w.Hll.resize((size_t)p.L * 9); w.bl.resize((size_t)p.L * 3); w.Dinv.resize((size_t)p.L * 9); w.z.resize((size_t)p.L * 3);
    w.Hodo.resize((size_t)n * n); w.bodo.resize(n); w.S.resize((size_t)n * n); w.bs.resize(n); w.bp.resize(n);
    w.xp.resize(n); w.xl.resize((size_t)p.L * 3);
    std::vector<double> poses(p.poses, p.poses + 3 * p.P), lms(p.lms, p.lms + 3 * (size_t)p.L), tp(poses), tl(lms), F;
    ba_ref_stats s;
    std::memset(&s, 0, sizeof(s));
    double lambda = 0, ni = 2;
    s.chi2_init = s.chi2_final = chi2(w, poses.data(), lms.data());
    double currentChi = s.chi2_init;
    bool ok = true;
    for (int it = 0; it < iters && ok; ++it) {
        linearize(w, poses.data(), lms.data());
        if (it == 0) {
            double maxd = 0;
            for (int a = 0; a < p.P; ++a) {
                if (p.fixed[a]) continue;
                double d[3] = {w.Hodo[(size_t)(3 * a) * n + 3 * a], w.Hodo[(size_t)(3 * a + 1) * n + 3 * a + 1], w.Hodo[(size_t)(3 * a + 2) * n + 3 * a + 2]};
                for (int t = w.ps_ptr[a]; t < w.ps_ptr[a + 1]; ++t)
                    for (int r = 0; r < 3; ++r) d[r] += w.Hppe[(size_t)w.ps_e[t] * 9 + r * 4];
                for (int r = 0; r < 3; ++r) maxd = std::max(maxd, std::fabs(d[r]));
            }
            for (int l = 0; l < p.L; ++l)
                for (int r = 0; r < 3; ++r) maxd = std::max(maxd, std::fabs(w.Hll[(size_t)l * 9 + r * 4]));
            lambda = 1e-5 * maxd;
            ni = 2;
        }
        double rho = 0;
        int qmax = 0;
        do {
            schur(w, lambda);
            w.xp = w.bs;
            bool ok2 = true;
            if (dpotrf_ptr && dpotrs_ptr) {
                F = w.S;   // symmetric: row-major == column-major
                char lo = 'L';
                int nn = n, one = 1, info = 0;
                ((dpotrf_fn)dpotrf_ptr)(&lo, &nn, F.data(), &nn, &info);
                ok2 = info == 0;
                if (ok2) ((dpotrs_fn)dpotrs_ptr)(&lo, &nn, &one, F.data(), &nn, w.xp.data(), &nn, &info);
            } else {
                F = w.S;
                ok2 = scalar_cholesky_solve(F, n, w.xp);
            }
            if (!ok2) std::fill(w.xp.begin(), w.xp.end(), 0.0);
            double scale = 0;
#pragma omp parallel for reduction(+ : scale) schedule(dynamic, 64)
            for (int l = 0; l < p.L; ++l) {   // x_l = z - Y^T x_p ; trial landmark
                double x[3] = {w.z[(size_t)l * 3], w.z[(size_t)l * 3 + 1], w.z[(size_t)l * 3 + 2]};
                for (int t = w.lm_ptr[l]; t < w.lm_ptr[l + 1]; ++t) {
                    const int k = w.lm_e[t], a = p.e_kf[k];
                    const double* y = &w.Y[(size_t)k * 9];
                    for (int c = 0; c < 3; ++c) x[c] -= y[c] * w.xp[3 * a] + y[3 + c] * w.xp[3 * a + 1] + y[6 + c] * w.xp[3 * a + 2];
                }
                for (int c = 0; c < 3; ++c) {
                    tl[(size_t)l * 3 + c] = lms[(size_t)l * 3 + c] + x[c];
                    scale += x[c] * (lambda * x[c] + w.bl[(size_t)l * 3 + c]);
                }
            }
            for (int a = 0; a < p.P; ++a) {
                for (int r = 0; r < 3; ++r) tp[3 * a + r] = poses[3 * a + r];
                if (p.fixed[a]) continue;
                tp[3 * a] += w.xp[3 * a]; tp[3 * a + 1] += w.xp[3 * a + 1];
                tp[3 * a + 2] = normalize_theta(tp[3 * a + 2] + w.xp[3 * a + 2]);
                for (int r = 0; r < 3; ++r) scale += w.xp[3 * a + r] * (lambda * w.xp[3 * a + r] + w.bp[3 * a + r]);
            }
            double tempChi = chi2(w, tp.data(), tl.data());
            if (!ok2) tempChi = std::numeric_limits<double>::max();
            ++s.trials; ++qmax;
            rho = (currentChi - tempChi) / (scale + 1e-3);
            if (s.n_rho < 256) s.rho_log[s.n_rho++] = rho;
            if (rho > 0 && std::isfinite(tempChi)) {
                double alpha = 1. - std::pow(2 * rho - 1, 3);
                alpha = std::min(alpha, 2. / 3.);
                lambda *= std::max(1. / 3., alpha);
                ni = 2;
                currentChi = tempChi;
                poses.swap(tp); lms.swap(tl);
            } else {
                lambda *= ni; ni *= 2;
            }
        } while (rho < 0 && qmax < 10);
        if (it < 64) { s.chi2_hist[it] = currentChi; s.lambda_hist[it] = lambda; s.trials_hist[it] = qmax; }
        s.iterations = it + 1;
        s.chi2_final = currentChi;
        if (qmax == 10 || rho == 0) { s.terminated = 1; ok = false; }
    }
    s.lambda_final = lambda;
    if (poses_out) std::memcpy(poses_out, poses.data(), poses.size() * sizeof(double));
    if (lms_out) std::memcpy(lms_out, lms.data(), lms.size() * sizeof(double));
    if (stats) *stats = s;
    return 0;
}

}  // extern "C"
