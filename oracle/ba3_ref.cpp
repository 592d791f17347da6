// ORACLE - TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py's cpu_baseline leg).  PARITY: the GRAPH and its cost are pinned
// against the reference's own code - Map::loadLocalGraph(optimizer, vpEdgesAll, vnAllIdx), src/optimizer.cpp and EdgeSE3ExpmapPrior
// compiled in oracle/_ref, g2o's two edge types written into the stand-in from their published definitions: the robust cost of
// the graph the reference builds equals ba3_ref_chi2 to 1e-9 (tests/test_ref_compiled.py).  The SOLVER stays UNPINNED: g2o
// cannot be built here; its iterations are a restatement, checked against an independent numpy / scipy
// model (tests/test_ba3_oracle.py).
//
// ba3_ref: the marginalising SE3-expmap local bundle adjustment of the reference - SURVEY.md section 8(f).2:
//   Map::loadLocalGraph(optimizer, vpEdgesAll, vnAllIdx)    /root/reference/src/Map.cpp:414-566
//       VertexSE3Expmap per key frame (Tcw; local ones free except the oldest / id 1, reference ones fixed)  :440-485
//       EdgeSE3ExpmapPrior per local key frame = addPlaneMotionSE3Expmap                                     :445, optimizer.cpp:236-314, 159-197
//       EdgeSE3Expmap between consecutive key frames (mOdoMeasureFrom)                                      :449-468
//       VertexSBAPointXYZ (marginalised) + EdgeProjectXYZ2UV, information invSigma2 * I, Huber(TH_HUBER)     :488-555
//   LocalMapper::removeOutlierChi2: LM optimize(10), then chi2() of every projection edge against 25         src/LocalMapper.cpp:172-230
// [3P g2o 20160424, restated from memory]: types_six_dof_expmap (VertexSE3Expmap::oplusImpl = exp(update) * estimate,
// EdgeProjectXYZ2UV::computeError / linearizeOplus, EdgeSE3Expmap::computeError = log(T_j^-1 C T_i) with Jacobians
// adj(T_j^-1 C) and -adj(T_i^-1 C^-1)), BlockSolverX Schur complement over the marginalised points, the Levenberg policy
// of ba_ref.cpp.  Dense LL^T stands in for CHOLMOD.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "se3_ref.h"

extern "C" {

struct ba3_problem {
    int32_t P, L, E, O;
    const double* poses;       // P x 12 (Tcw: R row-major, t)
    const uint8_t* fixed;      // P
    const double* lms;         // L x 3
    const int32_t* e_kf;       // E  pose of the observation
    const int32_t* e_lm;       // E  landmark
    const double* e_uv;        // E x 2
    const double* e_w;         // E  invSigma2 (information = w * I)
    const uint8_t* has_prior;  // P  EdgeSE3ExpmapPrior present
    const double* prior_meas;  // P x 12
    const double* prior_info;  // P x 36
    const int32_t* o_i;        // O  vertex 0 of the EdgeSE3Expmap
    const int32_t* o_j;        // O  vertex 1
    const double* o_meas;      // O x 12
    const double* o_info;      // O x 36
    double f, cx, cy, huber;
};

struct ba_ref_stats {   // layout of oracle/ba_ref.cpp
    int32_t iterations, trials, terminated;
    double chi2_init, chi2_final, lambda_final;
    double chi2_hist[64], lambda_hist[64];
    int32_t trials_hist[64];
    int32_t n_rho;
    double rho_log[256];
};
}

namespace {

struct St {
    std::vector<Se3> T;
    std::vector<double> X;
};

inline void huber(double e2, double d, double& r0, double& r1) {
    if (e2 <= d * d) { r0 = e2; r1 = 1; } else { const double q = std::sqrt(e2); r0 = 2 * q * d - d * d; r1 = d / q; }
}

// EdgeProjectXYZ2UV: e = obs - cam_map(T X); Jl (2x3, wrt the point) and Jp (2x6, wrt the pose update) as in g2o
inline void proj_edge(const ba3_problem& p, const Se3& T, const double* X, const double* uv, double e[2], double* Jp, double* Jl) {
    const double x = T.R[0] * X[0] + T.R[1] * X[1] + T.R[2] * X[2] + T.t[0];
    const double y = T.R[3] * X[0] + T.R[4] * X[1] + T.R[5] * X[2] + T.t[1];
    const double z = T.R[6] * X[0] + T.R[7] * X[1] + T.R[8] * X[2] + T.t[2];
    e[0] = uv[0] - (x / z * p.f + p.cx);
    e[1] = uv[1] - (y / z * p.f + p.cy);
    if (!Jp) return;
    const double z2 = z * z, f = p.f;
    const double tmp[6] = {f, 0, -x / z * f, 0, f, -y / z * f};
    for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 3; ++c)
            Jl[3 * r + c] = -1. / z * (tmp[3 * r] * T.R[c] + tmp[3 * r + 1] * T.R[3 + c] + tmp[3 * r + 2] * T.R[6 + c]);
    Jp[0] = x * y / z2 * f; Jp[1] = -(1 + (x * x / z2)) * f; Jp[2] = y / z * f; Jp[3] = -1. / z * f; Jp[4] = 0; Jp[5] = x / z2 * f;
    Jp[6] = (1 + y * y / z2) * f; Jp[7] = -x * y / z2 * f; Jp[8] = -x / z * f; Jp[9] = 0; Jp[10] = -1. / z * f; Jp[11] = y / z2 * f;
}

inline void odo_edge(const Se3& Ti, const Se3& Tj, const Se3& C, double e[6], double* Ji, double* Jj) {
    const Se3 TjinvC = se3_mul(se3_inv(Tj), C);
    se3_log(se3_mul(TjinvC, Ti), e);
    if (!Ji) return;
    se3_adj(TjinvC, Ji);
    double A[36];
    se3_adj(se3_mul(se3_inv(Ti), se3_inv(C)), A);
    for (int i = 0; i < 36; ++i) Jj[i] = -A[i];
}

double chi2_all(const ba3_problem& p, const St& s, double* edge_chi2) {
    double chi = 0;
    for (int k = 0; k < p.E; ++k) {
        double e[2], r0, r1;
        proj_edge(p, s.T[p.e_kf[k]], &s.X[3 * (size_t)p.e_lm[k]], p.e_uv + 2 * (size_t)k, e, nullptr, nullptr);
        const double c2 = p.e_w[k] * (e[0] * e[0] + e[1] * e[1]);
        if (edge_chi2) edge_chi2[k] = c2;
        huber(c2, p.huber, r0, r1);
        chi += r0;
    }
    for (int a = 0; a < p.P; ++a) {
        if (!p.has_prior[a]) continue;
        double e[6];
        se3_log(se3_mul(se3_from(p.prior_meas + 12 * (size_t)a), se3_inv(s.T[a])), e);
        const double* W = p.prior_info + 36 * (size_t)a;
        for (int r = 0; r < 6; ++r) {
            double v = 0;
            for (int c = 0; c < 6; ++c) v += W[6 * r + c] * e[c];
            chi += e[r] * v;
        }
    }
    for (int k = 0; k < p.O; ++k) {
        double e[6];
        odo_edge(s.T[p.o_i[k]], s.T[p.o_j[k]], se3_from(p.o_meas + 12 * (size_t)k), e, nullptr, nullptr);
        const double* W = p.o_info + 36 * (size_t)k;
        for (int r = 0; r < 6; ++r) {
            double v = 0;
            for (int c = 0; c < 6; ++c) v += W[6 * r + c] * e[c];
            chi += e[r] * v;
        }
    }
    return chi;
}

struct Sys {
    int n;
    std::vector<double> Hpp, bp, Hll, bl, Hpl;   // (6P)^2, 6P, 9L, 3L, 18E (6x3 per edge, zero for fixed poses)
    std::vector<int> lm_ptr, lm_e;
};

void build(const ba3_problem& p, const St& s, Sys& y) {
    const int n = 6 * p.P;
    y.n = n;
    y.Hpp.assign((size_t)n * n, 0.0); y.bp.assign(n, 0.0);
    y.Hll.assign((size_t)p.L * 9, 0.0); y.bl.assign((size_t)p.L * 3, 0.0); y.Hpl.assign((size_t)p.E * 18, 0.0);
    for (int k = 0; k < p.E; ++k) {
        const int a = p.e_kf[k], l = p.e_lm[k];
        double e[2], Jp[12], Jl[6], r0, r1;
        proj_edge(p, s.T[a], &s.X[3 * (size_t)l], p.e_uv + 2 * (size_t)k, e, Jp, Jl);
        const double w = p.e_w[k];
        huber(w * (e[0] * e[0] + e[1] * e[1]), p.huber, r0, r1);
        const double W = r1 * w;                 // weightedOmega = rho1 * w * I
        const double o0 = -W * e[0], o1 = -W * e[1];   // omega_r = -Omega e rho1
        double* Hl = &y.Hll[(size_t)l * 9];
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) Hl[3 * r + c] += W * (Jl[r] * Jl[c] + Jl[3 + r] * Jl[3 + c]);
            y.bl[(size_t)l * 3 + r] += Jl[r] * o0 + Jl[3 + r] * o1;
        }
        if (p.fixed[a]) continue;
        for (int r = 0; r < 6; ++r) {
            for (int c = 0; c < 6; ++c) y.Hpp[(size_t)(6 * a + r) * n + 6 * a + c] += W * (Jp[r] * Jp[c] + Jp[6 + r] * Jp[6 + c]);
            for (int c = 0; c < 3; ++c) y.Hpl[(size_t)k * 18 + 3 * r + c] = W * (Jp[r] * Jl[c] + Jp[6 + r] * Jl[3 + c]);
            y.bp[6 * a + r] += Jp[r] * o0 + Jp[6 + r] * o1;
        }
    }
    for (int a = 0; a < p.P; ++a) {   // EdgeSE3ExpmapPrior: J = -I  ->  H += Omega, b += Omega e
        if (!p.has_prior[a] || p.fixed[a]) continue;
        double e[6];
        se3_log(se3_mul(se3_from(p.prior_meas + 12 * (size_t)a), se3_inv(s.T[a])), e);
        const double* W = p.prior_info + 36 * (size_t)a;
        for (int r = 0; r < 6; ++r) {
            double v = 0;
            for (int c = 0; c < 6; ++c) { v += W[6 * r + c] * e[c]; y.Hpp[(size_t)(6 * a + r) * n + 6 * a + c] += W[6 * r + c]; }
            y.bp[6 * a + r] += v;
        }
    }
    for (int k = 0; k < p.O; ++k) {
        const int i = p.o_i[k], j = p.o_j[k];
        double e[6], Ji[36], Jj[36], WJi[36], WJj[36], We[6];
        odo_edge(s.T[i], s.T[j], se3_from(p.o_meas + 12 * (size_t)k), e, Ji, Jj);
        const double* W = p.o_info + 36 * (size_t)k;
        for (int r = 0; r < 6; ++r) {
            We[r] = 0;
            for (int c = 0; c < 6; ++c) {
                We[r] += W[6 * r + c] * e[c];
                double a = 0, b = 0;
                for (int q = 0; q < 6; ++q) { a += W[6 * r + q] * Ji[6 * q + c]; b += W[6 * r + q] * Jj[6 * q + c]; }
                WJi[6 * r + c] = a; WJj[6 * r + c] = b;
            }
        }
        const bool fi = !p.fixed[i], fj = !p.fixed[j];
        for (int r = 0; r < 6; ++r) {
            for (int c = 0; c < 6; ++c) {
                double ii = 0, jj = 0, ij = 0;
                for (int q = 0; q < 6; ++q) {
                    ii += Ji[6 * q + r] * WJi[6 * q + c];
                    jj += Jj[6 * q + r] * WJj[6 * q + c];
                    ij += Ji[6 * q + r] * WJj[6 * q + c];
                }
                if (fi) y.Hpp[(size_t)(6 * i + r) * n + 6 * i + c] += ii;
                if (fj) y.Hpp[(size_t)(6 * j + r) * n + 6 * j + c] += jj;
                if (fi && fj) { y.Hpp[(size_t)(6 * i + r) * n + 6 * j + c] += ij; y.Hpp[(size_t)(6 * j + c) * n + 6 * i + r] += ij; }
            }
            double bi = 0, bj = 0;
            for (int q = 0; q < 6; ++q) { bi += Ji[6 * q + r] * We[q]; bj += Jj[6 * q + r] * We[q]; }
            if (fi) y.bp[6 * i + r] -= bi;
            if (fj) y.bp[6 * j + r] -= bj;
        }
    }
}

inline void inv3(const double* M, double lam, double* Mi) {
    const double a = M[0] + lam, b = M[1], c = M[2], d = M[3], e = M[4] + lam, f = M[5], g = M[6], h = M[7], i = M[8] + lam;
    const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g, id = 1.0 / (a * A + b * B + c * C);
    Mi[0] = A * id; Mi[1] = -(b * i - c * h) * id; Mi[2] = (b * f - c * e) * id;
    Mi[3] = B * id; Mi[4] = (a * i - c * g) * id; Mi[5] = -(a * f - c * d) * id;
    Mi[6] = C * id; Mi[7] = -(a * h - b * g) * id; Mi[8] = (a * e - b * d) * id;
}

void schur(const ba3_problem& p, const Sys& y, double lam, std::vector<double>& S, std::vector<double>& bs, std::vector<double>& Dinv) {
    const int n = y.n;
    S = y.Hpp; bs = y.bp;
    for (int i = 0; i < n; ++i) S[(size_t)i * n + i] += lam;
    Dinv.assign((size_t)p.L * 9, 0.0);
    for (int l = 0; l < p.L; ++l) {
        double* Di = &Dinv[(size_t)l * 9];
        inv3(&y.Hll[(size_t)l * 9], lam, Di);
        const double* b = &y.bl[(size_t)l * 3];
        const double db[3] = {Di[0] * b[0] + Di[1] * b[1] + Di[2] * b[2], Di[3] * b[0] + Di[4] * b[1] + Di[5] * b[2],
                              Di[6] * b[0] + Di[7] * b[1] + Di[8] * b[2]};
        for (int u = y.lm_ptr[l]; u < y.lm_ptr[l + 1]; ++u) {
            const int ea = y.lm_e[u], pa = p.e_kf[ea];
            if (p.fixed[pa]) continue;
            const double* Ba = &y.Hpl[(size_t)ea * 18];
            double BD[18];
            for (int r = 0; r < 6; ++r)
                for (int c = 0; c < 3; ++c) BD[3 * r + c] = Ba[3 * r] * Di[c] + Ba[3 * r + 1] * Di[3 + c] + Ba[3 * r + 2] * Di[6 + c];
            for (int r = 0; r < 6; ++r) bs[6 * pa + r] -= Ba[3 * r] * db[0] + Ba[3 * r + 1] * db[1] + Ba[3 * r + 2] * db[2];
            for (int v = y.lm_ptr[l]; v < y.lm_ptr[l + 1]; ++v) {
                const int eb = y.lm_e[v], pb = p.e_kf[eb];
                if (p.fixed[pb]) continue;
                const double* Bb = &y.Hpl[(size_t)eb * 18];
                for (int r = 0; r < 6; ++r)
                    for (int c = 0; c < 6; ++c)
                        S[(size_t)(6 * pa + r) * n + 6 * pb + c] -= BD[3 * r] * Bb[3 * c] + BD[3 * r + 1] * Bb[3 * c + 1] + BD[3 * r + 2] * Bb[3 * c + 2];
            }
        }
    }
    for (int a = 0; a < p.P; ++a)
        if (p.fixed[a])
            for (int r = 0; r < 6; ++r) {
                const int i = 6 * a + r;
                for (int c = 0; c < n; ++c) { S[(size_t)i * n + c] = 0; S[(size_t)c * n + i] = 0; }
                S[(size_t)i * n + i] = 1.0;
                bs[i] = 0.0;
            }
}

bool chol_solve(std::vector<double>& A, int n, std::vector<double>& x) {
    for (int j = 0; j < n; ++j) {
        double* Aj = &A[(size_t)j * n];
        double d = Aj[j];
        for (int k = 0; k < j; ++k) d -= Aj[k] * Aj[k];
        if (!(d > 0.0)) return false;
        d = std::sqrt(d); Aj[j] = d;
        for (int i = j + 1; i < n; ++i) {
            double* Ai = &A[(size_t)i * n];
            double v = Ai[j];
            for (int k = 0; k < j; ++k) v -= Ai[k] * Aj[k];
            Ai[j] = v / d;
        }
    }
    for (int i = 0; i < n; ++i) { double v = x[i]; for (int k = 0; k < i; ++k) v -= A[(size_t)i * n + k] * x[k]; x[i] = v / A[(size_t)i * n + i]; }
    for (int i = n - 1; i >= 0; --i) { double v = x[i]; for (int k = i + 1; k < n; ++k) v -= A[(size_t)k * n + i] * x[k]; x[i] = v / A[(size_t)i * n + i]; }
    return true;
}

}  // namespace

extern "C" {

double ba3_ref_chi2(const ba3_problem* p, const double* poses12, const double* lms, double* edge_chi2) {
    St s;
    s.T.resize(p->P);
    for (int a = 0; a < p->P; ++a) s.T[a] = se3_from(poses12 + 12 * (size_t)a);
    s.X.assign(lms, lms + 3 * (size_t)p->L);
    return chi2_all(*p, s, edge_chi2);
}

// one projection / odometry edge (for Jacobian tests)
void ba3_ref_proj_edge(const ba3_problem* p, const double* pose12, const double* X, const double* uv, double* e, double* Jp, double* Jl) {
    proj_edge(*p, se3_from(pose12), X, uv, e, Jp, Jl);
}
void ba3_ref_odo_edge(const double* Ti12, const double* Tj12, const double* C12, double* e, double* Ji, double* Jj) {
    odo_edge(se3_from(Ti12), se3_from(Tj12), se3_from(C12), e, Ji, Jj);
}

// reduced system at the given state (S: 6P x 6P row-major, bs: 6P)
void ba3_ref_reduced_system(const ba3_problem* p, const double* poses12, const double* lms, double lambda, double* S, double* bs) {
    St s;
    s.T.resize(p->P);
    for (int a = 0; a < p->P; ++a) s.T[a] = se3_from(poses12 + 12 * (size_t)a);
    s.X.assign(lms, lms + 3 * (size_t)p->L);
    Sys y;
    y.lm_ptr.assign(p->L + 1, 0);
    for (int k = 0; k < p->E; ++k) y.lm_ptr[p->e_lm[k] + 1]++;
    for (int l = 0; l < p->L; ++l) y.lm_ptr[l + 1] += y.lm_ptr[l];
    y.lm_e.resize(p->E);
    { std::vector<int> f(y.lm_ptr.begin(), y.lm_ptr.end() - 1); for (int k = 0; k < p->E; ++k) y.lm_e[f[p->e_lm[k]]++] = k; }
    build(*p, s, y);
    std::vector<double> Sv, bv, Dinv;
    schur(*p, y, lambda, Sv, bv, Dinv);
    std::memcpy(S, Sv.data(), Sv.size() * sizeof(double));
    std::memcpy(bs, bv.data(), bv.size() * sizeof(double));
}

// Levenberg-Marquardt with g2o's policy (ba_ref.cpp); edge_chi2 (E, nullable) = chi2() of every projection edge at the
// final estimate, what LocalMapper::removeOutlierChi2 compares against 25
int ba3_ref_optimize(const ba3_problem* pp, int iters, double* poses_out12, double* lms_out, double* edge_chi2,
                     ba_ref_stats* stats) {
    const ba3_problem& p = *pp;
    St st, trial;
    st.T.resize(p.P);
    for (int a = 0; a < p.P; ++a) st.T[a] = se3_from(p.poses + 12 * (size_t)a);
    st.X.assign(p.lms, p.lms + 3 * (size_t)p.L);
    Sys y;
    y.lm_ptr.assign(p.L + 1, 0);
    for (int k = 0; k < p.E; ++k) y.lm_ptr[p.e_lm[k] + 1]++;
    for (int l = 0; l < p.L; ++l) y.lm_ptr[l + 1] += y.lm_ptr[l];
    y.lm_e.resize(p.E);
    { std::vector<int> f(y.lm_ptr.begin(), y.lm_ptr.end() - 1); for (int k = 0; k < p.E; ++k) y.lm_e[f[p.e_lm[k]]++] = k; }
    const int n = 6 * p.P;
    ba_ref_stats s;
    std::memset(&s, 0, sizeof(s));
    double lambda = 0, ni = 2;
    s.chi2_init = s.chi2_final = chi2_all(p, st, nullptr);
    std::vector<double> S, bs, Dinv, F, xp, xl((size_t)p.L * 3);
    bool ok = true;
    for (int it = 0; it < iters && ok; ++it) {
        double currentChi = chi2_all(p, st, nullptr);
        build(p, st, y);
        if (it == 0) {
            double maxd = 0;
            for (int a = 0; a < p.P; ++a)
                if (!p.fixed[a])
                    for (int r = 0; r < 6; ++r) maxd = std::max(maxd, std::fabs(y.Hpp[(size_t)(6 * a + r) * n + 6 * a + r]));
            for (int l = 0; l < p.L; ++l)
                for (int r = 0; r < 3; ++r) maxd = std::max(maxd, std::fabs(y.Hll[(size_t)l * 9 + 4 * r]));
            lambda = 1e-5 * maxd;
            ni = 2;
        }
        double rho = 0;
        int qmax = 0;
        do {
            schur(p, y, lambda, S, bs, Dinv);
            F = S; xp = bs;
            const bool ok2 = chol_solve(F, n, xp);
            if (!ok2) std::fill(xp.begin(), xp.end(), 0.0);
            double scale = 0;
            trial = st;
            for (int l = 0; l < p.L; ++l) {
                double c[3] = {y.bl[(size_t)l * 3], y.bl[(size_t)l * 3 + 1], y.bl[(size_t)l * 3 + 2]};
                for (int u = y.lm_ptr[l]; u < y.lm_ptr[l + 1]; ++u) {
                    const int e = y.lm_e[u], a = p.e_kf[e];
                    if (p.fixed[a]) continue;
                    const double* B = &y.Hpl[(size_t)e * 18];
                    for (int j = 0; j < 3; ++j)
                        for (int r = 0; r < 6; ++r) c[j] -= B[3 * r + j] * xp[6 * a + r];
                }
                const double* Di = &Dinv[(size_t)l * 9];
                for (int r = 0; r < 3; ++r) {
                    const double x = Di[3 * r] * c[0] + Di[3 * r + 1] * c[1] + Di[3 * r + 2] * c[2];
                    xl[(size_t)l * 3 + r] = x;
                    trial.X[(size_t)l * 3 + r] += x;
                    scale += x * (lambda * x + y.bl[(size_t)l * 3 + r]);
                }
            }
            for (int a = 0; a < p.P; ++a) {
                if (p.fixed[a]) continue;
                trial.T[a] = se3_mul(se3_exp(&xp[6 * a]), st.T[a]);   // VertexSE3Expmap::oplusImpl
                for (int r = 0; r < 6; ++r) scale += xp[6 * a + r] * (lambda * xp[6 * a + r] + y.bp[6 * a + r]);
            }
            double tempChi = chi2_all(p, trial, nullptr);
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
                st = trial;
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
    if (poses_out12) for (int a = 0; a < p.P; ++a) se3_to(st.T[a], poses_out12 + 12 * (size_t)a);
    if (lms_out) std::memcpy(lms_out, st.X.data(), st.X.size() * sizeof(double));
    if (edge_chi2) chi2_all(p, st, edge_chi2);
    if (stats) *stats = s;
    return 0;
}

}  // extern "C"
