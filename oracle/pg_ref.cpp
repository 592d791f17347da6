// ORACLE - TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py's cpu_baseline leg).  PARITY UNPINNED for g2o's part (its
// EdgeSE3 / EdgeSE3Prior / solver cannot be built here): a restatement, checked against a numpy / scipy
// model (tests/test_pg_oracle.py).  What se2lam wrote of it - pg_ref_plane_motion_prior = addVertexSE3PlaneMotion - is held to
// the reference's own src/optimizer.cpp compiled in oracle/_ref (tests/test_ref_compiled.py, 1e-9), and pg_ref_chi2 to the cost of
// the pose graph the compiled GlobalMapper::GlobalBA builds (EdgeSE3 / EdgeSE3Prior by g2o's definition in the stand-in, 1e-9).
//
// pg_ref: the pose graph of GlobalMapper::GlobalBA - SURVEY.md section 8(f).4:
//   /root/reference/src/GlobalMapper.cpp:328-535: one g2o::VertexSE3 per key frame (T_w_c; KF 0 fixed), one EdgeSE3Prior per
//       key frame from addVertexSE3PlaneMotion (src/optimizer.cpp:336-470, the non-OLD branch; ParameterSE3Offset =
//       identity), EdgeSE3 odometry edges (mOdoMeasureFrom) and EdgeSE3 feature edges (mFtrMeasureFrom, produced by
//       Sparsifier::DoMarginalizeSE3XYZ), Levenberg-Marquardt optimize(GLOBAL_ITER).
// [3P g2o 20160424 types/slam3d, restated from memory]:
//   VertexSE3::oplusImpl          estimate <- estimate * fromVectorMQT(update)
//   fromVectorMQT / toVectorMQT   (translation, compact quaternion q_xyz with w = sqrt(1 - |q_xyz|^2) >= 0)
//   EdgeSE3::computeError         toVectorMQT(Z^-1 * X_i^-1 * X_j);  EdgeSE3Prior: toVectorMQT(Z^-1 * X)
//   Jacobians                     the exact derivatives of those maps at update = 0 (g2o: isometry3d_gradients.h),
//                                 written here through quaternion algebra: for E = A D_i^-1 B D_j (A = Z^-1, B = X_i^-1 X_j)
//                                     d e / d d_j = [R_E 0; 0  w_E I + [q_E]x]
//                                     d e / d d_i = [-R_A  2 R_A [t_B]x; 0  -(L(q_A) R(q_B))_xyz,xyz]   (sign of q_E applied)
//   information matrices in the vector order (translation, rotation); no robust kernel.
// pose12 = rotation row-major (9) then translation (3).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

extern "C" {
struct pg_problem {
    int32_t P, O;
    const double* poses;       // P x 12 (T_w_c)
    const uint8_t* fixed;      // P
    const uint8_t* has_prior;  // P
    const double* prior_meas;  // P x 12
    const double* prior_info;  // P x 36  (translation, rotation)
    const int32_t* o_i;        // O  vertex 0 ("from")
    const int32_t* o_j;        // O  vertex 1 ("to")
    const double* o_meas;      // O x 12
    const double* o_info;      // O x 36
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

struct Iso { double R[9], t[3]; };
inline Iso iso_from(const double* p) { Iso T; std::memcpy(T.R, p, 72); std::memcpy(T.t, p + 9, 24); return T; }
inline void iso_to(const Iso& T, double* p) { std::memcpy(p, T.R, 72); std::memcpy(p + 9, T.t, 24); }
inline Iso iso_mul(const Iso& a, const Iso& b) {
    Iso c;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) c.R[3 * i + j] = a.R[3 * i] * b.R[j] + a.R[3 * i + 1] * b.R[3 + j] + a.R[3 * i + 2] * b.R[6 + j];
        c.t[i] = a.R[3 * i] * b.t[0] + a.R[3 * i + 1] * b.t[1] + a.R[3 * i + 2] * b.t[2] + a.t[i];
    }
    return c;
}
inline Iso iso_inv(const Iso& a) {
    Iso c;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c.R[3 * i + j] = a.R[3 * j + i];
    for (int i = 0; i < 3; ++i) c.t[i] = -(c.R[3 * i] * a.t[0] + c.R[3 * i + 1] * a.t[1] + c.R[3 * i + 2] * a.t[2]);
    return c;
}
// Eigen::Quaterniond(R) (w, x, y, z), normalised, w >= 0 (g2o::internal::normalize)
inline void quat_of(const double* R, double q[4]) {
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = std::sqrt(t + 1.0);
        q[0] = 0.5 * t; t = 0.5 / t;
        q[1] = (R[7] - R[5]) * t; q[2] = (R[2] - R[6]) * t; q[3] = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
        q[1 + i] = 0.5 * t; t = 0.5 / t;
        q[0] = (R[3 * k + j] - R[3 * j + k]) * t;
        q[1 + j] = (R[3 * j + i] + R[3 * i + j]) * t;
        q[1 + k] = (R[3 * k + i] + R[3 * i + k]) * t;
    }
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double s = (q[0] < 0 ? -1.0 : 1.0) / n;
    for (int a = 0; a < 4; ++a) q[a] *= s;
}
inline void mat_of_quat(const double q[4], double* R) {   // Eigen::Quaterniond::toRotationMatrix
    const double tx = 2 * q[1], ty = 2 * q[2], tz = 2 * q[3];
    const double twx = tx * q[0], twy = ty * q[0], twz = tz * q[0];
    const double txx = tx * q[1], txy = ty * q[1], txz = tz * q[1];
    const double tyy = ty * q[2], tyz = tz * q[2], tzz = tz * q[3];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
inline void to_mqt(const Iso& T, double v[6]) {
    double q[4];
    quat_of(T.R, q);
    v[0] = T.t[0]; v[1] = T.t[1]; v[2] = T.t[2];
    v[3] = q[1]; v[4] = q[2]; v[5] = q[3];
}
inline Iso from_mqt(const double v[6]) {
    Iso T;
    double w = 1 - (v[3] * v[3] + v[4] * v[4] + v[5] * v[5]);
    if (w < 0) {
        for (int i = 0; i < 9; ++i) T.R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    } else {
        const double q[4] = {std::sqrt(w), v[3], v[4], v[5]};
        mat_of_quat(q, T.R);
    }
    T.t[0] = v[0]; T.t[1] = v[1]; T.t[2] = v[2];
    return T;
}
// d toVectorMQT(E * fromVectorMQT(d)) / d d at 0  (6x6 row-major)
inline void jac_right(const Iso& E, double J[36]) {
    double q[4];
    quat_of(E.R, q);
    for (int i = 0; i < 36; ++i) J[i] = 0;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) J[6 * r + c] = E.R[3 * r + c];
    const double S[9] = {q[0], -q[3], q[2], q[3], q[0], -q[1], -q[2], q[1], q[0]};   // w I + [q_xyz]x
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) J[6 * (3 + r) + 3 + c] = S[3 * r + c];
}
// e = toVectorMQT(A D_i^-1 B); d e / d d_i at 0
inline void jac_left_inv(const Iso& A, const Iso& B, double J[36]) {
    double qa[4], qb[4], qe[4];
    quat_of(A.R, qa);
    quat_of(B.R, qb);
    const Iso E = iso_mul(A, B);
    quat_of(E.R, qe);
    // quaternion product qa * qb (before the sign normalisation of qe): its w decides the sign that was applied
    const double wprod = qa[0] * qb[0] - qa[1] * qb[1] - qa[2] * qb[2] - qa[3] * qb[3];
    const double sgn = wprod < 0 ? -1.0 : 1.0;
    for (int i = 0; i < 36; ++i) J[i] = 0;
    const double sk[9] = {0, -B.t[2], B.t[1], B.t[2], 0, -B.t[0], -B.t[1], B.t[0], 0};
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            J[6 * r + c] = -A.R[3 * r + c];
            J[6 * r + 3 + c] = 2 * (A.R[3 * r] * sk[c] + A.R[3 * r + 1] * sk[3 + c] + A.R[3 * r + 2] * sk[6 + c]);
        }
    // xyz part of qa * (0, u) * qb for the unit vectors u, with u = -v
    for (int c = 0; c < 3; ++c) {
        double u[4] = {0, 0, 0, 0};
        u[1 + c] = 1;
        double m[4], o[4];   // m = qa * u, o = m * qb
        m[0] = qa[0] * u[0] - qa[1] * u[1] - qa[2] * u[2] - qa[3] * u[3];
        m[1] = qa[0] * u[1] + qa[1] * u[0] + qa[2] * u[3] - qa[3] * u[2];
        m[2] = qa[0] * u[2] - qa[1] * u[3] + qa[2] * u[0] + qa[3] * u[1];
        m[3] = qa[0] * u[3] + qa[1] * u[2] - qa[2] * u[1] + qa[3] * u[0];
        o[1] = m[0] * qb[1] + m[1] * qb[0] + m[2] * qb[3] - m[3] * qb[2];
        o[2] = m[0] * qb[2] - m[1] * qb[3] + m[2] * qb[0] + m[3] * qb[1];
        o[3] = m[0] * qb[3] + m[1] * qb[2] - m[2] * qb[1] + m[3] * qb[0];
        for (int r = 0; r < 3; ++r) J[6 * (3 + r) + 3 + c] = -sgn * o[1 + r];
    }
}

inline double quad(const double* W, const double e[6]) {
    double s = 0;
    for (int r = 0; r < 6; ++r) {
        double v = 0;
        for (int c = 0; c < 6; ++c) v += W[6 * r + c] * e[c];
        s += e[r] * v;
    }
    return s;
}

double chi2_all(const pg_problem& p, const std::vector<Iso>& X, double* edge_chi2) {
    double chi = 0;
    for (int a = 0; a < p.P; ++a)
        if (p.has_prior[a]) {
            double e[6];
            to_mqt(iso_mul(iso_inv(iso_from(p.prior_meas + 12 * (size_t)a)), X[a]), e);
            chi += quad(p.prior_info + 36 * (size_t)a, e);
        }
    for (int k = 0; k < p.O; ++k) {
        double e[6];
        to_mqt(iso_mul(iso_inv(iso_from(p.o_meas + 12 * (size_t)k)), iso_mul(iso_inv(X[p.o_i[k]]), X[p.o_j[k]])), e);
        const double c2 = quad(p.o_info + 36 * (size_t)k, e);
        if (edge_chi2) edge_chi2[k] = c2;
        chi += c2;
    }
    return chi;
}

void build(const pg_problem& p, const std::vector<Iso>& X, std::vector<double>& H, std::vector<double>& b) {
    const int n = 6 * p.P;
    H.assign((size_t)n * n, 0.0);
    b.assign(n, 0.0);
    auto add = [&](int a, int c, const double* Ja, const double* Jc, const double* W) {   // H_ac += Ja' W Jc
        for (int r = 0; r < 6; ++r)
            for (int s = 0; s < 6; ++s) {
                double v = 0;
                for (int q = 0; q < 6; ++q) {
                    double wj = 0;
                    for (int u = 0; u < 6; ++u) wj += W[6 * q + u] * Jc[6 * u + s];
                    v += Ja[6 * q + r] * wj;
                }
                H[(size_t)(6 * a + r) * n + 6 * c + s] += v;
            }
    };
    auto addb = [&](int a, const double* Ja, const double* W, const double* e) {   // b_a -= Ja' W e
        double We[6];
        for (int r = 0; r < 6; ++r) { We[r] = 0; for (int c = 0; c < 6; ++c) We[r] += W[6 * r + c] * e[c]; }
        for (int r = 0; r < 6; ++r) { double v = 0; for (int q = 0; q < 6; ++q) v += Ja[6 * q + r] * We[q]; b[6 * a + r] -= v; }
    };
    for (int a = 0; a < p.P; ++a) {
        if (!p.has_prior[a] || p.fixed[a]) continue;
        const Iso E = iso_mul(iso_inv(iso_from(p.prior_meas + 12 * (size_t)a)), X[a]);
        double e[6], J[36];
        to_mqt(E, e);
        jac_right(E, J);
        add(a, a, J, J, p.prior_info + 36 * (size_t)a);
        addb(a, J, p.prior_info + 36 * (size_t)a, e);
    }
    for (int k = 0; k < p.O; ++k) {
        const int i = p.o_i[k], j = p.o_j[k];
        const Iso A = iso_inv(iso_from(p.o_meas + 12 * (size_t)k)), B = iso_mul(iso_inv(X[i]), X[j]);
        const Iso E = iso_mul(A, B);
        double e[6], Ji[36], Jj[36];
        to_mqt(E, e);
        jac_left_inv(A, B, Ji);
        jac_right(E, Jj);
        const double* W = p.o_info + 36 * (size_t)k;
        if (!p.fixed[i]) { add(i, i, Ji, Ji, W); addb(i, Ji, W, e); }
        if (!p.fixed[j]) { add(j, j, Jj, Jj, W); addb(j, Jj, W, e); }
        if (!p.fixed[i] && !p.fixed[j]) { add(i, j, Ji, Jj, W); add(j, i, Jj, Ji, W); }
    }
    for (int a = 0; a < p.P; ++a)
        if (p.fixed[a])
            for (int r = 0; r < 6; ++r) H[(size_t)(6 * a + r) * n + 6 * a + r] = 1.0;
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

double pg_ref_chi2(const pg_problem* p, const double* poses12, double* edge_chi2) {
    std::vector<Iso> X(p->P);
    for (int a = 0; a < p->P; ++a) X[a] = iso_from(poses12 + 12 * (size_t)a);
    return chi2_all(*p, X, edge_chi2);
}

// error and Jacobians of one EdgeSE3 (Jacobian tests): e (6), Ji, Jj (6x6)
void pg_ref_edge(const double* Xi12, const double* Xj12, const double* Z12, double* e, double* Ji, double* Jj) {
    const Iso A = iso_inv(iso_from(Z12)), B = iso_mul(iso_inv(iso_from(Xi12)), iso_from(Xj12));
    const Iso E = iso_mul(A, B);
    to_mqt(E, e);
    if (Ji) jac_left_inv(A, B, Ji);
    if (Jj) jac_right(E, Jj);
}
void pg_ref_oplus(const double* X12, const double* update6, double* out12) { iso_to(iso_mul(iso_from(X12), from_mqt(update6)), out12); }

void pg_ref_system(const pg_problem* p, const double* poses12, double lambda, double* H, double* b) {
    std::vector<Iso> X(p->P);
    for (int a = 0; a < p->P; ++a) X[a] = iso_from(poses12 + 12 * (size_t)a);
    std::vector<double> Hv, bv;
    build(*p, X, Hv, bv);
    const int n = 6 * p->P;
    for (int a = 0; a < p->P; ++a)
        if (!p->fixed[a])
            for (int r = 0; r < 6; ++r) Hv[(size_t)(6 * a + r) * n + 6 * a + r] += lambda;
    std::memcpy(H, Hv.data(), Hv.size() * sizeof(double));
    std::memcpy(b, bv.data(), bv.size() * sizeof(double));
}

int pg_ref_optimize(const pg_problem* pp, int iters, double* poses_out12, double* edge_chi2, ba_ref_stats* stats) {
    const pg_problem& p = *pp;
    const int n = 6 * p.P;
    std::vector<Iso> X(p.P), T(p.P);
    for (int a = 0; a < p.P; ++a) X[a] = iso_from(p.poses + 12 * (size_t)a);
    ba_ref_stats s;
    std::memset(&s, 0, sizeof(s));
    double lambda = 0, ni = 2;
    s.chi2_init = s.chi2_final = chi2_all(p, X, nullptr);
    std::vector<double> H, b, F, x;
    bool ok = true;
    for (int it = 0; it < iters && ok; ++it) {
        double currentChi = chi2_all(p, X, nullptr);
        build(p, X, H, b);
        if (it == 0) {
            double maxd = 0;
            for (int a = 0; a < p.P; ++a)
                if (!p.fixed[a])
                    for (int r = 0; r < 6; ++r) maxd = std::max(maxd, std::fabs(H[(size_t)(6 * a + r) * n + 6 * a + r]));
            lambda = 1e-5 * maxd;
            ni = 2;
        }
        double rho = 0;
        int qmax = 0;
        do {
            F = H;
            for (int a = 0; a < p.P; ++a)
                if (!p.fixed[a])
                    for (int r = 0; r < 6; ++r) F[(size_t)(6 * a + r) * n + 6 * a + r] += lambda;
            x = b;
            const bool ok2 = chol_solve(F, n, x);
            if (!ok2) std::fill(x.begin(), x.end(), 0.0);
            double scale = 0;
            for (int a = 0; a < p.P; ++a) {
                T[a] = X[a];
                if (p.fixed[a]) continue;
                T[a] = iso_mul(X[a], from_mqt(&x[6 * a]));     // VertexSE3::oplusImpl
                for (int r = 0; r < 6; ++r) scale += x[6 * a + r] * (lambda * x[6 * a + r] + b[6 * a + r]);
            }
            double tempChi = chi2_all(p, T, nullptr);
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
                X = T;
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
    if (poses_out12) for (int a = 0; a < p.P; ++a) iso_to(X[a], poses_out12 + 12 * (size_t)a);
    if (edge_chi2) chi2_all(p, X, edge_chi2);
    if (stats) *stats = s;
    return 0;
}

// addVertexSE3PlaneMotion (src/optimizer.cpp:336-470, the #else branch): measurement = T_w_c with the body's roll,
// pitch and height removed, information = AdjTR(Tbc)' diag(1e-4, 1e-4, z, xrot, yrot, 1e-4) AdjTR(Tbc), vector order
// (translation, rotation), AdjTR(T) = [R skew(t) R; 0 R] (:95-104)
void pg_ref_plane_motion_prior(const double* Twc12, const double* Tbc12, double xrot_info, double yrot_info, double z_info,
                               double* meas12, double* info36) {
    const Iso Twc = iso_from(Twc12), Tbc = iso_from(Tbc12);
    Iso Twb = iso_mul(Twc, iso_inv(Tbc));
    // rotation vector of R_wb (Eigen::AngleAxisd angle * axis) through its quaternion: yaw component only
    double q[4];
    quat_of(Twb.R, q);
    const double nv = std::sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    double yaw = 0;
    if (nv > 0) {
        const double angle = 2 * std::atan2(nv, q[0]);
        yaw = angle * q[3] / nv;
    }
    const double c = std::cos(yaw), sn = std::sin(yaw);
    // Eigen::Quaterniond(AngleAxis(yaw, z)).toRotationMatrix()
    const double qz[4] = {std::cos(0.5 * yaw), 0, 0, std::sin(0.5 * yaw)};
    (void)c; (void)sn;
    mat_of_quat(qz, Twb.R);
    Twb.t[2] = 0;
    iso_to(iso_mul(Twb, Tbc), meas12);
    double A[36] = {0};
    const double sk[9] = {0, -Tbc.t[2], Tbc.t[1], Tbc.t[2], 0, -Tbc.t[0], -Tbc.t[1], Tbc.t[0], 0};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            A[6 * i + j] = Tbc.R[3 * i + j];
            A[6 * (i + 3) + (j + 3)] = Tbc.R[3 * i + j];
            A[6 * i + (j + 3)] = sk[3 * i] * Tbc.R[j] + sk[3 * i + 1] * Tbc.R[3 + j] + sk[3 * i + 2] * Tbc.R[6 + j];
        }
    const double D[6] = {1e-4, 1e-4, z_info, xrot_info, yrot_info, 1e-4};
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double s = 0;
            for (int k = 0; k < 6; ++k) s += A[6 * k + i] * D[k] * A[6 * k + j];
            info36[6 * i + j] = s;
        }
}

}  // extern "C"
