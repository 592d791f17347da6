// ORACLE - TEST INFRASTRUCTURE ONLY.  PARITY: pinned against the reference's own src/sparsifier.cpp, compiled unmodified in
// oracle/_ref against a stand-in for the Eigen / g2o headers (tests/test_ref_compiled.py: relative pose bit-identical, InfoSE3
// on the same marginal Hessian to 1e-12, the marginal Hessian from the reference's per-measurement Hessians to 1e-12, end to
// end to 1e-5 - the construction's condition number is 1e15); also checked against an analytic numpy model
// (tests/test_sparsify.py).  The Eigen solvers underneath (LDLT, inverse, JacobiSVD) are the stand-in's, not the library's.
//
// Sparsifier::DoMarginalizeSE3XYZ (/root/reference/src/sparsifier.cpp:105-177) - SURVEY.md section 8(f).4: the feature
// constraint between two key frames.  Two key frames KF (T_w_c as g2o::SE3Quat), the N map points both observe, and
// per key frame and point a measurement information matrix (MeasSE3XYZ: idKF, idMP, info; z is not used):
//   H (12 + 3N)     = sum J' info J, J (3x9) the FORWARD-DIFFERENCE Jacobian (delta 1e-6) of z = KF^-1 * MP w.r.t.
//                     KF.toMinimalVector() = (t, q_xyz) and MP                                         (:59-104)
//   H11 += 1e-6 I,   H_marginal = H11 - H12 H22^-1 H21  (H22 is block diagonal: 3x3 per point)      (:153-166)
//   InfoSE3 (:219-275): J (6x12) = forward differences of (KF1^-1 KF2).toMinimalVector(), I = (J H^-1 J')^-1, symmetrised;
//                     SVD with singular values clamped to [1e-6, 1e4] (a negative eigenvalue becomes 1e-6), I = U S V',
//                     symmetrised - on a symmetric matrix this is the eigen-decomposition with clamped eigenvalues
//   z_out = KF1^-1 * KF2
// [3P g2o 20160424 SE3Quat: toMinimalVector / fromMinimalVector / inverse / map / operator*, Eigen quaternion-vector
// product] restated from memory; Eigen's LDLT / inverse() / JacobiSVD are replaced by per-point closed-form 3x3
// inverses, Gauss-Jordan with partial pivoting and a cyclic Jacobi eigen-solver (exact solvers: agree to round-off).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

struct Q3 { double w, x, y, z, t[3]; };   // SE3Quat: unit quaternion + translation

inline void rotate(const Q3& q, const double v[3], double o[3]) {   // Eigen: v + w * (2 qv x v) + qv x (2 qv x v)
    double uv[3] = {q.y * v[2] - q.z * v[1], q.z * v[0] - q.x * v[2], q.x * v[1] - q.y * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    o[0] = v[0] + q.w * uv[0] + (q.y * uv[2] - q.z * uv[1]);
    o[1] = v[1] + q.w * uv[1] + (q.z * uv[0] - q.x * uv[2]);
    o[2] = v[2] + q.w * uv[2] + (q.x * uv[1] - q.y * uv[0]);
}
inline void normalize_rot(Q3& q) {   // SE3Quat::normalizeRotation
    if (q.w < 0) { q.w = -q.w; q.x = -q.x; q.y = -q.y; q.z = -q.z; }
    const double n = std::sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    q.w /= n; q.x /= n; q.y /= n; q.z /= n;
}
inline Q3 from_pose12(const double* p) {   // SE3Quat(R, t): Eigen::Quaterniond(R), normalised
    const double* R = p;
    Q3 q;
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = std::sqrt(t + 1.0);
        q.w = 0.5 * t; t = 0.5 / t;
        q.x = (R[7] - R[5]) * t; q.y = (R[2] - R[6]) * t; q.z = (R[3] - R[1]) * t;
    } else if (!(R[4] > R[0]) && !(R[8] > R[0])) {
        t = std::sqrt(R[0] - R[4] - R[8] + 1.0);
        q.x = 0.5 * t; t = 0.5 / t;
        q.w = (R[7] - R[5]) * t; q.y = (R[3] + R[1]) * t; q.z = (R[6] + R[2]) * t;
    } else if (R[4] > R[0] && !(R[8] > R[4])) {
        t = std::sqrt(R[4] - R[8] - R[0] + 1.0);
        q.y = 0.5 * t; t = 0.5 / t;
        q.w = (R[2] - R[6]) * t; q.z = (R[7] + R[5]) * t; q.x = (R[1] + R[3]) * t;
    } else {
        t = std::sqrt(R[8] - R[0] - R[4] + 1.0);
        q.z = 0.5 * t; t = 0.5 / t;
        q.w = (R[3] - R[1]) * t; q.x = (R[2] + R[6]) * t; q.y = (R[5] + R[7]) * t;
    }
    q.t[0] = p[9]; q.t[1] = p[10]; q.t[2] = p[11];
    normalize_rot(q);
    return q;
}
inline void to_pose12(const Q3& q, double* p) {
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w, txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    p[0] = 1 - (tyy + tzz); p[1] = txy - twz; p[2] = txz + twy;
    p[3] = txy + twz; p[4] = 1 - (txx + tzz); p[5] = tyz - twx;
    p[6] = txz - twy; p[7] = tyz + twx; p[8] = 1 - (txx + tyy);
    p[9] = q.t[0]; p[10] = q.t[1]; p[11] = q.t[2];
}
inline void to_min(const Q3& q, double v[6]) { v[0] = q.t[0]; v[1] = q.t[1]; v[2] = q.t[2]; v[3] = q.x; v[4] = q.y; v[5] = q.z; }
inline Q3 from_min(const double v[6]) {
    Q3 q;
    const double w = 1. - v[3] * v[3] - v[4] * v[4] - v[5] * v[5];
    if (w > 0) { q.w = std::sqrt(w); q.x = v[3]; q.y = v[4]; q.z = v[5]; }
    else { q.w = 0; q.x = -v[3]; q.y = -v[4]; q.z = -v[5]; }
    q.t[0] = v[0]; q.t[1] = v[1]; q.t[2] = v[2];
    return q;
}
inline Q3 inverse(const Q3& q) {
    Q3 r;
    r.w = q.w; r.x = -q.x; r.y = -q.y; r.z = -q.z;
    const double m[3] = {q.t[0] * -1., q.t[1] * -1., q.t[2] * -1.};
    rotate(r, m, r.t);
    return r;
}
inline Q3 mul(const Q3& a, const Q3& b) {
    Q3 r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    double rt[3];
    rotate(a, b.t, rt);
    r.t[0] = a.t[0] + rt[0]; r.t[1] = a.t[1] + rt[1]; r.t[2] = a.t[2] + rt[2];
    normalize_rot(r);
    return r;
}
inline void map_inv(const Q3& kf, const double mp[3], double z[3]) {   // KF.inverse() * MP
    const Q3 inv = inverse(kf);
    double r[3];
    rotate(inv, mp, r);
    z[0] = r[0] + inv.t[0]; z[1] = r[1] + inv.t[1]; z[2] = r[2] + inv.t[2];
}

// JacobianSE3XYZ (:59-94): J is 3 x 9 row-major
void jacobian_se3xyz(const Q3& kf, const double mp[3], double J[27]) {
    const double delta = 1e-6;
    double zref[3], v6[6];
    map_inv(kf, mp, zref);
    to_min(kf, v6);
    for (int i = 0; i < 9; ++i) {
        double zd[3];
        if (i < 6) {
            double vd[6];
            std::memcpy(vd, v6, sizeof(vd));
            vd[i] += delta;
            map_inv(from_min(vd), mp, zd);
        } else {
            double md[3] = {mp[0], mp[1], mp[2]};
            md[i - 6] += delta;
            map_inv(kf, md, zd);
        }
        for (int r = 0; r < 3; ++r) J[9 * r + i] = (zd[r] - zref[r]) / delta;
    }
}

bool invert(double* A, int n) {   // Gauss-Jordan with partial pivoting, in place; row-major
    std::vector<double> B((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) B[(size_t)i * n + i] = 1.0;
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int r = c + 1; r < n; ++r)
            if (std::fabs(A[(size_t)r * n + c]) > std::fabs(A[(size_t)piv * n + c])) piv = r;
        if (A[(size_t)piv * n + c] == 0.0) return false;
        if (piv != c)
            for (int k = 0; k < n; ++k) { std::swap(A[(size_t)c * n + k], A[(size_t)piv * n + k]); std::swap(B[(size_t)c * n + k], B[(size_t)piv * n + k]); }
        const double d = 1.0 / A[(size_t)c * n + c];
        for (int k = 0; k < n; ++k) { A[(size_t)c * n + k] *= d; B[(size_t)c * n + k] *= d; }
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            const double f = A[(size_t)r * n + c];
            if (f == 0.0) continue;
            for (int k = 0; k < n; ++k) { A[(size_t)r * n + k] -= f * A[(size_t)c * n + k]; B[(size_t)r * n + k] -= f * B[(size_t)c * n + k]; }
        }
    }
    std::memcpy(A, B.data(), B.size() * sizeof(double));
    return true;
}

// symmetric 6x6: eigenvalues clamped as InfoSE3 does with singular values, then recomposed
void clamp_spectrum6(double* I) {
    double A[36], V[36];
    std::memcpy(A, I, sizeof(A));
    for (int i = 0; i < 36; ++i) V[i] = (i % 7 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        for (int p = 0; p < 5; ++p)
            for (int q = p + 1; q < 6; ++q) {
                const double apq = A[6 * p + q];
                if (apq == 0.0) continue;
                const double theta = (A[6 * q + q] - A[6 * p + p]) / (2 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 6; ++k) {
                    const double akp = A[6 * k + p], akq = A[6 * k + q];
                    A[6 * k + p] = c * akp - s * akq;
                    A[6 * k + q] = s * akp + c * akq;
                }
                for (int k = 0; k < 6; ++k) {
                    const double apk = A[6 * p + k], aqk = A[6 * q + k];
                    A[6 * p + k] = c * apk - s * aqk;
                    A[6 * q + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 6; ++k) {
                    const double vkp = V[6 * k + p], vkq = V[6 * k + q];
                    V[6 * k + p] = c * vkp - s * vkq;
                    V[6 * k + q] = s * vkp + c * vkq;
                }
            }
    }
    double lam[6];
    for (int k = 0; k < 6; ++k) {
        const double l = A[7 * k];
        lam[k] = l < 0 ? 1e-6 : std::fmin(std::fmax(l, 1e-6), 1e4);
    }
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) {
            double v = 0;
            for (int k = 0; k < 6; ++k) v += V[6 * r + k] * lam[k] * V[6 * c + k];
            I[6 * r + c] = v;
        }
    for (int r = 0; r < 6; ++r)
        for (int c = r + 1; c < 6; ++c) { const double m = 0.5 * (I[6 * r + c] + I[6 * c + r]); I[6 * r + c] = I[6 * c + r] = m; }
}

}  // namespace

extern "C" {

// kf12: 2 x 12 (T_w_c of KF0, KF1); mp: N x 3; measurements: M x (kf in {0, 1}, mp in [0, N)), info M x 9 (row-major 3x3).
// z_out12 = pose12 of KF0^-1 KF1; info_out36 row-major; h_marginal144 (nullable) = the marginalised 12 x 12 Hessian.
void sparsify_ref(const double* kf12, int N, const double* mp, int M, const int32_t* m_kf, const int32_t* m_mp,
                  const double* m_info, double* z_out12, double* info_out36, double* h_marginal144) {
    const Q3 KF[2] = {from_pose12(kf12), from_pose12(kf12 + 12)};
    double H11[144] = {0};
    std::vector<double> Hmm((size_t)N * 9, 0.0), Hkm((size_t)N * 36, 0.0);   // per point: 3x3 and [H_0m; H_1m] (12 x 3)
    for (int i = 0; i < M; ++i) {
        const int k = m_kf[i], m = m_mp[i];
        if (k != 0 && k != 1) continue;
        double J[27];
        jacobian_se3xyz(KF[k], mp + 3 * (size_t)m, J);
        const double* W = m_info + 9 * (size_t)i;
        double WJ[27];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 9; ++c) WJ[9 * r + c] = W[3 * r] * J[c] + W[3 * r + 1] * J[9 + c] + W[3 * r + 2] * J[18 + c];
        for (int a = 0; a < 9; ++a)
            for (int b = 0; b < 9; ++b) {
                const double v = J[a] * WJ[b] + J[9 + a] * WJ[9 + b] + J[18 + a] * WJ[18 + b];
                if (a < 6 && b < 6) H11[12 * (6 * k + a) + 6 * k + b] += v;
                else if (a >= 6 && b >= 6) Hmm[(size_t)m * 9 + 3 * (a - 6) + (b - 6)] += v;
                else if (a < 6) Hkm[(size_t)m * 36 + 3 * (6 * k + a) + (b - 6)] += v;
            }
    }
    for (int i = 0; i < 12; ++i) H11[13 * i] += 1e-6;
    for (int m = 0; m < N; ++m) {   // H11 -= H_1m H_mm^-1 H_m1
        double D[9];
        std::memcpy(D, &Hmm[(size_t)m * 9], sizeof(D));
        bool any = false;
        for (int i = 0; i < 9; ++i) any |= D[i] != 0.0;
        if (!any || !invert(D, 3)) continue;
        const double* B = &Hkm[(size_t)m * 36];
        double BD[36];
        for (int r = 0; r < 12; ++r)
            for (int c = 0; c < 3; ++c) BD[3 * r + c] = B[3 * r] * D[c] + B[3 * r + 1] * D[3 + c] + B[3 * r + 2] * D[6 + c];
        for (int r = 0; r < 12; ++r)
            for (int c = 0; c < 12; ++c) H11[12 * r + c] -= BD[3 * r] * B[3 * c] + BD[3 * r + 1] * B[3 * c + 1] + BD[3 * r + 2] * B[3 * c + 2];
    }
    if (h_marginal144) std::memcpy(h_marginal144, H11, sizeof(H11));
    // InfoSE3
    const Q3 zref_q = mul(inverse(KF[0]), KF[1]);
    double zref[6], v1[6], v2[6], J[72];
    to_min(zref_q, zref);
    to_min(KF[0], v1);
    to_min(KF[1], v2);
    const double delta = 1e-6;
    for (int i = 0; i < 12; ++i) {
        double zd[6], vd[6];
        if (i < 6) {
            std::memcpy(vd, v1, sizeof(vd)); vd[i] += delta;
            to_min(mul(inverse(from_min(vd)), KF[1]), zd);
        } else {
            std::memcpy(vd, v2, sizeof(vd)); vd[i - 6] += delta;
            to_min(mul(inverse(KF[0]), from_min(vd)), zd);
        }
        for (int r = 0; r < 6; ++r) J[12 * r + i] = (zd[r] - zref[r]) / delta;
    }
    double Hinv[144];
    std::memcpy(Hinv, H11, sizeof(Hinv));
    invert(Hinv, 12);
    double JH[72], Mx[36];
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 12; ++c) { double v = 0; for (int k = 0; k < 12; ++k) v += J[12 * r + k] * Hinv[12 * k + c]; JH[12 * r + c] = v; }
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) { double v = 0; for (int k = 0; k < 12; ++k) v += JH[12 * r + k] * J[12 * c + k]; Mx[6 * r + c] = v; }
    invert(Mx, 6);
    for (int r = 0; r < 6; ++r)
        for (int c = r + 1; c < 6; ++c) { const double m = 0.5 * (Mx[6 * r + c] + Mx[6 * c + r]); Mx[6 * r + c] = Mx[6 * c + r] = m; }
    clamp_spectrum6(Mx);
    std::memcpy(info_out36, Mx, sizeof(Mx));
    to_pose12(zref_q, z_out12);
}

}  // extern "C"
