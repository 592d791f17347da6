// ORACLE - TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see ba_ref.cpp).
// SE(3) algebra of g2o 20160424's SE3Quat [3P, restated from memory]: exp / log with vector order (omega, upsilon),
// products renormalised through a unit quaternion the way SE3Quat::normalizeRotation does, adj().  Shared by the
// pose-only restatement (ba_ref.cpp) and the marginalising SE3 bundle adjustment (ba3_ref.cpp).
// pose12 = rotation row-major (9) then translation (3): x_c = R x_w + t.
#pragma once
#include <cmath>
#include <cstring>

namespace {

struct Se3 {
    double R[9], t[3];
};
inline Se3 se3_from(const double* p) {
    Se3 T;
    std::memcpy(T.R, p, 9 * sizeof(double));
    std::memcpy(T.t, p + 9, 3 * sizeof(double));
    return T;
}
inline void se3_to(const Se3& T, double* p) {
    std::memcpy(p, T.R, 9 * sizeof(double));
    std::memcpy(p + 9, T.t, 3 * sizeof(double));
}
// Eigen::Quaterniond(R) as (x, y, z, w), then SE3Quat::normalizeRotation: w >= 0, unit norm
inline void quat_of_rotation(const double R[9], double q[4]) {
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = std::sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R[7] - R[5]) * t;
        q[1] = (R[2] - R[6]) * t;
        q[2] = (R[3] - R[1]) * t;
    } else if (R[0] >= R[4] && R[0] >= R[8]) {   // largest diagonal element first (Eigen), written out per case
        t = std::sqrt(R[0] - R[4] - R[8] + 1.0);
        q[0] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (R[7] - R[5]) * t;
        q[1] = (R[3] + R[1]) * t;
        q[2] = (R[6] + R[2]) * t;
    } else if (R[4] > R[0] && R[4] >= R[8]) {
        t = std::sqrt(R[4] - R[8] - R[0] + 1.0);
        q[1] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (R[2] - R[6]) * t;
        q[2] = (R[7] + R[5]) * t;
        q[0] = (R[1] + R[3]) * t;
    } else {
        t = std::sqrt(R[8] - R[0] - R[4] + 1.0);
        q[2] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (R[3] - R[1]) * t;
        q[0] = (R[2] + R[6]) * t;
        q[1] = (R[5] + R[7]) * t;
    }
    if (q[3] < 0)
        for (int a = 0; a < 4; ++a) q[a] = -q[a];
    const double nrm = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int a = 0; a < 4; ++a) q[a] /= nrm;
}
// z component of Eigen::AngleAxisd(q): angle = 2 atan2(|q_xyz|, w), axis = q_xyz / |q_xyz| - the rotation vector of a rotation
// THROUGH ITS QUATERNION, as src/optimizer.cpp:271-274 takes it; well conditioned near angle pi, where the logarithm of the
// rotation matrix (se3_log) loses ten digits (found by tools/fuzz_ref_backend.py against the compiled reference)
inline double rotation_vector_z(const double R[9]) {
    double q[4];
    quat_of_rotation(R, q);
    const double nv = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    return nv > 0 ? 2 * std::atan2(nv, q[3]) * q[2] / nv : 0.0;
}
inline void normalize_rotation(double R[9]) {
    // SE3Quat keeps a unit quaternion and renormalises it after every product / construction (normalizeRotation):
    // matrix -> quaternion (Eigen's conversion) -> normalise -> matrix
    double q[4];   // x, y, z, w
    quat_of_rotation(R, q);
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
inline Se3 se3_mul(const Se3& a, const Se3& b) {
    Se3 c;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) c.R[3 * i + j] = a.R[3 * i] * b.R[j] + a.R[3 * i + 1] * b.R[3 + j] + a.R[3 * i + 2] * b.R[6 + j];
        c.t[i] = a.R[3 * i] * b.t[0] + a.R[3 * i + 1] * b.t[1] + a.R[3 * i + 2] * b.t[2] + a.t[i];
    }
    normalize_rotation(c.R);
    return c;
}
inline Se3 se3_inv(const Se3& a) {
    Se3 c;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c.R[3 * i + j] = a.R[3 * j + i];
    for (int i = 0; i < 3; ++i) c.t[i] = -(c.R[3 * i] * a.t[0] + c.R[3 * i + 1] * a.t[1] + c.R[3 * i + 2] * a.t[2]);
    return c;
}
inline void skew3(const double v[3], double S[9]) {
    S[0] = 0; S[1] = -v[2]; S[2] = v[1];
    S[3] = v[2]; S[4] = 0; S[5] = -v[0];
    S[6] = -v[1]; S[7] = v[0]; S[8] = 0;
}
inline void mat3_mul(const double A[9], const double B[9], double C[9]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
// SE3Quat::exp, update = (omega, upsilon)
inline Se3 se3_exp(const double u[6]) {
    const double theta = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    double Om[9], Om2[9], V[9];
    skew3(u, Om);
    mat3_mul(Om, Om, Om2);
    Se3 T;
    if (theta < 0.00001) {
        for (int i = 0; i < 9; ++i) T.R[i] = (i % 4 == 0 ? 1.0 : 0.0) + Om[i] + Om2[i];
        std::memcpy(V, T.R, sizeof(V));
    } else {
        const double a = std::sin(theta) / theta, b = (1 - std::cos(theta)) / (theta * theta);
        const double c = (theta - std::sin(theta)) / (theta * theta * theta);
        for (int i = 0; i < 9; ++i) {
            T.R[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * Om[i] + b * Om2[i];
            V[i] = (i % 4 == 0 ? 1.0 : 0.0) + b * Om[i] + c * Om2[i];
        }
    }
    for (int i = 0; i < 3; ++i) T.t[i] = V[3 * i] * u[3] + V[3 * i + 1] * u[4] + V[3 * i + 2] * u[5];
    normalize_rotation(T.R);
    return T;
}
// SE3Quat::log -> (omega, upsilon)
inline void se3_log(const Se3& T, double out[6]) {
    const double* R = T.R;
    const double d = 0.5 * (R[0] + R[4] + R[8] - 1);
    const double dR[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
    double om[3], Om[9], Om2[9], Vi[9];
    if (d > 0.99999) {
        for (int i = 0; i < 3; ++i) om[i] = 0.5 * dR[i];
        skew3(om, Om);
        mat3_mul(Om, Om, Om2);
        for (int i = 0; i < 9; ++i) Vi[i] = (i % 4 == 0 ? 1.0 : 0.0) - 0.5 * Om[i] + (1. / 12.) * Om2[i];
    } else {
        const double theta = std::acos(d);
        const double k = theta / (2 * std::sqrt(1 - d * d));
        for (int i = 0; i < 3; ++i) om[i] = k * dR[i];
        skew3(om, Om);
        mat3_mul(Om, Om, Om2);
        const double c = (1 - theta / (2 * std::tan(theta / 2))) / (theta * theta);
        for (int i = 0; i < 9; ++i) Vi[i] = (i % 4 == 0 ? 1.0 : 0.0) - 0.5 * Om[i] + c * Om2[i];
    }
    for (int i = 0; i < 3; ++i) {
        out[i] = om[i];
        out[3 + i] = Vi[3 * i] * T.t[0] + Vi[3 * i + 1] * T.t[1] + Vi[3 * i + 2] * T.t[2];
    }
}

// SE3Quat::adj(): [R 0; skew(t) R  R] for the vector order (rotation, translation)
inline void se3_adj(const Se3& T, double A[36]) {
    double sk[9], sR[9];
    skew3(T.t, sk);
    mat3_mul(sk, T.R, sR);
    for (int i = 0; i < 36; ++i) A[i] = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            A[6 * i + j] = T.R[3 * i + j];
            A[6 * (i + 3) + (j + 3)] = T.R[3 * i + j];
            A[6 * (i + 3) + j] = sR[3 * i + j];
        }
}

}  // namespace
