// SE(3) algebra of g2o 20160424's SE3Quat on the device (and the host): exp / log with the vector order (omega, upsilon),
// products renormalised through a unit quaternion the way SE3Quat::normalizeRotation does, adj().  Shared by the
// pose-only bundle adjustment (pose_ba.hip) and the marginalising SE3-expmap bundle adjustment (ba.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

namespace se2gpu {
namespace {

struct Se3 {
    double R[9], t[3];   // x_c = R x_w + t, R row-major
};

__host__ __device__ inline void normalize_rotation(double R[9]) {
    // SE3Quat keeps a unit quaternion and renormalises it after every product / construction (normalizeRotation):
    // matrix -> quaternion (Eigen's conversion) -> normalise -> matrix
    double q[4];   // x, y, z, w
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R[7] - R[5]) * t;
        q[1] = (R[2] - R[6]) * t;
        q[2] = (R[3] - R[1]) * t;
    } else if (R[0] >= R[4] && R[0] >= R[8]) {   // largest diagonal element first (Eigen), written out per case
        t = sqrt(R[0] - R[4] - R[8] + 1.0);
        q[0] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (R[7] - R[5]) * t;
        q[1] = (R[3] + R[1]) * t;
        q[2] = (R[6] + R[2]) * t;
    } else if (R[4] > R[0] && R[4] >= R[8]) {
        t = sqrt(R[4] - R[8] - R[0] + 1.0);
        q[1] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (R[2] - R[6]) * t;
        q[2] = (R[7] + R[5]) * t;
        q[0] = (R[1] + R[3]) * t;
    } else {
        t = sqrt(R[8] - R[0] - R[4] + 1.0);
        q[2] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (R[3] - R[1]) * t;
        q[0] = (R[2] + R[6]) * t;
        q[1] = (R[5] + R[7]) * t;
    }
    if (q[3] < 0)
        #pragma unroll
        for (int a = 0; a < 4; ++a) q[a] = -q[a];
    const double nrm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    #pragma unroll
    for (int a = 0; a < 4; ++a) q[a] /= nrm;
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
__host__ __device__ inline Se3 se3_mul(const Se3& a, const Se3& b) {
    Se3 c;
    #pragma unroll
    for (int i = 0; i < 3; ++i) {
        #pragma unroll
        for (int j = 0; j < 3; ++j) c.R[3 * i + j] = a.R[3 * i] * b.R[j] + a.R[3 * i + 1] * b.R[3 + j] + a.R[3 * i + 2] * b.R[6 + j];
        c.t[i] = a.R[3 * i] * b.t[0] + a.R[3 * i + 1] * b.t[1] + a.R[3 * i + 2] * b.t[2] + a.t[i];
    }
    normalize_rotation(c.R);
    return c;
}
__host__ __device__ inline Se3 se3_inv(const Se3& a) {
    Se3 c;
    #pragma unroll
    for (int i = 0; i < 3; ++i)
        #pragma unroll
        for (int j = 0; j < 3; ++j) c.R[3 * i + j] = a.R[3 * j + i];
    #pragma unroll
    for (int i = 0; i < 3; ++i) c.t[i] = -(c.R[3 * i] * a.t[0] + c.R[3 * i + 1] * a.t[1] + c.R[3 * i + 2] * a.t[2]);
    return c;
}
__host__ __device__ inline void skew3(const double v[3], double S[9]) {
    S[0] = 0; S[1] = -v[2]; S[2] = v[1];
    S[3] = v[2]; S[4] = 0; S[5] = -v[0];
    S[6] = -v[1]; S[7] = v[0]; S[8] = 0;
}
__host__ __device__ inline void mat3_mul(const double A[9], const double B[9], double C[9]) {
    #pragma unroll
    for (int i = 0; i < 3; ++i)
        #pragma unroll
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
// SE3Quat::exp, update = (omega, upsilon)
__host__ __device__ inline Se3 se3_exp(const double u[6]) {
    const double theta = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    double Om[9], Om2[9], V[9];
    skew3(u, Om);
    mat3_mul(Om, Om, Om2);
    Se3 T;
    if (theta < 0.00001) {
        #pragma unroll
        for (int i = 0; i < 9; ++i) V[i] = T.R[i] = (i % 4 == 0 ? 1.0 : 0.0) + Om[i] + Om2[i];
    } else {
        const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta);
        const double c = (theta - sin(theta)) / (theta * theta * theta);
        #pragma unroll
        for (int i = 0; i < 9; ++i) {
            T.R[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * Om[i] + b * Om2[i];
            V[i] = (i % 4 == 0 ? 1.0 : 0.0) + b * Om[i] + c * Om2[i];
        }
    }
    #pragma unroll
    for (int i = 0; i < 3; ++i) T.t[i] = V[3 * i] * u[3] + V[3 * i + 1] * u[4] + V[3 * i + 2] * u[5];
    normalize_rotation(T.R);
    return T;
}
// SE3Quat::log -> (omega, upsilon)
__host__ __device__ inline void se3_log(const Se3& T, double out[6]) {
    const double* R = T.R;
    const double d = 0.5 * (R[0] + R[4] + R[8] - 1);
    const double dR[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
    double om[3], Om[9], Om2[9], Vi[9];
    if (d > 0.99999) {
        #pragma unroll
        for (int i = 0; i < 3; ++i) om[i] = 0.5 * dR[i];
        skew3(om, Om);
        mat3_mul(Om, Om, Om2);
        #pragma unroll
        for (int i = 0; i < 9; ++i) Vi[i] = (i % 4 == 0 ? 1.0 : 0.0) - 0.5 * Om[i] + (1. / 12.) * Om2[i];
    } else {
        const double theta = acos(d);
        const double k = theta / (2 * sqrt(1 - d * d));
        #pragma unroll
        for (int i = 0; i < 3; ++i) om[i] = k * dR[i];
        skew3(om, Om);
        mat3_mul(Om, Om, Om2);
        const double c = (1 - theta / (2 * tan(theta / 2))) / (theta * theta);
        #pragma unroll
        for (int i = 0; i < 9; ++i) Vi[i] = (i % 4 == 0 ? 1.0 : 0.0) - 0.5 * Om[i] + c * Om2[i];
    }
    #pragma unroll
    for (int i = 0; i < 3; ++i) {
        out[i] = om[i];
        out[3 + i] = Vi[3 * i] * T.t[0] + Vi[3 * i + 1] * T.t[1] + Vi[3 * i + 2] * T.t[2];
    }
}

// SE3Quat::adj(): [R 0; skew(t) R  R] for the vector order (rotation, translation), row-major 6x6
__host__ __device__ inline void se3_adj(const Se3& T, double A[36]) {
    double sk[9], sR[9];
    skew3(T.t, sk);
    mat3_mul(sk, T.R, sR);
    #pragma unroll
    for (int i = 0; i < 36; ++i) A[i] = 0;
    #pragma unroll
    for (int i = 0; i < 3; ++i)
        #pragma unroll
        for (int j = 0; j < 3; ++j) {
            A[6 * i + j] = T.R[3 * i + j];
            A[6 * (i + 3) + (j + 3)] = T.R[3 * i + j];
            A[6 * (i + 3) + j] = sR[3 * i + j];
        }
}
__host__ __device__ inline Se3 se3_load(const double* p) {   // pose12 = R row-major (9), t (3)
    Se3 T;
    #pragma unroll
    for (int i = 0; i < 9; ++i) T.R[i] = p[i];
    #pragma unroll
    for (int i = 0; i < 3; ++i) T.t[i] = p[9 + i];
    return T;
}
__host__ __device__ inline void se3_store(const Se3& T, double* p) {
    #pragma unroll
    for (int i = 0; i < 9; ++i) p[i] = T.R[i];
    #pragma unroll
    for (int i = 0; i < 3; ++i) p[9 + i] = T.t[i];
}

// ---- g2o::Isometry3D / slam3d arithmetic (VertexSE3, EdgeSE3, EdgeSE3Prior): plain matrix products (no quaternion
// renormalisation), the minimal vector (translation, compact quaternion q_xyz with w = sqrt(1 - |q_xyz|^2) >= 0) of
// g2o::internal::toVectorMQT / fromVectorMQT, and the exact derivatives of the edge errors at update = 0.
__host__ __device__ inline Se3 iso_mul(const Se3& a, const Se3& b) {
    Se3 c;
    #pragma unroll
    for (int i = 0; i < 3; ++i) {
        #pragma unroll
        for (int j = 0; j < 3; ++j) c.R[3 * i + j] = a.R[3 * i] * b.R[j] + a.R[3 * i + 1] * b.R[3 + j] + a.R[3 * i + 2] * b.R[6 + j];
        c.t[i] = a.R[3 * i] * b.t[0] + a.R[3 * i + 1] * b.t[1] + a.R[3 * i + 2] * b.t[2] + a.t[i];
    }
    return c;
}
// Eigen::Quaterniond(R) as (w, x, y, z), normalised, w >= 0; the three non-positive-trace cases written out (no
// dynamic indexing on the device)
__host__ __device__ inline void quat_of(const double* R, double q[4]) {
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[0] = 0.5 * t; t = 0.5 / t;
        q[1] = (R[7] - R[5]) * t; q[2] = (R[2] - R[6]) * t; q[3] = (R[3] - R[1]) * t;
    } else if (!(R[4] > R[0]) && !(R[8] > R[0])) {        // i = 0
        t = sqrt(R[0] - R[4] - R[8] + 1.0);
        q[1] = 0.5 * t; t = 0.5 / t;
        q[0] = (R[7] - R[5]) * t; q[2] = (R[3] + R[1]) * t; q[3] = (R[6] + R[2]) * t;
    } else if (R[4] > R[0] && !(R[8] > R[4])) {            // i = 1
        t = sqrt(R[4] - R[8] - R[0] + 1.0);
        q[2] = 0.5 * t; t = 0.5 / t;
        q[0] = (R[2] - R[6]) * t; q[3] = (R[7] + R[5]) * t; q[1] = (R[1] + R[3]) * t;
    } else {                                               // i = 2
        t = sqrt(R[8] - R[0] - R[4] + 1.0);
        q[3] = 0.5 * t; t = 0.5 / t;
        q[0] = (R[3] - R[1]) * t; q[1] = (R[2] + R[6]) * t; q[2] = (R[5] + R[7]) * t;
    }
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double s = (q[0] < 0 ? -1.0 : 1.0) / n;
    #pragma unroll
    for (int a = 0; a < 4; ++a) q[a] *= s;
}
__host__ __device__ inline void mat_of_quat(const double q[4], double* R) {
    const double tx = 2 * q[1], ty = 2 * q[2], tz = 2 * q[3];
    const double twx = tx * q[0], twy = ty * q[0], twz = tz * q[0];
    const double txx = tx * q[1], txy = ty * q[1], txz = tz * q[1];
    const double tyy = ty * q[2], tyz = tz * q[2], tzz = tz * q[3];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
__host__ __device__ inline void to_mqt(const Se3& T, double v[6]) {
    double q[4];
    quat_of(T.R, q);
    v[0] = T.t[0]; v[1] = T.t[1]; v[2] = T.t[2];
    v[3] = q[1]; v[4] = q[2]; v[5] = q[3];
}
__host__ __device__ inline Se3 from_mqt(const double v[6]) {
    Se3 T;
    const double w = 1 - (v[3] * v[3] + v[4] * v[4] + v[5] * v[5]);
    if (w < 0) {
        #pragma unroll
        for (int i = 0; i < 9; ++i) T.R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    } else {
        const double q[4] = {sqrt(w), v[3], v[4], v[5]};
        mat_of_quat(q, T.R);
    }
    T.t[0] = v[0]; T.t[1] = v[1]; T.t[2] = v[2];
    return T;
}
// d toVectorMQT(E * fromVectorMQT(d)) / d d at 0: [R_E 0; 0  w_E I + [q_E]x]
__host__ __device__ inline void mqt_jac_right(const Se3& E, double J[36]) {
    double q[4];
    quat_of(E.R, q);
    #pragma unroll
    for (int i = 0; i < 36; ++i) J[i] = 0;
    #pragma unroll
    for (int r = 0; r < 3; ++r)
        #pragma unroll
        for (int c = 0; c < 3; ++c) J[6 * r + c] = E.R[3 * r + c];
    J[21] = q[0]; J[22] = -q[3]; J[23] = q[2];
    J[27] = q[3]; J[28] = q[0]; J[29] = -q[1];
    J[33] = -q[2]; J[34] = q[1]; J[35] = q[0];
}
// e = toVectorMQT(A * fromVectorMQT(d)^-1 * B): d e / d d at 0: [-R_A  2 R_A [t_B]x; 0  -sgn (q_A (0, u) q_B)_xyz]
__host__ __device__ inline void mqt_jac_left_inv(const Se3& A, const Se3& B, double J[36]) {
    double qa[4], qb[4];
    quat_of(A.R, qa);
    quat_of(B.R, qb);
    const double wprod = qa[0] * qb[0] - qa[1] * qb[1] - qa[2] * qb[2] - qa[3] * qb[3];
    const double sgn = wprod < 0 ? -1.0 : 1.0;
    #pragma unroll
    for (int i = 0; i < 36; ++i) J[i] = 0;
    const double sk[9] = {0, -B.t[2], B.t[1], B.t[2], 0, -B.t[0], -B.t[1], B.t[0], 0};
    #pragma unroll
    for (int r = 0; r < 3; ++r)
        #pragma unroll
        for (int c = 0; c < 3; ++c) {
            J[6 * r + c] = -A.R[3 * r + c];
            J[6 * r + 3 + c] = 2 * (A.R[3 * r] * sk[c] + A.R[3 * r + 1] * sk[3 + c] + A.R[3 * r + 2] * sk[6 + c]);
        }
    #pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double u1 = c == 0 ? 1.0 : 0.0, u2 = c == 1 ? 1.0 : 0.0, u3 = c == 2 ? 1.0 : 0.0;
        const double m0 = -qa[1] * u1 - qa[2] * u2 - qa[3] * u3;
        const double m1 = qa[0] * u1 + qa[2] * u3 - qa[3] * u2;
        const double m2 = qa[0] * u2 - qa[1] * u3 + qa[3] * u1;
        const double m3 = qa[0] * u3 + qa[1] * u2 - qa[2] * u1;
        J[6 * 3 + 3 + c] = -sgn * (m0 * qb[1] + m1 * qb[0] + m2 * qb[3] - m3 * qb[2]);
        J[6 * 4 + 3 + c] = -sgn * (m0 * qb[2] - m1 * qb[3] + m2 * qb[0] + m3 * qb[1]);
        J[6 * 5 + 3 + c] = -sgn * (m0 * qb[3] + m1 * qb[2] - m2 * qb[1] + m3 * qb[0]);
    }
}

}  // namespace
}  // namespace se2gpu
