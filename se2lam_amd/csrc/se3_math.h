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

}  // namespace
}  // namespace se2gpu
