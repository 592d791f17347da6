// SE(2) odometry pre-integration between key frames - host code, mirrors the reference's names.
//   struct PreSE2                      /root/reference/include/se2lam/Frame.h:20-24
//   Se2::operator-                     /root/reference/src/Config.cpp:215-224   (float arithmetic, as the reference)
//   Track::updateFramePose (preint.)   /root/reference/src/Track.cpp:169-187
//   Track::resetLocalTrack (reset)     /root/reference/src/Track.cpp:199-204
// The result feeds the PreEdgeSE2 odometry edges of the bundle adjustment: Map::loadLocalGraph turns `meas` into the
// edge measurement and `cov` into its information matrix (src/Map.cpp:943-949) - see preSE2Information below and
// addEdgeSE2 in optimizer.h.  This is a 3x3 sequential recursion per odometry message: it stays on the host (SURVEY.md
// section 8, row a26); the restatement used by the synthetic graph generator is se2lam_amd/synth.py:_preintegrate.
#pragma once
#include <cmath>

#include "types.h"

namespace se2lam_amd {

struct PreSE2 {
    double meas[3];
    double cov[9];  // 3*3, RowMajor
};

inline void resetPreSE2(PreSE2& p) {
    for (int i = 0; i < 3; ++i) p.meas[i] = 0;
    for (int i = 0; i < 9; ++i) p.cov[i] = 0;
}

// odometry poses as the reference stores them (Se2: float x, y, theta)
struct Se2f {
    float x, y, theta;
};
inline double normalize_angle(double theta) {   // /root/reference/include/se2lam/Config.h:28-42: wrap to [-pi, pi)
    const double pi = 3.14159265358979323846;
    if (theta >= -pi && theta < pi) return theta;
    const double multiplier = std::floor(theta / (2 * pi));
    theta = theta - multiplier * 2 * pi;
    if (theta >= pi) theta -= 2 * pi;
    if (theta < -pi) theta += 2 * pi;
    return theta;
}
// a - b: pose of a expressed in the frame of b
inline Se2f se2Minus(const Se2f& a, const Se2f& b) {
    const float dx = a.x - b.x, dy = a.y - b.y;
    const float dth = (float)normalize_angle(a.theta - b.theta);   // float difference, wrapped in double, stored float
    const float c = std::cos(b.theta), s = std::sin(b.theta);
    return Se2f{c * dx + s * dy, -s * dx + c * dy, dth};
}

// One odometry message: odok = odom_now - odom_last; noise = Config::ODO_{X,Y,T}_NOISE (standard deviations; floats in the
// reference, and their squares are FLOAT products there - Track.cpp:183-185 - which is what the compiled reference showed:
// 0.002f * 0.002f = 4.00000044e-06, not 4.00000038e-06)
inline void updatePreSE2(PreSE2& p, const Se2f& odok, float noise_x, float noise_y, float noise_t) {
    const double ox = odok.x, oy = odok.y;
    const double c = std::cos(p.meas[2]), s = std::sin(p.meas[2]);   // Phi_ik = Rotation2D(meas[2])
    const double px = c * ox - s * oy, py = s * ox + c * oy;          // Phi_ik * odork
    // Ak = I with Ak(0:2, 2) = Phi_ik * (-oy, ox);  Bk = blkdiag(Phi_ik, 1)
    const double a02 = c * (-oy) - s * ox, a12 = s * (-oy) + c * ox;
    p.meas[0] += px;
    p.meas[1] += py;
    p.meas[2] += odok.theta;
    const double A[9] = {1, 0, a02, 0, 1, a12, 0, 0, 1};
    const double B[9] = {c, -s, 0, s, c, 0, 0, 0, 1};
    const float sx = noise_x * noise_x, sy = noise_y * noise_y, st = noise_t * noise_t;
    const double Sv[3] = {sx, sy, st};
    double AS[9], out[9];
    for (int r = 0; r < 3; ++r)
        for (int q = 0; q < 3; ++q) {
            double v = 0;
            for (int k = 0; k < 3; ++k) v += A[r * 3 + k] * p.cov[k * 3 + q];
            AS[r * 3 + q] = v;
        }
    for (int r = 0; r < 3; ++r)
        for (int q = 0; q < 3; ++q) {
            double v = 0;
            for (int k = 0; k < 3; ++k) v += AS[r * 3 + k] * A[q * 3 + k];          // A Sigma A^T
            for (int k = 0; k < 3; ++k) v += B[r * 3 + k] * Sv[k] * B[q * 3 + k];   // + B Sigma_v B^T
            out[r * 3 + q] = v;
        }
    for (int i = 0; i < 9; ++i) p.cov[i] = out[i];
}

// information matrix of the PreEdgeSE2 edge: cov^-1 (row-major 3x3); false if cov is singular
inline bool preSE2Information(const PreSE2& p, Matrix3D& info) {
    const double* m = p.cov;
    const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    if (!(std::fabs(det) > 0.0)) return false;
    const double id = 1.0 / det;
    info.m[0] = c00 * id;
    info.m[1] = (m[2] * m[7] - m[1] * m[8]) * id;
    info.m[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    info.m[3] = c01 * id;
    info.m[4] = (m[0] * m[8] - m[2] * m[6]) * id;
    info.m[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    info.m[6] = c02 * id;
    info.m[7] = (m[1] * m[6] - m[0] * m[7]) * id;
    info.m[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    return true;
}

}  // namespace se2lam_amd
