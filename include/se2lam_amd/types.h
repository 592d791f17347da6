// Value types that cross the reference's hot-path interfaces.
//
// The reference passes OpenCV / g2o / Eigen values (cv::Mat, cv::KeyPoint, g2o::SE2, g2o::SE3Quat, Eigen matrices).
// This image has none of those libraries, so the adapters are written against the POD mirrors below; where the real
// headers ARE present (`__has_include`), overloads taking the real types are compiled in as well and convert
// EXPLICITLY (conversions.h).  What is and is not a plain reinterpretation:
//   cv::KeyPoint, cv::Point2f   layout-identical to KeyPoint / Point2f (28 / 8 bytes, checked by static_assert when
//                               OpenCV is present): vectors of them are passed through without a copy
//   g2o::SE2                    Rotation2Dd + Vector2d (angle first): NOT the (x, y, theta) of SE2 -> converted
//   g2o::SE3Quat                unit quaternion + translation: NOT a rotation matrix -> converted
//   Eigen::Matrix2d / 3d        column-major by default; Matrix2D / Matrix3D here are row-major -> converted element-wise
//   cv::Mat                     reference-counted header; Mat8U / MatF are non-owning views of its data / step
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../se2gpu.h"

namespace se2lam_amd {

struct Point2f {          // cv::Point2f
    float x = 0, y = 0;
};

struct Point3f {          // cv::Point3f
    float x, y, z;
};

struct KeyPoint {         // cv::KeyPoint (28 bytes; layout == se2gpu_keypoint)
    Point2f pt;
    float size = 0, angle = -1, response = 0;
    int octave = 0, class_id = -1;
};
static_assert(sizeof(KeyPoint) == sizeof(se2gpu_keypoint), "cv::KeyPoint layout");

struct Mat8U {            // a CV_8UC1 cv::Mat view (rows x cols, row pitch `step`), or an owning n x 32 descriptor matrix
    int rows = 0, cols = 0;
    size_t step = 0;
    const uint8_t* data = nullptr;
    std::vector<uint8_t> owned;
    bool empty() const { return rows == 0 || cols == 0 || data == nullptr; }
    void create(int r, int c) {
        owned.assign((size_t)r * c, 0);
        rows = r; cols = c; step = (size_t)c; data = owned.data();
    }
    void release() { owned.clear(); rows = cols = 0; step = 0; data = nullptr; }
    uint8_t* ptr(int r) { return owned.data() + (size_t)r * step; }
    const uint8_t* ptr(int r) const { return data + (size_t)r * step; }
};

struct MatF {             // a CV_32F cv::Mat (Config::Kcam 3x3, Config::bTc 4x4, KeyFrame::Tcw 4x4), owning, row-major
    int rows = 0, cols = 0;
    std::vector<float> v;
    MatF() = default;
    MatF(int r, int c) : rows(r), cols(c), v((size_t)r * c, 0.f) {}
    static MatF eye(int n) { MatF m(n, n); for (int i = 0; i < n; ++i) m.v[(size_t)i * n + i] = 1.f; return m; }
    template <typename T> T& at(int r, int c) { static_assert(sizeof(T) == sizeof(float), "CV_32F"); return v[(size_t)r * cols + c]; }
    template <typename T> const T& at(int r, int c) const { static_assert(sizeof(T) == sizeof(float), "CV_32F"); return v[(size_t)r * cols + c]; }
};

struct SE2 {                                                 // g2o::SE2 by value: translation + angle
    double x = 0, y = 0, theta = 0;
    SE2() = default;
    SE2(double x_, double y_, double theta_) : x(x_), y(y_), theta(theta_) {}
};
struct Vector2D {                                            // g2o::Vector2D
    double v[2] = {0, 0};
    Vector2D() = default;
    Vector2D(double a, double b) : v{a, b} {}
    explicit Vector2D(const double* p) : v{p[0], p[1]} {}
    double& operator()(int i) { return v[i]; }
    double operator()(int i) const { return v[i]; }
};
struct Vector3D {                                            // g2o::Vector3D
    double v[3] = {0, 0, 0};
    Vector3D() = default;
    Vector3D(double a, double b, double c) : v{a, b, c} {}
    explicit Vector3D(const double* p) : v{p[0], p[1], p[2]} {}    // Vector3D(meas.meas), Map.cpp:951
    double& operator()(int i) { return v[i]; }
    double operator()(int i) const { return v[i]; }
};
struct Matrix2D {                                            // row-major 2x2
    double m[4] = {0, 0, 0, 0};
    Matrix2D() = default;
    Matrix2D(double a, double b, double c, double d) : m{a, b, c, d} {}
    static Matrix2D Identity() { return Matrix2D(1, 0, 0, 1); }
    double& operator()(int r, int c) { return m[2 * r + c]; }
    double operator()(int r, int c) const { return m[2 * r + c]; }
    Matrix2D inverse() const {                               // Sigma_all.inverse(), Map.cpp:1049
        const double id = 1.0 / (m[0] * m[3] - m[1] * m[2]);
        return Matrix2D(m[3] * id, -m[1] * id, -m[2] * id, m[0] * id);
    }
};
struct Matrix3D {                                            // row-major 3x3
    double m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double& operator()(int r, int c) { return m[3 * r + c]; }
    double operator()(int r, int c) const { return m[3 * r + c]; }
};
struct Matrix6d {                                            // row-major 6x6 (g2o::Matrix6d)
    double m[36] = {};
    double& operator()(int r, int c) { return m[6 * r + c]; }
    double operator()(int r, int c) const { return m[6 * r + c]; }
};
struct SE3Quat {                                             // g2o::SE3Quat by value, as rotation matrix + translation
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double t[3] = {0, 0, 0};
};

// toSE3Quat(const cv::Mat&) of the reference's converter.h for a 4x4 CV_32F pose (works on MatF and on cv::Mat)
template <typename MatT>
inline SE3Quat toSE3Quat(const MatT& T) {
    SE3Quat q;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) q.R[3 * r + c] = (double)T.template at<float>(r, c);
        q.t[r] = (double)T.template at<float>(r, 3);
    }
    return q;
}

// The reference has no error channel on these surfaces (asserts only); the adapters throw on a library error so a
// failure cannot pass silently.  Nothing is thrown across the C ABI itself.
inline void check(int rc, const char* what) {
    if (rc != SE2GPU_OK) throw std::runtime_error(std::string(what) + ": " + se2gpu_last_error());
}

}  // namespace se2lam_amd

#include "conversions.h"
