// Conversions between the real OpenCV / g2o / Eigen types and the POD mirrors of types.h, compiled in only where the
// third-party headers exist (`__has_include`); included at the end of types.h.  None of these is a cast: see the table
// at the top of types.h.  (This image has none of the three libraries, so this file is exercised by the maintainers'
// build, not by this repository's tests.)
#pragma once

#if defined(__has_include)
#if __has_include(<opencv2/core.hpp>)
#include <opencv2/core.hpp>
#define SE2LAM_AMD_HAVE_OPENCV 1
namespace se2lam_amd {
static_assert(sizeof(cv::KeyPoint) == sizeof(KeyPoint) && sizeof(cv::Point2f) == sizeof(Point2f),
              "cv::KeyPoint / cv::Point2f are passed through by reinterpretation");
inline const KeyPoint* mirror(const std::vector<cv::KeyPoint>& v) { return reinterpret_cast<const KeyPoint*>(v.data()); }
inline Mat8U view8U(const cv::Mat& m) {          // non-owning view of a CV_8UC1 matrix
    Mat8U o;
    if (m.empty()) return o;
    CV_Assert(m.type() == CV_8UC1);
    o.rows = m.rows; o.cols = m.cols; o.step = static_cast<size_t>(m.step);   // cv::MatStep converts to size_t; a stand-in's plain size_t passes through
    o.data = m.ptr<uint8_t>(0);
    return o;
}
inline MatF toMatF(const cv::Mat& m) {
    CV_Assert(m.type() == CV_32F);
    MatF o(m.rows, m.cols);
    for (int r = 0; r < m.rows; ++r)
        for (int c = 0; c < m.cols; ++c) o.at<float>(r, c) = m.at<float>(r, c);
    return o;
}
}  // namespace se2lam_amd
#endif

#if __has_include(<Eigen/Core>)
#include <Eigen/Core>
#define SE2LAM_AMD_HAVE_EIGEN 1
namespace se2lam_amd {
inline Vector2D mirror(const Eigen::Vector2d& v) { return Vector2D(v[0], v[1]); }
inline Vector3D mirror(const Eigen::Vector3d& v) { return Vector3D(v[0], v[1], v[2]); }
inline Matrix2D mirror(const Eigen::Matrix2d& a) { return Matrix2D(a(0, 0), a(0, 1), a(1, 0), a(1, 1)); }   // (r, c): storage-order agnostic
inline Matrix3D mirror(const Eigen::Matrix3d& a) {
    Matrix3D o;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) o(r, c) = a(r, c);
    return o;
}
inline Matrix6d mirror(const Eigen::Matrix<double, 6, 6>& a) {
    Matrix6d o;
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) o(r, c) = a(r, c);
    return o;
}
inline Eigen::Vector3d toEigen(const Vector3D& v) { return Eigen::Vector3d(v.v[0], v.v[1], v.v[2]); }
}  // namespace se2lam_amd
#endif

#if __has_include(<g2o/types/slam2d/se2.h>) && __has_include(<g2o/types/sba/types_six_dof_expmap.h>)
#include <g2o/types/sba/types_six_dof_expmap.h>
#include <g2o/types/slam2d/se2.h>
#define SE2LAM_AMD_HAVE_G2O 1
namespace se2lam_amd {
inline SE2 mirror(const g2o::SE2& p) { return SE2(p.translation()[0], p.translation()[1], p.rotation().angle()); }
inline g2o::SE2 toG2o(const SE2& p) { return g2o::SE2(p.x, p.y, p.theta); }
inline SE3Quat mirror(const g2o::SE3Quat& q) {   // unit quaternion -> rotation matrix, row-major
    SE3Quat o;
    const Eigen::Matrix3d R = q.rotation().toRotationMatrix();
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) o.R[3 * r + c] = R(r, c);
        o.t[r] = q.translation()[r];
    }
    return o;
}
inline g2o::SE3Quat toG2o(const SE3Quat& q) {
    Eigen::Matrix3d R;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) R(r, c) = q.R[3 * r + c];
    return g2o::SE3Quat(R, Eigen::Vector3d(q.t[0], q.t[1], q.t[2]));
}
}  // namespace se2lam_amd
#endif
#endif  // __has_include
