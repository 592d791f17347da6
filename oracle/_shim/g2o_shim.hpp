// ORACLE - TEST INFRASTRUCTURE ONLY (oracle/_ref).  Never compiled into, linked with or loaded by the product path.
//
// A stand-in for the Eigen 3 and g2o (tag 20160424_git) headers that /root/reference/src/EdgeSE2XYZ.cpp and
// include/se2lam/EdgeSE2XYZ.h include, just large enough for those two files to compile UNMODIFIED (oracle/Makefile, target
// `ref`): fixed-size matrices with comma initialisers, blocks, transposes and products; AngleAxis / Quaternion / Rotation2D;
// g2o's SE2, SE3Quat, CameraParameters::cam_map, internal::toEuler, skew, the two vertex types and BaseBinaryEdge's data
// members.  Neither library is installed in this image.  What oracle/_ref pins with it is what se2lam WROTE - the residual
// and the analytic Jacobians of EdgeSE2XYZ (src/EdgeSE2XYZ.cpp:61-106), SE2ToSE3 / SE3ToSE2 / d_inv_d_se2 (:16-39) and
// PreEdgeSE2 (include/se2lam/EdgeSE2XYZ.h:62-102) - evaluated through the reference's own statements.  The library
// formulas underneath (quaternion from angle-axis, q * v, SE3Quat's product / inverse / map, cam_map, toEuler) are written
// here from the published Eigen / g2o sources; g2o's block solver, Levenberg-Marquardt and CHOLMOD are not part of this at all.
#pragma once
#include <cmath>
#include <iostream>
#include <vector>

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW

namespace Eigen {

template <typename T, int R, int C> class Matrix;

// writable view of a BR x BC block of a matrix
template <typename M, typename T, int BR, int BC> class BlockRef {
    M& m;
    int r0, c0;
public:
    BlockRef(M& m_, int r, int c) : m(m_), r0(r), c0(c) {}
    T& operator()(int r, int c) { return m(r0 + r, c0 + c); }
    T operator()(int r, int c) const { return m(r0 + r, c0 + c); }
    BlockRef& operator=(const Matrix<T, BR, BC>& v) { for (int r = 0; r < BR; ++r) for (int c = 0; c < BC; ++c) m(r0 + r, c0 + c) = v(r, c); return *this; }
    BlockRef& operator=(const BlockRef& o) { return *this = Matrix<T, BR, BC>(o); }
    template <typename M2> BlockRef& operator=(const BlockRef<M2, T, BR, BC>& o) { return *this = Matrix<T, BR, BC>(o); }
    void setZero() { for (int r = 0; r < BR; ++r) for (int c = 0; c < BC; ++c) m(r0 + r, c0 + c) = T(0); }
    Matrix<T, BR, BC> operator-() const { return -Matrix<T, BR, BC>(*this); }   // (Matrix has the converting constructor)
};

template <typename T, int R, int C> class CommaInit {
    Matrix<T, R, C>& m;
    int i;
public:
    CommaInit(Matrix<T, R, C>& m_, T first) : m(m_), i(0) { put(first); }
    void put(T v) { m(i / C, i % C) = v; ++i; }
    CommaInit& operator,(T v) { put(v); return *this; }
};

template <typename T, int R, int C> class Matrix {
    T d[R * C];
public:
    enum { RowsAtCompileTime = R, ColsAtCompileTime = C };
    Matrix() { for (T& v : d) v = T(0); }
    Matrix(T x, T y) { static_assert(R * C == 2, "size"); d[0] = x; d[1] = y; }
    Matrix(T x, T y, T z) { static_assert(R * C == 3, "size"); d[0] = x; d[1] = y; d[2] = z; }
    template <typename M> Matrix(const BlockRef<M, T, R, C>& b) { for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) d[r * C + c] = b(r, c); }
    T& operator()(int r, int c) { return d[r * C + c]; }
    const T& operator()(int r, int c) const { return d[r * C + c]; }
    T& operator()(int i) { return d[i]; }
    const T& operator()(int i) const { return d[i]; }
    T& operator[](int i) { return d[i]; }
    const T& operator[](int i) const { return d[i]; }
    CommaInit<T, R, C> operator<<(T first) { return CommaInit<T, R, C>(*this, first); }
    void setZero() { for (T& v : d) v = T(0); }
    void setIdentity() { setZero(); for (int i = 0; i < (R < C ? R : C); ++i) (*this)(i, i) = T(1); }
    static Matrix Zero() { return Matrix(); }
    static Matrix Identity() { Matrix m; m.setIdentity(); return m; }
    static Matrix UnitX() { Matrix m; m[0] = T(1); return m; }
    static Matrix UnitY() { Matrix m; m[1] = T(1); return m; }
    static Matrix UnitZ() { Matrix m; m[2] = T(1); return m; }
    Matrix<T, C, R> transpose() const { Matrix<T, C, R> t; for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) t(c, r) = (*this)(r, c); return t; }
    template <int BR, int BC> BlockRef<Matrix, T, BR, BC> block(int r, int c) { return BlockRef<Matrix, T, BR, BC>(*this, r, c); }
    template <int BR, int BC> Matrix<T, BR, BC> block(int r, int c) const { Matrix<T, BR, BC> v; for (int i = 0; i < BR; ++i) for (int j = 0; j < BC; ++j) v(i, j) = (*this)(r + i, c + j); return v; }
    template <int N> BlockRef<Matrix, T, N, 1> head() { static_assert(C == 1, "vector"); return BlockRef<Matrix, T, N, 1>(*this, 0, 0); }
    template <int N> Matrix<T, N, 1> head() const { Matrix<T, N, 1> v; for (int i = 0; i < N; ++i) v[i] = d[i]; return v; }
    Matrix operator-() const { Matrix m; for (int i = 0; i < R * C; ++i) m.d[i] = -d[i]; return m; }
    Matrix& operator+=(const Matrix& o) { for (int i = 0; i < R * C; ++i) d[i] += o.d[i]; return *this; }
    Matrix& operator*=(T s) { for (int i = 0; i < R * C; ++i) d[i] *= s; return *this; }
    T dot(const Matrix& o) const { T s = T(0); for (int i = 0; i < R * C; ++i) s += d[i] * o.d[i]; return s; }
    T squaredNorm() const { return dot(*this); }
    T norm() const { return std::sqrt(squaredNorm()); }
    Matrix cross(const Matrix& o) const {
        static_assert(R * C == 3, "cross");
        return Matrix(d[1] * o.d[2] - d[2] * o.d[1], d[2] * o.d[0] - d[0] * o.d[2], d[0] * o.d[1] - d[1] * o.d[0]);
    }
};
template <typename T, int R, int C> Matrix<T, R, C> operator+(const Matrix<T, R, C>& a, const Matrix<T, R, C>& b) { Matrix<T, R, C> m; for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) m(r, c) = a(r, c) + b(r, c); return m; }
template <typename T, int R, int C> Matrix<T, R, C> operator-(const Matrix<T, R, C>& a, const Matrix<T, R, C>& b) { Matrix<T, R, C> m; for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) m(r, c) = a(r, c) - b(r, c); return m; }
template <typename T, int R, int C, typename M> Matrix<T, R, C> operator-(const Matrix<T, R, C>& a, const BlockRef<M, T, R, C>& b) { return a - Matrix<T, R, C>(b); }
template <typename T, int R, int K, int C> Matrix<T, R, C> operator*(const Matrix<T, R, K>& a, const Matrix<T, K, C>& b) {
    Matrix<T, R, C> m;
    for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) { T s = T(0); for (int k = 0; k < K; ++k) s += a(r, k) * b(k, c); m(r, c) = s; }
    return m;
}
template <typename T, int R, int C> Matrix<T, R, C> operator*(const Matrix<T, R, C>& a, T s) { Matrix<T, R, C> m; for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) m(r, c) = a(r, c) * s; return m; }
template <typename T, int R, int C> Matrix<T, R, C> operator*(T s, const Matrix<T, R, C>& a) { return a * s; }

typedef Matrix<double, 2, 1> Vector2d;
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<double, 2, 2> Matrix2d;
typedef Matrix<double, 3, 3> Matrix3d;

class AngleAxisd {
public:
    double angle_;
    Vector3d axis_;
    AngleAxisd(double a, const Vector3d& ax) : angle_(a), axis_(ax) {}
};

class Quaterniond {   // Eigen/src/Geometry/Quaternion.h
public:
    double x_, y_, z_, w_;
    Quaterniond() : x_(0), y_(0), z_(0), w_(1) {}
    Quaterniond(double w, double x, double y, double z) : x_(x), y_(y), z_(z), w_(w) {}
    Quaterniond(const AngleAxisd& aa) {
        const double ha = 0.5 * aa.angle_;
        w_ = std::cos(ha);
        const double s = std::sin(ha);
        x_ = s * aa.axis_[0]; y_ = s * aa.axis_[1]; z_ = s * aa.axis_[2];
    }
    explicit Quaterniond(const Matrix3d& m) {   // quaternionbase_assign_impl<Other, 3, 3>
        double t = m(0, 0) + m(1, 1) + m(2, 2);
        if (t > 0) {
            t = std::sqrt(t + 1.0);
            w_ = 0.5 * t;
            t = 0.5 / t;
            x_ = (m(2, 1) - m(1, 2)) * t; y_ = (m(0, 2) - m(2, 0)) * t; z_ = (m(1, 0) - m(0, 1)) * t;
        } else {
            int i = 0;
            if (m(1, 1) > m(0, 0)) i = 1;
            if (m(2, 2) > m(i, i)) i = 2;
            const int j = (i + 1) % 3, k = (j + 1) % 3;
            t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0);
            double q[3];
            q[i] = 0.5 * t;
            t = 0.5 / t;
            w_ = (m(k, j) - m(j, k)) * t;
            q[j] = (m(j, i) + m(i, j)) * t;
            q[k] = (m(k, i) + m(i, k)) * t;
            x_ = q[0]; y_ = q[1]; z_ = q[2];
        }
    }
    double w() const { return w_; } double x() const { return x_; } double y() const { return y_; } double z() const { return z_; }
    Vector3d vec() const { return Vector3d(x_, y_, z_); }
    Quaterniond conjugate() const { return Quaterniond(w_, -x_, -y_, -z_); }
    void normalize() { const double n = std::sqrt(w_ * w_ + x_ * x_ + y_ * y_ + z_ * z_); w_ /= n; x_ /= n; y_ /= n; z_ /= n; }
    void negate() { w_ = -w_; x_ = -x_; y_ = -y_; z_ = -z_; }
    Quaterniond operator*(const Quaterniond& b) const {   // quat_product
        return Quaterniond(w_ * b.w_ - x_ * b.x_ - y_ * b.y_ - z_ * b.z_, w_ * b.x_ + x_ * b.w_ + y_ * b.z_ - z_ * b.y_,
                           w_ * b.y_ + y_ * b.w_ + z_ * b.x_ - x_ * b.z_, w_ * b.z_ + z_ * b.w_ + x_ * b.y_ - y_ * b.x_);
    }
    Quaterniond& operator*=(const Quaterniond& b) { *this = *this * b; return *this; }
    Vector3d operator*(const Vector3d& v) const {          // _transformVector
        Vector3d uv = vec().cross(v);
        uv += uv;
        return v + w_ * uv + vec().cross(uv);
    }
    Matrix3d toRotationMatrix() const {
        Matrix3d res;
        const double tx = 2 * x_, ty = 2 * y_, tz = 2 * z_;
        const double twx = tx * w_, twy = ty * w_, twz = tz * w_, txx = tx * x_, txy = ty * x_, txz = tz * x_, tyy = ty * y_, tyz = tz * y_, tzz = tz * z_;
        res(0, 0) = 1 - (tyy + tzz); res(0, 1) = txy - twz; res(0, 2) = txz + twy;
        res(1, 0) = txy + twz; res(1, 1) = 1 - (txx + tzz); res(1, 2) = tyz - twx;
        res(2, 0) = txz - twy; res(2, 1) = tyz + twx; res(2, 2) = 1 - (txx + tyy);
        return res;
    }
    Matrix3d matrix() const { return toRotationMatrix(); }
};

class Rotation2Dd {
    double a;
public:
    Rotation2Dd(double angle = 0) : a(angle) {}
    double& angle() { return a; }
    double angle() const { return a; }
    Rotation2Dd inverse() const { return Rotation2Dd(-a); }
    Matrix2d toRotationMatrix() const { Matrix2d m; const double s = std::sin(a), c = std::cos(a); m << c, -s, s, c; return m; }
    Vector2d operator*(const Vector2d& v) const { return toRotationMatrix() * v; }
    Rotation2Dd operator*(const Rotation2Dd& o) const { return Rotation2Dd(a + o.a); }
};

}  // namespace Eigen

namespace g2o {

typedef Eigen::Vector2d Vector2D;
typedef Eigen::Vector3d Vector3D;
typedef Eigen::Matrix2d Matrix2D;
typedef Eigen::Matrix3d Matrix3D;

inline double normalize_theta(double theta) {   // g2o/stuff/misc.h
    if (theta >= -M_PI && theta < M_PI) return theta;
    const double multiplier = std::floor(theta / (2 * M_PI));
    theta = theta - multiplier * 2 * M_PI;
    if (theta >= M_PI) theta -= 2 * M_PI;
    if (theta < -M_PI) theta += 2 * M_PI;
    return theta;
}

class SE2 {   // g2o/types/slam2d/se2.h
    Eigen::Rotation2Dd _R;
    Vector2D _t;
public:
    SE2() : _R(0), _t(0, 0) {}
    SE2(double x, double y, double theta) : _R(theta), _t(x, y) {}
    const Vector2D& translation() const { return _t; }
    const Eigen::Rotation2Dd& rotation() const { return _R; }
    SE2 inverse() const {
        SE2 ret;
        ret._R = _R.inverse();
        ret._R.angle() = normalize_theta(ret._R.angle());
        ret._t = ret._R * (_t * -1.);
        return ret;
    }
    Vector3D toVector() const { return Vector3D(_t(0), _t(1), _R.angle()); }
};

class SE3Quat {   // g2o/types/slam3d/se3quat.h
    Eigen::Quaterniond _r;
    Vector3D _t;
public:
    SE3Quat() : _t(0, 0, 0) {}
    const Vector3D& translation() const { return _t; }
    const Eigen::Quaterniond& rotation() const { return _r; }
    void setTranslation(const Vector3D& t) { _t = t; }
    void setRotation(const Eigen::Quaterniond& r) { _r = r; }
    void normalizeRotation() { if (_r.w() < 0) _r.negate(); _r.normalize(); }
    SE3Quat operator*(const SE3Quat& tr2) const {
        SE3Quat result(*this);
        result._t += _r * tr2._t;
        result._r *= tr2._r;
        result.normalizeRotation();
        return result;
    }
    SE3Quat inverse() const {
        SE3Quat ret;
        ret._r = _r.conjugate();
        ret._t = ret._r * (_t * -1.);
        return ret;
    }
    Vector3D map(const Vector3D& xyz) const { return _r * xyz + _t; }
};

inline Matrix3D skew(const Vector3D& v) {   // g2o/types/sba/types_six_dof_expmap.h (se3_ops)
    Matrix3D m;
    m(0, 1) = -v(2); m(0, 2) = v(1); m(1, 2) = -v(0);
    m(1, 0) = v(2); m(2, 0) = -v(1); m(2, 1) = v(0);
    return m;
}

namespace internal {
inline Vector3D toEuler(const Matrix3D& R) {   // g2o/types/slam3d/isometry3d_mappings.cpp
    Eigen::Quaterniond q(R);
    const double q0 = q.w(), q1 = q.x(), q2 = q.y(), q3 = q.z();
    const double roll = std::atan2(2 * (q0 * q1 + q2 * q3), 1 - 2 * (q1 * q1 + q2 * q2));
    const double pitch = std::asin(2 * (q0 * q2 - q3 * q1));
    const double yaw = std::atan2(2 * (q0 * q3 + q1 * q2), 1 - 2 * (q2 * q2 + q3 * q3));
    return Vector3D(roll, pitch, yaw);
}
}  // namespace internal

class CameraParameters {   // g2o/types/sba/types_six_dof_expmap.{h,cpp}
public:
    double focal_length;
    Vector2D principle_point;
    double baseline;
    CameraParameters() : focal_length(1.), principle_point(0., 0.), baseline(0.5) {}
    CameraParameters(double f, const Vector2D& pp, double b) : focal_length(f), principle_point(pp), baseline(b) {}
    Vector2D cam_map(const Vector3D& trans_xyz) const {
        Vector2D proj(trans_xyz(0) / trans_xyz(2), trans_xyz(1) / trans_xyz(2));
        Vector2D res;
        res[0] = proj[0] * focal_length + principle_point[0];
        res[1] = proj[1] * focal_length + principle_point[1];
        return res;
    }
};

namespace HyperGraph { class Vertex { public: virtual ~Vertex() {} }; }

template <int D, typename T> class BaseVertex : public HyperGraph::Vertex {
protected:
    T _estimate;
public:
    enum { Dimension = D };
    const T& estimate() const { return _estimate; }
    void setEstimate(const T& e) { _estimate = e; }
};
class VertexSE2 : public BaseVertex<3, SE2> {};
class VertexSBAPointXYZ : public BaseVertex<3, Vector3D> {};

template <int D, typename E, typename VertexXi, typename VertexXj> class BaseBinaryEdge {
protected:
    std::vector<HyperGraph::Vertex*> _vertices;
    E _measurement;
    Eigen::Matrix<double, D, 1> _error;
    Eigen::Matrix<double, D, VertexXi::Dimension> _jacobianOplusXi;
    Eigen::Matrix<double, D, VertexXj::Dimension> _jacobianOplusXj;
public:
    BaseBinaryEdge() : _vertices(2, nullptr) {}
    virtual ~BaseBinaryEdge() {}
    void setVertex(size_t i, HyperGraph::Vertex* v) { _vertices[i] = v; }
    void setMeasurement(const E& m) { _measurement = m; }
    const Eigen::Matrix<double, D, 1>& error() const { return _error; }
    const Eigen::Matrix<double, D, VertexXi::Dimension>& jacobianOplusXi() const { return _jacobianOplusXi; }
    const Eigen::Matrix<double, D, VertexXj::Dimension>& jacobianOplusXj() const { return _jacobianOplusXj; }
    virtual void computeError() = 0;
    virtual void linearizeOplus() = 0;
    virtual bool read(std::istream& is) = 0;
    virtual bool write(std::ostream& os) const = 0;
};

}  // namespace g2o
