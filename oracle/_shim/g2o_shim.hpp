// ORACLE - TEST INFRASTRUCTURE ONLY (oracle/_ref).  Never compiled into, linked with or loaded by the product path.
//
// A stand-in for the Eigen 3 and g2o (tag 20160424_git) headers that /root/reference/src/EdgeSE2XYZ.cpp, src/optimizer.cpp,
// src/converter.cpp, src/sparsifier.cpp and include/se2lam/{EdgeSE2XYZ,optimizer,converter,sparsifier}.h include, just large enough for those files to compile
// UNMODIFIED (oracle/Makefile, target `ref`): fixed-size matrices with comma initialisers, blocks, transposes and products;
// AngleAxis / Quaternion / Rotation2D / Isometry3d; g2o's SE2, SE3Quat (product, inverse, map, log, adj), CameraParameters::cam_map,
// internal::toEuler / toSE3Quat / fromSE3Quat, skew, Huber's rho, and a RECORDING graph - vertices with id / fixed /
// marginalized, edges with vertices / measurement / information / robust kernel / parameter ids / level, a SparseOptimizer
// that stores what addVertex / addEdge / addParameter hand it.  Neither library is installed in this image.  What oracle/_ref
// pins with it is what se2lam WROTE - the residual and the analytic Jacobians of EdgeSE2XYZ (src/EdgeSE2XYZ.cpp:61-106),
// SE2ToSE3 / SE3ToSE2 / d_inv_d_se2 (:16-39), PreEdgeSE2 (include/se2lam/EdgeSE2XYZ.h:62-102), and all of src/optimizer.cpp:
// what each add* call puts into the graph, the plane-motion priors, EdgeSE3ExpmapPrior, the information reordering of
// addEdgeSE3Expmap, Jl / invJl / invJJl - and all of src/sparsifier.cpp (numeric Jacobians, marginalisation, InfoSE3 with its
// SVD clamp; MatrixXd, LDL', inverse() and JacobiSVD below are plain exact solvers, not Eigen's algorithms) - evaluated
// through the reference's own statements.  The library formulas
// underneath (quaternion from angle-axis / matrix, q * v, SE3Quat's product / inverse / map / log / adj, cam_map, toEuler) are
// written here from the published Eigen / g2o sources; g2o's own edge types (EdgeSE3Expmap, EdgeProjectXYZ2UV, EdgeSE3,
// EdgeSE3Prior, EdgeSE3PointXYZ) are recorded, not evaluated; the block solver, Levenberg-Marquardt and CHOLMOD are empty
// types: nothing is ever optimised here.
#pragma once
#include <cmath>
#include <algorithm>
#include <deque>
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <set>
#include <type_traits>
#include <vector>

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#define G2O_TYPES_SBA_API

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
    BlockRef& operator+=(const Matrix<T, BR, BC>& v) { for (int r = 0; r < BR; ++r) for (int c = 0; c < BC; ++c) m(r0 + r, c0 + c) += v(r, c); return *this; }
    Matrix<T, BR, BC> operator-() const { return -Matrix<T, BR, BC>(*this); }   // (Matrix has the converting constructor)
};

class MatrixXd;
template <typename T> using aligned_allocator = std::allocator<T>;
enum { ComputeFullU = 1, ComputeFullV = 2 };

// view of a run-time sized block (m.block(r, c, nr, nc)); M may be const
template <typename M> class DynBlock {
    M& m;
    int r0, c0, nr, nc;
public:
    DynBlock(M& m_, int r, int c, int nr_, int nc_) : m(m_), r0(r), c0(c), nr(nr_), nc(nc_) {}
    int rows() const { return nr; }
    int cols() const { return nc; }
    double operator()(int r, int c) const { return m(r0 + r, c0 + c); }
    template <typename Src> DynBlock& operator+=(const Src& s) { for (int r = 0; r < nr; ++r) for (int c = 0; c < nc; ++c) m(r0 + r, c0 + c) += s(r, c); return *this; }
    DynBlock& operator=(const MatrixXd& o);      // below
    template <typename Src> DynBlock& assign(const Src& s) { for (int r = 0; r < nr; ++r) for (int c = 0; c < nc; ++c) m(r0 + r, c0 + c) = s(r, c); return *this; }
    template <typename T, int R, int C> DynBlock& operator=(const Matrix<T, R, C>& v) { return assign(v); }
    template <typename M2> DynBlock& operator=(const DynBlock<M2>& o) { return assign(o); }
    DynBlock& operator=(const DynBlock& o) { return assign(o); }
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
    Matrix(const MatrixXd& o);                   // below (sizes must agree)
    explicit Matrix(const T* p) { for (int i = 0; i < R * C; ++i) d[i] = p[i]; }   // (vectors: Vector3D(meas.meas))
    int rows() const { return R; }
    int cols() const { return C; }
    DynBlock<Matrix> col(int c) { return DynBlock<Matrix>(*this, 0, c, R, 1); }
    Matrix inverse() const {                     // Gauss-Jordan with partial pivoting (Eigen: PartialPivLU; both exact solvers)
        static_assert(R == C, "square");
        Matrix a(*this), b = Identity();
        for (int k = 0; k < R; ++k) {
            int p = k;
            for (int r = k + 1; r < R; ++r) if (std::fabs(a(r, k)) > std::fabs(a(p, k))) p = r;
            if (p != k) for (int c = 0; c < C; ++c) { std::swap(a(k, c), a(p, c)); std::swap(b(k, c), b(p, c)); }
            const T inv = T(1) / a(k, k);
            for (int c = 0; c < C; ++c) { a(k, c) *= inv; b(k, c) *= inv; }
            for (int r = 0; r < R; ++r) {
                if (r == k) continue;
                const T f = a(r, k);
                if (f == T(0)) continue;
                for (int c = 0; c < C; ++c) { a(r, c) -= f * a(k, c); b(r, c) -= f * b(k, c); }
            }
        }
        return b;
    }
    static Matrix Zero() { return Matrix(); }
    static Matrix Zero(int, int) { return Matrix(); }
    DynBlock<Matrix> block(int r, int c, int nr, int nc) { return DynBlock<Matrix>(*this, r, c, nr, nc); }
    DynBlock<const Matrix> block(int r, int c, int nr, int nc) const { return DynBlock<const Matrix>(*this, r, c, nr, nc); }
    Matrix normalized() const { Matrix m(*this); const T n = norm(); for (int i = 0; i < R * C; ++i) m.d[i] /= n; return m; }
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
// a scalar of another arithmetic type (float, int): converted to the matrix's scalar first, as Eigen's scalar promotion does
template <typename T, int R, int C, typename S, typename std::enable_if<std::is_arithmetic<S>::value && !std::is_same<S, T>::value, int>::type = 0>
Matrix<T, R, C> operator*(const Matrix<T, R, C>& a, S s) { return a * T(s); }
template <typename T, int R, int C, typename S, typename std::enable_if<std::is_arithmetic<S>::value && !std::is_same<S, T>::value, int>::type = 0>
Matrix<T, R, C> operator*(S s, const Matrix<T, R, C>& a) { return a * T(s); }
template <typename T, int R, int C> Matrix<T, R, C> operator/(const Matrix<T, R, C>& a, T s) { Matrix<T, R, C> m; for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) m(r, c) = a(r, c) / s; return m; }
template <typename T, int R, int C> Matrix<T, R, C> operator/(const Matrix<T, R, C>& a, int s) { return a / T(s); }

template <typename T, int R, int C> std::ostream& operator<<(std::ostream& os, const Matrix<T, R, C>& m) {
    for (int r = 0; r < R; ++r) { for (int c = 0; c < C; ++c) os << (c ? " " : "") << m(r, c); os << "\n"; }
    return os;
}
typedef Matrix<double, 2, 1> Vector2d;
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<double, 2, 2> Matrix2d;
typedef Matrix<double, 3, 3> Matrix3d;
typedef Matrix<double, 4, 4> Matrix4d;

// Eigen::Map<Matrix3d, Options>(ptr): the second template argument of Eigen::Map is the ALIGNMENT option, not a storage
// order - Map<Matrix3d, RowMajor> (src/Map.cpp:951) therefore views the array in Matrix3d's own column-major order
enum { ColMajor = 0, RowMajor = 1 };
template <typename M, int Options = 0> class Map {
    double* p;
    enum { R = M::RowsAtCompileTime, C = M::ColsAtCompileTime };
public:
    explicit Map(double* p_) : p(p_) {}
    explicit Map(const double* p_) : p(const_cast<double*>(p_)) {}
    double& operator()(int r, int c) { return p[c * R + r]; }
    double operator()(int r, int c) const { return p[c * R + r]; }
    double& operator[](int i) { return p[i]; }
    double operator[](int i) const { return p[i]; }
    M eval() const { M m; for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) m(r, c) = (*this)(r, c); return m; }
    operator M() const { return eval(); }
    M inverse() const { return eval().inverse(); }
    Map& operator=(const M& m) { for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) (*this)(r, c) = m(r, c); return *this; }
    template <int N> BlockRef<Map, double, N, 1> head() { return BlockRef<Map, double, N, 1>(*this, 0, 0); }
};
template <typename T, int R, int K, typename M, int O> auto operator*(const Matrix<T, R, K>& a, const Map<M, O>& b) -> decltype(a * b.eval()) { return a * b.eval(); }
template <typename T, int K, int C, typename M, int O> auto operator*(const Map<M, O>& a, const Matrix<T, K, C>& b) -> decltype(a.eval() * b) { return a.eval() * b; }

// Eigen::MatrixXd: what src/sparsifier.cpp does with it - Zero / Identity, blocks, products, differences, ldlt().solve()
class LDLTXd;
class MatrixXd {
    int nr = 0, nc = 0;
    std::vector<double> d;
public:
    MatrixXd() {}
    MatrixXd(int r, int c) : nr(r), nc(c), d((size_t)r * c, 0.0) {}
    template <typename M> MatrixXd(const DynBlock<M>& b) : nr(b.rows()), nc(b.cols()), d((size_t)nr * nc) { for (int r = 0; r < nr; ++r) for (int c = 0; c < nc; ++c) (*this)(r, c) = b(r, c); }
    template <typename T, int R, int C> MatrixXd(const Matrix<T, R, C>& m) : nr(R), nc(C), d((size_t)R * C) { for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) (*this)(r, c) = m(r, c); }
    int rows() const { return nr; }
    int cols() const { return nc; }
    double& operator()(int r, int c) { return d[(size_t)r * nc + c]; }
    double operator()(int r, int c) const { return d[(size_t)r * nc + c]; }
    static MatrixXd Zero(int r, int c) { return MatrixXd(r, c); }
    static MatrixXd Identity(int r, int c) { MatrixXd m(r, c); for (int i = 0; i < std::min(r, c); ++i) m(i, i) = 1.0; return m; }
    DynBlock<MatrixXd> block(int r, int c, int br, int bc) { return DynBlock<MatrixXd>(*this, r, c, br, bc); }
    DynBlock<const MatrixXd> block(int r, int c, int br, int bc) const { return DynBlock<const MatrixXd>(*this, r, c, br, bc); }
    MatrixXd operator*(double s) const { MatrixXd m(*this); for (double& v : m.d) v *= s; return m; }
    MatrixXd operator*(const MatrixXd& o) const {
        MatrixXd m(nr, o.nc);
        for (int r = 0; r < nr; ++r) for (int c = 0; c < o.nc; ++c) { double s = 0; for (int k = 0; k < nc; ++k) s += (*this)(r, k) * o(k, c); m(r, c) = s; }
        return m;
    }
    MatrixXd operator-(const MatrixXd& o) const { MatrixXd m(*this); for (size_t i = 0; i < d.size(); ++i) m.d[i] -= o.d[i]; return m; }
    LDLTXd ldlt() const;
};
template <typename M> DynBlock<M>& DynBlock<M>::operator=(const MatrixXd& o) { return assign(o); }
template <typename T, int R, int C> Matrix<T, R, C>::Matrix(const MatrixXd& o) { for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) d[r * C + c] = o(r, c); }

// H22.ldlt().solve(H21): an exact solver of a symmetric positive definite system.  Eigen's LDLT pivots on the diagonal;
// this one is the plain Cholesky LDL' without pivoting - the two agree to round-off on the matrices the reference solves
// (block diagonal 3x3 sums of J' info J with a positive definite information)
class LDLTXd {
    MatrixXd L;
    std::vector<double> D;
public:
    explicit LDLTXd(const MatrixXd& A) : L(A.rows(), A.cols()), D(A.rows(), 0.0) {
        const int n = A.rows();
        for (int j = 0; j < n; ++j) {
            double dj = A(j, j);
            for (int k = 0; k < j; ++k) dj -= L(j, k) * L(j, k) * D[k];
            D[j] = dj;
            L(j, j) = 1.0;
            for (int i = j + 1; i < n; ++i) {
                double v = A(i, j);
                for (int k = 0; k < j; ++k) v -= L(i, k) * L(j, k) * D[k];
                L(i, j) = v / dj;
            }
        }
    }
    MatrixXd solve(const MatrixXd& B) const {
        const int n = L.rows(), m = B.cols();
        MatrixXd X(B);
        for (int c = 0; c < m; ++c) {
            for (int i = 0; i < n; ++i) { double v = X(i, c); for (int k = 0; k < i; ++k) v -= L(i, k) * X(k, c); X(i, c) = v; }
            for (int i = 0; i < n; ++i) X(i, c) /= D[i];
            for (int i = n - 1; i >= 0; --i) { double v = X(i, c); for (int k = i + 1; k < n; ++k) v -= L(k, i) * X(k, c); X(i, c) = v; }
        }
        return X;
    }
};
inline LDLTXd MatrixXd::ldlt() const { return LDLTXd(*this); }

// Eigen::JacobiSVD<MatrixXd>(A, ComputeFullU | ComputeFullV) of a square matrix: one-sided Jacobi (Hestenes) on the columns -
// A V = U S with orthogonal V, singular values sorted in decreasing order like Eigen's.  (Eigen's is two-sided; an SVD is
// unique up to the joint sign of a pair (u_i, v_i) and the basis inside a repeated singular value, neither of which the
// reference's use - sign of u_i . v_i, U S V' - can see.)
template <typename MatType> class JacobiSVD {
    MatrixXd U_, V_, S_;
public:
    template <typename M> JacobiSVD(const M& A, unsigned = 0) {
        const int n = A.rows();
        MatrixXd W(n, n), V = MatrixXd::Identity(n, n);
        for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) W(r, c) = A(r, c);
        for (int sweep = 0; sweep < 60; ++sweep) {
            double off = 0;
            for (int p = 0; p < n; ++p)
                for (int q = p + 1; q < n; ++q) {
                    double a = 0, b = 0, g = 0;
                    for (int r = 0; r < n; ++r) { a += W(r, p) * W(r, p); b += W(r, q) * W(r, q); g += W(r, p) * W(r, q); }
                    if (g == 0 || std::fabs(g) <= 1e-300) continue;
                    off = std::max(off, std::fabs(g) / std::sqrt(a * b));
                    const double zeta = (b - a) / (2 * g);
                    const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1 + zeta * zeta));
                    const double cs = 1 / std::sqrt(1 + t * t), sn = cs * t;
                    for (int r = 0; r < n; ++r) {
                        const double wp = W(r, p), wq = W(r, q);
                        W(r, p) = cs * wp - sn * wq; W(r, q) = sn * wp + cs * wq;
                        const double vp = V(r, p), vq = V(r, q);
                        V(r, p) = cs * vp - sn * vq; V(r, q) = sn * vp + cs * vq;
                    }
                }
            if (off < 1e-15) break;
        }
        std::vector<double> sv(n);
        std::vector<int> order(n);
        for (int c = 0; c < n; ++c) { double a = 0; for (int r = 0; r < n; ++r) a += W(r, c) * W(r, c); sv[c] = std::sqrt(a); order[c] = c; }
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return sv[x] > sv[y]; });
        U_ = MatrixXd(n, n); V_ = MatrixXd(n, n); S_ = MatrixXd(n, 1);
        for (int k = 0; k < n; ++k) {
            const int c = order[k];
            S_(k, 0) = sv[c];
            for (int r = 0; r < n; ++r) { U_(r, k) = sv[c] > 0 ? W(r, c) / sv[c] : (r == k ? 1.0 : 0.0); V_(r, k) = V(r, c); }
        }
    }
    const MatrixXd& singularValues() const { return S_; }
    const MatrixXd& matrixU() const { return U_; }
    const MatrixXd& matrixV() const { return V_; }
};

class Quaterniond;
class AngleAxisd {
public:
    double angle_;
    Vector3d axis_;
    AngleAxisd(double a, const Vector3d& ax) : angle_(a), axis_(ax) {}
    explicit AngleAxisd(const Quaterniond& q);      // below
    double angle() const { return angle_; }
    const Vector3d& axis() const { return axis_; }
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

// Eigen/src/Geometry/AngleAxis.h, operator=(QuaternionBase) of 3.3: angle = 2 atan2(|vec|, |w|), axis = vec / (+-|vec|)
// (3.2 takes 2 acos(w); the two agree to rounding for the unit quaternions with w >= 0 that SE3Quat hands out)
inline AngleAxisd::AngleAxisd(const Quaterniond& q) : angle_(0), axis_(1, 0, 0) {
    double n = q.vec().norm();
    if (n != 0) {
        angle_ = 2 * std::atan2(n, std::fabs(q.w()));
        if (q.w() < 0) n = -n;
        axis_ = q.vec() * (1.0 / n);
    }
}

// Eigen::Transform<double, 3, Isometry>: rotation matrix + translation
class Isometry3d {
    Matrix3d R_;
    Vector3d t_;
public:
    Isometry3d() { R_.setIdentity(); }
    static Isometry3d Identity() { return Isometry3d(); }
    explicit Isometry3d(const Quaterniond& q) : R_(q.toRotationMatrix()) {}
    Isometry3d& operator=(const Quaterniond& q) { R_ = q.toRotationMatrix(); t_.setZero(); return *this; }
    Vector3d& translation() { return t_; }
    const Vector3d& translation() const { return t_; }
    Matrix3d& linear() { return R_; }
    const Matrix3d& linear() const { return R_; }
    const Matrix3d& rotation() const { return R_; }
    Matrix4d matrix() const {
        Matrix4d m;
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) m(r, c) = R_(r, c); m(r, 3) = t_[r]; }
        m(3, 3) = 1;
        return m;
    }
    Isometry3d operator*(const Isometry3d& o) const { Isometry3d r; r.R_ = R_ * o.R_; r.t_ = R_ * o.t_ + t_; return r; }
    Isometry3d inverse() const { Isometry3d r; r.R_ = R_.transpose(); r.t_ = -(r.R_ * t_); return r; }
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

inline Matrix3D skew(const Vector3D& v) {   // g2o/types/sba/types_six_dof_expmap.h (se3_ops)
    Matrix3D m;
    m(0, 1) = -v(2); m(0, 2) = v(1); m(1, 2) = -v(0);
    m(1, 0) = v(2); m(2, 0) = -v(1); m(2, 1) = v(0);
    return m;
}
inline Vector3D deltaR(const Matrix3D& R) {   // se3_ops.hpp
    return Vector3D(R(2, 1) - R(1, 2), R(0, 2) - R(2, 0), R(1, 0) - R(0, 1));
}

typedef Eigen::Matrix<double, 6, 6> Matrix6d;
typedef Eigen::Matrix<double, 6, 1> Vector6d;
typedef Eigen::Isometry3d Isometry3D;

class SE3Quat {   // g2o/types/slam3d/se3quat.h
    Eigen::Quaterniond _r;
    Vector3D _t;
public:
    SE3Quat() : _t(0, 0, 0) {}
    SE3Quat(const Matrix3D& R, const Vector3D& t) : _r(Eigen::Quaterniond(R)), _t(t) { normalizeRotation(); }
    SE3Quat(const Eigen::Quaterniond& q, const Vector3D& t) : _r(q), _t(t) { normalizeRotation(); }
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
    Vector3D operator*(const Vector3D& v) const { return _t + _r * v; }
    Vector6d toMinimalVector() const {            // (translation, q_x, q_y, q_z)
        Vector6d v;
        v[0] = _t(0); v[1] = _t(1); v[2] = _t(2);
        v[3] = _r.x(); v[4] = _r.y(); v[5] = _r.z();
        return v;
    }
    void fromMinimalVector(const Vector6d& v) {
        const double w = 1. - v[3] * v[3] - v[4] * v[4] - v[5] * v[5];
        if (w > 0) _r = Eigen::Quaterniond(std::sqrt(w), v[3], v[4], v[5]);
        else _r = Eigen::Quaterniond(0, -v[3], -v[4], -v[5]);
        _t = Vector3D(v[0], v[1], v[2]);
    }
    Vector6d log() const {   // (rotation, translation)
        Vector6d res;
        const Matrix3D _R = _r.toRotationMatrix();
        const double d = 0.5 * (_R(0, 0) + _R(1, 1) + _R(2, 2) - 1);
        Vector3D omega;
        const Vector3D dR = deltaR(_R);
        Matrix3D V_inv;
        if (d > 0.99999) {
            omega = 0.5 * dR;
            const Matrix3D Omega = skew(omega);
            V_inv = Matrix3D::Identity() - 0.5 * Omega + (1. / 12.) * (Omega * Omega);
        } else {
            const double theta = std::acos(d);
            omega = theta / (2 * std::sqrt(1 - d * d)) * dR;
            const Matrix3D Omega = skew(omega);
            V_inv = Matrix3D::Identity() - 0.5 * Omega + ((1 - theta / (2 * std::tan(theta / 2))) / (theta * theta)) * (Omega * Omega);
        }
        const Vector3D upsilon = V_inv * _t;
        for (int i = 0; i < 3; i++) { res[i] = omega[i]; res[i + 3] = upsilon[i]; }
        return res;
    }
    Matrix6d adj() const {   // (rotation, translation) order: [R 0; skew(t) R  R]
        const Matrix3D R = _r.toRotationMatrix();
        Matrix6d res;
        res.block(0, 0, 3, 3) = R;
        res.block(3, 3, 3, 3) = R;
        res.block(3, 0, 3, 3) = skew(_t) * R;
        return res;
    }
    Eigen::Matrix4d to_homogeneous_matrix() const {
        Eigen::Matrix4d m;
        const Matrix3D R = _r.toRotationMatrix();
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) m(r, c) = R(r, c); m(r, 3) = _t[r]; }
        m(3, 3) = 1;
        return m;
    }
    operator Isometry3D() const {
        Isometry3D result(_r);
        result.translation() = _t;
        return result;
    }
};

namespace internal {
inline Vector3D toEuler(const Matrix3D& R) {   // g2o/types/slam3d/isometry3d_mappings.cpp
    Eigen::Quaterniond q(R);
    const double q0 = q.w(), q1 = q.x(), q2 = q.y(), q3 = q.z();
    const double roll = std::atan2(2 * (q0 * q1 + q2 * q3), 1 - 2 * (q1 * q1 + q2 * q2));
    const double pitch = std::asin(2 * (q0 * q2 - q3 * q1));
    const double yaw = std::atan2(2 * (q0 * q3 + q1 * q2), 1 - 2 * (q2 * q2 + q3 * q3));
    return Vector3D(roll, pitch, yaw);
}
inline SE3Quat toSE3Quat(const Isometry3D& t) { return SE3Quat(t.linear(), t.translation()); }
inline Isometry3D fromSE3Quat(const SE3Quat& t) {
    Isometry3D result(t.rotation());
    result.translation() = t.translation();
    return result;
}
}  // namespace internal

// ---- the graph: only what records what the reference's add* functions (src/optimizer.cpp) put into it.  No solver.
class Parameter {
    int _id = -1;
public:
    virtual ~Parameter() {}
    void setId(int id) { _id = id; }
    int id() const { return _id; }
};

class CameraParameters : public Parameter {   // g2o/types/sba/types_six_dof_expmap.{h,cpp}
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

class ParameterSE3Offset : public Parameter {
    Isometry3D _offset;
public:
    void setOffset(const Isometry3D& o) { _offset = o; }
    const Isometry3D& offset() const { return _offset; }
};

class RobustKernel {
protected:
    double _delta = 1.;
public:
    virtual ~RobustKernel() {}
    virtual void setDelta(double d) { _delta = d; }
    double delta() const { return _delta; }
    virtual double rho(double e2) const = 0;
};
class RobustKernelHuber : public RobustKernel {   // g2o/core/robust_kernel_impl.cpp: rho[0]
public:
    double rho(double e2) const override {
        const double dsqr = _delta * _delta;
        if (e2 <= dsqr) return e2;
        return 2 * std::sqrt(e2) * _delta - dsqr;
    }
};

namespace HyperGraph {
class Vertex {
protected:
    int _id = -1;
public:
    virtual ~Vertex() {}
    int id() const { return _id; }
    void setId(int id) { _id = id; }
};
class Edge {
protected:
    std::vector<Vertex*> _vertices;
public:
    explicit Edge(int n) : _vertices(n, nullptr) {}
    virtual ~Edge() {}
    std::vector<Vertex*>& vertices() { return _vertices; }
    void setVertex(size_t i, Vertex* v) { _vertices[i] = v; }
};
}  // namespace HyperGraph

namespace OptimizableGraph {
class Vertex : public HyperGraph::Vertex {
    bool _fixed = false, _marginalized = false;
public:
    void setFixed(bool f) { _fixed = f; }
    bool fixed() const { return _fixed; }
    void setMarginalized(bool m) { _marginalized = m; }
    bool marginalized() const { return _marginalized; }
};
class Edge : public HyperGraph::Edge {
    RobustKernel* _rk = nullptr;
    int _level = 0;
    std::vector<int> _parameterIds;
    const std::vector<Parameter*>* _graphParameters = nullptr;   // set by SparseOptimizer::addEdge (g2o: resolveParameters)
public:
    void setGraphParameters(const std::vector<Parameter*>* p) { _graphParameters = p; }
    const Parameter* parameterOf(int argNum) const {
        if (!_graphParameters) return nullptr;
        for (const Parameter* p : *_graphParameters) if (p->id() == parameterId(argNum)) return p;
        return nullptr;
    }
    explicit Edge(int n) : HyperGraph::Edge(n) {}
    ~Edge() override { delete _rk; }
    void setRobustKernel(RobustKernel* rk) { delete _rk; _rk = rk; }
    RobustKernel* robustKernel() const { return _rk; }
    void setLevel(int l) { _level = l; }
    int level() const { return _level; }
    bool setParameterId(int argNum, int paramId) {
        if ((int)_parameterIds.size() <= argNum) _parameterIds.resize(argNum + 1, -1);
        _parameterIds[argNum] = paramId;
        return true;
    }
    int parameterId(int argNum) const { return argNum < (int)_parameterIds.size() ? _parameterIds[argNum] : -1; }
    virtual void computeError() = 0;
    virtual void linearizeOplus() = 0;
    virtual double chi2() const = 0;
};
}  // namespace OptimizableGraph

template <int D, typename T> class BaseVertex : public OptimizableGraph::Vertex {
protected:
    T _estimate;
public:
    enum { Dimension = D };
    const T& estimate() const { return _estimate; }
    void setEstimate(const T& e) { _estimate = e; }
};
class VertexSE2 : public BaseVertex<3, SE2> {};
class VertexSBAPointXYZ : public BaseVertex<3, Vector3D> {};
class VertexSE3Expmap : public BaseVertex<6, SE3Quat> {};
class VertexSE3 : public BaseVertex<6, Isometry3D> {};
class VertexPointXYZ : public BaseVertex<3, Vector3D> {};

template <int D, typename E> class BaseEdge : public OptimizableGraph::Edge {
protected:
    E _measurement;
    Eigen::Matrix<double, D, D> _information;
    Eigen::Matrix<double, D, 1> _error;
public:
    explicit BaseEdge(int n) : OptimizableGraph::Edge(n) {}
    void setMeasurement(const E& m) { _measurement = m; }
    const E& measurement() const { return _measurement; }
    const Eigen::Matrix<double, D, D>& information() const { return _information; }
    Eigen::Matrix<double, D, D>& information() { return _information; }
    void setInformation(const Eigen::Matrix<double, D, D>& i) { _information = i; }
    const Eigen::Matrix<double, D, 1>& error() const { return _error; }
    double chi2() const override { return _error.dot(_information * _error); }
    virtual bool read(std::istream& is) = 0;
    virtual bool write(std::ostream& os) const = 0;
};

template <int D, typename E, typename VertexXi> class BaseUnaryEdge : public BaseEdge<D, E> {
protected:
    Eigen::Matrix<double, D, VertexXi::Dimension> _jacobianOplusXi;
public:
    BaseUnaryEdge() : BaseEdge<D, E>(1) {}
    const Eigen::Matrix<double, D, VertexXi::Dimension>& jacobianOplusXi() const { return _jacobianOplusXi; }
};

template <int D, typename E, typename VertexXi, typename VertexXj> class BaseBinaryEdge : public BaseEdge<D, E> {
protected:
    Eigen::Matrix<double, D, VertexXi::Dimension> _jacobianOplusXi;
    Eigen::Matrix<double, D, VertexXj::Dimension> _jacobianOplusXj;
public:
    BaseBinaryEdge() : BaseEdge<D, E>(2) {}
    const Eigen::Matrix<double, D, VertexXi::Dimension>& jacobianOplusXi() const { return _jacobianOplusXi; }
    const Eigen::Matrix<double, D, VertexXj::Dimension>& jacobianOplusXj() const { return _jacobianOplusXj; }
};

// g2o's own edge types that src/optimizer.cpp constructs and fills in.  The two of the SE3-expmap local graph are evaluated
// too - EdgeProjectXYZ2UV::computeError (types_six_dof_expmap.h: obs - cam_map(T.map(X)) with the camera parameter of id
// parameterId(0)) and EdgeSE3Expmap::computeError (log(T_1^-1 * measurement * T_0)), from the published g2o sources - so
// that the cost of a graph Map::loadLocalGraph built can be compared as a whole; the pose-graph types are recorded only.
#define SE2_SHIM_RECORD_ONLY                                               \
    void computeError() override {}                                        \
    void linearizeOplus() override {}                                      \
    bool read(std::istream&) override { return false; }                    \
    bool write(std::ostream&) const override { return false; }
class EdgeSE3Expmap : public BaseBinaryEdge<6, SE3Quat, VertexSE3Expmap, VertexSE3Expmap> {
public:
    void computeError() override {
        const VertexSE3Expmap* v1 = static_cast<const VertexSE3Expmap*>(_vertices[0]);
        const VertexSE3Expmap* v2 = static_cast<const VertexSE3Expmap*>(_vertices[1]);
        const SE3Quat C(_measurement);
        const SE3Quat error_ = v2->estimate().inverse() * C * v1->estimate();
        _error = error_.log();
    }
    void linearizeOplus() override {}
    bool read(std::istream&) override { return false; }
    bool write(std::ostream&) const override { return false; }
};
class EdgeProjectXYZ2UV : public BaseBinaryEdge<2, Vector2D, VertexSBAPointXYZ, VertexSE3Expmap> {
public:
    void computeError() override {
        const VertexSE3Expmap* v1 = static_cast<const VertexSE3Expmap*>(_vertices[1]);
        const VertexSBAPointXYZ* v2 = static_cast<const VertexSBAPointXYZ*>(_vertices[0]);
        const CameraParameters* cam = dynamic_cast<const CameraParameters*>(parameterOf(0));
        const Vector2D obs(_measurement);
        _error = obs - cam->cam_map(v1->estimate().map(v2->estimate()));
    }
    void linearizeOplus() override {}
    bool read(std::istream&) override { return false; }
    bool write(std::ostream&) const override { return false; }
};
namespace internal {
inline Vector6d toVectorMQT(const Isometry3D& t) {   // g2o/types/slam3d/isometry3d_mappings.cpp: (translation, q_xyz of the unit quaternion with w >= 0)
    Eigen::Quaterniond q(t.linear());
    q.normalize();
    Vector6d v;
    const double s = q.w() < 0 ? -1.0 : 1.0;
    v[3] = s * q.x(); v[4] = s * q.y(); v[5] = s * q.z();
    v[0] = t.translation()[0]; v[1] = t.translation()[1]; v[2] = t.translation()[2];
    return v;
}
}  // namespace internal
// EdgeSE3 / EdgeSE3Prior (g2o/types/slam3d): error = toVectorMQT(Z^-1 * X_i^-1 * X_j) resp. toVectorMQT(Z^-1 * X * offset) - evaluated
// so that the cost of GlobalMapper::GlobalBA's graph can be compared as a whole (from the published g2o sources)
class EdgeSE3 : public BaseBinaryEdge<6, Isometry3D, VertexSE3, VertexSE3> {
public:
    void computeError() override {
        const VertexSE3* from = static_cast<const VertexSE3*>(_vertices[0]);
        const VertexSE3* to = static_cast<const VertexSE3*>(_vertices[1]);
        _error = internal::toVectorMQT(_measurement.inverse() * from->estimate().inverse() * to->estimate());
    }
    void linearizeOplus() override {}
    bool read(std::istream&) override { return false; }
    bool write(std::ostream&) const override { return false; }
};
class EdgeSE3Prior : public BaseUnaryEdge<6, Isometry3D, VertexSE3> {
public:
    void computeError() override {
        const VertexSE3* v = static_cast<const VertexSE3*>(_vertices[0]);
        const ParameterSE3Offset* off = dynamic_cast<const ParameterSE3Offset*>(parameterOf(0));
        const Isometry3D n2w = off ? v->estimate() * off->offset() : v->estimate();
        _error = internal::toVectorMQT(_measurement.inverse() * n2w);
    }
    void linearizeOplus() override {}
    bool read(std::istream&) override { return false; }
    bool write(std::ostream&) const override { return false; }
};
class EdgeSE3PointXYZ : public BaseBinaryEdge<3, Vector3D, VertexSE3, VertexPointXYZ> { public: SE2_SHIM_RECORD_ONLY };
#undef SE2_SHIM_RECORD_ONLY

template <typename M> class LinearSolverCholmod {};
template <typename M> class LinearSolverEigen {};
class BlockSolverX {
public:
    typedef int PoseMatrixType;
    template <typename LS> explicit BlockSolverX(LS* ls) { delete ls; }
};
class OptimizationAlgorithmLevenberg {
public:
    explicit OptimizationAlgorithmLevenberg(BlockSolverX* bs) { delete bs; }
};

class SparseOptimizer {
    std::map<int, OptimizableGraph::Vertex*> _vertices;
    std::vector<OptimizableGraph::Edge*> _edges;
    std::set<OptimizableGraph::Edge*> _edgeSet;
    std::vector<Parameter*> _parameters;
    OptimizationAlgorithmLevenberg* _algorithm = nullptr;
    bool _verbose = false;
    bool* _stop = nullptr;
    int _level = 0;
public:
    SparseOptimizer() {}
    SparseOptimizer(const SparseOptimizer&) = delete;
    ~SparseOptimizer() {
        for (auto& kv : _vertices) delete kv.second;
        for (auto* e : _edges) delete e;
        for (auto* p : _parameters) delete p;
        delete _algorithm;
    }
    bool addVertex(OptimizableGraph::Vertex* v) { return _vertices.emplace(v->id(), v).second; }
    bool addEdge(OptimizableGraph::Edge* e) {       // HyperGraph::addEdge: an edge that is already in the graph is refused
        if (!_edgeSet.insert(e).second) return false;             // (Map.cpp:551-553 adds every projection edge a second time)
        e->setGraphParameters(&_parameters);
        _edges.push_back(e);
        return true;
    }
    bool addParameter(Parameter* p) { _parameters.push_back(p); return true; }
    OptimizableGraph::Vertex* vertex(int id) { auto it = _vertices.find(id); return it == _vertices.end() ? nullptr : it->second; }
    void setAlgorithm(OptimizationAlgorithmLevenberg* a) { delete _algorithm; _algorithm = a; }
    void setVerbose(bool v) { _verbose = v; }
    // Nothing is optimised in oracle/_ref: initializeOptimization / optimize hand the graph, as the reference built it, to
    // whoever registered a hook (the drivers: cost at the start, the recorded vertices and edges), and return.
    static std::function<void(SparseOptimizer&, int)>& optimizeHook() { static std::function<void(SparseOptimizer&, int)> h; return h; }
    bool initializeOptimization(int level = 0) { _level = level; return true; }
    int optimize(int iterations) { if (optimizeHook()) optimizeHook()(*this, iterations); return 0; }
    void setForceStopFlag(bool* f) { _stop = f; }
    bool* forceStopFlag() const { return _stop; }
    int level() const { return _level; }
    bool verbose() const { return _verbose; }
    bool hasAlgorithm() const { return _algorithm != nullptr; }
    const std::vector<OptimizableGraph::Edge*>& edges() const { return _edges; }
    const std::map<int, OptimizableGraph::Vertex*>& vertices() const { return _vertices; }
    const std::vector<Parameter*>& parameters() const { return _parameters; }
};

}  // namespace g2o
