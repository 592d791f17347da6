// ORACLE - TEST INFRASTRUCTURE ONLY (oracle/_ref).  See cv_shim.hpp.
//
// The OpenCV functions /root/reference's hot-path sources call, written from the published OpenCV 3.2 algorithms
// (modules/features2d/src/fast.cpp, fast_score.cpp, keypoint.cpp; modules/imgproc/src/imgwarp.cpp, smooth.cpp, filter.cpp;
// modules/core/src/copy.cpp, mathfuncs_core.cpp, lapack.cpp) over cv::Mat views - NOT over oracle/orb_ref.cpp's pyramid
// levels, and without looking at its formulation: FAST here is the rolling three-row scan with cornerScore<16>, there a
// score plane with a strict-maximum test; resize / blur / border here work on any view with a parent matrix.  Where the two
// agree bit for bit (tests/test_ref_compiled.py) two independent readings of the same library text agree; neither is
// the library.
#include "cv_shim.hpp"

#include <cfloat>
#include <cstdio>
#include <map>
#include <cstdlib>

namespace cv {

// ------------------------------------------------------------------------------------------------ Mat
Mat::Mat(const Mat& m, const Rect& roi)
    : flags(m.flags), rows(roi.height), cols(roi.width), step(m.step), data(m.data + (size_t)roi.y * m.step + (size_t)roi.x * m.elemSize()),
      datastart(m.datastart), dataend(m.dataend), owner(m.owner) {
    if (roi.x < 0 || roi.y < 0 || roi.width < 0 || roi.height < 0 || roi.x + roi.width > m.cols || roi.y + roi.height > m.rows)
        throw std::runtime_error("cv shim: ROI outside the matrix");
}

void Mat::locateROI(Size& whole, Point& ofs) const {   // core/src/matrix.cpp
    const size_t esz = elemSize();
    const ptrdiff_t delta1 = data - datastart, delta2 = dataend - datastart;
    if (delta1 == 0) ofs.x = ofs.y = 0;
    else {
        ofs.y = (int)(delta1 / (ptrdiff_t)step);
        ofs.x = (int)((delta1 - (ptrdiff_t)step * ofs.y) / (ptrdiff_t)esz);
    }
    const size_t minstep = (size_t)(ofs.x + cols) * esz;
    whole.height = (int)((delta2 - (ptrdiff_t)minstep) / (ptrdiff_t)step + 1);
    whole.height = std::max(whole.height, ofs.y + rows);
    whole.width = (int)((delta2 - (ptrdiff_t)step * (whole.height - 1)) / (ptrdiff_t)esz);
    whole.width = std::max(whole.width, ofs.x + cols);
}

void Mat::copyTo(Mat& dst) const {
    if (empty()) { dst.release(); return; }
    dst.create(rows, cols, flags);
    if (dst.data == data && dst.step == step) return;
    const size_t rb = (size_t)cols * elemSize();
    for (int r = 0; r < rows; ++r) std::memmove(dst.ptr(r), ptr(r), rb);
}

double Mat::getd(int r, int c) const {
    switch (depth()) {
        case CV_8U: return at<uchar>(r, c);
        case CV_32S: return at<int>(r, c);
        case CV_32F: return at<float>(r, c);
        case CV_64F: return at<double>(r, c);
    }
    throw std::runtime_error("cv shim: element type not supported");
}
void Mat::setd(int r, int c, double v) {
    switch (depth()) {
        case CV_8U: at<uchar>(r, c) = saturate_cast<uchar>(v); return;
        case CV_32S: at<int>(r, c) = cvRound(v); return;
        case CV_32F: at<float>(r, c) = (float)v; return;
        case CV_64F: at<double>(r, c) = v; return;
    }
    throw std::runtime_error("cv shim: element type not supported");
}
void Mat::convertTo(Mat& dst, int type) const {
    if (channels() != 1) throw std::runtime_error("cv shim: convertTo of a multi-channel matrix");
    Mat out(rows, cols, CV_MAKETYPE(CV_MAT_DEPTH(type), 1));
    for (int r = 0; r < rows; ++r) for (int c = 0; c < cols; ++c) out.setd(r, c, getd(r, c));
    dst = out;
}
Mat& Mat::setTo(const Scalar& s) {
    if (channels() != 1) throw std::runtime_error("cv shim: setTo on a multi-channel matrix");
    for (int r = 0; r < rows; ++r) for (int c = 0; c < cols; ++c) setd(r, c, s.val[0]);
    return *this;
}
static void need_float_fwd(const Mat& a) {
    if (a.channels() != 1 || (a.depth() != CV_32F && a.depth() != CV_64F)) throw std::runtime_error("cv shim: matrix arithmetic needs 32F / 64F");
}
MatExpr Mat::t() const {
    Mat out(cols, rows, flags);
    const size_t esz = elemSize();
    for (int r = 0; r < rows; ++r) for (int c = 0; c < cols; ++c) std::memcpy(out.ptr(c) + r * esz, ptr(r) + c * esz, esz);
    return MatExpr(out);
}
MatExpr Mat::inv() const {
    need_float_fwd(*this);
    if (rows != cols) throw std::runtime_error("cv shim: inv() of a non-square matrix");
    const int n = rows;
    std::vector<double> a((size_t)n * n), b((size_t)n * n, 0.0);
    for (int r = 0; r < n; ++r) { for (int c = 0; c < n; ++c) a[(size_t)r * n + c] = getd(r, c); b[(size_t)r * n + r] = 1.0; }
    for (int k = 0; k < n; ++k) {
        int p = k;
        for (int r = k + 1; r < n; ++r) if (std::fabs(a[(size_t)r * n + k]) > std::fabs(a[(size_t)p * n + k])) p = r;
        if (p != k) for (int c = 0; c < n; ++c) { std::swap(a[(size_t)k * n + c], a[(size_t)p * n + c]); std::swap(b[(size_t)k * n + c], b[(size_t)p * n + c]); }
        const double inv = 1.0 / a[(size_t)k * n + k];
        for (int c = 0; c < n; ++c) { a[(size_t)k * n + c] *= inv; b[(size_t)k * n + c] *= inv; }
        for (int r = 0; r < n; ++r) {
            if (r == k) continue;
            const double f = a[(size_t)r * n + k];
            if (f == 0) continue;
            for (int c = 0; c < n; ++c) { a[(size_t)r * n + c] -= f * a[(size_t)k * n + c]; b[(size_t)r * n + c] -= f * b[(size_t)k * n + c]; }
        }
    }
    Mat out(n, n, flags);
    for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) out.setd(r, c, b[(size_t)r * n + c]);
    return MatExpr(out);
}
double norm(const Mat& m) {
    double s = 0;
    for (int r = 0; r < m.rows; ++r) for (int c = 0; c < m.cols; ++c) { const double v = m.getd(r, c); s += v * v; }
    return std::sqrt(s);
}
MatExpr Mat::zeros(int r, int c, int type) {
    Mat out(r, c, type);
    for (int y = 0; y < r; ++y) std::memset(out.ptr(y), 0, (size_t)c * out.elemSize());
    return MatExpr(out);
}
MatExpr Mat::eye(int r, int c, int type) {
    Mat out = zeros(r, c, type);
    for (int i = 0; i < std::min(r, c); ++i) out.setd(i, i, 1.0);
    return MatExpr(out);
}

static void need_float(const Mat& a) {
    if (a.channels() != 1 || (a.depth() != CV_32F && a.depth() != CV_64F)) throw std::runtime_error("cv shim: matrix arithmetic needs 32F / 64F");
}
// gemm of core/src/matmul.cpp for small float matrices: every output element accumulates in double ("float x float -> double sums")
MatExpr operator*(const MatExpr& a, const MatExpr& b) {
    need_float(a.m); need_float(b.m);
    if (a.m.cols != b.m.rows || a.m.type() != b.m.type()) throw std::runtime_error("cv shim: product of mismatching matrices");
    Mat out(a.m.rows, b.m.cols, a.m.type());
    for (int r = 0; r < out.rows; ++r)
        for (int c = 0; c < out.cols; ++c) {
            double s = 0;
            for (int k = 0; k < a.m.cols; ++k) s += a.m.getd(r, k) * b.m.getd(k, c);
            out.setd(r, c, s);
        }
    return MatExpr(out);
}
template <typename F> static MatExpr elementwise(const Mat& a, F f) {
    need_float(a);
    Mat out(a.rows, a.cols, a.type());
    for (int r = 0; r < a.rows; ++r) for (int c = 0; c < a.cols; ++c) out.setd(r, c, f(r, c));
    return MatExpr(out);
}
MatExpr operator*(double s, const MatExpr& a) { return elementwise(a.m, [&](int r, int c) { return a.m.getd(r, c) * s; }); }
MatExpr operator*(const MatExpr& a, double s) { return s * a; }
MatExpr operator/(const MatExpr& a, double s) { return elementwise(a.m, [&](int r, int c) { return a.m.getd(r, c) / s; }); }
MatExpr operator-(const MatExpr& a) { return elementwise(a.m, [&](int r, int c) { return -a.m.getd(r, c); }); }
static void same_shape(const Mat& a, const Mat& b) {
    if (a.rows != b.rows || a.cols != b.cols || a.type() != b.type()) throw std::runtime_error("cv shim: sum of mismatching matrices");
}
MatExpr operator+(const MatExpr& a, const MatExpr& b) { same_shape(a.m, b.m); return elementwise(a.m, [&](int r, int c) { return a.m.getd(r, c) + b.m.getd(r, c); }); }
MatExpr operator-(const MatExpr& a, const MatExpr& b) { same_shape(a.m, b.m); return elementwise(a.m, [&](int r, int c) { return a.m.getd(r, c) - b.m.getd(r, c); }); }
std::ostream& operator<<(std::ostream& os, const Mat& m) {
    os << "[";
    for (int r = 0; r < m.rows; ++r) {
        for (int c = 0; c < m.cols; ++c) os << (c ? ", " : "") << (m.channels() == 1 ? m.getd(r, c) : 0.0);
        os << (r + 1 < m.rows ? ";\n " : "");
    }
    return os << "]";
}

// ------------------------------------------------------------------------------------------------ core
// mathfuncs_core.cpp: degree-7 odd polynomial, result in degrees
float fastAtan2(float y, float x) {
    static const float p1 = 0.9997878412794807f * (float)(180 / CV_PI), p3 = -0.3258083974640975f * (float)(180 / CV_PI),
                       p5 = 0.1555786518463281f * (float)(180 / CV_PI), p7 = -0.04432655554792128f * (float)(180 / CV_PI);
    const float ax = std::abs(x), ay = std::abs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

static int border_interpolate(int p, int len, int type) {   // core/src/copy.cpp borderInterpolate
    if ((unsigned)p < (unsigned)len) return p;
    if (type == BORDER_REPLICATE) return p < 0 ? 0 : len - 1;
    if (type == BORDER_REFLECT || type == BORDER_REFLECT_101) {
        const int delta = type == BORDER_REFLECT_101;
        if (len == 1) return 0;
        do {
            if (p < 0) p = -p - 1 + delta;
            else p = len - 1 - (p - len) - delta;
        } while ((unsigned)p >= (unsigned)len);
        return p;
    }
    if (type == BORDER_WRAP) { if (p < 0) p -= ((p - len + 1) / len) * len; if (p >= len) p %= len; return p; }
    if (type == BORDER_CONSTANT) return -1;
    throw std::runtime_error("cv shim: border type");
}

void copyMakeBorder(InputArray _src, OutputArray _dst, int top, int bottom, int left, int right, int borderType, const Scalar& value) {
    Mat src = _src.getMat();
    if (src.type() != CV_8UC1) throw std::runtime_error("cv shim: copyMakeBorder is 8UC1 only");
    if (src.isSubmatrix() && (borderType & BORDER_ISOLATED) == 0) {   // the pixels around a view count as image
        Size whole; Point ofs;
        src.locateROI(whole, ofs);
        const int dtop = std::min(ofs.y, top), dbottom = std::min(whole.height - src.rows - ofs.y, bottom);
        const int dleft = std::min(ofs.x, left), dright = std::min(whole.width - src.cols - ofs.x, right);
        src.data -= (size_t)dtop * src.step + dleft;
        src.rows += dtop + dbottom;
        src.cols += dleft + dright;
        top -= dtop; left -= dleft; bottom -= dbottom; right -= dright;
    }
    _dst.create(src.rows + top + bottom, src.cols + left + right, src.type());
    Mat dst = _dst.getMat();
    if (top == 0 && left == 0 && bottom == 0 && right == 0) {
        if (src.data != dst.data || src.step != dst.step) src.copyTo(dst);
        return;
    }
    borderType &= ~BORDER_ISOLATED;
    const int W = src.cols, H = src.rows;
    for (int y = 0; y < H; ++y) {   // interior (it may be the very same memory), then the row's left / right frame
        uchar* d = dst.ptr(y + top);
        const uchar* s = src.ptr(y);
        if (d + left != s) std::memmove(d + left, s, W);
        if (borderType == BORDER_CONSTANT) {
            for (int x = 0; x < left; ++x) d[x] = saturate_cast<uchar>(value.val[0]);
            for (int x = 0; x < right; ++x) d[left + W + x] = saturate_cast<uchar>(value.val[0]);
        } else {
            for (int x = 0; x < left; ++x) d[x] = d[left + border_interpolate(x - left, W, borderType)];
            for (int x = 0; x < right; ++x) d[left + W + x] = d[left + border_interpolate(W + x, W, borderType)];
        }
    }
    const size_t rb = (size_t)dst.cols;
    for (int y = 0; y < top; ++y) {
        if (borderType == BORDER_CONSTANT) std::memset(dst.ptr(y), saturate_cast<uchar>(value.val[0]), rb);
        else std::memcpy(dst.ptr(y), dst.ptr(top + border_interpolate(y - top, H, borderType)), rb);
    }
    for (int y = 0; y < bottom; ++y) {
        if (borderType == BORDER_CONSTANT) std::memset(dst.ptr(top + H + y), saturate_cast<uchar>(value.val[0]), rb);
        else std::memcpy(dst.ptr(top + H + y), dst.ptr(top + border_interpolate(H + y, H, borderType)), rb);
    }
}

// ------------------------------------------------------------------------------------------------ imgproc
// imgwarp.cpp: resize, INTER_LINEAR, 8UC1 - 11-bit fixed-point coefficients, HResizeLinear into int rows, VResizeLinear
// dst = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;  INTER_NEAREST for the mask pyramid
static inline short sat_short(float v) { const int i = cvRound(v); return (short)std::min(std::max(i, -32768), 32767); }
void resize(InputArray _src, OutputArray _dst, Size dsize, double inv_scale_x, double inv_scale_y, int interpolation) {
    ++shim_call_counts()[1];
    Mat src = _src.getMat();
    if (src.type() != CV_8UC1) throw std::runtime_error("cv shim: resize is 8UC1 only");
    const Size ssize = src.size();
    if (dsize.width == 0 || dsize.height == 0) {
        dsize = Size(cvRound(ssize.width * inv_scale_x), cvRound(ssize.height * inv_scale_y));
    } else {
        inv_scale_x = (double)dsize.width / ssize.width;
        inv_scale_y = (double)dsize.height / ssize.height;
    }
    _dst.create(dsize, src.type());
    Mat dst = _dst.getMat();
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    if (interpolation == INTER_NEAREST) {
        for (int y = 0; y < dsize.height; ++y) {
            const int sy = std::min(cvFloor(y * scale_y), ssize.height - 1);
            for (int x = 0; x < dsize.width; ++x) dst.at<uchar>(y, x) = src.at<uchar>(sy, std::min(cvFloor(x * scale_x), ssize.width - 1));
        }
        return;
    }
    if (interpolation != INTER_LINEAR) throw std::runtime_error("cv shim: interpolation mode");
    const int COEF = 2048, ksize = 2, ksize2 = 1;
    std::vector<int> xofs(dsize.width), yofs(dsize.height);
    std::vector<short> ialpha(dsize.width * ksize), ibeta(dsize.height * ksize);
    int xmin = 0, xmax = dsize.width;
    for (int dx = 0; dx < dsize.width; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cvFloor(fx);
        fx -= sx;
        if (sx < ksize2 - 1) { xmin = dx + 1; if (sx < 0) { fx = 0; sx = 0; } }
        if (sx + ksize2 >= ssize.width) { xmax = std::min(xmax, dx); if (sx >= ssize.width - 1) { fx = 0; sx = ssize.width - 1; } }
        xofs[dx] = sx;
        ialpha[dx * 2] = sat_short((1.f - fx) * COEF);
        ialpha[dx * 2 + 1] = sat_short(fx * COEF);
    }
    (void)xmin;
    for (int dy = 0; dy < dsize.height; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        const int sy = cvFloor(fy);
        fy -= sy;
        yofs[dy] = sy;
        ibeta[dy * 2] = sat_short((1.f - fy) * COEF);
        ibeta[dy * 2 + 1] = sat_short(fy * COEF);
    }
    // (the library keeps a ring of horizontally resized rows; recomputing them per output row gives the same integers)
    std::vector<int> rows_buf(2 * (size_t)dsize.width);
    int* R[2] = {rows_buf.data(), rows_buf.data() + dsize.width};
    for (int dy = 0; dy < dsize.height; dy++) {
        for (int k = 0; k < ksize; ++k) {
            int sy = yofs[dy] - ksize2 + 1 + k;
            sy = sy < 0 ? 0 : (sy >= ssize.height ? ssize.height - 1 : sy);
            const uchar* S = src.ptr(sy);
            int dx = 0;
            for (; dx < xmax; dx++) R[k][dx] = S[xofs[dx]] * ialpha[dx * 2] + S[xofs[dx] + 1] * ialpha[dx * 2 + 1];
            for (; dx < dsize.width; dx++) R[k][dx] = S[xofs[dx]] * COEF;
        }
        const int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
        uchar* D = dst.ptr(dy);
        for (int x = 0; x < dsize.width; x++)
            D[x] = (uchar)((((b0 * (R[0][x] >> 4)) >> 16) + ((b1 * (R[1][x] >> 4)) >> 16) + 2) >> 2);
    }
}

// smooth.cpp / filter.cpp: GaussianBlur on 8UC1 = separable filter with the float Gaussian kernel of getGaussianKernel()
// converted to 8-bit fixed point (createSeparableLinearFilter: both kernels x 256, rounded; int row sums; the column pass
// returns (sum + 2^15) >> 16, saturated).  A view without BORDER_ISOLATED reads the pixels around it in its parent
// matrix (FilterEngine::apply with the located ROI); only beyond the parent is the border extrapolated.
void GaussianBlur(InputArray _src, OutputArray _dst, Size ksize, double sigma1, double sigma2, int borderType) {
    ++shim_call_counts()[2];
    Mat src = _src.getMat();
    if (src.type() != CV_8UC1) throw std::runtime_error("cv shim: GaussianBlur is 8UC1 only");
    _dst.create(src.size(), src.type());
    Mat dst = _dst.getMat();
    if (sigma2 <= 0) sigma2 = sigma1;
    if (ksize.width <= 0 || ksize.height <= 0 || !(ksize.width & 1) || !(ksize.height & 1) || sigma1 <= 0)
        throw std::runtime_error("cv shim: GaussianBlur needs an explicit odd kernel size and sigma");
    auto taps = [](int n, double sigma) {   // getGaussianKernel(n, sigma, CV_32F), then convertTo(CV_32S, 256)
        std::vector<float> cf(n);
        const double scale2X = -0.5 / (sigma * sigma);
        double sum = 0;
        for (int i = 0; i < n; i++) {
            const double x = i - (n - 1) * 0.5;
            const double t = std::exp(scale2X * x * x);
            cf[i] = (float)t;
            sum += cf[i];
        }
        sum = 1. / sum;
        std::vector<int> k(n);
        for (int i = 0; i < n; i++) { cf[i] = (float)(cf[i] * sum); k[i] = cvRound((double)cf[i] * 256); }
        return k;
    };
    const std::vector<int> kx = taps(ksize.width, sigma1), ky = taps(ksize.height, sigma2);
    const int rx = ksize.width / 2, ry = ksize.height / 2;
    Size whole(src.cols, src.rows);
    Point ofs(0, 0);
    const bool isolated = (borderType & BORDER_ISOLATED) != 0;
    if (!isolated) src.locateROI(whole, ofs);
    const int bt = borderType & ~BORDER_ISOLATED;
    const uchar* base = src.data - (size_t)ofs.y * src.step - ofs.x;   // pixel (0, 0) of the parent matrix
    const int W = src.cols, H = src.rows;
    std::vector<int> hsum((size_t)(H + 2 * ry) * W);
    for (int y = -ry; y < H + ry; ++y) {
        const int py = border_interpolate(y + ofs.y, whole.height, bt);
        const uchar* row = base + (size_t)py * src.step;
        int* out = &hsum[(size_t)(y + ry) * W];
        for (int x = 0; x < W; ++x) {
            int s = 0;
            for (int k = -rx; k <= rx; ++k) s += kx[k + rx] * row[border_interpolate(x + k + ofs.x, whole.width, bt)];
            out[x] = s;
        }
    }
    for (int y = 0; y < H; ++y) {
        uchar* D = dst.ptr(y);
        for (int x = 0; x < W; ++x) {
            int s = 0;
            for (int k = 0; k <= 2 * ry; ++k) s += ky[k] * hsum[(size_t)(y + k) * W + x];
            D[x] = saturate_u8((s + (1 << 15)) >> 16);
        }
    }
}

void undistort(InputArray _src, OutputArray _dst, InputArray, InputArray _D, InputArray) {
    // cv::undistort with zero distortion and newCameraMatrix = cameraMatrix remaps every pixel onto itself (bilinear at integer
    // positions); the configurations the oracle runs have D = 0, anything else is refused here
    Mat D = _D.getMat();
    for (int r = 0; r < D.rows; ++r) for (int c = 0; c < D.cols * D.channels(); ++c)
        if ((D.depth() == CV_32F ? ((const float*)D.ptr(r))[c] : D.depth() == CV_64F ? ((const double*)D.ptr(r))[c] : 1.0) != 0.0)
            throw std::runtime_error("cv shim: undistort with distortion coefficients is not implemented");
    Mat src = _src.getMat();
    src.copyTo(_dst.getMatRef());
}
void undistortPoints(InputArray, OutputArray, InputArray, InputArray, InputArray, InputArray) {
    throw std::runtime_error("cv shim: undistortPoints is not implemented (the oracle's cameras have no distortion)");
}
// calib3d/src/calibration.cpp cvRodrigues2, rotation VECTOR -> matrix (the direction Track::calcSE3toXYZInfo uses, src/Track.cpp:298-299):
// computed in double, theta = |r|; below DBL_EPSILON the identity, otherwise c I + (1 - c) r r^T + s [r]x over the unit axis;
// the result has the depth of the input
void Rodrigues(InputArray _src, OutputArray _dst) {
    Mat src = _src.getMat();
    if (!((src.rows == 3 && src.cols == 1) || (src.rows == 1 && src.cols == 3)) || src.channels() != 1 || (src.depth() != CV_32F && src.depth() != CV_64F))
        throw std::runtime_error("cv shim: Rodrigues takes a 3x1 / 1x3 rotation vector of floats or doubles (the matrix -> vector direction is not implemented)");
    double r[3];
    for (int i = 0; i < 3; ++i) {
        const int rr = src.rows == 3 ? i : 0, cc = src.rows == 3 ? 0 : i;
        r[i] = src.depth() == CV_32F ? (double)src.at<float>(rr, cc) : src.at<double>(rr, cc);
    }
    const double theta = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (!(theta < DBL_EPSILON)) {
        const double c = std::cos(theta), s = std::sin(theta), c1 = 1. - c, itheta = theta ? 1. / theta : 0.;
        const double x = r[0] * itheta, y = r[1] * itheta, z = r[2] * itheta;
        const double rrt[9] = {x * x, x * y, x * z, x * y, y * y, y * z, x * z, y * z, z * z};
        const double rx[9] = {0, -z, y, z, 0, -x, -y, x, 0};
        const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        for (int k = 0; k < 9; ++k) R[k] = c * I[k] + c1 * rrt[k] + s * rx[k];
    }
    Mat out(3, 3, src.depth() == CV_32F ? CV_32FC1 : CV_64FC1);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        if (src.depth() == CV_32F) out.at<float>(i, j) = (float)R[3 * i + j]; else out.at<double>(i, j) = R[3 * i + j];
    }
    out.copyTo(_dst.getMatRef());
}

// how often the stand-in's image functions ran: the drop-in build (tests/dropin), where ORBextractor.cpp is the binding over libse2gpu,
// must never reach FAST / resize / GaussianBlur (oracle/ref_pipeline_driver.cpp reports the counts)
long long* shim_call_counts() { static long long c[4] = {0, 0, 0, 0}; return c; }
std::function<int(const float*, const float*, int, unsigned char*)>& shim_fundamental_hook() {
    static std::function<int(const float*, const float*, int, unsigned char*)> h;
    return h;
}

// ------------------------------------------------------------------------------------------------ features2d
// fast.cpp FAST_t<16> + fast_score.cpp cornerScore<16>: a pixel is a corner when more than 8 contiguous pixels of the
// 16-ring are all darker than v - t or all brighter than v + t; its score is the largest t for which that still holds
// (cornerScore, started at the call's threshold) minus nothing - the function returns (-b0 - 1); non-maximum suppression
// compares the score (stored as uchar, 0 where no corner) with the 8 neighbours of the 3 buffered rows.
static const int kRing[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
                                 {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};
static int corner_score16(const uchar* ptr, const int pixel[25], int threshold) {
    const int K = 8, N = K * 3 + 1;
    const int v = ptr[0];
    short d[N];
    for (int k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min((int)d[k + 1], (int)d[k + 2]);
        a = std::min(a, (int)d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, (int)d[k + 4]);
        a = std::min(a, (int)d[k + 5]);
        a = std::min(a, (int)d[k + 6]);
        a = std::min(a, (int)d[k + 7]);
        a = std::min(a, (int)d[k + 8]);
        a0 = std::max(a0, std::min(a, (int)d[k]));
        a0 = std::max(a0, std::min(a, (int)d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max((int)d[k + 1], (int)d[k + 2]);
        b = std::max(b, (int)d[k + 3]);
        b = std::max(b, (int)d[k + 4]);
        b = std::max(b, (int)d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, (int)d[k + 6]);
        b = std::max(b, (int)d[k + 7]);
        b = std::max(b, (int)d[k + 8]);
        b0 = std::min(b0, std::max(b, (int)d[k]));
        b0 = std::min(b0, std::max(b, (int)d[k + 9]));
    }
    return -b0 - 1;
}
void FAST(InputArray _img, std::vector<KeyPoint>& keypoints, int threshold, bool nonmax_suppression) {
    ++shim_call_counts()[0];
    Mat img = _img.getMat();
    if (img.type() != CV_8UC1) throw std::runtime_error("cv shim: FAST is 8UC1 only");
    const int K = 8, N = 16 + K + 1;
    int pixel[25];
    for (int k = 0; k < 16; ++k) pixel[k] = kRing[k][0] + kRing[k][1] * (int)img.step;
    for (int k = 16; k < 25; ++k) pixel[k] = pixel[k - 16];
    keypoints.clear();
    threshold = std::min(std::max(threshold, 0), 255);
    if (img.cols < 7 || img.rows < 7) return;
    std::vector<uchar> bufs(3 * (size_t)img.cols, 0);
    std::vector<std::vector<int>> cps(3);
    uchar* buf[3] = {bufs.data(), bufs.data() + img.cols, bufs.data() + 2 * (size_t)img.cols};
    for (int i = 3; i < img.rows - 2; i++) {
        const uchar* ptr = img.ptr(i) + 3;
        uchar* curr = buf[(i - 3) % 3];
        std::vector<int>& cornerpos = cps[(i - 3) % 3];
        std::memset(curr, 0, img.cols);
        cornerpos.clear();
        if (i < img.rows - 3) {
            for (int j = 3; j < img.cols - 3; j++, ptr++) {
                const int v = ptr[0];
                bool corner = false;
                {   // darker arc
                    const int vt = v - threshold;
                    int count = 0;
                    for (int k = 0; k < N; k++) {
                        if (ptr[pixel[k]] < vt) { if (++count > K) { corner = true; break; } }
                        else count = 0;
                    }
                }
                if (!corner) {   // brighter arc
                    const int vt = v + threshold;
                    int count = 0;
                    for (int k = 0; k < N; k++) {
                        if (ptr[pixel[k]] > vt) { if (++count > K) { corner = true; break; } }
                        else count = 0;
                    }
                }
                if (corner) {
                    cornerpos.push_back(j);
                    if (nonmax_suppression) curr[j] = (uchar)corner_score16(ptr, pixel, threshold);
                }
            }
        }
        if (i == 3) continue;
        const uchar* prev = buf[(i - 4 + 3) % 3];
        const uchar* pprev = buf[(i - 5 + 3) % 3];
        const std::vector<int>& cp = cps[(i - 4 + 3) % 3];
        for (int j : cp) {
            const int score = prev[j];
            if (!nonmax_suppression ||
                (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] && score > pprev[j] && score > pprev[j + 1] &&
                 score > curr[j - 1] && score > curr[j] && score > curr[j + 1]))
                keypoints.push_back(KeyPoint((float)j, (float)(i - 1), 7.f, -1, (float)score));
        }
    }
}

// keypoint.cpp KeyPointsFilter::retainBest, as OpenCV 3.2 writes it: std::nth_element by response, then std::partition keeps
// everything that ties with the n-th response.  The reference cuts the list to n right afterwards (ORBextractor.cpp:693-694,
// 708-709), so which of the tied points survive, and in what order they enter the level's list, is what THIS toolchain's
// std::nth_element leaves in the first n places: libstdc++'s introselect, the library a GCC build of the reference links.  That
// is the default since round 5 (it is what a real build does; the restatement and the HIP path compute the same permutation,
// oracle/stl_nth.h).  SE2_REF_RETAIN=stable is the order rounds 1-4 defined instead - the n best by (response descending,
// position in the list ascending) - kept as a labelled alternative.
void KeyPointsFilter::retainBest(std::vector<KeyPoint>& keypoints, int n_points) {
    if (n_points < 0 || keypoints.size() <= (size_t)n_points) return;
    if (n_points == 0) { keypoints.clear(); return; }
    static const bool stable = [] { const char* e = std::getenv("SE2_REF_RETAIN"); return e && std::string(e) == "stable"; }();
    if (stable) {
        std::stable_sort(keypoints.begin(), keypoints.end(), [](const KeyPoint& a, const KeyPoint& b) { return a.response > b.response; });
        keypoints.resize(n_points);
        return;
    }
    std::nth_element(keypoints.begin(), keypoints.begin() + n_points, keypoints.end(),
                     [](const KeyPoint& a, const KeyPoint& b) { return a.response > b.response; });
    const float ambiguous = keypoints[n_points - 1].response;
    auto new_end = std::partition(keypoints.begin() + n_points, keypoints.end(), [&](const KeyPoint& k) { return k.response >= ambiguous; });
    keypoints.resize(new_end - keypoints.begin());
}

// ------------------------------------------------------------------------------------------------ SVD
// lapack.cpp JacobiSVDImpl_<float>: one-sided Jacobi on the rows of A^T, products and norms accumulated in double, the
// rotation applied in float; singular values sorted descending; missing left vectors are not needed by cvu::triangulate
// (it reads vt only) and are not completed here.
void SVD::compute(InputArray _src, OutputArray _w, OutputArray _u, OutputArray _vt, int) {
    Mat src = _src.getMat();
    if (src.type() != CV_32FC1 || src.rows < src.cols) throw std::runtime_error("cv shim: SVD of a 32F matrix with rows >= cols only");
    const int m = src.rows, n = src.cols;
    Mat At = src.t();                    // n x m: row i = column i of A
    Mat Vt = Mat::eye(n, n, CV_32FC1);
    std::vector<double> W(n);
    for (int i = 0; i < n; i++) {
        double sd = 0;
        for (int k = 0; k < m; k++) { const float t = At.at<float>(i, k); sd += (double)t * t; }
        W[i] = sd;
    }
    const float eps = FLT_EPSILON * 2;
    const int max_iter = std::max(m, 30);
    for (int iter = 0; iter < max_iter; iter++) {
        bool changed = false;
        for (int i = 0; i < n - 1; i++)
            for (int j = i + 1; j < n; j++) {
                float* Ai = At.ptr<float>(i);
                float* Aj = At.ptr<float>(j);
                double a = W[i], p = 0, b = W[j];
                for (int k = 0; k < m; k++) p += (double)Ai[k] * Aj[k];
                if (std::abs(p) <= eps * std::sqrt((double)a * b)) continue;
                p *= 2;
                const double beta = a - b, gamma = hypot((double)p, beta);
                float c, s;
                if (beta < 0) {
                    const double delta = (gamma - beta) * 0.5;
                    s = (float)std::sqrt(delta / gamma);
                    c = (float)(p / (gamma * s * 2));
                } else {
                    c = (float)std::sqrt((gamma + beta) / (gamma * 2));
                    s = (float)(p / (gamma * c * 2));
                }
                a = b = 0;
                for (int k = 0; k < m; k++) {
                    const float t0 = c * Ai[k] + s * Aj[k];
                    const float t1 = -s * Ai[k] + c * Aj[k];
                    Ai[k] = t0; Aj[k] = t1;
                    a += (double)t0 * t0; b += (double)t1 * t1;
                }
                W[i] = a; W[j] = b;
                changed = true;
                float* Vi = Vt.ptr<float>(i);
                float* Vj = Vt.ptr<float>(j);
                for (int k = 0; k < n; k++) {
                    const float t0 = c * Vi[k] + s * Vj[k];
                    const float t1 = -s * Vi[k] + c * Vj[k];
                    Vi[k] = t0; Vj[k] = t1;
                }
            }
        if (!changed) break;
    }
    for (int i = 0; i < n; i++) {
        double sd = 0;
        for (int k = 0; k < m; k++) { const float t = At.at<float>(i, k); sd += (double)t * t; }
        W[i] = std::sqrt(sd);
    }
    for (int i = 0; i < n - 1; i++) {
        int j = i;
        for (int k = i + 1; k < n; k++) if (W[j] < W[k]) j = k;
        if (i != j) {
            std::swap(W[i], W[j]);
            for (int k = 0; k < m; k++) std::swap(At.at<float>(i, k), At.at<float>(j, k));
            for (int k = 0; k < n; k++) std::swap(Vt.at<float>(i, k), Vt.at<float>(j, k));
        }
    }
    Mat w(n, 1, CV_32FC1);
    for (int i = 0; i < n; i++) w.at<float>(i, 0) = (float)W[i];
    w.copyTo(_w.getMatRef());
    Vt.copyTo(_vt.getMatRef());
    Mat U(m, n, CV_32FC1);
    for (int i = 0; i < n; i++)
        for (int k = 0; k < m; k++) U.at<float>(k, i) = W[i] > 0 ? (float)(At.at<float>(i, k) / W[i]) : 0.f;
    U.copyTo(_u.getMatRef());
}

}  // namespace cv

// ------------------------------------------------------------------------------------------------ FileStorage (structure only)
namespace cv {
namespace {
std::map<std::string, std::vector<FsNodePtr>>& fs_files() {
    static std::map<std::string, std::vector<FsNodePtr>> files;
    return files;
}
char fs_dt(int depth) {
    switch (depth) {
        case CV_8U: return 'u';
        case CV_8S: return 'c';
        case CV_16U: return 'w';
        case CV_16S: return 's';
        case CV_32S: return 'i';
        case CV_32F: return 'f';
        case CV_64F: return 'd';
    }
    throw std::runtime_error("cv shim: FileStorage: matrix depth");
}
int fs_depth(char dt) {
    switch (dt) {
        case 'u': return CV_8U;
        case 'c': return CV_8S;
        case 'w': return CV_16U;
        case 's': return CV_16S;
        case 'i': return CV_32S;
        case 'f': return CV_32F;
        case 'd': return CV_64F;
    }
    throw std::runtime_error("cv shim: FileStorage: matrix dt");
}
void fs_dump_node(const FsNode& n, std::ostringstream& os) {
    char buf[32];
    switch (n.kind) {
        case FsNode::INT: os << "I " << n.i << "\n"; break;
        case FsNode::REAL: {
            uint64_t b;
            std::memcpy(&b, &n.r, 8);
            std::snprintf(buf, sizeof buf, "%016llx", (unsigned long long)b);
            os << "R " << buf << "\n";
            break;
        }
        case FsNode::STR: os << "S " << n.s << "\n"; break;
        case FsNode::SEQ:
        case FsNode::MAP: {
            const bool map = n.kind == FsNode::MAP;
            os << (map ? "{" : "[") << (n.flow ? ":" : "") << "\n";
            for (const auto& kv : n.kids) {
                if (map) os << "K " << kv.first << "\n";
                fs_dump_node(*kv.second, os);
            }
            os << (map ? "}" : "]") << "\n";
            break;
        }
        case FsNode::MAT: {
            os << "M " << n.m.rows << " " << n.m.cols << " " << fs_dt(n.m.depth()) << " ";
            const size_t rb = (size_t)n.m.cols * n.m.elemSize();
            for (int r = 0; r < n.m.rows; ++r)
                for (size_t k = 0; k < rb; ++k) { std::snprintf(buf, sizeof buf, "%02x", n.m.ptr(r)[k]); os << buf; }
            os << "\n";
            break;
        }
        default: throw std::runtime_error("cv shim: FileStorage: empty node in a file");
    }
}
}  // namespace

std::vector<FsNodePtr>& shim_fs_file(const std::string& path) { return fs_files()[path]; }
bool shim_fs_exists(const std::string& path) { return fs_files().count(path) != 0; }
void shim_fs_erase(const std::string& path) { fs_files().erase(path); }

FsNodePtr shim_fs_node(const Mat& m) {      // cvWrite of a CvMat: !!opencv-matrix {rows, cols, dt, data}; an empty Mat is 0 x 0 of 'u'
    if (m.channels() != 1 && !m.empty()) throw std::runtime_error("cv shim: FileStorage: multi-channel matrix");
    FsNodePtr p = std::make_shared<FsNode>();
    p->kind = FsNode::MAT;
    if (m.empty()) p->m = Mat();
    else p->m = m.clone();
    return p;
}
void shim_fs_read(const FileNode& n, Mat& v) {
    if (n.empty()) { v.release(); return; }     // read(node, mat, default_mat): the default is an empty matrix
    if (n.kind() != FsNode::MAT) throw std::runtime_error("cv shim: FileNode is not a matrix");
    v = n.n->m.clone();
}

std::string shim_fs_dump(const std::string& path) {
    if (!shim_fs_exists(path)) throw std::runtime_error("cv shim: no such file: " + path);
    std::ostringstream os;
    for (const FsNodePtr& d : shim_fs_file(path)) {
        os << "D\n";
        for (const auto& kv : d->kids) { os << "K " << kv.first << "\n"; fs_dump_node(*kv.second, os); }
    }
    return os.str();
}

void shim_fs_inject(const std::string& path, const std::string& events) {
    shim_fs_erase(path);
    std::vector<FsNodePtr>& docs = shim_fs_file(path);
    std::vector<FsNodePtr> open;      // open containers; open[0] = the current document
    std::string key;
    std::istringstream is(events);
    std::string line;
    auto add = [&](const FsNodePtr& v) {
        if (open.empty()) throw std::runtime_error("cv shim: inject: value before the first D");
        FsNode& t = *open.back();
        if (t.kind == FsNode::MAP && key.empty()) throw std::runtime_error("cv shim: inject: value without a key inside a map");
        t.kids.emplace_back(t.kind == FsNode::MAP ? key : std::string(), v);
        key.clear();
    };
    while (std::getline(is, line)) {
        if (line.empty()) continue;
        const char c = line[0];
        const std::string rest = line.size() > 2 ? line.substr(2) : std::string();
        FsNodePtr p = std::make_shared<FsNode>();
        if (c == 'D') { p->kind = FsNode::MAP; docs.push_back(p); open.assign(1, p); }
        else if (c == 'K') key = rest;
        else if (c == 'I') { p->kind = FsNode::INT; p->i = std::stoi(rest); add(p); }
        else if (c == 'R') { p->kind = FsNode::REAL; const uint64_t b = std::stoull(rest, nullptr, 16); std::memcpy(&p->r, &b, 8); add(p); }
        else if (c == 'S') { p->kind = FsNode::STR; p->s = rest; add(p); }
        else if (c == '[' || c == '{') { p->kind = c == '[' ? FsNode::SEQ : FsNode::MAP; p->flow = line.size() > 1 && line[1] == ':'; add(p); open.push_back(p); }
        else if (c == ']' || c == '}') {
            if (open.size() < 2 || open.back()->kind != (c == ']' ? FsNode::SEQ : FsNode::MAP)) throw std::runtime_error("cv shim: inject: unbalanced " + line);
            open.pop_back();
        } else if (c == 'M') {
            std::istringstream ms(rest);
            int rows, cols; char dt; std::string hex;
            ms >> rows >> cols >> dt >> hex;
            p->kind = FsNode::MAT;
            if (rows > 0 && cols > 0) {
                p->m.create(rows, cols, fs_depth(dt));
                const size_t bytes = (size_t)rows * cols * p->m.elemSize();
                if (hex.size() != 2 * bytes) throw std::runtime_error("cv shim: inject: matrix data length");
                for (size_t k = 0; k < bytes; ++k) p->m.data[k] = (uchar)std::stoi(hex.substr(2 * k, 2), nullptr, 16);
            }
            add(p);
        } else throw std::runtime_error("cv shim: inject: unknown line " + line);
    }
}
}  // namespace cv
