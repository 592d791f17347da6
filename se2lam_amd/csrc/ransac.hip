// Track::removeOutliers (/root/reference/src/Track.cpp:308-344) - SURVEY.md 8(f).3: the epipolar outlier filter that
// follows MatchByWindow in Track::mTrack.  The reference calls cv::findFundamentalMat(pt1, pt2, mask) with the defaults
// (FM_RANSAC, 3 px, confidence 0.99), i.e. [3P, OpenCV 3.2 calib3d fundam.cpp / ptsetreg.cpp]
//   fewer than 7 points  no mask (every match is dropped by Track.cpp:338-341)
//   exactly 7            7-point algorithm, all points kept
//   8 .. 14              LMedS over 7-point samples (median of the FP32 errors, sorted as integers)
//   15 and more          RANSAC over 7-point samples, cv::RNG(-1), at most 1000 iterations with the adaptive stop
//
// RANSAC is sequential only through its stop rule.  The sample indices depend on nothing but the RNG and the point
// count, so the host draws all of them (7000 multiply-with-carry steps), and
//   k_fm_models   one thread per sample: null space of the 7x9 system (Householder QR of its transpose), the cubic
//                 det(lambda F1 + (1 - lambda) F2) = 0, up to three fundamental matrices               (FP64)
//   k_fm_score    one wave per (sample, model): inlier count over all correspondences by wave ballot, or the LMedS
//                 median by a rank sort across lanes
//   k_fm_select   one workgroup: replays the reference's loop over the per-sample records - only prefix maxima of the
//                 inlier count can change the state, found 64 samples at a time with a wave prefix-max - then writes
//                 the mask of the winning model
// run in one stream with a single download (mask + count).
//
// The null-space basis is arbitrary in OpenCV as well (SVD::FULL_UV completes V from seeded random vectors); the det = 0
// members of the pencil do not depend on it.  cv::solveCubic's acos / cos / cubeRoot are replaced by Newton iterations
// made of + - * / sqrt, with the same branches and root order, so that the result does not depend on a math library;
// the CPU test restatement runs the identical operation sequence and the masks agree bit for bit.  Compiled with
// -ffp-contract=off.  Agreement with OpenCV itself is unpinned (not installed here).
#include "track_ws.h"

#include <algorithm>

namespace se2gpu {
namespace {

constexpr int kMaxIters = 1000;     // createRANSACPointSetRegistrator / createLMeDSPointSetRegistrator default
constexpr int kModelPoints = 7;
constexpr double kConfidence = 0.99, kThreshold = 3.0, kLmedsOutlierRatio = 0.45;

__device__ inline double fm_cbrt(double v) {   // v >= 0
    if (!(v > 0)) return v;
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    const int ex = (int)((bits >> 52) & 0x7ff) - 1023;
    const int q = ex >= 0 ? ex / 3 : -((-ex + 2) / 3);
    double y = __longlong_as_double((long long)((unsigned long long)(q + 1023) << 52));
    for (int it = 0; it < 12; ++it) y = (2.0 * y + v / (y * y)) / 3.0;
    return y;
}

// cos(acos(r) / 3) = the root of 4c^3 - 3c = r in [1/2, 1]; Newton from c = 1 descends monotonically
__device__ inline double fm_cos_third(double r) {
    if (r > 1.0) r = 1.0;
    if (r < -1.0) r = -1.0;
    double c = 1.0;
    for (int it = 0; it < 64; ++it) {
        const double gp = 12.0 * c * c - 3.0;
        if (!(gp > 0)) break;
        const double g = (4.0 * c * c - 3.0) * c - r;
        const double cn = c - g / gp;
        if (cn == c) break;
        c = cn;
    }
    return c;
}

// cv::solveCubic: roots of c0 x^3 + c1 x^2 + c2 x + c3 in the library's order
__device__ inline int fm_solve_cubic(const double c[4], double x[3]) {
    double a0 = c[0], a1 = c[1], a2 = c[2], a3 = c[3];
    int n = 0;
    x[0] = x[1] = x[2] = 0;
    if (a0 == 0) {
        if (a1 == 0) {
            if (a2 == 0) n = a3 == 0 ? -1 : 0;
            else { x[0] = -a3 / a2; n = 1; }
        } else {
            double d = a2 * a2 - 4 * a1 * a3;
            if (d >= 0) {
                d = sqrt(d);
                const double q1 = (-a2 + d) * 0.5, q2 = (a2 + d) * -0.5;
                if (fabs(q1) > fabs(q2)) { x[0] = q1 / a1; x[1] = a3 / q1; }
                else { x[0] = q2 / a1; x[1] = a3 / q2; }
                n = d > 0 ? 2 : 1;
            }
        }
    } else {
        a0 = 1. / a0;
        a1 *= a0; a2 *= a0; a3 *= a0;
        const double Q = (a1 * a1 - 3 * a2) * (1. / 9);
        const double R = (2 * a1 * a1 * a1 - 9 * a1 * a2 + 27 * a3) * (1. / 54);
        const double Qcubed = Q * Q * Q;
        double d = Qcubed - R * R;
        if (d >= 0) {   // three real roots: -2 sqrt(Q) cos(theta/3 + 2 k pi/3) - a1/3, theta = acos(R / sqrt(Q^3))
            const double ct = fm_cos_third(R / sqrt(Qcubed));
            const double st = sqrt(1.0 - ct * ct);
            const double t0 = -2 * sqrt(Q), t2 = a1 * (1. / 3);
            const double h = 0.8660254037844386;
            x[0] = t0 * ct - t2;
            x[1] = t0 * (-0.5 * ct - h * st) - t2;
            x[2] = t0 * (-0.5 * ct + h * st) - t2;
            n = 3;
        } else {
            d = sqrt(-d);
            double e = fm_cbrt(fabs(R) + d);
            if (R > 0) e = -e;
            x[0] = (e + Q / e) - a1 * (1. / 3);
            n = 1;
        }
    }
    return n;
}

// (f1, f2): the last two columns of Q in the Householder QR of a^T (9x7) span the null space of a (7x9)
__device__ inline void fm_null_space(const double a[7][9], double f1[9], double f2[9]) {
    double M[9][7], V[7][9], beta[7];
    for (int r = 0; r < 9; ++r)
        for (int c = 0; c < 7; ++c) M[r][c] = a[c][r];
    for (int k = 0; k < 7; ++k) {
        double sigma = 0;
        for (int r = k; r < 9; ++r) sigma += M[r][k] * M[r][k];
        const double norm = sqrt(sigma);
        const double alpha = M[k][k] > 0 ? -norm : norm;
        for (int r = 0; r < 9; ++r) V[k][r] = r < k ? 0.0 : M[r][k];
        V[k][k] = M[k][k] - alpha;
        double vn = 0;
        for (int r = k; r < 9; ++r) vn += V[k][r] * V[k][r];
        beta[k] = vn > 0 ? 2.0 / vn : 0.0;
        for (int c = k + 1; c < 7; ++c) {
            double s = 0;
            for (int r = k; r < 9; ++r) s += V[k][r] * M[r][c];
            s *= beta[k];
            for (int r = k; r < 9; ++r) M[r][c] -= s * V[k][r];
        }
    }
    for (int j = 0; j < 2; ++j) {
        double q[9];
        for (int r = 0; r < 9; ++r) q[r] = r == 7 + j ? 1.0 : 0.0;
        for (int k = 6; k >= 0; --k) {
            double s = 0;
            for (int r = k; r < 9; ++r) s += V[k][r] * q[r];
            s *= beta[k];
            for (int r = k; r < 9; ++r) q[r] -= s * V[k][r];
        }
        for (int r = 0; r < 9; ++r) (j == 0 ? f1 : f2)[r] = q[r];
    }
}

// run7Point (fundam.cpp): up to three row-major 3x3 matrices from the 7 correspondences idx[0..6]
__device__ inline int fm_run7point(const float2* __restrict__ m1, const float2* __restrict__ m2, const int* idx,
                                   double* F) {
    double a[7][9], f1[9], f2[9], c[4], r[3];
    for (int i = 0; i < 7; ++i) {
        const float2 p = m1[idx[i]], q = m2[idx[i]];
        const double x0 = p.x, y0 = p.y, x1 = q.x, y1 = q.y;
        a[i][0] = x1 * x0; a[i][1] = x1 * y0; a[i][2] = x1;
        a[i][3] = y1 * x0; a[i][4] = y1 * y0; a[i][5] = y1;
        a[i][6] = x0; a[i][7] = y0; a[i][8] = 1;
    }
    fm_null_space(a, f1, f2);
    for (int i = 0; i < 9; ++i) f1[i] -= f2[i];
    double t0 = f2[4] * f2[8] - f2[5] * f2[7];
    double t1 = f2[3] * f2[8] - f2[5] * f2[6];
    double t2 = f2[3] * f2[7] - f2[4] * f2[6];
    c[3] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2;
    c[2] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2 - f1[3] * (f2[1] * f2[8] - f2[2] * f2[7]) +
           f1[4] * (f2[0] * f2[8] - f2[2] * f2[6]) - f1[5] * (f2[0] * f2[7] - f2[1] * f2[6]) +
           f1[6] * (f2[1] * f2[5] - f2[2] * f2[4]) - f1[7] * (f2[0] * f2[5] - f2[2] * f2[3]) +
           f1[8] * (f2[0] * f2[4] - f2[1] * f2[3]);
    t0 = f1[4] * f1[8] - f1[5] * f1[7];
    t1 = f1[3] * f1[8] - f1[5] * f1[6];
    t2 = f1[3] * f1[7] - f1[4] * f1[6];
    c[0] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2;
    c[1] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2 - f2[3] * (f1[1] * f1[8] - f1[2] * f1[7]) +
           f2[4] * (f1[0] * f1[8] - f1[2] * f1[6]) - f2[5] * (f1[0] * f1[7] - f1[1] * f1[6]) +
           f2[6] * (f1[1] * f1[5] - f1[2] * f1[4]) - f2[7] * (f1[0] * f1[5] - f1[2] * f1[3]) +
           f2[8] * (f1[0] * f1[4] - f1[1] * f1[3]);
    const int n = fm_solve_cubic(c, r);
    if (n < 1 || n > 3) return n;
    for (int k = 0; k < n; ++k) {
        double* f = F + 9 * k;
        double lambda = r[k], mu = 1.;
        const double s = f1[8] * r[k] + f2[8];
        if (fabs(s) > 2.220446049250313e-16) {   // normalise F(3,3) to 1
            mu = 1. / s;
            lambda *= mu;
            f[8] = 1.;
        } else
            f[8] = 0.;
        for (int i = 0; i < 8; ++i) f[i] = f1[i] * lambda + f2[i] * mu;
    }
    return n;
}

// FMEstimatorCallback::computeError: the larger of the two squared point-to-epipolar-line distances, rounded to FP32
__device__ inline float fm_error(const double* F, float2 p1, float2 p2) {
    double a = F[0] * p1.x + F[1] * p1.y + F[2];
    double b = F[3] * p1.x + F[4] * p1.y + F[5];
    double c = F[6] * p1.x + F[7] * p1.y + F[8];
    const double s2 = 1. / (a * a + b * b);
    const double d2 = p2.x * a + p2.y * b + c;
    a = F[0] * p2.x + F[3] * p2.y + F[6];
    b = F[1] * p2.x + F[4] * p2.y + F[7];
    c = F[2] * p2.x + F[5] * p2.y + F[8];
    const double s1 = 1. / (a * a + b * b);
    const double d1 = p1.x * a + p1.y * b + c;
    const double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
    return (float)((e1 < e2) ? e2 : e1);   // std::max: a NaN in e1 wins, a NaN in e2 loses
}

// cv::RANSACUpdateNumIters
__host__ __device__ inline int fm_update_num_iters(double p, double ep, int model_points, int max_iters) {
    p = fmax(p, 0.); p = fmin(p, 1.);
    ep = fmax(ep, 0.); ep = fmin(ep, 1.);
    double num = fmax(1. - p, 2.2250738585072014e-308);
    double denom = 1. - pow(1. - ep, (double)model_points);
    if (denom < 2.2250738585072014e-308) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)rint(num / denom);
}

__global__ __launch_bounds__(64) void k_fm_models(const float2* __restrict__ m1, const float2* __restrict__ m2,
                                                   const int* __restrict__ subsets, int niters,
                                                   double* __restrict__ F, int* __restrict__ nmodels) {
    const int it = blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= niters) return;
    int idx[7];
    for (int i = 0; i < 7; ++i) idx[i] = subsets[it * 7 + i];
    double Fl[27];
    for (int i = 0; i < 27; ++i) Fl[i] = 0;
    const int nm = fm_run7point(m1, m2, idx, Fl);
    nmodels[it] = nm;
    for (int i = 0; i < 27; ++i) F[(size_t)it * 27 + i] = Fl[i];
}

// score[it*3 + k]: RANSAC - number of correspondences with error <= thr (as double); LMedS - the median error.
// Models that do not exist score -1 (RANSAC) / +inf (LMedS).
__global__ __launch_bounds__(256) void k_fm_score(const float2* __restrict__ m1, const float2* __restrict__ m2, int n,
                                                   const double* __restrict__ F, const int* __restrict__ nmodels,
                                                   int niters, int lmeds, float thr, double* __restrict__ score) {
    const int w = blockIdx.x * 4 + threadIdx.x / 64;
    const int lane = threadIdx.x & 63;
    if (w >= niters * 3) return;
    const int it = w / 3, k = w - 3 * it;
    const int nm = nmodels[it];
    if (k >= nm) {   // also nm <= 0
        if (lane == 0) score[w] = lmeds ? __longlong_as_double(0x7ff0000000000000ll) : -1.0;
        return;
    }
    double Fm[9];
    for (int i = 0; i < 9; ++i) Fm[i] = F[(size_t)it * 27 + 9 * k + i];
    if (!lmeds) {
        int good = 0;
        for (int i = lane; i < n; i += 64) good += fm_error(Fm, m1[i], m2[i]) <= thr;
        for (int s = 1; s < 64; s <<= 1) good += __shfl_xor(good, s);
        if (lane == 0) score[w] = (double)good;
        return;
    }
    // LMedS, n <= 14: std::sort of the float bit patterns AS INTS; an x86 default NaN carries the sign bit, so every
    // NaN gets the key 0xffc00000 and sorts first
    int key = INT32_MAX;
    if (lane < n) {
        const float e = fm_error(Fm, m1[lane], m2[lane]);
        key = e != e ? (int)0xffc00000 : __float_as_int(e);
    }
    int rank = 0;
    for (int j = 0; j < n; ++j) {
        const int kj = __shfl(key, j);
        rank += (kj < key) || (kj == key && j < lane);
    }
    const unsigned long long hi_m = __ballot(lane < n && rank == n / 2);
    const unsigned long long lo_m = __ballot(lane < n && rank == n / 2 - 1);
    const float hi = __int_as_float(__shfl(key, __ffsll((long long)hi_m) - 1));
    const float lo = __int_as_float(__shfl(key, __ffsll((long long)lo_m) - 1));
    const double median = n % 2 != 0 ? (double)hi : (double)((lo + hi) * 0.5);
    if (lane == 0) score[w] = median;
}

// Replays RANSACPointSetRegistrator::run / LMeDSPointSetRegistrator::run over the per-model scores and writes the mask.
// out_mask[n], out_info = {inliers, best sample, best model, iterations the reference would have run}
__global__ __launch_bounds__(256) void k_fm_select(const float2* __restrict__ m1, const float2* __restrict__ m2, int n,
                                                    const double* __restrict__ F, const double* __restrict__ score,
                                                    int niters, int lmeds, uint8_t* __restrict__ out_mask,
                                                    int* __restrict__ out_info) {
    __shared__ double s_F[9];
    __shared__ float s_thr;
    __shared__ int s_best, s_iters, s_count;
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) s_count = 0;
    if (tid < 64) {
        int best = -1, used = niters;
        float thr = (float)(kThreshold * kThreshold);
        if (!lmeds) {
            int max_good = 0, limit = niters;
            for (int c0 = 0; c0 < limit; c0 += 64) {
                const int it = c0 + lane;
                int v = -1, kf = 0;
                if (it < niters) {
                    for (int k = 0; k < 3; ++k) {
                        const int g = (int)score[it * 3 + k];
                        if (g > v) { v = g; kf = k; }   // the first model reaching the sample's maximum owns it
                    }
                }
                // exclusive prefix maximum over the lanes of the chunk
                int pm = v;
                for (int d = 1; d < 64; d <<= 1) {
                    const int o = __shfl_up(pm, d);
                    if (lane >= d) pm = max(pm, o);
                }
                int ex = __shfl_up(pm, 1);
                if (lane == 0) ex = -1;
                unsigned long long rec = __ballot(v > max(ex, max(max_good, kModelPoints - 1)));
                while (rec) {   // record breakers in sample order: the only samples that change the loop state
                    const int b = __ffsll((long long)rec) - 1;
                    rec &= rec - 1;
                    if (c0 + b >= limit) { rec = 0; break; }
                    const int g = __shfl(v, b);
                    best = (c0 + b) * 3 + __shfl(kf, b);
                    max_good = g;
                    limit = fm_update_num_iters(kConfidence, (double)(n - g) / n, kModelPoints, limit);
                }
            }
            used = limit;
        } else {
            // first strict minimum of the medians in (sample, model) order
            double bm = 1.7976931348623157e308;
            int bi = INT32_MAX;
            for (int w = lane; w < niters * 3; w += 64) {
                const double m = score[w];
                if (m < bm) { bm = m; bi = w; }
            }
            for (int s = 1; s < 64; s <<= 1) {
                const double om = __shfl_xor(bm, s);
                const int oi = __shfl_xor(bi, s);
                if (om < bm || (om == bm && oi < bi)) { bm = om; bi = oi; }
            }
            if (bm < 1.7976931348623157e308) {
                best = bi;
                double sigma = 2.5 * 1.4826 * (1 + 5. / (n - kModelPoints)) * sqrt(bm);
                sigma = fmax(sigma, 0.001);
                thr = (float)(sigma * sigma);
            }
        }
        if (lane == 0) { s_best = best; s_thr = thr; s_iters = used; }
        if (best >= 0 && lane < 9) s_F[lane] = F[(size_t)(best / 3) * 27 + 9 * (best % 3) + lane];
    }
    __syncthreads();
    const int best = s_best;
    int good = 0;
    for (int i = tid; i < n; i += 256) {
        uint8_t mk = 0;
        if (best >= 0) mk = fm_error(s_F, m1[i], m2[i]) <= s_thr;
        out_mask[i] = mk;
        good += mk;
    }
    for (int s = 1; s < 64; s <<= 1) good += __shfl_xor(good, s);
    if (lane == 0 && good) atomicAdd(&s_count, good);
    __syncthreads();
    if (tid == 0) {
        out_info[0] = s_count;
        out_info[1] = best >= 0 ? best / 3 : -1;
        out_info[2] = best >= 0 ? best % 3 : -1;
        out_info[3] = s_iters;
    }
}

struct CvRng {   // cv::RNG (multiply-with-carry), seeded with (uint64)-1 by both registrators
    uint64_t state;
    explicit CvRng(uint64_t s) : state(s ? s : 0xffffffffull) {}
    unsigned next() {
        state = (uint64_t)(unsigned)state * 4164903690u + (unsigned)(state >> 32);
        return (unsigned)state;
    }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

}  // namespace
}  // namespace se2gpu

using namespace se2gpu;

extern "C" int se2gpu_track_create(se2gpu_track** out) {
    SE2_REQUIRE(out, SE2GPU_ERR_INVALID, "track_create: NULL argument");
    *out = nullptr;
    SE2_REQUIRE(have_device(), SE2GPU_ERR_NO_DEVICE, "no HIP device visible (libse2gpu has no CPU fallback)");
    se2gpu_track* h = new se2gpu_track();
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
        delete h;
        return SE2GPU_ERR_HIP;
    }
    *out = h;
    return SE2GPU_OK;
}

extern "C" void se2gpu_track_destroy(se2gpu_track* h) {
    if (!h) return;
    if (h->stream) {
        (void)hipStreamSynchronize(h->stream);
        (void)hipStreamDestroy(h->stream);
    }
    delete h;
}

extern "C" int se2gpu_track_fundamental_mask(se2gpu_track* h, const float* pt1, const float* pt2, int n, uint8_t* mask,
                                             int* n_inliers) {
    SE2_REQUIRE(h && n_inliers, SE2GPU_ERR_INVALID, "fundamental_mask: NULL argument");
    SE2_REQUIRE(n >= 0, SE2GPU_ERR_INVALID, "fundamental_mask: negative size");
    *n_inliers = 0;
    h->last_info[0] = 0; h->last_info[1] = h->last_info[2] = -1; h->last_info[3] = 0;
    if (n == 0) return SE2GPU_OK;
    SE2_REQUIRE(pt1 && pt2 && mask, SE2GPU_ERR_INVALID, "fundamental_mask: NULL buffer");
    if (n < kModelPoints) {   // findFundamentalMat returns before creating the mask
        std::memset(mask, 0, (size_t)n);
        return SE2GPU_OK;
    }
    if (n == kModelPoints) {  // runKernel only; mask.setTo(1)
        std::memset(mask, 1, (size_t)n);
        *n_inliers = n;
        return SE2GPU_OK;
    }
    const int lmeds = n < 15;
    const int niters = lmeds ? std::max(fm_update_num_iters(kConfidence, kLmedsOutlierRatio, kModelPoints, kMaxIters), 3)
                             : kMaxIters;
    if (h->subsets_n != n) {  // PointSetRegistrator::getSubset: 7 distinct indices by rejection, one RNG for the run
        CvRng rng((uint64_t)-1);
        h->subsets.resize((size_t)kMaxIters * 7);
        for (int it = 0; it < kMaxIters; ++it) {
            int* idx = &h->subsets[(size_t)it * 7];
            for (int i = 0; i < 7; ++i) {
                for (;;) {
                    const int v = idx[i] = rng.uniform(0, n);
                    int j = 0;
                    for (; j < i; ++j)
                        if (v == idx[j]) break;
                    if (j == i) break;
                }
            }
        }
        h->subsets_n = n;
    }
    hipStream_t st = h->stream;
    // one upload: [pt1 | pt2 | subsets]
    const size_t pts_b = (size_t)n * 2 * sizeof(float), sub_b = (size_t)niters * 7 * sizeof(int);
    const size_t in_b = 2 * pts_b + sub_b;
    SE2_CHECK(h->h_in.reserve(in_b));
    SE2_CHECK(h->d_in.reserve(in_b));
    std::memcpy(h->h_in.p, pt1, pts_b);
    std::memcpy(h->h_in.p + pts_b, pt2, pts_b);
    std::memcpy(h->h_in.p + 2 * pts_b, h->subsets.data(), sub_b);
    SE2_HIP(hipMemcpyAsync(h->d_in.p, h->h_in.p, in_b, hipMemcpyHostToDevice, st));
    const float2* d_m1 = (const float2*)h->d_in.p;
    const float2* d_m2 = (const float2*)(h->d_in.p + pts_b);
    const int* d_sub = (const int*)(h->d_in.p + 2 * pts_b);
    SE2_CHECK(h->d_F.reserve((size_t)kMaxIters * 27));
    SE2_CHECK(h->d_score.reserve((size_t)kMaxIters * 3));
    SE2_CHECK(h->d_nm.reserve((size_t)kMaxIters));
    const size_t mask_b = ((size_t)n + 15) & ~(size_t)15;
    SE2_CHECK(h->d_out.reserve(mask_b + 4 * sizeof(int)));
    SE2_CHECK(h->h_out.reserve(mask_b + 4 * sizeof(int)));
    hipLaunchKernelGGL(k_fm_models, dim3((niters + 63) / 64), dim3(64), 0, st, d_m1, d_m2, d_sub, niters, h->d_F.p,
                       h->d_nm.p);
    hipLaunchKernelGGL(k_fm_score, dim3((niters * 3 + 3) / 4), dim3(256), 0, st, d_m1, d_m2, n, h->d_F.p, h->d_nm.p,
                       niters, lmeds, (float)(kThreshold * kThreshold), h->d_score.p);
    hipLaunchKernelGGL(k_fm_select, dim3(1), dim3(256), 0, st, d_m1, d_m2, n, h->d_F.p, h->d_score.p, niters, lmeds,
                       h->d_out.p, (int*)(h->d_out.p + mask_b));
    SE2_HIP(hipGetLastError());
    SE2_HIP(hipMemcpyAsync(h->h_out.p, h->d_out.p, mask_b + 4 * sizeof(int), hipMemcpyDeviceToHost, st));
    SE2_HIP(hipStreamSynchronize(st));
    std::memcpy(mask, h->h_out.p, (size_t)n);
    std::memcpy(h->last_info, h->h_out.p + mask_b, sizeof(h->last_info));
    *n_inliers = h->last_info[0];
    return SE2GPU_OK;
}

extern "C" int se2gpu_track_remove_outliers(se2gpu_track* h, const se2gpu_keypoint* kps1, int n1,
                                            const se2gpu_keypoint* kps2, int n2, int32_t* matches, int* n_inliers) {
    SE2_REQUIRE(h && n_inliers, SE2GPU_ERR_INVALID, "remove_outliers: NULL argument");
    SE2_REQUIRE(n1 >= 0 && n2 >= 0, SE2GPU_ERR_INVALID, "remove_outliers: negative size");
    *n_inliers = 0;
    if (n1 == 0) return SE2GPU_OK;
    SE2_REQUIRE(kps1 && matches && (n2 == 0 || kps2), SE2GPU_ERR_INVALID, "remove_outliers: NULL buffer");
    // Track.cpp:315-322: the matched pairs, in key-point order
    h->pt1.clear(); h->pt2.clear(); h->idx.clear();
    for (int i = 0; i < n1; ++i) {
        const int m = matches[i];
        if (m < 0) continue;
        SE2_REQUIRE(m < n2, SE2GPU_ERR_INVALID, "remove_outliers: match %d of feature %d is out of range", m, i);
        h->idx.push_back(i);
        h->pt1.push_back(kps1[i].x); h->pt1.push_back(kps1[i].y);
        h->pt2.push_back(kps2[m].x); h->pt2.push_back(kps2[m].y);
    }
    const int n = (int)h->idx.size();
    int inl = 0;
    h->mask.assign((size_t)std::max(n, 1), 0);
    if (n) SE2_CHECK(se2gpu_track_fundamental_mask(h, h->pt1.data(), h->pt2.data(), n, h->mask.data(), &inl));
    for (int i = 0; i < n; ++i)
        if (!h->mask[i]) matches[h->idx[i]] = -1;
    if (inl < 10) {   // Track.cpp:338-341: too few inliers, the whole frame is not trusted
        inl = 0;
        for (int i = 0; i < n1; ++i) matches[i] = -1;
    }
    *n_inliers = inl;
    return SE2GPU_OK;
}

// the winning (sample, model) and the number of iterations the reference loop would have executed, for tests / tracing
extern "C" int se2gpu_track_last_ransac(const se2gpu_track* h, int info[4]) {
    SE2_REQUIRE(h && info, SE2GPU_ERR_INVALID, "last_ransac: NULL argument");
    std::memcpy(info, h->last_info, sizeof(h->last_info));
    return SE2GPU_OK;
}
