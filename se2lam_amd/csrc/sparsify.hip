// libse2gpu - Sparsifier::DoMarginalizeSE3XYZ (/root/reference/src/sparsifier.cpp:105-275) for a BATCH of key-frame pairs:
// the feature constraints GlobalMapper::CreateFeatEdge hands to GlobalBA (src/GlobalMapper.cpp:744-840) - SURVEY.md 8(f).4.
//
// One wave per pair.  Lane l takes the map points l, l + 64, ...: for each, the forward-difference Jacobians
// (delta 1e-6, the reference's) of z = KF^-1 * MP w.r.t. KF.toMinimalVector() and MP for both key frames, the point's
// 3x3 block and its 12x3 coupling, and the point's term of the Schur complement H11 - H12 H22^-1 H21 (H22 is block
// diagonal).  The 12x12 terms are summed over the wave (fixed butterfly order: deterministic); lane 0 finishes the pair:
// regulariser, 12x12 inverse, the 6x12 forward-difference Jacobian of (KF1^-1 KF2).toMinimalVector(), I = (J H^-1 J')^-1,
// spectrum clamped to [1e-6, 1e4] (cyclic Jacobi on the symmetric 6x6), z_out = KF1^-1 KF2.
// The step to I is ill-conditioned by construction (six gauge freedoms held by a 1e-6 regulariser), so this file is
// compiled with -ffp-contract=off and follows the operation order of oracle/sparsify_ref.cpp statement by statement:
// the forward differences, which decide the result, are then reproduced to the bit.
// [3P g2o 20160424 SE3Quat, Eigen quaternion arithmetic] restated; Eigen's LDLT / inverse / JacobiSVD replaced by exact
// small solvers (closed form 3x3 via Gauss-Jordan, Gauss-Jordan with partial pivoting, Jacobi eigen-solver).
#include <cmath>
#include <vector>

#include "common.h"

using namespace se2gpu;

namespace {

struct Q3 { double w, x, y, z, t[3]; };

__device__ inline void rotate(const Q3& q, const double v[3], double o[3]) {
    double uv[3] = {q.y * v[2] - q.z * v[1], q.z * v[0] - q.x * v[2], q.x * v[1] - q.y * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    o[0] = v[0] + q.w * uv[0] + (q.y * uv[2] - q.z * uv[1]);
    o[1] = v[1] + q.w * uv[1] + (q.z * uv[0] - q.x * uv[2]);
    o[2] = v[2] + q.w * uv[2] + (q.x * uv[1] - q.y * uv[0]);
}
__device__ inline void normalize_rot(Q3& q) {
    if (q.w < 0) { q.w = -q.w; q.x = -q.x; q.y = -q.y; q.z = -q.z; }
    const double n = sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    q.w /= n; q.x /= n; q.y /= n; q.z /= n;
}
__device__ inline Q3 from_pose12(const double* p) {
    const double* R = p;
    Q3 q;
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q.w = 0.5 * t; t = 0.5 / t;
        q.x = (R[7] - R[5]) * t; q.y = (R[2] - R[6]) * t; q.z = (R[3] - R[1]) * t;
    } else if (!(R[4] > R[0]) && !(R[8] > R[0])) {
        t = sqrt(R[0] - R[4] - R[8] + 1.0);
        q.x = 0.5 * t; t = 0.5 / t;
        q.w = (R[7] - R[5]) * t; q.y = (R[3] + R[1]) * t; q.z = (R[6] + R[2]) * t;
    } else if (R[4] > R[0] && !(R[8] > R[4])) {
        t = sqrt(R[4] - R[8] - R[0] + 1.0);
        q.y = 0.5 * t; t = 0.5 / t;
        q.w = (R[2] - R[6]) * t; q.z = (R[7] + R[5]) * t; q.x = (R[1] + R[3]) * t;
    } else {
        t = sqrt(R[8] - R[0] - R[4] + 1.0);
        q.z = 0.5 * t; t = 0.5 / t;
        q.w = (R[3] - R[1]) * t; q.x = (R[2] + R[6]) * t; q.y = (R[5] + R[7]) * t;
    }
    q.t[0] = p[9]; q.t[1] = p[10]; q.t[2] = p[11];
    normalize_rot(q);
    return q;
}
__device__ inline void to_pose12(const Q3& q, double* p) {
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w, txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    p[0] = 1 - (tyy + tzz); p[1] = txy - twz; p[2] = txz + twy;
    p[3] = txy + twz; p[4] = 1 - (txx + tzz); p[5] = tyz - twx;
    p[6] = txz - twy; p[7] = tyz + twx; p[8] = 1 - (txx + tyy);
    p[9] = q.t[0]; p[10] = q.t[1]; p[11] = q.t[2];
}
__device__ inline void to_min(const Q3& q, double v[6]) { v[0] = q.t[0]; v[1] = q.t[1]; v[2] = q.t[2]; v[3] = q.x; v[4] = q.y; v[5] = q.z; }
__device__ inline Q3 from_min(const double v[6]) {
    Q3 q;
    const double w = 1. - v[3] * v[3] - v[4] * v[4] - v[5] * v[5];
    if (w > 0) { q.w = sqrt(w); q.x = v[3]; q.y = v[4]; q.z = v[5]; }
    else { q.w = 0; q.x = -v[3]; q.y = -v[4]; q.z = -v[5]; }
    q.t[0] = v[0]; q.t[1] = v[1]; q.t[2] = v[2];
    return q;
}
__device__ inline Q3 inverse(const Q3& q) {
    Q3 r;
    r.w = q.w; r.x = -q.x; r.y = -q.y; r.z = -q.z;
    const double m[3] = {q.t[0] * -1., q.t[1] * -1., q.t[2] * -1.};
    rotate(r, m, r.t);
    return r;
}
__device__ inline Q3 mul(const Q3& a, const Q3& b) {
    Q3 r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    double rt[3];
    rotate(a, b.t, rt);
    r.t[0] = a.t[0] + rt[0]; r.t[1] = a.t[1] + rt[1]; r.t[2] = a.t[2] + rt[2];
    normalize_rot(r);
    return r;
}
__device__ inline void map_inv(const Q3& kf, const double mp[3], double z[3]) {
    const Q3 inv = inverse(kf);
    double r[3];
    rotate(inv, mp, r);
    z[0] = r[0] + inv.t[0]; z[1] = r[1] + inv.t[1]; z[2] = r[2] + inv.t[2];
}
__device__ void jacobian_se3xyz(const Q3& kf, const double mp[3], double J[27]) {
    const double delta = 1e-6;
    double zref[3], v6[6];
    map_inv(kf, mp, zref);
    to_min(kf, v6);
    for (int i = 0; i < 9; ++i) {
        double zd[3];
        if (i < 6) {
            double vd[6];
            for (int k = 0; k < 6; ++k) vd[k] = v6[k];
            vd[i] += delta;
            map_inv(from_min(vd), mp, zd);
        } else {
            double md[3] = {mp[0], mp[1], mp[2]};
            md[i - 6] += delta;
            map_inv(kf, md, zd);
        }
        for (int r = 0; r < 3; ++r) J[9 * r + i] = (zd[r] - zref[r]) / delta;
    }
}
// Gauss-Jordan with partial pivoting, in place (A <- A^-1), B = n x n scratch; row-major
__device__ bool invert(double* A, double* B, int n) {
    for (int i = 0; i < n * n; ++i) B[i] = 0.0;
    for (int i = 0; i < n; ++i) B[i * n + i] = 1.0;
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int r = c + 1; r < n; ++r)
            if (fabs(A[r * n + c]) > fabs(A[piv * n + c])) piv = r;
        if (A[piv * n + c] == 0.0) return false;
        if (piv != c)
            for (int k = 0; k < n; ++k) {
                double t = A[c * n + k]; A[c * n + k] = A[piv * n + k]; A[piv * n + k] = t;
                t = B[c * n + k]; B[c * n + k] = B[piv * n + k]; B[piv * n + k] = t;
            }
        const double d = 1.0 / A[c * n + c];
        for (int k = 0; k < n; ++k) { A[c * n + k] *= d; B[c * n + k] *= d; }
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            const double f = A[r * n + c];
            if (f == 0.0) continue;
            for (int k = 0; k < n; ++k) { A[r * n + k] -= f * A[c * n + k]; B[r * n + k] -= f * B[c * n + k]; }
        }
    }
    for (int i = 0; i < n * n; ++i) A[i] = B[i];
    return true;
}
__device__ void clamp_spectrum6(double* I) {
    double A[36], V[36];
    for (int i = 0; i < 36; ++i) { A[i] = I[i]; V[i] = (i % 7 == 0) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 30; ++sweep)
        for (int p = 0; p < 5; ++p)
            for (int q = p + 1; q < 6; ++q) {
                const double apq = A[6 * p + q];
                if (apq == 0.0) continue;
                const double theta = (A[6 * q + q] - A[6 * p + p]) / (2 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 6; ++k) {
                    const double akp = A[6 * k + p], akq = A[6 * k + q];
                    A[6 * k + p] = c * akp - s * akq;
                    A[6 * k + q] = s * akp + c * akq;
                }
                for (int k = 0; k < 6; ++k) {
                    const double apk = A[6 * p + k], aqk = A[6 * q + k];
                    A[6 * p + k] = c * apk - s * aqk;
                    A[6 * q + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 6; ++k) {
                    const double vkp = V[6 * k + p], vkq = V[6 * k + q];
                    V[6 * k + p] = c * vkp - s * vkq;
                    V[6 * k + q] = s * vkp + c * vkq;
                }
            }
    double lam[6];
    for (int k = 0; k < 6; ++k) {
        const double l = A[7 * k];
        lam[k] = l < 0 ? 1e-6 : fmin(fmax(l, 1e-6), 1e4);
    }
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) {
            double v = 0;
            for (int k = 0; k < 6; ++k) v += V[6 * r + k] * lam[k] * V[6 * c + k];
            I[6 * r + c] = v;
        }
    for (int r = 0; r < 6; ++r)
        for (int c = r + 1; c < 6; ++c) { const double m = 0.5 * (I[6 * r + c] + I[6 * c + r]); I[6 * r + c] = I[6 * c + r] = m; }
}

// mp_ptr: points of pair p = [mp_ptr[p], mp_ptr[p+1]); mm_ptr: measurements of point j = [mm_ptr[j], mm_ptr[j+1]) in the
// order they were given; m_kf in {0, 1}; scratch: 288 doubles per POINT (its key-frame blocks and its Schur term), so that
// the sums can be taken in exactly the reference's order whatever lane computed a point.
__global__ __launch_bounds__(64) void k_sparsify(int npairs, const double* __restrict__ kf12, const int* __restrict__ mp_ptr,
                                                 const double* __restrict__ mp_xyz, const int* __restrict__ mm_ptr,
                                                 const int* __restrict__ m_kf, const double* __restrict__ m_info,
                                                 double* __restrict__ scratch, double* __restrict__ z_out,
                                                 double* __restrict__ info_out) {
    const int p = blockIdx.x;
    if (p >= npairs) return;
    const int lane = threadIdx.x;
    const Q3 KF0 = from_pose12(kf12 + 24 * (size_t)p), KF1 = from_pose12(kf12 + 24 * (size_t)p + 12);
    __shared__ double h11s[144];
    for (int j = mp_ptr[p] + lane; j < mp_ptr[p + 1]; j += 64) {
        // this point's part of H11 (12 x 12): [0,144) its J' W J key-frame blocks, [144,288) its Schur term
        double* acc = scratch + (size_t)j * 288;
        for (int i = 0; i < 288; ++i) acc[i] = 0.0;
        double Hmm[9], Hkm[36];
        for (int i = 0; i < 9; ++i) Hmm[i] = 0.0;
        for (int i = 0; i < 36; ++i) Hkm[i] = 0.0;
        const double mp[3] = {mp_xyz[3 * (size_t)j], mp_xyz[3 * (size_t)j + 1], mp_xyz[3 * (size_t)j + 2]};
        for (int i = mm_ptr[j]; i < mm_ptr[j + 1]; ++i) {
            const int k = m_kf[i];
            double J[27], WJ[27];
            jacobian_se3xyz(k ? KF1 : KF0, mp, J);
            const double* W = m_info + 9 * (size_t)i;
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 9; ++c) WJ[9 * r + c] = W[3 * r] * J[c] + W[3 * r + 1] * J[9 + c] + W[3 * r + 2] * J[18 + c];
            for (int a = 0; a < 9; ++a)
                for (int b = 0; b < 9; ++b) {
                    const double v = J[a] * WJ[b] + J[9 + a] * WJ[9 + b] + J[18 + a] * WJ[18 + b];
                    if (a < 6 && b < 6) acc[12 * (6 * k + a) + 6 * k + b] += v;
                    else if (a >= 6 && b >= 6) Hmm[3 * (a - 6) + (b - 6)] += v;
                    else if (a < 6) Hkm[3 * (6 * k + a) + (b - 6)] += v;
                }
        }
        bool any = false;
        for (int i = 0; i < 9; ++i) any |= Hmm[i] != 0.0;
        double B9[9];
        if (!any || !invert(Hmm, B9, 3)) continue;
        double BD[36];
        for (int r = 0; r < 12; ++r)
            for (int c = 0; c < 3; ++c) BD[3 * r + c] = Hkm[3 * r] * Hmm[c] + Hkm[3 * r + 1] * Hmm[3 + c] + Hkm[3 * r + 2] * Hmm[6 + c];
        for (int r = 0; r < 12; ++r)
            for (int c = 0; c < 12; ++c)
                acc[144 + 12 * r + c] += BD[3 * r] * Hkm[3 * c] + BD[3 * r + 1] * Hkm[3 * c + 1] + BD[3 * r + 2] * Hkm[3 * c + 2];
    }
    __threadfence_block();
    __syncthreads();
    // Order of the sums: the reference (and the oracle) add the key-frame blocks measurement by measurement (= point by
    // point: the measurements are grouped by point), then the regulariser, then subtract the Schur terms point by point.
    // H11 is kept regular only by that 1e-6 I, so the order is part of the result (a pre-summed H11 moved the clamped
    // spectrum by up to 100 % at 220 points): every entry is summed by one lane over the points in exactly that order.
    for (int i = lane; i < 144; i += 64) {
        double hsum = 0.0;
        for (int j = mp_ptr[p]; j < mp_ptr[p + 1]; ++j) hsum += scratch[(size_t)j * 288 + i];
        if (i % 13 == 0) hsum += 1e-6;
        for (int j = mp_ptr[p]; j < mp_ptr[p + 1]; ++j) hsum -= scratch[(size_t)j * 288 + 144 + i];
        h11s[i] = hsum;
    }
    __syncthreads();
    if (lane != 0) return;
    double H11[144];
    for (int i = 0; i < 144; ++i) H11[i] = h11s[i];
    const Q3 zref_q = mul(inverse(KF0), KF1);
    double zref[6], v1[6], v2[6], J[72];
    to_min(zref_q, zref);
    to_min(KF0, v1);
    to_min(KF1, v2);
    const double delta = 1e-6;
    for (int i = 0; i < 12; ++i) {
        double zd[6], vd[6];
        if (i < 6) {
            for (int k = 0; k < 6; ++k) vd[k] = v1[k];
            vd[i] += delta;
            to_min(mul(inverse(from_min(vd)), KF1), zd);
        } else {
            for (int k = 0; k < 6; ++k) vd[k] = v2[k];
            vd[i - 6] += delta;
            to_min(mul(inverse(KF0), from_min(vd)), zd);
        }
        for (int r = 0; r < 6; ++r) J[12 * r + i] = (zd[r] - zref[r]) / delta;
    }
    double B144[144];
    invert(H11, B144, 12);
    double JH[72], Mx[36], B36[36];
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 12; ++c) { double v = 0; for (int k = 0; k < 12; ++k) v += J[12 * r + k] * H11[12 * k + c]; JH[12 * r + c] = v; }
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) { double v = 0; for (int k = 0; k < 12; ++k) v += JH[12 * r + k] * J[12 * c + k]; Mx[6 * r + c] = v; }
    invert(Mx, B36, 6);
    for (int r = 0; r < 6; ++r)
        for (int c = r + 1; c < 6; ++c) { const double m = 0.5 * (Mx[6 * r + c] + Mx[6 * c + r]); Mx[6 * r + c] = Mx[6 * c + r] = m; }
    clamp_spectrum6(Mx);
    for (int i = 0; i < 36; ++i) info_out[36 * (size_t)p + i] = Mx[i];
    to_pose12(zref_q, z_out + 12 * (size_t)p);
}

}  // namespace

extern "C" {

int se2gpu_sparsify_se3xyz(int npairs, const double* kf12, const int32_t* mp_ptr, const double* mp_xyz,
                           const int32_t* m_ptr, const int32_t* m_kf, const int32_t* m_mp, const double* m_info,
                           double* z_out12, double* info_out36) {
    SE2_REQUIRE(have_device(), SE2GPU_ERR_NO_DEVICE, "no HIP device visible (libse2gpu has no CPU fallback)");
    SE2_REQUIRE(npairs >= 0, SE2GPU_ERR_INVALID, "sparsify: negative pair count");
    if (npairs == 0) return SE2GPU_OK;
    SE2_REQUIRE(kf12 && mp_ptr && m_ptr && z_out12 && info_out36, SE2GPU_ERR_INVALID, "sparsify: NULL argument");
    // the CSR pointers index host and device arrays: start at 0, never decrease
    SE2_REQUIRE(mp_ptr[0] == 0 && m_ptr[0] == 0, SE2GPU_ERR_INVALID, "sparsify: mp_ptr / m_ptr must start at 0");
    for (int p = 0; p < npairs; ++p)
        SE2_REQUIRE(mp_ptr[p + 1] >= mp_ptr[p] && m_ptr[p + 1] >= m_ptr[p], SE2GPU_ERR_INVALID,
                    "sparsify: mp_ptr / m_ptr decrease at pair %d", p);
    const int NP = mp_ptr[npairs], NM = m_ptr[npairs];
    SE2_REQUIRE(NP >= 0 && NM >= 0 && (NP == 0 || mp_xyz) && (NM == 0 || (m_kf && m_mp && m_info)), SE2GPU_ERR_INVALID,
                "sparsify: NULL array");
    // measurements grouped by point (stable): a point's list in the order its measurements were given; measurements of a
    // key frame other than 0 / 1 are dropped (sparsifier.cpp:117-119)
    std::vector<int> mm_ptr((size_t)NP + 1, 0), kf_s, src;
    for (int p = 0; p < npairs; ++p) {
        const int n = mp_ptr[p + 1] - mp_ptr[p];
        for (int i = m_ptr[p]; i < m_ptr[p + 1]; ++i) {
            if (m_kf[i] != 0 && m_kf[i] != 1) continue;
            SE2_REQUIRE(m_mp[i] >= 0 && m_mp[i] < n, SE2GPU_ERR_INVALID, "sparsify: pair %d, measurement %d names point %d of %d", p, i, m_mp[i], n);
            mm_ptr[(size_t)mp_ptr[p] + m_mp[i] + 1]++;
        }
    }
    for (int j = 0; j < NP; ++j) mm_ptr[j + 1] += mm_ptr[j];
    kf_s.resize(mm_ptr[NP]);
    src.resize(mm_ptr[NP]);
    {
        std::vector<int> fill(mm_ptr.begin(), mm_ptr.end() - 1);
        for (int p = 0; p < npairs; ++p)
            for (int i = m_ptr[p]; i < m_ptr[p + 1]; ++i) {
                if (m_kf[i] != 0 && m_kf[i] != 1) continue;
                const int t = fill[(size_t)mp_ptr[p] + m_mp[i]]++;
                kf_s[t] = m_kf[i];
                src[t] = i;
            }
    }
    std::vector<double> info_s(9 * (size_t)src.size());
    for (size_t t = 0; t < src.size(); ++t) std::memcpy(&info_s[9 * t], m_info + 9 * (size_t)src[t], 72);
    DevBuf<double> d_kf, d_mp, d_info, d_scratch, d_z, d_out;
    DevBuf<int> d_mpp, d_mmp, d_mkf;
    hipStream_t st = nullptr;
    SE2_CHECK(d_kf.upload(kf12, 24 * (size_t)npairs, st));
    SE2_CHECK(d_mpp.upload(mp_ptr, (size_t)npairs + 1, st));
    // (no pair has points: mp_xyz may be NULL - Sparsifier::DoMarginalizeSE3XYZBatch passes xyz.data() of an empty vector)
    SE2_CHECK(d_mp.reserve(3 * (size_t)std::max(NP, 1)));
    if (NP) SE2_HIP(hipMemcpyAsync(d_mp.p, mp_xyz, 3 * (size_t)NP * 8, hipMemcpyHostToDevice, st));
    SE2_CHECK(d_mmp.upload(mm_ptr.data(), mm_ptr.size(), st));
    SE2_CHECK(d_mkf.reserve(std::max<size_t>(kf_s.size(), 1)));
    if (!kf_s.empty()) SE2_HIP(hipMemcpyAsync(d_mkf.p, kf_s.data(), kf_s.size() * 4, hipMemcpyHostToDevice, st));
    SE2_CHECK(d_info.reserve(std::max<size_t>(info_s.size(), 1)));
    if (!info_s.empty()) SE2_HIP(hipMemcpyAsync(d_info.p, info_s.data(), info_s.size() * 8, hipMemcpyHostToDevice, st));
    SE2_CHECK(d_scratch.reserve((size_t)std::max(NP, 1) * 288));
    SE2_CHECK(d_z.reserve(12 * (size_t)npairs));
    SE2_CHECK(d_out.reserve(36 * (size_t)npairs));
    hipLaunchKernelGGL(k_sparsify, dim3(npairs), dim3(64), 0, st, npairs, d_kf.p, d_mpp.p, d_mp.p, d_mmp.p, d_mkf.p, d_info.p,
                       d_scratch.p, d_z.p, d_out.p);
    SE2_HIP(hipGetLastError());
    SE2_HIP(hipMemcpyAsync(z_out12, d_z.p, 12 * (size_t)npairs * 8, hipMemcpyDeviceToHost, st));
    SE2_HIP(hipMemcpyAsync(info_out36, d_out.p, 36 * (size_t)npairs * 8, hipMemcpyDeviceToHost, st));
    SE2_HIP(hipStreamSynchronize(st));
    return SE2GPU_OK;
}

}  // extern "C"
