// Track::doTriangulate (/root/reference/src/Track.cpp:378-419) for all matches of a frame pair at once - SURVEY.md 8(f).3.
//   cvu::triangulate   /root/reference/src/cvutil.cpp:46-59   linear (DLT) triangulation: the right singular vector of the
//                                                            smallest singular value of the 4x4 system, dehomogenised
//   Config::acceptDepth /root/reference/src/Config.cpp:188-190
//   cvu::checkParallax /root/reference/src/cvutil.cpp:95-101
// One thread per feature of the reference key frame.  cv::SVD::compute is OpenCV's one-sided Jacobi (Hestenes) in FP32;
// the same algorithm runs here in FP64 on the FP32 system matrix, the point is rounded to FP32 at the end.  The CPU test
// restatement executes the identical operation sequence, so the two agree bit for bit; agreement with OpenCV's own
// float iteration is unpinned (OpenCV is not installed), expected ~1e-5 relative.
// Compiled with -ffp-contract=off like the matchers.
#include "track_ws.h"

namespace se2gpu {
namespace {

// smallest right singular vector of the 4x4 matrix a (row-major); v4 = that vector (unnormalised sign)
__device__ inline void smallest_right_singular_vector(const double a_in[16], double v4[4]) {
    double At[4][4], Vt[4][4], W[4];   // At: rows = columns of A (one-sided Jacobi works on A^T), Vt: accumulates V^T
    for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 4; ++k) {
            At[i][k] = a_in[k * 4 + i];
            Vt[i][k] = i == k ? 1.0 : 0.0;
        }
    for (int i = 0; i < 4; ++i) {
        double sd = 0;
        for (int k = 0; k < 4; ++k) sd += At[i][k] * At[i][k];
        W[i] = sd;
    }
    const double eps = 2.220446049250313e-16 * 10;
    for (int iter = 0; iter < 30; ++iter) {
        bool changed = false;
        for (int i = 0; i < 3; ++i)
            for (int j = i + 1; j < 4; ++j) {
                double a = W[i], p = 0, b = W[j];
                for (int k = 0; k < 4; ++k) p += At[i][k] * At[j][k];
                if (fabs(p) <= eps * sqrt(a * b)) continue;
                p *= 2;
                const double beta = a - b, gamma = sqrt(p * p + beta * beta);
                double c, s;
                if (beta < 0) {
                    const double delta = (gamma - beta) * 0.5;
                    s = sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                } else {
                    c = sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }
                a = 0;
                b = 0;
                for (int k = 0; k < 4; ++k) {
                    const double t0 = c * At[i][k] + s * At[j][k];
                    const double t1 = -s * At[i][k] + c * At[j][k];
                    At[i][k] = t0;
                    At[j][k] = t1;
                    a += t0 * t0;
                    b += t1 * t1;
                }
                W[i] = a;
                W[j] = b;
                changed = true;
                for (int k = 0; k < 4; ++k) {
                    const double t0 = c * Vt[i][k] + s * Vt[j][k];
                    const double t1 = -s * Vt[i][k] + c * Vt[j][k];
                    Vt[i][k] = t0;
                    Vt[j][k] = t1;
                }
            }
        if (!changed) break;
    }
    int m = 0;
    for (int i = 1; i < 4; ++i)
        if (W[i] < W[m]) m = i;   // first minimum
    for (int k = 0; k < 4; ++k) v4[k] = Vt[m][k];
}

__global__ void k_triangulate(int n, const se2gpu_keypoint* __restrict__ kps_ref,
                              const se2gpu_keypoint* __restrict__ kps_cur, int n_cur, int* __restrict__ match_idx,
                              const uint8_t* __restrict__ has_obs, const float* __restrict__ P1,
                              const float* __restrict__ P2, float ox, float oy, float oz, float lower, float upper,
                              float min_cos, float* __restrict__ pos, uint8_t* __restrict__ good,
                              int* __restrict__ counters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    good[i] = 0;
    const int mi = match_idx[i];
    if (mi < 0 || mi >= n_cur) return;
    if (has_obs && has_obs[i]) {   // Track.cpp:393-397: the map point of the key frame is kept by the caller
        atomicAdd(&counters[1], 1);
        return;
    }
    const float x1 = kps_ref[i].x, y1 = kps_ref[i].y, x2 = kps_cur[mi].x, y2 = kps_cur[mi].y;
    double A[16];
    for (int c = 0; c < 4; ++c) {   // rows in float as cv::Mat arithmetic does, then widened
        A[0 + c] = (double)(x1 * P1[8 + c] - P1[0 + c]);
        A[4 + c] = (double)(y1 * P1[8 + c] - P1[4 + c]);
        A[8 + c] = (double)(x2 * P2[8 + c] - P2[0 + c]);
        A[12 + c] = (double)(y2 * P2[8 + c] - P2[4 + c]);
    }
    double v[4];
    smallest_right_singular_vector(A, v);
    const float w = (float)v[3];
    const float px = (float)v[0] / w, py = (float)v[1] / w, pz = (float)v[2] / w;
    if (pz >= lower && pz <= upper) {
        pos[3 * i] = px;       // mLocalMPs[i] = pos only for an accepted depth (Track.cpp:407-408); zero elsewhere
        pos[3 * i + 1] = py;
        pos[3 * i + 2] = pz;
        // checkParallax(o1 = 0, o2 = Ocam, pt3 = pos): |p1 . p2| / (|p1| |p2|) < minCos
        const float q0 = px - ox, q1 = py - oy, q2 = pz - oz;
        const float dotf = px * q0 + py * q1 + pz * q2;                                  // Point3_<float>::dot: float arithmetic
        const double dot = (double)dotf;                                                 // cv::norm(scalar) = |.| in double
        const double n1 = sqrt((double)px * px + (double)py * py + (double)pz * pz);     // cv::norm(Point3f) in double
        const double n2 = sqrt((double)q0 * q0 + (double)q1 * q1 + (double)q2 * q2);
        const float cosp = (float)(fabs(dot) / (n1 * n2));
        if (cosp < min_cos) {
            good[i] = 1;
            atomicAdd(&counters[0], 1);
        }
    } else {
        match_idx[i] = -1;   // Track.cpp:413-415
    }
}

}  // namespace
}  // namespace se2gpu

using namespace se2gpu;

extern "C" int se2gpu_triangulate(int n, const se2gpu_keypoint* kps_ref, const se2gpu_keypoint* kps_cur, int n_cur,
                                  int32_t* match_idx, const uint8_t* has_observation, const float* P_ref,
                                  const float* P_cur, const float* Ocam, float lower_depth, float upper_depth,
                                  int min_degree, float* pos_out, uint8_t* good_parallax, int* n_good,
                                  int* n_tracked_old) {
    SE2_REQUIRE(have_device(), SE2GPU_ERR_NO_DEVICE, "no HIP device visible (libse2gpu has no CPU fallback)");
    SE2_REQUIRE(n >= 0 && n_cur >= 0, SE2GPU_ERR_INVALID, "triangulate: negative size");
    SE2_REQUIRE(min_degree >= 1 && min_degree <= 4, SE2GPU_ERR_INVALID, "triangulate: minDegree must be 1..4");
    if (n_good) *n_good = 0;
    if (n_tracked_old) *n_tracked_old = 0;
    if (n == 0) return SE2GPU_OK;
    SE2_REQUIRE(kps_ref && kps_cur && match_idx && P_ref && P_cur && Ocam && pos_out && good_parallax,
                SE2GPU_ERR_INVALID, "triangulate: NULL argument");
    for (int i = 0; i < n; ++i)   // a match that points past the current frame's features is the caller's error, not "no match"
        SE2_REQUIRE(match_idx[i] < n_cur, SE2GPU_ERR_INVALID, "triangulate: match_idx[%d] = %d but the frame has %d features", i,
                    match_idx[i], n_cur);
    const float minCos[4] = {0.9998f, 0.9994f, 0.9986f, 0.9976f};   // cvutil.cpp:96
    DevBuf<se2gpu_keypoint> d_k1, d_k2;
    DevBuf<int> d_m, d_cnt;
    DevBuf<uint8_t> d_obs, d_good;
    DevBuf<float> d_P, d_pos;
    hipStream_t st = nullptr;
    SE2_CHECK(d_k1.upload(kps_ref, (size_t)n, st));
    SE2_CHECK(d_k2.upload(kps_cur, (size_t)std::max(n_cur, 1), st));
    SE2_CHECK(d_m.upload(match_idx, (size_t)n, st));
    if (has_observation) SE2_CHECK(d_obs.upload(has_observation, (size_t)n, st));
    float Pm[24];
    std::memcpy(Pm, P_ref, 12 * sizeof(float));
    std::memcpy(Pm + 12, P_cur, 12 * sizeof(float));
    SE2_CHECK(d_P.upload(Pm, 24, st));
    SE2_CHECK(d_pos.reserve(3 * (size_t)n));
    SE2_CHECK(d_good.reserve((size_t)n));
    SE2_CHECK(d_cnt.reserve(2));
    SE2_HIP(hipMemsetAsync(d_cnt.p, 0, 2 * sizeof(int), st));
    SE2_HIP(hipMemsetAsync(d_pos.p, 0, 3 * (size_t)n * sizeof(float), st));
    hipLaunchKernelGGL(k_triangulate, dim3((n + 127) / 128), dim3(128), 0, st, n, d_k1.p, d_k2.p, n_cur, d_m.p,
                       has_observation ? d_obs.p : (const uint8_t*)nullptr, d_P.p, d_P.p + 12, Ocam[0], Ocam[1], Ocam[2],
                       lower_depth, upper_depth, minCos[min_degree - 1], d_pos.p, d_good.p, d_cnt.p);
    SE2_HIP(hipGetLastError());
    int cnt[2] = {0, 0};
    SE2_HIP(hipMemcpyAsync(pos_out, d_pos.p, 3 * (size_t)n * sizeof(float), hipMemcpyDeviceToHost, st));
    SE2_HIP(hipMemcpyAsync(good_parallax, d_good.p, (size_t)n, hipMemcpyDeviceToHost, st));
    SE2_HIP(hipMemcpyAsync(match_idx, d_m.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, st));
    SE2_HIP(hipMemcpyAsync(cnt, d_cnt.p, sizeof(cnt), hipMemcpyDeviceToHost, st));
    SE2_HIP(hipStreamSynchronize(st));
    if (n_good) *n_good = cnt[0];
    if (n_tracked_old) *n_tracked_old = cnt[1];
    return SE2GPU_OK;
}

// The same pass with the persistent workspace of a tracking thread: one packed upload, one packed download.
extern "C" int se2gpu_track_triangulate(se2gpu_track* h, int n, const se2gpu_keypoint* kps_ref,
                                        const se2gpu_keypoint* kps_cur, int n_cur, int32_t* match_idx,
                                        const uint8_t* has_observation, const float* P_ref, const float* P_cur,
                                        const float* Ocam, float lower_depth, float upper_depth, int min_degree,
                                        float* pos_out, uint8_t* good_parallax, int* n_good, int* n_tracked_old) {
    SE2_REQUIRE(h, SE2GPU_ERR_INVALID, "track_triangulate: NULL handle");
    SE2_REQUIRE(n >= 0 && n_cur >= 0, SE2GPU_ERR_INVALID, "triangulate: negative size");
    SE2_REQUIRE(min_degree >= 1 && min_degree <= 4, SE2GPU_ERR_INVALID, "triangulate: minDegree must be 1..4");
    if (n_good) *n_good = 0;
    if (n_tracked_old) *n_tracked_old = 0;
    if (n == 0) return SE2GPU_OK;
    SE2_REQUIRE(kps_ref && kps_cur && match_idx && P_ref && P_cur && Ocam && pos_out && good_parallax,
                SE2GPU_ERR_INVALID, "triangulate: NULL argument");
    for (int i = 0; i < n; ++i)   // a match that points past the current frame's features is the caller's error, not "no match"
        SE2_REQUIRE(match_idx[i] < n_cur, SE2GPU_ERR_INVALID, "triangulate: match_idx[%d] = %d but the frame has %d features", i,
                    match_idx[i], n_cur);
    const float minCos[4] = {0.9998f, 0.9994f, 0.9986f, 0.9976f};   // cvutil.cpp:96
    auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    // input  [kps_ref | kps_cur | match_idx | has_obs | P (24 floats)], output [pos | match_idx | counters | good]
    const size_t o_k1 = 0, o_k2 = up16(o_k1 + (size_t)n * sizeof(se2gpu_keypoint));
    const size_t o_m = up16(o_k2 + (size_t)std::max(n_cur, 1) * sizeof(se2gpu_keypoint));
    const size_t o_obs = up16(o_m + (size_t)n * sizeof(int)), o_P = up16(o_obs + (size_t)n);
    const size_t in_b = o_P + 24 * sizeof(float);
    const size_t q_pos = 0, q_m = up16(q_pos + 3 * (size_t)n * sizeof(float)), q_cnt = up16(q_m + (size_t)n * sizeof(int));
    const size_t q_good = q_cnt + 16, out_b = up16(q_good + (size_t)n);
    SE2_CHECK(h->h_in.reserve(in_b));
    SE2_CHECK(h->d_in.reserve(in_b));
    SE2_CHECK(h->h_out.reserve(out_b));
    SE2_CHECK(h->d_out.reserve(out_b));
    uint8_t* hi = h->h_in.p;
    std::memcpy(hi + o_k1, kps_ref, (size_t)n * sizeof(se2gpu_keypoint));
    if (n_cur) std::memcpy(hi + o_k2, kps_cur, (size_t)n_cur * sizeof(se2gpu_keypoint));
    std::memcpy(hi + o_m, match_idx, (size_t)n * sizeof(int));
    if (has_observation) std::memcpy(hi + o_obs, has_observation, (size_t)n);
    std::memcpy(hi + o_P, P_ref, 12 * sizeof(float));
    std::memcpy(hi + o_P + 12 * sizeof(float), P_cur, 12 * sizeof(float));
    hipStream_t st = h->stream;
    SE2_HIP(hipMemcpyAsync(h->d_in.p, hi, in_b, hipMemcpyHostToDevice, st));
    uint8_t* di = h->d_in.p;
    uint8_t* dout = h->d_out.p;
    SE2_HIP(hipMemsetAsync(dout, 0, q_good, st));   // positions, counters
    // the kernel updates match_idx in place: work on the copy inside the output block
    SE2_HIP(hipMemcpyAsync(dout + q_m, di + o_m, (size_t)n * sizeof(int), hipMemcpyDeviceToDevice, st));
    const float* dP = (const float*)(di + o_P);
    hipLaunchKernelGGL(k_triangulate, dim3((n + 127) / 128), dim3(128), 0, st, n, (const se2gpu_keypoint*)(di + o_k1),
                       (const se2gpu_keypoint*)(di + o_k2), n_cur, (int*)(dout + q_m),
                       has_observation ? (const uint8_t*)(di + o_obs) : (const uint8_t*)nullptr, dP, dP + 12, Ocam[0],
                       Ocam[1], Ocam[2], lower_depth, upper_depth, minCos[min_degree - 1], (float*)(dout + q_pos),
                       dout + q_good, (int*)(dout + q_cnt));
    SE2_HIP(hipGetLastError());
    SE2_HIP(hipMemcpyAsync(h->h_out.p, dout, out_b, hipMemcpyDeviceToHost, st));
    SE2_HIP(hipStreamSynchronize(st));
    const uint8_t* ho = h->h_out.p;
    std::memcpy(pos_out, ho + q_pos, 3 * (size_t)n * sizeof(float));
    std::memcpy(match_idx, ho + q_m, (size_t)n * sizeof(int));
    std::memcpy(good_parallax, ho + q_good, (size_t)n);
    int cnt[2];
    std::memcpy(cnt, ho + q_cnt, sizeof(cnt));
    if (n_good) *n_good = cnt[0];
    if (n_tracked_old) *n_tracked_old = cnt[1];
    return SE2GPU_OK;
}
