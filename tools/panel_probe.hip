// Stage timing of the single-wave Cholesky panel kernel (copy of k_chol_panel of csrc/ba.hip with s_memtime stamps).
//   hipcc --offload-arch=gfx950 -O3 tools/panel_probe.hip -o tools/bin/panel_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int kNB = 32;
#ifndef WPE
#define WPE
#endif
__device__ inline double bcast_lane(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ inline double fast_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}
__device__ inline double fast_rsqrt(double d) {
    double y = __builtin_amdgcn_rsq(d);
    y = y * fma(-0.5 * d * y, y, 1.5);
    y = y * fma(-0.5 * d * y, y, 1.5);
    return y;
}
// VARIANT 0: product kernel.  1: bulk through v_readlane of lane j's row (symmetry), no LDS.  2: no bulk at all
// (chain only).  3: no chain (bulk only, inv constant).
template <int VARIANT>
__global__ __launch_bounds__(64) WPE void k_panel(double* __restrict__ A, double* __restrict__ R, int ld, int n, int nt, int k,
                                               double* __restrict__ fail, long long* __restrict__ stamps) {
    __shared__ __attribute__((aligned(16))) double colA[64];
    __shared__ __attribute__((aligned(16))) double colB[64];
    long long t0 = clock64();
    const int lane = threadIdx.x;
    const int nS = nt - k;
    const bool isR = (int)blockIdx.x >= nS;
    const int tr = isR ? (int)blockIdx.x - nS : k + (int)blockIdx.x;
    const bool isDiag = !isR && tr == k;
    const int c0 = kNB * k;
    const int ncol = min(kNB, n - c0);
    const int r = lane & 31;
    double* rowp;
    if (lane < kNB) rowp = A + (size_t)(c0 + r) * ld + c0;
    else if (!isR) rowp = A + (size_t)(kNB * tr + r) * ld + c0;
    else rowp = R + (size_t)(kNB * tr + r) * ld + c0;
    if (isDiag && lane >= kNB) rowp = R + (size_t)(c0 + r) * ld + c0;
    double m[kNB];
    {
        const double2* rp2 = reinterpret_cast<const double2*>(rowp);
#pragma unroll
        for (int c = 0; c < kNB; c += 2) {
            const double2 v = rp2[c / 2];
            m[c] = v.x;
            m[c + 1] = v.y;
        }
        const bool ident = isDiag && lane >= kNB;
#pragma unroll
        for (int c = 0; c < kNB; ++c) m[c] = ident ? (c == r ? 1.0 : 0.0) : m[c];
#pragma unroll
        for (int c = 0; c < kNB; ++c)
            if (c >= ncol) m[c] = (lane < kNB && c == r) ? 1e300 : 0.0;
    }
    // force the loads to have landed before the stamp
    double chk = 0;
#pragma unroll
    for (int c = 0; c < kNB; ++c) chk += m[c];
    if (chk == 1.2345e-300) m[0] += 1.0;
    long long t1 = clock64();
    double inv = fast_rcp(bcast_lane(m[0], 0));
#pragma unroll
    for (int j = 0; j < kNB; ++j) {
        const double mr = m[j] * inv;
        if (VARIANT != 3) {
            if (j + 1 < kNB) {
                m[j + 1] = fma(-mr, bcast_lane(m[j], j + 1), m[j + 1]);
                inv = fast_rcp(bcast_lane(m[j + 1], j + 1));
            }
        }
        if (VARIANT == 0 || VARIANT == 3) {
            if (j + 2 < kNB) {
                double* col = (j & 1) ? colB : colA;
                col[lane] = m[j];
#pragma unroll
                for (int c = j + 2; c < kNB; ++c) m[c] = fma(-mr, col[c], m[c]);
            }
        } else if (VARIANT == 4 || VARIANT == 5) {
            // DS ops pinned between ALU-permeable scheduling barriers: write, then ALL reads of the column back to back
            if (j + 2 < kNB) {
                double* col = (j & 1) ? colB : colA;
                double cv[kNB];
                __builtin_amdgcn_sched_barrier(VARIANT == 4 ? 0x7 : 0x6);
                col[lane] = m[j];
#pragma unroll
                for (int c = j + 2; c < kNB; ++c) cv[c] = col[c];
                __builtin_amdgcn_sched_barrier(VARIANT == 4 ? 0x7 : 0x6);
#pragma unroll
                for (int c = j + 2; c < kNB; ++c) m[c] = fma(-mr, cv[c], m[c]);
            }
        } else if (VARIANT == 1) {
#pragma unroll
            for (int c = j + 2; c < kNB; ++c) m[c] = fma(-mr, bcast_lane(m[c], j), m[c]);
        }
    }
    double chk2 = 0;
#pragma unroll
    for (int c = 0; c < kNB; ++c) chk2 += m[c];
    if (chk2 == 1.2345e-300) m[0] += 1.0;
    long long t2 = clock64();
    bool bad = false;
    double out[kNB];
#pragma unroll
    for (int c = 0; c < kNB; ++c) {
        const double d = bcast_lane(m[c], c);
        const bool pos = (d > 0.0) & (d < __builtin_inf());
        bad |= (c < ncol) & !pos;
        out[c] = m[c] * fast_rsqrt(((c < ncol) & pos) ? d : 1.0);
    }
    double chk3 = 0;
#pragma unroll
    for (int c = 0; c < kNB; ++c) chk3 += out[c];
    if (chk3 == 1.2345e-300) out[0] += 1.0;
    long long t3 = clock64();
    {
        double2* wp2 = reinterpret_cast<double2*>(rowp);
        if (ncol == kNB) {
            if (lane >= kNB || isDiag) {
#pragma unroll
                for (int c = 0; c < kNB; c += 2) wp2[c / 2] = make_double2(out[c], out[c + 1]);
            }
        } else {
#pragma unroll
            for (int c = 0; c < kNB; ++c)
                if (c < ncol && (lane >= kNB || (isDiag && c <= r))) rowp[c] = out[c];
        }
    }
    if (bad && isDiag && lane == 0) fail[0] = 1.0;
    long long t4 = clock64();
    if (lane == 0 && blockIdx.x < 4) {
        long long* s = stamps + blockIdx.x * 8;
        s[0] = t1 - t0; s[1] = t2 - t1; s[2] = t3 - t2; s[3] = t4 - t3; s[4] = wall_clock64();
    }
}

template <int V>
int run(const char* name, double* dA, double* dR, const std::vector<double>& hA, int ld, int n, int nt, double* fail, long long* st) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int k : {0, 9}) {
        CK(hipMemcpy(dA, hA.data(), hA.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemset(dR, 0, hA.size() * 8));
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_panel<V>, dim3(nt), dim3(64), 0, 0, dA, dR, ld, n, nt, k, fail, st);
        CK(hipMemcpy(dA, hA.data(), hA.size() * 8, hipMemcpyHostToDevice));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        const int reps = 200;
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_panel<V>, dim3(nt), dim3(64), 0, 0, dA, dR, ld, n, nt, k, fail, st);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        long long h[8]; CK(hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost));
        printf("%-34s k=%d: %.2f us/launch (200 back-to-back); diag WG clk: load %lld  loop %lld  rsqrt %lld  store %lld\n", name, k,
               ms * 1e3 / reps, h[0], h[1], h[2], h[3]);
    }
    return 0;
}

int main() {
    const int n = 600, ld = 608, nt = ld / kNB;
    std::vector<double> hA((size_t)ld * ld, 0.0);
    // SPD: diagonally dominant random symmetric
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (double)(s >> 8) / (1 << 24) - 0.5; };
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            double v = (i == j) ? 400.0 + rnd() : rnd();
            hA[(size_t)i * ld + j] = v; hA[(size_t)j * ld + i] = v;
        }
    for (int j = 0; j < n; ++j) hA[(size_t)n * ld + j] = rnd();
    double *dA, *dR, *fail; long long* st;
    CK(hipMalloc(&dA, hA.size() * 8)); CK(hipMalloc(&dR, hA.size() * 8)); CK(hipMalloc(&fail, 64)); CK(hipMalloc(&st, 1024));
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("V0 product (LDS bulk)", dA, dR, hA, ld, n, nt, fail, st);
        run<1>("V1 readlane bulk (symmetry)", dA, dR, hA, ld, n, nt, fail, st);
        run<2>("V2 chain only", dA, dR, hA, ld, n, nt, fail, st);
        run<3>("V3 bulk only", dA, dR, hA, ld, n, nt, fail, st);
        run<4>("V4 DS pinned, ALU(0x7) may cross", dA, dR, hA, ld, n, nt, fail, st);
        run<5>("V5 DS pinned, VALU|SALU may cross", dA, dR, hA, ld, n, nt, fail, st);
    }
    return 0;
}
