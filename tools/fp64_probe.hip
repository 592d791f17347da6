// Latency / issue-rate probes for the single-wave FP64 pivot chain of k_chol_panel (gfx950).
//   hipcc --offload-arch=gfx950 -O3 tools/fp64_probe.hip -o /tmp/fp64_probe && /tmp/fp64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int N = 4096;

template <int W>
__global__ void k_fma_indep(double* out) {  // W independent chains per lane
    double a[W];
    for (int w = 0; w < W; ++w) a[w] = out[threadIdx.x + 64 * w];
    const double b = out[1000];
    long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int w = 0; w < W; ++w) a[w] = fma(a[w], b, 1e-9);
    }
    long long t1 = clock64();
    double s = 0;
    for (int w = 0; w < W; ++w) s += a[w];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) out[2000] = (double)(t1 - t0) / N;
}
__global__ void k_mul_chain(double* out) {
    double a = out[threadIdx.x];
    const double b = out[1000];
    long long t0 = clock64();
#pragma unroll 8
    for (int i = 0; i < N; ++i) a = a * b;
    long long t1 = clock64();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) out[2000] = (double)(t1 - t0) / N;
}
__global__ void k_f32_chain(double* out) {
    float a = (float)out[threadIdx.x];
    const float b = (float)out[1000];
    long long t0 = clock64();
#pragma unroll 8
    for (int i = 0; i < N; ++i) a = fmaf(a, b, 1e-9f);
    long long t1 = clock64();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) out[2000] = (double)(t1 - t0) / N;
}
__global__ void k_ldexp_chain(double* out) {
    double a = out[threadIdx.x];
    const int e = (int)out[1001];
    long long t0 = clock64();
#pragma unroll 8
    for (int i = 0; i < N; ++i) a = ldexp(a, e + (i & 1));
    long long t1 = clock64();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) out[2000] = (double)(t1 - t0) / N;
}
__global__ void k_rcp_chain(double* out) {
    double a = out[threadIdx.x] + 1.5;
    long long t0 = clock64();
#pragma unroll 8
    for (int i = 0; i < N; ++i) a = __builtin_amdgcn_rcp(a);
    long long t1 = clock64();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) out[2000] = (double)(t1 - t0) / N;
}
__global__ void k_readlane_fma_chain(double* out) {  // fma -> readlane (lane 5) -> fma ...
    double a = out[threadIdx.x] + 1.0;
    long long t0 = clock64();
#pragma unroll 8
    for (int i = 0; i < N; ++i) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(a), 5);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(a), 5);
        a = fma(a, __hiloint2double(hi, lo), 1e-9);
        a = a - floor(a) + 1.0;  // keep bounded (2 more dependent ops)
    }
    long long t1 = clock64();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) out[2000] = (double)(t1 - t0) / N;
}
__global__ void k_floor_chain(double* out) {  // the bounding ops alone
    double a = out[threadIdx.x] + 1.0;
    const double b = out[1000] + 1.0;
    long long t0 = clock64();
#pragma unroll 8
    for (int i = 0; i < N; ++i) {
        a = fma(a, b, 1e-9);
        a = a - floor(a) + 1.0;
    }
    long long t1 = clock64();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) out[2000] = (double)(t1 - t0) / N;
}
__global__ void k_scalar_chain(double* out) {  // readlane -> scalar int exponent strip -> fma
    double a = out[threadIdx.x] + 1.0;
    long long t0 = clock64();
#pragma unroll 8
    for (int i = 0; i < N; ++i) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(a), 5);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(a), 5);
        const double f = __hiloint2double((hi & 0x000fffff) | 0x3ff00000, lo);
        a = fma(a, f, 1e-9);
        a = a - floor(a) + 1.0;
    }
    long long t1 = clock64();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) out[2000] = (double)(t1 - t0) / N;
}
__global__ void k_lds_bcast(double* out) {  // write column, 16 broadcast reads + 16 independent fmas
    __shared__ double col[64];
    double m[16];
    for (int c = 0; c < 16; ++c) m[c] = out[threadIdx.x + c];
    double a = out[threadIdx.x];
    long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N; ++i) {
        col[threadIdx.x] = a;
#pragma unroll
        for (int c = 0; c < 16; ++c) m[c] = fma(-a, col[c + 8], m[c]);
        a = m[i & 15] * 0.5;
    }
    long long t1 = clock64();
    double s = 0;
    for (int c = 0; c < 16; ++c) s += m[c];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) out[2000] = (double)(t1 - t0) / N;
}
__global__ void k_readlane_bulk(double* out) {  // same with 16 readlane pairs instead of LDS
    double m[16];
    for (int c = 0; c < 16; ++c) m[c] = out[threadIdx.x + c];
    double a = out[threadIdx.x];
    long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int lo = __builtin_amdgcn_readlane(__double2loint(a), c + 8);
            const int hi = __builtin_amdgcn_readlane(__double2hiint(a), c + 8);
            m[c] = fma(-a, __hiloint2double(hi, lo), m[c]);
        }
        a = m[i & 15] * 0.5;
    }
    long long t1 = clock64();
    double s = 0;
    for (int c = 0; c < 16; ++c) s += m[c];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) out[2000] = (double)(t1 - t0) / N;
}

int main() {
    double* d; CK(hipMalloc(&d, 1 << 20));
    double h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = 1.0 + 1e-7 * i;
    h[1000] = 1.0000001; h[1001] = 0.0;
#define RUN(name, ...) do { CK(hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice)); hipLaunchKernelGGL(__VA_ARGS__, dim3(1), dim3(64), 0, 0, d); \
    CK(hipDeviceSynchronize()); double v; CK(hipMemcpy(&v, d + 2000, 8, hipMemcpyDeviceToHost)); printf("%-44s %8.1f clk / iteration\n", name, v); } while (0)
    for (int rep = 0; rep < 2; ++rep) {
        RUN("fp64 fma, 1 chain", k_fma_indep<1>);
        RUN("fp64 fma, 2 independent chains", k_fma_indep<2>);
        RUN("fp64 fma, 4 independent chains", k_fma_indep<4>);
        RUN("fp64 fma, 8 independent chains", k_fma_indep<8>);
        RUN("fp64 fma, 16 independent chains", k_fma_indep<16>);
        RUN("fp64 mul chain", k_mul_chain);
        RUN("fp32 fma chain", k_f32_chain);
        RUN("fp64 ldexp chain", k_ldexp_chain);
        RUN("fp64 rcp chain", k_rcp_chain);
        RUN("fma + 2 bounding ops (baseline for next two)", k_floor_chain);
        RUN("readlane x2 -> fma + 2 bounding ops", k_readlane_fma_chain);
        RUN("readlane x2 -> s_and_or -> fma + 2 bounding", k_scalar_chain);
        RUN("LDS bcast: write + 16 reads + 16 fma + mul", k_lds_bcast);
        RUN("readlane bulk: 32 readlane + 16 fma + mul", k_readlane_bulk);
    }
    return 0;
}
