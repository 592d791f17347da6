// Micro-benchmarks used to understand the latency floor of the (tiny, launch-bound) BA kernels on MI355X.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o /tmp/microbench && /tmp/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty() {}
__global__ void k_fma_chain(double* out, int n) {
    double a = out[threadIdx.x], b = 1.0000001;
    long long t0 = clock64();
    long long w0 = wall_clock64();
    for (int i = 0; i < n; ++i) a = fma(a, b, 1e-9);
    long long t1 = clock64();
    long long w1 = wall_clock64();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) { out[64] = (double)(t1 - t0); out[65] = (double)(w1 - w0); }
}
__global__ void k_touch(double* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0; }

int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double* d; CK(hipMalloc(&d, 1 << 24));
    CK(hipMemset(d, 0, 1 << 24));
    float ms;
    // (a) empty kernel: back-to-back launches
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("empty kernel, 1000 back-to-back launches: %.3f us each\n", ms);
    }
    // (b) event pair around ONE empty kernel
    double tot = 0;
    for (int i = 0; i < 200; ++i) {
        CK(hipEventRecord(e0, st)); hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms;
    }
    printf("event pair around one empty kernel: %.3f us\n", tot / 200 * 1e3);
    // (c) dependent kernels touching 1 MB
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(k_touch, dim3(512), dim3(256), 0, st, d, 131072);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("1 MB rmw kernel, 1000 dependent launches: %.3f us each\n", ms);
    // (d) shader clock under a sparse load: one wave, dependent fp64 FMA chain
    for (int n : {20000, 200000, 2000000}) {
        CK(hipEventRecord(e0, st)); hipLaunchKernelGGL(k_fma_chain, dim3(1), dim3(64), 0, st, d, n);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        double h[66]; CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
        printf("fma chain n=%d: %.1f us, %.2f shader-clk/fma, %.2f ns/fma, shader clk ~ %.0f MHz (wall_clock64 %.0f ticks)\n", n,
               ms * 1e3, h[64] / n, ms * 1e6 / n, h[64] / (ms * 1e3), h[65]);
    }
    // (e) sync latency: tiny D2H after a kernel
    double hv; tot = 0;
    for (int i = 0; i < 200; ++i) {
        auto t0 = std::chrono::high_resolution_clock::now();
        hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
        CK(hipMemcpyAsync(&hv, d, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
        tot += std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
    }
    printf("launch + 8-byte D2H + stream sync (host wall): %.2f us\n", tot / 200);
    return 0;
}
