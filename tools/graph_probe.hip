// Is a hipGraph launch of a kernel chain cheaper than the same chain launched kernel by kernel?  (window batches of the BA
// are bound by the launch rate, docs/history/DESIGN_rounds_1-5.md 4.1.2)
//   hipcc --offload-arch=gfx950 -O3 tools/graph_probe.hip -o tools/bin/graph_probe && tools/bin/graph_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_touch(double* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0; }
int main() {
    const int kChain = 56, kStreams = 16, kReps = 50;
    std::vector<hipStream_t> st(kStreams);
    std::vector<double*> d(kStreams);
    for (int s = 0; s < kStreams; ++s) { CK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking)); CK(hipMalloc(&d[s], 1 << 16)); }
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    // direct launches, all streams
    for (int w = 0; w < 2; ++w) {
        auto t0 = now();
        for (int r = 0; r < kReps; ++r)
            for (int s = 0; s < kStreams; ++s)
                for (int k = 0; k < kChain; ++k) hipLaunchKernelGGL(k_touch, dim3(8), dim3(256), 0, st[s], d[s], 2048);
        auto t1 = now();
        for (int s = 0; s < kStreams; ++s) CK(hipStreamSynchronize(st[s]));
        auto t2 = now();
        if (w) printf("direct : enqueue %.2f us per kernel, total %.2f us per kernel (%d streams x %d kernels x %d)\n",
                      us(t0, t1) / (kReps * kStreams * kChain), us(t0, t2) / (kReps * kStreams * kChain), kStreams, kChain, kReps);
    }
    // one graph per stream
    std::vector<hipGraphExec_t> ge(kStreams);
    for (int s = 0; s < kStreams; ++s) {
        hipGraph_t g;
        CK(hipStreamBeginCapture(st[s], hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < kChain; ++k) hipLaunchKernelGGL(k_touch, dim3(8), dim3(256), 0, st[s], d[s], 2048);
        CK(hipStreamEndCapture(st[s], &g));
        CK(hipGraphInstantiate(&ge[s], g, nullptr, nullptr, 0));
    }
    for (int w = 0; w < 2; ++w) {
        auto t0 = now();
        for (int r = 0; r < kReps; ++r)
            for (int s = 0; s < kStreams; ++s) CK(hipGraphLaunch(ge[s], st[s]));
        auto t1 = now();
        for (int s = 0; s < kStreams; ++s) CK(hipStreamSynchronize(st[s]));
        auto t2 = now();
        if (w) printf("graph  : enqueue %.2f us per kernel, total %.2f us per kernel\n", us(t0, t1) / (kReps * kStreams * kChain),
                      us(t0, t2) / (kReps * kStreams * kChain));
    }
    // one stream only: is the dependent-kernel boundary itself shorter inside a graph?
    for (int w = 0; w < 2; ++w) {
        auto t0 = now();
        for (int r = 0; r < kReps; ++r)
            for (int k = 0; k < kChain; ++k) hipLaunchKernelGGL(k_touch, dim3(8), dim3(256), 0, st[0], d[0], 2048);
        CK(hipStreamSynchronize(st[0]));
        auto t2 = now();
        if (w) printf("direct, 1 stream: %.2f us per kernel\n", us(t0, t2) / (kReps * kChain));
    }
    for (int w = 0; w < 2; ++w) {
        auto t0 = now();
        for (int r = 0; r < kReps; ++r) CK(hipGraphLaunch(ge[0], st[0]));
        CK(hipStreamSynchronize(st[0]));
        auto t2 = now();
        if (w) printf("graph,  1 stream: %.2f us per kernel\n", us(t0, t2) / (kReps * kChain));
    }
    return 0;
}
