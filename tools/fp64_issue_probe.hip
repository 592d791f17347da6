// How fast does ONE SIMD issue FP64 FMAs - from one wave, and from two waves sharing it?  (workgroup of 64 x nw threads:
// waves 0..3 land on the four SIMDs of a CU, waves 4..7 on the same four again)
//   hipcc --offload-arch=gfx950 -O3 tools/fp64_issue_probe.hip -o /tmp/fip && /tmp/fip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS>
__global__ void k(double* o, long long* clk, int iters) {
    double a[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) a[c] = 1.0 + threadIdx.x * 1e-9 + c;
    const double m = 1.0000001, b = 1e-9;
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) a[c] = __builtin_fma(a[c], m, b);
    }
    const long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += a[c];
    o[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) clk[threadIdx.x >> 6] = t1 - t0;
}
template <int CHAINS>
void run(int nw, double* o, long long* clk) {
    const int iters = 2000;
    hipLaunchKernelGGL(k<CHAINS>, dim3(1), dim3(64 * nw), 0, 0, o, clk, iters);
    long long h[16];
    (void)hipMemcpy(h, clk, sizeof(long long) * nw, hipMemcpyDeviceToHost);
    long long mx = 0;
    for (int w = 0; w < nw; ++w) mx = h[w] > mx ? h[w] : mx;
    printf("%d wave(s) in the workgroup, %2d independent chains each: %6.2f clk per FMA and wave, %6.2f clk per FMA on the busiest SIMD\n",
           nw, CHAINS, (double)mx / (iters * CHAINS), (double)mx / (iters * CHAINS) / ((nw + 3) / 4));
}
int main() {
    double* o; long long* clk;
    (void)hipMalloc(&o, 8 * 1024); (void)hipMalloc(&clk, 8 * 16);
    for (int nw : {1, 4, 8, 16}) { run<1>(nw, o, clk); run<4>(nw, o, clk); run<8>(nw, o, clk); }
    return 0;
}
