// accuracy of v_rcp_f64 / v_rsq_f64 seeds with 0, 1, 2 Newton steps (decides the pivot chain length of k_chol_panel)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d = x[i];
    double r0 = __builtin_amdgcn_rcp(d);
    double r1 = fma(fma(-d, r0, 1.0), r0, r0);
    double r2 = fma(fma(-d, r1, 1.0), r1, r1);
    double s0 = __builtin_amdgcn_rsq(d);
    double s1 = s0 * fma(-0.5 * d * s0, s0, 1.5);
    double s2 = s1 * fma(-0.5 * d * s1, s1, 1.5);
    out[6 * i + 0] = r0; out[6 * i + 1] = r1; out[6 * i + 2] = r2;
    out[6 * i + 3] = s0; out[6 * i + 4] = s1; out[6 * i + 5] = s2;
}
int main() {
    const int n = 1 << 16;
    std::vector<double> h(n), o(6 * n);
    for (int i = 0; i < n; ++i) h[i] = std::exp((i % 4096) * 0.01 - 10.0) * (1.0 + (i * 2654435761u % 1000003) / 1000003.0);
    double *d, *r;
    hipMalloc(&d, n * 8); hipMalloc(&r, 6 * n * 8);
    hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, r, n);
    hipMemcpy(o.data(), r, 6 * n * 8, hipMemcpyDeviceToHost);
    double e[6] = {0};
    for (int i = 0; i < n; ++i) {
        const long double tr = 1.0L / h[i], ts = 1.0L / sqrtl(h[i]);
        for (int q = 0; q < 3; ++q) e[q] = fmax(e[q], (double)fabsl((o[6 * i + q] - tr) / tr));
        for (int q = 3; q < 6; ++q) e[q] = fmax(e[q], (double)fabsl((o[6 * i + q] - ts) / ts));
    }
    printf("max rel err  rcp: seed %.3e, 1 NR %.3e, 2 NR %.3e | rsq: seed %.3e, 1 NR %.3e, 2 NR %.3e (eps 1.1e-16)\n", e[0], e[1], e[2], e[3], e[4], e[5]);
    return 0;
}
