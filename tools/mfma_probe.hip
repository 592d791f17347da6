// Operand / result layout of v_mfma_f64_16x16x4_f64 on gfx950, determined empirically.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o tools/bin/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double d4_t __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, double* D, long long* clk) {
    // hypothesis: A[i][k]: lane = i + 16 k ; B[k][j]: lane = j + 16 k ; D[i][j]: lane = j + 16 * (i / 4), reg = i % 4
    const int l = threadIdx.x;
    const double a = A[(l % 16) * 4 + l / 16];   // A is 16x4 row-major
    const double b = B[(l / 16) * 16 + l % 16];  // B is 4x16 row-major
    d4_t c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) D[l * 4 + v] = c[v];
    // issue-rate: 64 dependent-free MFMAs on 4 accumulators
    d4_t c0 = c, c1 = c, c2 = c, c3 = c;
    long long t0 = clock64();
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    long long t1 = clock64();
    D[256 + l] = c0[0] + c1[1] + c2[2] + c3[3];
    if (l == 0) clk[0] = t1 - t0;
}
int main() {
    double hA[64], hB[64], hD[512];
    for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 4; ++kk) hA[i * 4 + kk] = 1.0 + i + 100.0 * kk;
    for (int kk = 0; kk < 4; ++kk) for (int j = 0; j < 16; ++j) hB[kk * 16 + j] = 0.5 + j * 0.25 + 7.0 * kk;
    double *dA, *dB, *dD; long long* dc;
    CK(hipMalloc(&dA, sizeof(hA))); CK(hipMalloc(&dB, sizeof(hB))); CK(hipMalloc(&dD, sizeof(hD))); CK(hipMalloc(&dc, 64));
    CK(hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, dc);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost));
    long long c; CK(hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost));
    // reference
    double ref[16][16];
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int kk = 0; kk < 4; ++kk) s += hA[i * 4 + kk] * hB[kk * 16 + j]; ref[i][j] = s; }
    // test hypotheses for D layout
    int okA = 1, okB = 1;
    for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
        const double got = hD[l * 4 + v];
        if (got != ref[4 * (l / 16) + v][l % 16]) okA = 0;   // i = 4*(lane/16) + v
        if (got != ref[(l / 16) + 4 * v][l % 16]) okB = 0;   // i = lane/16 + 4*v
    }
    printf("D layout i = 4*(lane/16)+v, j = lane%%16 : %s\n", okA ? "MATCH" : "no");
    printf("D layout i = lane/16+4*v,   j = lane%%16 : %s\n", okB ? "MATCH" : "no");
    if (!okA && !okB) { for (int l = 0; l < 64; l += 9) printf("lane %d: %g %g %g %g\n", l, hD[l*4], hD[l*4+1], hD[l*4+2], hD[l*4+3]); printf("ref[0][0..3] %g %g %g %g ref[1][0] %g ref[4][0] %g\n", ref[0][0], ref[0][1], ref[0][2], ref[0][3], ref[1][0], ref[4][0]); }
    printf("64 independent v_mfma_f64_16x16x4 (4 accumulators): %.1f clk each\n", (double)c / 64.0);
    return 0;
}
