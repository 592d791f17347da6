// Issue rate of the packed min/max flavours on gfx950 (which one should carry FAST's ring extrema?):
//   hipcc --offload-arch=gfx950 -O3 tools/pkminmax_probe.hip -o /tmp/pkprobe && /tmp/pkprobe
// 8 independent accumulator chains per lane, 4 waves per SIMD, all CUs: time per instruction per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef short short2v __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, int n) {
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = out[(threadIdx.x + 64 * i) & 1023];
    uint32_t b = out[threadIdx.x & 63], c = out[(threadIdx.x + 7) & 63];
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (MODE == 1) asm volatile("v_pk_minimum3_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (MODE == 2) asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (MODE == 3) asm volatile("v_min3_i16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (MODE == 4) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (MODE == 5) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (MODE == 6) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    uint32_t* d; CK(hipMalloc(&d, 1 << 24)); CK(hipMemset(d, 0x11, 1 << 24));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int n = 20000, blocks = 256 * 4;   // 4 workgroups of 4 waves per CU = 4 waves per SIMD
    const char* names[] = {"v_pk_min_i16", "v_pk_minimum3_f16", "v_pk_min_f16", "v_min3_i16", "v_pk_maximum3_f16", "v_perm_b32", "v_pk_max_i16"};
    for (int mode = 0; mode < 7; ++mode) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, n); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, n); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, n); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d, n); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, d, n); break;
                case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, d, n); break;
                case 6: hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(256), 0, 0, d, n); break;
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        // per SIMD: 4 waves x n x 8 instructions
        const double instr_per_simd = 4.0 * n * 8;
        printf("%-20s %8.3f ms   %.2f ns per wave-instruction per SIMD  (= %.2f cycles at 2.4 GHz)\n", names[mode], best,
               best * 1e6 / instr_per_simd, best * 1e6 / instr_per_simd * 2.4);
    }
    return 0;
}
