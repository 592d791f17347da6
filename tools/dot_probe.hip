// v_alignbyte_b32 / v_dot4_u32_u8 / v_dot2_u32_u16 semantics on gfx950, and the v_ashr_pk_u8_i32 miscompile of ROCm 7.2:
//   hipcc --offload-arch=gfx950 -O3 tools/dot_probe.hip -o /tmp/dp && /tmp/dp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__global__ void k(const unsigned* p, unsigned* o) {
    const unsigned a = p[0], b = p[1];
    o[0] = __builtin_amdgcn_alignbyte(b, a, 1);
    o[1] = __builtin_amdgcn_alignbyte(b, a, 2);
    o[2] = __builtin_amdgcn_alignbyte(b, a, 3);
    o[3] = __builtin_amdgcn_alignbyte(b, a, p[2]);      // shift from a register (3)
    o[4] = __builtin_amdgcn_udot4(a, 0x04030201u, 5u, false);
    o[5] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, 0x00030002u), 7u, false);
    o[6] = __builtin_amdgcn_perm(b, a, 0x0c050c02u);
}
// the horizontal pass of k_blur on one 12-byte window
__global__ void k2(const unsigned* p, int* o) {
    const unsigned w0 = p[threadIdx.x * 3], w1 = p[threadIdx.x * 3 + 1], w2 = p[threadIdx.x * 3 + 2];
    constexpr unsigned K0 = 18u | 34u << 8 | 49u << 16 | 55u << 24, K1 = 49u | 34u << 8 | 18u << 16;
    int h[4];
    h[0] = (int)__builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 1), K1,
                                       __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 1), K0, 0u, false), false);
    h[1] = (int)__builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 2), K1,
                                       __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 2), K0, 0u, false), false);
    h[2] = (int)__builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 3), K1,
                                       __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 3), K0, 0u, false), false);
    h[3] = (int)__builtin_amdgcn_udot4(w2, K1, __builtin_amdgcn_udot4(w1, K0, 0u, false), false);
    for (int q = 0; q < 4; ++q) o[threadIdx.x * 4 + q] = h[q];
}
// four signed sums -> four saturated bytes, the way k_blur's column pass used to write it
__global__ void k3(const int* p, unsigned* o) {
    unsigned out = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int sacc = p[threadIdx.x * 4 + q] * 257 + p[256 + q];
        const int v = min(max((sacc + (1 << 15)) >> 16, 0), 255);
        out |= (unsigned)v << (8 * q);
    }
    o[threadIdx.x] = out;
}
int main() {
    unsigned h[3] = {0x44332211u, 0x88776655u, 3u}, *d, *o, r[7];
    (void)hipMalloc(&d, 12); (void)hipMalloc(&o, 28);
    (void)hipMemcpy(d, h, 12, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d, o);
    (void)hipMemcpy(r, o, 28, hipMemcpyDeviceToHost);
    printf("alignbyte 1 %08x (55443322)  2 %08x (66554433)  3 %08x (77665544)  reg3 %08x\n", r[0], r[1], r[2], r[3]);
    printf("dot4 %u (0x11*1+0x22*2+0x33*3+0x44*4+5 = %u)  dot2 %u (0x2211*2+0x4433*3+7 = %u)  perm %08x (00660033)\n", r[4],
           0x11 * 1 + 0x22 * 2 + 0x33 * 3 + 0x44 * 4 + 5, r[5], 0x2211 * 2 + 0x4433 * 3 + 7, r[6]);
    unsigned char win[64 * 12];
    for (int i = 0; i < 64 * 12; ++i) win[i] = (unsigned char)((i * 37 + (i >> 3) * 11) & 0xff);
    unsigned* dw; int* dh; int hh[256];
    (void)hipMalloc(&dw, sizeof(win)); (void)hipMalloc(&dh, sizeof(hh));
    (void)hipMemcpy(dw, win, sizeof(win), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, 0, dw, dh);
    (void)hipMemcpy(hh, dh, sizeof(hh), hipMemcpyDeviceToHost);
    const int taps[7] = {18, 34, 49, 55, 49, 34, 18};
    int bad[4] = {0, 0, 0, 0};
    for (int t = 0; t < 64; ++t)
        for (int j = 0; j < 4; ++j) {
            int e = 0;
            for (int k = 0; k < 7; ++k) e += taps[k] * win[t * 12 + j + 1 + k];
            if (e != hh[t * 4 + j]) ++bad[j];
        }
    printf("blur row pass: mismatches per pixel of the group %d %d %d %d (of 64)\n", bad[0], bad[1], bad[2], bad[3]);
    int hs[260], *ds; unsigned *dq, hq[64];
    for (int i = 0; i < 256; ++i) hs[i] = (i * 7919) % 65536;
    for (int i = 256; i < 260; ++i) hs[i] = 0;
    (void)hipMalloc(&ds, sizeof(hs)); (void)hipMalloc(&dq, sizeof(hq));
    (void)hipMemcpy(ds, hs, sizeof(hs), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k3, dim3(1), dim3(64), 0, 0, ds, dq);
    (void)hipMemcpy(hq, dq, sizeof(hq), hipMemcpyDeviceToHost);
    int badq[4] = {0, 0, 0, 0};
    for (int t = 0; t < 64; ++t)
        for (int q = 0; q < 4; ++q) {
            const int sacc = hs[t * 4 + q] * 257;
            int v = (sacc + (1 << 15)) >> 16; v = v < 0 ? 0 : (v > 255 ? 255 : v);
            if (((hq[t] >> (8 * q)) & 0xff) != (unsigned)v) ++badq[q];
        }
    printf("min(max(x >> 16, 0), 255) of four sums packed into a dword: wrong bytes per position %d %d %d %d (of 64)\n", badq[0],
           badq[1], badq[2], badq[3]);
    return 0;
}
