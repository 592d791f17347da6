// Tile hand-off between two workgroups of ONE XCD through its L2 (round 6): does a hand-off whose loads only bypass the vector L1
// ("sc0": workgroup scope) and whose stores are acknowledged by the L2 beat k_chol_tiles' agent-scope protocol ("sc1": stores
// written through to memory, loads served from there) - and does it deliver the payload?  Same payload and checks as
// tools/xcd_probe.hip (64 lanes x 6 x 16 B, the round number in every word).  Partners: workgroups 0 and 8 (same XCD), 0 and 16,
// and 0 and 1 (different XCDs: the L2-scope variants must FAIL there - stale data or a spin time-out - or the probe proves nothing).
//   S_AGENT   stores sc1, loads sc1                      [k_chol_tiles today]
//   S_L2      stores plain, loads sc0
//   S_L2B     stores sc0, loads sc0
//   S_MIXED   stores sc1 (written through: any XCD may read), loads sc0 (the same-XCD consumer's share of the saving)
//   S_INV     stores plain, loads plain behind "buffer_inv sc0" (the vector L1 invalidated: the load is served by the L2)
//   S_INV_WT  stores sc1, loads plain behind "buffer_inv sc0"
//   hipcc --offload-arch=gfx950 -O3 tools/l2scope_probe.hip -o /tmp/l2scope_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ROUNDS = 400;
typedef double d2_t __attribute__((ext_vector_type(2)));
enum { S_AGENT = 0, S_L2 = 1, S_L2B = 2, S_MIXED = 3, S_INV = 4, S_INV_WT = 5 };

template <int S> __device__ inline void st(double* p, d2_t v) {
    if (S == S_AGENT || S == S_MIXED) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
    if (S == S_L2 || S == S_INV) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
    if (S == S_INV_WT) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
    if (S == S_L2B) asm volatile("global_store_dwordx4 %0, %1, off sc0\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
template <int S> __device__ inline d2_t ld(const double* p) {
    d2_t v;
    if (S == S_AGENT) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    else if (S == S_INV || S == S_INV_WT) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int S> __device__ inline void stf(unsigned* p, unsigned v) {
    if (S == S_AGENT || S == S_MIXED) asm volatile("global_store_dword %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
    if (S == S_L2 || S == S_INV) asm volatile("global_store_dword %0, %1, off" : : "v"(p), "v"(v) : "memory");
    if (S == S_INV_WT) asm volatile("global_store_dword %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
    if (S == S_L2B) asm volatile("global_store_dword %0, %1, off sc0" : : "v"(p), "v"(v) : "memory");
}
template <int S> __device__ inline unsigned ldf(const unsigned* p) {
    unsigned v;
    if (S == S_AGENT) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (S == S_INV || S == S_INV_WT) asm volatile("buffer_inv sc0\n\tglobal_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int S>
__global__ __launch_bounds__(64) void k_pingpong(int a, int b, double* tile, unsigned* flags, long long* out) {
    const int me = (int)blockIdx.x == a ? 0 : ((int)blockIdx.x == b ? 1 : -1);
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0 && blockIdx.x < 32) out[8 + blockIdx.x] = xcc & 0xf;
    if (me < 0) return;
    const int ln = threadIdx.x;
    double* mine = tile + (size_t)me * 1024 + ln * 2;
    const double* theirs = tile + (size_t)(1 - me) * 1024 + ln * 2;
    long long errors = 0, spins = 0;
    const long long t0 = wall_clock64();
    for (int r = 0; r < ROUNDS; ++r) {
        const unsigned s = 2 * r + me + 1;      // partner 0 starts (s = 1), then 1 (s = 2), ...
        if (s > 1) {
            const double want = (double)(s - 1);
            const long long w0 = wall_clock64();
            while (__builtin_amdgcn_readfirstlane(ldf<S>(flags)) != s - 1) {
                __builtin_amdgcn_s_sleep(1);
                ++spins;
                if (wall_clock64() - w0 > 1000000ll) { if (ln == 0) { out[2 + me] = -1; out[me] = wall_clock64() - t0; } return; }
            }
            d2_t x[6];
            if (S == S_INV || S == S_INV_WT) asm volatile("buffer_inv sc0" ::: "memory");
#pragma unroll
            for (int q = 0; q < 6; ++q) x[q] = ld<S>(theirs + 128 * q);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]) : : "memory");
#pragma unroll
            for (int q = 0; q < 6; ++q) errors += (x[q].x != want) + (x[q].y != want + 0.5);
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) st<S>(mine + 128 * q, d2_t{(double)s, (double)s + 0.5});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (ln == 0) stf<S>(flags, s);
    }
    const long long t1 = wall_clock64();
    if (ln == 0) { out[me] = t1 - t0; out[2 + me] = errors; out[4 + me] = spins; }
}

template <int S> int run(const char* name, double* tile, unsigned* flags, long long* out) {
    for (int b : {8, 16, 1}) {
        CK(hipMemset(tile, 0, 1 << 16)); CK(hipMemset(flags, 0, 256)); CK(hipMemset(out, 0, 1024));
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL((k_pingpong<S>), dim3(128), dim3(64), 0, 0, 0, b, tile, flags, out);
        CK(hipDeviceSynchronize());
        long long h[40]; CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
        printf("%-44s WG 0 (xcc %lld) <-> WG %2d (xcc %lld): %6.3f us per hand-off, payload errors %lld / %lld, polls per hand-off %.1f%s\n", name,
               h[8], b, h[8 + b], (double)h[1] / 100.0 / (2 * ROUNDS), h[2], h[3], (double)(h[4] + h[5]) / (2 * ROUNDS),
               (h[2] < 0 || h[3] < 0) ? "  [SPIN TIMEOUT]" : "");
    }
    return 0;
}

int main() {
    double* tile; unsigned* flags; long long* out;
    CK(hipMalloc(&tile, 1 << 16)); CK(hipMalloc(&flags, 256)); CK(hipMalloc(&out, 1024));
    for (int rep = 0; rep < 1; ++rep) {
        if (run<S_AGENT>("stores sc1, loads sc1  [k_chol_tiles]", tile, flags, out)) return 1;
        if (run<S_L2>("stores plain, loads sc0", tile, flags, out)) return 1;
        if (run<S_L2B>("stores sc0, loads sc0", tile, flags, out)) return 1;
        if (run<S_MIXED>("stores sc1, loads sc0", tile, flags, out)) return 1;
        if (run<S_INV>("stores plain, buffer_inv sc0 + plain loads", tile, flags, out)) return 1;
        if (run<S_INV_WT>("stores sc1, buffer_inv sc0 + plain loads", tile, flags, out)) return 1;
    }
    return 0;
}
