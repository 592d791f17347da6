// Tile hand-off between two workgroups on MI355X, by protocol and by placement (same XCD: workgroup ids 0 and 8; another
// XCD: 0 and 1).  The payload (64 lanes x 6 x 16 B = one slab of k_chol_tiles) carries the round number and the consumer
// checks it: a protocol that reads stale or torn lines shows up as errors, not as a fast number.
//   P_FLAG    payload stores (sc1) -> s_waitcnt vmcnt(0) -> flag store (sc1) | flag poll (sc1) -> payload loads (sc1)   [k_chol_tiles]
//   P_NOWAIT  as P_FLAG without the s_waitcnt between payload and flag (NOT safe: counts how often the payload is late)
//   P_TAGGED  no flag: every 128-byte line of the payload carries the round number in its last 8 bytes, written by the same
//             store instruction as the line's data; the consumer polls the payload itself until all tags have arrived
//   P_FLAGONLY  no payload
//   P_STRIDED   P_FLAG with the access pattern k_chol_tiles has today: lane = tile row (row stride 5 KB), lanes 32..63 store
//               8 x 16 B each (4 to M, 4 to MR: 64 B of a row per array), the consumer's lane = (row, half) loads 6 x 16 B
// store flavours for the flag: sc1 / sc0 sc1 / atomic swap without return
//   hipcc --offload-arch=gfx950 -O3 tools/xcd_probe.hip -o tools/bin/xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ROUNDS = 400;
typedef double d2_t __attribute__((ext_vector_type(2)));
enum { P_FLAG = 0, P_NOWAIT = 1, P_TAGGED = 2, P_FLAGONLY = 3, P_STRIDED = 4, P_STRIDED_NOWAIT = 5 };

__device__ inline void st(double* p, d2_t v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory"); }
__device__ inline d2_t ld(const double* p) {
    d2_t v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int F> __device__ inline void stf(unsigned* p, unsigned v) {
    if (F == 0) asm volatile("global_store_dword %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
    if (F == 1) asm volatile("global_store_dword %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
    if (F == 2) asm volatile("global_atomic_swap %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ inline unsigned ldf(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// one wave per partner.  Payload layout: store q of lane l covers bytes [1024 q + 16 l, +16): one instruction = 8 lines of 128 B
template <int P, int F>
__global__ __launch_bounds__(64) void k_pingpong(int a, int b, double* tile, double* big, unsigned* flags, long long* out) {
    const int me = (int)blockIdx.x == a ? 0 : ((int)blockIdx.x == b ? 1 : -1);
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0 && blockIdx.x < 32) out[8 + blockIdx.x] = xcc & 0xf;
    if (me < 0) return;
    const int ln = threadIdx.x;
    double* mine = tile + (size_t)me * 1024 + ln * 2;
    const double* theirs = tile + (size_t)(1 - me) * 1024 + ln * 2;
    const bool tagl = (ln & 7) == 7;     // this lane's .y is the last 8 bytes of a 128-byte line
    long long errors = 0, spins = 0;
    const long long t0 = wall_clock64();
    for (int r = 0; r < ROUNDS; ++r) {
        const unsigned s = 2 * r + me + 1;      // partner 0 starts (s = 1), then 1 (s = 2), ...
        if (s > 1) {
            const double want = (double)(s - 1);
            const long long w0 = wall_clock64();
            d2_t x[6];
            if (P == P_TAGGED) {
                for (;;) {
#pragma unroll
                    for (int q = 0; q < 6; ++q) x[q] = ld(theirs + 128 * q);
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]) : : "memory");
                    bool ok = true;
#pragma unroll
                    for (int q = 0; q < 6; ++q) ok = ok && (!tagl || x[q].y == want);
                    if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
                    ++spins;
                    if (wall_clock64() - w0 > 1000000ll) { if (ln == 0) out[2 + me] = -1; return; }
                }
#pragma unroll
                for (int q = 0; q < 6; ++q) errors += (x[q].x != want) + (!tagl && x[q].y != want + 0.5);
            } else {
                while (__builtin_amdgcn_readfirstlane(ldf(flags)) != s - 1) {
                    __builtin_amdgcn_s_sleep(1);
                    ++spins;
                    if (wall_clock64() - w0 > 1000000ll) { if (ln == 0) out[2 + me] = -1; return; }
                }
                if (P == P_STRIDED || P == P_STRIDED_NOWAIT) {
                    const double* rowM = big + (size_t)(1 - me) * (1 << 20) / 8 + (size_t)(ln >> 1) * 640 + 4 * (ln & 1);
                    const double* rowR = rowM + (1 << 19) / 8;
                    x[0] = ld(rowM); x[1] = ld(rowM + 2); x[2] = ld(rowR); x[3] = ld(rowR + 2); x[4] = ld(rowR); x[5] = ld(rowR + 2);
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]) : : "memory");
#pragma unroll
                    for (int q = 0; q < 6; ++q) errors += (x[q].x != want) + (x[q].y != want + 0.5);
                } else if (P != P_FLAGONLY) {
#pragma unroll
                    for (int q = 0; q < 6; ++q) x[q] = ld(theirs + 128 * q);
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]) : : "memory");
#pragma unroll
                    for (int q = 0; q < 6; ++q) errors += (x[q].x != want) + (x[q].y != want + 0.5);
                }
            }
        }
        if (P == P_STRIDED || P == P_STRIDED_NOWAIT) {
            if (ln >= 32) {
                double* rowM = big + (size_t)me * (1 << 20) / 8 + (size_t)(ln - 32) * 640;        // 5 KB row stride
                double* rowR = rowM + (1 << 19) / 8;
#pragma unroll
                for (int q = 0; q < 4; ++q) { st(rowM + 2 * q, d2_t{(double)s, (double)s + 0.5}); st(rowR + 2 * q, d2_t{(double)s, (double)s + 0.5}); }
            }
        } else if (P != P_FLAGONLY) {
#pragma unroll
            for (int q = 0; q < 6; ++q) st(mine + 128 * q, d2_t{(double)s, (P == P_TAGGED && tagl) ? (double)s : (double)s + 0.5});
        }
        if (P == P_FLAG || P == P_STRIDED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (P != P_TAGGED && ln == 0) stf<F>(flags, s);
    }
    const long long t1 = wall_clock64();
    if (ln == 0) { out[me] = t1 - t0; out[2 + me] = errors; out[4 + me] = spins; }
}

template <int P, int F> int run(const char* name, double* tile, double* big, unsigned* flags, long long* out) {
    for (int b : {8, 1, 4}) {
        CK(hipMemset(tile, 0, 1 << 16)); CK(hipMemset(big, 0, 2 << 20)); CK(hipMemset(flags, 0, 256)); CK(hipMemset(out, 0, 1024));
        hipLaunchKernelGGL((k_pingpong<P, F>), dim3(128), dim3(64), 0, 0, 0, b, tile, big, flags, out);
        CK(hipDeviceSynchronize());
        long long h[40]; CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
        printf("%-50s WG 0 (xcc %lld) <-> WG %2d (xcc %lld): %6.3f us per hand-off, payload errors %lld / %lld, polls per hand-off %.1f%s\n", name,
               h[8], b, h[8 + b], (double)h[1] / 100.0 / (2 * ROUNDS), h[2], h[3], (double)(h[4] + h[5]) / (2 * ROUNDS),
               (h[2] < 0 || h[3] < 0) ? "  [SPIN TIMEOUT]" : "");
    }
    return 0;
}

int main() {
    double* tile; double* big; unsigned* flags; long long* out;
    CK(hipMalloc(&tile, 1 << 16)); CK(hipMalloc(&big, 2 << 20)); CK(hipMalloc(&flags, 256)); CK(hipMalloc(&out, 1024));
    if (run<P_FLAG, 0>("payload, wait, flag (sc1)  [k_chol_tiles]", tile, big, flags, out)) return 1;
    if (run<P_FLAG, 1>("payload, wait, flag (sc0 sc1)", tile, big, flags, out)) return 1;
    if (run<P_FLAG, 2>("payload, wait, flag (atomic swap)", tile, big, flags, out)) return 1;
    if (run<P_NOWAIT, 0>("payload, NO wait, flag (sc1)", tile, big, flags, out)) return 1;
    if (run<P_STRIDED, 0>("k_chol_tiles' pattern: strided rows, wait, flag", tile, big, flags, out)) return 1;
    if (run<P_STRIDED_NOWAIT, 0>("strided rows, NO wait, flag", tile, big, flags, out)) return 1;
    if (run<P_TAGGED, 0>("tagged 128-byte lines, no flag", tile, big, flags, out)) return 1;
    if (run<P_FLAGONLY, 0>("flag only (sc1)", tile, big, flags, out)) return 1;
    if (run<P_FLAGONLY, 1>("flag only (sc0 sc1)", tile, big, flags, out)) return 1;
    if (run<P_FLAGONLY, 2>("flag only (atomic swap)", tile, big, flags, out)) return 1;
    return 0;
}
