// Inter-workgroup hand-off latency on MI355X: two workgroups ping-pong an 8 KB tile through global memory with
// agent-scope release/acquire flags (bounded spins).  Compares same-XCD (workgroup ids 0 and 8) with cross-XCD (0 and 1).
//   hipcc --offload-arch=gfx950 -O3 tools/flag_probe.hip -o tools/bin/flag_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ROUNDS = 200;
constexpr int SPIN_MAX = 1 << 20;

__device__ inline bool wait_flag(unsigned* f, unsigned want) {
    for (int i = 0; i < SPIN_MAX; ++i) {
        if (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

// partner ids: a and b (block indices).  payload doubles per thread: 4 (256 threads -> 8 KB)
__global__ __launch_bounds__(256) void k_pingpong(int a, int b, double* tile, unsigned* flags, long long* out, int payload) {
    const int me = (int)blockIdx.x == a ? 0 : ((int)blockIdx.x == b ? 1 : -1);
    if (me < 0) return;
    __shared__ int ok;
    double v[4] = {1, 2, 3, 4};
    long long t0 = wall_clock64();
    for (int r = 0; r < ROUNDS; ++r) {
        // turn: even half-steps belong to 0, odd to 1.  step s = 2r + me
        const unsigned s = 2 * r + me;
        if (threadIdx.x == 0) ok = wait_flag(flags, s) ? 1 : 0;  // flags[0] counts completed half-steps
        __syncthreads();
        if (!ok) { if (threadIdx.x == 0) out[2 + me] = -1; return; }
        if (payload) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] += __builtin_nontemporal_load(tile + threadIdx.x * 4 + q) * 0.5;
#pragma unroll
            for (int q = 0; q < 4; ++q) tile[threadIdx.x * 4 + q] = v[q];
        }
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(flags, s + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[me] = t1 - t0; out[2 + me] = (long long)v[0]; }
}

int main() {
    double* tile; unsigned* flags; long long* out;
    CK(hipMalloc(&tile, 1 << 16)); CK(hipMalloc(&flags, 256)); CK(hipMalloc(&out, 256));
    for (int payload = 0; payload < 2; ++payload)
        for (int b : {8, 1, 4, 16, 64}) {
            CK(hipMemset(tile, 0, 1 << 16)); CK(hipMemset(flags, 0, 256)); CK(hipMemset(out, 0, 256));
            hipLaunchKernelGGL(k_pingpong, dim3(128), dim3(256), 0, 0, 0, b, tile, flags, out, payload);
            CK(hipDeviceSynchronize());
            long long h[4]; CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
            // wall_clock64 ticks at 100 MHz
            printf("WG 0 <-> WG %2d  payload %s: %7.3f us per hand-off (one direction)%s\n", b, payload ? "8 KB r+w" : "none   ",
                   (double)h[0] / 100.0 / (2 * ROUNDS), (h[2] < 0 || h[3] < 0) ? "  [SPIN TIMEOUT]" : "");
        }
    return 0;
}
