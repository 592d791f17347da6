// The dense pose solve of csrc/ba.hip (k_chol_tiles: LDL^T of the augmented system as one dataflow launch over tiles, x = R y)
// with 64-WIDE block columns and SIXTEEN waves per task - docs/history/DESIGN_rounds_1-5.md 8.1 (ii) built as a standalone prototype, so that it can be
// measured against k_chol_tiles before any of the product is touched:
//   * update phase: the 64 x 64 x 64 tile products on the matrix cores, wave (a, b) of the 4 x 4 grid owns the 16 x 16 quadrant
//     (a, b) of T and of the private copy of D; the operand tiles travel in eight 8-column slabs - waves 0..7 fetch and stage the
//     slabs of L(j, m) (M and MR), waves 8..15 those of the task's own tile row (MR) - and every wave multiplies the slabs in
//     ascending order as they appear in LDS;
//   * elimination of the stacked [D; T]: tools/chol64_probe.hip - waves 0..7 on the pivot chain over the D rows, waves 8..15 over
//     the T rows one LDS hop behind; the T waves own the results, so they publish: slab w (M then MR, 4 column pairs x 64 rows x
//     16 bytes: every store instruction fills 1 KB of whole lines) and its flag, no workgroup barrier in between;
//   * x = R y in the last tasks of the same launch, sixteen lanes per row.
// The plan is built on the host from a tile pattern (symbolic fill on tiles): dense, or `nd` = two uncoupled arcs and a separator,
// the nested-dissection shape the product's plan gives the 200 key-frame window (3 + 3 + 5 tiles of 64: eight block columns on the
// chain instead of fourteen of 32).
//   hipcc --offload-arch=gfx950 -O3 -I tools/waveemu tools/chol64_solve.hip -o tools/bin/chol64_solve && tools/bin/chol64_solve nd 3 5 20   (or: n [repetitions])
// NOT YET RUN ON A GPU (written when round 4's GPU minutes were spent).  Its logic - plan, flags, staging counters, MFMA operand
// mapping, elimination protocol, publish layout, x tasks - runs on the CPU under tools/waveemu (tasks in launch order, the waves of
// a task interleaved at random):
//   g++ -O2 -std=c++17 -pthread -DWAVEEMU -I tools/waveemu -x c++ tools/chol64_solve.hip -o /tmp/chol64_solve_emu && /tmp/chol64_solve_emu 150
#ifdef WAVEEMU
#include "waveemu.h"
#else
#include <hip/hip_runtime.h>
#endif
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

constexpr int NB = 64;                        // tile size = columns of a block column
constexpr int CW = 8;                         // columns per wave = columns per slab
constexpr int ND = NB / CW;                   // D waves; as many T waves
constexpr int kSlabs = NB / CW;
constexpr int kSlabDoubles = 2 * CW * NB;     // a published slab: M then MR, each 4 column pairs x 64 rows x 2 doubles
constexpr int kTileDoubles = kSlabs * kSlabDoubles;
constexpr int CU = 2;                         // published columns taken per poll in the catch-up phase (register budget: 128 VGPRs)
#ifdef WAVEEMU
#define __host__
#endif
__host__ __device__ inline size_t pub_tile(int kind, int i, int j, int nt, int nbc) { return ((size_t)(kind * nt + i) * nbc + j) * kTileDoubles; }

#ifdef WAVEEMU
static inline double bcast_lane(double v, int lane) { return waveemu_readlane(v, lane); }
#define WAVE_LOCKSTEP() waveemu::wave_barrier()
#define WAVE_ONE_LANE(lane) ((lane) == 0)      // a store every lane of a wave executes in one instruction: a lagging fibre would bring the value back      // see tools/chol64_probe.hip
using std::min;
#define LOAD_AGENT(v, p) v = d2_t{(p)[0], (p)[1]}
#define WAIT_VM4(a, b, c, d)
#define WAIT_VM8(a, b, c, d, e, f, g, h)
// (the hand-off's producer side at its weakest, as in tools/chol32_emu.cpp: a write-through store is visible to nobody until the wave
// waits for it; a flag raised before that wait overtakes its payload and SE2_EMU_RESIDENT > 0 shows it)
struct PendingStore { double* p; d2_t v; };
static thread_local std::vector<PendingStore> g_pending[1024];
static inline void store_agent(double* p, d2_t v) { g_pending[waveemu::g()->cur].push_back(PendingStore{p, v}); }
static inline unsigned poll_agent(const unsigned* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
static inline void wait_vm() {
    std::vector<PendingStore>& q = g_pending[waveemu::g()->cur];
    for (const PendingStore& st : q) { st.p[0] = st.v[0]; st.p[1] = st.v[1]; }
    q.clear();
    __atomic_thread_fence(__ATOMIC_RELEASE);
}
#else
#define WAVE_LOCKSTEP()
#define WAVE_ONE_LANE(lane) true
typedef double d2_t __attribute__((ext_vector_type(2)));
typedef double d4_t __attribute__((ext_vector_type(4)));
__device__ inline double bcast_lane(double v, int lane) {  // lane: wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
// agent-scope loads / write-through stores of the tile hand-off (csrc/ba.hip)
// (the loads of one staging step are all issued before the one wait that ties their results)
#define LOAD_AGENT(v, p) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory")
#define WAIT_VM4(a, b, c, d) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "memory")
#define WAIT_VM8(a, b, c, d, e, f, g, h) \
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : : "memory")
__device__ inline void store_agent(double* p, d2_t v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory"); }
__device__ inline unsigned poll_agent(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ inline void wait_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif

// tasks[b] = {tile row | kind << 16, block column, first, one past the last entry of its dependency list}; kind 0: L tile (or the
// diagonal task), 1: R tile, 2: x task of a tile row.  deps: block columns m < j, bit 15 = the task's own tile row has a tile there.
__global__ __launch_bounds__(1024) void k_chol64(const double* __restrict__ A, double* __restrict__ PUB, double* __restrict__ YU, int ld, int n,
                                                 int nbc, const int4* __restrict__ tasks, const int* __restrict__ deps,
                                                 unsigned* __restrict__ flagA, unsigned* __restrict__ flagR, unsigned epoch,
                                                 double* __restrict__ fail, double* __restrict__ xout, long long* __restrict__ dbg) {
    const unsigned bx = blockIdx.x;
    const int nt = ld / NB;
    constexpr int kTile = NB * (NB + 2);
    __shared__ __attribute__((aligned(16))) double LD[3 * kTile + NB * NB + NB];
    double (*Tb)[NB + 2] = reinterpret_cast<double (*)[NB + 2]>(LD);              // M(j, m); then the updated D
    double (*Tc)[NB + 2] = reinterpret_cast<double (*)[NB + 2]>(LD + kTile);      // MR(j, m)
    double (*Ta)[NB + 2] = reinterpret_cast<double (*)[NB + 2]>(LD + 2 * kTile);  // MR(i, m); then the updated T
    double (*COLV)[NB] = reinterpret_cast<double (*)[NB]>(LD + 3 * kTile);        // pivot rows of the elimination
    double* YUS = LD + 3 * kTile + NB * NB;                                       // the rhs row's values, D wave -> T wave (diagonal task)
    double (*MRD)[NB] = reinterpret_cast<double (*)[NB]>(LD);                     // multiplier columns, D rows: over Tb once every wave holds its columns
    double (*MRT)[NB] = reinterpret_cast<double (*)[NB]>(LD + 2 * kTile);         // ... T rows: over Ta
    __shared__ int ok_s, readyD, readyT, slab_s[kSlabs];
    __shared__ int deps_s[64];
    const int tid = threadIdx.x;
    if (tid == 0) { ok_s = 1; readyD = 0; readyT = 0; }
    if (tid < kSlabs) slab_s[tid] = 0;
    const int4 tk = tasks[bx];
    if (tid < tk.w - tk.z && tid < 64) deps_s[tid] = deps[tk.z + tid];
    __syncthreads();
    long long* stamp = dbg ? dbg + (size_t)bx * 8 : nullptr;
    if (stamp && tid == 0) stamp[0] = wall_clock64();
    const int ln = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if ((tk.x >> 16) == 2) {
        // ---- x(r) = sum_{j >= r} MR_R(r, j) y_un(j): sixteen lanes per row, four columns each
        const int r = tk.x & 0xffff;
        const int it = n / NB;                          // tile row of the rhs row
        const int row = tid / 16, c4 = (tid % 16) * 4;
        double acc = 0.0;
        for (int dq = tk.z; dq < tk.w; ++dq) {
            const int j = deps_s[dq - tk.z];
            if (tid == 0) {
                bool ok = true;
                for (int sl = 0; sl < kSlabs; ++sl) {
                    const unsigned* fr = flagR + ((size_t)r * nbc + j) * kSlabs + sl;
                    const unsigned* fy = (it == j ? flagR : flagA) + ((size_t)it * nbc + j) * kSlabs + sl;
                    const long long t0 = wall_clock64();
                    while (poll_agent(fr) != epoch || poll_agent(fy) != epoch) {
                        __builtin_amdgcn_s_sleep(2);
#ifndef WAVEEMU
                        if (wall_clock64() - t0 > 200000000ll) { ok = false; break; }
#endif
                    }
                    (void)t0;
                }
                ok_s = ok ? 1 : 0;
            }
            __syncthreads();
            if (!ok_s) {
                if (tid == 0) fail[0] = 1e6;
                return;
            }
            const int c0 = NB * j + c4;
            const double* rt = PUB + pub_tile(1, r, j, nt, nbc) + (c4 >> 3) * kSlabDoubles + kSlabDoubles / 2;   // MR_R(r, j): the slab of these 4 columns
            const int p0 = (c4 & 7) >> 1;
            d2_t m0, m1, y0, y1;
            LOAD_AGENT(m0, rt + (p0 * NB + row) * 2); LOAD_AGENT(m1, rt + ((p0 + 1) * NB + row) * 2);
            LOAD_AGENT(y0, YU + c0); LOAD_AGENT(y1, YU + c0 + 2);
            WAIT_VM4(m0, m1, y0, y1);
            if (c0 + 0 < n) acc += m0[0] * y0[0];
            if (c0 + 1 < n) acc += m0[1] * y0[1];
            if (c0 + 2 < n) acc += m1[0] * y1[0];
            if (c0 + 3 < n) acc += m1[1] * y1[1];
            __syncthreads();
        }
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        acc += __shfl_xor(acc, 4);
        acc += __shfl_xor(acc, 8);
        if ((tid & 15) == 0 && NB * r + row < n) xout[NB * r + row] = acc;
        return;
    }
    const bool isR = (tk.x >> 16) != 0;
    const int i = tk.x & 0xffff, j = tk.y;
    const bool isDiag = !isR && i == j;
    // ---- update phase: D(j, j) -= MR(j, m) M(j, m)^T and T(i, j) -= MR(i, m) M(j, m)^T over the block columns m < j
    const int qi = 16 * (wv >> 2), qj = 16 * (wv & 3);
    const int orow = qi + (ln >> 4), ocol = qj + (ln & 15);   // output element v: (orow + 4 v, ocol)
    const int arow = ln & 15, acol = ln >> 4;                 // operand element of a k-chunk: (arow, 4 chunk + acol)
    d4_t T0 = {0, 0, 0, 0}, D0 = {0, 0, 0, 0}, accT = {0, 0, 0, 0}, accD = {0, 0, 0, 0};
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        D0[v] = A[(size_t)(NB * j + orow + 4 * v) * ld + NB * j + ocol];
        if (!isR && !isDiag) T0[v] = A[(size_t)(NB * i + orow + 4 * v) * ld + NB * j + ocol];
    }
    const int sw = wv & (ND - 1);            // the slab this wave stages ...
    const bool stT = wv >= ND;               // ... of the own tile row (MR) - or of L(j, m) (M and MR)
    for (int dq = tk.z; dq < tk.w; ++dq) {
        const int dep = deps_s[dq - tk.z];
        const int m = dep & 0x7fff;
        const bool hasT = (dep >> 15) != 0;
        __syncthreads();   // everyone is done with the LDS tiles of the previous column
        if (!ok_s) {
            if (tid == 0) fail[0] = 1e6;
            return;
        }
        const int need = 2 * (dq - tk.z + 1);        // both stagers of a slab have added themselves
        const unsigned* f_own = stT ? (isR ? flagR : flagA) + ((size_t)i * nbc + m) * kSlabs + sw : flagA + ((size_t)j * nbc + m) * kSlabs + sw;
        bool staged = false;
        if (stT && !hasT) {                          // nothing to fetch: the T products of this column are skipped
            WAVE_LOCKSTEP();
            if (ln == 0) __hip_atomic_fetch_add(&slab_s[sw], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            staged = true;
        }
        int mult = 0;
        long long t0 = 0;
        while (mult < kSlabs) {
            bool progress = false;
            // (one LDS read per wave: the same value in every lane - said explicitly, the branch below holds wave collectives)
            if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&slab_s[mult], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) >= need) {
                const int k0 = CW * mult + acol;
                const double b0 = Tb[qj + arow][k0], b1 = Tb[qj + arow][k0 + 4];
                const double c0 = Tc[qi + arow][k0], c1 = Tc[qi + arow][k0 + 4];
                double a0 = 0.0, a1 = 0.0;
                if (hasT) { a0 = Ta[qi + arow][k0]; a1 = Ta[qi + arow][k0 + 4]; }
                __builtin_amdgcn_sched_barrier(0);
                accD = __builtin_amdgcn_mfma_f64_16x16x4f64(c0, b0, accD, 0, 0, 0);
                if (hasT) accT = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, accT, 0, 0, 0);
                accD = __builtin_amdgcn_mfma_f64_16x16x4f64(c1, b1, accD, 0, 0, 0);
                if (hasT) accT = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, accT, 0, 0, 0);
                ++mult;
                progress = true;
            }
            if (!staged) {
                bool here = __builtin_amdgcn_readfirstlane((int)poll_agent(f_own)) == (int)epoch;
#ifndef WAVEEMU
                if (!here && !progress) {   // 2 s without the flag: report, and count the slab as staged so that nobody waits for it
                    if (t0 == 0) t0 = wall_clock64();
                    else if (wall_clock64() - t0 > 200000000ll) {
                        if (ln == 0) ok_s = 0;
                        here = true;
                    }
                }
#endif
                (void)t0;
                if (here) {
                    // slab sw of the tile: lane = row, four column pairs; every load instruction covers 1 KB of contiguous bytes
                    const double* t = PUB + (stT ? pub_tile(isR ? 1 : 0, i, m, nt, nbc) : pub_tile(0, j, m, nt, nbc)) + sw * kSlabDoubles + ln * 2;
                    const int sc = CW * sw;
                    d2_t r0, r1, r2, r3;
                    LOAD_AGENT(r0, t + kSlabDoubles / 2); LOAD_AGENT(r1, t + kSlabDoubles / 2 + NB * 2);
                    LOAD_AGENT(r2, t + kSlabDoubles / 2 + 2 * NB * 2); LOAD_AGENT(r3, t + kSlabDoubles / 2 + 3 * NB * 2);
                    if (stT) {
                        WAIT_VM4(r0, r1, r2, r3);
                        double* d = &Ta[ln][sc];
                        d[0] = r0[0]; d[1] = r0[1]; d[2] = r1[0]; d[3] = r1[1]; d[4] = r2[0]; d[5] = r2[1]; d[6] = r3[0]; d[7] = r3[1];
                    } else {
                        d2_t m0, m1, m2, m3;
                        LOAD_AGENT(m0, t); LOAD_AGENT(m1, t + NB * 2); LOAD_AGENT(m2, t + 2 * NB * 2); LOAD_AGENT(m3, t + 3 * NB * 2);
                        WAIT_VM8(r0, r1, r2, r3, m0, m1, m2, m3);
                        double* d = &Tc[ln][sc];
                        d[0] = r0[0]; d[1] = r0[1]; d[2] = r1[0]; d[3] = r1[1]; d[4] = r2[0]; d[5] = r2[1]; d[6] = r3[0]; d[7] = r3[1];
                        d = &Tb[ln][sc];
                        d[0] = m0[0]; d[1] = m0[1]; d[2] = m1[0]; d[3] = m1[1]; d[4] = m2[0]; d[5] = m2[1]; d[6] = m3[0]; d[7] = m3[1];
                    }
                    asm volatile("" ::: "memory");
                    WAVE_LOCKSTEP();
                    if (ln == 0) __hip_atomic_fetch_add(&slab_s[sw], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    staged = true;
                    progress = true;
                }
            }
            if (!progress) {
                if (staged) __builtin_amdgcn_s_sleep(1);
                else __builtin_amdgcn_s_sleep(2);
            }
        }
    }
    __syncthreads();
    if (!ok_s) {
        if (tid == 0) fail[0] = 1e6;
        return;
    }
    if (stamp && tid == 0) stamp[1] = wall_clock64();
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        Ta[orow + 4 * v][ocol] = T0[v] - accT[v];
        Tb[orow + 4 * v][ocol] = D0[v] - accD[v];
    }
    __syncthreads();
    // ---- elimination of the stacked [D; T] (tools/chol64_probe.hip): lane = row, wave w of each half owns columns 8 w .. 8 w + 7
    const int lane = ln;
    const bool isT = stT;
    const int w = sw, cb = CW * w;
    const int c0 = NB * j;
    const int ncol = min(NB, n - c0);
    const int yrow = (!isR && i == n / NB) ? n % NB : -1;    // the rhs row lives in this task: its values after the elimination are y_un
    double mm[CW], mrs[CW];
    {
        const bool ident = isDiag && isT;            // the diagonal task eliminates [D; I]: R(j, j)
        const double* lp = isT ? &Ta[lane][cb] : &Tb[lane][cb];
#pragma unroll
        for (int q = 0; q < CW; ++q) mm[q] = ident ? (cb + q == lane ? 1.0 : 0.0) : lp[q];
        if (ncol < NB) {   // last panel only: columns past n become a decoupled block (huge diagonal, zero elsewhere)
#pragma unroll
            for (int q = 0; q < CW; ++q)
                if (cb + q >= ncol) mm[q] = (!isT && cb + q == lane) ? 1e300 : 0.0;
        }
    }
    __syncthreads();   // every wave holds its columns: MRD / MRT may overwrite Tb / Ta   (k_chol_tiles does this with a counter; a barrier here)
    for (int done = 0; done < cb; done += CU) {
        while (__hip_atomic_load(&readyD, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < done + CU) __builtin_amdgcn_s_sleep(1);
        if (isT)
            while (__hip_atomic_load(&readyT, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < done + CU) __builtin_amdgcn_s_sleep(1);
        double mrv[CU], cv[CU][CW];
#pragma unroll
        for (int u = 0; u < CU; ++u) {
            mrv[u] = isT ? MRT[done + u][lane] : MRD[done + u][lane];
#pragma unroll
            for (int q = 0; q < CW; ++q) cv[u][q] = COLV[done + u][cb + q];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < CU; ++u)
#pragma unroll
            for (int q = 0; q < CW; ++q) mm[q] = fma(-mrv[u], cv[u][q], mm[q]);
    }
    double pmin = 1e300;
    if (!isT) {
        double* mrc = &MRD[cb][lane];
        double* colv = &COLV[cb][lane];
        double mr_prev = 0.0, rvb[2][CW];
        double piv = bcast_lane(mm[0], cb);
        double x0 = __builtin_amdgcn_rcp(piv), e = fma(-piv, x0, 1.0);
#pragma unroll
        for (int q = 0; q < CW; ++q) {
            const int jj = cb + q;
            pmin = fmin(pmin, piv);
            const double mr0 = mm[q] * x0;
            const double mr = fma(mr0, e, mr0);
            mrs[q] = mr;
            mrc[q * NB] = mr;
            colv[q * NB] = mm[q];
            if (lane == yrow) YUS[jj] = mm[q];      // (diagonal task of the rhs row's tile row: its T wave publishes y_un)
            asm volatile("" ::: "memory");
            WAVE_LOCKSTEP();
            if (WAVE_ONE_LANE(lane)) __hip_atomic_store(&readyD, jj + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // (all lanes store the same value in one instruction; the emulator lets one fibre do it)
            asm volatile("" ::: "memory");
#pragma unroll
            for (int q2 = q + 2; q2 < CW; ++q2) rvb[q & 1][q2] = COLV[jj][cb + q2];
            if (q > 0) {
#pragma unroll
                for (int q2 = q + 1; q2 < CW; ++q2) mm[q2] = fma(-mr_prev, rvb[(q & 1) ^ 1][q2], mm[q2]);
            }
            if (q + 1 < CW) {
                mm[q + 1] = fma(-mr, bcast_lane(mm[q], jj + 1), mm[q + 1]);
                piv = bcast_lane(mm[q + 1], jj + 1);
                x0 = __builtin_amdgcn_rcp(piv);
                e = fma(-piv, x0, 1.0);
            }
            mr_prev = mr;
        }
        // pivots must be positive and finite (pad pivots are 1e300); a NaN shows in the pivot row's own multiplier
        const double chk = bcast_lane(mrs[CW - 1], cb + CW - 1);
        const bool bad = !(pmin > 0.0) | !(pmin < __builtin_inf()) | !(chk == chk);
        if (bad && isDiag && lane == 0) fail[0] = 1.0;
        if (stamp && lane == 0 && w == ND - 1) stamp[2] = wall_clock64();
        return;
    }
    // ---- T wave: the block of eight pivots at once, when the D wave of this block has finished; then its slab and the flag
    while (__hip_atomic_load(&readyD, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < cb + CW) __builtin_amdgcn_s_sleep(1);
    {
        double cvn[CW], pvn = COLV[cb][cb];
#pragma unroll
        for (int q2 = 1; q2 < CW; ++q2) cvn[q2] = COLV[cb][cb + q2];
#pragma unroll
        for (int q = 0; q < CW; ++q) {
            double cv[CW];
            const double pv = pvn;
#pragma unroll
            for (int q2 = q + 1; q2 < CW; ++q2) cv[q2] = cvn[q2];
            if (q + 1 < CW) {
                pvn = COLV[cb + q + 1][cb + q + 1];
#pragma unroll
                for (int q2 = q + 2; q2 < CW; ++q2) cvn[q2] = COLV[cb + q + 1][cb + q2];
            }
            const double x0 = __builtin_amdgcn_rcp(pv), e = fma(-pv, x0, 1.0);
            const double mr0 = mm[q] * x0;
            const double mr = fma(mr0, e, mr0);
            mrs[q] = mr;
            MRT[cb + q][lane] = mr;
#pragma unroll
            for (int q2 = q + 1; q2 < CW; ++q2) mm[q2] = fma(-mr, cv[q2], mm[q2]);
        }
    }
    asm volatile("" ::: "memory");
    WAVE_LOCKSTEP();
    if (WAVE_ONE_LANE(lane)) __hip_atomic_store(&readyT, cb + CW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // (all lanes store the same value in one instruction; the emulator lets one fibre do it)
    {
        double* pb = PUB + pub_tile(isR || isDiag ? 1 : 0, i, j, nt, nbc) + w * kSlabDoubles + lane * 2;
#pragma unroll
        for (int q = 0; q < CW; q += 2) {
            store_agent(pb + (q / 2) * NB * 2, d2_t{mm[q], mm[q + 1]});
            store_agent(pb + kSlabDoubles / 2 + (q / 2) * NB * 2, d2_t{mrs[q], mrs[q + 1]});
        }
    }
    // y_un = the rhs row after the elimination of this block column: a T row of the task in the rhs row's tile row, or - in the
    // diagonal task of that tile row - a D row, handed over in LDS by the D wave of these columns
    if (yrow >= 0) {
        if (isDiag) {
            if (lane < CW / 2) store_agent(YU + c0 + cb + 2 * lane, d2_t{YUS[cb + 2 * lane], YUS[cb + 2 * lane + 1]});
        } else if (lane == yrow) {
#pragma unroll
            for (int q = 0; q < CW; q += 2) store_agent(YU + c0 + cb + q, d2_t{mm[q], mm[q + 1]});
        }
    }
    wait_vm();          // slab w is this wave's alone: its write-through stores have landed -> its flag
    WAVE_LOCKSTEP();
    if (lane == 0)
        __hip_atomic_store((isR || isDiag ? flagR : flagA) + ((size_t)i * nbc + j) * kSlabs + w, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (stamp && lane == 0 && w == ND - 1) stamp[3] = wall_clock64();
}

#include "chol_host.h"

int main(int argc, char** argv) {
    // chol64_solve [n [repetitions [interleavings]]]: dense;  chol64_solve nd [arc tiles [separator tiles [repetitions ...]]]: two
    // uncoupled arcs and a separator that couples to both (the nested-dissection shape of the 200 key-frame window: 3 + 3 + 5 tiles)
    const bool nd = argc > 1 && std::string(argv[1]) == "nd";
    const int arc = nd ? (argc > 2 ? std::atoi(argv[2]) : 3) : 0, sep = nd ? (argc > 3 ? std::atoi(argv[3]) : 5) : 0;
    const bool file_mode = argc > 2 && std::string(argv[1]) == "file";
    int n = nd ? NB * (2 * arc + sep) : (argc > 1 && !file_mode ? std::atoi(argv[1]) : 600);
    const int reps = file_mode ? (argc > 3 ? std::atoi(argv[3]) : 1) : nd ? (argc > 4 ? std::atoi(argv[4]) : 1) : (argc > 2 ? std::atoi(argv[2]) : 1);
    const int seeds_arg = file_mode ? 4 : nd ? 5 : 3;
    // chol64_solve file <path> [repetitions [interleavings]]: a system and ITS PLAN from a file - int32 {n, ld, tasks, dependency entries},
    // the tasks (4 int32 each), the dependency entries, A (ld x ld doubles, the rhs in row n), as tests/test_wave_protocol_probe.py
    // writes it from the product's own se2gpu_ba_debug_solve_plan_tile(..., 64, ...)
    const bool from_file = argc > 2 && std::string(argv[1]) == "file";
    CholSystem S;
    if (from_file) {
        if (!chol_system_from_file(NB, argv[2], S)) { std::printf("cannot read %s\n", argv[2]); return 2; }
    } else {
        S = chol_system(NB, n, arc, sep);
    }
    const int ld = S.ld, nt = S.nt, nbc = S.nbc, chain = S.chain;
    n = S.n;
    const std::vector<double>& A = S.A;
    const Plan& p2 = S.plan;
    // (tile rows: nt of them - the rhs row may open one of its own; block columns: nbc)
    std::vector<double> PUB(2 * (size_t)nt * nbc * kTileDoubles, 0.0), YU(ld, 0.0), x(n, 0.0);
    std::vector<unsigned> flagA((size_t)nt * nbc * kSlabs, 0u), flagR((size_t)nt * nbc * kSlabs, 0u);
    std::vector<long long> dbg(p2.tasks.size() * 8, 0);
    double fail = 0.0;
    const int ntask = (int)p2.tasks.size();
    std::printf("n = %d%s: ld %d, %d tile rows, %d block columns (%d on the longest chain), %d tasks of 1024 threads\n", n, nd ? " (two arcs + separator)" : "", ld, nt, nbc, chain, ntask);
#ifdef WAVEEMU
    const unsigned nseeds = argc > seeds_arg ? (unsigned)std::atoi(argv[seeds_arg]) : 2;
    int rc = 0, same = 0;
    std::vector<double> x_first;
    for (unsigned seed = 1; seed <= nseeds; ++seed) {
        std::fill(PUB.begin(), PUB.end(), 0.0); std::fill(YU.begin(), YU.end(), 0.0); std::fill(x.begin(), x.end(), 0.0);
        unsigned long long sw = 0;
        // SE2_EMU_RESIDENT=k: the tasks of a launch side by side on OS threads, k in flight, dispatched in index order; default: one after
        // the other (launch order = dependency order: every flag a task polls is already up)
        const unsigned resident = std::getenv("SE2_EMU_RESIDENT") ? (unsigned)std::atoi(std::getenv("SE2_EMU_RESIDENT")) : 0u;
        for (int rep = 0; rep < reps; ++rep) {
            auto task = [&]() {
                k_chol64(A.data(), PUB.data(), YU.data(), ld, n, nbc, p2.tasks.data(), p2.deps.data(), flagA.data(), flagR.data(), seed * 100u + rep + 1u, &fail,
                         x.data(), dbg.data());
                wait_vm();      // (the end of a wave completes what it has in flight)
            };
            if (resident > 0) waveemu::run_grid(1024, ntask, resident, seed, task);
            else
                for (int t = 0; t < ntask; ++t) sw += waveemu::run_group(1024, t, ntask, seed * 7919u + t, task);
        }
        std::printf("interleaving %u: %llu switches\n", seed, sw);
#else
#define CK(x_) do { hipError_t e_ = (x_); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x_, hipGetErrorString(e_)); return 1; } } while (0)
    double *dA, *dP, *dY, *dF, *dX; unsigned *dfa, *dfr; int4* dT; int* dD; long long* dG;
    CK(hipMalloc(&dA, A.size() * 8)); CK(hipMalloc(&dP, PUB.size() * 8)); CK(hipMalloc(&dY, YU.size() * 8)); CK(hipMalloc(&dF, 8)); CK(hipMalloc(&dX, x.size() * 8));
    CK(hipMalloc(&dfa, flagA.size() * 4)); CK(hipMalloc(&dfr, flagR.size() * 4)); CK(hipMalloc(&dT, p2.tasks.size() * sizeof(int4)));
    CK(hipMalloc(&dD, std::max<size_t>(1, p2.deps.size()) * 4)); CK(hipMalloc(&dG, dbg.size() * 8));
    CK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice)); CK(hipMemset(dP, 0, PUB.size() * 8)); CK(hipMemset(dY, 0, YU.size() * 8));
    CK(hipMemset(dF, 0, 8)); CK(hipMemset(dfa, 0, flagA.size() * 4)); CK(hipMemset(dfr, 0, flagR.size() * 4)); CK(hipMemset(dG, 0, dbg.size() * 8));
    CK(hipMemcpy(dT, p2.tasks.data(), p2.tasks.size() * sizeof(int4), hipMemcpyHostToDevice)); CK(hipMemcpy(dD, p2.deps.data(), p2.deps.size() * 4, hipMemcpyHostToDevice));
    {
        hipDeviceProp_t prop;
        CK(hipGetDeviceProperties(&prop, 0));
        std::printf("%s: %zu bytes of LDS per workgroup allowed, %d CUs\n", prop.name, (size_t)prop.sharedMemPerBlock, prop.multiProcessorCount);
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    int rc = 0;
    for (int rep = 0; rep < std::max(reps, 3); ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_chol64, dim3(ntask), dim3(1024), 0, 0, dA, dP, dY, ld, n, nbc, dT, dD, dfa, dfr, (unsigned)(rep + 1), dF, dX, dG);
        CK(hipGetLastError());      // (135 KB of static LDS: a launch the device refuses shows here, not as a wrong answer)
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    std::printf("k_chol64: best of %d launches %.1f us\n", std::max(reps, 3), best * 1e3f);
    CK(hipMemcpy(x.data(), dX, x.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&fail, dF, 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(dbg.data(), dG, dbg.size() * 8, hipMemcpyDeviceToHost));
    {
#endif
        const double res = chol_residual(S, x);
        std::printf("  |A x - b|_inf / |b|_inf = %.3e %s, failure flag %g\n", res, res < 1e-11 ? "(ok)" : "(MISMATCH)", fail);
        if (!(res < 1e-11) || fail != 0.0) rc = 1;
#ifdef WAVEEMU
        // the order in which a task takes its slabs and columns is fixed by the plan, not by the schedule: the same BITS every time
        if (x_first.empty()) x_first = x;
        else if (std::memcmp(x_first.data(), x.data(), x.size() * sizeof(double)) != 0) { std::printf("  the solution differs in its bits from the first interleaving's\n"); rc = 1; }
        else ++same;
#endif
    }
#ifdef WAVEEMU
    std::printf("%u interleavings, %d of %u bit-identical to the first\n", nseeds, same, nseeds - 1);
#endif
#ifndef WAVEEMU
    // the chain: per block column the diagonal task's stamps {start, update done, D wave 7 done, last slab flagged} in 100 MHz ticks
    for (int t = 0; t < ntask; ++t)
        if ((p2.tasks[t].x >> 16) == 0 && (p2.tasks[t].x & 0xffff) == p2.tasks[t].y)
            std::printf("  diagonal task of block column %d: update %.2f us, pivots %.2f us, published %.2f us after its start; started %.2f us after task 0\n", p2.tasks[t].y,
                        (dbg[t * 8 + 1] - dbg[t * 8]) / 100.0, (dbg[t * 8 + 2] - dbg[t * 8 + 1]) / 100.0, (dbg[t * 8 + 3] - dbg[t * 8]) / 100.0,
                        (dbg[t * 8] - dbg[0]) / 100.0);
#endif
    return rc;
}
