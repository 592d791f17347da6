// The 64-wide block column of docs/history/DESIGN_rounds_1-5.md 8.1 (ii) as a standalone experiment: the elimination of the stacked [D; T] - a 64 x 64
// diagonal copy D and a 64 x 64 tile T - by SIXTEEN waves, in the LDL^T form of k_chol_tiles (csrc/ba.hip):
//      M = the unnormalised elimination result, MR = M diag(1 / pivot);   [D; T] = [MR_D; MR_T] diag(pivot) MR_D^T
//   waves 0..7  own columns 8 w .. 8 w + 7 over the 64 D rows (lane = row): the pivot chain, as wave w of k_chol_tiles today -
//               catch up with the columns of the waves before them as those appear in LDS, then eliminate their own eight
//               (pivot through v_readlane, reciprocal seed + Newton residual, the pivot row's values back from LDS);
//   waves 8..15 own the same columns over the 64 T rows, one LDS hop behind: they need only the pivot row (COLV) of a
//               column, recompute its reciprocal with the same two instructions (bit-identical multipliers), and take a whole
//               block of eight pivots at once when the D wave of that block has finished - nothing of theirs is on the chain.
// What the experiment is to answer on the GPU: does a pivot still cost ~100 ns with sixteen waves on the LDS (then a 64-wide
// column is 6.4 us + one hand-off instead of 2 x (3.2 us + one hand-off)), and how far behind do the T waves end.
//   hipcc --offload-arch=gfx950 -O3 [-DCATCH=4] tools/chol64_probe.hip -o tools/bin/chol64_probe && tools/bin/chol64_probe [blocks [repetitions]]
// (gfx950, -O3, CATCH=2: 104 VGPRs, no scratch, 98,312 bytes of LDS = one workgroup per CU; not yet run on a GPU - written at the end
// of round 4 when the round's GPU minutes were spent.)
// The LOGIC (progress counters, who reads what when, the result) is checked without a GPU by tools/waveemu:
//   g++ -O2 -std=c++17 -pthread -DWAVEEMU -I tools/waveemu -x c++ tools/chol64_probe.hip -o /tmp/chol64_emu && /tmp/chol64_emu
#ifdef WAVEEMU
#include "waveemu.h"
#else
#include <hip/hip_runtime.h>
#endif
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

constexpr int NB = 64;     // columns of the block column = D rows = T rows
constexpr int CW = 8;      // columns per wave
constexpr int ND = NB / CW;   // D waves (and as many T waves)
#ifndef CATCH
#define CATCH 2     // 4 (what k_chol_tiles does with its four waves) needs 6 VGPRs more than the 128 a wave of a 1024-thread workgroup has: it spills
#endif
#ifndef PIVCHK
#define PIVCHK 0      // 1: the pivot test on the high dword of the pivot in scalar registers (s_min_i32 / s_max_i32) instead of v_max_f64 + v_min_f64
#endif
constexpr int CU = CATCH;      // published columns taken per poll in the catch-up phase

#ifdef WAVEEMU
static inline double bcast_lane(double v, int lane) { return waveemu_readlane(v, lane); }
static inline int hi_word(double v) { long long b; memcpy(&b, &v, 8); return (int)(b >> 32); }
using std::min; using std::max;
// The lanes of a wave execute an LDS instruction together: what one lane wrote, every lane of the SAME wave reads back in the next
// instruction, and a counter stored after a write is behind the writes of all 64 lanes.  The emulator's lanes are separate
// fibres, so the places that rely on this say so.
#define WAVE_LOCKSTEP() waveemu::wave_barrier()
#define WAVE_ONE_LANE(lane) ((lane) == 0)      // a store every lane of a wave executes in one instruction: a lagging fibre would bring the value back
#else
#define WAVE_LOCKSTEP()
#define WAVE_ONE_LANE(lane) true
__device__ inline double bcast_lane(double v, int lane) {  // lane: wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ inline int hi_word(double v) { return __double2hiint(v); }
#endif

// out: MD | MRD | MT | MRT, each NB x NB row-major;  stamps[wave]: 100 MHz ticks {start of own columns, end}
__global__ __launch_bounds__(1024) void k_elim64(const double* __restrict__ Din, const double* __restrict__ Tin, double* __restrict__ out,
                                                 long long* __restrict__ stamps, double* __restrict__ fail, int reps) {
    __shared__ __attribute__((aligned(16))) double COLV[NB][NB];   // [jj][c]: element (jj, c) when column jj is eliminated (= column jj's D rows, by symmetry)
    __shared__ __attribute__((aligned(16))) double MRD[NB][NB];    // [jj][row]: multiplier column jj, D rows
    __shared__ __attribute__((aligned(16))) double MRT[NB][NB];    // ... T rows
    __shared__ int readyD, readyT;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool isT = wv >= ND;
    const int w = wv % ND, cb = CW * w;
    const size_t blk = (size_t)blockIdx.x * NB * NB;
    for (int rep = 0; rep < reps; ++rep) {
        if (tid == 0) { readyD = 0; readyT = 0; }
        double m[CW], mrs[CW];
        {
            const double* src = (isT ? Tin : Din) + blk + (size_t)lane * NB + cb;
#pragma unroll
            for (int q = 0; q < CW; ++q) m[q] = src[q];
        }
        __syncthreads();
        long long t_begin = wall_clock64();
        // ---- the columns of the waves before this one, CU at a time as they are published
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
                for (int q = 0; q < CW; ++q) m[q] = fma(-mrv[u], cv[u][q], m[q]);
        }
        long long t_own = wall_clock64();
        double pmin = 1e300;
        int hmin = 0x7fffffff, hmax = 0;      // PIVCHK: a pivot is positive and finite iff 0 < its high dword < 0x7ff00000 (denormal pivots count as failures)
        if (!isT) {
            // ---- D wave: the pivot chain of its eight columns (the q loop of k_chol_tiles with 64 D rows in the wave)
            double* mrc = &MRD[cb][lane];
            double* colv = &COLV[cb][lane];
            double mr_prev = 0.0, rvb[2][CW];
            double piv = bcast_lane(m[0], cb);
            double x0 = __builtin_amdgcn_rcp(piv), e = fma(-piv, x0, 1.0);
#pragma unroll
            for (int q = 0; q < CW; ++q) {
                const int jj = cb + q;
#if PIVCHK
                { const int h = hi_word(piv); hmin = min(hmin, h); hmax = max(hmax, h); }
#else
                pmin = fmin(pmin, piv);
#endif
                const double mr0 = m[q] * x0;
                const double mr = fma(mr0, e, mr0);
                mrs[q] = mr;
                mrc[q * NB] = mr;
                colv[q * NB] = m[q];
                asm volatile("" ::: "memory");
                WAVE_LOCKSTEP();
                if (WAVE_ONE_LANE(lane)) __hip_atomic_store(&readyD, jj + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // (all lanes store the same value in one instruction; the emulator lets one fibre do it)
                asm volatile("" ::: "memory");
#pragma unroll
                for (int q2 = q + 2; q2 < CW; ++q2) rvb[q & 1][q2] = COLV[jj][cb + q2];
                if (q > 0) {
#pragma unroll
                    for (int q2 = q + 1; q2 < CW; ++q2) m[q2] = fma(-mr_prev, rvb[(q & 1) ^ 1][q2], m[q2]);
                }
                if (q + 1 < CW) {
                    m[q + 1] = fma(-mr, bcast_lane(m[q], jj + 1), m[q + 1]);
                    piv = bcast_lane(m[q + 1], jj + 1);
                    x0 = __builtin_amdgcn_rcp(piv);
                    e = fma(-piv, x0, 1.0);
                }
                mr_prev = mr;
            }
        } else {
            // ---- T wave: the whole block of eight pivots at once, when the D wave of this block has finished
            while (__hip_atomic_load(&readyD, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < cb + CW) __builtin_amdgcn_s_sleep(1);
            // (two pivot rows in registers at a time: sixteen waves of a workgroup share the register file at 128 VGPRs each)
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
                const double mr0 = m[q] * x0;
                const double mr = fma(mr0, e, mr0);
                mrs[q] = mr;
                MRT[cb + q][lane] = mr;
#pragma unroll
                for (int q2 = q + 1; q2 < CW; ++q2) m[q2] = fma(-mr, cv[q2], m[q2]);
            }
            asm volatile("" ::: "memory");
            WAVE_LOCKSTEP();
            if (WAVE_ONE_LANE(lane)) __hip_atomic_store(&readyT, cb + CW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // (all lanes store the same value in one instruction; the emulator lets one fibre do it)
        }
        long long t_end = wall_clock64();
        if (rep == reps - 1) {
            double* o = out + 4 * blk + (isT ? 2 : 0) * (size_t)NB * NB + (size_t)lane * NB + cb;
#pragma unroll
            for (int q = 0; q < CW; ++q) { o[q] = m[q]; o[(size_t)NB * NB + q] = mrs[q]; }
            if (lane == 0) {
                stamps[(size_t)blockIdx.x * 48 + 3 * wv] = t_own - t_begin;
                stamps[(size_t)blockIdx.x * 48 + 3 * wv + 1] = t_end - t_begin;
                stamps[(size_t)blockIdx.x * 48 + 3 * wv + 2] = t_end - t_own;
            }
            if (!isT && (PIVCHK ? (hmin <= 0 || hmax >= 0x7ff00000) : !(pmin > 0.0)) && lane == 0) fail[0] = 1.0;
        }
        __syncthreads();   // the LDS arrays are rewritten by the next repetition
    }
}

// ---- host: a random SPD D and a random T, the elimination in plain loops, the comparison
static void reference(const std::vector<double>& D, const std::vector<double>& T, std::vector<double>& M, std::vector<double>& MR) {
    // stacked [D; T] (2 NB rows): column by column, m -= (m_jj / piv) (pivot row)
    std::vector<double> S(2 * NB * NB);
    for (int r = 0; r < NB; ++r) for (int c = 0; c < NB; ++c) { S[r * NB + c] = D[r * NB + c]; S[(NB + r) * NB + c] = T[r * NB + c]; }
    M.assign(2 * NB * NB, 0.0); MR.assign(2 * NB * NB, 0.0);
    for (int j = 0; j < NB; ++j) {
        const double piv = S[j * NB + j];
        std::vector<double> prow(NB);
        for (int c = 0; c < NB; ++c) prow[c] = S[j * NB + c];
        for (int r = 0; r < 2 * NB; ++r) {
            const double mr = S[r * NB + j] / piv;
            M[r * NB + j] = S[r * NB + j];
            MR[r * NB + j] = mr;
            for (int c = j + 1; c < NB; ++c) S[r * NB + c] -= mr * prow[c];
        }
    }
}

int main(int argc, char** argv) {
    const int nblk = argc > 1 ? std::atoi(argv[1]) : 1;
    const int reps = argc > 2 ? std::atoi(argv[2]) : 1;
    std::vector<double> D((size_t)nblk * NB * NB), T((size_t)nblk * NB * NB);
    unsigned long long s = 88172645463325252ull;
    auto unit = [&]() { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return (double)((s * 2685821657736338717ull) >> 11) * (1.0 / 9007199254740992.0); };
    for (int b = 0; b < nblk; ++b) {
        std::vector<double> G(NB * NB);
        for (double& v : G) v = unit() - 0.5;
        double* d = &D[(size_t)b * NB * NB];
        for (int r = 0; r < NB; ++r)
            for (int c = 0; c < NB; ++c) {
                double a = r == c ? 4.0 : 0.0;
                for (int k = 0; k < NB; ++k) a += G[r * NB + k] * G[c * NB + k];
                d[r * NB + c] = a;
            }
        for (int k = 0; k < NB * NB; ++k) T[(size_t)b * NB * NB + k] = 4.0 * (unit() - 0.5);
    }
    std::vector<double> out((size_t)nblk * 4 * NB * NB);
    std::vector<long long> stamps((size_t)nblk * 48);
    double fail = 0.0;
#ifdef WAVEEMU
    const unsigned nseeds = argc > 3 ? (unsigned)std::atoi(argv[3]) : 6;
    for (unsigned seed = 0; seed < nseeds; ++seed) {
        std::fill(out.begin(), out.end(), -1.0);
        for (int b = 0; b < nblk; ++b) {
            const unsigned long long sw = waveemu::run_group(1024, b, nblk, seed, [&]() { k_elim64(D.data(), T.data(), out.data(), stamps.data(), &fail, reps); });
            if (b == 0) std::printf("interleaving %u: %llu switches\n", seed, sw);
        }
#else
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
    double *dD, *dT, *dO, *dF; long long* dS;
    CK(hipMalloc(&dD, D.size() * 8)); CK(hipMalloc(&dT, T.size() * 8)); CK(hipMalloc(&dO, out.size() * 8)); CK(hipMalloc(&dF, 8));
    CK(hipMalloc(&dS, stamps.size() * 8));
    CK(hipMemcpy(dD, D.data(), D.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dT, T.data(), T.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(dF, 0, 8));
    for (int pass = 0; pass < 2; ++pass) {      // the second launch is the warm one
        hipLaunchKernelGGL(k_elim64, dim3(nblk), dim3(1024), 0, 0, dD, dT, dO, dS, dF, reps);
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(out.data(), dO, out.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(stamps.data(), dS, stamps.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&fail, dF, 8, hipMemcpyDeviceToHost));
    {
#endif
        double worst = 0.0;
        for (int b = 0; b < nblk; ++b) {
            std::vector<double> M, MR;
            reference(std::vector<double>(D.begin() + (size_t)b * NB * NB, D.begin() + (size_t)(b + 1) * NB * NB),
                      std::vector<double>(T.begin() + (size_t)b * NB * NB, T.begin() + (size_t)(b + 1) * NB * NB), M, MR);
            const double* o = &out[(size_t)b * 4 * NB * NB];
            for (int part = 0; part < 2; ++part)        // D rows, T rows
                for (int r = 0; r < NB; ++r)
                    for (int c = 0; c < NB; ++c) {
                        const double wm = M[(part * NB + r) * NB + c], wr = MR[(part * NB + r) * NB + c];
                        const double gm = o[(2 * part) * NB * NB + r * NB + c], gr = o[(2 * part + 1) * NB * NB + r * NB + c];
                        // (rows above the diagonal of the D part hold round-off of an exact zero: compared absolutely)
                        const double dm = std::fabs(gm - wm) / std::fmax(1.0, std::fabs(wm)), dr = std::fabs(gr - wr) / std::fmax(1.0, std::fabs(wr));
                        if (!(dm <= worst)) worst = dm;     // (a NaN stays)
                        if (!(dr <= worst) && worst == worst) worst = dr;
                    }
        }
        std::printf("blocks %d, repetitions %d: worst difference to the plain elimination %.3e %s, pivot failure flag %g\n", nblk, reps, worst,
                    worst < 1e-11 ? "(ok)" : "(MISMATCH)", fail);
        if (!(worst < 1e-11)) return 1;
    }
#ifndef WAVEEMU
    std::printf("100 MHz ticks, block 0, last repetition: wave | catch-up | own columns | total\n");
    for (int wv = 0; wv < 16; ++wv)
        std::printf("  %s wave %d: %5lld %5lld %5lld\n", wv < 8 ? "D" : "T", wv % 8, stamps[3 * wv], stamps[3 * wv + 2], stamps[3 * wv + 1]);
    std::printf("the chain: D wave 7 ends %.2f us after the start (64 pivots: %.0f ns each); the last T wave %.2f us\n", stamps[3 * 7 + 1] / 100.0,
                stamps[3 * 7 + 1] * 10.0 / 64.0, stamps[3 * 15 + 1] / 100.0);
#endif
    return 0;
}
