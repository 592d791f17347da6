// waveemu: runs ONE workgroup of a wave-level HIP kernel on the CPU, for checking a wave protocol (LDS hand-offs, progress
// counters, readlane broadcasts, barriers) on a machine without a GPU.  NOT part of the product and not a portability
// layer: nothing under se2lam_amd/ or include/ knows it; the probes under tools/ that include it (-DWAVEEMU) are
// experiments that are timed on the GPU with hipcc and only have their LOGIC checked here.
//
// Model: every work-item is a fibre (ucontext) on one OS thread; a fibre runs until it yields (s_sleep, a barrier, a wave
// collective), the scheduler then picks the next wave at random (seeded) and inside it the next lane round-robin, so that
// many interleavings of the waves are exercised; plain loads / stores need no atomics.  Wave collectives (readlane,
// readfirstlane, shuffles) go through a per-wave exchange buffer with a wave barrier on both sides, so the lanes of a wave
// must call them in wave-uniform control flow - as on the hardware.  A watchdog on the number of switches reports a
// deadlock instead of hanging.  Not modelled: timing, memory ordering between waves (stores are visible at once), the
// register file, bank conflicts.
#pragma once
// (fibres are switched with _setjmp / _longjmp - swapcontext makes a system call per switch - and a fortified longjmp refuses to
// jump between stacks: this header has to be the first include)
#undef _FORTIFY_SOURCE
#include <setjmp.h>
#include <ucontext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <functional>
#include <thread>
#include <memory>
#include <random>
#include <vector>

namespace waveemu {
struct Dim3 { unsigned x = 1, y = 1, z = 1; };
struct Fibre {
    ucontext_t ctx;
    jmp_buf jb;
    bool started = false;
    std::unique_ptr<char[]> stack;      // (not zero-filled: only the pages a fibre touches are ever mapped)
    bool done = false;
    int where = 0;      // what the fibre last yielded in: 1 __syncthreads, 2 a wave collective, 100 + n s_sleep(n)
};
struct Wave {
    double slot[64], slot2[64];
    int arrived = 0;
    unsigned gen = 0;
};
struct Group {
    int nthreads = 0, cur = 0;
    std::vector<Fibre> fibres;
    std::vector<Wave> waves;
    ucontext_t sched;
    jmp_buf sched_jb;
    int bar_arrived = 0;
    unsigned bar_gen = 0;
    unsigned long long switches = 0, limit = 0;
    std::function<void()> body;
    Dim3 block_idx, block_dim, grid_dim;
};
inline Group*& g() { static thread_local Group* p = nullptr; return p; }      // (one workgroup per OS thread: run_grid)
inline void yield() {
    Group* G = g();
    if (++G->switches > G->limit) {
        std::fprintf(stderr, "waveemu: %llu switches without finishing - deadlock? (workgroup %u, thread %d was running)\n", G->switches, G->block_idx.x, G->cur);
        for (size_t w = 0; w < G->waves.size(); ++w) {       // where every wave's lanes are waiting
            std::fprintf(stderr, "  wave %zu:", w);
            int last = -1, run = 0;
            for (int l = 0; l <= 64; ++l) {
                const size_t t = w * 64 + l;
                const int wh = l < 64 && t < G->fibres.size() ? (G->fibres[t].done ? -2 : G->fibres[t].where) : -3;
                if (wh != last) {
                    if (run) std::fprintf(stderr, " %dx%s%d", run, last == -2 ? "done" : last == 1 ? "syncthreads" : last == 2 ? "collective" : last >= 100 ? "sleep" : "?", last >= 100 ? last - 100 : 0);
                    last = wh; run = 0;
                }
                ++run;
            }
            std::fprintf(stderr, "\n");
        }
        std::abort();
    }
    if (_setjmp(G->fibres[G->cur].jb) == 0) _longjmp(G->sched_jb, 1);
}
inline void trampoline() {
    Group* G = g();
    G->body();
    G->fibres[G->cur].done = true;
    _longjmp(G->sched_jb, 1);
}
// runs one workgroup of `nthreads` work-items; `seed` picks the interleaving
inline unsigned long long run_group(int nthreads, unsigned bx, unsigned nblocks, unsigned seed, std::function<void()> body,
                                    unsigned long long limit = 200000000ull) {
    Group G;
    g() = &G;
    G.nthreads = nthreads;
    G.fibres.resize(nthreads);
    G.waves.resize((nthreads + 63) / 64);
    G.body = body;
    G.limit = limit;
    G.block_idx.x = bx; G.block_dim.x = nthreads; G.grid_dim.x = nblocks;
    for (int t = 0; t < nthreads; ++t) {
        Fibre& f = G.fibres[t];
        constexpr size_t kStack = 256 * 1024;
        f.stack.reset(new char[kStack]);
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack.get();
        f.ctx.uc_stack.ss_size = kStack;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, (void (*)())trampoline, 0);
    }
    std::mt19937 rng(seed);
    const int nw = (int)G.waves.size();
    std::vector<int> next_lane(nw, 0);
    // every wave gets a weight from the seed - 1, 4, 16 or 64 - and is picked in proportion to it: some interleavings starve a wave
    // for a long time while the others run ahead, which is what finds a missing wait
    std::vector<int> ticket;
    for (int w = 0; w < nw; ++w) {
        const int weight = seed == 0 ? 1 : 1 << (2 * (int)(rng() % 4));
        for (int k = 0; k < weight; ++k) ticket.push_back(w);
    }
    int live = nthreads;
    while (live > 0) {
        // a wave by its weight, then up to 64 of its lanes in turn (a burst keeps the lanes of a wave close together, as they are)
        const int w = seed == 0 ? (int)(G.switches % nw) : ticket[rng() % ticket.size()];
        const int burst = seed == 0 ? 64 : 1 + (int)(rng() % 64);
        for (int k = 0; k < burst && live > 0; ++k) {
            const int lane = next_lane[w];
            next_lane[w] = (lane + 1) % 64;
            const int t = w * 64 + lane;
            if (t >= nthreads || G.fibres[t].done) continue;
            G.cur = t;
            if (_setjmp(G.sched_jb) == 0) {
                if (G.fibres[t].started) _longjmp(G.fibres[t].jb, 1);
                G.fibres[t].started = true;
                setcontext(&G.fibres[t].ctx);
            }
            if (G.fibres[t].done) --live;
        }
        if (++G.switches > G.limit) { std::fprintf(stderr, "waveemu: scheduler limit reached - deadlock?\n"); std::abort(); }
    }
    g() = nullptr;
    return G.switches;
}

// A launch of `nblocks` workgroups with at most `resident` of them in flight, each on an OS thread of its own and dispatched in
// order of their index as the hardware does: what a dataflow kernel relies on when its grid exceeds what is resident (a task
// may wait only for tasks with a lower index).  The workgroups really run side by side - global memory is shared as it is.
inline void run_grid(int nthreads, unsigned nblocks, unsigned resident, unsigned seed, std::function<void()> body,
                     unsigned long long limit = 4000000000ull) {
    std::atomic<unsigned> next{0};
    std::vector<std::thread> pool;
    for (unsigned k = 0; k < resident; ++k)
        pool.emplace_back([&]() {
            for (;;) {
                const unsigned b = next.fetch_add(1);
                if (b >= nblocks) return;
                run_group(nthreads, b, nblocks, seed * 7919u + b, body, limit);
            }
        });
    for (std::thread& t : pool) t.join();
}

inline void block_barrier() {
    Group* G = g();
    const unsigned gen = G->bar_gen;
    if (++G->bar_arrived == G->nthreads) { G->bar_arrived = 0; ++G->bar_gen; return; }
    G->fibres[G->cur].where = 1;
    while (G->bar_gen == gen) yield();
}
inline void wave_barrier() {
    Group* G = g();
    Wave& W = G->waves[G->cur / 64];
    const int width = std::min(64, G->nthreads - (G->cur / 64) * 64);
    const unsigned gen = W.gen;
    if (++W.arrived == width) { W.arrived = 0; ++W.gen; return; }
    G->fibres[G->cur].where = 2;
    while (W.gen == gen) yield();
}
// every lane hands in v; afterwards every lane can read any lane's value
inline double wave_read(double v, int lane) {
    Group* G = g();
    Wave& W = G->waves[G->cur / 64];
    W.slot[G->cur % 64] = v;
    wave_barrier();
    const double r = W.slot[lane];
    wave_barrier();
    return r;
}
}  // namespace waveemu

// ---- the spellings a kernel uses -------------------------------------------------------------------------------------
#define __global__
#define __device__
#define __shared__ static thread_local      // one copy per OS thread = per workgroup in flight
#define __launch_bounds__(...)
#define __restrict__
#define __forceinline__ inline
#define threadIdx (waveemu::Dim3{(unsigned)waveemu::g()->cur, 0, 0})
#define blockIdx (waveemu::g()->block_idx)
#define blockDim (waveemu::g()->block_dim)
#define gridDim (waveemu::g()->grid_dim)
#define __ATOMIC_SCOPE_IGNORED 0
#define __HIP_MEMORY_SCOPE_WORKGROUP 0
#define __HIP_MEMORY_SCOPE_AGENT 0
// (acquire / release whatever the kernel asks for: on the host they cost nothing and keep the compiler from moving a payload
// access across a flag when workgroups run on several OS threads)
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), __ATOMIC_ACQUIRE)
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), __ATOMIC_RELEASE)
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add((p), (v), __ATOMIC_ACQ_REL)
inline void __syncthreads() { waveemu::block_barrier(); }
inline void __builtin_amdgcn_s_sleep(int n) { waveemu::g()->fibres[waveemu::g()->cur].where = 100 + n; waveemu::yield(); }
inline void __builtin_amdgcn_sched_barrier(int) {}
inline double __builtin_amdgcn_rcp(double x) { return (double)(float)(1.0 / x); }   // v_rcp_f64 is a ~single-precision seed: the kernels refine it
inline int __builtin_amdgcn_readfirstlane(int v) { return (int)waveemu::wave_read((double)v, 0); }
inline double waveemu_readlane(double v, int lane) { return waveemu::wave_read(v, lane); }
inline double __shfl_xor(double v, int mask) { return waveemu::wave_read(v, (waveemu::g()->cur % 64) ^ mask); }
// clang's ext_vector_type(2 / 4) of double as far as the kernels use them: brace initialisation, .x / .y, [i]
struct d2_t {
    double x, y;
    double& operator[](int i) { return i ? y : x; }
    const double& operator[](int i) const { return i ? y : x; }
};
struct d4_t {
    double v[4];
    double& operator[](int i) { return v[i]; }
    const double& operator[](int i) const { return v[i]; }
};
struct int4 { int x, y, z, w; };
struct double2 { double x, y; };
inline long long __shfl_xor(long long v, int mask) {
    double d; std::memcpy(&d, &v, 8);
    d = waveemu::wave_read(d, (waveemu::g()->cur % 64) ^ mask);
    std::memcpy(&v, &d, 8);
    return v;
}
inline int __double2hiint(double d) { long long v; std::memcpy(&v, &d, 8); return (int)(v >> 32); }
inline long long __double_as_longlong(double d) { long long v; std::memcpy(&v, &d, 8); return v; }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { const unsigned long long o = *p; *p += v; return o; }
// v_mfma_f64_16x16x4_f64 in the layout tools/mfma_probe.hip found on gfx950: A[i][k] in lane i + 16 k, B[k][j] in lane j + 16 k,
// D[i][j] in lane j + 16 (i % 4), register i / 4
inline d4_t __builtin_amdgcn_mfma_f64_16x16x4f64(double a, double b, d4_t c, int, int, int) {
    waveemu::Group* G = waveemu::g();
    waveemu::Wave& W = G->waves[G->cur / 64];
    const int lane = G->cur % 64;
    W.slot[lane] = a; W.slot2[lane] = b;
    waveemu::wave_barrier();
    const int j = lane & 15, ib = lane >> 4;
    for (int v = 0; v < 4; ++v) {
        const int i = ib + 4 * v;
        double acc = c[v];
        for (int k = 0; k < 4; ++k) acc = std::fma(W.slot[i + 16 * k], W.slot2[j + 16 * k], acc);
        c[v] = acc;
    }
    waveemu::wave_barrier();
    return c;
}
inline long long wall_clock64() { return (long long)waveemu::g()->switches; }
inline long long clock64() { return (long long)waveemu::g()->switches; }
