// The SHIPPED dense pose solve - d_chol_tiles of se2lam_amd/csrc/ba.hip, its own lines (experiments/waveemu/extract_chol_tiles.py cuts them
// out at build time and lists its few substitutions) - run on the CPU by experiments/waveemu: one task after the other in launch order, the
// four waves of a task interleaved at random from a seed.  What this looks for is a LOGIC race inside a workgroup that a
// hardware soak only meets once in hundreds of runs: the slab counters against the staging tiles, the multiplier columns that are
// overlaid on Tc / Ta (`loaded_s`), `ready_s` against COLV / MRC, the x tasks.  What it cannot see: the memory ordering between
// workgroups (the emulator's stores are visible at once) - that is what tools/soak_fresh.sh checks on the GPU.  Test harness only.
//   python experiments/waveemu/extract_chol_tiles.py /tmp/chol32_body.inc
//   g++ -O2 -std=c++17 -pthread -I experiments/waveemu -I /tmp experiments/chol32_emu.cpp -o /tmp/chol32_emu && /tmp/chol32_emu nd 3 5 1 20
//   SE2_EMU_RESIDENT=4 /tmp/chol32_emu 100 50     (the tasks of a launch side by side, four in flight, dispatched in index order)
#include "waveemu.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using std::min;
using std::max;
#define __host__
#define WAVE_LOCKSTEP() waveemu::wave_barrier()
#define SE2_WAIT_VM6(a, b, c, d, e, f) WAIT_VM()
static inline void WAIT_VM();
struct BaCtl { int done; };
static inline double bcast_lane(double v, int lane) { return waveemu_readlane(v, lane); }
static inline d2_t load_agent(const double* p) { return d2_t{p[0], p[1]}; }
// The hand-off's producer side, modelled at its weakest: a write-through store is not visible to anybody until the wave waits for
// it (`s_waitcnt vmcnt(0)` = WAIT_VM()) - the latest moment the hardware may complete it.  A flag raised before that wait reaches a
// consumer ahead of the payload, and the consumer's solution is wrong: `SE2_EMU_RESIDENT` > 0 (tasks side by side) then shows it.
struct PendingStore { double* p; d2_t v; };
static thread_local std::vector<PendingStore> g_pending[256];
static inline void store_agent(double* p, d2_t v) { g_pending[waveemu::g()->cur].push_back(PendingStore{p, v}); }
static inline void WAIT_VM() {
    std::vector<PendingStore>& q = g_pending[waveemu::g()->cur];
    for (const PendingStore& s : q) { s.p[0] = s.v.x; s.p[1] = s.v.y; }
    q.clear();
    __atomic_thread_fence(__ATOMIC_RELEASE);
}
static inline unsigned poll_agent(const unsigned* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }

#include "chol32_body.inc"
#include "chol_host.h"

int main(int argc, char** argv) {
    // chol32_emu n [interleavings [first seed]]   |   chol32_emu nd <arc tiles> <separator tiles> [first seed [interleavings]]
    const bool nd = argc > 1 && std::string(argv[1]) == "nd";
    const int arc = nd ? (argc > 2 ? std::atoi(argv[2]) : 3) : 0, sep = nd ? (argc > 3 ? std::atoi(argv[3]) : 5) : 0;
    const bool file_mode = argc > 2 && std::string(argv[1]) == "file";
    const int n = nd || file_mode ? 0 : (argc > 1 ? std::atoi(argv[1]) : 100);
    const unsigned seed0 = file_mode ? (argc > 3 ? (unsigned)std::atoi(argv[3]) : 1u) : nd ? (argc > 4 ? (unsigned)std::atoi(argv[4]) : 1u) : (argc > 3 ? (unsigned)std::atoi(argv[3]) : 1u);
    const unsigned nseeds = file_mode ? (argc > 4 ? (unsigned)std::atoi(argv[4]) : 2u) : nd ? (argc > 5 ? (unsigned)std::atoi(argv[5]) : 3u) : (argc > 2 ? (unsigned)std::atoi(argv[2]) : 3u);
    // SE2_EMU_RESIDENT=k: the tasks of a launch run side by side, k at a time (default: one after the other)
    const unsigned resident = std::getenv("SE2_EMU_RESIDENT") ? (unsigned)std::atoi(std::getenv("SE2_EMU_RESIDENT")) : 0u;
    // chol32_emu file <path> [first seed [interleavings]]: a system and its plan from a file (chol_host.h)
    const bool from_file = argc > 2 && std::string(argv[1]) == "file";
    CholSystem S;
    if (from_file) {
        if (!chol_system_from_file(kNB, argv[2], S)) { std::printf("cannot read %s\n", argv[2]); return 2; }
    } else {
        S = chol_system(kNB, n, arc, sep);
    }
    // SE2_EMU_INDEFINITE=1: one diagonal entry negated - the diagonal task of that block column must raise the failure flag
    const bool indefinite = std::getenv("SE2_EMU_INDEFINITE") != nullptr;
    if (indefinite) { const int k = S.n / 2; S.A[(size_t)k * S.ld + k] = -S.A[(size_t)k * S.ld + k]; }
    const int ntask = (int)S.plan.tasks.size();
    if (std::getenv("SE2_EMU_PRINT_PLAN")) {      // the plan only: tests compare it with the product's solve_plan_build
        for (const int4& t : S.plan.tasks) std::printf("T %d %d %d %d\n", t.x, t.y, t.z, t.w);
        for (int d : S.plan.deps) std::printf("D %d\n", d);
        return 0;
    }
    std::printf("n = %d%s: ld %d, %d tile rows, %d block columns (%d on the longest chain), %d tasks of 256 threads\n", S.n, nd ? " (two arcs + separator)" : "", S.ld,
                S.nt, S.nbc, S.chain, ntask);
    std::vector<double> PUB(2 * (size_t)S.nt * S.nbc * kSlabs * kSlabDoubles + S.ld, 0.0), x(S.n, 0.0);
    std::vector<unsigned> flagA((size_t)S.nt * S.nbc * kSlabs, 0u), flagR((size_t)S.nt * S.nbc * kSlabs, 0u);
    double fail = 0.0;
    int rc = 0, same = 0;
    std::vector<double> x_first;
    for (unsigned seed = seed0; seed < seed0 + nseeds; ++seed) {
        std::fill(PUB.begin(), PUB.end(), 0.0);
        std::fill(x.begin(), x.end(), 0.0);
        double* YU = PUB.data() + 2 * (size_t)S.nt * S.nbc * kSlabs * kSlabDoubles;
        unsigned epoch = seed;
        unsigned long long sw = 0;
        static unsigned long long head = 0;   // the launch's task counter (round 6): a workgroup's task is the number it draws, not its block index
        auto task = [&]() {
            d_chol_tiles<false>(blockIdx.x, S.A.data(), PUB.data(), YU, S.ld, S.n, S.nbc, S.plan.tasks.data(), S.plan.deps.data(), nullptr, flagA.data(), flagR.data(),
                                &epoch, &fail, nullptr, nullptr, x.data(), nullptr, &head, ntask);
            WAIT_VM();      // (the end of a wave completes what it has in flight)
        };
        if (resident > 0)       // the whole launch side by side: `resident` workgroups in flight on OS threads, dispatched in index order
            waveemu::run_grid(256, ntask, resident, seed, task);
        else
            for (int t = 0; t < ntask; ++t) sw += waveemu::run_group(256, t, ntask, seed * 7919u + t, task);
        const double res = chol_residual(S, x);
        std::printf("interleaving %u: %llu switches, |A x - b|_inf / |b|_inf = %.3e %s, failure flag %g\n", seed, sw, res, res < 1e-11 ? "(ok)" : "(MISMATCH)", fail);
        if (indefinite) {
            std::printf("  (indefinite system: the failure flag is %s)\n", fail == 1.0 ? "raised, as it must be" : "NOT raised");
            return fail == 1.0 ? 0 : 1;
        }
        if (!(res < 1e-11) || fail != 0.0) rc = 1;
        // the order in which a task takes its slabs and columns is fixed by the plan, not by the schedule: every interleaving must give
        // the same BITS
        if (x_first.empty()) x_first = x;
        else if (std::memcmp(x_first.data(), x.data(), x.size() * sizeof(double)) == 0) ++same;
        else { std::printf("interleaving %u: the solution differs in its bits from the first interleaving's\n", seed); rc = 1; }
    }
    std::printf("%u interleavings, %d of %u bit-identical to the first\n", nseeds, same, nseeds - 1);
    return rc;
}
