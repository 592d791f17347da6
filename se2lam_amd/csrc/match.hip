// libse2gpu - ORB matchers on gfx950 (MI355X).
//
// Replaces se2lam::ORBmatcher (/root/reference/src/ORBmatcher.cpp, include/se2lam/ORBmatcher.h:42-80):
//   DescriptorDistance :110-126, ComputeThreeMaxima :64-105, MatchByWindow :278-381, MatchByProjection :383-454
// and the Frame grid they search (/root/reference/src/Frame.cpp:209-286: PosInGrid uses round(), GetFeaturesInArea
// returns indices in (cell x, cell y, insertion) order; 64 x 48 cells, include/se2lam/Frame.h:26-27).
//
// The reference's matchers are order-dependent greedy passes: query i sees vMatchesDistance as left by queries
// 0..i-1 and ties go to the earliest candidate.  The split used here keeps that exactly:
//   k_grid_order     per target frame: features sorted by (cell x, cell y, index) = the order GetFeaturesInArea
//                    enumerates them (LDS bitonic sort of 28-bit keys) + the start of every grid column in that list
//   k_cand_*         one wave per query: window / level / square test over the grid-column slice of the sorted list
//                    that the search window covers, wave-ballot ordered
//                    compaction, 256-bit Hamming distance (4 x popcll) -> per-query candidate list (idx, dist)
//                    - the parallel O(N1*N2) part
//   k_resolve_*      one workgroup per frame pair: reproduces the greedy pass on the pre-computed lists as a fixed-point
//                    iteration over ALL queries at once (exactly the sequential result, see k_resolve_window);
//                    rotation histogram, ComputeThreeMaxima, prev-matched update; parallel across pairs
// Compiled with -ffp-contract=off (float grid / projection arithmetic must round as the reference's).
#include <algorithm>
#include <climits>
#include <cstdlib>
#include <mutex>
#include <cmath>

#include "common.h"

using namespace se2gpu;

namespace {

constexpr int kGridCols = 64, kGridRows = 48;
constexpr int kThHigh = 100, kThLow = 75, kHisto = 30;
constexpr int kMaxCand = 128;     // candidates kept per query
constexpr int kMaxFeat = 8192;    // features per frame the LDS sort can hold

struct Bounds {
    float min_x, min_y, max_x, max_y, wInv, hInv;
};

// per-frame grid record: [0] = number of features in the grid, [1 + cx] = first position of grid column cx in the
// sorted list (cx = 0 .. kGridCols; the last one = the number again)
constexpr int kGridRec = kGridCols + 2;

__device__ __forceinline__ int hamming256(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b) {
    const unsigned long long* pa = (const unsigned long long*)a;
    const unsigned long long* pb = (const unsigned long long*)b;
    return __popcll(pa[0] ^ pb[0]) + __popcll(pa[1] ^ pb[1]) + __popcll(pa[2] ^ pb[2]) + __popcll(pa[3] ^ pb[3]);
}

// ---------------------------------------------------------------------------------------------
// k_grid_order: one workgroup per frame.  sorted[f][i] = (cell << 16) | idx for the features that fall in the grid,
// ascending (cell = ix * 48 + iy); n_grid[f] = their number.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_grid_order(Bounds bd, const se2gpu_keypoint* __restrict__ kps,
                                                     const int* __restrict__ counts, const int* __restrict__ frames,
                                                     int cap, uint32_t* __restrict__ sorted, int* __restrict__ n_grid) {
    __shared__ uint32_t keys[kMaxFeat];
    __shared__ int s_n;
    const int f = frames ? frames[blockIdx.x] : blockIdx.x;
    const int n = min(counts[f], min(cap, kMaxFeat));
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    int npad = 1;
    while (npad < n) npad <<= 1;
    for (int i = threadIdx.x; i < npad; i += 256) {
        uint32_t key = 0xffffffffu;
        if (i < n) {
            const se2gpu_keypoint kp = kps[(size_t)f * cap + i];
            const int px = (int)roundf((kp.x - bd.min_x) * bd.wInv);
            const int py = (int)roundf((kp.y - bd.min_y) * bd.hInv);
            if (px >= 0 && px < kGridCols && py >= 0 && py < kGridRows) {
                key = ((uint32_t)(px * kGridRows + py) << 16) | (uint32_t)i;
                atomicAdd(&s_n, 1);
            }
        }
        keys[i] = key;
    }
    __syncthreads();
    for (int k = 2; k <= npad; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < npad; i += 256) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const uint32_t a = keys[i], b = keys[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < n; i += 256) sorted[(size_t)f * cap + i] = keys[i];
    const int ng = s_n;
    if (threadIdx.x == 0) n_grid[(size_t)f * kGridRec] = ng;
    if (threadIdx.x <= kGridCols) {   // lower bound of column cx in the sorted keys (invalid keys 0xffffffff sort last)
        const uint32_t want = (uint32_t)(threadIdx.x * kGridRows) << 16;
        int lo = 0, hi = ng;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (keys[mid] < want) lo = mid + 1; else hi = mid;
        }
        n_grid[(size_t)f * kGridRec + 1 + threadIdx.x] = lo;
    }
}

// The target frame's grid-ordered list and the three key-point fields the window test reads, staged in LDS once per
// workgroup (16 B per feature): a query's scan then waits on LDS instead of on two dependent global loads.
struct TargetLds {
    const uint32_t* sorted;
    const float* x;
    const float* y;
    const int* octave;
};
// queries per workgroup: 64 (16 per wave) when there are many pairs to fill the chip with, 8 for a single pair
constexpr int kCandQueriesBatch = 64, kCandQueriesSingle = 8;
__device__ __forceinline__ TargetLds stage_target(int* lds, int cap, const se2gpu_keypoint* __restrict__ kps2,
                                                  const uint32_t* __restrict__ sorted2, int n2) {
    uint32_t* s_sorted = (uint32_t*)lds;
    float* s_x = (float*)(lds + cap);
    float* s_y = (float*)(lds + 2 * cap);
    int* s_oct = lds + 3 * cap;
    for (int i = threadIdx.x; i < n2; i += blockDim.x) {
        s_sorted[i] = sorted2[i];
        const se2gpu_keypoint kp = kps2[i];
        s_x[i] = kp.x;
        s_y[i] = kp.y;
        s_oct[i] = kp.octave;
    }
    __syncthreads();
    return TargetLds{s_sorted, s_x, s_y, s_oct};
}

// Candidate scan of one query by one wave (Frame::GetFeaturesInArea + DescriptorDistance).
// Returns the number of candidates (may exceed kMaxCand: the caller flags overflow); out[s] = (idx << 12) | dist.
__device__ __forceinline__ int scan_candidates(const Bounds& bd, float x, float y, float r, int minLevel, int maxLevel,
                                               const uint8_t* __restrict__ d1, const TargetLds& tg,
                                               const uint8_t* __restrict__ desc2, const int* __restrict__ grid2,
                                               const uint8_t* __restrict__ excl, uint32_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    int nMinCellX = (int)floorf((x - bd.min_x - r) * bd.wInv);
    nMinCellX = max(0, nMinCellX);
    if (nMinCellX >= kGridCols) return 0;
    int nMaxCellX = (int)ceilf((x - bd.min_x + r) * bd.wInv);
    nMaxCellX = min(kGridCols - 1, nMaxCellX);
    if (nMaxCellX < 0) return 0;
    int nMinCellY = (int)floorf((y - bd.min_y - r) * bd.hInv);
    nMinCellY = max(0, nMinCellY);
    if (nMinCellY >= kGridRows) return 0;
    int nMaxCellY = (int)ceilf((y - bd.min_y + r) * bd.hInv);
    nMaxCellY = min(kGridRows - 1, nMaxCellY);
    if (nMaxCellY < 0) return 0;
    const bool checkLevels = !(minLevel == -1 && maxLevel == -1);
    int count = 0;
    // the list is sorted by grid column first: only the slice of columns nMinCellX .. nMaxCellX is visited
    const int cbeg = grid2[1 + nMinCellX], cend = grid2[2 + nMaxCellX];
    for (int c0 = cbeg; c0 < cend; c0 += 64) {
        const int pos = c0 + lane;
        bool ok = false;
        int idx = 0;
        if (pos < cend) {
            const uint32_t pk = tg.sorted[pos];
            const int cell = (int)(pk >> 16);
            idx = (int)(pk & 0xffffu);
            const int cx = cell / kGridRows, cy = cell % kGridRows;
            if (cx >= nMinCellX && cx <= nMaxCellX && cy >= nMinCellY && cy <= nMaxCellY) {
                const int oct = tg.octave[idx];
                ok = (!checkLevels || (oct >= minLevel && oct <= maxLevel)) && !(fabsf(tg.x[idx] - x) > r) &&
                     !(fabsf(tg.y[idx] - y) > r) && !(excl && excl[idx]);
            }
        }
        const unsigned long long m = __ballot(ok);
        if (ok) {
            const int slot = count + __popcll(m & ((1ull << lane) - 1ull));
            if (slot < kMaxCand) out[slot] = ((uint32_t)idx << 12) | (uint32_t)hamming256(d1, desc2 + 32 * (size_t)idx);
        }
        count += __popcll(m);
    }
    return count;
}

// MatchByWindow candidates: query i1 of pair p = frame a's key point, searched around prev_xy in frame b.
__global__ __launch_bounds__(256) void k_cand_window(Bounds bd, const se2gpu_keypoint* __restrict__ kps,
                                                      const uint8_t* __restrict__ desc, const int* __restrict__ counts,
                                                      int cap, const int* __restrict__ pair_a,
                                                      const int* __restrict__ pair_b, const float* __restrict__ prev_xy,
                                                      const uint32_t* __restrict__ sorted, const int* __restrict__ n_grid,
                                                      int win, int level_offset, int min_level, int max_level,
                                                      uint32_t* __restrict__ cand, int* __restrict__ ncand, int npairs, int qpb) {
    // pairs are the fast grid dimension (padded to a multiple of 8): a pair's workgroups share an XCD and its L2
    extern __shared__ __attribute__((aligned(16))) int lds[];
    const int p = blockIdx.x;
    if (p >= npairs) return;
    const int fa = pair_a[p], fb = pair_b[p];
    const int n1 = min(counts[fa], cap), n2 = min(counts[fb], cap);
    const int q0 = blockIdx.y * qpb;
    if (q0 >= n1) return;
    const TargetLds tg = stage_target(lds, cap, kps + (size_t)fb * cap, sorted + (size_t)fb * cap, n2);
    // (one wave per query: the query index is wave-uniform - said so, its key point, previous position and descriptor address are scalar)
    for (int i1 = q0 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x / 64)); i1 < min(q0 + qpb, n1); i1 += 4) {
        const se2gpu_keypoint kp1 = kps[(size_t)fa * cap + i1];
        const int level1 = kp1.octave;
        int n = 0;
        if (!(level1 > max_level || level1 < min_level)) {
            const int minLevel2 = level1 - level_offset > 0 ? level1 - level_offset : 0;
            n = scan_candidates(bd, prev_xy[((size_t)p * cap + i1) * 2], prev_xy[((size_t)p * cap + i1) * 2 + 1], (float)win,
                                minLevel2, level1 + level_offset, desc + ((size_t)fa * cap + i1) * 32, tg,
                                desc + (size_t)fb * cap * 32, n_grid + (size_t)fb * kGridRec, nullptr,
                                cand + ((size_t)p * cap + i1) * kMaxCand);
        }
        if ((threadIdx.x & 63) == 0) ncand[(size_t)p * cap + i1] = n;
    }
}

// wave-uniform arg-min over the lanes with `valid`: smallest dist, then lowest lane.  Returns lane or -1.
__device__ __forceinline__ int wave_argmin(bool valid, int dist) {
    unsigned long long m = __ballot(valid);
    if (m == 0) return -1;
#pragma unroll
    for (int b = 8; b >= 0; --b) {
        const unsigned long long z = __ballot(valid && (((dist >> b) & 1) == 0)) & m;
        if (z) m = z;
    }
    return __ffsll((long long)m) - 1;
}

struct Best2 {
    int d1, p1, d2, p2;  // best (dist, position), second best; INT_MAX / -1 when absent
};
__device__ __forceinline__ void best2_push(Best2& b, int d, int p) {  // (d, p) ordering, p increasing across pushes
    if (d < b.d1 || (d == b.d1 && p < b.p1)) {
        b.d2 = b.d1; b.p2 = b.p1;
        b.d1 = d; b.p1 = p;
    } else if (d < b.d2 || (d == b.d2 && p < b.p2)) {
        b.d2 = d; b.p2 = p;
    }
}

// Stage the candidate lists of `nq` queries (fixed stride kMaxCand in HBM) into LDS as one compact array so that
// the greedy pass never waits on HBM.  off[0..nq] = exclusive prefix of min(ncand, kMaxCand); entries that
// do not fit in `cap_e` stay in HBM (the pass falls back to global reads for those queries).  Whole workgroup; the
// prefix is taken by the first wave.
__device__ __forceinline__ void stage_candidates(const uint32_t* __restrict__ cand, const int* __restrict__ ncand, int nq,
                                                 int* off, uint32_t* ce, int cap_e, int* __restrict__ overflow) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    if (tid < 64) {
        const int lane = tid;
        const int per = (nq + 63) / 64;
        int local = 0;
        for (int i = lane * per; i < min(nq, (lane + 1) * per); ++i) {
            int n = ncand[i];
            if (n > kMaxCand) { atomicOr(overflow, 1); n = kMaxCand; }   // (informational: such queries take the spill scan)
            local += n;
        }
        int incl = local;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(incl, d);
            if (lane >= d) incl += v;
        }
        int run = incl - local;
        for (int i = lane * per; i < min(nq, (lane + 1) * per); ++i) {
            off[i] = run;
            run += min(ncand[i], kMaxCand);
        }
        if (lane == 63) off[nq] = incl;
    }
    __syncthreads();
    const int tcopy = min(off[nq], cap_e);
    for (int e = tid; e < tcopy; e += nthr) {
        int lo = 0, hi = nq;  // largest i with off[i] <= e
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (off[mid] <= e) lo = mid; else hi = mid;
        }
        ce[e] = cand[(size_t)lo * kMaxCand + (e - off[lo])];
    }
    __syncthreads();
}

// (L bins; the indices keep the caller's values where no bin qualifies, as ORBmatcher::ComputeThreeMaxima leaves them)
__host__ __device__ __forceinline__ void three_maxima_n(const int* hist, int L, int& ind1, int& ind2, int& ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = hist[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}
__device__ __forceinline__ void three_maxima(const int* hist, int& ind1, int& ind2, int& ind3) {
    ind1 = ind2 = ind3 = -1;
    three_maxima_n(hist, kHisto, ind1, ind2, ind3);
}

// Best / second-best scan of the candidate list of query q (ORBmatcher.cpp:308-325; ties go to the earliest list
// position) against the EFFECTIVE vMatchesDistance the sequential pass would see when it reaches q: the committed value
// (earlier chunks) lowered by the tentative acceptances of the queries below q.  The tentative acceptors of a target
// form a linked list (head[target] -> next[query] -> ...), rebuilt before every sweep; accD[query] is the distance a
// query accepts its target with.
struct LaneBest {
    int bd, bp, bd2, bp2;   // best / second-best distance and list position (-1: none)
    int bi, bi2;            // ... and the target features they belong to
};
__device__ __forceinline__ void lane_best_push(LaneBest& b, int dist, int t, int idx) {
    if (dist < b.bd) { b.bd2 = b.bd; b.bp2 = b.bp; b.bi2 = b.bi; b.bd = dist; b.bp = t; b.bi = idx; }
    else if (dist < b.bd2) { b.bd2 = dist; b.bp2 = t; b.bi2 = idx; }
}
__device__ __forceinline__ LaneBest query_scan(const uint32_t* cl, int n, int q, const int* vMatchesDistance,
                                               const int* head, const int* next, const int* accD) {
    LaneBest b{INT_MAX, -1, INT_MAX, -1, -1, -1};
    for (int t = 0; t < n; ++t) {
        const uint32_t pk = cl[t];
        const int dist = (int)(pk & 0xfffu), idx = (int)(pk >> 12);
        int eff = vMatchesDistance ? vMatchesDistance[idx] : INT_MAX;
        for (int a = head[idx]; a >= 0; a = next[a])
            if (a < q) eff = min(eff, accD[a]);
        if (!(eff <= dist)) lane_best_push(b, dist, t, idx);
    }
    return b;
}

// The same scan for a query whose search window holds more candidates than a list keeps (kMaxCand): the candidates are
// enumerated again from the target frame's grid order - the cells, the level band, the square test and the Hamming
// distance exactly as scan_candidates takes them, in the same order - by the query's own thread, every sweep.  Slow (a
// clustered frame: hundreds of key points inside one 40 x 40 window) but exact: the reference has no such limit
// (ORBmatcher.cpp:298-325 walks whatever GetFeaturesInArea returns), and a tracker must not lose the frame over it.
struct SpillTarget {
    Bounds bd;
    const uint32_t* sorted;          // the target frame's grid order (cell << 16 | feature)
    const int* grid;                 // its column offsets (kGridRec ints)
    const se2gpu_keypoint* kps;      // target key points
    const uint8_t* desc;             // target descriptors
    const uint8_t* excl;             // features to skip (KeyFrame::hasObservation), or nullptr
};
// rec (nullable): the (feature, distance) pairs are written there in enumeration order, rec_cap entries at most - the scan of
// the first sweep feeds the replay of the later ones (query_replay_spill)
__device__ inline LaneBest query_scan_spill(const SpillTarget& tg, float x, float y, float r, int minLevel, int maxLevel,
                                            const uint8_t* __restrict__ d1, int q, const int* vMatchesDistance,
                                            const int* head, const int* next, const int* accD, uint2* __restrict__ rec = nullptr,
                                            int rec_cap = 0, int* rec_count = nullptr) {
    LaneBest b{INT_MAX, -1, INT_MAX, -1, -1, -1};
    if (rec_count) *rec_count = 0;
    const Bounds& bd = tg.bd;
    int nMinCellX = (int)floorf((x - bd.min_x - r) * bd.wInv);
    nMinCellX = max(0, nMinCellX);
    if (nMinCellX >= kGridCols) return b;
    int nMaxCellX = (int)ceilf((x - bd.min_x + r) * bd.wInv);
    nMaxCellX = min(kGridCols - 1, nMaxCellX);
    if (nMaxCellX < 0) return b;
    int nMinCellY = (int)floorf((y - bd.min_y - r) * bd.hInv);
    nMinCellY = max(0, nMinCellY);
    if (nMinCellY >= kGridRows) return b;
    int nMaxCellY = (int)ceilf((y - bd.min_y + r) * bd.hInv);
    nMaxCellY = min(kGridRows - 1, nMaxCellY);
    if (nMaxCellY < 0) return b;
    const bool checkLevels = !(minLevel == -1 && maxLevel == -1);
    const int cbeg = tg.grid[1 + nMinCellX], cend = tg.grid[2 + nMaxCellX];
    int t = 0;
    for (int pos = cbeg; pos < cend; ++pos) {
        const uint32_t pk = tg.sorted[pos];
        const int cell = (int)(pk >> 16), idx = (int)(pk & 0xffffu);
        const int cx = cell / kGridRows, cy = cell % kGridRows;
        if (!(cx >= nMinCellX && cx <= nMaxCellX && cy >= nMinCellY && cy <= nMaxCellY)) continue;
        const se2gpu_keypoint kp = tg.kps[idx];
        const bool ok = (!checkLevels || (kp.octave >= minLevel && kp.octave <= maxLevel)) && !(fabsf(kp.x - x) > r) &&
                        !(fabsf(kp.y - y) > r) && !(tg.excl && tg.excl[idx]);
        if (!ok) continue;
        const int dist = hamming256(d1, tg.desc + 32 * (size_t)idx);
        if (rec && t < rec_cap) rec[t] = make_uint2((uint32_t)idx, (uint32_t)dist);
        int eff = vMatchesDistance ? vMatchesDistance[idx] : INT_MAX;
        for (int a = head[idx]; a >= 0; a = next[a])
            if (a < q) eff = min(eff, accD[a]);
        if (!(eff <= dist)) lane_best_push(b, dist, t, idx);
        ++t;
    }
    if (rec_count) *rec_count = t;
    return b;
}
// The later sweeps of a spilled query: its candidates do not change between sweeps (the grid, the window and the distances are
// fixed), only the tentative acceptances they are weighed against do - so the list recorded by the first sweep is replayed
// instead of walking the grid and recomputing every Hamming distance again (the re-scan per sweep).
__device__ inline LaneBest query_replay_spill(const uint2* __restrict__ rec, int cnt, int q, const int* vMatchesDistance,
                                              const int* head, const int* next, const int* accD) {
    LaneBest b{INT_MAX, -1, INT_MAX, -1, -1, -1};
    for (int t = 0; t < cnt; ++t) {
        const uint2 e = rec[t];
        const int idx = (int)e.x, dist = (int)e.y;
        int eff = vMatchesDistance ? vMatchesDistance[idx] : INT_MAX;
        for (int a = head[idx]; a >= 0; a = next[a])
            if (a < q) eff = min(eff, accD[a]);
        if (!(eff <= dist)) lane_best_push(b, dist, t, idx);
    }
    return b;
}
// A spilled query's place in its workgroup's spill arena: taken once, in the first sweep (an LDS bump allocator; ncand is the
// exact number of candidates the scan will enumerate); {-1, 0} when the arena is full - the query then re-scans every sweep
__device__ inline LaneBest query_spill(const SpillTarget& tg, float x, float y, float r, int minLevel, int maxLevel,
                                       const uint8_t* __restrict__ d1, int q, const int* vMatchesDistance, const int* head,
                                       const int* next, const int* accD, int sweep, int need, uint2* __restrict__ arena, int arena_cap,
                                       int* used, int2* __restrict__ place) {
    if (sweep == 0) {
        int pos = -1;
        if (arena) {
            pos = atomicAdd(used, need);
            if (pos + need > arena_cap) pos = -1;
        }
        int cnt = 0;
        const LaneBest b = query_scan_spill(tg, x, y, r, minLevel, maxLevel, d1, q, vMatchesDistance, head, next, accD,
                                            pos >= 0 ? arena + pos : nullptr, need, &cnt);
        *place = (pos >= 0 && cnt <= need) ? make_int2(pos, cnt) : make_int2(-1, 0);
        return b;
    }
    const int2 pl = *place;
    if (pl.x >= 0) return query_replay_spill(arena + pl.x, pl.y, q, vMatchesDistance, head, next, accD);
    return query_scan_spill(tg, x, y, r, minLevel, maxLevel, d1, q, vMatchesDistance, head, next, accD);
}

// MatchByWindow greedy pass (ORBmatcher.cpp:292-377), one workgroup per pair, parallel over ALL queries.
// The reference processes queries in index order; query i depends on earlier queries only through
// vMatchesDistance[idx] of its own candidates.  The whole pass is iterated to a FIXED POINT: in every sweep each query
// recomputes (best, second best, accept?) against the tentative acceptances the queries below it held in the previous
// sweep.  Query 0 is final after sweep 1, and a query is final one sweep after all queries below it are, so any fixed
// point is exactly the sequential result (induction on the query index); dependency chains are short, a handful of
// sweeps suffices in practice (worst case n1 + 1).  Afterwards the highest accepting query of a target owns it (the
// sequential eviction chain), every accept feeds the rotation histogram.
// Dynamic LDS (ints, capE each): head next accT accD accT2 accD2 bin_of m12 ang1 ang2 | hist[32] ind[4] | off[capE + 4]
//                                | staged candidate entries.
__global__ __launch_bounds__(1024) void k_resolve_window(const se2gpu_keypoint* __restrict__ kps,
                                                          const int* __restrict__ counts, int cap,
                                                          const int* __restrict__ pair_a, const int* __restrict__ pair_b,
                                                          const uint32_t* __restrict__ cand, const int* __restrict__ ncand,
                                                          float nnratio, int cand_lds, int* __restrict__ matches12,
                                                          float* __restrict__ prev_xy, int* __restrict__ nmatches,
                                                          int* __restrict__ overflow, Bounds bd,
                                                          const uint8_t* __restrict__ desc, const uint32_t* __restrict__ sorted,
                                                          const int* __restrict__ n_grid, int win, int level_offset,
                                                          int min_level, int max_level, uint2* __restrict__ spill, int spill_cap,
                                                          int2* __restrict__ spill_place) {
    extern __shared__ __attribute__((aligned(16))) int lds[];
    __shared__ int s_spill_used;
    if (threadIdx.x == 0) s_spill_used = 0;
    const int p = blockIdx.x;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int fa = pair_a[p], fb = pair_b[p];
    const int n1 = min(counts[fa], cap), n2 = min(counts[fb], cap);
    const int capE = (cap + 3) & ~3;
    int* head = lds;
    int* next = lds + capE;
    int* accT = lds + 2 * capE;
    int* accD = lds + 3 * capE;
    int* accT2 = lds + 4 * capE;
    int* accD2 = lds + 5 * capE;
    int* bin_of = lds + 6 * capE;
    int* m12 = lds + 7 * capE;
    float* ang1 = (float*)(lds + 8 * capE);
    float* ang2 = (float*)(lds + 9 * capE);
    int* hist = lds + 10 * capE;                     // 32
    int* s_ind = lds + 10 * capE + 32;               // 4
    int* off = lds + 10 * capE + 36;                 // cap + 1 entries
    uint32_t* ce = (uint32_t*)(lds + 11 * capE + 40);
    const se2gpu_keypoint* k1 = kps + (size_t)fa * cap;
    const se2gpu_keypoint* k2 = kps + (size_t)fb * cap;
    for (int i = tid; i < n2; i += nthr) ang2[i] = k2[i].angle;
    for (int i = tid; i < n1; i += nthr) { bin_of[i] = -1; m12[i] = -1; accT[i] = -1; accD[i] = 0; ang1[i] = k1[i].angle; }
    if (tid < kHisto) hist[tid] = 0;
    const uint32_t* cand_p = cand + (size_t)p * cap * kMaxCand;
    stage_candidates(cand_p, ncand + (size_t)p * cap, n1, off, ce, cand_lds, overflow);
    for (int sweep = 0; sweep <= n1 + 1; ++sweep) {
        for (int t = tid; t < n2; t += nthr) head[t] = -1;
        __syncthreads();
        for (int q = tid; q < n1; q += nthr)
            if (accT[q] >= 0) next[q] = atomicExch(&head[accT[q]], q);
        __syncthreads();
        int changed = 0;
        for (int q = tid; q < n1; q += nthr) {
            const int o0 = off[q], n = off[q + 1] - o0;
            const uint32_t* cl = (o0 + n <= cand_lds) ? ce + o0 : cand_p + (size_t)q * kMaxCand;
            LaneBest b;
            if (ncand[(size_t)p * cap + q] > kMaxCand) {   // more candidates than the list keeps: exact scan of the grid
                const int level1 = k1[q].octave;
                const int minLevel2 = level1 - level_offset > 0 ? level1 - level_offset : 0;
                const SpillTarget tg{bd, sorted + (size_t)fb * cap, n_grid + (size_t)fb * kGridRec, k2, desc + (size_t)fb * cap * 32, nullptr};
                b = query_spill(tg, prev_xy[((size_t)p * cap + q) * 2], prev_xy[((size_t)p * cap + q) * 2 + 1], (float)win,
                                minLevel2, level1 + level_offset, desc + ((size_t)fa * cap + q) * 32, q, nullptr, head, next, accD, sweep,
                                ncand[(size_t)p * cap + q], spill ? spill + (size_t)p * spill_cap : nullptr, spill_cap, &s_spill_used,
                                spill_place + (size_t)p * cap + q);
            } else {
                b = query_scan(cl, n, q, nullptr, head, next, accD);
            }
            const bool nacc = b.bp >= 0 && b.bd <= kThLow && (float)b.bd < (float)b.bd2 * nnratio;
            const int nidx = nacc ? b.bi : -1;
            const int nd = nacc ? b.bd : 0;
            changed |= (nidx != accT[q]) | (nd != accD[q]);
            accT2[q] = nidx;
            accD2[q] = nd;
        }
        if (!__syncthreads_or(changed)) break;       // fixed point: accT / accD and the lists built from them are final
        int* t0 = accT; accT = accT2; accT2 = t0;
        int* t1 = accD; accD = accD2; accD2 = t1;
    }
    // commit
    const float factor = (float)kHisto / 360.0f;
    for (int q = tid; q < n1; q += nthr) {
        const int t = accT[q];
        if (t < 0) continue;
        float rot = ang1[q] - ang2[t];
        if (rot < 0.0f) rot += 360.f;
        int bin = (int)roundf(rot * factor);
        if (bin == kHisto) bin = 0;
        bin_of[q] = bin;
        atomicAdd(&hist[bin], 1);
        int owner = -1;
        for (int a = head[t]; a >= 0; a = next[a]) owner = max(owner, a);
        if (owner == q) m12[q] = t;                   // the last accepting query keeps the target, the others were evicted
    }
    __syncthreads();
    if (tid == 0) { three_maxima(hist, s_ind[0], s_ind[1], s_ind[2]); s_ind[3] = 0; }
    __syncthreads();
    int cnt = 0;
    for (int i = tid; i < n1; i += nthr) {
        const int bin = bin_of[i];
        int m = m12[i];
        if (bin >= 0 && bin != s_ind[0] && bin != s_ind[1] && bin != s_ind[2] && m >= 0) m = -1;
        if (m >= 0) {
            ++cnt;
            prev_xy[((size_t)p * cap + i) * 2] = k2[m].x;
            prev_xy[((size_t)p * cap + i) * 2 + 1] = k2[m].y;
        }
        matches12[(size_t)p * cap + i] = m;
    }
    for (int sft = 1; sft < 64; sft <<= 1) cnt += __shfl_xor(cnt, sft);
    if ((tid & 63) == 0 && cnt) atomicAdd(&s_ind[3], cnt);
    __syncthreads();
    if (tid == 0) nmatches[p] = s_ind[3];
}

// prev_xy of pair p = key point positions of frame a (Track::resetLocalTrack, Track.cpp:194)
__global__ void k_init_prev(const se2gpu_keypoint* __restrict__ kps, const int* __restrict__ counts, int cap,
                            const int* __restrict__ pair_a, float* __restrict__ prev_xy) {
    const int p = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int fa = pair_a[p];
    if (i >= min(counts[fa], cap)) return;
    prev_xy[((size_t)p * cap + i) * 2] = kps[(size_t)fa * cap + i].x;
    prev_xy[((size_t)p * cap + i) * 2 + 1] = kps[(size_t)fa * cap + i].y;
}

// MatchByProjection candidates: query i = map point, projected with cvu::camprjc(K, cvu::se3map(Tcw, pos)).
struct ProjCam {
    float T[12];
    float fx, fy, cx, cy;
};
__global__ __launch_bounds__(256) void k_cand_projection(Bounds bd, ProjCam cam, const float* __restrict__ mp_pos,
                                                          const uint8_t* __restrict__ mp_desc,
                                                          const int* __restrict__ mp_octave,
                                                          const uint8_t* __restrict__ mp_skip, int m,
                                                          const se2gpu_keypoint* __restrict__ kps,
                                                          const uint8_t* __restrict__ desc,
                                                          const uint8_t* __restrict__ kf_observed,
                                                          const uint32_t* __restrict__ sorted,
                                                          const int* __restrict__ n_grid, int win, int level_offset,
                                                          uint32_t* __restrict__ cand, int* __restrict__ ncand, int n_feat, int qpb,
                                                          float* __restrict__ proj_xy) {
    extern __shared__ __attribute__((aligned(16))) int lds[];
    const int q0 = blockIdx.x * qpb;
    const TargetLds tg = stage_target(lds, n_feat, kps, sorted, n_feat);
    for (int i = q0 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x / 64)); i < min(q0 + qpb, m); i += 4) {   // wave-uniform map point
        int n = 0;
        if (!mp_skip[i]) {
            const float X = mp_pos[3 * i], Y = mp_pos[3 * i + 1], Z = mp_pos[3 * i + 2];
            float pc[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                float s = 0;
                s += cam.T[4 * r + 0] * X;
                s += cam.T[4 * r + 1] * Y;
                s += cam.T[4 * r + 2] * Z;
                pc[r] = s + cam.T[4 * r + 3];
            }
            float u = 0, v = 0, w = 0;
            u += cam.fx * pc[0]; u += 0.f * pc[1]; u += cam.cx * pc[2];
            v += 0.f * pc[0]; v += cam.fy * pc[1]; v += cam.cy * pc[2];
            w += 0.f * pc[0]; w += 0.f * pc[1]; w += 1.f * pc[2];
            const float px = u / w, py = v / w;
            if ((threadIdx.x & 63) == 0) { proj_xy[2 * (size_t)i] = px; proj_xy[2 * (size_t)i + 1] = py; }   // (for the spill scan)
            if (px >= bd.min_x && px <= bd.max_x && py >= bd.min_y && py <= bd.max_y) {
                const int predictLevel = mp_octave[i];
                const int levelWinSize = predictLevel * win;
                const int minLevel = predictLevel > level_offset ? predictLevel - level_offset : 0;
                n = scan_candidates(bd, px, py, (float)levelWinSize, minLevel, predictLevel + level_offset,
                                    mp_desc + 32 * (size_t)i, tg, desc, n_grid, kf_observed, cand + (size_t)i * kMaxCand);
            }
        }
        if ((threadIdx.x & 63) == 0) ncand[i] = n;
    }
}

// MatchByProjection greedy pass (ORBmatcher.cpp:390-451), one workgroup, parallel over map points: the same fixed-point
// scheme as k_resolve_window, applied to chunks of `chunk` map points (the candidate lists of a chunk are staged in LDS);
// a chunk starts from the vMatchesDistance the earlier chunks committed.  On commit the highest accepting map point of a
// feature owns vMatchesIdxMP[feature] (the reference overwrites the earlier owner) and leaves its - smallest - distance.
// Dynamic LDS (ints): head[nE] vMatchesDistance[nE] octave[nE] | next accT accD accT2 accD2 [chunk each] |
//                     off[chunk + 4] | staged candidate entries.
__global__ __launch_bounds__(1024) void k_resolve_projection(const se2gpu_keypoint* __restrict__ kps, int n, int m,
                                                              const uint32_t* __restrict__ cand,
                                                              const int* __restrict__ ncand, float nnratio,
                                                              int chunk, int cand_lds, int* __restrict__ match_idx,
                                                              int* __restrict__ nmatches, int* __restrict__ overflow, Bounds bd,
                                                              const uint8_t* __restrict__ desc,
                                                              const uint8_t* __restrict__ kf_observed,
                                                              const uint32_t* __restrict__ sorted,
                                                              const int* __restrict__ n_grid, const float* __restrict__ proj_xy,
                                                              const uint8_t* __restrict__ mp_desc,
                                                              const int* __restrict__ mp_octave, int win, int level_offset,
                                                              uint2* __restrict__ spill, int spill_cap, int2* __restrict__ spill_place) {
    extern __shared__ __attribute__((aligned(16))) int lds[];
    __shared__ int s_cnt, s_spill_used;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int nE = (n + 3) & ~3;
    int* head = lds;
    int* vMatchesDistance = lds + nE;
    int* octave = lds + 2 * nE;
    int* next = lds + 3 * nE;
    int* accT = next + chunk;
    int* accD = accT + chunk;
    int* accT2 = accD + chunk;
    int* accD2 = accT2 + chunk;
    int* off = accD2 + chunk;                            // chunk + 1 entries
    uint32_t* ce = (uint32_t*)(off + chunk + 4);
    for (int i = tid; i < n; i += nthr) { vMatchesDistance[i] = INT_MAX; octave[i] = kps[i].octave; match_idx[i] = -1; }
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    for (int i0 = 0; i0 < m; i0 += chunk) {
        const int mq = min(chunk, m - i0);
        const uint32_t* cand_c = cand + (size_t)i0 * kMaxCand;
        stage_candidates(cand_c, ncand + i0, mq, off, ce, cand_lds, overflow);
        for (int q = tid; q < mq; q += nthr) { accT[q] = -1; accD[q] = 0; }
        if (tid == 0) s_spill_used = 0;   // (the arena is the chunk's; a barrier follows before the first use)
        for (int sweep = 0; sweep <= mq + 1; ++sweep) {
            for (int t = tid; t < n; t += nthr) head[t] = -1;
            __syncthreads();
            for (int q = tid; q < mq; q += nthr)
                if (accT[q] >= 0) next[q] = atomicExch(&head[accT[q]], q);
            __syncthreads();
            int changed = 0;
            for (int q = tid; q < mq; q += nthr) {
                const int o0 = off[q], nc = off[q + 1] - o0;
                const uint32_t* cl = (o0 + nc <= cand_lds) ? ce + o0 : cand_c + (size_t)q * kMaxCand;
                LaneBest b;
                if (ncand[i0 + q] > kMaxCand) {   // more candidates than the list keeps: exact scan of the key frame's grid
                    const int predictLevel = mp_octave[i0 + q];
                    const int minLevel = predictLevel > level_offset ? predictLevel - level_offset : 0;
                    const SpillTarget tg{bd, sorted, n_grid, kps, desc, kf_observed};
                    b = query_spill(tg, proj_xy[2 * (size_t)(i0 + q)], proj_xy[2 * (size_t)(i0 + q) + 1],
                                    (float)(predictLevel * win), minLevel, predictLevel + level_offset,
                                    mp_desc + 32 * (size_t)(i0 + q), q, vMatchesDistance, head, next, accD, sweep, ncand[i0 + q], spill,
                                    spill_cap, &s_spill_used, spill_place + i0 + q);
                } else {
                    b = query_scan(cl, nc, q, vMatchesDistance, head, next, accD);
                }
                bool nacc = b.bp >= 0 && b.bd <= kThHigh;
                int nidx = -1;
                if (nacc) {
                    nidx = b.bi;
                    const int bestLevel2 = b.bp2 >= 0 ? octave[b.bi2] : -1;
                    if (octave[nidx] == bestLevel2 && (float)b.bd > nnratio * (float)b.bd2) { nacc = false; nidx = -1; }
                }
                const int nd = nacc ? b.bd : 0;
                changed |= (nidx != accT[q]) | (nd != accD[q]);
                accT2[q] = nidx;
                accD2[q] = nd;
            }
            if (!__syncthreads_or(changed)) break;
            int* t0 = accT; accT = accT2; accT2 = t0;
            int* t1 = accD; accD = accD2; accD2 = t1;
        }
        for (int q = tid; q < mq; q += nthr) {           // commit the chunk
            const int t = accT[q];
            if (t < 0) continue;
            int owner = -1;
            for (int a = head[t]; a >= 0; a = next[a]) owner = max(owner, a);
            if (owner == q) {                            // the last accepting map point keeps the feature
                match_idx[t] = i0 + q;
                vMatchesDistance[t] = accD[q];
            }
        }
        __syncthreads();
    }
    int cnt = 0;
    for (int i = tid; i < n; i += nthr) cnt += match_idx[i] >= 0;
    for (int s2 = 1; s2 < 64; s2 <<= 1) cnt += __shfl_xor(cnt, s2);
    if ((tid & 63) == 0 && cnt) atomicAdd(&s_cnt, cnt);
    __syncthreads();
    if (tid == 0) nmatches[0] = s_cnt;
}

// ---------------------------------------------------------------------------------------------
// SearchByBoW (ORBmatcher.cpp:128-276).  A feature belongs to exactly one vocabulary node, so the vbMatched2
// dependency never crosses nodes: one wave per pair of equal nodes replays the reference's loop for that node
// (idx1 in list order; best / second best over the still unmatched idx2 of the node by wave-ballot arg-min);
// nodes run in parallel.  The rotation histogram is global (30 atomic counters).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_bow_match(const int2* __restrict__ node_pairs, const int* __restrict__ ptr1,
                                                   const int* __restrict__ idx1s, const int* __restrict__ ptr2,
                                                   const int* __restrict__ idx2s, const se2gpu_keypoint* __restrict__ kps1,
                                                   const uint8_t* __restrict__ desc1, const uint8_t* __restrict__ has1,
                                                   const se2gpu_keypoint* __restrict__ kps2,
                                                   const uint8_t* __restrict__ desc2, const uint8_t* __restrict__ has2,
                                                   int mp_only, float nnratio, int check_ori, int* __restrict__ matches12,
                                                   int* __restrict__ bin_of, int* __restrict__ hist) {
    extern __shared__ unsigned char matched2[];   // vbMatched2 of this node's features: sized by the host to the largest node
    const int lane = threadIdx.x;
    const int2 np = node_pairs[blockIdx.x];
    const int a0 = ptr1[np.x], a1 = ptr1[np.x + 1], b0 = ptr2[np.y], b1 = ptr2[np.y + 1];
    const int m2 = b1 - b0;
    for (int i = lane; i < m2; i += 64) matched2[i] = 0;
    __syncthreads();
    const float factor = (float)kHisto / 360.0f;
    for (int i1 = a0; i1 < a1; ++i1) {
        const int idx1 = idx1s[i1];
        if (mp_only && !has1[idx1]) continue;
        const uint8_t* d1 = desc1 + 32 * (size_t)idx1;
        Best2 b{INT_MAX, -1, INT_MAX, -1};
        for (int c0 = 0; c0 < m2; c0 += 64) {
            const int pos = c0 + lane;
            bool valid = false;
            int dist = 0;
            if (pos < m2) {
                const int idx2 = idx2s[b0 + pos];
                valid = !(mp_only && !has2[idx2]) && !matched2[pos];
                if (valid) dist = hamming256(d1, desc2 + 32 * (size_t)idx2);
            }
            const int l1 = wave_argmin(valid, dist);
            if (l1 < 0) continue;
            best2_push(b, __shfl(dist, l1), c0 + l1);
            const int l2 = wave_argmin(valid && lane != l1, dist);
            if (l2 >= 0) best2_push(b, __shfl(dist, l2), c0 + l2);
        }
        if (b.p1 >= 0 && b.d1 < kThLow && (float)b.d1 < nnratio * (float)b.d2) {
            if (lane == 0) {
                const int idx2 = idx2s[b0 + b.p1];
                matches12[idx1] = idx2;
                matched2[b.p1] = 1;
                if (check_ori) {
                    float rot = kps1[idx1].angle - kps2[idx2].angle;
                    if (rot < 0.0f) rot += 360.0f;
                    int bin = (int)roundf(rot * factor);
                    if (bin == kHisto) bin = 0;
                    bin_of[idx1] = bin;
                    atomicAdd(&hist[bin], 1);
                }
            }
        }
        __syncthreads();
    }
}

__global__ void k_bow_finish(int n1, int check_ori, const int* __restrict__ bin_of, const int* __restrict__ hist,
                             int* __restrict__ matches12, int* __restrict__ nmatches) {
    __shared__ int s_ind[3];
    __shared__ int s_cnt;
    if (threadIdx.x == 0) {
        s_cnt = 0;
        s_ind[0] = s_ind[1] = s_ind[2] = -1;
        if (check_ori) three_maxima(hist, s_ind[0], s_ind[1], s_ind[2]);
    }
    __syncthreads();
    int cnt = 0;
    for (int i = threadIdx.x; i < n1; i += blockDim.x) {
        int m = matches12[i];
        if (check_ori && m >= 0) {
            const int bin = bin_of[i];
            if (bin != s_ind[0] && bin != s_ind[1] && bin != s_ind[2]) { m = -1; matches12[i] = -1; }
        }
        cnt += m >= 0;
    }
    atomicAdd(&s_cnt, cnt);
    __syncthreads();
    if (threadIdx.x == 0) nmatches[0] = s_cnt;
}

Bounds make_bounds(const se2gpu_frame_bounds& b) {
    Bounds o;
    o.min_x = b.min_x; o.min_y = b.min_y; o.max_x = b.max_x; o.max_y = b.max_y;
    o.wInv = (float)kGridCols / (b.max_x - b.min_x);   // Frame.cpp:40-41
    o.hInv = (float)kGridRows / (b.max_y - b.min_y);
    return o;
}

}  // namespace

struct se2gpu_matcher {
    hipStream_t own_stream = nullptr, stream = nullptr;
    int max_features = 0, max_batch = 1, device = 0;
    // scratch for frame sets of up to `nframes_cap` frames with stride `cap_cur`
    DevBuf<uint32_t> sorted, cand;
    DevBuf<uint2> spill;          // (feature, distance) lists of the queries with more than kMaxCand candidates (query_spill)
    DevBuf<int2> spill_place;
    DevBuf<int> n_grid, ncand, overflow, pair_a, pair_b, counts, matches, nmatches, mp_octave;
    DevBuf<float> prev, mp_pos, proj_xy;
    DevBuf<se2gpu_keypoint> kps;
    DevBuf<uint8_t> desc, mp_desc, mp_skip, kf_obs, has1, has2;
    DevBuf<int> fvp1, fvi1, fvp2, fvi2, bin_of, hist;
    DevBuf<int2> node_pairs;
    // single-call paths (host buffers in, host buffers out): one packed block each way
    PinBuf<uint8_t> stage_h;
    DevBuf<uint8_t> stage_d;
    long long spill_calls = 0;   // host-buffer / batch calls in which some query took the exact spill scan (se2gpu_matcher_spill_calls)
    ~se2gpu_matcher() {
        if (own_stream) (void)hipStreamDestroy(own_stream);
    }
};

namespace {

int check_overflow(se2gpu_matcher* h) {
    int ov = 0;
    SE2_HIP(hipMemcpyAsync(&ov, h->overflow.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    SE2_HIP(hipStreamSynchronize(h->stream));
    // (a search window with more than kMaxCand candidates is no error any more: those queries were resolved by the exact
    // spill scan; the flag only says that it happened)
    if (ov) {
        ++h->spill_calls;
        SE2_HIP(hipMemsetAsync(h->overflow.p, 0, sizeof(int), h->stream));
    }
    return SE2GPU_OK;
}

// Dynamic LDS above 64 KiB has to be allowed per kernel AND per device: the candidate kernels stage 16 B per target
// feature (up to kMaxFeat features = 128 KiB), the greedy passes keep their state and the staged candidate lists in
// kResolveLds.
constexpr size_t kResolveLds = 120 * 1024;   // of the CU's 160 KiB: one workgroup per pair and CU
int lds_attributes() {
    static std::mutex mu;
    static bool done[64] = {};
    int dev = 0;
    SE2_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    if (dev >= 0 && dev < 64 && done[dev]) return SE2GPU_OK;
    SE2_HIP(hipFuncSetAttribute((const void*)k_cand_window, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxFeat * 16));
    SE2_HIP(hipFuncSetAttribute((const void*)k_cand_projection, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxFeat * 16));
    SE2_HIP(hipFuncSetAttribute((const void*)k_resolve_window, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kResolveLds));
    SE2_HIP(hipFuncSetAttribute((const void*)k_resolve_projection, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kResolveLds));
    if (dev >= 0 && dev < 64) done[dev] = true;
    return SE2GPU_OK;
}

// MatchByWindow over `npairs` pairs of frames of a device-resident frame set
int window_batch(se2gpu_matcher* h, const Bounds& bd, const se2gpu_keypoint* d_kps, const uint8_t* d_desc,
                 const int32_t* d_counts, int cap, int nframes_hint, const int32_t* d_pair_a, const int32_t* d_pair_b,
                 int npairs, int win, int level_offset, int min_level, int max_level, float nnratio, float* d_prev,
                 bool init_prev, int32_t* d_matches12, int32_t* d_nmatches, int* d_overflow = nullptr) {
    if (!d_overflow) d_overflow = h->overflow.p;
    hipStream_t st = h->stream;
    SE2_REQUIRE(cap <= kMaxFeat && cap <= 65536, SE2GPU_ERR_CAPACITY, "cap %d exceeds the matcher limit %d", cap, kMaxFeat);
    // grid order of every frame that appears as a target (all frames < nframes_hint: cheap)
    SE2_CHECK(h->sorted.reserve((size_t)nframes_hint * cap));
    SE2_CHECK(h->n_grid.reserve((size_t)nframes_hint * kGridRec));
    SE2_CHECK(h->cand.reserve((size_t)npairs * cap * kMaxCand));
    SE2_CHECK(h->ncand.reserve((size_t)npairs * cap));
    hipLaunchKernelGGL(k_grid_order, dim3(nframes_hint), dim3(256), 0, st, bd, d_kps, d_counts, (const int*)nullptr, cap,
                       h->sorted.p, h->n_grid.p);
    if (init_prev)
        hipLaunchKernelGGL(k_init_prev, dim3((cap + 255) / 256, npairs), dim3(256), 0, st, d_kps, d_counts, cap, d_pair_a,
                           d_prev);
    SE2_CHECK(lds_attributes());
    const int qpb = npairs >= 32 ? kCandQueriesBatch : kCandQueriesSingle;
    hipLaunchKernelGGL(k_cand_window, dim3((npairs + 7) & ~7, (cap + qpb - 1) / qpb), dim3(256), (size_t)cap * 16, st, bd,
                       d_kps, d_desc, d_counts, cap, d_pair_a, d_pair_b, d_prev, h->sorted.p, h->n_grid.p, win,
                       level_offset, min_level, max_level, h->cand.p, h->ncand.p, npairs, qpb);
    const size_t fixed_lds = ((size_t)11 * ((cap + 3) & ~3) + 48) * sizeof(int);
    constexpr size_t kLdsBudget = kResolveLds;
    SE2_REQUIRE(fixed_lds + 4096 <= kLdsBudget, SE2GPU_ERR_CAPACITY,
                "cap %d needs %zu B of LDS in the resolve pass (limit %zu)", cap, fixed_lds, kLdsBudget);
    const int cand_lds = (int)((kLdsBudget - 1024 - fixed_lds) / sizeof(int));  // staged candidate entries
    const size_t lds = fixed_lds + (size_t)cand_lds * sizeof(int);
    const int resolve_threads = std::min(1024, std::max(64, (cap + 63) & ~63));
    // spill arena: 8 x cap (feature, distance) entries per pair - e.g. sixteen queries that see half of the other frame each
    const int spill_cap = 8 * cap;
    SE2_CHECK(h->spill.reserve((size_t)npairs * spill_cap));
    SE2_CHECK(h->spill_place.reserve((size_t)npairs * cap));
    hipLaunchKernelGGL(k_resolve_window, dim3(npairs), dim3(resolve_threads), lds, st, d_kps, d_counts, cap, d_pair_a,
                       d_pair_b, h->cand.p, h->ncand.p, nnratio, cand_lds, d_matches12, d_prev, d_nmatches, d_overflow, bd, d_desc,
                       (const uint32_t*)h->sorted.p, (const int*)h->n_grid.p, win, level_offset, min_level, max_level, h->spill.p,
                       spill_cap, h->spill_place.p);
    SE2_HIP(hipGetLastError());
    return SE2GPU_OK;
}

}  // namespace

extern "C" {

// ORBmatcher::ComputeThreeMaxima (ORBmatcher.h:57, ORBmatcher.cpp:64-105) as the public member it is in the reference: the
// three fullest of L histogram bins, the second / third dropped below a tenth of the first.  Host utility - the same
// function the resolve kernels run on the device.  ind1..3 are in/out (the reference leaves them untouched where no bin
// qualifies; its callers start from -1).
int se2gpu_three_maxima(const int32_t* counts, int L, int* ind1, int* ind2, int* ind3) {
    SE2_REQUIRE(L >= 0 && (L == 0 || counts) && ind1 && ind2 && ind3, SE2GPU_ERR_INVALID, "three_maxima: bad argument");
    three_maxima_n(counts, L, *ind1, *ind2, *ind3);
    return SE2GPU_OK;
}

// The reference constructs its ORBmatcher on the stack of every call (Track.cpp:131, LocalMapper.cpp:117).  Destroyed
// handles are parked with their stream and buffers (per device, at most kPoolMax) and handed out again by
// se2gpu_matcher_create; SE2GPU_MATCHER_POOL=0 disables this.
namespace {
constexpr size_t kPoolMax = 4;
static std::mutex g_mt_pool_mu;
static std::vector<se2gpu_matcher*> g_mt_pool;
static bool mt_pool_enabled() {
    static const bool on = [] { const char* e = std::getenv("SE2GPU_MATCHER_POOL"); return !(e && e[0] == '0'); }();
    return on;
}
}  // namespace

int se2gpu_matcher_create(int max_features, int max_batch, se2gpu_matcher** out) {
    SE2_REQUIRE(out, SE2GPU_ERR_INVALID, "matcher_create: out is NULL");
    SE2_REQUIRE(have_device(), SE2GPU_ERR_NO_DEVICE, "no HIP device visible (libse2gpu has no CPU fallback)");
    SE2_REQUIRE(max_features > 0 && max_features <= kMaxFeat, SE2GPU_ERR_INVALID, "max_features must be in 1..%d", kMaxFeat);
    int dev = 0;
    SE2_HIP(hipGetDevice(&dev));
    {
        std::lock_guard<std::mutex> lk(g_mt_pool_mu);
        for (size_t i = 0; i < g_mt_pool.size(); ++i)
            if (g_mt_pool[i]->device == dev) {
                se2gpu_matcher* h = g_mt_pool[i];
                g_mt_pool.erase(g_mt_pool.begin() + (ptrdiff_t)i);
                h->max_features = max_features;
                h->max_batch = std::max(1, max_batch);
                h->spill_calls = 0;   // a matcher handed out again starts like a new one
                *out = h;
                return SE2GPU_OK;
            }
    }
    se2gpu_matcher* h = new se2gpu_matcher;
    h->device = dev;
    h->max_features = max_features;
    h->max_batch = std::max(1, max_batch);
    if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        set_error("hipStreamCreate failed");
        return SE2GPU_ERR_HIP;
    }
    h->stream = h->own_stream;
    if (h->overflow.reserve(1) != SE2GPU_OK || hipMemsetAsync(h->overflow.p, 0, sizeof(int), h->stream) != hipSuccess) {
        delete h;
        return SE2GPU_ERR_HIP;
    }
    *out = h;
    return SE2GPU_OK;
}

void se2gpu_matcher_destroy(se2gpu_matcher* h) {
    if (!h) return;
    if (mt_pool_enabled()) {
        (void)hipStreamSynchronize(h->stream);
        h->stream = h->own_stream;
        (void)hipMemsetAsync(h->overflow.p, 0, sizeof(int), h->stream);
        std::lock_guard<std::mutex> lk(g_mt_pool_mu);
        if (g_mt_pool.size() < kPoolMax) {
            g_mt_pool.push_back(h);
            return;
        }
    }
    delete h;
}
void* se2gpu_matcher_stream(se2gpu_matcher* h) { return h ? (void*)h->stream : nullptr; }

int se2gpu_matcher_set_stream(se2gpu_matcher* h, void* s) {
    SE2_REQUIRE(h, SE2GPU_ERR_INVALID, "matcher handle is NULL");
    h->stream = s ? (hipStream_t)s : h->own_stream;
    return SE2GPU_OK;
}

// Calls of this handle so far in which at least one query had more than 128 candidates in its search window and was
// resolved by the exact spill scan (one thread walks the grid in the first sweep of the fixed point and records the list; the
// later sweeps replay it).  A caller on
// the tracking thread can watch the counter: a frame pair that moves it pays a latency the usual pair does not.
int se2gpu_matcher_spill_calls(const se2gpu_matcher* h, long long* calls) {
    SE2_REQUIRE(h && calls, SE2GPU_ERR_INVALID, "matcher_spill_calls: NULL argument");
    *calls = h->spill_calls;
    return SE2GPU_OK;
}

int se2gpu_matcher_sync(se2gpu_matcher* h) {
    SE2_REQUIRE(h, SE2GPU_ERR_INVALID, "matcher handle is NULL");
    return check_overflow(h);
}

int se2gpu_match_window_batch_device(se2gpu_matcher* h, const se2gpu_frame_bounds* bounds,
                                     const se2gpu_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_counts,
                                     int cap, const int32_t* d_pair_a, const int32_t* d_pair_b, int npairs, int win_size,
                                     int level_offset, int min_level, int max_level, float nnratio,
                                     int32_t* d_matches12, int32_t* d_nmatches) {
    SE2_REQUIRE(h && bounds && d_kps && d_desc && d_counts && d_pair_a && d_pair_b && d_matches12 && d_nmatches,
                SE2GPU_ERR_INVALID, "match_window_batch: NULL argument");
    SE2_REQUIRE(npairs >= 1 && cap >= 1, SE2GPU_ERR_INVALID, "match_window_batch: bad sizes");
    // frames referenced by the pairs are 0..max_batch-1 of the extractor's output arrays
    const int nframes = h->max_batch;
    SE2_CHECK(h->prev.reserve((size_t)npairs * cap * 2));
    return window_batch(h, make_bounds(*bounds), d_kps, d_desc, d_counts, cap, nframes, d_pair_a, d_pair_b, npairs,
                        win_size, level_offset, min_level, max_level, nnratio, h->prev.p, true, d_matches12, d_nmatches);
}

int se2gpu_match_window(se2gpu_matcher* h, const se2gpu_frame_bounds* bounds, const se2gpu_keypoint* kps1,
                        const uint8_t* desc1, int n1, const se2gpu_keypoint* kps2, const uint8_t* desc2, int n2,
                        float* prev_xy, int win_size, int level_offset, int min_level, int max_level, float nnratio,
                        int32_t* matches12, int* n_matches) {
    SE2_REQUIRE(h && bounds && n_matches, SE2GPU_ERR_INVALID, "match_window: NULL argument");
    SE2_REQUIRE(n1 >= 0 && n2 >= 0 && n1 <= h->max_features && n2 <= h->max_features, SE2GPU_ERR_CAPACITY,
                "match_window: %d / %d features exceed max_features %d", n1, n2, h->max_features);
    *n_matches = 0;
    if (n1 == 0) return SE2GPU_OK;
    SE2_REQUIRE(kps1 && desc1 && prev_xy && matches12 && (n2 == 0 || (kps2 && desc2)), SE2GPU_ERR_INVALID,
                "match_window: NULL buffer");
    hipStream_t st = h->stream;
    const int cap = std::max(std::max(n1, n2), 1);
    // one block both ways: [matches12 | prev_xy | nmatches, overflow, counts(2), pair_a, pair_b | kps (2 cap) | desc (2 cap)]
    auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_m = 0, o_prev = up16(o_m + (size_t)cap * sizeof(int)), o_sc = up16(o_prev + (size_t)cap * 2 * sizeof(float));
    const size_t o_kps = o_sc + 32, o_desc = up16(o_kps + 2 * (size_t)cap * sizeof(se2gpu_keypoint));
    const size_t total = o_desc + 2 * (size_t)cap * 32;
    SE2_CHECK(h->stage_h.reserve(total));
    SE2_CHECK(h->stage_d.reserve(total));
    uint8_t* hs = h->stage_h.p;
    uint8_t* ds = h->stage_d.p;
    std::memcpy(hs + o_prev, prev_xy, (size_t)n1 * 2 * sizeof(float));
    const int sc[8] = {0, 0, n1, n2, 0, 1, 0, 0};
    std::memcpy(hs + o_sc, sc, sizeof(sc));
    std::memcpy(hs + o_kps, kps1, (size_t)n1 * sizeof(se2gpu_keypoint));
    std::memcpy(hs + o_desc, desc1, (size_t)n1 * 32);
    if (n2) {
        std::memcpy(hs + o_kps + (size_t)cap * sizeof(se2gpu_keypoint), kps2, (size_t)n2 * sizeof(se2gpu_keypoint));
        std::memcpy(hs + o_desc + (size_t)cap * 32, desc2, (size_t)n2 * 32);
    }
    SE2_HIP(hipMemcpyAsync(ds + o_prev, hs + o_prev, total - o_prev, hipMemcpyHostToDevice, st));
    int* d_sc = (int*)(ds + o_sc);
    SE2_CHECK(window_batch(h, make_bounds(*bounds), (const se2gpu_keypoint*)(ds + o_kps), ds + o_desc, d_sc + 2, cap, 2,
                           d_sc + 4, d_sc + 5, 1, win_size, level_offset, min_level, max_level, nnratio,
                           (float*)(ds + o_prev), false, (int32_t*)(ds + o_m), d_sc, d_sc + 1));
    SE2_HIP(hipMemcpyAsync(hs, ds, o_kps, hipMemcpyDeviceToHost, st));
    SE2_HIP(hipStreamSynchronize(st));
    const int* r = (const int*)(hs + o_sc);
    std::memcpy(matches12, hs + o_m, (size_t)n1 * sizeof(int));
    std::memcpy(prev_xy, hs + o_prev, (size_t)n1 * 2 * sizeof(float));
    *n_matches = r[0];
    if (r[1]) ++h->spill_calls;
    return SE2GPU_OK;
}

int se2gpu_match_projection(se2gpu_matcher* h, const se2gpu_frame_bounds* bounds, const float* mp_pos,
                            const uint8_t* mp_desc, const int32_t* mp_octave, const uint8_t* mp_skip, int m,
                            const float* Tcw, float fx, float fy, float cx, float cy, const se2gpu_keypoint* kps,
                            const uint8_t* desc, const uint8_t* kf_observed, int n, int win_size, int level_offset,
                            float nnratio, int32_t* match_idx_mp, int* n_matches) {
    SE2_REQUIRE(h && bounds && n_matches && Tcw, SE2GPU_ERR_INVALID, "match_projection: NULL argument");
    SE2_REQUIRE(n >= 0 && n <= h->max_features && m >= 0, SE2GPU_ERR_CAPACITY, "match_projection: %d features exceed %d", n,
                h->max_features);
    *n_matches = 0;
    if (n == 0) return SE2GPU_OK;
    SE2_REQUIRE(kps && desc && kf_observed && match_idx_mp && (m == 0 || (mp_pos && mp_desc && mp_octave && mp_skip)),
                SE2GPU_ERR_INVALID, "match_projection: NULL buffer");
    hipStream_t st = h->stream;
    const int mm = std::max(m, 1);
    SE2_CHECK(h->sorted.reserve((size_t)n));
    SE2_CHECK(h->n_grid.reserve(kGridRec));
    SE2_CHECK(h->cand.reserve((size_t)mm * kMaxCand));
    SE2_CHECK(h->ncand.reserve((size_t)mm));
    SE2_CHECK(h->proj_xy.reserve(2 * (size_t)mm));
    // one block both ways: [match_idx (n) | nmatches, overflow, count | kps | desc | kf_obs | mp_pos | mp_octave | mp_desc | mp_skip]
    auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_m = 0, o_sc = up16((size_t)n * sizeof(int)), o_kps = o_sc + 16;
    const size_t o_desc = up16(o_kps + (size_t)n * sizeof(se2gpu_keypoint)), o_obs = o_desc + (size_t)n * 32;
    const size_t o_pos = up16(o_obs + (size_t)n), o_oct = up16(o_pos + 3 * (size_t)mm * sizeof(float));
    const size_t o_mdesc = up16(o_oct + (size_t)mm * sizeof(int)), o_skip = o_mdesc + 32 * (size_t)mm;
    const size_t total = up16(o_skip + (size_t)mm);
    SE2_CHECK(h->stage_h.reserve(total));
    SE2_CHECK(h->stage_d.reserve(total));
    uint8_t* hs = h->stage_h.p;
    uint8_t* ds = h->stage_d.p;
    const int sc[4] = {0, 0, n, 0};
    std::memcpy(hs + o_sc, sc, sizeof(sc));
    std::memcpy(hs + o_kps, kps, (size_t)n * sizeof(se2gpu_keypoint));
    std::memcpy(hs + o_desc, desc, (size_t)n * 32);
    std::memcpy(hs + o_obs, kf_observed, (size_t)n);
    if (m) {
        std::memcpy(hs + o_pos, mp_pos, 3 * (size_t)m * sizeof(float));
        std::memcpy(hs + o_oct, mp_octave, (size_t)m * sizeof(int));
        std::memcpy(hs + o_mdesc, mp_desc, 32 * (size_t)m);
        std::memcpy(hs + o_skip, mp_skip, (size_t)m);
    }
    SE2_HIP(hipMemcpyAsync(ds + o_sc, hs + o_sc, total - o_sc, hipMemcpyHostToDevice, st));
    const se2gpu_keypoint* d_kps = (const se2gpu_keypoint*)(ds + o_kps);
    int* d_sc = (int*)(ds + o_sc);
    const Bounds bd = make_bounds(*bounds);
    ProjCam cam;
    std::memcpy(cam.T, Tcw, sizeof(cam.T));
    cam.fx = fx; cam.fy = fy; cam.cx = cx; cam.cy = cy;
    hipLaunchKernelGGL(k_grid_order, dim3(1), dim3(256), 0, st, bd, d_kps, d_sc + 2, (const int*)nullptr, n, h->sorted.p,
                       h->n_grid.p);
    if (m) {
        SE2_CHECK(lds_attributes());
        const int qpb = kCandQueriesSingle;
        hipLaunchKernelGGL(k_cand_projection, dim3((m + qpb - 1) / qpb), dim3(256), (size_t)n * 16, st, bd, cam,
                           (const float*)(ds + o_pos), ds + o_mdesc, (const int*)(ds + o_oct), ds + o_skip, m, d_kps,
                           ds + o_desc, ds + o_obs, h->sorted.p, h->n_grid.p, win_size, level_offset, h->cand.p, h->ncand.p,
                           n, qpb, h->proj_xy.p);
    }
    {
        const int chunk = 1024;
        constexpr size_t kLdsBudget = kResolveLds;
        const size_t nE = ((size_t)n + 3) & ~(size_t)3;
        const size_t fixed_lds = (3 * nE + 6 * (size_t)chunk + 8) * sizeof(int);
        SE2_REQUIRE(fixed_lds + 4096 <= kLdsBudget, SE2GPU_ERR_CAPACITY, "%d key-frame features need %zu B of LDS", n, fixed_lds);
        SE2_CHECK(lds_attributes());
        const int cand_lds = (int)((kLdsBudget - 1024 - fixed_lds) / sizeof(int));
        const int spill_cap = 16 * std::max(n, 1024);   // per chunk of 1024 map points
        SE2_CHECK(h->spill.reserve((size_t)spill_cap));
        SE2_CHECK(h->spill_place.reserve((size_t)std::max(m, 1)));
        hipLaunchKernelGGL(k_resolve_projection, dim3(1), dim3(1024), fixed_lds + (size_t)cand_lds * sizeof(int), st, d_kps,
                           n, m, h->cand.p, h->ncand.p, nnratio, chunk, cand_lds, (int*)(ds + o_m), d_sc, d_sc + 1, bd,
                           (const uint8_t*)(ds + o_desc), (const uint8_t*)(ds + o_obs), (const uint32_t*)h->sorted.p,
                           (const int*)h->n_grid.p, (const float*)h->proj_xy.p, (const uint8_t*)(ds + o_mdesc),
                           (const int*)(ds + o_oct), win_size, level_offset, h->spill.p, spill_cap, h->spill_place.p);
    }
    SE2_HIP(hipGetLastError());
    SE2_HIP(hipMemcpyAsync(hs, ds, o_kps, hipMemcpyDeviceToHost, st));
    SE2_HIP(hipStreamSynchronize(st));
    const int* r = (const int*)(hs + o_sc);
    std::memcpy(match_idx_mp, hs + o_m, (size_t)n * sizeof(int));
    *n_matches = r[0];
    if (r[1]) ++h->spill_calls;
    return SE2GPU_OK;
}

int se2gpu_search_by_bow(se2gpu_matcher* h, const se2gpu_keypoint* kps1, const uint8_t* desc1, int n1,
                         const int32_t* fv1_nodes, const int32_t* fv1_ptr, const int32_t* fv1_idx, int nn1,
                         const uint8_t* has_mp1, const se2gpu_keypoint* kps2, const uint8_t* desc2, int n2,
                         const int32_t* fv2_nodes, const int32_t* fv2_ptr, const int32_t* fv2_idx, int nn2,
                         const uint8_t* has_mp2, int mp_only, float nnratio, int check_orientation, int32_t* matches12,
                         int* n_matches) {
    SE2_REQUIRE(h && n_matches, SE2GPU_ERR_INVALID, "search_by_bow: NULL argument");
    SE2_REQUIRE(n1 >= 0 && n2 >= 0 && n1 <= h->max_features && n2 <= h->max_features && nn1 >= 0 && nn2 >= 0,
                SE2GPU_ERR_CAPACITY, "search_by_bow: sizes out of range");
    *n_matches = 0;
    if (n1 == 0) return SE2GPU_OK;
    SE2_REQUIRE(kps1 && desc1 && matches12 && (nn1 == 0 || (fv1_nodes && fv1_ptr && fv1_idx)) &&
                    (n2 == 0 || (kps2 && desc2)) && (nn2 == 0 || (fv2_nodes && fv2_ptr && fv2_idx)) &&
                    (!mp_only || (has_mp1 && (n2 == 0 || has_mp2))),
                SE2GPU_ERR_INVALID, "search_by_bow: NULL buffer");
    // merge walk over the two ascending node lists (host: hundreds of nodes) -> pairs of equal nodes
    std::vector<int2> pairs;
    for (int a = 0, b = 0; a < nn1 && b < nn2;) {
        if (fv1_nodes[a] == fv2_nodes[b]) { pairs.push_back(make_int2(a, b)); ++a; ++b; }
        else if (fv1_nodes[a] < fv2_nodes[b]) ++a;
        else ++b;
    }
    for (int k = 0; k < nn1; ++k)
        SE2_REQUIRE(fv1_ptr[k] <= fv1_ptr[k + 1], SE2GPU_ERR_INVALID, "search_by_bow: fv1_ptr is not monotone");
    for (int k = 0; k < nn2; ++k) {
        SE2_REQUIRE(fv2_ptr[k] <= fv2_ptr[k + 1], SE2GPU_ERR_INVALID, "search_by_bow: fv2_ptr is not monotone");
    }
    int max_node2 = 64;   // LDS bytes of k_bow_match: one flag per feature of the largest node (a key frame has <= max_features)
    for (int k = 0; k < nn2; ++k) max_node2 = std::max(max_node2, fv2_ptr[k + 1] - fv2_ptr[k]);
    SE2_REQUIRE(max_node2 <= 64 * 1024, SE2GPU_ERR_CAPACITY, "search_by_bow: %d features in one vocabulary node", max_node2);
    hipStream_t st = h->stream;
    const int t1 = nn1 ? fv1_ptr[nn1] : 0, t2 = nn2 ? fv2_ptr[nn2] : 0;
    SE2_CHECK(h->kps.reserve((size_t)n1 + n2 + 1));
    SE2_CHECK(h->desc.reserve(((size_t)n1 + n2 + 1) * 32));
    SE2_CHECK(h->has1.reserve((size_t)n1 + 1));
    SE2_CHECK(h->has2.reserve((size_t)n2 + 1));
    SE2_CHECK(h->fvp1.reserve((size_t)nn1 + 2));
    SE2_CHECK(h->fvp2.reserve((size_t)nn2 + 2));
    SE2_CHECK(h->fvi1.reserve((size_t)t1 + 1));
    SE2_CHECK(h->fvi2.reserve((size_t)t2 + 1));
    SE2_CHECK(h->matches.reserve((size_t)n1));
    SE2_CHECK(h->bin_of.reserve((size_t)n1));
    SE2_CHECK(h->hist.reserve(32));
    SE2_CHECK(h->nmatches.reserve(1));
    SE2_CHECK(h->node_pairs.reserve(pairs.size() + 1));
    se2gpu_keypoint* d_k2 = h->kps.p + n1;
    uint8_t* d_d2 = h->desc.p + (size_t)n1 * 32;
    SE2_HIP(hipMemcpyAsync(h->kps.p, kps1, (size_t)n1 * sizeof(se2gpu_keypoint), hipMemcpyHostToDevice, st));
    SE2_HIP(hipMemcpyAsync(h->desc.p, desc1, (size_t)n1 * 32, hipMemcpyHostToDevice, st));
    if (n2) {
        SE2_HIP(hipMemcpyAsync(d_k2, kps2, (size_t)n2 * sizeof(se2gpu_keypoint), hipMemcpyHostToDevice, st));
        SE2_HIP(hipMemcpyAsync(d_d2, desc2, (size_t)n2 * 32, hipMemcpyHostToDevice, st));
    }
    if (mp_only) {
        SE2_HIP(hipMemcpyAsync(h->has1.p, has_mp1, (size_t)n1, hipMemcpyHostToDevice, st));
        if (n2) SE2_HIP(hipMemcpyAsync(h->has2.p, has_mp2, (size_t)n2, hipMemcpyHostToDevice, st));
    }
    if (nn1) {
        SE2_HIP(hipMemcpyAsync(h->fvp1.p, fv1_ptr, ((size_t)nn1 + 1) * sizeof(int), hipMemcpyHostToDevice, st));
        if (t1) SE2_HIP(hipMemcpyAsync(h->fvi1.p, fv1_idx, (size_t)t1 * sizeof(int), hipMemcpyHostToDevice, st));
    }
    if (nn2) {
        SE2_HIP(hipMemcpyAsync(h->fvp2.p, fv2_ptr, ((size_t)nn2 + 1) * sizeof(int), hipMemcpyHostToDevice, st));
        if (t2) SE2_HIP(hipMemcpyAsync(h->fvi2.p, fv2_idx, (size_t)t2 * sizeof(int), hipMemcpyHostToDevice, st));
    }
    if (!pairs.empty())
        SE2_HIP(hipMemcpyAsync(h->node_pairs.p, pairs.data(), pairs.size() * sizeof(int2), hipMemcpyHostToDevice, st));
    SE2_HIP(hipMemsetAsync(h->matches.p, 0xff, (size_t)n1 * sizeof(int), st));
    SE2_HIP(hipMemsetAsync(h->bin_of.p, 0xff, (size_t)n1 * sizeof(int), st));
    SE2_HIP(hipMemsetAsync(h->hist.p, 0, 32 * sizeof(int), st));
    if (!pairs.empty())
        hipLaunchKernelGGL(k_bow_match, dim3((unsigned)pairs.size()), dim3(64), (size_t)((max_node2 + 63) & ~63), st, h->node_pairs.p, h->fvp1.p, h->fvi1.p,
                           h->fvp2.p, h->fvi2.p, h->kps.p, h->desc.p, h->has1.p, d_k2, d_d2, h->has2.p, mp_only, nnratio,
                           check_orientation, h->matches.p, h->bin_of.p, h->hist.p);
    hipLaunchKernelGGL(k_bow_finish, dim3(1), dim3(256), 0, st, n1, check_orientation, h->bin_of.p, h->hist.p, h->matches.p,
                       h->nmatches.p);
    SE2_HIP(hipGetLastError());
    SE2_HIP(hipMemcpyAsync(matches12, h->matches.p, (size_t)n1 * sizeof(int), hipMemcpyDeviceToHost, st));
    SE2_HIP(hipMemcpyAsync(n_matches, h->nmatches.p, sizeof(int), hipMemcpyDeviceToHost, st));
    SE2_HIP(hipStreamSynchronize(st));
    return SE2GPU_OK;
}

}  // extern "C"
