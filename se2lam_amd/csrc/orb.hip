// libse2gpu - ORB extractor on gfx950 (MI355X), batched over frames.
//
// Replaces se2lam::ORBextractor (/root/reference/src/ORBextractor.cpp, include/se2lam/ORBextractor.h:38-83):
//   ctor :463-520, ComputePyramid :790-831, ComputeKeyPoints :531-716, IC_Angle :130-157,
//   GaussianBlur call :769, computeOrbDescriptor :161-200, operator() :727-788.
// The OpenCV primitives it calls are restated with the semantics DESIGN.md defines
// (FAST-9/16 + cornerScore + in-cell NMS, 11-bit fixed-point bilinear resize, reflect-101 border, 8-bit
// fixed-point 7x7 Gaussian, retainBest as libstdc++'s nth_element leaves it, fastAtan2, round-half-even).
//
// Pipeline (every kernel covers the whole batch; all stages stay in HBM, no host round trip):
//   k_level0 / k_resize   pyramid level k from level k-1, border filled in the same pass; resize coefficients from host
//                         tables over the bordered row, the taps of a thread from one 64-bit load per source row
//   k_fast_score[_sparse] S(x,y) = FAST-9/16 score for every level in one launch, in-cell non-max suppression fused:
//                         writes a sparse plane (S where S > 7 and a strict in-cell maximum, else 0)
//   k_cell_collect        one workgroup per (frame, level, cell): what cv::FAST returns for the cell (threshold 20 /
//                         fallback 7) in row-major order -> the cell's list (<true>: HARRIS_SCORE, responses re-scored)
//   k_cell_retain         one wave per cell: quota redistribution replayed, then retainBest + resize = libstdc++'s
//                         nth_element (introselect) step for step, the partition spread over the lanes
//   k_level_select        one workgroup per (frame, level): the cells' retained corners in cell order, the level-wide
//                         retainBest + resize by the same introselect, key points out in the reference's order
//   k_orientation         one wave per keypoint: integer intensity-centroid moments, fastAtan2
//   k_blur                blurred pyramid: the 16 px frame keeps the un-blurred reflect copies (as in the reference's
//                         in-place ROI blur), the interior gets the 7x7 fixed-point Gaussian (register window, packed
//                         u16 horizontal taps), on a side stream beside the key-point chain; the frame is written by
//                         k_level0 / k_resize
//   k_angle_trig          cos / sin of the key-point angle (double, rounded once), one thread per key point
//   k_describe            one wave per two keypoints: lane l evaluates pattern pairs l, l+64, l+128, l+192 and the four
//                         64-bit wave ballots ARE the 256-bit descriptor; also writes the final cv::KeyPoint
// Compiled with -ffp-contract=off (se2lam_amd/build.py): the float index arithmetic must round as the
// reference's non-FMA build does.
#include <algorithm>
#include <cfloat>
#include <cmath>

#include <cstdlib>

#include "common.h"

using namespace se2gpu;

namespace {

constexpr int kMaxLevels = 12;
constexpr int kEdge = 16;          // EDGE_THRESHOLD
constexpr int kPatch = 31;         // PATCH_SIZE
constexpr int kHalfPatch = 15;     // HALF_PATCH_SIZE
constexpr int kSortCap = 4096;     // candidates (S > 7, local maxima) one cell can hold in LDS

__constant__ signed char c_pattern[1024] = {
#include "orb_pattern_31.inc"
};

struct Geom {
    int nlevels;
    int rows, cols;
    int w[kMaxLevels], h[kMaxLevels], stride[kMaxLevels];
    unsigned off[kMaxLevels];            // byte offset of the level's bordered buffer inside a frame block
    unsigned frame_bytes;
    int quota[kMaxLevels], gcols[kMaxLevels], grows[kMaxLevels], cellW[kMaxLevels], cellH[kMaxLevels];
    int nfc[kMaxLevels];                 // nfeaturesCell
    // The scan area of a level is the union of its cells' FAST windows: x in [16, sx1), y in [16, sy1).  Normally sx1 = w - 16;
    // when the cells are so wide that the last column starts at or beyond w - 16 ((gcols - 1) * cellW >= w - 32: its window is
    // empty, ORBextractor.cpp:598-603) the column before it still spans its full cellW and scans up to 13 px of the reflect
    // frame - the reference does exactly that (beyond 13 px its cv::Mat::colRange raises).  Same for rows.  skip: bit 0 /
    // bit 1 = the last cell column / row is not even visited by the reference's loops (hX <= 0 / hY <= 0: `continue`), which
    // leaves those cells open in the quota redistribution instead of closing them with zero key points.
    int sx1[kMaxLevels], sy1[kMaxLevels], skip[kMaxLevels];
    int level_cap;                       // entries of k_level_select's list: max over the levels of quota + 2 * cells (rounded up)
    int cell_base[kMaxLevels + 1];       // prefix sum of cells per level
    float scale[kMaxLevels];             // mvScaleFactor
    float patch[kMaxLevels];             // (float)(int)(PATCH_SIZE * mvScaleFactor[level])
    int tile_base[kMaxLevels + 1];       // prefix sums of work tiles per level (set per kernel family)
    // candidate lists written by the score kernel, one per list tile (lst_tw x lst_th pixels of the scan area; a strip of
    // k_fast_score or a tile of k_fast_score_sparse): list (column tc, row tr) of level l = lst_base[l] + tr * tiles_x + tc
    int lst_tw, lst_th, lst_cap;
    int lst_base[kMaxLevels + 1];
    // per-cell lists of k_cell_collect: a cell of level l owns lcap[l] entries (no cell can hold more corners: in-cell maxima
    // are never adjacent) at lcell_off[l] + cell * lcap[l] of its frame's block (lcell_off[nlevels] entries); levels whose
    // cells can exceed the LDS paths also own 2 * lcap[l] uint32 of scratch per cell, at lscr_off[l] (sort / position lists)
    int lcap[kMaxLevels];
    unsigned lcell_off[kMaxLevels + 1], lscr_off[kMaxLevels + 1];
    int umax[16];
    int nfeatures;
    int fast_th;
    int harris;                          // scoreType == ORB::HARRIS_SCORE: retain by Harris response
    int nframes;                         // frames of this launch
};

// Every batch kernel is launched on a (frames rounded up to 8, blocks per frame) grid.  Workgroups reach the 8 XCDs
// round-robin in launch order (x fastest), so frame f lives on XCD f % 8 in EVERY kernel: the halos, discs and patches
// that neighbouring workgroups of a frame share meet in one L2, and all XCDs work on the same block index at the same
// time (equal load).
#define SE2_FRAME_GRID(f, bx)          \
    const int f = (int)blockIdx.x;     \
    const int bx = (int)blockIdx.y;    \
    if (f >= g.nframes) return

// the same with a block offset: a launch that covers only the tiles of some levels (pipelined batches, see orb_run)
#define SE2_FRAME_GRID_OFF(f, bx, off)       \
    const int f = (int)blockIdx.x;           \
    const int bx = (int)blockIdx.y + (off);  \
    if (f >= g.nframes) return

// interior pixel (x, y) of level l of frame f
__device__ __forceinline__ size_t pix(const Geom& g, int f, int l, int y, int x) {
    return (size_t)f * g.frame_bytes + g.off[l] + (size_t)(y + kEdge) * g.stride[l] + (x + kEdge);
}

__host__ __device__ __forceinline__ int reflect101(int p, int n) { return p < 0 ? -p : (p >= n ? 2 * n - 2 - p : p); }

// ---------------------------------------------------------------------------------------------
// pyramid
// ---------------------------------------------------------------------------------------------
// level 0: copyMakeBorder(image, 16 px, BORDER_REFLECT_101)
// One thread = one 16-byte chunk of a bordered row.  The item space is [interior chunks of all rows | the other chunks]:
// an interior chunk is one aligned 128-bit load, a chunk that touches the reflected frame or the row padding gathers
// bytes - keeping the two kinds in separate waves keeps the gather out of 95 % of them.
__global__ __launch_bounds__(256) void k_level0(Geom g, const uint8_t* __restrict__ imgs, int pitch,
                                                 uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur) {
    SE2_FRAME_GRID(f, bx);
    const int W = g.w[0], H = g.h[0], stride = g.stride[0];
    const int nch = stride / 16, rows = H + 2 * kEdge;
    const bool aligned = (W % 16 == 0) && (pitch % 16 == 0);    // interior chunks exist only then
    const int nin = aligned ? W / 16 : 0;                        // interior chunks per row: X0 = 16 .. W
    const int item = bx * 256 + threadIdx.x;
    int Y, c;
    bool interior;
    if (item < nin * rows) {
        Y = item / nin;
        c = item - Y * nin + 1;
        interior = true;
    } else {
        const int j = item - nin * rows, nb = nch - nin;
        Y = j / nb;
        if (Y >= rows) return;
        c = j - Y * nb;
        if (c >= 1) c += nin;                                    // chunk 0, then the chunks past the interior
        interior = false;
    }
    const int X0 = c * 16;                                       // first of 16 columns of the bordered buffer
    const int sy = reflect101(Y - kEdge, H);
    const uint8_t* src = imgs + (size_t)f * pitch * H + (size_t)sy * pitch;
    uint4 v;
    if (interior) {
        v = *reinterpret_cast<const uint4*>(src + (X0 - kEdge));
    } else {
        uint32_t wv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t acc = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int X = X0 + 4 * k + q;
                uint32_t b = 0;
                if (X < W + 2 * kEdge) b = src[reflect101(X - kEdge, W)];
                acc |= b << (8 * q);
            }
            wv[k] = acc;
        }
        v = make_uint4(wv[0], wv[1], wv[2], wv[3]);
    }
    const size_t off = (size_t)f * g.frame_bytes + g.off[0] + (size_t)Y * stride + X0;
    *reinterpret_cast<uint4*>(pyr + off) = v;
    // the blurred pyramid keeps the un-blurred frame (see k_blur): frame rows, the left chunk, the chunks from the right edge on
    if (!interior || Y < kEdge || Y >= H + kEdge) *reinterpret_cast<uint4*>(blur + off) = v;
}

// level l >= 1: cv::resize(level l-1, INTER_LINEAR) + reflect-101 border, 4 output bytes per thread.
// ytab[dy] = {sy0, sy1, b0, b1}.  xtab has one entry per column X of the BORDERED destination row (so the reflection
// costs nothing per pixel), built on the host with exactly cv::resize's arithmetic (fx = (float)((dx + 0.5) * scale_x
// - 0.5) in double, 11-bit rounding in float, the S[sx] * ONE tail): two int4 per group of 4 columns,
// {sx | valid << 31} x 4 and {a0 | a1 << 16} x 4.  The two source bytes S[sx], S[sx + 1] of a row come from one
// (unaligned) 64-bit load per source row and column group.
struct ResizeTab {
    const int4* ytab;
    const int4* xtab;
    int ngroups;   // stride / 4
};

// A thread owns one column group (4 output bytes) and walks down kResizeRows rows: the 32 B of x coefficients are
// fetched once per thread instead of once per 4 output bytes (they were three quarters of this kernel's load traffic),
// the y entry of a row is one broadcast load per wave.  Workgroup = 4 waves = 4 row bands of one 64-group column tile.
constexpr int kResizeRows = 8;
__global__ __launch_bounds__(256) void k_resize(Geom g, int l, ResizeTab t, uint8_t* __restrict__ pyr,
                                                 uint8_t* __restrict__ blur) {
    SE2_FRAME_GRID(f, bx);
    const int H = g.h[l], stride = g.stride[l], rows = H + 2 * kEdge;
    const int ctiles = (t.ngroups + 63) / 64;
    const int xg = (bx % ctiles) * 64 + (int)(threadIdx.x & 63);
    // (the row band is the wave's: told to the compiler, the y table entries become scalar loads - one level less in the
    // table -> source row -> pixel chain of dependent loads, and the row addresses live in scalar registers)
    const int Y0 = __builtin_amdgcn_readfirstlane(((bx / ctiles) * 4 + (int)(threadIdx.x >> 6)) * kResizeRows);
    if (xg >= t.ngroups || Y0 >= rows) return;
    const int4 xs = t.xtab[2 * xg], xa = t.xtab[2 * xg + 1];
    const int sxv[4] = {xs.x, xs.y, xs.z, xs.w};
    const int av[4] = {xa.x, xa.y, xa.z, xa.w};
    // The source columns of the four output pixels span at most 4 (scale 1.2, also across the reflection), so all
    // eight taps of a source row lie in ONE (unaligned) 64-bit word starting at the smallest column.
    const int sx0 = sxv[0] & 0xffff, sx1 = sxv[1] & 0xffff, sx2 = sxv[2] & 0xffff, sx3 = sxv[3] & 0xffff;
    const int sb = min(min(sx0, sx1), min(sx2, sx3));
    // tap pair {S[sx], S[sx + 1]} of output pixel q as two u16 lanes: one v_perm_b32 out of the 64-bit word (selector bytes
    // i, zero, i + 1, zero with i = sx - sb), and the horizontal interpolation S[sx] * a0 + S[sx + 1] * a1 is one
    // v_dot2_u32_u16 with the table's {a0 | a1 << 16} word - two instructions per pixel and source row instead of seven
    typedef unsigned short ushort2r __attribute__((ext_vector_type(2)));
    uint32_t sel[4];
    ushort2r aw[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t i = (uint32_t)((sxv[q] & 0xffff) - sb);
        sel[q] = i | 0x0c000c00u | ((i + 1u) << 16);
        aw[q] = __builtin_bit_cast(ushort2r, (uint32_t)av[q]);
    }
    const uint8_t* src = pyr + pix(g, f, l - 1, 0, 0) + sb;
    const int sstride = g.stride[l - 1];
    const size_t obase = (size_t)f * g.frame_bytes + g.off[l] + 4 * xg;
    const bool frame_col = 4 * xg < 16 || 4 * xg >= ((kEdge + g.w[l]) / 16) * 16;
    const int Yend = min(Y0 + kResizeRows, rows);
#pragma unroll 4
    for (int Y = Y0; Y < Yend; ++Y) {
        const int4 yt = t.ytab[reflect101(Y - kEdge, H)];
        const unsigned long long w0 = *reinterpret_cast<const unsigned long long*>(src + (ptrdiff_t)yt.x * sstride);
        const unsigned long long w1 = *reinterpret_cast<const unsigned long long*>(src + (ptrdiff_t)yt.y * sstride);
        uint32_t v = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const ushort2r p0 = __builtin_bit_cast(ushort2r, __builtin_amdgcn_perm((uint32_t)(w0 >> 32), (uint32_t)w0, sel[q]));
            const ushort2r p1 = __builtin_bit_cast(ushort2r, __builtin_amdgcn_perm((uint32_t)(w1 >> 32), (uint32_t)w1, sel[q]));
            const int r0 = (int)__builtin_amdgcn_udot2(p0, aw[q], 0u, false);
            const int r1 = (int)__builtin_amdgcn_udot2(p1, aw[q], 0u, false);
            uint32_t b = (uint32_t)((((yt.z * (r0 >> 4)) >> 16) + ((yt.w * (r1 >> 4)) >> 16) + 2) >> 2) & 0xffu;
            if (sxv[q] >= 0) b = 0;   // bit 31 = valid; the padding of the row stride stays 0
            v |= b << (8 * q);
        }
        const size_t off = obase + (size_t)Y * stride;
        *(uint32_t*)(pyr + off) = v;
        // un-blurred frame of the blurred pyramid, in whole 16-byte chunks as k_blur expects: frame rows, chunk 0, and
        // everything from the chunk that holds the first column right of the interior
        if (frame_col || Y < kEdge || Y >= H + kEdge) *(uint32_t*)(blur + off) = v;
    }
}

// ---------------------------------------------------------------------------------------------
// FAST-9/16 score map.  S = max over the 16 arcs of 9 contiguous circle pixels of min(v - p) and of min(p - v),
// clamped to [0, 255].  Corner at threshold t  <=>  S > t;  cv::FAST's cornerScore  =  S - 1.
// One thread per pixel of the scan area [16, w-16) x [16, h-16) of every level (tiles of 64 x 4).
// ---------------------------------------------------------------------------------------------
typedef short short2v __attribute__((ext_vector_type(2)));

// Two pixels at once in packed 16-bit lanes.  With p[k] the 16 ring pixels and v the centre,
// min_arc(v - p) = v - max_arc(p)  and  min_arc(p - v) = min_arc(p) - v,  so the extrema are taken on the raw ring values
// and only two subtractions remain:   S = max(0, v - min_k max9[k], max_k min9[k] - v).
// Every arc of 9 is an arc of 8 plus one end point, and an arc of 8 starting at an odd ring position j serves both the
// arc of 9 that starts at j - 1 and the one that starts at j:
//     m8[j] = min(p[j .. j+7]) = min(a4[j], a4[j+4]),  a4 = minima of 4 consecutive pixels (by doubling: 8 + 8 operations),
//     max over the two arcs of their minimum = min(m8[j], max(p[j-1], p[j+8])) = min3(a4[j], a4[j+4], max(p[j-1], p[j+8]))
// The three-input steps are gfx950's v_pk_minimum3_f16 / v_pk_maximum3_f16 (full rate, tools/pkminmax_probe.hip): pixel
// values 0..255 held in 16-bit lanes are positive f16 denormals, whose order is the integer order of their bit patterns,
// and minimum / maximum return one operand unchanged (f16 denormals are not flushed: .amdhsa_float_denorm_mode_16_64 3,
// the code object default) - so they ARE three-input u16 extrema.  36 packed operations per extremum instead of 47.
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ short2v min3_pk(short2v a, short2v b, short2v c) {
    return __builtin_bit_cast(short2v, __builtin_elementwise_minimum(
        __builtin_elementwise_minimum(__builtin_bit_cast(half2v, a), __builtin_bit_cast(half2v, b)), __builtin_bit_cast(half2v, c)));
}
__device__ __forceinline__ short2v max3_pk(short2v a, short2v b, short2v c) {
    return __builtin_bit_cast(short2v, __builtin_elementwise_maximum(
        __builtin_elementwise_maximum(__builtin_bit_cast(half2v, a), __builtin_bit_cast(half2v, b)), __builtin_bit_cast(half2v, c)));
}
__device__ __forceinline__ short2v fast_score_pk(const short2v p[16], short2v v) {
    short2v a2[8], A2[8], a4[8], A4[8];   // lower case: minima, upper case: maxima; index i <-> odd ring position 2 i + 1
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a2[i] = __builtin_elementwise_min(p[(2 * i + 1) & 15], p[(2 * i + 2) & 15]);
        A2[i] = __builtin_elementwise_max(p[(2 * i + 1) & 15], p[(2 * i + 2) & 15]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {         // four consecutive ring pixels from position 2 i + 1
        a4[i] = __builtin_elementwise_min(a2[i], a2[(i + 1) & 7]);
        A4[i] = __builtin_elementwise_max(A2[i], A2[(i + 1) & 7]);
    }
    short2v lo[8], hi[8];                 // per odd position: max over its two arcs of their minimum / min of their maximum
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int j = 2 * i + 1;
        lo[i] = min3_pk(a4[i], a4[(i + 2) & 7], __builtin_elementwise_max(p[(j - 1) & 15], p[(j + 8) & 15]));
        hi[i] = max3_pk(A4[i], A4[(i + 2) & 7], __builtin_elementwise_min(p[(j - 1) & 15], p[(j + 8) & 15]));
    }
    const short2v bmn = __builtin_elementwise_max(max3_pk(lo[0], lo[1], lo[2]), max3_pk(max3_pk(lo[3], lo[4], lo[5]), lo[6], lo[7]));
    const short2v bmx = __builtin_elementwise_min(min3_pk(hi[0], hi[1], hi[2]), min3_pk(min3_pk(hi[3], hi[4], hi[5]), hi[6], hi[7]));
    const short2v zero = {0, 0};
    return __builtin_elementwise_max(zero, __builtin_elementwise_max(v - bmx, bmn - v));
}

// One thread = 4 horizontally adjacent pixels of a vertical strip of kScoreRows rows.  A 7-row sliding window lives
// in registers; every input byte is loaded once per thread with 32-bit loads (three aligned dwords per row: pixels
// x0-4 .. x0+7) and EXPANDED once, when its row enters the window, into the nine byte pairs {b, b+1}, b = 1..9, as
// packed u16 lanes (v_perm_b32) - exactly the operands the two pixel pairs of the thread need from that row at every
// ring position, for the seven iterations the row stays in the window.  The window is a circular buffer with static
// slot indices: the row loop is unrolled by 7.  Workgroup = 4 waves = 4 strips; a wave spans 64 column groups of
// which the first and last are halo (their scores feed the neighbours' non-max suppression, they write nothing), and
// each strip computes one extra row above and below for the same reason (kScoreRows + 2 = 21 = 3 x 7 iterations).
//
// The kernel computes S' = S where S > 7 and S is a strict maximum over the 8-neighbours that lie in the SAME CELL
// (cv::FAST's non-max suppression inside one FAST call, ORBextractor.cpp:616-623; cells of level l tile the scan area
// [16, w-16) x [16, h-16) in steps of cellW x cellH), else 0 - and keeps only the non-zero results: every wave appends
// the 4-pixel groups that hold a survivor to the candidate list of its strip, {(y << 12) | x of the group, its four S'
// bytes}, at a slot from a wave ballot (no atomics, no score plane: a strip writes some hundred bytes instead of 4.7 KB
// and k_cell_collect reads those instead of scanning the plane).  A strip has 62 x 19 groups, which is the capacity of
// its list: it cannot overflow.
constexpr int kScoreRows = 19;
constexpr int kScoreGroups = 62;  // useful column groups per wave

// pairs {b, b+1}, b = 1..9, of a 12-byte window row {w0, w1, w2} as zero-extended u16 lanes
__device__ __forceinline__ void expand_row(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t (&e)[9]) {
    e[0] = __builtin_amdgcn_perm(w1, w0, 0x0c020c01u);   // bytes 1, 2
    e[1] = __builtin_amdgcn_perm(w1, w0, 0x0c030c02u);   // 2, 3
    e[2] = __builtin_amdgcn_perm(w1, w0, 0x0c040c03u);   // 3, 4
    e[3] = __builtin_amdgcn_perm(w2, w1, 0x0c010c00u);   // 4, 5
    e[4] = __builtin_amdgcn_perm(w2, w1, 0x0c020c01u);   // 5, 6
    e[5] = __builtin_amdgcn_perm(w2, w1, 0x0c030c02u);   // 6, 7
    e[6] = __builtin_amdgcn_perm(w2, w1, 0x0c040c03u);   // 7, 8
    e[7] = __builtin_amdgcn_perm(w2, w2, 0x0c010c00u);   // 8, 9
    e[8] = __builtin_amdgcn_perm(w2, w2, 0x0c020c01u);   // 9, 10
}
// window row r of iteration phase PH lives in slot (PH + r) % 7; pair b of that row
#define SE2_WP(r, b) __builtin_bit_cast(short2v, E[(PH + (r)) % 7][(b) - 1])
template <int PH, int C>  // C = centre byte of the first pixel of the pair inside the 12-byte window (4 or 6)
__device__ __forceinline__ short2v score_pair(const uint32_t (&E)[7][9]) {
    short2v p[16];
    p[0] = SE2_WP(6, C);      p[1] = SE2_WP(6, C + 1);  p[2] = SE2_WP(5, C + 2);  p[3] = SE2_WP(4, C + 3);
    p[4] = SE2_WP(3, C + 3);  p[5] = SE2_WP(2, C + 3);  p[6] = SE2_WP(1, C + 2);  p[7] = SE2_WP(0, C + 1);
    p[8] = SE2_WP(0, C);      p[9] = SE2_WP(0, C - 1);  p[10] = SE2_WP(1, C - 2); p[11] = SE2_WP(2, C - 3);
    p[12] = SE2_WP(3, C - 3); p[13] = SE2_WP(4, C - 3); p[14] = SE2_WP(5, C - 2); p[15] = SE2_WP(6, C - 1);
    return fast_score_pk(p, SE2_WP(3, C));
}
// c > 7 ? c : 0 and, generally, "keep a where a > b" on packed int16 lanes: (b - a) >> 15 (arithmetic) is all ones
// exactly where a > b
__device__ __forceinline__ short2v keep_greater(short2v a, short2v b) {
    const short2v m = (b - a) >> 15;
    return a & m;
}
#undef SE2_WP

constexpr int kStripCap = kScoreGroups * kScoreRows;   // 4-pixel groups of one strip

__global__ __launch_bounds__(256) void k_fast_score(Geom g, const uint8_t* __restrict__ pyr, uint2* __restrict__ lst_ent,
                                                     int* __restrict__ lst_cnt, int bx0) {
    SE2_FRAME_GRID_OFF(f, bx, bx0);
    int l = 0;
    while (l + 1 < g.nlevels && bx >= g.tile_base[l + 1]) ++l;
    const int t = bx - g.tile_base[l];
    const int H = g.h[l], stride = g.stride[l];
    const int X1 = g.sx1[l], Y1 = g.sy1[l];   // end of the scan area (w - 16, h - 16, or up to 13 px beyond: Geom::sx1)
    const int sw = X1 - kEdge;
    const int tiles_x = (sw + 4 * kScoreGroups - 1) / (4 * kScoreGroups);
    const int lane = threadIdx.x & 63;
    const int x0 = kEdge + (t % tiles_x) * (4 * kScoreGroups) + (lane - 1) * 4;  // lane 0 / 63 = halo groups
    // (the strip is the wave's: said so, its rows, the row addresses and the scan-area tests stay in scalar registers - this
    // kernel is bound by VALU issue, and every address computed per lane is VALU work)
    const int strip = __builtin_amdgcn_readfirstlane((t / tiles_x) * 4 + (int)(threadIdx.x >> 6));
    const int y0 = kEdge + strip * kScoreRows;
    if (y0 >= Y1) return;  // wave-uniform
    const bool xin = x0 >= kEdge && x0 < X1;            // this lane's group starts inside the scan area
    const uint8_t* base = pyr + pix(g, f, l, 0, 0);
    // candidate list of this strip
    const size_t lst = (size_t)f * g.lst_base[g.nlevels] + g.lst_base[l] + strip * tiles_x + (t % tiles_x);
    uint2* ent = lst_ent + lst * kStripCap;
    int nent = 0;                                              // wave-uniform
    const int xc = min(max(x0, kEdge), X1 - 1) & ~3;    // clamped (aligned) load position for halo lanes outside
    uint32_t E[7][9];
    const int ymax = H + kEdge - 1;   // last row of the bordered plane (strips at the bottom clamp their look-ahead)
    // window rows y0-4 .. y0+1 (slots 0..5) so that the first computed score row is y0-1
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const uint32_t* p = (const uint32_t*)(base + (ptrdiff_t)min(y0 - 4 + r, ymax) * stride + xc - 4);
        expand_row(p[0], p[1], p[2], E[r]);
    }
    const int yend = min(y0 + kScoreRows, Y1);
    const int nvalid = xin ? min(4, X1 - x0) : 0;
    // cell-edge flags of the 4 pixels (horizontal): bit q of eL / eR
    const int cellW = g.cellW[l], cellH = g.cellH[l];
    unsigned eL = 0, eR = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int xr = x0 + q - kEdge;
        if (xr >= 0 && xr % cellW == 0) eL |= 1u << q;
        if (xr >= 0 && (xr % cellW == cellW - 1 || x0 + q == X1 - 1)) eR |= 1u << q;
    }
    // Non-max suppression state, all in packed int16 pairs A = pixels {0,1}, B = pixels {2,3} of this thread:
    //   c   scores with S <= 7 already zeroed (a score <= 7 can neither win nor beat a winner, so this changes nothing)
    //   Hc  max over the left / right neighbour that lies in the same cell        (centre-row contribution)
    //   H3  max(Hc, c)                                                            (contribution of the row above / below)
    // mLA.. are the per-thread cell-edge masks (all ones where that neighbour is inside the cell), vA / vB the valid-
    // pixel masks of the last column group.
    auto pairmask = [](bool lo, bool hi) { return (uint32_t)(lo ? 0xffffu : 0u) | (hi ? 0xffff0000u : 0u); };
    const uint32_t mLA = pairmask(!(eL & 1), !(eL & 2)), mLB = pairmask(!(eL & 4), !(eL & 8));
    const uint32_t mRA = pairmask(!(eR & 1), !(eR & 2)), mRB = pairmask(!(eR & 4), !(eR & 8));
    const uint32_t vA = pairmask(nvalid > 0, nvalid > 1), vB = pairmask(nvalid > 2, nvalid > 3);
    const short2v seven = {7, 7};
    uint32_t H3ppA = 0, H3ppB = 0, H3pA = 0, H3pB = 0, HcpA = 0, HcpB = 0, cpA = 0, cpB = 0;
    // row of the decided output row yo = y - 1 inside its cell, kept incrementally (one modulo here instead of two per row)
    int ymod = (((y0 - 2 - kEdge) % cellH) + cellH) % cellH;
    auto step = [&](auto phc, int y) {
        constexpr int PH = decltype(phc)::value;
        {   // row y+3 enters the window (slot of window row 6)
            const uint32_t* p = (const uint32_t*)(base + (ptrdiff_t)min(y + 3, ymax) * stride + xc - 4);
            expand_row(p[0], p[1], p[2], E[(PH + 6) % 7]);
        }
        uint32_t cA = 0, cB = 0;
        if (y >= kEdge && y < Y1) {   // rows outside the scan area score 0
            cA = __builtin_bit_cast(uint32_t, keep_greater(score_pair<PH, 4>(E), seven)) & vA;
            cB = __builtin_bit_cast(uint32_t, keep_greater(score_pair<PH, 6>(E), seven)) & vB;
        }
        const uint32_t lB = __shfl_up(cB, 1), rA = __shfl_down(cA, 1);
        const uint32_t LA = (cA << 16) | (lane == 0 ? 0u : lB >> 16);        // {s-1, s0}
        const uint32_t MID = (cA >> 16) | (cB << 16);                         // {s1, s2}
        const uint32_t RB = (cB >> 16) | (lane == 63 ? 0u : rA << 16);        // {s3, s4}
        const short2v hcA = __builtin_elementwise_max(__builtin_bit_cast(short2v, LA & mLA), __builtin_bit_cast(short2v, MID & mRA));
        const short2v hcB = __builtin_elementwise_max(__builtin_bit_cast(short2v, MID & mLB), __builtin_bit_cast(short2v, RB & mRB));
        const uint32_t HcA = __builtin_bit_cast(uint32_t, hcA), HcB = __builtin_bit_cast(uint32_t, hcB);
        const uint32_t H3A = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(hcA, __builtin_bit_cast(short2v, cA)));
        const uint32_t H3B = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(hcB, __builtin_bit_cast(short2v, cB)));
        const int yo = y - 1;  // row whose suppression can now be decided
        uint32_t o4 = 0;   // the four S' bytes of this thread's group in row yo
        if (yo >= y0 && yo < yend && lane >= 1 && lane <= kScoreGroups && nvalid > 0) {
            const uint32_t tT = (ymod == 0) ? 0u : 0xffffffffu;                                            // row above in the cell?
            const uint32_t tB = ((ymod == cellH - 1) || yo == Y1 - 1) ? 0u : 0xffffffffu;          // row below?
            const short2v mA = __builtin_elementwise_max(__builtin_elementwise_max(__builtin_bit_cast(short2v, HcpA),
                                                                                     __builtin_bit_cast(short2v, H3ppA & tT)),
                                                         __builtin_bit_cast(short2v, H3A & tB));
            const short2v mB = __builtin_elementwise_max(__builtin_elementwise_max(__builtin_bit_cast(short2v, HcpB),
                                                                                     __builtin_bit_cast(short2v, H3ppB & tT)),
                                                         __builtin_bit_cast(short2v, H3B & tB));
            const uint32_t oA = __builtin_bit_cast(uint32_t, keep_greater(__builtin_bit_cast(short2v, cpA), mA));
            const uint32_t oB = __builtin_bit_cast(uint32_t, keep_greater(__builtin_bit_cast(short2v, cpB), mB));
            o4 = __builtin_amdgcn_perm(oB, oA, 0x06040200u);
        }
        const unsigned long long hit = __ballot(o4 != 0);
        if (o4 != 0) ent[nent + __builtin_amdgcn_mbcnt_hi((uint32_t)(hit >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hit, 0))] =
            make_uint2(((uint32_t)yo << 12) | (uint32_t)x0, o4);
        nent += __popcll(hit);
        H3ppA = H3pA; H3ppB = H3pB;
        H3pA = H3A; H3pB = H3B; HcpA = HcA; HcpB = HcB; cpA = cA; cpB = cB;
        ymod = (ymod + 1 == cellH) ? 0 : ymod + 1;
    };
    for (int y = y0 - 1; y <= yend; y += 7) {   // rows past yend only feed guarded code
        step(std::integral_constant<int, 0>{}, y);
        step(std::integral_constant<int, 1>{}, y + 1);
        step(std::integral_constant<int, 2>{}, y + 2);
        step(std::integral_constant<int, 3>{}, y + 3);
        step(std::integral_constant<int, 4>{}, y + 4);
        step(std::integral_constant<int, 5>{}, y + 5);
        step(std::integral_constant<int, 6>{}, y + 6);
    }
    if (lane == 0) lst_cnt[lst] = nent;
}

// ---------------------------------------------------------------------------------------------
// k_fast_score_sparse: the same plane as k_fast_score (S' = S where S > 7 and a strict in-cell maximum, else 0), computed
// only where it can be non-zero.  A 9-arc of the 16-ring always contains two ADJACENT compass points (ring positions
// 0, 4, 8, 12), so S > 7 needs both of them brighter than v + 7 or both darker than v - 7: a 4-pixel test that rejects
// flat areas, noise and straight edges (typically > 90 % of the pixels).  One workgroup = one 128 x 32 tile staged in
// LDS with its halo:
//   1  compass test for every pixel of the tile + 1 px (packed int16 pairs, 4 pixels per thread and step); survivors
//      are appended to a candidate list in LDS
//   2  the full FAST-9 score for the candidates only, two per lane in the packed arithmetic of fast_score_pk, ring
//      bytes gathered from the LDS tile; scores <= 7 dropped; written into a sparse LDS score tile
//   3  in-cell non-max suppression per surviving candidate against its 8 neighbours in that tile (raw scores, as cv::FAST);
//      survivors go to the candidate list of the tile, in k_fast_score's entry format with one pixel per entry
// Dense corners only cost time, never correctness: the candidate list holds every pixel of the tile if need be.
// On the benchmark texture (4000 overlapping rectangles: 15 % of the pixels pass the compass test, 6 % have S > 7) this
// kernel takes 737 us per 256 frames against 631 us for k_fast_score, which is therefore the default; the two meet at
// about one candidate in ten pixels, and on imagery with fewer corners this one's cost tends to the compass pass alone
// (about a fifth of the dense kernel's arithmetic).  The extractor picks between the two from the candidate density this
// kernel measures (se2gpu_orb handle, score_mode); SE2GPU_ORB_SCORE=dense / sparse pins one.
// ---------------------------------------------------------------------------------------------
constexpr int kFsTW = 128, kFsTH = 32;
constexpr int kFsLW = kFsTW + 16;          // LDS image tile: columns x0-8 .. x0+TW+8
constexpr int kFsLH = kFsTH + 8;           //                 rows    y0-4 .. y0+TH+4
constexpr int kFsSW = kFsTW + 8;           // score tile: columns x0-4 .. x0+TW+4 (whole 4-pixel groups), rows y0-1 .. y0+TH
constexpr int kFsSH = kFsTH + 2;
constexpr int kFsMaxCand = kFsSW * kFsSH;
constexpr int kFsListCap = 2048;           // survivors one tile can list (in-cell maxima are never adjacent: <= ~1/4 of the
                                           // 4096 pixels plus cell-boundary effects; more raises the overflow flag)

// candidate bits of a pixel pair: bit 15 / bit 31 set where two adjacent compass points are both >= v + 8 or <= v - 8
__device__ __forceinline__ uint32_t compass_pair(short2v v, short2v n, short2v e, short2v s, short2v w) {
    const short2v dn = n - v, de = e - v, ds = s - v, dw = w - v;
    const short2v mx = __builtin_elementwise_max(
        __builtin_elementwise_max(__builtin_elementwise_min(dn, de), __builtin_elementwise_min(de, ds)),
        __builtin_elementwise_max(__builtin_elementwise_min(ds, dw), __builtin_elementwise_min(dw, dn)));
    const short2v mn = __builtin_elementwise_min(
        __builtin_elementwise_min(__builtin_elementwise_max(dn, de), __builtin_elementwise_max(de, ds)),
        __builtin_elementwise_min(__builtin_elementwise_max(ds, dw), __builtin_elementwise_max(dw, dn)));
    const short2v eight = {8, 8}, zero = {0, 0};
    const short2v a = mx - eight;               // >= 0: bright candidate
    const short2v b = zero - mn - eight;        // >= 0: dark candidate
    return ~(__builtin_bit_cast(uint32_t, a) & __builtin_bit_cast(uint32_t, b)) & 0x80008000u;
}

__global__ __launch_bounds__(256) void k_fast_score_sparse(Geom g, const uint8_t* __restrict__ pyr, uint2* __restrict__ lst_ent,
                                                           int* __restrict__ lst_cnt, int* __restrict__ cand_count,
                                                           int* __restrict__ overflow, int bx0) {
    __shared__ uint32_t s_img[(kFsLW / 4) * kFsLH];
    __shared__ uint32_t s_sc[(kFsSW / 4) * kFsSH];
    __shared__ uint16_t s_cand[kFsMaxCand];
    __shared__ int s_n, s_nout;
    SE2_FRAME_GRID_OFF(f, tlin, bx0);
    int l = 0;
    while (l + 1 < g.nlevels && tlin >= g.tile_base[l + 1]) ++l;
    const int t = tlin - g.tile_base[l];
    const int H = g.h[l], stride = g.stride[l];
    const int X1 = g.sx1[l], Y1 = g.sy1[l];   // end of the scan area (Geom::sx1)
    const int tiles_x = (X1 - kEdge + kFsTW - 1) / kFsTW;
    const int x0 = kEdge + (t % tiles_x) * kFsTW, y0 = kEdge + (t / tiles_x) * kFsTH;
    const int tid = threadIdx.x;
    const uint8_t* plane = pyr + (size_t)f * g.frame_bytes + g.off[l];   // bordered plane: pixel (x, y) at (y+16)*stride + x+16
    const size_t lst = (size_t)f * g.lst_base[g.nlevels] + g.lst_base[l] + t;
    uint2* ent = lst_ent + lst * kFsListCap;
    constexpr int LWd = kFsLW / 4, SWd = kFsSW / 4;
    // 0. stage the image tile (rows / columns outside the bordered plane are clamped: they only feed pixels that are
    //    not scored), clear the sparse tiles
    const int last_row = H + 2 * kEdge - 1, last_dw = stride / 4 - 1;
    for (int i = tid; i < LWd * kFsLH; i += 256) {
        const int r = i / LWd, c = i - r * LWd;
        const int by = min(max(y0 - 4 + r + kEdge, 0), last_row);
        const int bx = min(max((x0 - 8 + kEdge) / 4 + c, 0), last_dw);
        s_img[i] = *(const uint32_t*)(plane + (size_t)by * stride + 4 * bx);
    }
    for (int i = tid; i < SWd * kFsSH; i += 256) s_sc[i] = 0;
    if (tid == 0) { s_n = 0; s_nout = 0; }
    __syncthreads();
    // 1. compass test.  Thread = (column group gx = tid & 31, row phase tid >> 5): the 128 tile columns, rows -1 .. TH.
    const int xlo = max(kEdge, x0 - 1), xhi = min(X1, x0 + kFsTW + 1);   // columns that need a score
    {
        const int gx = (tid & 31) + 1;                                   // score-tile group: pixels x0 + 4 (gx - 1) ..
        const int xg = x0 - 4 + 4 * gx;
        unsigned colmask = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (xg + q >= xlo && xg + q < xhi) colmask |= 1u << q;
        for (int gy = tid >> 5; gy < kFsSH; gy += 8) {
            const int y = y0 + gy - 1;
            if (y < kEdge || y >= Y1 || !colmask) continue;
            const uint32_t* row = s_img + (gy + 3) * LWd + gx;           // row y, columns xg-4 .. xg+7
            const uint32_t w0 = row[0], w1 = row[1], w2 = row[2];
            const uint32_t n1 = row[1 - 3 * LWd], s1 = row[1 + 3 * LWd];  // rows y-3 / y+3, columns xg .. xg+3
            const short2v vA = __builtin_bit_cast(short2v, __builtin_amdgcn_perm(w1, w1, 0x0c010c00u));   // pixels 0, 1
            const short2v vB = __builtin_bit_cast(short2v, __builtin_amdgcn_perm(w1, w1, 0x0c030c02u));   // pixels 2, 3
            const short2v eA = __builtin_bit_cast(short2v, __builtin_amdgcn_perm(w2, w1, 0x0c040c03u));   // x+3: bytes 7, 8
            const short2v eB = __builtin_bit_cast(short2v, __builtin_amdgcn_perm(w2, w2, 0x0c020c01u));   //      bytes 9, 10
            const short2v wA = __builtin_bit_cast(short2v, __builtin_amdgcn_perm(w0, w0, 0x0c020c01u));   // x-3: bytes 1, 2
            const short2v wB = __builtin_bit_cast(short2v, __builtin_amdgcn_perm(w1, w0, 0x0c040c03u));   //      bytes 3, 4
            const short2v nA = __builtin_bit_cast(short2v, __builtin_amdgcn_perm(n1, n1, 0x0c010c00u));
            const short2v nB = __builtin_bit_cast(short2v, __builtin_amdgcn_perm(n1, n1, 0x0c030c02u));
            const short2v sA = __builtin_bit_cast(short2v, __builtin_amdgcn_perm(s1, s1, 0x0c010c00u));
            const short2v sB = __builtin_bit_cast(short2v, __builtin_amdgcn_perm(s1, s1, 0x0c030c02u));
            const uint32_t cA = compass_pair(vA, nA, eA, sA, wA), cB = compass_pair(vB, nB, eB, sB, wB);
            const unsigned flags = (((cA >> 15) & 1u) | ((cA >> 30) & 2u) | ((cB >> 13) & 4u) | ((cB >> 28) & 8u)) & colmask;
            if (flags) {
                int at = atomicAdd(&s_n, __popc(flags));
                const int code = (gy << 8) | (4 * gx);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (flags & (1u << q)) s_cand[at++] = (uint16_t)(code + q);
            }
        }
    }
    const uint8_t* img8 = (const uint8_t*)s_img;
    if (tid < 2 * kFsSH) {   // the two halo columns x0 - 1 and x0 + TW, one pixel per thread
        const int side = tid >= kFsSH, gy = tid - side * kFsSH;
        const int sx = side ? kFsTW + 4 : 3;
        const int x = x0 - 4 + sx, y = y0 + gy - 1;
        if (y >= kEdge && y < Y1 && x >= xlo && x < xhi) {
            const int c = (gy + 3) * kFsLW + sx + 4;
            const int v = img8[c];
            const int dn = img8[c - 3 * kFsLW] - v, ds = img8[c + 3 * kFsLW] - v, de = img8[c + 3] - v, dw = img8[c - 3] - v;
            const int mx = max(max(min(dn, de), min(de, ds)), max(min(ds, dw), min(dw, dn)));
            const int mn = min(min(max(dn, de), max(de, ds)), min(max(ds, dw), max(dw, dn)));
            if (mx >= 8 || mn <= -8) s_cand[atomicAdd(&s_n, 1)] = (uint16_t)((gy << 8) | sx);
        }
    }
    __syncthreads();
    // 2. FAST-9 score of the candidates, two per lane.  Candidate code = (score-tile row << 8) | score-tile column.
    const int n = s_n;
    if (tid == 0 && cand_count) atomicAdd(&cand_count[f], n);   // candidate density of the frame: picks the kernel
    uint8_t* sc8 = (uint8_t*)s_sc;
    const short2v seven = {7, 7};
    for (int b = 2 * tid; b < n; b += 512) {
        const int k0 = s_cand[b], k1 = (b + 1 < n) ? s_cand[b + 1] : k0;
        const int c0 = ((k0 >> 8) + 3) * kFsLW + (k0 & 255) + 4;        // byte of the centre pixel in the image tile
        const int c1 = ((k1 >> 8) + 3) * kFsLW + (k1 & 255) + 4;
        auto pk = [&](int dy, int dx) {
            const int o = dy * kFsLW + dx;
            return __builtin_bit_cast(short2v, (uint32_t)img8[c0 + o] | ((uint32_t)img8[c1 + o] << 16));
        };
        short2v p[16];
        p[0] = pk(3, 0);    p[1] = pk(3, 1);    p[2] = pk(2, 2);    p[3] = pk(1, 3);
        p[4] = pk(0, 3);    p[5] = pk(-1, 3);   p[6] = pk(-2, 2);   p[7] = pk(-3, 1);
        p[8] = pk(-3, 0);   p[9] = pk(-3, -1);  p[10] = pk(-2, -2); p[11] = pk(-1, -3);
        p[12] = pk(0, -3);  p[13] = pk(1, -3);  p[14] = pk(2, -2);  p[15] = pk(3, -1);
        const short2v sc = keep_greater(fast_score_pk(p, pk(0, 0)), seven);
        sc8[(k0 >> 8) * kFsSW + (k0 & 255)] = (uint8_t)sc.x;
        if (b + 1 < n) sc8[(k1 >> 8) * kFsSW + (k1 & 255)] = (uint8_t)sc.y;
    }
    __syncthreads();
    // 3. non-max suppression inside the cell (cv::FAST compares raw scores of the neighbours of one FAST call = one cell)
    const int cellW = g.cellW[l], cellH = g.cellH[l];
    const int xr0 = (x0 - kEdge) % cellW, yr0 = (y0 - kEdge) % cellH;    // workgroup-uniform
    for (int b = tid; b < n; b += 256) {
        const int k = s_cand[b];
        const int sy = k >> 8, sx = k & 255;
        const int lx = sx - 4, ly = sy - 1;
        if (lx < 0 || lx >= kFsTW || ly < 0 || ly >= kFsTH) continue;     // halo: another tile's pixel
        const int i = sy * kFsSW + sx;
        const int c = sc8[i];
        if (!c) continue;
        int xm = xr0 + lx, ym = yr0 + ly;                                  // position inside the cell
        while (xm >= cellW) xm -= cellW;
        while (ym >= cellH) ym -= cellH;
        const bool L = xm != 0, R = !(xm == cellW - 1 || x0 + lx == X1 - 1);
        const bool T = ym != 0, B = !(ym == cellH - 1 || y0 + ly == Y1 - 1);
        int m = 0;
        if (L) m = max(m, (int)sc8[i - 1]);
        if (R) m = max(m, (int)sc8[i + 1]);
        if (T) {
            m = max(m, (int)sc8[i - kFsSW]);
            if (L) m = max(m, (int)sc8[i - kFsSW - 1]);
            if (R) m = max(m, (int)sc8[i - kFsSW + 1]);
        }
        if (B) {
            m = max(m, (int)sc8[i + kFsSW]);
            if (L) m = max(m, (int)sc8[i + kFsSW - 1]);
            if (R) m = max(m, (int)sc8[i + kFsSW + 1]);
        }
        if (c > m && x0 + lx < X1 && y0 + ly < Y1) {
            const int slot = atomicAdd(&s_nout, 1);
            if (slot < kFsListCap) ent[slot] = make_uint2(((uint32_t)(y0 + ly) << 12) | (uint32_t)(x0 + lx), (uint32_t)c);
        }
    }
    __syncthreads();
    if (tid == 0) {
        if (s_nout > kFsListCap) atomicOr(overflow, 1);
        lst_cnt[lst] = min(s_nout, kFsListCap);
    }
}

// ---------------------------------------------------------------------------------------------
// Key-point retention (ORBextractor.cpp:616-710).  cv::FAST hands every cell its corners in row-major order; the reference
// then cuts each cell, and afterwards the level's concatenated list, with KeyPointsFilter::retainBest(v, n) + v.resize(n):
// std::nth_element by response and the first n entries of the permutation it leaves.  Which of several equal-response
// corners survive, and the order in which the survivors reach the output (MatchByWindow is order-dependent), are therefore
// properties of libstdc++'s introselect - a fixed function of the input - and the kernels below compute exactly that
// permutation (rounds 1-4 sorted by (response, y, x) instead, which keeps the same points only up to ties):
//   k_cell_collect  one workgroup per cell: the cell's FAST output (threshold 20, or 7 if that gave <= 3 corners) in row-major
//                   order -> the cell's list, + its size
//   k_cell_retain   one wave per cell: the level's quota redistribution (:631-679) replayed, then introselect on the list
//   k_level_select  one workgroup per (frame, level): the cells' first nToRetain entries in cell order, introselect with the
//                   level's quota, key points out in that order
// A list entry: FAST_SCORE  uint32  S << 24 | y << 12 | x         (response = S - 1 = cornerScore)
//               HARRIS      uint64  ordered(response) << 32 | y << 12 | x
// and the comparator of retainBest, a.response > b.response, is key(a) > key(b) on the high part.
// ---------------------------------------------------------------------------------------------
// HarrisResponses(cellImage, pts, 7, 0.04f) of ORBextractor.cpp:85-126 for one key point (x, y) of the un-blurred level:
// 7x7 block of 3x3 Sobel windows, integer sums, then the reference's float expression (compiled without contraction).
__device__ __forceinline__ float harris_response(const uint8_t* __restrict__ lvl, int stride, int x, int y) {
    const uint8_t* p0 = lvl + (ptrdiff_t)(y - 3) * stride + (x - 3);
    int a = 0, b = 0, c = 0;
    for (int i = 0; i < 7; ++i) {
        const uint8_t* r0 = p0 + (ptrdiff_t)(i - 1) * stride;
        const uint8_t* r1 = r0 + stride;
        const uint8_t* r2 = r1 + stride;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int Ix = ((int)r1[j + 1] - (int)r1[j - 1]) * 2 + ((int)r0[j + 1] - (int)r0[j - 1]) + ((int)r2[j + 1] - (int)r2[j - 1]);
            const int Iy = ((int)r2[j] - (int)r0[j]) * 2 + ((int)r2[j - 1] - (int)r0[j - 1]) + ((int)r2[j + 1] - (int)r0[j + 1]);
            a += Ix * Ix;
            b += Iy * Iy;
            c += Ix * Iy;
        }
    }
    float scale = (1 << 2) * 7 * 255.0f;
    scale = 1.0f / scale;
    const float scale_sq_sq = scale * scale * scale * scale;
    return ((float)a * b - (float)c * c - 0.04f * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
}

// se2gpu_orb_debug_score: the S' plane of one level of one frame, rebuilt from the candidate lists (plane zeroed before)
__global__ __launch_bounds__(256) void k_scatter_lists(Geom g, int f, int l, const uint2* __restrict__ lst_ent,
                                                        const int* __restrict__ lst_cnt, uint8_t* __restrict__ plane) {
    const int id = g.lst_base[l] + (int)blockIdx.x;
    if (id >= g.lst_base[l + 1]) return;
    const size_t lst = (size_t)f * g.lst_base[g.nlevels] + id;
    const int n = lst_cnt[lst];
    for (int e = threadIdx.x; e < n; e += 256) {
        const uint2 en = lst_ent[lst * g.lst_cap + e];
        const int y = (int)(en.x >> 12), x0 = (int)(en.x & 0xfffu);
        for (int q = 0; q < 4; ++q) {
            const uint32_t sc = (en.y >> (8 * q)) & 0xffu;
            if (sc) plane[(size_t)y * g.w[l] + x0 + q] = (uint8_t)sc;
        }
    }
}

// ---- list entries
template <bool HARRIS> struct Ent;
template <> struct Ent<false> {
    using T = uint32_t;
    static __device__ __forceinline__ uint32_t key(T e) { return e >> 24; }
    static __device__ __forceinline__ uint32_t pos(T e) { return e & 0x00ffffffu; }
    static __device__ __forceinline__ float resp(T e) { return (float)((int)(e >> 24) - 1); }   // cornerScore = S - 1
};
template <> struct Ent<true> {
    using T = unsigned long long;
    static __device__ __forceinline__ uint32_t key(T e) { return (uint32_t)(e >> 32); }
    static __device__ __forceinline__ uint32_t pos(T e) { return (uint32_t)e & 0x00ffffffu; }
    static __device__ __forceinline__ float resp(T e) {
        uint32_t b = (uint32_t)(e >> 32);
        b ^= (b >> 31) ? 0x80000000u : 0xffffffffu;
        return __uint_as_float(b);
    }
    static __device__ __forceinline__ T make(float r, uint32_t pos) {   // float order -> unsigned order (-0 counts as +0)
        uint32_t b = __float_as_uint(r);
        if (b == 0x80000000u) b = 0;
        b ^= (b >> 31) ? 0xffffffffu : 0x80000000u;
        return ((T)b << 32) | pos;
    }
};

// memory written by some lanes of a wave and read by others afterwards (LDS or global): order it
__device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

// ---- libstdc++'s heap select (bits/stl_heap.h + __heap_select), the branch introselect takes when its depth limit runs out.
// Executed by ONE lane: it is reached on adversarial inputs only (median-of-three killers), never on image data.
template <bool H>
__device__ void seq_adjust_heap(typename Ent<H>::T* first, int hole, int len, typename Ent<H>::T value) {
    using E = Ent<H>;
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (E::key(first[child]) > E::key(first[child - 1])) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;                     // __push_heap
    while (hole > top && E::key(first[parent]) > E::key(value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}
template <bool H>
__device__ void seq_heap_select(typename Ent<H>::T* first, int middle, int last) {   // offsets relative to first
    using E = Ent<H>;
    const int len = middle;
    if (len >= 2)
        for (int parent = (len - 2) / 2;; --parent) {   // __make_heap
            seq_adjust_heap<H>(first, parent, len, first[parent]);
            if (parent == 0) break;
        }
    for (int i = middle; i < last; ++i)
        if (E::key(first[i]) > E::key(first[0])) {      // __pop_heap(first, middle, i)
            const typename E::T v = first[i];
            first[i] = first[0];
            seq_adjust_heap<H>(first, 0, len, v);
        }
}

// ---- the rounds of introselect once its range fits the wave: one element per lane in a register (lane p holds a[base + p]),
// the same steps as introselect_wave below with lane shuffles instead of LDS traffic - a third of the instructions, and most
// cells (~100 corners) spend all but their first round here.  Returns false when the depth limit ran out (the range is
// written back, the caller takes the heap-select branch); true when the selection is finished.
__device__ __forceinline__ uint32_t lane_read(uint32_t v, int i) { return (uint32_t)__builtin_amdgcn_readlane((int)v, i); }
__device__ __forceinline__ unsigned long long lane_read(unsigned long long v, int i) {
    return ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), i) << 32) |
           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, i);
}
template <bool H, class PosT>
__device__ bool introselect_tail(typename Ent<H>::T* a, PosT* lp, PosT* rp, int& first, int& last, const int nth, int& depth) {
    using E = Ent<H>;
    using T = typename E::T;
    const int lane = (int)(threadIdx.x & 63);
    const unsigned long long lt = (1ull << lane) - 1ull, gt = lane == 63 ? 0ull : (~0ull << (lane + 1));
    const int base = first, len0 = last - first;
    T v = lane < len0 ? a[base + lane] : (T)0;
    bool done = true;
    while (last - first > 3) {
        if (depth == 0) { done = false; break; }
        --depth;
        const int i0 = __builtin_amdgcn_readfirstlane(first - base), hi = __builtin_amdgcn_readfirstlane(last - base);
        const int ia = i0 + 1, ib = i0 + (hi - i0) / 2, ic = hi - 1;
        const T ea = lane_read(v, ia), eb = lane_read(v, ib), ec = lane_read(v, ic), er = lane_read(v, i0);
        const uint32_t ka = E::key(ea), kb = E::key(eb), kc = E::key(ec);
        int src;                                              // __move_median_to_first(first, first + 1, mid, last - 1)
        if (ka > kb) src = (kb > kc) ? ib : (ka > kc) ? ic : ia;
        else src = (ka > kc) ? ia : (kb > kc) ? ic : ib;
        const T es = src == ia ? ea : src == ib ? eb : ec;
        if (lane == src) v = er;
        if (lane == i0) v = es;
        const uint32_t pk = E::key(es);
        const bool inr = lane > i0 && lane < hi;
        const uint32_t k = E::key(v);
        const bool isL = inr && !(k > pk), isR = inr && !(pk > k);
        const unsigned long long mL = __builtin_amdgcn_ballot_w64(isL), mR = __builtin_amdgcn_ballot_w64(isR);
        const int cL = __popcll(mL), cR = __popcll(mR);
        const int rL = __popcll(mL & lt), rR = __popcll(mR & gt);     // rank from the left / from the right
        if (isL) lp[rL] = (PosT)lane;
        if (isR) rp[rR] = (PosT)lane;
        wave_fence();
        const int m = min(cL, cR);
        const bool ok = lane < m && (int)lp[lane] < (int)rp[lane];
        const int K = __popcll(__builtin_amdgcn_ballot_w64(ok));
        int from = lane;                                      // the K swaps as one permutation
        if (isL && rL < K) from = (int)rp[rL];
        else if (isR && rR < K) from = (int)lp[rR];
        v = __shfl(v, from);
        int cut = K > 0 ? (int)rp[K - 1] : hi;
        if (K < cL) cut = min(cut, (int)lp[K]);
        cut = __builtin_amdgcn_readfirstlane(cut) + base;
        wave_fence();
        if (cut <= nth) first = cut;
        else last = cut;
    }
    if (done) {                                               // __insertion_sort on at most three elements
        const int i0 = __builtin_amdgcn_readfirstlane(first - base), cnt = __builtin_amdgcn_readfirstlane(last - first);
        T e[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) e[q] = lane_read(v, min(i0 + q, 63));
        if (cnt >= 2 && E::key(e[1]) > E::key(e[0])) { const T t = e[0]; e[0] = e[1]; e[1] = t; }
        if (cnt >= 3) {
            if (E::key(e[2]) > E::key(e[0])) { const T t = e[2]; e[2] = e[1]; e[1] = e[0]; e[0] = t; }
            else if (E::key(e[2]) > E::key(e[1])) { const T t = e[2]; e[2] = e[1]; e[1] = t; }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q)
            if (q < cnt && lane == i0 + q) v = e[q];
    }
    if (lane < len0) a[base + lane] = v;
    wave_fence();
    return done;
}

// ---- std::nth_element(a, a + nth, a + n, response greater) of libstdc++ (bits/stl_algo.h __introselect), by one wave.
// The control flow is the library's, step for step: depth limit 2 lg n, median of (first + 1, mid, last - 1) moved to first,
// the unguarded Hoare partition of [first + 1, last) around it, the side that holds nth kept, insertion sort below four
// elements.  Only the partition is spread over the lanes, and it is the same partition: the sequential loop swaps the k-th
// element from the left that is not greater than the pivot with the k-th element from the right that is not smaller, for
// k = 1, 2, ... while the former lies left of the latter (the elements it passes over never move, so "k-th" can be counted
// in the array as it stood), and returns where the left scan stops next: the (K+1)-th such element from the left, or the
// place the K-th swap put one if that comes first.  lp / rp: scratch for the two position lists (n entries each).
template <bool H, class PosT>
__device__ void introselect_wave(typename Ent<H>::T* a, PosT* lp, PosT* rp, int n, int nth) {
    using E = Ent<H>;
    using T = typename E::T;
    const int lane = (int)(threadIdx.x & 63);
    const unsigned long long lt = (1ull << lane) - 1ull;
    int first = 0, last = n;
    int depth = 2 * (31 - __clz(n));
    while (last - first > 3) {
        if (last - first <= 64 && depth > 0 && introselect_tail<H, PosT>(a, lp, rp, first, last, nth, depth)) return;
        if (depth == 0) {
            if (lane == 0) {
                seq_heap_select<H>(a + first, nth + 1 - first, last - first);
                const T t = a[first];
                a[first] = a[nth];
                a[nth] = t;
            }
            wave_fence();
            return;
        }
        --depth;
        if (lane == 0) {   // __move_median_to_first(first, first + 1, mid, last - 1)
            const int ia = first + 1, ib = first + (last - first) / 2, ic = last - 1;
            const T ea = a[ia], eb = a[ib], ec = a[ic], er = a[first];
            const uint32_t ka = E::key(ea), kb = E::key(eb), kc = E::key(ec);
            int src;
            if (ka > kb) src = (kb > kc) ? ib : (ka > kc) ? ic : ia;
            else src = (ka > kc) ? ia : (kb > kc) ? ic : ib;
            a[first] = src == ia ? ea : src == ib ? eb : ec;
            a[src] = er;
        }
        wave_fence();
        const uint32_t pk = E::key(a[first]);
        const int lo = first + 1, hi = last;
        int cL = 0, cR = 0;
        for (int b = lo; b < hi; b += 64) {
            const int p = b + lane;
            bool isL = false, isR = false;
            if (p < hi) {
                const uint32_t k = E::key(a[p]);
                isL = !(k > pk);     // the left scan `while (comp(*first, *pivot)) ++first` stops here
                isR = !(pk > k);     // the right scan `while (comp(*pivot, *last)) --last` stops here
            }
            const unsigned long long mL = __builtin_amdgcn_ballot_w64(isL), mR = __builtin_amdgcn_ballot_w64(isR);
            if (isL) lp[cL + __popcll(mL & lt)] = (PosT)p;
            if (isR) rp[cR + __popcll(mR & lt)] = (PosT)p;   // counted from the left; the k-th from the right is rp[cR - 1 - k]
            cL += __popcll(mL);
            cR += __popcll(mR);
        }
        wave_fence();
        const int m = min(cL, cR);
        int K = 0;                                            // swaps the sequential loop performs
        for (int b = 0; b < m; b += 64) {
            const int k = b + lane;
            const bool ok = k < m && (int)lp[k] < (int)rp[cR - 1 - k];
            const int c = __popcll(__builtin_amdgcn_ballot_w64(ok));   // (true on a prefix: lp ascends, rp[cR-1-k] descends)
            K += c;
            if (c < 64) break;
        }
        for (int b = 0; b < K; b += 64) {
            const int k = b + lane;
            if (k < K) {
                const int i = (int)lp[k], j = (int)rp[cR - 1 - k];
                const T ei = a[i], ej = a[j];
                a[i] = ej;
                a[j] = ei;
            }
        }
        int cut = K > 0 ? (int)rp[cR - K] : hi;
        if (K < cL) cut = min(cut, (int)lp[K]);
        wave_fence();
        if (cut <= nth) first = cut;
        else last = cut;
    }
    if (lane == 0)                                            // __insertion_sort on at most three elements
        for (int i = first + 1; i < last; ++i) {
            const T v = a[i];
            int hole = i;
            while (hole > first && E::key(v) > E::key(a[hole - 1])) {
                a[hole] = a[hole - 1];
                --hole;
            }
            a[hole] = v;
        }
    wave_fence();
}

// bitonic sort of npad (a power of two) uint32 keys by the 256 threads of a workgroup; LDS or global memory
__device__ __forceinline__ void bitonic_sort_u32(uint32_t* keys, int npad) {
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
}

// ---------------------------------------------------------------------------------------------
// k_cell_collect_w: what cv::FAST returns for the cell, in the order it returns it - one WAVE per cell (four cells per
// workgroup, no barriers).  The kernel is bound by the instructions a wave issues per cell (33 k cells per batch), so:
//   * the cell's rectangle, its candidate lists and its list in HBM come from a table built with the geometry (CellGeo: the
//     dozen integer divisions they take cost more than the rest of the preamble);
//   * all in-cell candidates (S > 7) go into ONE array, one ballot per pixel column; the FAST(20) / FAST(7) choice
//     (ORBextractor.cpp:616-623) is a count taken on the way and a filter afterwards;
//   * row-major order by a counting sort over the cell's rows (an ordinary cell: ~100 corners on ~75 rows), then a rank
//     inside each row's few entries - a tenth of the comparisons of a full rank sort.
// A cell with more than kWCap candidates, more than 64 lists or more than kWRows rows is left to the workgroup kernel below
// (cell_total = kCellDeferred + an entry in the deferred queue).  Same lists, same sort key, same output as there.
// ---------------------------------------------------------------------------------------------
constexpr int kCellDeferred = -1;
constexpr int kWCap = 512, kWRows = 128;
constexpr int kMaxListsPerCell = 64;
// per cell and list layout (dense strips / sparse tiles), two int4:
//   a = {xa | ya << 16, xb | yb << 16, id of the first list, ntc | min(nl, 255) << 8 | lists per row << 16 | level << 28}
//   b = {list offset in the frame's block (entries), scratch offset (uint32), lcap, bits of 1.0f / ntc}
struct CellGeoRec { int4 a, b; };
template <bool HARRIS>
__global__ __launch_bounds__(256) void k_cell_collect_w(Geom g, const CellGeoRec* __restrict__ geo, const uint2* __restrict__ lst_ent,
                                                         const int* __restrict__ lst_cnt, const uint8_t* __restrict__ pyr,
                                                         typename Ent<HARRIS>::T* __restrict__ cell_ent, int* __restrict__ cell_total,
                                                         int* __restrict__ defer_q) {
    using E = Ent<HARRIS>;
    __shared__ uint32_t s_keys[4][kWCap], s_tmp[4][kWCap];
    __shared__ int s_cnt[4][kWRows], s_rs[4][kWRows];
    __shared__ int s_offs[4][kMaxListsPerCell + 1];
    __shared__ unsigned s_lsts[4][kMaxListsPerCell];
    const int f = (int)blockIdx.x;
    if (f >= g.nframes) return;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    const int cell = (int)blockIdx.y * 4 + wave;
    const int ncells_frame = g.cell_base[g.nlevels];
    if (cell >= ncells_frame) return;
    uint32_t* keys = s_keys[wave];
    uint32_t* tmp = s_tmp[wave];
    int* cnt = s_cnt[wave];
    int* rs = s_rs[wave];
    int* s_off = s_offs[wave];
    unsigned* s_lst = s_lsts[wave];
    const int4 ga = geo[cell].a, gb = geo[cell].b;
    const int xa = ga.x & 0xffff, ya = (int)((unsigned)ga.x >> 16), xb = ga.y & 0xffff, yb = (int)((unsigned)ga.y >> 16);
    const int ntc = ga.w & 0xff, nl = (ga.w >> 8) & 0xff, ltx = (ga.w >> 16) & 0xfff, l = (int)((unsigned)ga.w >> 28);
    int* tot_out = cell_total + (size_t)f * ncells_frame + cell;
    auto defer = [&]() {
        if (lane == 0) {
            *tot_out = kCellDeferred;
            defer_q[1 + atomicAdd(&defer_q[0], 1)] = (f << 16) | cell;
        }
    };
    if (nl == 0) {
        if (lane == 0) *tot_out = 0;
        return;
    }
    const int ch = yb - ya;
    if (nl > kMaxListsPerCell || ch > kWRows) { defer(); return; }
    {   // the cell's lists: counts -> exclusive prefix
        int c = 0;
        if (lane < nl) {
            const int row = (int)(((float)lane + 0.5f) * __int_as_float(gb.w));   // lane / ntc (exact: see orb_configure)
            const unsigned id = (unsigned)(ga.z + row * ltx + (lane - row * ntc));
            s_lst[lane] = id;
            c = lst_cnt[(size_t)f * g.lst_base[g.nlevels] + id];
        }
        int incl = c;
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(incl, d);
            if (lane >= d) incl += v;
        }
        s_off[lane + 1] = incl;
        if (lane == 0) s_off[0] = 0;
        cnt[lane] = 0;
        cnt[lane + 64] = 0;
    }
    wave_fence();
    const int nent = s_off[nl];
    const uint2* ebase = lst_ent + (size_t)f * g.lst_base[g.nlevels] * g.lst_cap;
    int n_all = 0, n20 = 0;   // in-cell candidates (S > 7) / those above fast_th
    const int step0 = nl > 1 ? 1 << (31 - __clz(nl - 1)) : 0;   // first step of the binary search over the nl list starts
    bool odd = false;         // an entry with three in-cell maxima among its four pixels (impossible: they are never adjacent)
    for (int e0 = 0; e0 < nent; e0 += 64) {
        const int e = e0 + lane;
        uint2 en = make_uint2(0u, 0u);
        unsigned mask = 0;    // which of the entry's four pixels are in-cell candidates
        if (e < nent) {
            int k = 0;        // the list that holds entry e
            for (int step = step0; step > 0; step >>= 1)
                if (k + step < nl && s_off[k + step] <= e) k += step;
            en = ebase[(size_t)s_lst[k] * g.lst_cap + (e - s_off[k])];
            const int y = (int)(en.x >> 12), x0 = (int)(en.x & 0xfffu);
            // non-zero bytes of en.y -> bits 0..3; columns of the cell -> bits [xa - x0, xb - x0) of the four
            unsigned t = en.y | (en.y >> 4);
            t |= t >> 2;
            t |= t >> 1;
            const unsigned nz = ((t & 0x01010101u) * 0x01020408u) >> 24;
            const int qlo = min(max(xa - x0, 0), 4), qhi = min(max(xb - x0, 0), 4);
            const unsigned cols = ((1u << qhi) - 1u) & ~((1u << qlo) - 1u);
            mask = (y >= ya && y < yb) ? (nz & cols) : 0u;
        }
        // slots (the order inside keys[] does not matter: it is sorted below): an entry holds at most two candidates
        const unsigned m2 = mask & (mask - 1u);
        const unsigned long long b1 = __builtin_amdgcn_ballot_w64(mask != 0), b2 = __builtin_amdgcn_ballot_w64(m2 != 0);
        odd |= (m2 & (m2 - 1u)) != 0;
        const int slot = n_all + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b1, 0u)) +
                         (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b2, 0u));
        const uint32_t ybits = (en.x >> 12) << 20;
        const int x0 = (int)(en.x & 0xfffu);
        if (mask && slot < kWCap) {
            const int q = __ffs(mask) - 1;
            keys[slot] = ybits | ((uint32_t)(x0 + q) << 8) | ((en.y >> (8 * q)) & 0xffu);
        }
        if (m2 && slot + 1 < kWCap) {
            const int q = __ffs(m2) - 1;
            keys[slot + 1] = ybits | ((uint32_t)(x0 + q) << 8) | ((en.y >> (8 * q)) & 0xffu);
        }
        n_all += __popcll(b1) + __popcll(b2);
    }
    if (n_all > kWCap || __builtin_amdgcn_ballot_w64(odd) != 0) { defer(); return; }
    wave_fence();
    for (int i0 = 0; i0 < n_all; i0 += 64) {
        const int i = i0 + lane;
        n20 += __popcll(__builtin_amdgcn_ballot_w64(i < n_all && (int)(keys[i] & 0xffu) > g.fast_th));
    }
    // threshold choice of ORBextractor.cpp:616-623: FAST(20); if it yields <= 3 key points, FAST(7) - the corners at 7 are the
    // strict in-cell maxima of all scores above 7, those at 20 the ones among them that score above 20
    const int lo = n20 > 3 ? g.fast_th : 7;
    const int n = n20 > 3 ? n20 : n_all;
    const int lcap = gb.z;
    if (n > lcap) { defer(); return; }   // (cannot happen: in-cell maxima are never adjacent)
    wave_fence();
    // counting sort by row: slot inside the row (arrival order), row starts, placement, then the rank inside the row by x
    int slotv[kWCap / 64];
#pragma unroll
    for (int u = 0; u < kWCap / 64; ++u) {
        const int i = lane + 64 * u;
        slotv[u] = -1;
        if (64 * u >= n_all) continue;   // (uniform)
        if (i < n_all) {
            const uint32_t key = keys[i];
            if ((int)(key & 0xffu) > lo) slotv[u] = atomicAdd(&cnt[(int)(key >> 20) - ya], 1);
        }
    }
    wave_fence();
    {
        const int c0 = cnt[lane], c1 = cnt[lane + 64];
        int i0 = c0, i1 = c1;
        for (int d = 1; d < 64; d <<= 1) {
            const int v0 = __shfl_up(i0, d), v1 = __shfl_up(i1, d);
            if (lane >= d) { i0 += v0; i1 += v1; }
        }
        const int t0 = __shfl(i0, 63);
        rs[lane] = i0 - c0;
        rs[lane + 64] = t0 + i1 - c1;
    }
    wave_fence();
#pragma unroll
    for (int u = 0; u < kWCap / 64; ++u) {
        const int i = lane + 64 * u;
        if (64 * u >= n_all) continue;   // (uniform)
        if (i < n_all && slotv[u] >= 0) {
            const uint32_t key = keys[i];
            tmp[rs[(int)(key >> 20) - ya] + slotv[u]] = key;
        }
    }
    wave_fence();
    typename E::T* out = cell_ent + (size_t)f * g.lcell_off[g.nlevels] + (unsigned)gb.x;
    const uint8_t* lvl = pyr + pix(g, f, l, 0, 0);
    const int stride = g.stride[l];
    for (int p = lane; p < n; p += 64) {
        const uint32_t key = tmp[p];
        const int r = (int)(key >> 20) - ya;
        const int b0 = rs[r], b1 = b0 + cnt[r];
        int rank = b0;
        for (int j = b0; j < b1; ++j) rank += tmp[j] < key;
        const uint32_t pos = key >> 8;
        if constexpr (HARRIS) out[rank] = E::make(harris_response(lvl, stride, (int)(pos & 0xfff), (int)(pos >> 12)), pos);
        else out[rank] = (key << 24) | pos;
    }
    if (lane == 0) *tot_out = n;
}

// ---------------------------------------------------------------------------------------------
// cell_collect_wg / k_cell_collect_big: the same for the cells k_cell_collect_w left (huge cells, cells full of corners):
// one WORKGROUP per cell.
// Cell (i, j) of level l scans x in [16 + j*cellW, min(16 + (j+1)*cellW, w-16)), same for y: the cells tile the
// scan area exactly (cell window = cell +- 3 px, cv::FAST skips a 3 px rim, ORBextractor.cpp:569-608).
// sort key = (y << 12 | x) << 8 | S: ascending = row-major, the order of cv::FAST's output vector
// ---------------------------------------------------------------------------------------------
constexpr int kRankMax = 768;      // up to here a rank sort (no barriers) beats the bitonic network
template <bool HARRIS>
__device__ void cell_collect_wg(const Geom& g, const int f, const int cell, const uint2* __restrict__ lst_ent,
                                const int* __restrict__ lst_cnt, const uint8_t* __restrict__ pyr,
                                typename Ent<HARRIS>::T* __restrict__ cell_ent, uint32_t* __restrict__ cell_scr,
                                int* __restrict__ cell_total, int* __restrict__ overflow) {
    using E = Ent<HARRIS>;
    __shared__ uint32_t keys[kSortCap];
    __shared__ int s_nw, s_n20;   // candidates with 7 < S <= fast_th (stored from the back of keys[]) / with S > fast_th (front)
    __shared__ int s_off[kMaxListsPerCell + 1];                    // exclusive prefix of the entry counts of the cell's lists
    __shared__ unsigned s_lst[kMaxListsPerCell];
    int l = 0;
    while (l + 1 < g.nlevels && cell >= g.cell_base[l + 1]) ++l;
    const int ci = (cell - g.cell_base[l]) / g.gcols[l], cj = (cell - g.cell_base[l]) % g.gcols[l];
    const int xa = kEdge + cj * g.cellW[l], ya = kEdge + ci * g.cellH[l];
    const int xb = (cj == g.gcols[l] - 1) ? g.w[l] - kEdge : xa + g.cellW[l];
    const int yb = (ci == g.grows[l] - 1) ? g.h[l] - kEdge : ya + g.cellH[l];
    if (threadIdx.x == 0) { s_nw = 0; s_n20 = 0; }
    const int cw = xb - xa, ch = yb - ya;
    const int stride = g.stride[l];
    const int lcap = g.lcap[l];
    typename E::T* out = cell_ent + (size_t)f * g.lcell_off[g.nlevels] + g.lcell_off[l] + (size_t)(cell - g.cell_base[l]) * lcap;
    // the candidate lists whose tiles meet the cell: columns tc0..tc1 x rows tr0..tr1 of the level's list grid
    const int ltx = (g.sx1[l] - kEdge + g.lst_tw - 1) / g.lst_tw;
    const int tc0 = (xa - kEdge) / g.lst_tw, tc1 = (xb - 1 - kEdge) / g.lst_tw;
    const int tr0 = (ya - kEdge) / g.lst_th, tr1 = (yb - 1 - kEdge) / g.lst_th;
    const int ntc = tc1 - tc0 + 1;
    const int nl_all = (cw > 0 && ch > 0) ? ntc * (tr1 - tr0 + 1) : 0;
    const uint2* ebase = lst_ent + (size_t)f * g.lst_base[g.nlevels] * g.lst_cap;
    // every candidate of the cell, kMaxListsPerCell lists at a time (one round for ordinary cells; a 600 x 300 cell of a
    // 150-feature extractor takes two).  All threads call it together (it contains barriers).
    auto walk = [&](auto&& visit) {
        for (int k0 = 0; k0 < nl_all; k0 += kMaxListsPerCell) {
            const int nl = min(nl_all - k0, kMaxListsPerCell);
            __syncthreads();   // (the table of the previous round has been consumed; the counters are initialised)
            if ((int)threadIdx.x < 64) {   // wave 0: counts -> exclusive prefix
                const int k = threadIdx.x;
                int c = 0;
                if (k < nl) {
                    const int kk = k0 + k;
                    const unsigned id = (unsigned)(g.lst_base[l] + (tr0 + kk / ntc) * ltx + tc0 + kk % ntc);
                    s_lst[k] = id;
                    c = lst_cnt[(size_t)f * g.lst_base[g.nlevels] + id];
                }
                int incl = c;
                for (int d = 1; d < 64; d <<= 1) {
                    const int v = __shfl_up(incl, d);
                    if (k >= d) incl += v;
                }
                s_off[k + 1] = incl;
                if (k == 0) s_off[0] = 0;
            }
            __syncthreads();
            const int total = s_off[nl];
            for (int e = threadIdx.x; e < total; e += 256) {
                int k = 0;
                while (s_off[k + 1] <= e) ++k;
                const uint2 en = ebase[(size_t)s_lst[k] * g.lst_cap + (e - s_off[k])];
                const int y = (int)(en.x >> 12), x0 = (int)(en.x & 0xfffu);
                if (y < ya || y >= yb) continue;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int sc = (int)((en.y >> (8 * q)) & 0xffu);
                    const int x = x0 + q;
                    if (sc == 0 || x < xa || x >= xb) continue;
                    visit(sc, ((((uint32_t)y << 12) | (uint32_t)x) << 8) | (uint32_t)sc);
                }
            }
        }
    };
    walk([&](int sc, uint32_t key) {
        if (sc > g.fast_th) {
            const int slot = atomicAdd(&s_n20, 1);
            if (slot < kSortCap) keys[slot] = key;
        } else {
            const int slot = atomicAdd(&s_nw, 1);
            if (slot < kSortCap) keys[kSortCap - 1 - slot] = key;
        }
    });
    __syncthreads();
    // threshold choice of ORBextractor.cpp:616-623: FAST(20); if it yields <= 3 keypoints, FAST(7) - the corners at 7 are
    // the strict in-cell maxima of all scores above 7, those at 20 the ones among them that score above 20
    const int n20 = s_n20, nw = s_nw;
    const int total = n20 > 3 ? n20 : n20 + nw;
    const uint8_t* lvl = pyr + pix(g, f, l, 0, 0);
    auto emit = [&](int i, uint32_t key) {   // sorted place i of the cell's list
        const uint32_t pos = key >> 8;
        if constexpr (HARRIS) out[i] = E::make(harris_response(lvl, stride, (int)(pos & 0xfff), (int)(pos >> 12)), pos);
        else out[i] = (key << 24) | pos;
    };
    if (threadIdx.x == 0) cell_total[(size_t)f * g.cell_base[g.nlevels] + cell] = total <= lcap ? total : 0;
    if (total > lcap) {   // (in-cell maxima are never adjacent, so a cell holds at most ceil(w/2) ceil(h/2) of them = lcap)
        if (threadIdx.x == 0) atomicOr(overflow, 1);
        return;
    }
    if (n20 + nw > kSortCap) {
        // a huge cell full of corners (front and back ran into each other): collect what FAST returns once more, straight into
        // the cell's scratch list in global memory, and sort it there.  lcap is a power of two on such levels.
        uint32_t* scr = cell_scr + (size_t)f * g.lscr_off[g.nlevels] + g.lscr_off[l] + (size_t)(cell - g.cell_base[l]) * 2 * lcap;
        const int lo = n20 > 3 ? g.fast_th : 7;
        __syncthreads();
        if (threadIdx.x == 0) s_n20 = 0;
        walk([&](int sc, uint32_t key) {
            if (sc > lo) scr[atomicAdd(&s_n20, 1)] = key;
        });
        int npad = 1;
        while (npad < total) npad <<= 1;
        for (int i = total + threadIdx.x; i < npad; i += 256) scr[i] = 0xffffffffu;
        __syncthreads();
        bitonic_sort_u32(scr, npad);
        for (int i = threadIdx.x; i < total; i += 256) emit(i, scr[i]);
        return;
    }
    int n = n20;
    if (n20 <= 3) {
        uint32_t mv[kSortCap / 256];   // source and destination ranges may overlap: through registers
#pragma unroll
        for (int u = 0; u < kSortCap / 256; ++u) {
            const int i = threadIdx.x + 256 * u;
            mv[u] = i < nw ? keys[kSortCap - 1 - i] : 0u;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kSortCap / 256; ++u) {
            const int i = threadIdx.x + 256 * u;
            if (i < nw) keys[n20 + i] = mv[u];
        }
        __syncthreads();
        n = n20 + nw;
    }
    if (n <= kRankMax) {
        // a cell of ordinary size (~100 corners): rank sort.  A key's rank = the number of smaller keys (keys are unique: they
        // contain the position); every thread reads the same keys[j] (LDS broadcast), no barriers
        for (int i = threadIdx.x; i < n; i += 256) {
            const uint32_t ki = keys[i];
            int rank = 0;
            for (int j = 0; j < n; ++j) rank += keys[j] < ki;
            emit(rank, ki);
        }
        return;
    }
    int npad = 1;
    while (npad < n) npad <<= 1;
    for (int i = n + threadIdx.x; i < npad; i += 256) keys[i] = 0xffffffffu;
    __syncthreads();
    bitonic_sort_u32(keys, npad);
    for (int i = threadIdx.x; i < n; i += 256) emit(i, keys[i]);
}

// the deferred cells of a batch (defer_q = {count, (frame << 16 | cell) ...}, filled by k_cell_collect_w), a fixed small grid
template <bool HARRIS>
__global__ __launch_bounds__(256) void k_cell_collect_big(Geom g, const int* __restrict__ defer_q, const uint2* __restrict__ lst_ent,
                                                           const int* __restrict__ lst_cnt, const uint8_t* __restrict__ pyr,
                                                           typename Ent<HARRIS>::T* __restrict__ cell_ent, uint32_t* __restrict__ cell_scr,
                                                           int* __restrict__ cell_total, int* __restrict__ overflow) {
    const int nq = defer_q[0];
    for (int it = (int)blockIdx.x; it < nq; it += (int)gridDim.x) {
        const int v = defer_q[1 + it];
        cell_collect_wg<HARRIS>(g, v >> 16, v & 0xffff, lst_ent, lst_cnt, pyr, cell_ent, cell_scr, cell_total, overflow);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Quota replay of one level by one wave: lane c + 64 k owns cell c + 64 k (<= kQuotaCells = 1024 cells).  The reference's loop
// (:640-679) is a fixed-point over passes whose body does not depend on the cell order, so a pass is one wave step.
// n_ret / c_off receive the per-cell retain counts and their exclusive prefix; returns their sum.
// ---------------------------------------------------------------------------------------------
// skip / gcols: cells of the last column (bit 0) / row (bit 1) that the reference's loops never visit (Geom::skip) keep
// nTotal = nToRetain = 0 and bNoMore = false - the redistribution passes close them, not the first pass.
constexpr int kQuotaPerLane = 16, kQuotaCells = 64 * kQuotaPerLane;
__device__ __forceinline__ int level_quota(const int* tot, int nCells, int nfc, int* n_ret, int* c_off, int skip, int gcols) {
    const int lane = threadIdx.x & 63;
    int t[kQuotaPerLane], r[kQuotaPerLane];
    bool nm[kQuotaPerLane];
    int dist = 0, cnt = 0;
#pragma unroll
    for (int k = 0; k < kQuotaPerLane; ++k) {
        const int c = lane + 64 * k;
        t[k] = c < nCells ? tot[c] : 0;
        r[k] = 0;
        nm[k] = true;
        if (c < nCells) {
            const bool skipped = skip && (((skip & 1) && c % gcols == gcols - 1) || ((skip & 2) && c >= nCells - gcols));
            if (skipped) { t[k] = 0; nm[k] = false; }
            else if (t[k] > nfc) { r[k] = nfc; nm[k] = false; }
            else { r[k] = t[k]; dist += nfc - t[k]; ++cnt; }
        }
    }
    for (int s = 1; s < 64; s <<= 1) { dist += __shfl_xor(dist, s); cnt += __shfl_xor(cnt, s); }
    while (dist > 0 && cnt < nCells) {
        const int nNew = nfc + (int)ceilf((float)dist / (float)(nCells - cnt));
        int d2 = 0, c2 = 0;
#pragma unroll
        for (int k = 0; k < kQuotaPerLane; ++k)
            if (!nm[k]) {
                if (t[k] > nNew) { r[k] = nNew; }
                else { r[k] = t[k]; d2 += nNew - t[k]; nm[k] = true; ++c2; }
            }
        for (int s = 1; s < 64; s <<= 1) { d2 += __shfl_xor(d2, s); c2 += __shfl_xor(c2, s); }
        dist = d2;
        cnt += c2;
    }
    int base = 0;
#pragma unroll
    for (int k = 0; k < kQuotaPerLane; ++k) {
        if (64 * k >= nCells) break;   // (uniform: most levels have fewer than 64 cells)
        const int c = lane + 64 * k;
        int incl = r[k];
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(incl, d);
            if (lane >= d) incl += v;
        }
        if (c < nCells) { n_ret[c] = r[k]; c_off[c] = base + incl - r[k]; }
        base += __shfl(incl, 63);
    }
    return base;
}

// ---------------------------------------------------------------------------------------------
// k_quota: nToRetain of every cell (ORBextractor.cpp:631-679), one wave per (frame, level); cell_plan[cell] = {nToRetain,
// place of the cell's first retained corner in the level's list}
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_quota(Geom g, const int* __restrict__ cell_total, int2* __restrict__ cell_plan,
                                              int* __restrict__ defer_q) {
    __shared__ int n_ret[kQuotaCells], c_off[kQuotaCells];
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) defer_q[0] = 0;   // (k_cell_collect_big is done: next batch)
    SE2_FRAME_GRID(f, l);
    const int ncells_frame = g.cell_base[g.nlevels];
    const int nc = g.gcols[l] * g.grows[l];
    level_quota(cell_total + (size_t)f * ncells_frame + g.cell_base[l], nc, g.nfc[l], n_ret, c_off, g.skip[l], g.gcols[l]);
    wave_fence();
    int2* out = cell_plan + (size_t)f * ncells_frame + g.cell_base[l];
    for (int c = threadIdx.x; c < nc; c += 64) out[c] = make_int2(n_ret[c], c_off[c]);
}

// ---------------------------------------------------------------------------------------------
// k_cell_retain: KeyPointsFilter::retainBest(keysCell, nToRetain) + resize (ORBextractor.cpp:692-694), one wave per cell.
// A cell that keeps everything (most cells of sparse imagery) is done - its list is already in the reference's order.  Otherwise introselect over the list: in LDS up to
// kSelCap corners, in place in global memory (with the cell's scratch lists) beyond.
// ---------------------------------------------------------------------------------------------
constexpr int kSelCap = 512;      // corners a cell can hold for the LDS path of k_cell_retain (more: in place in global memory)
template <bool HARRIS>
__global__ __launch_bounds__(256) void k_cell_retain(Geom g, const CellGeoRec* __restrict__ geo, typename Ent<HARRIS>::T* __restrict__ cell_ent,
                                                      uint32_t* __restrict__ cell_scr, const int* __restrict__ cell_total,
                                                      const int2* __restrict__ cell_plan, int* __restrict__ overflow) {
    using E = Ent<HARRIS>;
    using T = typename E::T;
    __shared__ T s_a[4][kSelCap];
    __shared__ uint16_t s_lp[4][kSelCap], s_rp[4][kSelCap];
    const int f = (int)blockIdx.x;
    if (f >= g.nframes) return;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    const int cell = (int)blockIdx.y * 4 + wave;
    const int ncells_frame = g.cell_base[g.nlevels];
    if (cell >= ncells_frame) return;
    const int total = cell_total[(size_t)f * ncells_frame + cell];
    const int nret = cell_plan[(size_t)f * ncells_frame + cell].x;
    if (total <= nret) return;
    const int4 gb = geo[cell].b;
    const int lcap = gb.z;
    T* list = cell_ent + (size_t)f * g.lcell_off[g.nlevels] + (unsigned)gb.x;
    if (total <= kSelCap) {
        T* a = s_a[wave];
        for (int i = lane; i < total; i += 64) a[i] = list[i];
        wave_fence();
        introselect_wave<HARRIS, uint16_t>(a, s_lp[wave], s_rp[wave], total, nret);
        for (int i = lane; i < nret; i += 64) list[i] = a[i];
    } else {
        if (lcap <= kSelCap) {   // (cannot happen: total <= lcap)
            if (lane == 0) atomicOr(overflow, 2);
            return;
        }
        uint32_t* scr = cell_scr + (size_t)f * g.lscr_off[g.nlevels] + (unsigned)gb.y;
        introselect_wave<HARRIS, uint32_t>(list, scr, scr + lcap, total, nret);
    }
}

// ---------------------------------------------------------------------------------------------
// per-(frame, level) selection: the cells' retained corners in cell row-major order (ORBextractor.cpp:687-705), the
// level-wide retainBest + resize (:706-710).  Output: kp_list[f][i] = {level, x, y, response} in the reference's order;
// counts[f].  A level's place in the output is the number of key points the lower levels keep, which follows from k_quota's
// plan alone - no workgroup waits for another.
// ---------------------------------------------------------------------------------------------
template <bool HARRIS>
__global__ __launch_bounds__(256) void k_level_select(Geom g, const typename Ent<HARRIS>::T* __restrict__ cell_ent,
                                                       const int2* __restrict__ cell_plan, int4* __restrict__ kp_list,
                                                       int* __restrict__ counts, int cap, int* __restrict__ overflow) {
    using E = Ent<HARRIS>;
    using T = typename E::T;
    // dynamic LDS: the level's list and the two position lists of its introselect, g.level_cap entries each (the largest
    // quota + 2 cells of any level bounds what the cells can hand over; a few hundred entries with the usual parameters)
    extern __shared__ __attribute__((aligned(16))) unsigned char level_lds[];
    const int kLevelCap = g.level_cap;
    T* lst = reinterpret_cast<T*>(level_lds);
    uint16_t* s_lp = reinterpret_cast<uint16_t*>(lst + kLevelCap);
    uint16_t* s_rp = s_lp + kLevelCap;
    __shared__ int s_kept[kMaxLevels];
    SE2_FRAME_GRID(f, l);
    const int ncells_frame = g.cell_base[g.nlevels];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int2* plan_f = cell_plan + (size_t)f * ncells_frame;
    if (threadIdx.x <= (unsigned)l) {                      // entries of a level before its level-wide retain: last cell's place + count
        const bool cells = g.cell_base[threadIdx.x + 1] > g.cell_base[threadIdx.x];   // (a level can have none: orb_configure)
        const int2 last = cells ? plan_f[g.cell_base[threadIdx.x + 1] - 1] : make_int2(0, 0);
        int o = last.x + last.y;
        if (o > kLevelCap) { atomicOr(overflow, 4); o = kLevelCap; }
        s_kept[threadIdx.x] = o;
    }
    __syncthreads();
    const int nCells = g.gcols[l] * g.grows[l];
    const int n = s_kept[l];
    int out_n = 0;                                         // key points of the lower levels
    for (int lv = 0; lv < l; ++lv) out_n += min(s_kept[lv], g.quota[lv]);
    // the cells' lists one after the other: a wave takes every fourth cell
    const T* lbase = cell_ent + (size_t)f * g.lcell_off[g.nlevels] + g.lcell_off[l];
    for (int c = wave; c < nCells; c += 4) {
        const T* src = lbase + (size_t)c * g.lcap[l];
        const int2 pl = plan_f[g.cell_base[l] + c];
        for (int i = (int)(threadIdx.x & 63); i < pl.x; i += 64)
            if (pl.y + i < kLevelCap) lst[pl.y + i] = src[i];
    }
    __syncthreads();
    const int quota = g.quota[l];
    if (n > quota) {
        if (wave == 0) introselect_wave<HARRIS, uint16_t>(lst, s_lp, s_rp, n, quota);
        __syncthreads();
    }
    const int keep = min(n, quota);
    for (int i = threadIdx.x; i < keep; i += 256) {
        const int pos = out_n + i;
        if (pos < cap) {
            const T e = lst[i];
            const uint32_t p = E::pos(e);
            kp_list[(size_t)f * cap + pos] = make_int4(l, (int)(p & 0xfff), (int)(p >> 12), __float_as_int(E::resp(e)));
        } else {
            atomicOr(overflow, 8);
        }
    }
    if (l == g.nlevels - 1 && threadIdx.x == 0) counts[f] = min(out_n + keep, cap);
}

// se2gpu_orb_debug_nth_element: introselect_wave on caller data (tests): LDS path up to kSelCap entries, global path beyond
__global__ __launch_bounds__(64) void k_debug_nth(unsigned long long* a, uint32_t* scr, int n, int nth, int force_global) {
    __shared__ unsigned long long s_a[kSelCap];
    __shared__ uint16_t s_lp[kSelCap], s_rp[kSelCap];
    if (n <= kSelCap && !force_global) {
        for (int i = threadIdx.x; i < n; i += 64) s_a[i] = a[i];
        wave_fence();
        introselect_wave<true, uint16_t>(s_a, s_lp, s_rp, n, nth);
        for (int i = threadIdx.x; i < n; i += 64) a[i] = s_a[i];
    } else {
        introselect_wave<true, uint32_t>(a, scr, scr + n, n, nth);
    }
}

// ---------------------------------------------------------------------------------------------
// orientation: intensity centroid over the radius-15 disc (IC_Angle, ORBextractor.cpp:130-157), 16 lanes per keypoint
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {  // cv::fastAtan2
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

constexpr int kOriPerWave = 4;   // key points per wave of k_orientation
__global__ __launch_bounds__(256) void k_orientation(Geom g, const uint8_t* __restrict__ pyr,
                                                      const int4* __restrict__ kp_list, const int* __restrict__ counts,
                                                      int cap, float* __restrict__ angles, const int4* __restrict__ wtab) {
    // lane = (row offset v = 0..15, column octet c = 0..3), c fastest; a lane reads 8 bytes of row +v and of row -v with two
    // (unaligned) 32-bit loads each - the 749 disc pixels of a key point in 4 load instructions instead of 47 byte loads per
    // lane.  Integer moments: the summation order is irrelevant.
    // A wave takes FOUR key points (round 5): the kernel is bound by the latency of its dependent loads (record -> level
    // geometry -> pixels; SQ counters: a wave is parked 77 % of its life) at 32 waves per CU, so one key point per wave meant
    // 32 rounds of one latency chain each.  Four records and sixteen pixel loads per lane are in flight together, and the
    // eight moments of the four key points are summed over the lanes by a transposing butterfly (11 shuffles instead of 48).
    SE2_FRAME_GRID(f, bx);
    const int kbase = __builtin_amdgcn_readfirstlane((bx * 4 + (int)(threadIdx.x / 64)) * kOriPerWave);
    const int lane = threadIdx.x & 63;
    // (column octet fastest: the four lanes of a disc row are neighbours and their 8-byte pieces one 32-byte run - the texture
    // path then sees 16 rows per load instruction instead of 64 scattered dwords)
    const int v = lane >> 2, c = lane & 3;
    const int n = counts[f];
    if (kbase >= n) return;
    const int4 w = wtab[lane];
    uint32_t P0[kOriPerWave], P1[kOriPerWave], Q0[kOriPerWave], Q1[kOriPerWave];
#pragma unroll
    for (int i = 0; i < kOriPerWave; ++i) {
        P0[i] = P1[i] = Q0[i] = Q1[i] = 0;
        if (kbase + i < n) {   // (uniform; the records are scalar loads)
            const int4 kp = kp_list[(size_t)f * cap + kbase + i];
            const int stride = g.stride[kp.x];
            const uint8_t* center = pyr + pix(g, f, kp.x, kp.z, kp.y);
            const int u0 = -16 + 8 * c;
            const uint8_t* pp = center + u0 + v * stride;
            const uint8_t* pm = center + u0 - v * stride;
            P0[i] = *reinterpret_cast<const uint32_t*>(pp);
            P1[i] = *reinterpret_cast<const uint32_t*>(pp + 4);
            Q0[i] = *reinterpret_cast<const uint32_t*>(pm);
            Q1[i] = *reinterpret_cast<const uint32_t*>(pm + 4);
        }
    }
    int val[2 * kOriPerWave];   // {m10, m01} of the four key points: this lane's share
#pragma unroll
    for (int i = 0; i < kOriPerWave; ++i) {
        const uint32_t p0 = P0[i], p1 = P1[i];
        const uint32_t q0 = v == 0 ? 0u : Q0[i], q1 = v == 0 ? 0u : Q1[i];   // the centre row counts once
        // m10 = sum u (I(u, v) + I(u, -v)), m01 = v sum (I(u, v) - I(u, -v)) over the disc row: with the byte weights
        // {u + 16} and {1} of the lane's eight columns (zero outside the disc) these are eight v_dot4_u32_u8
        const uint32_t su = __builtin_amdgcn_udot4(p0, (uint32_t)w.x, __builtin_amdgcn_udot4(p1, (uint32_t)w.y,
                            __builtin_amdgcn_udot4(q0, (uint32_t)w.x, __builtin_amdgcn_udot4(q1, (uint32_t)w.y, 0u, false), false), false), false);
        const uint32_t sp = __builtin_amdgcn_udot4(p0, (uint32_t)w.z, __builtin_amdgcn_udot4(p1, (uint32_t)w.w, 0u, false), false);
        const uint32_t sm = __builtin_amdgcn_udot4(q0, (uint32_t)w.z, __builtin_amdgcn_udot4(q1, (uint32_t)w.w, 0u, false), false);
        val[2 * i] = (int)su - 16 * (int)(sp + sm);
        val[2 * i + 1] = v * ((int)sp - (int)sm);
    }
    // transposing butterfly: a step with lane mask M halves the values a lane carries - the lane keeps the half its bit of M
    // selects, sends the other half to its partner and adds what the partner sends.  After M = 32, 16, 8 a lane holds the sum
    // over 8 lanes of value number 4 b32 + 2 b16 + b8; three plain steps finish the sum over the remaining lanes.
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool hi = lane & 32;
        const int keep = hi ? val[j + 4] : val[j], send = hi ? val[j] : val[j + 4];
        val[j] = keep + __shfl_xor(send, 32);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const bool hi = lane & 16;
        const int keep = hi ? val[j + 2] : val[j], send = hi ? val[j] : val[j + 2];
        val[j] = keep + __shfl_xor(send, 16);
    }
    {
        const bool hi = lane & 8;
        const int keep = hi ? val[1] : val[0], send = hi ? val[0] : val[1];
        val[0] = keep + __shfl_xor(send, 8);
    }
    int tot = val[0];
    tot += __shfl_xor(tot, 4);
    tot += __shfl_xor(tot, 2);
    tot += __shfl_xor(tot, 1);
    // lanes with bit 8 clear hold m10 of key point i = 2 b32 + b16, their partners (bit 8 set) its m01
    const int other = __shfl_xor(tot, 8);
    const int i = ((lane >> 5) & 1) * 2 + ((lane >> 4) & 1);
    if ((lane & 15) == 0 && kbase + i < n) angles[(size_t)f * cap + kbase + i] = fast_atan2_deg((float)other, (float)tot);
}

// ---------------------------------------------------------------------------------------------
// 7x7 Gaussian, sigma 2, 8-bit fixed point: taps {18,34,49,55,49,34,18} per pass, (v + 2^15) >> 16, saturate.
// Tile = 64 x 16 outputs per workgroup; reads the un-blurred pyramid (whose frame holds reflect-101 copies).
// ---------------------------------------------------------------------------------------------
constexpr int kBlurRows = 36;
// One thread = 4 adjacent output pixels of a vertical strip of kBlurRows rows; the horizontal 7-tap sums of the last
// 7 rows stay in registers (sliding window), inputs come in as three aligned dwords per row.  No LDS.
// The blurred pyramid keeps the UN-blurred 16 px reflect frame of every level (the descriptor pattern of a key point
// 16 px from the edge reaches a few pixels into it).  The frame (about 14 % of a pyramid) is written by k_level0 /
// k_resize together with the pyramid itself: the 2 x 16 full frame rows, for every interior row its first 16-byte chunk
// and the chunks from the one that straddles the right edge of the interior onwards (the interior bytes that chunk
// carries are overwritten here).
__global__ __launch_bounds__(256) void k_blur(Geom g, const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur, int bx0) {
    SE2_FRAME_GRID_OFF(f, bx, bx0);
    int l = 0;
    while (l + 1 < g.nlevels && bx >= g.tile_base[l + 1]) ++l;
    const int t = bx - g.tile_base[l];
    const int W = g.w[l], H = g.h[l], stride = g.stride[l];
    const int tiles_x = (W + 255) / 256;
    const int x0 = (t % tiles_x) * 256 + (threadIdx.x & 63) * 4;
    const int y0 = __builtin_amdgcn_readfirstlane(((t / tiles_x) * 4 + (int)(threadIdx.x >> 6)) * kBlurRows);   // the wave's band: scalar rows
    if (x0 >= W || y0 >= H) return;
    const uint8_t* src = pyr + pix(g, f, l, 0, 0);
    uint8_t* dst = blur + pix(g, f, l, 0, 0);
    // Horizontal 7-tap sums of a row (they fit 16 bits: 255 * 257 = 65535): output pixel j needs the bytes j + 1 .. j + 7 of
    // the 12-byte window; v_alignbyte_b32 brings them to the front of two dwords and two v_dot4_u32_u8 with the taps
    // {18,34,49,55} and {49,34,18,0} add them up - 14 instructions per four pixels (27 as packed 16-bit multiply-adds).
    auto hrow_w = [&](uint32_t w0, uint32_t w1, uint32_t w2, int (&h)[4]) {
        constexpr uint32_t K0 = 18u | 34u << 8 | 49u << 16 | 55u << 24, K1 = 49u | 34u << 8 | 18u << 16;
        h[0] = (int)__builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 1), K1,
                                           __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 1), K0, 0u, false), false);
        h[1] = (int)__builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 2), K1,
                                           __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 2), K0, 0u, false), false);
        h[2] = (int)__builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 3), K1,
                                           __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 3), K0, 0u, false), false);
        h[3] = (int)__builtin_amdgcn_udot4(w2, K1, __builtin_amdgcn_udot4(w1, K0, 0u, false), false);
    };
    auto hrow = [&](int y, int (&h)[4]) {
        const uint32_t* p = (const uint32_t*)(src + (ptrdiff_t)y * stride + x0 - 4);
        hrow_w(p[0], p[1], p[2], h);
    };
    // The vertical pass takes the seven horizontal sums h[y-3 .. y+3] of a pixel two at a time: the window holds PAIRS of
    // consecutive rows, P[r] = h[r] | h[r+1] << 16 (a sum fits 16 bits), and
    //     18 h[y-3] + 34 h[y-2] | 49 h[y-1] + 55 h[y] | 49 h[y+1] + 34 h[y+2] | 18 h[y+3]
    // is three v_dot2_u32_u16 on P[y-3], P[y-1], P[y+1] and one multiply-add - with the v_lshl_or that forms the new pair five
    // instructions per pixel instead of seven (three adds of the symmetric rows + four multiply-adds; round 5).  The window of
    // six pairs P[y-3 .. y+2] is a circular buffer with static slots: the row loop is unrolled by 6 (kBlurRows is a multiple of 6).
    typedef unsigned short ushort2b __attribute__((ext_vector_type(2)));
    const ushort2b K01 = {18, 34}, K23 = {49, 55}, K45 = {49, 34};
    uint32_t pw[6][4];      // pw[(PH + i) % 6] = P[y - 3 + i] at the step of row y (i = 0..4), P[y + 2] is formed there
    int hprev[4];           // h[y + 2]
    {
        int h0[4], h1[4];
        hrow(y0 - 3, h0);
#pragma unroll
        for (int r = 0; r < 5; ++r) {   // pairs P[y0-3 .. y0+1] from the rows y0-3 .. y0+2
            hrow(y0 - 2 + r, h1);
#pragma unroll
            for (int q = 0; q < 4; ++q) { pw[r][q] = (uint32_t)h0[q] | ((uint32_t)h1[q] << 16); h0[q] = h1[q]; }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) hprev[q] = h0[q];
    }
    const int yend = min(y0 + kBlurRows, H);
    const int nvalid = min(4, W - x0);
    // the row entering the window is fetched two steps ahead, so that its latency hides behind the steps' arithmetic
    const int ylast = H + kEdge - 1;               // last row of the bordered plane
    uint32_t n0, n1, n2, m0, m1, m2;               // rows y + 3 and y + 4 of the coming step
    {
        const uint32_t* p = (const uint32_t*)(src + (ptrdiff_t)min(y0 + 3, ylast) * stride + x0 - 4);
        n0 = p[0]; n1 = p[1]; n2 = p[2];
        const uint32_t* q = (const uint32_t*)(src + (ptrdiff_t)min(y0 + 4, ylast) * stride + x0 - 4);
        m0 = q[0]; m1 = q[1]; m2 = q[2];
    }
    auto step = [&](auto phc, int y) {
        constexpr int PH = decltype(phc)::value;   // pair i of this iteration lives in slot (PH + i) % 6
        if (y >= yend) return;                     // uniform
        const uint32_t c0 = n0, c1 = n1, c2 = n2;
        n0 = m0; n1 = m1; n2 = m2;
        {
            const uint32_t* p = (const uint32_t*)(src + (ptrdiff_t)min(y + 5, ylast) * stride + x0 - 4);
            m0 = p[0]; m1 = p[1]; m2 = p[2];
        }
        int hn[4];                                 // h[y + 3]
        hrow_w(c0, c1, c2, hn);
        int sacc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            pw[(PH + 5) % 6][q] = (uint32_t)hprev[q] | ((uint32_t)hn[q] << 16);   // P[y + 2]
            hprev[q] = hn[q];
            uint32_t acc = 18u * (uint32_t)hn[q] + (1u << 15);
            acc = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2b, pw[PH % 6][q]), K01, acc, false);
            acc = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2b, pw[(PH + 2) % 6][q]), K23, acc, false);
            acc = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2b, pw[(PH + 4) % 6][q]), K45, acc, false);
            sacc[q] = (int)acc;
        }
        // (v + 2^15) >> 16, saturated to a byte, two pixels per instruction (gfx950's v_ashr_pk_u8_i32: the low half of the
        // result holds the two bytes, the upper half is NOT cleared - v_perm_b32 takes only the low halves).  Written with
        // the builtin on purpose: the plain form min(max(x >> 16, 0), 255) is matched to the same instruction by ROCm 7.2's
        // compiler, which then ORs the third pixel into the uncleared upper half (tools/dot_probe.hip reproduces that;
        // every third pixel of a group came out as q2 | q0).
        const uint32_t p01 = __builtin_amdgcn_ashr_pk_u8_i32(sacc[0], sacc[1], 16);
        const uint32_t p23 = __builtin_amdgcn_ashr_pk_u8_i32(sacc[2], sacc[3], 16);
        const uint32_t out = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
        uint32_t* o = (uint32_t*)(dst + (size_t)y * stride + x0);
        if (nvalid == 4) {
            *o = out;
        } else {  // last column group: keep the frame bytes (un-blurred reflect copies) to the right of the interior
            const uint32_t keep = ~((1u << (8 * nvalid)) - 1u);
            *o = (*o & keep) | (out & ~keep);
        }
    };
    for (int y = y0; y < yend; y += 6) {
        step(std::integral_constant<int, 0>{}, y);
        step(std::integral_constant<int, 1>{}, y + 1);
        step(std::integral_constant<int, 2>{}, y + 2);
        step(std::integral_constant<int, 3>{}, y + 3);
        step(std::integral_constant<int, 4>{}, y + 4);
        step(std::integral_constant<int, 5>{}, y + 5);
    }
}

// ---------------------------------------------------------------------------------------------
// descriptors + final key points.  One wave per keypoint; lane l evaluates bits l, l+64, l+128, l+192.
// ---------------------------------------------------------------------------------------------
// cos / sin of the key-point angle, in double and rounded once to float as computeOrbDescriptor does
// (ORBextractor.cpp:166-167).  One THREAD per key point: inside k_describe (one wave per key point) the double-precision
// sin and cos would be executed once per wave, i.e. 64 times more often.
// The reference's `(float)cos(angle)` / `(float)sin(angle)` take a FLOAT angle under `using namespace std`: the float overloads,
// i.e. libm's cosf / sinf - glibc's are one ulp off the rounded double value on 0.1 % of the arguments, and one descriptor in two
// million then samples a neighbouring pixel.  glibc's routine (flt-32, 2.28 and later) is plain double arithmetic: one multiply
// with 2/pi (scaled by 2^24, so that the quadrant is bits 24..31 of the truncated product), one multiply-subtract with pi/2, a
// degree-7 sine or degree-8 cosine polynomial, one rounding to float.  Written out here it gives the bits the reference gets on a
// glibc host (the CPU checker holds the same lines to libm on every float of [0, 2 pi]: tests/c_glibc_sincosf_check.c); contraction is off in
// this translation unit, which is one of the two forms checked.
__device__ __forceinline__ float glibc_sincosf_poly(double x, double x2, int n, double csign) {
    if ((n & 1) == 0) {
        const double x3 = x * x2, t1 = 0x1.1107605230bc4p-7 + x2 * -0x1.994eb3774cf24p-13, x7 = x3 * x2, s = x + x3 * -0x1.555545995a603p-3;
        return (float)(s + x7 * t1);
    }
    const double c0 = csign * 0x1p0, c1 = csign * -0x1.ffffffd0c621cp-2, c2 = csign * 0x1.55553e1068f19p-5, c3 = csign * -0x1.6c087e89a359dp-10,
                 c4 = csign * 0x1.99343027bf8c3p-16;
    const double x4 = x2 * x2, k2 = c3 + x2 * c4, k1 = c0 + x2 * c1, x6 = x4 * x2, c = k1 + x4 * c2;
    return (float)(c + x6 * k2);
}
__device__ __forceinline__ float glibc_sincosf(float y, int cosine) {   // 0 <= y < 120
    double x = y;
    const unsigned top = (__float_as_uint(y) >> 20) & 0x7ffu;
    if (top < ((__float_as_uint(0x1.921FB6p-1f) >> 20) & 0x7ffu)) {       // |y| < pi / 4
        if (top < ((__float_as_uint(0x1p-12f) >> 20) & 0x7ffu)) return cosine ? 1.0f : y;
        return glibc_sincosf_poly(x, x * x, cosine, 1.0);
    }
    const double r = x * 0x1.45F306DC9C883p+23;
    const int n = ((int)r + 0x800000) >> 24;
    x = x - n * 0x1.921FB54442D18p0;
    const double sgn = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    return glibc_sincosf_poly(x * sgn, x * x, n ^ cosine, (n & 2) ? -1.0 : 1.0);
}
__global__ __launch_bounds__(256) void k_angle_trig(const int* __restrict__ counts, int cap,
                                                     const float* __restrict__ angles, float2* __restrict__ cs) {
    const int f = blockIdx.y;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= min(counts[f], cap)) return;
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    const float angle = angles[(size_t)f * cap + k] * factorPI;
    cs[(size_t)f * cap + k] = make_float2(glibc_sincosf(angle, 1), glibc_sincosf(angle, 0));
}

__global__ __launch_bounds__(256) void k_describe(Geom g, const uint8_t* __restrict__ blur,
                                                   const int4* __restrict__ kp_list, const int* __restrict__ counts,
                                                   int cap, const float* __restrict__ angles,
                                                   const float2* __restrict__ cs,
                                                   se2gpu_keypoint* __restrict__ kps, uint8_t* __restrict__ desc) {
    // One wave handles TWO key points.  The 512 sample bytes of a key point lie within 18.4 px of it (the farthest pattern
    // point, whatever the rotation): the wave first copies the 39 x 40 byte patch around the key point from the blurred
    // level into LDS with dword loads (7 per lane instead of 8 scattered byte gathers, each of which costs the texture
    // path a pass per lane), then gathers the bytes there.  The patch region is private to the wave: no barrier.
    constexpr int kPR = 19, kPW = 40, kPH = 2 * kPR + 1;                 // rows -19 .. 19, columns -20 .. 19
    __shared__ uint32_t patch[8][kPH * (kPW / 4)];
    SE2_FRAME_GRID(f, bx);
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / 64));   // (wave-uniform, said so: key-point records by scalar loads)
    const int k0 = 2 * (bx * 4 + wv);
    const int lane = threadIdx.x & 63;
    const int n = counts[f];
    if (k0 >= n) return;  // wave-uniform
    const bool two = k0 + 1 < n;
    int4 kp[2];
    float ang[2];
    float2 ab[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int k = (u == 0 || two) ? k0 + u : k0;
        kp[u] = kp_list[(size_t)f * cap + k];
        ang[u] = angles[(size_t)f * cap + k];
        ab[u] = cs[(size_t)f * cap + k];
    }
    constexpr int kPD = kPH * (kPW / 4);                                   // 390 dwords per patch
    uint32_t pv[2][(kPD + 63) / 64];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int stride = g.stride[kp[u].x];
        const uint8_t* center = blur + pix(g, f, kp[u].x, kp[u].z, kp[u].y);
#pragma unroll
        for (int i = 0; i < (kPD + 63) / 64; ++i) {
            const int e = min(lane + 64 * i, kPD - 1);
            const int row = e / (kPW / 4), col = e - row * (kPW / 4);
            const uint8_t* p = center + (ptrdiff_t)(row - kPR) * stride + (4 * col - kPW / 2);
            pv[u][i] = *reinterpret_cast<const uint32_t*>(p);              // (unaligned) dword, as in k_orientation
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < (kPD + 63) / 64; ++i)
            if (lane + 64 * i < kPD) patch[2 * wv + u][lane + 64 * i] = pv[u][i];
    // The rotation of a test pair (x0, y0), (x1, y1) is two packed multiplies and a packed add / subtract per coordinate
    // (v_pk_mul_f32 / v_pk_add_f32: IEEE per lane, the products and the sum rounded separately as in the reference's scalar
    // x*b + y*a - contraction is off in this translation unit); the pattern bytes are unpacked once for both key points.
    typedef float float2p __attribute__((ext_vector_type(2)));
    int t0[2][4], t1[2][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int pw = reinterpret_cast<const int*>(c_pattern)[64 * q + lane];   // the 4 pattern bytes in one load
        const float2p X = {(float)(signed char)(pw & 0xff), (float)(signed char)((pw >> 16) & 0xff)};
        const float2p Y = {(float)(signed char)((pw >> 8) & 0xff), (float)(signed char)((pw >> 24) & 0xff)};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float2p A = {ab[u].x, ab[u].x}, Bv = {ab[u].y, ab[u].y};
            const uint8_t* pc = reinterpret_cast<const uint8_t*>(patch[2 * wv + u]) + kPR * kPW + kPW / 2;   // the key point
            const float2p R = X * Bv + Y * A, Cc = X * A - Y * Bv;
            const int r0 = (int)rintf(R.x), c0 = (int)rintf(Cc.x);
            const int r1 = (int)rintf(R.y), c1 = (int)rintf(Cc.y);
            t0[u][q] = pc[r0 * kPW + c0];
            t1[u][q] = pc[r1 * kPW + c1];
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (u == 1 && !two) break;
        const int k = k0 + u;
        unsigned long long words[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) words[q] = __ballot(t0[u][q] < t1[u][q]);
        if (lane < 4) {
            unsigned long long w = lane == 0 ? words[0] : (lane == 1 ? words[1] : (lane == 2 ? words[2] : words[3]));
            *(unsigned long long*)(desc + ((size_t)f * cap + k) * 32 + 8 * lane) = w;
        }
        if (lane == 0) {
            se2gpu_keypoint o;
            o.x = (float)kp[u].y;
            o.y = (float)kp[u].z;
            if (kp[u].x != 0) {
                const float sc = g.scale[kp[u].x];
                o.x *= sc;
                o.y *= sc;
            }
            o.size = g.patch[kp[u].x];
            o.angle = ang[u];
            o.response = __int_as_float(kp[u].w);
            o.octave = kp[u].x;
            o.class_id = -1;
            kps[(size_t)f * cap + k] = o;
        }
    }
}

inline int cv_round_f(float v) { return (int)std::nearbyintf(v); }
inline int cv_round_d(double v) { return (int)std::nearbyint(v); }
inline int cv_floor_f(float v) { int i = (int)v; return i - (i > v); }
inline int cv_ceil_d(double v) { int i = (int)v; return i + (i < v); }

}  // namespace

// ---------------------------------------------------------------------------------------------
// handle
// ---------------------------------------------------------------------------------------------
struct se2gpu_orb {
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipStream_t side_stream = nullptr;          // blurred pyramid: independent of the key-point chain, runs beside it
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipStream_t pyr_stream = nullptr;           // pipelined batches: the pyramid chain; score / blur of level l follow ev_lvl[l]
    hipEvent_t ev_lvl[kMaxLevels] = {};
    LaunchProfile prof;
    se2gpu_orb_params params{};
    double scaleFactor = 1.2;
    std::vector<float> mvScale, mvInvScale;
    std::vector<int> quota;
    int umax[16];
    Geom g{};
    int max_batch = 1;
    int last_batch = 0;
    DevBuf<uint8_t> pyr, blur, score, img;   // score: only se2gpu_orb_debug_score materialises the plane
    DevBuf<uint2> lst_ent;                   // candidate lists of the score kernel (see Geom::lst_*)
    DevBuf<int> lst_cnt;
    int dense_lst_base[kMaxLevels + 1], sparse_lst_base[kMaxLevels + 1];
    DevBuf<uint32_t> cell_keys;
    DevBuf<int> cell_total, counts, overflow;
    DevBuf<int2> cell_plan;                  // k_quota: {nToRetain, place in the level's list} per cell
    DevBuf<int4> cell_geo;                   // CellGeoRec per cell: [0, ncell) for the dense score kernel's lists, [ncell, 2 ncell) sparse
    DevBuf<int> defer_q;                     // cells k_cell_collect_w left to k_cell_collect_big: {count, entries ...}
    bool defer_dirty = false;                // a run was cut short between k_cell_collect_w and k_quota: defer_q[0] may not be zero
    DevBuf<int4> kp_list, tabs;
    DevBuf<float> angles;
    DevBuf<uint32_t> cell_scr;               // scratch of the huge-cell paths (Geom::lscr_off)
    DevBuf<float2> angle_cs;
    DevBuf<se2gpu_keypoint> kps;
    DevBuf<uint8_t> desc;
    PinBuf<uint8_t> stage_h;      // single-frame path: pinned image staging and the packed result block
    DevBuf<uint8_t> out_d;
    std::vector<size_t> xtab_off, ytab_off;  // offsets (in int4) into tabs, per level
    size_t orient_off = 0;                   // ... and of k_orientation's 64 weight entries
    int score_tiles = 0, blur_tiles = 0;
    Geom last_lists{};                       // geometry (with the list layout) of the last score launch
    int score_tile_base[kMaxLevels + 1], sparse_tile_base[kMaxLevels + 1], blur_tile_base[kMaxLevels + 1];
    // Which FAST kernel: SE2GPU_ORB_SCORE = dense | sparse | auto (default).  In auto mode the candidate kernel counts the
    // pixels that pass its compass test; it is used while fewer than kSparseBelow of the scanned pixels do, and every
    // kProbeEvery-th batch scored by the dense kernel goes through it instead to look again.  The count comes back
    // asynchronously (never waited for), so a decision lags one or two batches behind the imagery.
    int score_mode = 0;                  // 0 auto, 1 dense, 2 sparse
    bool use_sparse = false;             // auto: current choice
    bool fs_pending = false;             // a count download is in flight
    int fs_frames = 0, since_probe = 1 << 30;
    double fs_density = -1.0;            // last measured candidates / scanned pixel
    long long scan_pixels = 0;           // per frame, all levels
    DevBuf<int> fs_count;
    PinBuf<int> h_fs_count;
    hipEvent_t ev_fs = nullptr;
    ~se2gpu_orb() {
        if (own_stream) (void)hipStreamDestroy(own_stream);
        if (side_stream) (void)hipStreamDestroy(side_stream);
        if (pyr_stream) (void)hipStreamDestroy(pyr_stream);
        for (hipEvent_t e : ev_lvl)
            if (e) (void)hipEventDestroy(e);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (ev_fs) (void)hipEventDestroy(ev_fs);
    }
};

namespace {

// Geometry for a rows x cols image (ORBextractor.cpp:540-556, 794-795) + resize tables (cv::resize INTER_LINEAR).
int orb_configure(se2gpu_orb* h, int rows, int cols) {
    Geom& g = h->g;
    if (g.rows == rows && g.cols == cols && g.nlevels == h->params.nlevels) return SE2GPU_OK;
    const int L = h->params.nlevels;
    g.nlevels = L;
    g.rows = rows; g.cols = cols;
    g.nfeatures = h->params.nfeatures;
    g.fast_th = h->params.fast_th;
    g.harris = h->params.score_type == 0;
    for (int i = 0; i < 16; ++i) g.umax[i] = h->umax[i];
    unsigned off = 0;
    const float imageRatio = (float)cols / rows;
    g.cell_base[0] = 0;
    for (int l = 0; l < L; ++l) {
        const float scale = h->mvInvScale[l];
        g.w[l] = cv_round_f((float)cols * scale);
        g.h[l] = cv_round_f((float)rows * scale);
        SE2_REQUIRE(g.w[l] > 2 * kEdge + 6 && g.h[l] > 2 * kEdge + 6 && g.w[l] < 4096 && g.h[l] < 4096,
                    SE2GPU_ERR_INVALID, "pyramid level %d is %dx%d: unsupported image size", l, g.w[l], g.h[l]);
        g.stride[l] = ((g.w[l] + 2 * kEdge + 63) / 64) * 64;
        g.off[l] = off;
        off += (unsigned)g.stride[l] * (g.h[l] + 2 * kEdge);
        g.quota[l] = h->quota[l];
        g.gcols[l] = (int)std::sqrt((float)g.quota[l] / (5 * imageRatio));
        g.grows[l] = (int)(imageRatio * g.gcols[l]);
        g.scale[l] = h->mvScale[l];
        g.patch[l] = (float)(int)(kPatch * h->mvScale[l]);
        if (g.gcols[l] < 1 || g.grows[l] < 1) {
            // A level whose quota is too small for one cell (levelCols = sqrt(quota / (5 ratio)) = 0, or levelRows = ratio *
            // levelCols = 0 on a portrait image): the reference's loops over levelRows x levelCols then visit nothing and the
            // level contributes no key point (ORBextractor.cpp:541-716; e.g. the top level of 150 features over 8 levels).
            // Here: no cells, an empty scan area (so no score tiles and no candidate lists either).
            g.gcols[l] = g.grows[l] = 0;
            g.cellW[l] = g.cellH[l] = 1;
            g.nfc[l] = 0;
            g.cell_base[l + 1] = g.cell_base[l];
            g.sx1[l] = g.sy1[l] = kEdge;
            g.skip[l] = 0;
            continue;
        }
        SE2_REQUIRE(g.gcols[l] * g.grows[l] <= kQuotaCells, SE2GPU_ERR_INVALID, "level %d: unsupported cell grid %dx%d (more than 1024 cells)", l,
                    g.gcols[l], g.grows[l]);
        const int W = g.w[l] - 2 * kEdge, H = g.h[l] - 2 * kEdge;
        g.cellW[l] = (int)std::ceil((float)W / g.gcols[l]);
        g.cellH[l] = (int)std::ceil((float)H / g.grows[l]);
        g.nfc[l] = (int)std::ceil((float)g.quota[l] / (g.gcols[l] * g.grows[l]));
        g.cell_base[l + 1] = g.cell_base[l] + g.gcols[l] * g.grows[l];
        // Where the last cell column / row starts: inside the scan area (the usual case); at or up to 13 px beyond its end
        // (its own window is empty or not visited, the one before it scans into the reflect frame: Geom::sx1, skip); further
        // out the reference itself raises (cv::Mat::colRange / rowRange of a cell window outside the level, ORBextractor.cpp:608)
        const int ex = g.cellW[l] * (g.gcols[l] - 1), ey = g.cellH[l] * (g.grows[l] - 1);
        SE2_REQUIRE(ex <= W + 13 && ey <= H + 13, SE2GPU_ERR_INVALID,
                    "level %d: a cell window of the %dx%d grid lies outside the %dx%d level (the reference raises on this geometry)",
                    l, g.gcols[l], g.grows[l], g.w[l], g.h[l]);
        g.sx1[l] = kEdge + std::max(W, ex);
        g.sy1[l] = kEdge + std::max(H, ey);
        g.skip[l] = (ex >= W + 6 ? 1 : 0) | (ey >= H + 6 ? 2 : 0);
    }
    g.frame_bytes = (off + 255u) & ~255u;
    {   // k_level_select's list: every redistribution pass hands the cells at most quota + cells entries in total
        int lc = 64;
        for (int l = 0; l < L; ++l) lc = std::max(lc, g.quota[l] + 2 * g.gcols[l] * g.grows[l]);
        lc = (lc + 63) & ~63;
        // (64 bytes of the 64 KiB are k_level_select's static LDS: the launch at the exact limit would be refused by the runtime)
        SE2_REQUIRE((size_t)lc * ((g.harris ? 8 : 4) + 4) <= 64 * 1024 - 64 && lc < 65536, SE2GPU_ERR_INVALID,
                    "%d features on one pyramid level exceed the level list (at most %d)", lc, g.harris ? 5440 : 8128);
        g.level_cap = lc;
    }
    // per-cell list capacity: in-cell maxima are never adjacent, so ceil(cellW / 2) * ceil(cellH / 2) bounds a cell's corners
    g.lcell_off[0] = 0;
    g.lscr_off[0] = 0;
    for (int l = 0; l < L; ++l) {
        const long long bound = (long long)((g.cellW[l] + 1) / 2) * ((g.cellH[l] + 1) / 2);
        long long lc = ((bound + 63) / 64) * 64;
        if (bound > kSortCap) { lc = 1; while (lc < bound) lc <<= 1; }   // sorted in global memory by a bitonic network
        const long long nc = (long long)g.gcols[l] * g.grows[l];
        SE2_REQUIRE(lc <= (1 << 24) && g.lcell_off[l] + nc * lc < (1ll << 31) && g.lscr_off[l] + 2 * nc * lc < (1ll << 31), SE2GPU_ERR_INVALID,
                    "level %d: cells of %dx%d pixels are too large", l, g.cellW[l], g.cellH[l]);
        g.lcap[l] = (int)lc;
        g.lcell_off[l + 1] = g.lcell_off[l] + (unsigned)(nc * lc);
        g.lscr_off[l + 1] = g.lscr_off[l] + (lc > kSelCap ? (unsigned)(2 * nc * lc) : 0u);
    }
    // tiles
    h->score_tile_base[0] = 0;
    h->sparse_tile_base[0] = 0;
    h->blur_tile_base[0] = 0;
    for (int l = 0; l < L; ++l) {
        const int sw = g.sx1[l] - kEdge, sh = g.sy1[l] - kEdge;   // the scan area (Geom::sx1)
        h->score_tile_base[l + 1] = h->score_tile_base[l] + ((sw + 4 * kScoreGroups - 1) / (4 * kScoreGroups)) * ((sh + 4 * kScoreRows - 1) / (4 * kScoreRows));
        h->sparse_tile_base[l + 1] = h->sparse_tile_base[l] + ((sw + kFsTW - 1) / kFsTW) * ((sh + kFsTH - 1) / kFsTH);
        if (l == 0) h->scan_pixels = 0;
        h->scan_pixels += (long long)std::max(sw, 0) * std::max(sh, 0);
        h->blur_tile_base[l + 1] = h->blur_tile_base[l] + ((g.w[l] + 255) / 256) * ((g.h[l] + 4 * kBlurRows - 1) / (4 * kBlurRows));
        // candidate lists: one per strip of k_fast_score (62 groups x 19 rows) / per tile of k_fast_score_sparse
        if (l == 0) h->dense_lst_base[0] = h->sparse_lst_base[0] = 0;
        h->dense_lst_base[l + 1] = h->dense_lst_base[l] + ((sw + 4 * kScoreGroups - 1) / (4 * kScoreGroups)) * ((sh + kScoreRows - 1) / kScoreRows);
        h->sparse_lst_base[l + 1] = h->sparse_tile_base[l + 1];
    }
    // resize tables
    std::vector<int4> tabs, geo;   // (host sources of asynchronous uploads: alive until the synchronisation at the end)
    h->xtab_off.assign(L, 0);
    h->ytab_off.assign(L, 0);
    for (int l = 1; l < L; ++l) {
        const int sw = g.w[l - 1], sh = g.h[l - 1], dw = g.w[l], dh = g.h[l];
        const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
        const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
        // cv::resize switches to the S[sx]*ONE tail at the first dx whose sx+1 leaves the row and stays there; the
        // kernel applies the test per pixel, which is identical as long as sx is non-decreasing in dx (it is).
        std::vector<int4> yt(dh);
        auto clip = [](int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; };
        for (int dy = 0; dy < dh; dy++) {
            float fy = (float)((dy + 0.5) * scale_y - 0.5);
            int sy = cv_floor_f(fy);
            fy -= sy;
            const float cb0 = 1.f - fy, cb1 = fy;
            auto sat = [](int v) { return std::min(std::max(v, -32768), 32767); };
            yt[dy] = make_int4(clip(sy, 0, sh), clip(sy + 1, 0, sh), sat(cv_round_f(cb0 * 2048)), sat(cv_round_f(cb1 * 2048)));
        }
        h->ytab_off[l] = tabs.size();
        tabs.insert(tabs.end(), yt.begin(), yt.end());
        // x table over the bordered destination row (stride[l] columns, groups of 4)
        const int ng = g.stride[l] / 4;
        std::vector<int4> xt(2 * (size_t)ng, make_int4(0, 0, 0, 0));
        int last_sx = 0;   // padding columns repeat the last valid source column (bit 31 clear = "write 0"), so that
                           // the smallest column of a group of four is always one of its real taps
        for (int X = 0; X < g.stride[l]; ++X) {
            int w0 = last_sx, w1 = 0;
            if (X < dw + 2 * kEdge) {
                const int dx = reflect101(X - kEdge, dw);
                float fx = (float)((dx + 0.5) * scale_x - 0.5);
                int sx = cv_floor_f(fx);
                fx -= sx;
                if (sx < 0) { fx = 0; sx = 0; }
                int a0, a1;
                if (sx + 1 >= sw) {            // dx >= xmax: S[sx] * ONE (cv::resize's tail loop)
                    if (sx >= sw - 1) sx = sw - 1;
                    a0 = 2048; a1 = 0;
                } else {
                    a0 = cv_round_f((1.f - fx) * 2048.f);
                    a1 = cv_round_f(fx * 2048.f);
                }
                w0 = sx | (int)0x80000000u;
                w1 = a0 | (a1 << 16);
                last_sx = sx;
            }
            int* e0 = &xt[2 * (size_t)(X / 4)].x;
            int* e1 = &xt[2 * (size_t)(X / 4) + 1].x;
            e0[X & 3] = w0;
            e1[X & 3] = w1;
        }
        h->xtab_off[l] = tabs.size();
        tabs.insert(tabs.end(), xt.begin(), xt.end());
    }
    // k_orientation's weights: lane = (row offset v, column octet c) of the radius-15 disc; for its eight columns u the
    // bytes {u + 16 | 0} (x, y) and {1 | 0} (z, w), zero outside the disc row (|u| > umax[v])
    h->orient_off = tabs.size();
    for (int lane = 0; lane < 64; ++lane) {
        const int v = lane >> 2, c = lane & 3, d = h->umax[v], u0 = -16 + 8 * c;
        uint32_t w[4] = {0, 0, 0, 0};
        for (int i = 0; i < 8; ++i) {
            const int u = u0 + i;
            if (u < -d || u > d) continue;
            w[i >> 2] |= (uint32_t)(u + 16) << (8 * (i & 3));
            w[2 + (i >> 2)] |= 1u << (8 * (i & 3));
        }
        tabs.push_back(make_int4((int)w[0], (int)w[1], (int)w[2], (int)w[3]));
    }
    SE2_CHECK(h->tabs.upload(tabs, h->stream));
    {   // CellGeoRec tables (k_cell_collect_w, k_cell_retain): rectangle, candidate lists and list place of every cell
        const int ncell = g.cell_base[L];
        geo.assign(4 * (size_t)ncell, make_int4(0, 0, 0, 0));
        for (int layout = 0; layout < 2; ++layout) {
            const int tw = layout == 0 ? 4 * kScoreGroups : kFsTW, th = layout == 0 ? kScoreRows : kFsTH;
            const int* lbase = layout == 0 ? h->dense_lst_base : h->sparse_lst_base;
            for (int l = 0; l < L; ++l)
                for (int c = 0; c < g.gcols[l] * g.grows[l]; ++c) {
                    const int ci = c / g.gcols[l], cj = c % g.gcols[l];
                    const int xa = kEdge + cj * g.cellW[l], ya = kEdge + ci * g.cellH[l];
                    const int xb = (cj == g.gcols[l] - 1) ? g.w[l] - kEdge : xa + g.cellW[l];
                    const int yb = (ci == g.grows[l] - 1) ? g.h[l] - kEdge : ya + g.cellH[l];
                    const int ltx = (g.sx1[l] - kEdge + tw - 1) / tw;
                    const int tc0 = (xa - kEdge) / tw, tc1 = (xb - 1 - kEdge) / tw, tr0 = (ya - kEdge) / th, tr1 = (yb - 1 - kEdge) / th;
                    const int ntc = tc1 - tc0 + 1;
                    const int nl = (xb > xa && yb > ya) ? ntc * (tr1 - tr0 + 1) : 0;
                    // (lane + 0.5f) * (1.0f / ntc) truncates to lane / ntc for lane < 64: lane + 0.5 is never a multiple of ntc
                    // and stays 0.5 / ntc >= 1 / 128 away from one, the float error is ~1e-5
                    const float rcp = 1.0f / (float)std::max(ntc, 1);
                    int rb;
                    std::memcpy(&rb, &rcp, 4);
                    SE2_REQUIRE(ltx < 4096 && std::min(ntc, 255) == (ntc & 0xff), SE2GPU_ERR_INVALID, "level %d: cell / list geometry out of range", l);
                    int4 a = make_int4(xa | (ya << 16), xb | (yb << 16), lbase[l] + tr0 * ltx + tc0,
                                       (ntc & 0xff) | (std::min(nl, 255) << 8) | (ltx << 16) | (int)((unsigned)l << 28));
                    if (nl > 255 || ntc > 255) a.w = (a.w & ~0xffff) | 1 | (255 << 8);   // (deferred to the workgroup kernel, which derives its own lists)
                    const int4 b = make_int4((int)(g.lcell_off[l] + (unsigned)c * (unsigned)g.lcap[l]),
                                             g.lcap[l] > kSelCap ? (int)(g.lscr_off[l] + 2u * (unsigned)c * (unsigned)g.lcap[l]) : 0, g.lcap[l], rb);
                    const size_t at = 2 * ((size_t)layout * ncell + g.cell_base[l] + c);
                    geo[at] = a;
                    geo[at + 1] = b;
                }
        }
        if (geo.empty()) geo.resize(4);   // (no level has a cell)
        SE2_CHECK(h->cell_geo.upload(geo, h->stream));
    }
    // buffers
    const size_t B = (size_t)h->max_batch;
    SE2_CHECK(h->pyr.reserve(B * g.frame_bytes + 256));    // (+ slack: a patch copy of a key point in the reflect frame may read a few bytes past a row)
    SE2_CHECK(h->blur.reserve(B * g.frame_bytes + 256));
    SE2_CHECK(h->lst_cnt.reserve(std::max<size_t>(1, B * (size_t)std::max(h->dense_lst_base[L], h->sparse_lst_base[L]))));
    SE2_CHECK(h->lst_ent.reserve(std::max<size_t>(1, B * std::max((size_t)h->dense_lst_base[L] * kStripCap, (size_t)h->sparse_lst_base[L] * kFsListCap))));
    SE2_CHECK(h->cell_keys.reserve(std::max<size_t>(64, B * (size_t)g.lcell_off[L] * (g.harris ? 2 : 1))));   // uint32 entries, uint64 with HARRIS_SCORE
    SE2_CHECK(h->cell_scr.reserve(std::max<size_t>(1, B * (size_t)g.lscr_off[L])));
    SE2_CHECK(h->cell_total.reserve(std::max<size_t>(1, B * g.cell_base[L])));
    SE2_CHECK(h->cell_plan.reserve(std::max<size_t>(1, B * g.cell_base[L])));
    SE2_CHECK(h->defer_q.reserve(1 + B * g.cell_base[L]));
    SE2_HIP(hipMemsetAsync(h->defer_q.p, 0, sizeof(int), h->stream));
    SE2_CHECK(h->overflow.reserve(1));
    SE2_HIP(hipMemsetAsync(h->overflow.p, 0, sizeof(int), h->stream));
    SE2_HIP(hipStreamSynchronize(h->stream));
    return SE2GPU_OK;
}

// the device pipeline for `nframes` frames already in d_imgs (pitch = cols)
int orb_run(se2gpu_orb* h, const uint8_t* d_imgs, int pitch, int nframes, se2gpu_keypoint* d_kps, uint8_t* d_desc,
            int32_t* d_counts, int cap) {
    Geom g = h->g;
    g.nframes = nframes;
    hipStream_t st = h->stream;
    const int L = g.nlevels;
    const unsigned F8 = 8u * (unsigned)((nframes + 7) / 8);   // grid x: frames (see SE2_FRAME_GRID)
    SE2_CHECK(h->kp_list.reserve((size_t)h->max_batch * cap));
    SE2_CHECK(h->angles.reserve((size_t)h->max_batch * cap));
    SE2_CHECK(h->angle_cs.reserve((size_t)h->max_batch * cap));
    // Batches are pipelined by level: the pyramid chain (k_level0, k_resize x 7: latency-bound, half the VALU idle) runs on
    // its own stream and the score and blur launches of level l start as soon as level l exists, instead of after the
    // whole pyramid - the score and blur kernels are VALU-bound and fill the gaps of the resize chain.  A single frame
    // (the tracking thread's call) keeps the short serial sequence: there the extra launches would cost more than they hide.
    static const int pipe_min = [] { const char* e = std::getenv("SE2GPU_ORB_PIPELINE_MIN"); return e ? std::atoi(e) : 16; }();
    const bool piped = !h->prof.enabled && nframes >= pipe_min;
    if (piped && !h->pyr_stream) {
        SE2_HIP(hipStreamCreateWithFlags(&h->pyr_stream, hipStreamNonBlocking));
        for (int l = 0; l < kMaxLevels; ++l) SE2_HIP(hipEventCreateWithFlags(&h->ev_lvl[l], hipEventDisableTiming));
    }
    hipStream_t sp = piped ? h->pyr_stream : st;
    // blurred pyramid: beside the key-point chain, on the handle's side stream (none with SE2GPU_ORB_SIDE_STREAM=0)
    hipStream_t sb = (h->prof.enabled || !h->side_stream) ? st : h->side_stream;
    if (piped || sb != st) SE2_HIP(hipEventRecord(h->ev_fork, st));
    if (piped) SE2_HIP(hipStreamWaitEvent(sp, h->ev_fork, 0));
    if (sb != st) SE2_HIP(hipStreamWaitEvent(sb, h->ev_fork, 0));
    // ---- score kernel choice (dense / sparse by the measured candidate density)
    constexpr double kSparseBelow = 0.10;   // the two kernels cost the same at about one candidate per ten pixels
    constexpr int kProbeEvery = 32;
    bool run_sparse = h->score_mode == 2, count = false;
    if (h->score_mode == 0) {
        if (h->fs_pending && hipEventQuery(h->ev_fs) == hipSuccess) {
            long long tot = 0;
            for (int i = 0; i < h->fs_frames; ++i) tot += h->h_fs_count.p[i];
            h->fs_density = (double)tot / ((double)h->fs_frames * (double)std::max(h->scan_pixels, 1ll));
            h->use_sparse = h->fs_density < kSparseBelow;
            h->fs_pending = false;
        }
        const bool probe = !h->fs_pending && (h->use_sparse || h->since_probe >= kProbeEvery);
        run_sparse = h->use_sparse || probe;
        count = probe;
        h->since_probe = probe ? 0 : h->since_probe + 1;
    }
    if (run_sparse && count) {
        SE2_CHECK(h->fs_count.reserve((size_t)h->max_batch));
        SE2_CHECK(h->h_fs_count.reserve((size_t)h->max_batch));
        SE2_HIP(hipMemsetAsync(h->fs_count.p, 0, (size_t)nframes * sizeof(int), st));
    }
    Geom gb = g;   // blur tiles
    for (int l = 0; l <= L; ++l) gb.tile_base[l] = h->blur_tile_base[l];
    if (run_sparse) {
        for (int l = 0; l <= L; ++l) { g.tile_base[l] = h->sparse_tile_base[l]; g.lst_base[l] = h->sparse_lst_base[l]; }
        g.lst_tw = kFsTW; g.lst_th = kFsTH; g.lst_cap = kFsListCap;
    } else {
        for (int l = 0; l <= L; ++l) { g.tile_base[l] = h->score_tile_base[l]; g.lst_base[l] = h->dense_lst_base[l]; }
        g.lst_tw = 4 * kScoreGroups; g.lst_th = kScoreRows; g.lst_cap = kStripCap;
    }
    // score / blur of the levels [l0, l1)
    auto score_levels = [&](int l0, int l1) {
        const int b0 = g.tile_base[l0], nb = g.tile_base[l1] - b0;
        if (nb <= 0) return;
        if (run_sparse)
            SE2_LAUNCH(h->prof, st, "k_fast_score", k_fast_score_sparse, dim3(F8, nb), dim3(256), 0, g, h->pyr.p, h->lst_ent.p,
                       h->lst_cnt.p, count ? h->fs_count.p : (int*)nullptr, h->overflow.p, b0);
        else
            SE2_LAUNCH(h->prof, st, "k_fast_score", k_fast_score, dim3(F8, nb), dim3(256), 0, g, h->pyr.p, h->lst_ent.p,
                       h->lst_cnt.p, b0);
    };
    auto blur_levels = [&](int l0, int l1) {
        const int b0 = gb.tile_base[l0], nb = gb.tile_base[l1] - b0;
        if (nb > 0) SE2_LAUNCH(h->prof, sb, "k_blur", k_blur, dim3(F8, nb), dim3(256), 0, gb, h->pyr.p, h->blur.p, b0);
    };
    // ---- pyramid; in a pipelined batch level l's consumers follow ev_lvl[l].  The small upper levels go out together.
    const int lgroup = std::min(L, 3);   // levels >= lgroup are scored / blurred by one launch each, after the last resize
    for (int l = 0; l < L; ++l) {
        if (l == 0) {
            dim3 grid(F8, ((g.stride[0] / 16) * (g.h[0] + 2 * kEdge) + 255) / 256);
            SE2_LAUNCH(h->prof, sp, "k_level0", k_level0, grid, dim3(256), 0, g, d_imgs, pitch, h->pyr.p, h->blur.p);
        } else {
            const int ng = g.stride[l] / 4;
            ResizeTab t{h->tabs.p + h->ytab_off[l], h->tabs.p + h->xtab_off[l], ng};
            dim3 grid(F8, ((ng + 63) / 64) * ((g.h[l] + 2 * kEdge + 4 * kResizeRows - 1) / (4 * kResizeRows)));
            SE2_LAUNCH(h->prof, sp, "k_resize", k_resize, grid, dim3(256), 0, g, l, t, h->pyr.p, h->blur.p);
        }
        if (!piped) continue;
        if (l < lgroup || l == L - 1) {
            SE2_HIP(hipEventRecord(h->ev_lvl[l], sp));
            SE2_HIP(hipStreamWaitEvent(st, h->ev_lvl[l], 0));
            SE2_HIP(hipStreamWaitEvent(sb, h->ev_lvl[l], 0));
            if (l < lgroup) { score_levels(l, l + 1); blur_levels(l, l + 1); }
            else { score_levels(lgroup, L); blur_levels(lgroup, L); }
        }
    }
    if (!piped) {
        if (sb != st) {   // (the fork above was recorded before the pyramid: the blur stream must see the pyramid)
            SE2_HIP(hipEventRecord(h->ev_fork, st));
            SE2_HIP(hipStreamWaitEvent(sb, h->ev_fork, 0));
        }
        blur_levels(0, L);
        score_levels(0, L);
    }
    if (sb != st) SE2_HIP(hipEventRecord(h->ev_join, sb));
    if (run_sparse && count) {
        SE2_HIP(hipMemcpyAsync(h->h_fs_count.p, h->fs_count.p, (size_t)nframes * sizeof(int), hipMemcpyDeviceToHost, st));
        SE2_HIP(hipEventRecord(h->ev_fs, st));
        h->fs_pending = true;
        h->fs_frames = nframes;
    }
    h->last_lists = g;
    // Only k_quota puts the deferred-cell count back to zero.  Should a run ever leave between k_cell_collect_w and k_quota (a launch
    // error), the count would carry into the next batch and k_cell_collect_w would append beyond the queue: a run that
    // did not get past k_quota leaves the flag set, and the next one clears the word first.
    if (h->defer_dirty) SE2_HIP(hipMemsetAsync(h->defer_q.p, 0, sizeof(int), st));
    h->defer_dirty = true;
    const unsigned ncell = (unsigned)g.cell_base[L];
    const CellGeoRec* geo = reinterpret_cast<const CellGeoRec*>(h->cell_geo.p) + (run_sparse ? ncell : 0u);
    constexpr unsigned kBigGrid = 512;   // workgroups that share the deferred cells of a batch (normally none)
    if (g.harris) {
        using T = Ent<true>::T;
        T* ent = reinterpret_cast<T*>(h->cell_keys.p);
        if (ncell)
            SE2_LAUNCH(h->prof, st, "k_cell_collect", k_cell_collect_w<true>, dim3(F8, (ncell + 3) / 4), dim3(256), 0, g, geo, h->lst_ent.p,
                       h->lst_cnt.p, h->pyr.p, ent, h->cell_total.p, h->defer_q.p);
        SE2_LAUNCH(h->prof, st, "k_cell_collect_big", k_cell_collect_big<true>, dim3(kBigGrid), dim3(256), 0, g, h->defer_q.p,
                   h->lst_ent.p, h->lst_cnt.p, h->pyr.p, ent, h->cell_scr.p, h->cell_total.p, h->overflow.p);
        SE2_LAUNCH(h->prof, st, "k_quota", k_quota, dim3(F8, L), dim3(64), 0, g, h->cell_total.p, h->cell_plan.p, h->defer_q.p);
        h->defer_dirty = false;
        if (ncell)
            SE2_LAUNCH(h->prof, st, "k_cell_retain", k_cell_retain<true>, dim3(F8, (ncell + 3) / 4), dim3(256), 0, g, geo, ent,
                       h->cell_scr.p, h->cell_total.p, h->cell_plan.p, h->overflow.p);
        SE2_LAUNCH(h->prof, st, "k_level_select", k_level_select<true>, dim3(F8, L), dim3(256), (size_t)g.level_cap * 12, g, ent, h->cell_plan.p,
                   h->kp_list.p, d_counts, cap, h->overflow.p);
    } else {
        if (ncell)   // (no level may have a cell: a handful of features over many levels, orb_configure)
            SE2_LAUNCH(h->prof, st, "k_cell_collect", k_cell_collect_w<false>, dim3(F8, (ncell + 3) / 4), dim3(256), 0, g, geo, h->lst_ent.p,
                       h->lst_cnt.p, h->pyr.p, h->cell_keys.p, h->cell_total.p, h->defer_q.p);
        SE2_LAUNCH(h->prof, st, "k_cell_collect_big", k_cell_collect_big<false>, dim3(kBigGrid), dim3(256), 0, g, h->defer_q.p,
                   h->lst_ent.p, h->lst_cnt.p, h->pyr.p, h->cell_keys.p, h->cell_scr.p, h->cell_total.p, h->overflow.p);
        SE2_LAUNCH(h->prof, st, "k_quota", k_quota, dim3(F8, L), dim3(64), 0, g, h->cell_total.p, h->cell_plan.p, h->defer_q.p);
        h->defer_dirty = false;
        if (ncell)
            SE2_LAUNCH(h->prof, st, "k_cell_retain", k_cell_retain<false>, dim3(F8, (ncell + 3) / 4), dim3(256), 0, g, geo,
                       h->cell_keys.p, h->cell_scr.p, h->cell_total.p, h->cell_plan.p, h->overflow.p);
        SE2_LAUNCH(h->prof, st, "k_level_select", k_level_select<false>, dim3(F8, L), dim3(256), (size_t)g.level_cap * 8, g, h->cell_keys.p,
                   h->cell_plan.p, h->kp_list.p, d_counts, cap, h->overflow.p);
    }
    SE2_LAUNCH(h->prof, st, "k_orientation", k_orientation, dim3(F8, (cap + 4 * kOriPerWave - 1) / (4 * kOriPerWave)), dim3(256), 0, g, h->pyr.p,
               h->kp_list.p, d_counts, cap, h->angles.p, h->tabs.p + h->orient_off);
    SE2_LAUNCH(h->prof, st, "k_angle_trig", k_angle_trig, dim3((cap + 255) / 256, nframes), dim3(256), 0, d_counts, cap,
               h->angles.p, h->angle_cs.p);
    if (sb != st) SE2_HIP(hipStreamWaitEvent(st, h->ev_join, 0));
    SE2_LAUNCH(h->prof, st, "k_describe", k_describe, dim3(F8, (cap + 7) / 8), dim3(256), 0, g, h->blur.p,
               h->kp_list.p, d_counts, cap, h->angles.p, h->angle_cs.p, d_kps, d_desc);
    SE2_HIP(hipGetLastError());
    h->last_batch = nframes;
    return SE2GPU_OK;
}

int orb_check_overflow(se2gpu_orb* h) {
    int ov = 0;
    SE2_HIP(hipMemcpyAsync(&ov, h->overflow.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    SE2_HIP(hipStreamSynchronize(h->stream));
    if (ov) {
        SE2_HIP(hipMemsetAsync(h->overflow.p, 0, sizeof(int), h->stream));
        set_error("ORB extractor: internal capacity overflow (mask %d): cell candidates / level list / output cap", ov);
        return SE2GPU_ERR_CAPACITY;
    }
    return SE2GPU_OK;
}

}  // namespace

extern "C" {

int se2gpu_orb_create(const se2gpu_orb_params* params, se2gpu_orb** out) {
    SE2_REQUIRE(params && out, SE2GPU_ERR_INVALID, "orb_create: NULL argument");
    SE2_REQUIRE(have_device(), SE2GPU_ERR_NO_DEVICE, "no HIP device visible (libse2gpu has no CPU fallback)");
    SE2_REQUIRE(params->score_type == 1 || params->score_type == 0, SE2GPU_ERR_INVALID,
                "score_type must be 1 (ORB::FAST_SCORE) or 0 (ORB::HARRIS_SCORE)");
    SE2_REQUIRE(params->nlevels >= 1 && params->nlevels <= kMaxLevels && params->nfeatures > 0 &&
                    params->scale_factor > 1.0f && params->fast_th >= 7 && params->fast_th < 255,
                SE2GPU_ERR_INVALID, "orb_create: parameter out of range");
    SE2_REQUIRE(params->max_batch < 32768, SE2GPU_ERR_INVALID, "max_batch %d: at most 32767 frames per batch", params->max_batch);   // (before the handle exists: nothing to free on this path)
    se2gpu_orb* h = new se2gpu_orb;
    h->params = *params;
    if (h->params.max_rows <= 0 || h->params.max_cols <= 0) { h->params.max_rows = 480; h->params.max_cols = 640; }
    h->max_batch = std::max(1, params->max_batch);
    if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        set_error("hipStreamCreate failed");
        return SE2GPU_ERR_HIP;
    }
    h->stream = h->own_stream;
    if (const char* e = std::getenv("SE2GPU_ORB_SCORE"))
        h->score_mode = std::strcmp(e, "dense") == 0 ? 1 : (std::strcmp(e, "sparse") == 0 ? 2 : 0);
    // The side stream is created here, right after the handle's own stream, and not on first use: streams land on the
    // device's few hardware queues in creation order, and a side stream that ends up on the queue of its own handle's main
    // stream (or of another handle's) overlaps nothing - measured: 154k instead of 165k frames/s with two handles in flight.
    // SE2GPU_ORB_SIDE_STREAM=0: no side stream (callers that keep three or more batches in flight, one queue per handle).
    static const bool side_on = [] { const char* e = std::getenv("SE2GPU_ORB_SIDE_STREAM"); return !(e && e[0] == '0'); }();
    if ((side_on && hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking) != hipSuccess) ||
        hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_fs, hipEventDisableTiming) != hipSuccess) {
        set_error("hipStreamCreate / hipEventCreate failed");
        delete h;
        return SE2GPU_ERR_HIP;
    }
    // ORBextractor::ORBextractor (ORBextractor.cpp:463-520); scaleFactor is a double member built from a float
    const int L = params->nlevels;
    h->scaleFactor = (double)params->scale_factor;
    h->mvScale.resize(L); h->mvInvScale.resize(L); h->quota.resize(L);
    h->mvScale[0] = 1;
    for (int i = 1; i < L; i++) h->mvScale[i] = (float)(h->mvScale[i - 1] * h->scaleFactor);
    const float invScaleFactor = (float)(1.0f / h->scaleFactor);
    h->mvInvScale[0] = 1;
    for (int i = 1; i < L; i++) h->mvInvScale[i] = h->mvInvScale[i - 1] * invScaleFactor;
    const float factor = (float)(1.0 / h->scaleFactor);
    float nDesired = params->nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)L));
    int sum = 0;
    for (int level = 0; level < L - 1; level++) {
        h->quota[level] = cv_round_f(nDesired);
        sum += h->quota[level];
        nDesired *= factor;
    }
    h->quota[L - 1] = std::max(params->nfeatures - sum, 0);
    {
        int um[kHalfPatch + 2] = {0};
        int v, v0;
        const int vmax = cv_floor_f(kHalfPatch * std::sqrt(2.f) / 2 + 1);
        const int vmin = cv_ceil_d(kHalfPatch * std::sqrt(2.f) / 2);
        const double hp2 = kHalfPatch * kHalfPatch;
        for (v = 0; v <= vmax; ++v) um[v] = cv_round_d(std::sqrt(hp2 - v * v));
        for (v = kHalfPatch, v0 = 0; v >= vmin; --v) {
            while (um[v0] == um[v0 + 1]) ++v0;
            um[v] = v0;
            ++v0;
        }
        for (int i = 0; i < 16; ++i) h->umax[i] = um[i];
    }
    const int rc = orb_configure(h, h->params.max_rows, h->params.max_cols);
    if (rc != SE2GPU_OK) {
        delete h;
        return rc;
    }
    *out = h;
    return SE2GPU_OK;
}

void se2gpu_orb_destroy(se2gpu_orb* h) { delete h; }
int se2gpu_orb_levels(const se2gpu_orb* h) { return h ? h->params.nlevels : 0; }
float se2gpu_orb_scale_factor(const se2gpu_orb* h) { return h ? (float)h->scaleFactor : 0.f; }
void* se2gpu_orb_stream(se2gpu_orb* h) { return h ? (void*)h->stream : nullptr; }

int se2gpu_orb_set_stream(se2gpu_orb* h, void* s) {
    SE2_REQUIRE(h, SE2GPU_ERR_INVALID, "orb handle is NULL");
    h->stream = s ? (hipStream_t)s : h->own_stream;
    return SE2GPU_OK;
}

int se2gpu_orb_sync(se2gpu_orb* h) {
    SE2_REQUIRE(h, SE2GPU_ERR_INVALID, "orb handle is NULL");
    SE2_HIP(hipStreamSynchronize(h->stream));
    return orb_check_overflow(h);
}

int se2gpu_orb_extract_batch_device(se2gpu_orb* h, const uint8_t* d_imgs, int nframes, int rows, int cols,
                                    se2gpu_keypoint* d_kps, uint8_t* d_desc, int32_t* d_counts, int cap) {
    SE2_REQUIRE(h && d_imgs && d_kps && d_desc && d_counts, SE2GPU_ERR_INVALID, "extract_batch: NULL argument");
    SE2_REQUIRE(nframes >= 1 && nframes <= h->max_batch, SE2GPU_ERR_CAPACITY, "nframes %d exceeds max_batch %d", nframes,
                h->max_batch);
    SE2_REQUIRE(rows > 0 && cols > 0, SE2GPU_ERR_INVALID, "image %dx%d", rows, cols);
    // max_rows / max_cols are a sizing hint: a larger image grows the buffers (the reference's extractor takes any size)
    h->params.max_rows = std::max(h->params.max_rows, rows);
    h->params.max_cols = std::max(h->params.max_cols, cols);
    SE2_REQUIRE(cap > 0, SE2GPU_ERR_INVALID, "cap must be positive");
    SE2_CHECK(orb_configure(h, rows, cols));
    return orb_run(h, d_imgs, cols, nframes, d_kps, d_desc, d_counts, cap);
}

int se2gpu_orb_extract(se2gpu_orb* h, const uint8_t* img, int rows, int cols, size_t step, const uint8_t* mask,
                       se2gpu_keypoint* kps, uint8_t* desc, int cap, int* n_out) {
    SE2_REQUIRE(h && n_out, SE2GPU_ERR_INVALID, "orb_extract: NULL argument");
    // A mask is accepted and has no effect, exactly as in the reference: ComputePyramid builds mvMaskPyramid from it
    // (ORBextractor.cpp:797-828) but ComputeKeyPoints calls cv::FAST without it (:616, :622) - no key point is ever masked.
    (void)mask;
    *n_out = 0;
    if (!img || rows == 0 || cols == 0) return SE2GPU_OK;  // _image.empty(): silent return (ORBextractor.cpp:730)
    SE2_REQUIRE(kps && desc && cap > 0, SE2GPU_ERR_INVALID, "orb_extract: NULL output");
    SE2_REQUIRE(rows > 0 && cols > 0, SE2GPU_ERR_INVALID, "image %dx%d", rows, cols);
    // max_rows / max_cols are a sizing hint: a larger image grows the buffers (the reference's extractor takes any size)
    h->params.max_rows = std::max(h->params.max_rows, rows);
    h->params.max_cols = std::max(h->params.max_cols, cols);
    SE2_CHECK(orb_configure(h, rows, cols));
    hipStream_t st = h->stream;
    SE2_CHECK(h->img.reserve((size_t)h->params.max_rows * h->params.max_cols));
    // result block [count, overflow | key points (cap) | descriptors (cap)]: one download, one synchronisation
    const size_t o_kps = 16, o_desc = (o_kps + (size_t)cap * sizeof(se2gpu_keypoint) + 15) & ~(size_t)15;
    const size_t out_b = o_desc + (size_t)cap * 32, img_b = (size_t)rows * cols;
    SE2_CHECK(h->out_d.reserve(out_b));
    SE2_CHECK(h->stage_h.reserve(std::max(out_b, img_b)));
    for (int r = 0; r < rows; ++r) std::memcpy(h->stage_h.p + (size_t)r * cols, img + (size_t)r * step, (size_t)cols);
    SE2_HIP(hipMemcpyAsync(h->img.p, h->stage_h.p, img_b, hipMemcpyHostToDevice, st));
    uint8_t* d_out = h->out_d.p;
    SE2_CHECK(orb_run(h, h->img.p, cols, 1, (se2gpu_keypoint*)(d_out + o_kps), d_out + o_desc, (int*)d_out, cap));
    SE2_HIP(hipMemcpyAsync(d_out + 4, h->overflow.p, sizeof(int), hipMemcpyDeviceToDevice, st));
    SE2_HIP(hipMemcpyAsync(h->stage_h.p, d_out, out_b, hipMemcpyDeviceToHost, st));   // the upload has completed: stream order
    SE2_HIP(hipStreamSynchronize(st));
    const int* hdr = (const int*)h->stage_h.p;
    if (hdr[1]) {
        SE2_HIP(hipMemsetAsync(h->overflow.p, 0, sizeof(int), st));
        set_error("ORB extractor: internal capacity overflow (mask %d): cell candidates / level list / output cap", hdr[1]);
        return SE2GPU_ERR_CAPACITY;
    }
    const int n = std::min(hdr[0], cap);
    if (n) {
        std::memcpy(kps, h->stage_h.p + o_kps, (size_t)n * sizeof(se2gpu_keypoint));
        std::memcpy(desc, h->stage_h.p + o_desc, (size_t)n * 32);
    }
    *n_out = n;
    return SE2GPU_OK;
}

int se2gpu_orb_debug_level(se2gpu_orb* h, int frame, int level, int blurred, uint8_t* out, size_t out_cap, int* rows,
                           int* cols) {
    SE2_REQUIRE(h && out && rows && cols, SE2GPU_ERR_INVALID, "debug_level: NULL argument");
    SE2_REQUIRE(frame >= 0 && frame < h->last_batch && level >= 0 && level < h->g.nlevels, SE2GPU_ERR_INVALID,
                "debug_level: frame/level out of range");
    const Geom& g = h->g;
    const int e = (blurred & 2) ? kEdge : 0;   // bit 1: include the 16 px frame
    const int ow = g.w[level] + 2 * e, oh = g.h[level] + 2 * e;
    SE2_REQUIRE(out_cap >= (size_t)ow * oh, SE2GPU_ERR_CAPACITY, "debug_level: buffer too small");
    SE2_HIP(hipStreamSynchronize(h->stream));
    if (h->side_stream) SE2_HIP(hipStreamSynchronize(h->side_stream));
    const uint8_t* src = ((blurred & 1) ? h->blur.p : h->pyr.p) + (size_t)frame * g.frame_bytes + g.off[level] +
                         (size_t)(kEdge - e) * g.stride[level] + (kEdge - e);
    SE2_HIP(hipMemcpy2D(out, ow, src, g.stride[level], ow, oh, hipMemcpyDeviceToHost));
    *rows = oh;
    *cols = ow;
    return SE2GPU_OK;
}

int se2gpu_orb_debug_score(se2gpu_orb* h, int frame, int level, uint8_t* out, size_t out_cap, int* rows, int* cols) {
    SE2_REQUIRE(h && out && rows && cols, SE2GPU_ERR_INVALID, "debug_score: NULL argument");
    SE2_REQUIRE(frame >= 0 && frame < h->last_batch && level >= 0 && level < h->g.nlevels, SE2GPU_ERR_INVALID,
                "debug_score: frame/level out of range");
    const Geom& g = h->g;
    SE2_REQUIRE(out_cap >= (size_t)g.w[level] * g.h[level], SE2GPU_ERR_CAPACITY, "debug_score: buffer too small");
    const Geom& gl = h->last_lists;
    const size_t bytes = (size_t)g.w[level] * g.h[level];
    SE2_CHECK(h->score.reserve(bytes));
    SE2_HIP(hipMemsetAsync(h->score.p, 0, bytes, h->stream));
    hipLaunchKernelGGL(k_scatter_lists, dim3(gl.lst_base[level + 1] - gl.lst_base[level]), dim3(256), 0, h->stream, gl, frame,
                       level, h->lst_ent.p, h->lst_cnt.p, h->score.p);
    SE2_HIP(hipMemcpyAsync(out, h->score.p, bytes, hipMemcpyDeviceToHost, h->stream));
    SE2_HIP(hipStreamSynchronize(h->stream));
    *rows = g.h[level];
    *cols = g.w[level];
    return SE2GPU_OK;
}

// {kernel the next batch would be scored with (0 dense, 1 candidates only), last measured candidate density * 1e6 or -1}
int se2gpu_orb_score_kernel(se2gpu_orb* h, int info[2]) {
    SE2_REQUIRE(h && info, SE2GPU_ERR_INVALID, "orb_score_kernel: NULL argument");
    info[0] = h->score_mode == 2 || (h->score_mode == 0 && h->use_sparse) ? 1 : 0;
    info[1] = h->fs_density < 0 ? -1 : (int)(h->fs_density * 1e6);
    return SE2GPU_OK;
}

int se2gpu_orb_profile(se2gpu_orb* h, int enable) {
    SE2_REQUIRE(h, SE2GPU_ERR_INVALID, "orb handle is NULL");
    h->prof.enabled = enable != 0;
    h->prof.reset();
    return SE2GPU_OK;
}

int se2gpu_orb_debug_nth_element(uint64_t* entries, int n, int nth, int force_global) {
    SE2_REQUIRE(entries && n > 0 && nth >= 0 && nth < n && n <= (1 << 24), SE2GPU_ERR_INVALID, "nth_element: bad arguments");
    SE2_REQUIRE(have_device(), SE2GPU_ERR_NO_DEVICE, "no HIP device");
    DevBuf<unsigned long long> a;
    DevBuf<uint32_t> scr;
    SE2_CHECK(a.reserve((size_t)n));
    SE2_CHECK(scr.reserve(2 * (size_t)n));
    SE2_HIP(hipMemcpy(a.p, entries, (size_t)n * sizeof(uint64_t), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_debug_nth, dim3(1), dim3(64), 0, nullptr, a.p, scr.p, n, nth, force_global);
    SE2_HIP(hipGetLastError());
    SE2_HIP(hipDeviceSynchronize());
    SE2_HIP(hipMemcpy(entries, a.p, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return SE2GPU_OK;
}

int se2gpu_orb_profile_get(se2gpu_orb* h, int idx, const char** name, double* ms, int64_t* launches) {
    SE2_REQUIRE(h, SE2GPU_ERR_INVALID, "orb handle is NULL");
    if (idx < 0 || idx >= (int)h->prof.slots.size()) return SE2GPU_ERR_INVALID;
    if (name) *name = h->prof.slots[idx].name;
    if (ms) *ms = h->prof.slots[idx].ms;
    if (launches) *launches = h->prof.slots[idx].launches;
    return SE2GPU_OK;
}

}  // extern "C"
