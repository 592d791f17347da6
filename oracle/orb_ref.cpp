// ORACLE - TEST INFRASTRUCTURE ONLY.  Never linked, imported or called by the product path
// (se2lam_amd/, include/).  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline`
// leg may use it, and only as the checker / reported CPU baseline.
//
// PARITY: pinned against the REFERENCE'S OWN CODE for everything se2lam wrote, unpinned for the OpenCV arithmetic under it.
// /root/reference holds no golden vectors for this path and OpenCV is not installed in this image - but its extractor
// compiles: oracle/_ref (`make -C oracle ref`) builds /root/reference/src/ORBextractor.cpp unmodified against a stand-in for
// the OpenCV headers (oracle/_shim), and tests/test_ref_compiled.py + tools/fuzz_ref.py hold this restatement to it, key point
// for key point and descriptor byte for byte (ten config-1 frames, both score types, odd sizes, noise, 700 random cases).
// That pins the constructor's tables, the pyramid loop, the cell grid / quota redistribution / 20-7 threshold rule, the
// level cut, IC_Angle, computeOrbDescriptor, HarrisResponses and operator().  The OpenCV functions those call (FAST, resize,
// copyMakeBorder, GaussianBlur, retainBest, fastAtan2, cvRound) are restated here and, independently, in
// oracle/_shim/cv_shim.cpp; the two agree bit for bit, the real library could not be run.
// One known, understood difference: the descriptor's steering cosine / sine - the reference's expression resolves to libm's
// cosf / sinf, this restatement rounds the double values (see Extractor::descriptor; orb_ref_set_trig_libm switches): about
// one descriptor in two million differs by one or two bits under glibc 2.35, none in libm mode.  Pinned
// from the tree itself in tests/test_orb_oracle.py: the 256x4 pattern table (sha256), umax, per-level quotas / sizes / cell
// grids, EDGE_THRESHOLD / PATCH_SIZE, the Gaussian taps.
//
// CPU restatement (single thread, no dependencies) of se2lam::ORBextractor:
//   ctor                 /root/reference/src/ORBextractor.cpp:463-520
//   ComputePyramid       :790-831    (resize INTER_LINEAR + copyMakeBorder REFLECT_101, 16 px)
//   ComputeKeyPoints     :531-716    (per-cell FAST 20 / fallback 7, quota redistribution, retain best)
//   IC_Angle             :130-157    computeOrbDescriptor :161-200   operator() :727-788
// OpenCV semantics restated (version: "2.4.x / 3.1 above", README.MD:27; CI = Ubuntu 18.04 libopencv 3.2):
//   cv::FAST (FAST-9/16, threshold, non-max suppression, score = cornerScore<16>)
//   cv::resize INTER_LINEAR 8UC1: 11-bit fixed-point coefficients, src = (dst+0.5)*scale-0.5,
//                                 vertical pass (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2
//   cv::GaussianBlur 7x7 sigma 2 on 8U: separable, taps round(k*256) = {18,34,49,55,49,34,18}, (v + 2^15) >> 16
//   cv::KeyPointsFilter::retainBest: std::nth_element by response + the reference's resize(n) = the first n entries of the
//       permutation libstdc++'s introselect leaves (stl_nth.h: restated, held to this machine's std::nth_element) - which of
//       the tied key points stay and in what order they enter the level's list is what a GCC build of the reference does
//       (round 5; rounds 1-4 defined an order of their own: orb_ref_set_retain_stable).
//   cv::fastAtan2 (degree-7 odd polynomial), cvRound = round-half-to-even, cvFloor, cvCeil.
//   cos/sin of the keypoint angle: the float overloads = libm's cosf / sinf; glibc's algorithm restated (glibc_flt32, mode 2).
//
// Build: g++ -O2 -ffp-contract=off (oracle/Makefile): x*b + y*a must be mul, mul, add (no FMA).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "stl_nth.h"

extern "C" {
struct orb_ref_params {
    int32_t nfeatures;    // 1000
    float scale_factor;   // 1.2f
    int32_t nlevels;      // 8
    int32_t fast_th;      // 20
    int32_t score_type;   // 1 = ORB::FAST_SCORE (the reference's default), 0 = ORB::HARRIS_SCORE (ORBextractor.h:44)
};
struct orb_ref_keypoint {  // cv::KeyPoint layout
    float x, y, size, angle, response;
    int32_t octave, class_id;
};
}

namespace {

const int PATCH_SIZE = 31, HALF_PATCH_SIZE = 15, EDGE_THRESHOLD = 16;

const int kPattern[256 * 4] = {
#include "orb_pattern_31.inc"
};

static int g_retain_stable = 0;   // see Extractor::retain_best
static int g_trig_libm = 2;   // see Extractor::descriptor: 0 rounded double cosine / sine, 1 libm's cosf / sinf, 2 glibc's algorithm restated

// glibc's sinf / cosf (2.28 and later: sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h, s_sincosf_data.c - the routines ARM
// contributed) for the arguments a key-point angle can take, 0 <= y < 120: a range reduction by one multiply with 2/pi and one
// multiply-subtract with pi/2, then a degree-7 sine or degree-8 cosine polynomial, all in double, rounded to float once.  Plain
// IEEE double arithmetic - so it is the same on every machine, which libm's result is not obliged to be - and on this glibc
// (2.35) it IS libm: tools/glibc_sincosf_check.c compares it with sinf / cosf on all 1,086,918,621 floats of [0, 2 pi], compiled
// both without and with FMA contraction: no difference (profiles/r04_glibc_sincosf_check.txt).
namespace glibc_flt32 {
const double hpi_inv = 0x1.45F306DC9C883p+23;   // 2 / pi * 2^24: the quadrant ends up in bits 24..31 of the truncated product
const double hpi = 0x1.921FB54442D18p0;
const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5, c3 = -0x1.6c087e89a359dp-10, c4 = 0x1.99343027bf8c3p-16;
const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
inline uint32_t abstop12(float x) { uint32_t u; std::memcpy(&u, &x, 4); return (u >> 20) & 0x7ff; }
// sine polynomial in quadrant n even, cosine polynomial (negated in quadrants 2, 3) in n odd
inline float poly(double x, double x2, int n, double csign) {
    if ((n & 1) == 0) {
        const double x3 = x * x2, t1 = s2 + x2 * s3, x7 = x3 * x2, s = x + x3 * s1;
        return (float)(s + x7 * t1);
    }
    const double x4 = x2 * x2, k2 = csign * c3 + x2 * (csign * c4), k1 = csign * c0 + x2 * (csign * c1), x6 = x4 * x2, c = k1 + x4 * (csign * c2);
    return (float)(c + x6 * k2);
}
inline float sincos(float y, int cosine) {
    double x = y;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {            // |y| < pi / 4
        if (abstop12(y) < abstop12(0x1p-12f)) return cosine ? 1.0f : y;
        return poly(x, x * x, cosine, 1.0);
    }
    const double r = x * hpi_inv;
    const int n = ((int32_t)r + 0x800000) >> 24;
    x = x - n * hpi;
    const double sgn = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    return poly(x * sgn, x * x, n ^ cosine, (n & 2) ? -1.0 : 1.0);
}
}  // namespace glibc_flt32
inline int cv_round(double v) { return (int)std::nearbyint(v); }  // default rounding mode: half to even
inline int cv_round(float v) { return (int)std::nearbyintf(v); }
inline int cv_floor(float v) { int i = (int)v; return i - (i > v); }
inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }
inline int cv_ceil(double v) { int i = (int)v; return i + (i < v); }

// cv::fastAtan2 (OpenCV 3.x mathfuncs_core), degrees in [0, 360)
float fast_atan2(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    float ax = std::abs(x), ay = std::abs(y);
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

struct Level {
    int w = 0, h = 0, stride = 0;       // interior size, stride of the bordered buffer
    std::vector<uint8_t> buf, blur;     // (h+32) x (w+32) with a 16 px border
    uint8_t* at(int y, int x) { return &buf[(size_t)(y + EDGE_THRESHOLD) * stride + x + EDGE_THRESHOLD]; }
    const uint8_t* at(int y, int x) const { return &buf[(size_t)(y + EDGE_THRESHOLD) * stride + x + EDGE_THRESHOLD]; }
    uint8_t* bat(int y, int x) { return &blur[(size_t)(y + EDGE_THRESHOLD) * stride + x + EDGE_THRESHOLD]; }
};

struct Extractor {
    int nfeatures, nlevels, fastTh, scoreType;
    double scaleFactor;  // member is double (ORBextractor.h:68), constructed from a float
    std::vector<float> mvScaleFactor, mvInvScaleFactor;
    std::vector<int> mnFeaturesPerLevel, umax;
    std::vector<Level> pyr;

    explicit Extractor(const orb_ref_params& p)
        : nfeatures(p.nfeatures), nlevels(p.nlevels), fastTh(p.fast_th), scoreType(p.score_type),
          scaleFactor(p.scale_factor) {
        mvScaleFactor.resize(nlevels);
        mvScaleFactor[0] = 1;
        for (int i = 1; i < nlevels; i++) mvScaleFactor[i] = (float)(mvScaleFactor[i - 1] * scaleFactor);
        float invScaleFactor = (float)(1.0f / scaleFactor);
        mvInvScaleFactor.resize(nlevels);
        mvInvScaleFactor[0] = 1;
        for (int i = 1; i < nlevels; i++) mvInvScaleFactor[i] = mvInvScaleFactor[i - 1] * invScaleFactor;
        mnFeaturesPerLevel.resize(nlevels);
        float factor = (float)(1.0 / scaleFactor);
        float nDesired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
        int sum = 0;
        for (int level = 0; level < nlevels - 1; level++) {
            mnFeaturesPerLevel[level] = cv_round(nDesired);
            sum += mnFeaturesPerLevel[level];
            nDesired *= factor;
        }
        mnFeaturesPerLevel[nlevels - 1] = std::max(nfeatures - sum, 0);
        // circular patch row extents (ORBextractor.cpp:503-519)
        umax.assign(HALF_PATCH_SIZE + 1, 0);
        int v, v0, vmax = cv_floor(HALF_PATCH_SIZE * std::sqrt(2.f) / 2 + 1);
        int vmin = cv_ceil(HALF_PATCH_SIZE * std::sqrt(2.f) / 2);
        const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
        for (v = 0; v <= vmax; ++v) umax[v] = cv_round(std::sqrt(hp2 - v * v));
        for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
            while (umax[v0] == umax[v0 + 1]) ++v0;
            umax[v] = v0;
            ++v0;
        }
    }

    void level_size(int rows, int cols, int level, int& w, int& h) const {
        float scale = mvInvScaleFactor[level];
        w = cv_round((float)cols * scale);
        h = cv_round((float)rows * scale);
    }

    // BORDER_REFLECT_101 fill of the 16 px frame around the interior
    static void make_border(Level& L) {
        const int W = L.w, H = L.h, B = EDGE_THRESHOLD;
        auto refl = [](int p, int n) { return p < 0 ? -p : (p >= n ? 2 * n - 2 - p : p); };
        for (int y = -B; y < H + B; ++y) {
            const int sy = refl(y, H);
            for (int x = -B; x < W + B; ++x) {
                if (x >= 0 && x < W && y >= 0 && y < H) continue;
                *L.at(y, x) = *L.at(sy, refl(x, W));
            }
        }
    }

    // cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR), 8UC1, OpenCV 3.2 fixed-point path
    static void resize_linear(const Level& S, Level& D) {
        const int sw = S.w, sh = S.h, dw = D.w, dh = D.h;
        const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
        const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
        const int SCALE = 2048;
        std::vector<int> xofs(dw), yofs(dh);
        std::vector<short> ialpha(2 * dw), ibeta(2 * dh);
        int xmax = dw;
        for (int dx = 0; dx < dw; dx++) {
            float fx = (float)((dx + 0.5) * scale_x - 0.5);
            int sx = cv_floor(fx);
            fx -= sx;
            if (sx < 0) { fx = 0; sx = 0; }
            if (sx + 1 >= sw) {
                xmax = std::min(xmax, dx);
                if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
            }
            xofs[dx] = sx;
            float cb[2] = {1.f - fx, fx};
            for (int k = 0; k < 2; k++) {
                int v = cv_round(cb[k] * SCALE);
                ialpha[dx * 2 + k] = (short)std::min(std::max(v, -32768), 32767);
            }
        }
        for (int dy = 0; dy < dh; dy++) {
            float fy = (float)((dy + 0.5) * scale_y - 0.5);
            int sy = cv_floor(fy);
            fy -= sy;
            yofs[dy] = sy;
            float cb[2] = {1.f - fy, fy};
            for (int k = 0; k < 2; k++) {
                int v = cv_round(cb[k] * SCALE);
                ibeta[dy * 2 + k] = (short)std::min(std::max(v, -32768), 32767);
            }
        }
        std::vector<int> row0(dw), row1(dw);
        auto hresize = [&](int sy, std::vector<int>& out) {
            const uint8_t* Sp = S.at(sy, 0);
            int dx = 0;
            for (; dx < xmax; dx++) {
                const int sx = xofs[dx];
                out[dx] = Sp[sx] * ialpha[dx * 2] + Sp[sx + 1] * ialpha[dx * 2 + 1];
            }
            for (; dx < dw; dx++) out[dx] = Sp[xofs[dx]] * SCALE;
        };
        auto clip = [](int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; };
        for (int dy = 0; dy < dh; dy++) {
            const int sy0 = clip(yofs[dy], 0, sh), sy1 = clip(yofs[dy] + 1, 0, sh);
            hresize(sy0, row0);
            hresize(sy1, row1);
            const int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
            uint8_t* Dp = D.at(dy, 0);
            for (int x = 0; x < dw; x++)
                Dp[x] = (uint8_t)((((b0 * (row0[x] >> 4)) >> 16) + ((b1 * (row1[x] >> 4)) >> 16) + 2) >> 2);
        }
    }

    void compute_pyramid(const uint8_t* img, int rows, int cols, int step) {
        pyr.assign(nlevels, Level());
        for (int level = 0; level < nlevels; ++level) {
            Level& L = pyr[level];
            level_size(rows, cols, level, L.w, L.h);
            L.stride = L.w + 2 * EDGE_THRESHOLD;
            L.buf.assign((size_t)L.stride * (L.h + 2 * EDGE_THRESHOLD), 0);
            if (level == 0) {
                for (int y = 0; y < rows; ++y) std::memcpy(L.at(y, 0), img + (size_t)y * step, cols);
            } else {
                resize_linear(pyr[level - 1], L);
            }
            make_border(L);
        }
    }

    // FAST-9/16 score S(x,y) = max over the 16 arcs of 9 contiguous circle pixels of min(v - p_i) and of
    // min(p_i - v), clamped at 0.  A pixel is a corner at threshold t iff S > t, and cv::FAST's
    // cornerScore<16> equals S - 1 for such a pixel; non-max suppression inside a FAST call compares
    // cornerScore over the 8 neighbours that lie in the call's scan area (others count as 0).
    static int fast_score(const uint8_t* p, int stride) {
        static const int ox[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
        static const int oy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
        int d[25];
        const int v = p[0];
        for (int k = 0; k < 16; ++k) d[k] = v - p[oy[k] * stride + ox[k]];
        for (int k = 16; k < 25; ++k) d[k] = d[k - 16];
        int best = 0;
        for (int k = 0; k < 16; ++k) {
            int mn = d[k], mx = d[k];
            for (int m = 1; m < 9; ++m) {
                mn = std::min(mn, d[k + m]);
                mx = std::max(mx, d[k + m]);
            }
            best = std::max(best, std::max(mn, -mx));
        }
        return best;
    }

    void score_map(int level, std::vector<uint8_t>& S) const {
        const Level& L = pyr[level];
        S.assign((size_t)L.w * L.h, 0);
        for (int y = 3; y < L.h - 3; ++y)
            for (int x = 3; x < L.w - 3; ++x) S[(size_t)y * L.w + x] = (uint8_t)fast_score(L.at(y, x), L.stride);
    }

    struct KP {
        int x, y;     // level coordinates (interior)
        float resp;   // cornerScore = S - 1 (exact in float), or the Harris response
    };

    // HarrisResponses(cellImage, pts, 7, HARRIS_K) of ORBextractor.cpp:85-126 for one key point: the cell image is a
    // view into the (un-blurred) level image, so the 7x7 block of 3x3 Sobel windows is read from the level itself.
    // Float expression evaluated in the reference's order (no contraction: -ffp-contract=off).
    float harris_response(const Level& L, int x, int y) const {
        const int blockSize = 7, r = blockSize / 2;
        const float harris_k = 0.04f;
        float scale = (1 << 2) * blockSize * 255.0f;
        scale = 1.0f / scale;
        const float scale_sq_sq = scale * scale * scale * scale;
        const int step = L.stride;
        const uint8_t* ptr0 = L.at(y - r, x - r);
        int a = 0, b = 0, c = 0;
        for (int i = 0; i < blockSize; i++)
            for (int j = 0; j < blockSize; j++) {
                const uint8_t* ptr = ptr0 + i * step + j;
                const int Ix = (ptr[1] - ptr[-1]) * 2 + (ptr[-step + 1] - ptr[-step - 1]) + (ptr[step + 1] - ptr[step - 1]);
                const int Iy = (ptr[step] - ptr[-step]) * 2 + (ptr[step - 1] - ptr[-step - 1]) + (ptr[step + 1] - ptr[-step + 1]);
                a += Ix * Ix;
                b += Iy * Iy;
                c += Ix * Iy;
            }
        return ((float)a * b - (float)c * c - harris_k * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
    }

    // cv::FAST(cell, kps, thr, true) over the cell window [x0, x0+w) x [y0, y0+h): scans rows/cols 3..dim-4.
    static void fast_cell(const std::vector<uint8_t>& S, int W, int x0, int y0, int w, int h, int thr,
                          std::vector<KP>& out) {
        out.clear();
        const int xa = x0 + 3, xb = x0 + w - 3, ya = y0 + 3, yb = y0 + h - 3;  // scan area [xa,xb) x [ya,yb)
        auto sc = [&](int x, int y) -> int {
            if (x < xa || x >= xb || y < ya || y >= yb) return 0;
            const int s = S[(size_t)y * W + x];
            return s > thr ? s - 1 : 0;
        };
        for (int y = ya; y < yb; ++y)
            for (int x = xa; x < xb; ++x) {
                const int s = sc(x, y);
                if (s == 0) continue;
                if (s > sc(x - 1, y) && s > sc(x + 1, y) && s > sc(x - 1, y - 1) && s > sc(x, y - 1) &&
                    s > sc(x + 1, y - 1) && s > sc(x - 1, y + 1) && s > sc(x, y + 1) && s > sc(x + 1, y + 1))
                    out.push_back(KP{x, y, (float)s});
            }
    }

    struct LevelGeom {
        int levelCols, levelRows, cellW, cellH, nCells, nfeaturesCell, W, H, maxBorderX, maxBorderY;
    };
    LevelGeom geometry(int level) const {
        LevelGeom g;
        const float imageRatio = (float)pyr[0].w / pyr[0].h;
        const int nDesired = mnFeaturesPerLevel[level];
        g.levelCols = (int)std::sqrt((float)nDesired / (5 * imageRatio));
        g.levelRows = (int)(imageRatio * g.levelCols);
        g.maxBorderX = pyr[level].w - EDGE_THRESHOLD;
        g.maxBorderY = pyr[level].h - EDGE_THRESHOLD;
        g.W = g.maxBorderX - EDGE_THRESHOLD;
        g.H = g.maxBorderY - EDGE_THRESHOLD;
        g.cellW = (int)std::ceil((float)g.W / g.levelCols);
        g.cellH = (int)std::ceil((float)g.H / g.levelRows);
        g.nCells = g.levelRows * g.levelCols;
        g.nfeaturesCell = (int)std::ceil((float)nDesired / g.nCells);
        return g;
    }

    // cv::KeyPointsFilter::retainBest(v, n) followed by the reference's v.resize(n): see stl_nth.h.  g_retain_stable = 1 is the
    // order rounds 1-4 defined instead (the n best by response, ties by position in the list, list order kept) - one of the
    // outcomes the C++ standard allows, not the one a GCC build of the reference produces; kept as a labelled alternative.
    static void retain_best(std::vector<KP>& v, int n) {
        if (n < 0 || (int)v.size() <= n) return;
        if (n == 0) { v.clear(); return; }
        if (g_retain_stable) {
            std::vector<int> idx(v.size());
            for (size_t q = 0; q < idx.size(); ++q) idx[q] = (int)q;
            std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return v[a].resp > v[b].resp; });
            std::vector<KP> kept;
            for (int q = 0; q < n; ++q) kept.push_back(v[idx[q]]);
            v.swap(kept);
            return;
        }
        stl_nth::nth_element(v.data(), v.data() + n, v.data() + v.size(), [](const KP& a, const KP& b) { return a.resp > b.resp; });
        v.resize(n);
    }

    void compute_keypoints(std::vector<std::vector<orb_ref_keypoint>>& all) {
        all.assign(nlevels, {});
        for (int level = 0; level < nlevels; ++level) {
            const int nDesired = mnFeaturesPerLevel[level];
            const LevelGeom g = geometry(level);
            const int levelCols = g.levelCols, levelRows = g.levelRows, cellW = g.cellW, cellH = g.cellH;
            const int nCells = g.nCells, nfeaturesCell = g.nfeaturesCell;
            std::vector<uint8_t> S;
            score_map(level, S);
            std::vector<std::vector<std::vector<KP>>> cellKP(levelRows, std::vector<std::vector<KP>>(levelCols));
            std::vector<std::vector<int>> nToRetain(levelRows, std::vector<int>(levelCols, 0));
            std::vector<std::vector<int>> nTotal(levelRows, std::vector<int>(levelCols, 0));
            std::vector<std::vector<char>> bNoMore(levelRows, std::vector<char>(levelCols, 0));
            std::vector<int> iniXCol(levelCols, 0), iniYRow(levelRows, 0);
            int nNoMore = 0, nToDistribute = 0;
            float hY = (float)(cellH + 6);
            for (int i = 0; i < levelRows; i++) {
                const float iniY = (float)(EDGE_THRESHOLD + i * cellH - 3);
                iniYRow[i] = (int)iniY;
                if (i == levelRows - 1) {
                    hY = g.maxBorderY + 3 - iniY;
                    if (hY <= 0) continue;
                }
                float hX = (float)(cellW + 6);
                for (int j = 0; j < levelCols; j++) {
                    float iniX;
                    if (i == 0) {
                        iniX = (float)(EDGE_THRESHOLD + j * cellW - 3);
                        iniXCol[j] = (int)iniX;
                    } else {
                        iniX = (float)iniXCol[j];
                    }
                    if (j == levelCols - 1) {
                        hX = g.maxBorderX + 3 - iniX;
                        if (hX <= 0) continue;
                    }
                    // rowRange(iniY, iniY+hY).colRange(iniX, iniX+hX): float -> int conversions
                    const int y0 = (int)iniY, y1 = (int)(iniY + hY), x0 = (int)iniX, x1 = (int)(iniX + hX);
                    std::vector<KP>& kps = cellKP[i][j];
                    fast_cell(S, pyr[level].w, x0, y0, x1 - x0, y1 - y0, fastTh, kps);
                    if (kps.size() <= 3) fast_cell(S, pyr[level].w, x0, y0, x1 - x0, y1 - y0, 7, kps);
                    if (scoreType == 0)   // ORB::HARRIS_SCORE (ORBextractor.cpp:625-629)
                        for (KP& k : kps) k.resp = harris_response(pyr[level], k.x, k.y);
                    const int nKeys = (int)kps.size();
                    nTotal[i][j] = nKeys;
                    if (nKeys > nfeaturesCell) {
                        nToRetain[i][j] = nfeaturesCell;
                        bNoMore[i][j] = 0;
                    } else {
                        nToRetain[i][j] = nKeys;
                        nToDistribute += nfeaturesCell - nKeys;
                        bNoMore[i][j] = 1;
                        nNoMore++;
                    }
                }
            }
            while (nToDistribute > 0 && nNoMore < nCells) {
                const int nNew = nfeaturesCell + (int)std::ceil((float)nToDistribute / (nCells - nNoMore));
                nToDistribute = 0;
                for (int i = 0; i < levelRows; i++)
                    for (int j = 0; j < levelCols; j++)
                        if (!bNoMore[i][j]) {
                            if (nTotal[i][j] > nNew) {
                                nToRetain[i][j] = nNew;
                                bNoMore[i][j] = 0;
                            } else {
                                nToRetain[i][j] = nTotal[i][j];
                                nToDistribute += nNew - nTotal[i][j];
                                bNoMore[i][j] = 1;
                                nNoMore++;
                            }
                        }
            }
            std::vector<KP> list;
            const int scaledPatchSize = (int)(PATCH_SIZE * mvScaleFactor[level]);
            for (int i = 0; i < levelRows; i++)
                for (int j = 0; j < levelCols; j++) {
                    std::vector<KP>& kc = cellKP[i][j];
                    // KeyPointsFilter::retainBest(keysCell, nToRetain) + resize (ORBextractor.cpp:692-694): the first nToRetain
                    // entries of what libstdc++'s nth_element leaves of FAST's row-major list (stl_nth.h)
                    retain_best(kc, nToRetain[i][j]);
                    for (const KP& k : kc) list.push_back(k);
                }
            if ((int)list.size() > nDesired) retain_best(list, nDesired);   // :706-710
            for (const KP& k : list) {
                orb_ref_keypoint kp;
                kp.x = (float)k.x;
                kp.y = (float)k.y;
                kp.size = (float)scaledPatchSize;
                kp.angle = -1;
                kp.response = k.resp;
                kp.octave = level;
                kp.class_id = -1;
                all[level].push_back(kp);
            }
        }
        for (int level = 0; level < nlevels; ++level)
            for (auto& kp : all[level]) kp.angle = ic_angle(pyr[level], kp.x, kp.y);
    }

    float ic_angle(const Level& L, float px, float py) const {
        int m_01 = 0, m_10 = 0;
        const uint8_t* center = L.at(cv_round(py), cv_round(px));
        for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
        const int step = L.stride;
        for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
            int v_sum = 0;
            const int d = umax[v];
            for (int u = -d; u <= d; ++u) {
                const int val_plus = center[u + v * step], val_minus = center[u - v * step];
                v_sum += (val_plus - val_minus);
                m_10 += u * (val_plus + val_minus);
            }
            m_01 += v * v_sum;
        }
        return fast_atan2((float)m_01, (float)m_10);
    }

    static void gaussian_taps(int taps[7]) {  // getGaussianKernel(7, 2, CV_32F) -> convertTo(CV_32S, 256)
        float cf[7];
        double sum = 0;
        const double scale2X = -0.5 / (2.0 * 2.0);
        for (int i = 0; i < 7; i++) {
            const double x = i - 3.0;
            cf[i] = (float)std::exp(scale2X * x * x);
            sum += cf[i];
        }
        sum = 1. / sum;
        for (int i = 0; i < 7; i++) {
            cf[i] = (float)(cf[i] * sum);
            taps[i] = cv_round((double)cf[i] * 256.0);
        }
    }

    // GaussianBlur(level, level, Size(7,7), 2, 2, BORDER_REFLECT_101) on the interior; the 16 px frame keeps the
    // un-blurred reflect copies (the blur writes the ROI only, ORBextractor.cpp:768-769,796-798).
    void blur_level(int level) {
        Level& L = pyr[level];
        L.blur = L.buf;
        int k[7];
        gaussian_taps(k);
        std::vector<int> tmp((size_t)(L.h + 6) * L.w);
        for (int y = -3; y < L.h + 3; ++y)
            for (int x = 0; x < L.w; ++x) {
                const uint8_t* p = L.at(y, x);
                int s = 0;
                for (int t = 0; t < 7; ++t) s += k[t] * p[t - 3];
                tmp[(size_t)(y + 3) * L.w + x] = s;
            }
        for (int y = 0; y < L.h; ++y)
            for (int x = 0; x < L.w; ++x) {
                int s = 0;
                for (int t = 0; t < 7; ++t) s += k[t] * tmp[(size_t)(y + t) * L.w + x];
                const int v = (s + (1 << 15)) >> 16;
                *L.bat(y, x) = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
    }

    void descriptor(const Level& L, const orb_ref_keypoint& kpt, uint8_t* desc) const {
        const float factorPI = (float)(3.14159265358979323846 / 180.f);
        const float angle = (float)kpt.angle * factorPI;
        // The reference writes `(float)cos(angle)` with a float argument under `using namespace std` (src/ORBextractor.cpp:166): C++
        // picks the FLOAT overloads, i.e. libm's cosf / sinf - which are not correctly rounded (glibc 2.35: sinf(0.509999156f) is one
        // ulp below the rounded double sine), so what the reference computes depends on the libm it is linked with.  The restatement
        // (and the HIP path) round the DOUBLE cosine / sine to float - libm-independent; one descriptor in about two million
        // differs in two bits from the reference compiled here, where a sampling coordinate sits on a .5 tie (found by
        // tools/fuzz_ref.py).  g_trig_libm = 1 switches this restatement to cosf / sinf so that the comparison with oracle/_ref is
        // exact on the machine at hand; 2 to glibc's algorithm written out (above): the same bits as 1 on a glibc machine, and the
        // form a GPU kernel can compute.
        float a, b;
        if (g_trig_libm == 2) { a = glibc_flt32::sincos(angle, 1); b = glibc_flt32::sincos(angle, 0); }
        else if (g_trig_libm) { a = cosf(angle); b = sinf(angle); }
        else { a = (float)std::cos((double)angle); b = (float)std::sin((double)angle); }
        const uint8_t* center = &L.blur[(size_t)(cv_round(kpt.y) + EDGE_THRESHOLD) * L.stride + cv_round(kpt.x) + EDGE_THRESHOLD];
        const int step = L.stride;
        const int* pat = kPattern;
        auto get = [&](int idx) -> int {
            const int px = pat[2 * idx], py = pat[2 * idx + 1];
            return center[cv_round(px * b + py * a) * step + cv_round(px * a - py * b)];
        };
        for (int i = 0; i < 32; ++i, pat += 32) {
            int val = 0;
            for (int bit = 0; bit < 8; ++bit) {
                const int t0 = get(2 * bit), t1 = get(2 * bit + 1);
                val |= (t0 < t1) << bit;
            }
            desc[i] = (uint8_t)val;
        }
    }

    int run(const uint8_t* img, int rows, int cols, int step, orb_ref_keypoint* kps, uint8_t* desc, int cap) {
        if (rows <= 0 || cols <= 0) return 0;
        compute_pyramid(img, rows, cols, step);
        std::vector<std::vector<orb_ref_keypoint>> all;
        compute_keypoints(all);
        int n = 0;
        for (int level = 0; level < nlevels; ++level) {
            if (all[level].empty()) continue;
            blur_level(level);
            for (auto& kp : all[level]) {
                if (n >= cap) return -1;
                descriptor(pyr[level], kp, desc + (size_t)n * 32);
                orb_ref_keypoint o = kp;
                if (level != 0) {
                    const float scale = mvScaleFactor[level];
                    o.x *= scale;
                    o.y *= scale;
                }
                kps[n++] = o;
            }
        }
        return n;
    }
};

}  // namespace

extern "C" {

int orb_ref_extract(const orb_ref_params* p, const uint8_t* img, int rows, int cols, int step, orb_ref_keypoint* kps,
                    uint8_t* desc, int cap, int* n_out) {
    Extractor ex(*p);
    const int n = ex.run(img, rows, cols, step, kps, desc, cap);
    if (n < 0) return -1;
    *n_out = n;
    return 0;
}

// tables: scale (nlevels), inv scale (nlevels), quotas (nlevels), umax (16)
void orb_ref_tables(const orb_ref_params* p, float* scale, float* inv_scale, int32_t* quota, int32_t* umax16) {
    Extractor ex(*p);
    for (int i = 0; i < p->nlevels; ++i) {
        scale[i] = ex.mvScaleFactor[i];
        inv_scale[i] = ex.mvInvScaleFactor[i];
        quota[i] = ex.mnFeaturesPerLevel[i];
    }
    for (int i = 0; i < 16; ++i) umax16[i] = ex.umax[i];
}

// out[level] = {w, h, levelCols, levelRows, cellW, cellH, nfeaturesCell}
void orb_ref_geometry(const orb_ref_params* p, int rows, int cols, int32_t* out7) {
    Extractor ex(*p);
    ex.pyr.assign(p->nlevels, Level());
    for (int l = 0; l < p->nlevels; ++l) ex.level_size(rows, cols, l, ex.pyr[l].w, ex.pyr[l].h);
    for (int l = 0; l < p->nlevels; ++l) {
        auto g = ex.geometry(l);
        int32_t* o = out7 + 7 * l;
        o[0] = ex.pyr[l].w; o[1] = ex.pyr[l].h; o[2] = g.levelCols; o[3] = g.levelRows;
        o[4] = g.cellW; o[5] = g.cellH; o[6] = g.nfeaturesCell;
    }
}

// interior of pyramid level `level` (blurred != 0: after GaussianBlur), tight pitch
int orb_ref_level(const orb_ref_params* p, const uint8_t* img, int rows, int cols, int step, int level, int blurred,
                  uint8_t* out, int* w, int* h) {
    Extractor ex(*p);
    ex.compute_pyramid(img, rows, cols, step);
    Level& L = ex.pyr[level];
    if (blurred & 1) ex.blur_level(level);
    const bool bl = blurred & 1;
    if (blurred & 2) {   // with the 16 px frame: (h + 32) x (w + 32)
        const int e = EDGE_THRESHOLD;
        *w = L.w + 2 * e; *h = L.h + 2 * e;
        for (int y = -e; y < L.h + e; ++y)
            std::memcpy(out + (size_t)(y + e) * (L.w + 2 * e), bl ? L.bat(y, -e) : L.at(y, -e), L.w + 2 * e);
        return 0;
    }
    *w = L.w; *h = L.h;
    for (int y = 0; y < L.h; ++y) std::memcpy(out + (size_t)y * L.w, bl ? L.bat(y, 0) : L.at(y, 0), L.w);
    return 0;
}

int orb_ref_score(const orb_ref_params* p, const uint8_t* img, int rows, int cols, int step, int level, uint8_t* out) {
    Extractor ex(*p);
    ex.compute_pyramid(img, rows, cols, step);
    std::vector<uint8_t> S;
    ex.score_map(level, S);
    std::memcpy(out, S.data(), S.size());
    return 0;
}

void orb_ref_pattern(int32_t* out1024) { for (int i = 0; i < 1024; ++i) out1024[i] = kPattern[i]; }
void orb_ref_gaussian_taps(int32_t* out7) { int t[7]; Extractor::gaussian_taps(t); for (int i = 0; i < 7; ++i) out7[i] = t[i]; }
float orb_ref_fast_atan2(float y, float x) { return fast_atan2(y, x); }
int orb_ref_cv_round(float v) { return cv_round(v); }
void orb_ref_set_trig_libm(int mode) { g_trig_libm = mode; }
void orb_ref_set_retain_stable(int on) { g_retain_stable = on; }
// stl_nth.h on caller data (entries: high 32 bits = key, larger is better; low 32 bits travel along), this machine's
// std::nth_element on the same, and a median-of-three killer for it (McIlroy's adversary run against std::nth_element).
void orb_ref_nth_element(uint64_t* e, int n, int nth) {
    stl_nth::nth_element(e, e + nth, e + n, [](uint64_t a, uint64_t b) { return (uint32_t)(a >> 32) > (uint32_t)(b >> 32); });
}
void orb_ref_std_nth_element(uint64_t* e, int n, int nth) {
    std::nth_element(e, e + nth, e + n, [](uint64_t a, uint64_t b) { return (uint32_t)(a >> 32) > (uint32_t)(b >> 32); });
}
long orb_ref_nth_heap_selects() { return stl_nth::heap_selects(); }
void orb_ref_nth_killer(int n, int nth, uint32_t* keys) {
    std::vector<int> val(n, n), idx(n);
    int nsolid = 0, candidate = 0;
    for (int i = 0; i < n; ++i) idx[i] = i;
    auto less = [&](int x, int y) {
        if (val[x] == n && val[y] == n) {
            if (x == candidate) val[x] = nsolid++;
            else val[y] = nsolid++;
        }
        if (val[x] == n) candidate = x;
        else if (val[y] == n) candidate = y;
        return val[x] < val[y];
    };
    std::nth_element(idx.begin(), idx.begin() + nth, idx.end(), less);
    for (int i = 0; i < n; ++i) keys[i] = (uint32_t)(n - val[i]);   // "less" on val = "greater" on n - val
}
float orb_ref_glibc_sincosf(float y, int cosine) { return glibc_flt32::sincos(y, cosine); }
int orb_ref_fast_score(const uint8_t* center, int stride) { return Extractor::fast_score(center, stride); }

}  // extern "C"
