// ORACLE - TEST INFRASTRUCTURE ONLY.  Never linked, imported or called by the product path
// (se2lam_amd/, include/).  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline`
// leg may use it, and only as the checker / reported CPU baseline.
//
// PARITY: the matchers and the frame grid are pinned against the REFERENCE'S OWN CODE - oracle/_ref compiles
// /root/reference/src/ORBmatcher.cpp, Frame.cpp and cvutil.cpp unmodified (`make -C oracle ref`), and
// tests/test_ref_compiled.py + tools/fuzz_ref.py compare match lists, counts and the updated vbPrevMatched exactly
// (MatchByWindow, MatchByProjection incl. cvu::se3map / camprjc, SearchByBoW over the reference's DBoW2::FeatureVector,
// GetFeaturesInArea, ComputeThreeMaxima, DescriptorDistance).  The reference ships no fixtures; also pinned
// from the tree: TH_LOW = 75, TH_HIGH = 100, HISTO_LENGTH = 30 (ORBmatcher.cpp:45-47), the 64x48 grid (Frame.h:26-27),
// DescriptorDistance against a naive bit count (tests/test_match_oracle.py).  findFundamentalMat and the SVD of
// cvu::triangulate are OpenCV arithmetic and stay unpinned.
//
// CPU restatement (single thread, no dependencies) of se2lam::ORBmatcher and the Frame grid it searches:
//   DescriptorDistance      /root/reference/src/ORBmatcher.cpp:110-126   (SWAR popcount over 8 x 32 bit)
//   ComputeThreeMaxima      :64-105
//   MatchByWindow           :278-381
//   MatchByProjection       :383-454      (+ cvu::se3map / cvu::camprjc float arithmetic, src/cvutil.cpp:89-106)
//   SearchByBoW             :128-276      (DBoW2 feature vectors passed as CSR; the vocabulary itself is host code)
//   Frame::PosInGrid        /root/reference/src/Frame.cpp:209-219   (NOTE: round(), not floor)
//   Frame::GetFeaturesInArea :222-286     (cell range by floor/ceil, level filter, square |dx|,|dy| <= r test,
//                                          results in (cell x, cell y, insertion) order)
//   grid element sizes      :37-44, FRAME_GRID_COLS = 64, FRAME_GRID_ROWS = 48 (include/se2lam/Frame.h:26-27)
// Build: g++ -O2 -ffp-contract=off (oracle/Makefile).
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

extern "C" {
struct match_ref_keypoint {  // cv::KeyPoint layout
    float x, y, size, angle, response;
    int32_t octave, class_id;
};
struct match_ref_bounds {
    float min_x, min_y, max_x, max_y;
};
}

namespace {

const int TH_HIGH = 100, TH_LOW = 75, HISTO_LENGTH = 30;
const int GRID_ROWS = 48, GRID_COLS = 64;

int descriptor_distance(const uint8_t* a, const uint8_t* b) {
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb;
        std::memcpy(&pa, a + 4 * i, 4);
        std::memcpy(&pb, b + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

struct Grid {
    match_ref_bounds b;
    float wInv, hInv;
    const match_ref_keypoint* kps;
    int n;
    std::vector<std::vector<std::vector<int>>> cells;  // [ix][iy]

    Grid(const match_ref_bounds& bounds, const match_ref_keypoint* k, int n_) : b(bounds), kps(k), n(n_) {
        wInv = (float)GRID_COLS / (b.max_x - b.min_x);
        hInv = (float)GRID_ROWS / (b.max_y - b.min_y);
        cells.assign(GRID_COLS, std::vector<std::vector<int>>(GRID_ROWS));
        for (int i = 0; i < n; i++) {
            const int px = (int)std::round((kps[i].x - b.min_x) * wInv);
            const int py = (int)std::round((kps[i].y - b.min_y) * hInv);
            if (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) continue;
            cells[px][py].push_back(i);
        }
    }

    void features_in_area(float x, float y, float r, int minLevel, int maxLevel, std::vector<int>& out) const {
        out.clear();
        int nMinCellX = (int)std::floor((x - b.min_x - r) * wInv);
        nMinCellX = std::max(0, nMinCellX);
        if (nMinCellX >= GRID_COLS) return;
        int nMaxCellX = (int)std::ceil((x - b.min_x + r) * wInv);
        nMaxCellX = std::min(GRID_COLS - 1, nMaxCellX);
        if (nMaxCellX < 0) return;
        int nMinCellY = (int)std::floor((y - b.min_y - r) * hInv);
        nMinCellY = std::max(0, nMinCellY);
        if (nMinCellY >= GRID_ROWS) return;
        int nMaxCellY = (int)std::ceil((y - b.min_y + r) * hInv);
        nMaxCellY = std::min(GRID_ROWS - 1, nMaxCellY);
        if (nMaxCellY < 0) return;
        bool bCheckLevels = true, bSameLevel = false;
        if (minLevel == -1 && maxLevel == -1) bCheckLevels = false;
        else if (minLevel == maxLevel) bSameLevel = true;
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
            for (int iy = nMinCellY; iy <= nMaxCellY; iy++)
                for (int idx : cells[ix][iy]) {
                    const match_ref_keypoint& kp = kps[idx];
                    if (bCheckLevels && !bSameLevel) {
                        if (kp.octave < minLevel || kp.octave > maxLevel) continue;
                    } else if (bSameLevel) {
                        if (kp.octave != minLevel) continue;
                    }
                    if (std::abs(kp.x - x) > r || std::abs(kp.y - y) > r) continue;
                    out.push_back(idx);
                }
    }
};

void three_maxima(const std::vector<int>* histo, int L, int& ind1, int& ind2, int& ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = (int)histo[i].size();
        if (s > max1) {
            max3 = max2; max2 = max1; max1 = s;
            ind3 = ind2; ind2 = ind1; ind1 = i;
        } else if (s > max2) {
            max3 = max2; max2 = s;
            ind3 = ind2; ind2 = i;
        } else if (s > max3) {
            max3 = s;
            ind3 = i;
        }
    }
    if (max2 < 0.1f * (float)max1) {
        ind2 = -1;
        ind3 = -1;
    } else if (max3 < 0.1f * (float)max1) {
        ind3 = -1;
    }
}

}  // namespace

extern "C" {

int match_ref_hamming(const uint8_t* a, const uint8_t* b) { return descriptor_distance(a, b); }

int match_ref_features_in_area(const match_ref_bounds* bounds, const match_ref_keypoint* kps, int n, float x, float y,
                               float r, int minLevel, int maxLevel, int32_t* out, int cap) {
    Grid g(*bounds, kps, n);
    std::vector<int> v;
    g.features_in_area(x, y, r, minLevel, maxLevel, v);
    for (size_t i = 0; i < v.size() && (int)i < cap; ++i) out[i] = v[i];
    return (int)v.size();
}

// MatchByWindow(frame1, frame2, vbPrevMatched, winSize, vnMatches12, levelOffset, minLevel, maxLevel)
int match_ref_window(const match_ref_bounds* bounds, const match_ref_keypoint* kps1, const uint8_t* desc1, int n1,
                     const match_ref_keypoint* kps2, const uint8_t* desc2, int n2, float* prev_xy, int winSize,
                     int levelOffset, int minLevel, int maxLevel, float nnratio, int32_t* vnMatches12) {
    int nmatches = 0;
    for (int i = 0; i < n1; ++i) vnMatches12[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = (float)HISTO_LENGTH / 360.0f;
    std::vector<int> vMatchesDistance(n2, INT_MAX), vnMatches21(n2, -1), vIndices2;
    Grid grid(*bounds, kps2, n2);
    for (int i1 = 0; i1 < n1; i1++) {
        const match_ref_keypoint& kp1 = kps1[i1];
        const int level1 = kp1.octave;
        if (level1 > maxLevel || level1 < minLevel) continue;
        const int minLevel2 = level1 - levelOffset > 0 ? level1 - levelOffset : 0;
        grid.features_in_area(prev_xy[2 * i1], prev_xy[2 * i1 + 1], (float)winSize, minLevel2, level1 + levelOffset,
                              vIndices2);
        if (vIndices2.empty()) continue;
        const uint8_t* d1 = desc1 + 32 * (size_t)i1;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int i2 : vIndices2) {
            const int dist = descriptor_distance(d1, desc2 + 32 * (size_t)i2);
            if (vMatchesDistance[i2] <= dist) continue;
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestIdx2 = i2;
            } else if (dist < bestDist2) {
                bestDist2 = dist;
            }
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * nnratio) {
                if (vnMatches21[bestIdx2] >= 0) {
                    vnMatches12[vnMatches21[bestIdx2]] = -1;
                    nmatches--;
                }
                vnMatches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchesDistance[bestIdx2] = bestDist;
                nmatches++;
                float rot = kps1[i1].angle - kps2[bestIdx2].angle;
                if (rot < 0.0) rot += 360.f;
                int bin = (int)std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(i1);
            }
        }
    }
    {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx1 : rotHist[i])
                if (vnMatches12[idx1] >= 0) {
                    vnMatches12[idx1] = -1;
                    nmatches--;
                }
        }
    }
    for (int i1 = 0; i1 < n1; i1++)
        if (vnMatches12[i1] >= 0) {
            prev_xy[2 * i1] = kps2[vnMatches12[i1]].x;
            prev_xy[2 * i1 + 1] = kps2[vnMatches12[i1]].y;
        }
    return nmatches;
}

// MatchByProjection(pNewKF, localMPs, winSize, levelOffset, vMatchesIdxMP) with ORBmatcher(nnratio)
//   mp_skip[i] != 0  <=>  the reference `continue`s at ORBmatcher.cpp:392-395 (null / bad parallax / already observed)
//   kf_observed[idx] <=>  pNewKF->hasObservation(idx) (:417)
int match_ref_projection(const match_ref_bounds* bounds, const float* mp_pos, const uint8_t* mp_desc,
                         const int32_t* mp_octave, const uint8_t* mp_skip, int m, const float* Tcw, float fx, float fy,
                         float cx, float cy, const match_ref_keypoint* kps, const uint8_t* desc,
                         const uint8_t* kf_observed, int n, int winSize, int levelOffset, float nnratio,
                         int32_t* vMatchesIdxMP) {
    int nmatches = 0;
    for (int i = 0; i < n; ++i) vMatchesIdxMP[i] = -1;
    std::vector<int> vMatchesDistance(n, INT_MAX), vNear;
    Grid grid(*bounds, kps, n);
    for (int i = 0; i < m; i++) {
        if (mp_skip[i]) continue;
        // cvu::se3map: R*pt + t with Matx33f * Point3f (s = 0; s += a(i,k)*b(k)) then + t
        const float X = mp_pos[3 * i], Y = mp_pos[3 * i + 1], Z = mp_pos[3 * i + 2];
        float pc[3];
        for (int r = 0; r < 3; ++r) {
            float s = 0;
            s += Tcw[4 * r + 0] * X;
            s += Tcw[4 * r + 1] * Y;
            s += Tcw[4 * r + 2] * Z;
            pc[r] = s + Tcw[4 * r + 3];
        }
        // cvu::camprjc: uvw = Matx33f(K) * pt, K = [fx 0 cx; 0 fy cy; 0 0 1]
        float u = 0, v = 0, w = 0;
        u += fx * pc[0]; u += 0.f * pc[1]; u += cx * pc[2];
        v += 0.f * pc[0]; v += fy * pc[1]; v += cy * pc[2];
        w += 0.f * pc[0]; w += 0.f * pc[1]; w += 1.f * pc[2];
        const float px = u / w, py = v / w;
        if (!(px >= bounds->min_x && px <= bounds->max_x && py >= bounds->min_y && py <= bounds->max_y)) continue;
        const int predictLevel = mp_octave[i];
        const int levelWinSize = predictLevel * winSize;
        const int minLevel = predictLevel > levelOffset ? predictLevel - levelOffset : 0;
        grid.features_in_area(px, py, (float)levelWinSize, minLevel, predictLevel + levelOffset, vNear);
        if (vNear.empty()) continue;
        int bestDist = INT_MAX, bestLevel = -1, bestDist2 = INT_MAX, bestLevel2 = -1, bestIdx = -1;
        for (int idx : vNear) {
            if (kf_observed[idx]) continue;
            const int dist = descriptor_distance(mp_desc + 32 * (size_t)i, desc + 32 * (size_t)idx);
            if (vMatchesDistance[idx] <= dist) continue;
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestLevel2 = bestLevel;
                bestLevel = kps[idx].octave;
                bestIdx = idx;
            } else if (dist < bestDist2) {
                bestLevel2 = kps[idx].octave;
                bestDist2 = dist;
            }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            if (vMatchesIdxMP[bestIdx] >= 0) {
                vMatchesIdxMP[bestIdx] = -1;
                nmatches--;
            }
            vMatchesIdxMP[bestIdx] = i;
            vMatchesDistance[bestIdx] = bestDist;
            nmatches++;
        }
    }
    return nmatches;
}

// SearchByBoW (ORBmatcher.cpp:128-276); feature vectors as CSR with ascending node ids
int match_ref_search_by_bow(const match_ref_keypoint* kps1, const uint8_t* desc1, int n1, const int32_t* fv1_nodes,
                            const int32_t* fv1_ptr, const int32_t* fv1_idx, int nn1, const uint8_t* has_mp1,
                            const match_ref_keypoint* kps2, const uint8_t* desc2, int n2, const int32_t* fv2_nodes,
                            const int32_t* fv2_ptr, const int32_t* fv2_idx, int nn2, const uint8_t* has_mp2, int mp_only,
                            float nnratio, int check_orientation, int32_t* matches12) {
    for (int i = 0; i < n1; ++i) matches12[i] = -1;
    std::vector<char> vbMatched2(n2, 0);
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = (float)HISTO_LENGTH / 360.0f;
    int nmatches = 0;
    int a = 0, b = 0;
    while (a < nn1 && b < nn2) {
        if (fv1_nodes[a] == fv2_nodes[b]) {
            for (int i1 = fv1_ptr[a]; i1 < fv1_ptr[a + 1]; i1++) {
                const int idx1 = fv1_idx[i1];
                if (mp_only && !has_mp1[idx1]) continue;
                const uint8_t* d1 = desc1 + 32 * (size_t)idx1;
                int bestDist1 = INT_MAX, bestIdx2 = -1, bestDist2 = INT_MAX;
                for (int i2 = fv2_ptr[b]; i2 < fv2_ptr[b + 1]; i2++) {
                    const int idx2 = fv2_idx[i2];
                    if (mp_only && !has_mp2[idx2]) continue;
                    if (vbMatched2[idx2]) continue;
                    const int dist = descriptor_distance(d1, desc2 + 32 * (size_t)idx2);
                    if (dist < bestDist1) {
                        bestDist2 = bestDist1;
                        bestDist1 = dist;
                        bestIdx2 = idx2;
                    } else if (dist < bestDist2) {
                        bestDist2 = dist;
                    }
                }
                if (bestDist1 < TH_LOW) {
                    if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
                        matches12[idx1] = bestIdx2;
                        vbMatched2[bestIdx2] = 1;
                        if (check_orientation) {
                            float rot = kps1[idx1].angle - kps2[bestIdx2].angle;
                            if (rot < 0.0) rot += 360.0f;
                            int bin = (int)std::round(rot * factor);
                            if (bin == HISTO_LENGTH) bin = 0;
                            rotHist[bin].push_back(idx1);
                        }
                        nmatches++;
                    }
                }
            }
            a++;
            b++;
        } else if (fv1_nodes[a] < fv2_nodes[b]) {
            while (a < nn1 && fv1_nodes[a] < fv2_nodes[b]) a++;   // lower_bound
        } else {
            while (b < nn2 && fv2_nodes[b] < fv1_nodes[a]) b++;
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx1 : rotHist[i]) {
                matches12[idx1] = -1;
                nmatches--;
            }
        }
    }
    return nmatches;
}

// Track::doTriangulate for all matches of a frame pair (/root/reference/src/Track.cpp:378-419), with
// cvu::triangulate (src/cvutil.cpp:46-59), Config::acceptDepth (src/Config.cpp:188-190) and cvu::checkParallax
// (src/cvutil.cpp:92-98).  cv::SVD::compute is OpenCV's one-sided Jacobi in FP32 (third-party, not in the tree); the
// same Hestenes iteration is run here in FP64 on the FP32 system matrix.  PARITY UNPINNED against OpenCV's own float
// iteration (expected agreement ~1e-5 relative on well-conditioned points).
static void smallest_right_singular_vector(const double a_in[16], double v4[4]) {
    double At[4][4], Vt[4][4], W[4];
    for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 4; ++k) {
            At[i][k] = a_in[k * 4 + i];
            Vt[i][k] = i == k ? 1.0 : 0.0;
        }
    for (int i = 0; i < 4; ++i) {
        double sd = 0;
        for (int k = 0; k < 4; ++k) sd += At[i][k] * At[i][k];
        W[i] = sd;
    }
    const double eps = 2.220446049250313e-16 * 10;
    for (int iter = 0; iter < 30; ++iter) {
        bool changed = false;
        for (int i = 0; i < 3; ++i)
            for (int j = i + 1; j < 4; ++j) {
                double a = W[i], p = 0, b = W[j];
                for (int k = 0; k < 4; ++k) p += At[i][k] * At[j][k];
                if (std::fabs(p) <= eps * std::sqrt(a * b)) continue;
                p *= 2;
                const double beta = a - b, gamma = std::sqrt(p * p + beta * beta);
                double c, sn;
                if (beta < 0) {
                    const double delta = (gamma - beta) * 0.5;
                    sn = std::sqrt(delta / gamma);
                    c = p / (gamma * sn * 2);
                } else {
                    c = std::sqrt((gamma + beta) / (gamma * 2));
                    sn = p / (gamma * c * 2);
                }
                a = 0;
                b = 0;
                for (int k = 0; k < 4; ++k) {
                    const double t0 = c * At[i][k] + sn * At[j][k];
                    const double t1 = -sn * At[i][k] + c * At[j][k];
                    At[i][k] = t0;
                    At[j][k] = t1;
                    a += t0 * t0;
                    b += t1 * t1;
                }
                W[i] = a;
                W[j] = b;
                changed = true;
                for (int k = 0; k < 4; ++k) {
                    const double t0 = c * Vt[i][k] + sn * Vt[j][k];
                    const double t1 = -sn * Vt[i][k] + c * Vt[j][k];
                    Vt[i][k] = t0;
                    Vt[j][k] = t1;
                }
            }
        if (!changed) break;
    }
    int m = 0;
    for (int i = 1; i < 4; ++i)
        if (W[i] < W[m]) m = i;
    for (int k = 0; k < 4; ++k) v4[k] = Vt[m][k];
}

int match_ref_triangulate(int n, const match_ref_keypoint* kps_ref, const match_ref_keypoint* kps_cur, int n_cur,
                          int32_t* match_idx, const uint8_t* has_obs, const float* P1, const float* P2,
                          const float* Ocam, float lower, float upper, int min_degree, float* pos, uint8_t* good,
                          int* n_tracked_old) {
    const float minCos[4] = {0.9998f, 0.9994f, 0.9986f, 0.9976f};
    int ngood = 0, nold = 0;
    for (int i = 0; i < n; ++i) {
        good[i] = 0;
        pos[3 * i] = pos[3 * i + 1] = pos[3 * i + 2] = 0.f;
        const int mi = match_idx[i];
        if (mi < 0 || mi >= n_cur) continue;
        if (has_obs && has_obs[i]) { nold++; continue; }
        const float x1 = kps_ref[i].x, y1 = kps_ref[i].y, x2 = kps_cur[mi].x, y2 = kps_cur[mi].y;
        double A[16];
        for (int c = 0; c < 4; ++c) {
            A[0 + c] = (double)(x1 * P1[8 + c] - P1[0 + c]);
            A[4 + c] = (double)(y1 * P1[8 + c] - P1[4 + c]);
            A[8 + c] = (double)(x2 * P2[8 + c] - P2[0 + c]);
            A[12 + c] = (double)(y2 * P2[8 + c] - P2[4 + c]);
        }
        double v[4];
        smallest_right_singular_vector(A, v);
        const float w = (float)v[3];
        const float px = (float)v[0] / w, py = (float)v[1] / w, pz = (float)v[2] / w;
        if (pz >= lower && pz <= upper) {
            pos[3 * i] = px; pos[3 * i + 1] = py; pos[3 * i + 2] = pz;   // Track.cpp:407-408: only for an accepted depth
            const float q0 = px - Ocam[0], q1 = py - Ocam[1], q2 = pz - Ocam[2];
            const float dotf = px * q0 + py * q1 + pz * q2;   // cv::Point3_<float>::dot: float arithmetic (cvutil.cpp:96)
            const double dot = (double)dotf;
            const double n1 = std::sqrt((double)px * px + (double)py * py + (double)pz * pz);
            const double n2 = std::sqrt((double)q0 * q0 + (double)q1 * q1 + (double)q2 * q2);
            const float cosp = (float)(std::fabs(dot) / (n1 * n2));
            if (cosp < minCos[min_degree - 1]) { good[i] = 1; ngood++; }
        } else {
            match_idx[i] = -1;
        }
    }
    if (n_tracked_old) *n_tracked_old = nold;
    return ngood;
}

// ---------------------------------------------------------------------------------------------
// Track::removeOutliers (src/Track.cpp:308-344) = cv::findFundamentalMat(pt1, pt2, mask) with the defaults
// FM_RANSAC, param1 = 3, param2 = 0.99 [3P: OpenCV 3.2 modules/calib3d/src/fundam.cpp + ptsetreg.cpp, restated from
// memory - parity with the real library is UNPINNED]:
//   npoints < 7          -> no mask                                 (=> every match is discarded, Track.cpp:338-341)
//   npoints == 7         -> 7-point, mask = all ones
//   8 <= npoints < 15    -> LMedS  (7-point samples, 0.45 outlier ratio, median of the FP32 errors sorted AS INTS)
//   npoints >= 15        -> RANSAC (7-point samples, cv::RNG(-1), <= 1000 iterations, adaptive stop)
// run7Point needs a basis of the 2-d null space of the 7x9 system.  cv::SVD's FULL_UV completes V with seeded random
// vectors, i.e. an arbitrary basis; the fundamental matrices are the det = 0 members of the pencil and do not depend on
// the basis, so a Householder QR of A^T supplies it here.  cv::solveCubic's acos/cos/cubeRoot are replaced by Newton
// iterations built from + - * / sqrt only (same branch structure and root order), so that every platform executing this
// operation sequence gets the same bits.
// ---------------------------------------------------------------------------------------------
}  // extern "C"

namespace {

struct CvRng {  // cv::RNG: multiply-with-carry
    uint64_t state;
    explicit CvRng(uint64_t s) : state(s ? s : 0xffffffffull) {}
    unsigned next() {
        state = (uint64_t)(unsigned)state * 4164903690u + (unsigned)(state >> 32);
        return (unsigned)state;
    }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

// PointSetRegistrator::getSubset with Callback::checkSubset == true: 7 distinct indices by rejection
void fm_subset(CvRng& rng, int count, int idx[7]) {
    for (int i = 0; i < 7; ++i) {
        for (;;) {
            const int v = idx[i] = rng.uniform(0, count);
            int j = 0;
            for (; j < i; ++j)
                if (v == idx[j]) break;
            if (j == i) break;
        }
    }
}

double fm_cbrt(double v) {  // v >= 0
    if (!(v > 0)) return v;
    uint64_t bits;
    std::memcpy(&bits, &v, 8);
    const int ex = (int)((bits >> 52) & 0x7ff) - 1023;
    const int q = ex >= 0 ? ex / 3 : -((-ex + 2) / 3);
    const uint64_t yb = (uint64_t)(q + 1023) << 52;
    double y;
    std::memcpy(&y, &yb, 8);
    for (int it = 0; it < 12; ++it) y = (2.0 * y + v / (y * y)) / 3.0;
    return y;
}

// cos(acos(r) / 3): the root of 4c^3 - 3c = r in [1/2, 1], Newton from c = 1 (monotone)
double fm_cos_third(double r) {
    if (r > 1.0) r = 1.0;
    if (r < -1.0) r = -1.0;
    double c = 1.0;
    for (int it = 0; it < 64; ++it) {
        const double gp = 12.0 * c * c - 3.0;
        if (!(gp > 0)) break;
        const double g = (4.0 * c * c - 3.0) * c - r;
        const double cn = c - g / gp;
        if (cn == c) break;
        c = cn;
    }
    return c;
}

// cv::solveCubic for c[0] x^3 + c[1] x^2 + c[2] x + c[3]
int fm_solve_cubic(const double c[4], double x[3]) {
    double a0 = c[0], a1 = c[1], a2 = c[2], a3 = c[3];
    int n = 0;
    x[0] = x[1] = x[2] = 0;
    if (a0 == 0) {
        if (a1 == 0) {
            if (a2 == 0) n = a3 == 0 ? -1 : 0;
            else { x[0] = -a3 / a2; n = 1; }
        } else {
            double d = a2 * a2 - 4 * a1 * a3;
            if (d >= 0) {
                d = std::sqrt(d);
                const double q1 = (-a2 + d) * 0.5, q2 = (a2 + d) * -0.5;
                if (std::fabs(q1) > std::fabs(q2)) { x[0] = q1 / a1; x[1] = a3 / q1; }
                else { x[0] = q2 / a1; x[1] = a3 / q2; }
                n = d > 0 ? 2 : 1;
            }
        }
    } else {
        a0 = 1. / a0;
        a1 *= a0; a2 *= a0; a3 *= a0;
        const double Q = (a1 * a1 - 3 * a2) * (1. / 9);
        const double R = (2 * a1 * a1 * a1 - 9 * a1 * a2 + 27 * a3) * (1. / 54);
        const double Qcubed = Q * Q * Q;
        double d = Qcubed - R * R;
        if (d >= 0) {
            // theta = acos(R / sqrt(Q^3)); x_k = -2 sqrt(Q) cos(theta/3 + 2 k pi/3) - a1/3
            const double ct = fm_cos_third(R / std::sqrt(Qcubed));
            const double st = std::sqrt(1.0 - ct * ct);
            const double t0 = -2 * std::sqrt(Q), t2 = a1 * (1. / 3);
            const double h = 0.8660254037844386;  // sin(2 pi / 3)
            x[0] = t0 * ct - t2;
            x[1] = t0 * (-0.5 * ct - h * st) - t2;
            x[2] = t0 * (-0.5 * ct + h * st) - t2;
            n = 3;
        } else {
            d = std::sqrt(-d);
            double e = fm_cbrt(std::fabs(R) + d);
            if (R > 0) e = -e;
            x[0] = (e + Q / e) - a1 * (1. / 3);
            n = 1;
        }
    }
    return n;
}

// basis (f1, f2) of the null space of the 7x9 matrix a: last two columns of Q in the Householder QR of a^T
void fm_null_space(const double a[7][9], double f1[9], double f2[9]) {
    double M[9][7], V[7][9], beta[7];
    for (int r = 0; r < 9; ++r)
        for (int c = 0; c < 7; ++c) M[r][c] = a[c][r];
    for (int k = 0; k < 7; ++k) {
        double sigma = 0;
        for (int r = k; r < 9; ++r) sigma += M[r][k] * M[r][k];
        const double norm = std::sqrt(sigma);
        const double alpha = M[k][k] > 0 ? -norm : norm;
        for (int r = 0; r < 9; ++r) V[k][r] = r < k ? 0.0 : M[r][k];
        V[k][k] = M[k][k] - alpha;
        double vn = 0;
        for (int r = k; r < 9; ++r) vn += V[k][r] * V[k][r];
        beta[k] = vn > 0 ? 2.0 / vn : 0.0;
        for (int c = k + 1; c < 7; ++c) {
            double s = 0;
            for (int r = k; r < 9; ++r) s += V[k][r] * M[r][c];
            s *= beta[k];
            for (int r = k; r < 9; ++r) M[r][c] -= s * V[k][r];
        }
    }
    for (int j = 0; j < 2; ++j) {
        double q[9];
        for (int r = 0; r < 9; ++r) q[r] = r == 7 + j ? 1.0 : 0.0;
        for (int k = 6; k >= 0; --k) {
            double s = 0;
            for (int r = k; r < 9; ++r) s += V[k][r] * q[r];
            s *= beta[k];
            for (int r = k; r < 9; ++r) q[r] -= s * V[k][r];
        }
        for (int r = 0; r < 9; ++r) (j == 0 ? f1 : f2)[r] = q[r];
    }
}

// run7Point (fundam.cpp): up to three 3x3 matrices (row-major) from 7 correspondences
int fm_run7point(const float* m1, const float* m2, const int idx[7], double F[27]) {
    double a[7][9], f1[9], f2[9], c[4], r[3];
    for (int i = 0; i < 7; ++i) {
        const double x0 = m1[2 * idx[i]], y0 = m1[2 * idx[i] + 1];
        const double x1 = m2[2 * idx[i]], y1 = m2[2 * idx[i] + 1];
        a[i][0] = x1 * x0; a[i][1] = x1 * y0; a[i][2] = x1;
        a[i][3] = y1 * x0; a[i][4] = y1 * y0; a[i][5] = y1;
        a[i][6] = x0; a[i][7] = y0; a[i][8] = 1;
    }
    fm_null_space(a, f1, f2);
    for (int i = 0; i < 9; ++i) f1[i] -= f2[i];
    double t0 = f2[4] * f2[8] - f2[5] * f2[7];
    double t1 = f2[3] * f2[8] - f2[5] * f2[6];
    double t2 = f2[3] * f2[7] - f2[4] * f2[6];
    c[3] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2;
    c[2] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2 - f1[3] * (f2[1] * f2[8] - f2[2] * f2[7]) +
           f1[4] * (f2[0] * f2[8] - f2[2] * f2[6]) - f1[5] * (f2[0] * f2[7] - f2[1] * f2[6]) +
           f1[6] * (f2[1] * f2[5] - f2[2] * f2[4]) - f1[7] * (f2[0] * f2[5] - f2[2] * f2[3]) +
           f1[8] * (f2[0] * f2[4] - f2[1] * f2[3]);
    t0 = f1[4] * f1[8] - f1[5] * f1[7];
    t1 = f1[3] * f1[8] - f1[5] * f1[6];
    t2 = f1[3] * f1[7] - f1[4] * f1[6];
    c[0] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2;
    c[1] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2 - f2[3] * (f1[1] * f1[8] - f1[2] * f1[7]) +
           f2[4] * (f1[0] * f1[8] - f1[2] * f1[6]) - f2[5] * (f1[0] * f1[7] - f1[1] * f1[6]) +
           f2[6] * (f1[1] * f1[5] - f1[2] * f1[4]) - f2[7] * (f1[0] * f1[5] - f1[2] * f1[3]) +
           f2[8] * (f1[0] * f1[4] - f1[1] * f1[3]);
    const int n = fm_solve_cubic(c, r);
    if (n < 1 || n > 3) return n;
    for (int k = 0; k < n; ++k) {
        double* f = F + 9 * k;
        double lambda = r[k], mu = 1.;
        const double s = f1[8] * r[k] + f2[8];
        if (std::fabs(s) > 2.220446049250313e-16) {
            mu = 1. / s;
            lambda *= mu;
            f[8] = 1.;
        } else
            f[8] = 0.;
        for (int i = 0; i < 8; ++i) f[i] = f1[i] * lambda + f2[i] * mu;
    }
    return n;
}

// FMEstimatorCallback::computeError: symmetric squared epipolar distance (max of the two), FP32 result
float fm_error(const double F[9], float x1, float y1, float x2, float y2) {
    double a = F[0] * x1 + F[1] * y1 + F[2];
    double b = F[3] * x1 + F[4] * y1 + F[5];
    double c = F[6] * x1 + F[7] * y1 + F[8];
    const double s2 = 1. / (a * a + b * b);
    const double d2 = x2 * a + y2 * b + c;
    a = F[0] * x2 + F[3] * y2 + F[6];
    b = F[1] * x2 + F[4] * y2 + F[7];
    c = F[2] * x2 + F[5] * y2 + F[8];
    const double s1 = 1. / (a * a + b * b);
    const double d1 = x1 * a + y1 * b + c;
    const double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
    return (float)((e1 < e2) ? e2 : e1);  // std::max(e1, e2)
}

int fm_update_num_iters(double p, double ep, int modelPoints, int maxIters) {  // cv::RANSACUpdateNumIters
    p = std::fmax(p, 0.); p = std::fmin(p, 1.);
    ep = std::fmax(ep, 0.); ep = std::fmin(ep, 1.);
    double num = std::fmax(1. - p, 2.2250738585072014e-308);
    double denom = 1. - std::pow(1. - ep, modelPoints);
    if (denom < 2.2250738585072014e-308) return 0;
    num = std::log(num);
    denom = std::log(denom);
    return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : (int)std::nearbyint(num / denom);
}

int32_t fm_sort_key(float e) {  // LMedS sorts the float errors as ints; an x86 default NaN has the sign bit set
    int32_t k;
    if (e != e) return (int32_t)0xffc00000;
    std::memcpy(&k, &e, 4);
    return k;
}

}  // namespace

extern "C" {

// mask[i] (n entries) for the n correspondences m1[i] <-> m2[i] (x, y interleaved); returns the number of inliers,
// 0 with an all-zero mask when findFundamentalMat leaves the mask empty or finds no model
int match_ref_fundamental_mask(const float* m1, const float* m2, int n, uint8_t* mask) {
    for (int i = 0; i < n; ++i) mask[i] = 0;
    if (n < 7) return 0;
    if (n == 7) {
        for (int i = 0; i < n; ++i) mask[i] = 1;
        return n;
    }
    CvRng rng((uint64_t)-1);
    double F[27], best[9];
    int idx[7];
    if (n >= 15) {
        int niters = 1000, maxGood = 0;
        const float t = (float)(3.0 * 3.0);
        std::vector<uint8_t> cur(n);
        for (int iter = 0; iter < niters; ++iter) {
            fm_subset(rng, n, idx);
            const int nm = fm_run7point(m1, m2, idx, F);
            if (nm <= 0) continue;
            for (int k = 0; k < nm; ++k) {
                int good = 0;
                for (int i = 0; i < n; ++i) {
                    cur[i] = fm_error(F + 9 * k, m1[2 * i], m1[2 * i + 1], m2[2 * i], m2[2 * i + 1]) <= t;
                    good += cur[i];
                }
                if (good > std::max(maxGood, 6)) {
                    std::memcpy(mask, cur.data(), n);
                    maxGood = good;
                    niters = fm_update_num_iters(0.99, (double)(n - good) / n, 7, niters);
                }
            }
        }
        return maxGood;
    }
    // LMedS
    int niters = std::max(fm_update_num_iters(0.99, 0.45, 7, 1000), 3);
    double minMedian = 1.7976931348623157e308;
    std::vector<int32_t> keys(n);
    for (int iter = 0; iter < niters; ++iter) {
        fm_subset(rng, n, idx);
        const int nm = fm_run7point(m1, m2, idx, F);
        if (nm <= 0) continue;
        for (int k = 0; k < nm; ++k) {
            for (int i = 0; i < n; ++i)
                keys[i] = fm_sort_key(fm_error(F + 9 * k, m1[2 * i], m1[2 * i + 1], m2[2 * i], m2[2 * i + 1]));
            std::sort(keys.begin(), keys.end());
            float lo, hi;
            std::memcpy(&lo, &keys[n / 2 - 1], 4);
            std::memcpy(&hi, &keys[n / 2], 4);
            const double median = n % 2 != 0 ? (double)hi : (double)((lo + hi) * 0.5);
            if (median < minMedian) {
                minMedian = median;
                std::memcpy(best, F + 9 * k, sizeof(best));
            }
        }
    }
    if (!(minMedian < 1.7976931348623157e308)) return 0;
    double sigma = 2.5 * 1.4826 * (1 + 5. / (n - 7)) * std::sqrt(minMedian);
    sigma = std::fmax(sigma, 0.001);
    const float t = (float)(sigma * sigma);
    int good = 0;
    for (int i = 0; i < n; ++i) {
        mask[i] = fm_error(best, m1[2 * i], m1[2 * i + 1], m2[2 * i], m2[2 * i + 1]) <= t;
        good += mask[i];
    }
    return good;
}

// Track::removeOutliers (src/Track.cpp:308-344): matches[i] = -1 for the outliers; fewer than 10 inliers => every
// match is discarded and 0 is returned
int match_ref_remove_outliers(const match_ref_keypoint* kp1, int n1, const match_ref_keypoint* kp2, int n2,
                              int32_t* matches) {
    std::vector<float> pt1, pt2;
    std::vector<int> idx;
    for (int i = 0; i < n1; ++i) {
        if (matches[i] < 0) continue;
        idx.push_back(i);
        pt1.push_back(kp1[i].x); pt1.push_back(kp1[i].y);
        pt2.push_back(kp2[matches[i]].x); pt2.push_back(kp2[matches[i]].y);
    }
    (void)n2;
    const int n = (int)idx.size();
    std::vector<uint8_t> mask(std::max(n, 1));
    int nInlier = n ? match_ref_fundamental_mask(pt1.data(), pt2.data(), n, mask.data()) : 0;
    for (int i = 0; i < n; ++i)
        if (!mask[i]) matches[idx[i]] = -1;
    if (nInlier < 10) {
        nInlier = 0;
        for (int i = 0; i < n1; ++i) matches[i] = -1;
    }
    return nInlier;
}

// test hook: the 7-point models of one sample (run7Point), for checks against the definition (det F = 0, the seven
// correspondences on their epipolar lines)
int match_ref_seven_point(const float* m1, const float* m2, const int* idx7, double* F27) {
    return fm_run7point(m1, m2, idx7, F27);
}

// test hook: the first `count` samples PointSetRegistrator::getSubset draws for n points (7 indices each)
void match_ref_ransac_subsets(int n, int count, int* out) {
    CvRng rng((uint64_t)-1);
    for (int k = 0; k < count; ++k) fm_subset(rng, n, out + 7 * k);
}

}  // extern "C"
