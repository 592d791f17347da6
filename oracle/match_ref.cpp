// ORACLE - TEST INFRASTRUCTURE ONLY.  Never linked, imported or called by the product path
// (se2lam_amd/, include/).  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline`
// leg may use it, and only as the checker / reported CPU baseline.
//
// PARITY UNPINNED for end-to-end match lists (the reference ships no fixtures); the integer parts
// ARE pinned from the tree: TH_LOW = 75, TH_HIGH = 100, HISTO_LENGTH = 30 (ORBmatcher.cpp:45-47), the 64x48 grid
// (Frame.h:26-27) and DescriptorDistance against a naive bit count (tests/test_match_oracle.py).
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
        pos[3 * i] = px; pos[3 * i + 1] = py; pos[3 * i + 2] = pz;
        if (pz >= lower && pz <= upper) {
            const float q0 = px - Ocam[0], q1 = py - Ocam[1], q2 = pz - Ocam[2];
            const double dot = (double)px * q0 + (double)py * q1 + (double)pz * q2;
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

}  // extern "C"
