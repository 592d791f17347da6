// Map::updateLocalGraph() (/root/reference/src/Map.cpp:285-331) and Map::loadLocalGraph(SlamOptimizer&) (Map.cpp:891-1022)
// on flat views of the map, for the LocalMapper thread that owns the pointer graph.  The reference walks
// std::set<PtrKeyFrame> / std::find per observation; a caller of this header walks its sets ONCE per key frame, appends
// plain numbers to the two builders below, and the library does the rest (window search on the host, vertex numbering,
// fixed rule, covariance inversion and the per-observation information matrices on the device).
#pragma once
#include <cstdint>
#include <vector>

#include "../se2gpu.h"
#include "optimizer.h"
#include "preintegration.h"
#include "types.h"

namespace se2lam_amd {

// CSR view of the whole map for updateLocalGraph: positions are indices into the order key frames / map points were added
class MapView {
public:
    int addKeyFrame(int id) { kf_id_.push_back(id); covis_.emplace_back(); kf_mp_.emplace_back(); return (int)kf_id_.size() - 1; }
    int addMapPoint(int id) { mp_id_.push_back(id); mp_kf_.emplace_back(); return (int)mp_id_.size() - 1; }
    void addCovisibility(int kf, int other) { covis_[kf].push_back(other); }          // KeyFrame::getAllCovisibleKFs()
    void addObservation(int kf, int mp) { kf_mp_[kf].push_back(mp); mp_kf_[mp].push_back(kf); }   // getAllObsMPs / getObservations
    // Map::updateLocalGraph() with mCurrentKF = position `currentKF`: fills mLocalGraphKFs, mRefKFs, mLocalGraphMPs (positions)
    void updateLocalGraph(int currentKF, std::vector<int32_t>& localKFs, std::vector<int32_t>& refKFs,
                          std::vector<int32_t>& localMPs, int searchLevel = 3) const {
        std::vector<int32_t> cp, ci, kp, ki, mp, mi;
        csr(covis_, cp, ci);
        csr(kf_mp_, kp, ki);
        csr(mp_kf_, mp, mi);
        se2gpu_map_view v{};
        v.n_kf = (int32_t)kf_id_.size();
        v.n_mp = (int32_t)mp_id_.size();
        v.kf_id = kf_id_.data();
        v.covis_ptr = cp.data(); v.covis_idx = ci.data();
        v.kf_mp_ptr = kp.data(); v.kf_mp_idx = ki.data();
        v.mp_id = mp_id_.data();
        v.mp_kf_ptr = mp.data(); v.mp_kf_idx = mi.data();
        localKFs.resize(v.n_kf); refKFs.resize(v.n_kf); localMPs.resize(v.n_mp);
        int nl = 0, nr = 0, nm = 0;
        check(se2gpu_map_update_local_graph(&v, currentKF, searchLevel, localKFs.data(), &nl, refKFs.data(), &nr,
                                            localMPs.data(), &nm), "Map::updateLocalGraph");
        localKFs.resize(nl); refKFs.resize(nr); localMPs.resize(nm);
    }

private:
    static void csr(const std::vector<std::vector<int32_t> >& rows, std::vector<int32_t>& ptr, std::vector<int32_t>& idx) {
        ptr.assign(rows.size() + 1, 0);
        for (size_t i = 0; i < rows.size(); ++i) ptr[i + 1] = ptr[i] + (int32_t)rows[i].size();
        idx.reserve(ptr.back());
        for (const auto& r : rows) idx.insert(idx.end(), r.begin(), r.end());
        if (idx.empty()) idx.push_back(0);      // never hand the C ABI a null pointer
    }
    std::vector<int32_t> kf_id_, mp_id_;
    std::vector<std::vector<int32_t> > covis_, kf_mp_, mp_kf_;
};

// The local window as the numbers Map::loadLocalGraph reads from it.  Order of calls = the reference's order:
// every key frame of mLocalGraphKFs, then every key frame of mRefKFs, then the map points each followed by its observations.
class LocalGraph {
public:
    // Config::Kcam, Config::bTc, Config::TH_HUBER, Config::PLANEMOTION_XROT_INFO / _Z_INFO
    template <typename MatK, typename MatT>
    LocalGraph(const MatK& Kcam, const MatT& bTc, float thHuber, float xrotInfo = 1e6f, float zInfo = 1.f) {
        g_.fx = Kcam.template at<float>(0, 0);
        g_.cx = Kcam.template at<float>(0, 2);
        g_.cy = Kcam.template at<float>(1, 2);
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) g_.Rbc[3 * r + c] = bTc.template at<float>(r, c);
            g_.tbc[r] = bTc.template at<float>(r, 3);
        }
        g_.huber_delta = thHuber;
        g_.xrot_info = xrotInfo;
        g_.z_info = zInfo;
    }
    // a key frame of mLocalGraphKFs: KeyFrame::id, Twb, Tcw (4x4 CV_32F); returns its position
    template <typename MatT>
    int addLocalKF(int id, const Se2f& Twb, const MatT& Tcw) {
        if (n_ref_) throw std::runtime_error("LocalGraph: local key frames come before reference key frames");
        ++n_local_;
        odo_to_.push_back(-1);
        odo_meas_.insert(odo_meas_.end(), 3, 0.0);
        odo_cov_.insert(odo_cov_.end(), 9, 0.0);
        return pushKF(id, Twb, Tcw);
    }
    template <typename MatT>
    int addRefKF(int id, const Se2f& Twb, const MatT& Tcw) { ++n_ref_; return pushKF(id, Twb, Tcw); }
    // KeyFrame::preOdomFromSelf of local key frame `from` when its target `to` is in mLocalGraphKFs (Map.cpp:928-943)
    void setOdometry(int from, int to, const PreSE2& pre) {
        odo_to_[from] = to;
        for (int i = 0; i < 3; ++i) odo_meas_[3 * from + i] = pre.meas[i];
        for (int i = 0; i < 9; ++i) odo_cov_[9 * from + i] = pre.cov[i];
    }
    int addMapPoint(float x, float y, float z) { mp_pos_.insert(mp_pos_.end(), {x, y, z}); return (int)mp_pos_.size() / 3 - 1; }
    // one observation of the map point added last: observing key frame (position; -1 = in neither list), keyPointsUn[ftrIdx].pt,
    // mViewMPs[ftrIdx], mvLevelSigma2[octave]   (Map.cpp:990-1019)
    void addObservation(int kf, float u, float v, float lx, float ly, float lz, float sigma2) {
        obs_mp_.push_back((int32_t)mp_pos_.size() / 3 - 1);
        obs_kf_.push_back(kf);
        obs_uv_.insert(obs_uv_.end(), {u, v});
        obs_lc_.insert(obs_lc_.end(), {lx, ly, lz});
        obs_sigma2_.push_back(sigma2);
    }
    // Map::loadLocalGraph(optimizer): vertex ids afterwards are local key frame i -> i, reference key frame i -> nLocal + i,
    // map point i -> nLocal + nRef + 1 + i (Map.cpp:906, 956, 975)
    void load(SlamOptimizer& optimizer) {
        g_.n_local_kf = n_local_;
        g_.n_ref_kf = n_ref_;
        g_.n_mp = (int32_t)mp_pos_.size() / 3;
        g_.n_obs = (int32_t)obs_mp_.size();
        g_.kf_id = kf_id_.data();
        g_.kf_Twb = kf_Twb_.data();
        g_.kf_Rcw = kf_Rcw_.data();
        g_.odo_to = odo_to_.data();
        g_.odo_meas = odo_meas_.data();
        g_.odo_cov = odo_cov_.data();
        g_.mp_pos = mp_pos_.data();
        g_.obs_mp = obs_mp_.data();
        g_.obs_kf = obs_kf_.data();
        g_.obs_uv = obs_uv_.data();
        g_.obs_lc = obs_lc_.data();
        g_.obs_sigma2 = obs_sigma2_.data();
        check(se2gpu_ba_load_local_graph(optimizer.handle(), &g_), "Map::loadLocalGraph");
    }
    int maxKFid() const { return n_local_ + n_ref_ + 1; }                    // Map.cpp:975
    int vertexIdMP(int mp) const { return maxKFid() + mp; }

private:
    template <typename MatT>
    int pushKF(int id, const Se2f& Twb, const MatT& Tcw) {
        kf_id_.push_back(id);
        kf_Twb_.insert(kf_Twb_.end(), {Twb.x, Twb.y, Twb.theta});
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) kf_Rcw_.push_back(Tcw.template at<float>(r, c));
        return (int)kf_id_.size() - 1;
    }
    se2gpu_local_graph g_{};
    int n_local_ = 0, n_ref_ = 0;
    std::vector<int32_t> kf_id_, odo_to_, obs_mp_, obs_kf_;
    std::vector<float> kf_Twb_, kf_Rcw_, mp_pos_, obs_uv_, obs_lc_, obs_sigma2_;
    std::vector<double> odo_meas_, odo_cov_;
};

}  // namespace se2lam_amd
