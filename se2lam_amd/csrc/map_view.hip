// libse2gpu - Map::updateLocalGraph (/root/reference/src/Map.cpp:285-331) on a POD / CSR view of the map: SURVEY.md
// section 8 row a25.  The reference walks its pointer graph with std::set copies (one set<PtrKeyFrame> per key frame and
// hop, one set<PtrMapPoint> per key frame); here the covisibility graph and the observation lists are CSR arrays the
// caller keeps next to its map, and the window selection is three passes over flat arrays with mark vectors.
// Host code by nature (a few hundred vertices, pointer-chasing, sequential): it produces the index lists that
// se2gpu_ba_load_local_graph takes, so the whole localBA preparation runs on POD data.
#include <algorithm>
#include <vector>

#include "common.h"

using namespace se2gpu;

extern "C" {

int se2gpu_map_update_local_graph(const se2gpu_map_view* m, int current_kf, int search_level, int32_t* local_kfs,
                                  int* n_local, int32_t* ref_kfs, int* n_ref, int32_t* local_mps, int* n_mps) {
    SE2_REQUIRE(m && n_local && n_ref && n_mps, SE2GPU_ERR_INVALID, "update_local_graph: NULL argument");
    const int K = m->n_kf, M = m->n_mp;
    SE2_REQUIRE(K > 0 && M >= 0 && current_kf >= 0 && current_kf < K, SE2GPU_ERR_INVALID,
                "update_local_graph: current key frame %d of %d", current_kf, K);
    SE2_REQUIRE(m->kf_id && m->covis_ptr && m->kf_mp_ptr && (M == 0 || (m->mp_id && m->mp_kf_ptr)), SE2GPU_ERR_INVALID,
                "update_local_graph: NULL array");
    auto csr_ok = [](const int32_t* ptr, int n) {   // starts at 0, never decreases
        if (ptr[0] != 0) return false;
        for (int i = 0; i < n; ++i)
            if (ptr[i + 1] < ptr[i]) return false;
        return true;
    };
    SE2_REQUIRE(csr_ok(m->covis_ptr, K) && csr_ok(m->kf_mp_ptr, K) && (M == 0 || csr_ok(m->mp_kf_ptr, M)), SE2GPU_ERR_INVALID,
                "update_local_graph: a CSR pointer array does not start at 0 or decreases");
    SE2_REQUIRE((m->covis_ptr[K] == 0 || m->covis_idx) && (m->kf_mp_ptr[K] == 0 || m->kf_mp_idx) &&
                (M == 0 || m->mp_kf_ptr[M] == 0 || m->mp_kf_idx), SE2GPU_ERR_INVALID, "update_local_graph: NULL index array");
    // setLocalKFs: mCurrentKF and everything within `searchLevel` covisibility hops (Map.cpp:298-308); a pass expands
    // the members of the previous pass only - members found earlier have been expanded already
    std::vector<uint8_t> is_local(K, 0), is_mp(M, 0), is_ref(K, 0);
    std::vector<int> frontier{current_kf}, next, locals{current_kf};
    is_local[current_kf] = 1;
    for (int lv = 0; lv < search_level && !frontier.empty(); ++lv) {
        next.clear();
        for (int a : frontier)
            for (int t = m->covis_ptr[a]; t < m->covis_ptr[a + 1]; ++t) {
                const int b = m->covis_idx[t];
                SE2_REQUIRE(b >= 0 && b < K, SE2GPU_ERR_INVALID, "covisibility entry %d out of range", b);
                if (!is_local[b]) { is_local[b] = 1; next.push_back(b); locals.push_back(b); }
            }
        frontier.swap(next);
    }
    // setLocalMPs: every map point a local key frame observes (getAllObsMPs(false), :310-315)
    std::vector<int> mps;
    for (int a : locals)
        for (int t = m->kf_mp_ptr[a]; t < m->kf_mp_ptr[a + 1]; ++t) {
            const int p = m->kf_mp_idx[t];
            SE2_REQUIRE(p >= 0 && p < M, SE2GPU_ERR_INVALID, "observation entry %d out of range", p);
            if (!is_mp[p]) { is_mp[p] = 1; mps.push_back(p); }
        }
    // setRefKFs: the other key frames that observe a local map point (:317-326)
    std::vector<int> refs;
    for (int p : mps)
        for (int t = m->mp_kf_ptr[p]; t < m->mp_kf_ptr[p + 1]; ++t) {
            const int b = m->mp_kf_idx[t];
            SE2_REQUIRE(b >= 0 && b < K, SE2GPU_ERR_INVALID, "observer entry %d out of range", b);
            if (!is_local[b] && !is_ref[b]) { is_ref[b] = 1; refs.push_back(b); }
        }
    // the reference's sets are ordered by id (KeyFrame::IdLessThan, MapPoint::IdLessThan): so are the output vectors
    auto by_kf_id = [&](int a, int b) { return m->kf_id[a] < m->kf_id[b] || (m->kf_id[a] == m->kf_id[b] && a < b); };
    std::sort(locals.begin(), locals.end(), by_kf_id);
    std::sort(refs.begin(), refs.end(), by_kf_id);
    std::sort(mps.begin(), mps.end(), [&](int a, int b) { return m->mp_id[a] < m->mp_id[b] || (m->mp_id[a] == m->mp_id[b] && a < b); });
    *n_local = (int)locals.size(); *n_ref = (int)refs.size(); *n_mps = (int)mps.size();
    if (local_kfs) std::copy(locals.begin(), locals.end(), local_kfs);
    if (ref_kfs) std::copy(refs.begin(), refs.end(), ref_kfs);
    if (local_mps) std::copy(mps.begin(), mps.end(), local_mps);
    return SE2GPU_OK;
}

}  // extern "C"
