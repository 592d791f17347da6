// ORBVocabulary (/root/reference/include/se2lam/ORBVocabulary.h: DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>,
// /root/reference/Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h) - the three calls se2lam makes:
//   loadFromBinaryFile(strVocFile)                        OdoSLAM.cpp:45          (TemplatedVocabulary.h:1478-1521)
//   transform(vCurrentDesc, mBowVec, mFeatVec, 4)         KeyFrame.cpp:251        (TemplatedVocabulary.h:1150-1216, 1241-1280)
//   score(BowVecCurr, BowVec)                             GlobalMapper.cpp:237, Localizer.cpp:360   (ScoringObject.cpp)
// Host code by design ("SearchByBoW with host DBoW2 transform"): a key frame's 1000 descriptors walk
// a k-ary tree of depth L once per key frame.  What it produces for the device is the FeatureVector in the CSR form that
// se2gpu_search_by_bow / ORBmatcher::SearchByBoW take (FeatureVectorCSR::view()).
//
// The tree lives in flat arrays; nothing of DBoW2, OpenCV or boost is needed.  Restated from the file format and the
// algorithm, not compiled from the reference's sources.  Faithful details:
//   * children keep the order in which their records appear in the file; the nearest child is the FIRST one at the minimal
//     Hamming distance (`d < best_d`, TemplatedVocabulary.h:1263-1272);
//   * a word with weight 0 ("stopped") contributes neither to the BowVector nor to the FeatureVector (:1182, :1203);
//   * TF_IDF / TF add the word's stored weight once per occurrence, IDF / BINARY once per word; L1, L2, chi-square, KL and
//     Bhattacharyya scoring normalise the vector (L1 norm, L2 for L2 scoring), dot-product scoring divides by the number of
//     words instead (:1190-1196) - only for TF_IDF / TF, as in the reference;
//   * the node recorded for a feature is the one `levelsup` levels above the leaves (L - levelsup from the root), the root
//     when that is not positive.
// Deviations (both documented, neither reachable with a well-formed vocabulary):
//   * the reference's loader reads one record past the end of the file (`while (!f.eof())`) and so appends a copy of the
//     last node as an extra child of its parent; being last among equal distances it is never chosen, and it is not
//     created here - size() is the true number of words (the reference reports one more);
//   * if a leaf is reached above level L - levelsup the reference leaves the node id uninitialised; here it is the leaf.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "ORBmatcher.h"   // FeatureVectorView

namespace se2lam_amd {

typedef uint32_t WordId;
typedef uint32_t NodeId;
typedef double WordValue;

enum WeightingType { TF_IDF = 0, TF = 1, IDF = 2, BINARY = 3 };                                          // BowVector.h:38-44
enum ScoringType { L1_NORM = 0, L2_NORM = 1, CHI_SQUARE = 2, KL = 3, BHATTACHARYYA = 4, DOT_PRODUCT = 5 };  // :47-55

// DBoW2::BowVector (std::map<WordId, WordValue>) as a vector sorted by word id
struct BowVector {
    std::vector<WordId> word;
    std::vector<WordValue> value;
    bool empty() const { return word.empty(); }
    size_t size() const { return word.size(); }
    void clear() { word.clear(); value.clear(); }
};

// DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned>>) as CSR: ascending node ids, ascending feature indices
struct FeatureVectorCSR {
    std::vector<int32_t> nodes, ptr, idx;
    bool empty() const { return nodes.empty(); }
    void clear() { nodes.clear(); ptr.clear(); idx.clear(); }
    FeatureVectorView view(const uint8_t* hasMapPoint = nullptr) const {
        FeatureVectorView v;
        v.nodes = nodes.data(); v.ptr = ptr.data(); v.idx = idx.data();
        v.numNodes = (int)nodes.size();
        v.hasMapPoint = hasMapPoint;
        return v;
    }
};

class ORBVocabulary {
public:
    static const int kDescBytes = 32;   // FORB::L

    bool empty() const { return m_words.empty(); }
    unsigned size() const { return (unsigned)m_words.size(); }   // number of words
    int getBranchingFactor() const { return m_k; }
    int getDepthLevels() const { return m_L; }
    WeightingType getWeightingType() const { return m_weighting; }
    ScoringType getScoringType() const { return m_scoring; }
    unsigned nodes() const { return (unsigned)m_parent.size(); }   // incl. the root (node 0)

    // header: nb_nodes (= nodes incl. root), size_node (41), k, L, scoring, weighting; then per node 1 .. nb_nodes-1:
    // parent (int32), descriptor (32 bytes), weight (float), is_leaf (1 byte)
    bool loadFromBinaryFile(const std::string& filename) {
        clear();
        std::FILE* f = std::fopen(filename.c_str(), "rb");
        if (!f) return false;
        uint32_t nb_nodes = 0, size_node = 0;
        int32_t k = 0, L = 0, scoring = 0, weighting = 0;
        bool ok = std::fread(&nb_nodes, 4, 1, f) == 1 && std::fread(&size_node, 4, 1, f) == 1 && std::fread(&k, 4, 1, f) == 1 &&
                  std::fread(&L, 4, 1, f) == 1 && std::fread(&scoring, 4, 1, f) == 1 && std::fread(&weighting, 4, 1, f) == 1;
        ok = ok && size_node == 4 + kDescBytes + 4 + 1 && nb_nodes >= 1 && k >= 1 && L >= 0 && scoring >= 0 && scoring <= 5 &&
             weighting >= 0 && weighting <= 3;
        if (!ok) { std::fclose(f); return false; }
        std::vector<uint8_t> rec((size_t)size_node * (nb_nodes - 1));
        const size_t got = rec.empty() ? 0 : std::fread(rec.data(), size_node, nb_nodes - 1, f);
        std::fclose(f);
        if (got != nb_nodes - 1) return false;
        m_k = k; m_L = L;
        m_scoring = (ScoringType)scoring; m_weighting = (WeightingType)weighting;
        const uint32_t N = nb_nodes;
        m_parent.assign(N, 0); m_weight.assign(N, 0.0); m_word.assign(N, -1); m_leaf.assign(N, 0);
        m_desc.assign((size_t)N * kDescBytes, 0);
        std::vector<int32_t> count(N + 1, 0);
        for (uint32_t id = 1; id < N; ++id) {
            const uint8_t* r = rec.data() + (size_t)(id - 1) * size_node;
            int32_t parent; float w;
            std::memcpy(&parent, r, 4);
            std::memcpy(&w, r + 4 + kDescBytes, 4);
            if (parent < 0 || (uint32_t)parent >= id) { clear(); return false; }   // a parent precedes its children in the file
            m_parent[id] = parent;
            std::memcpy(&m_desc[(size_t)id * kDescBytes], r + 4, kDescBytes);
            m_weight[id] = (WordValue)w;
            m_leaf[id] = r[4 + kDescBytes + 4] ? 1 : 0;
            if (m_leaf[id]) { m_word[id] = (int32_t)m_words.size(); m_words.push_back(id); }
            ++count[parent + 1];
        }
        // children in file order (CSR)
        m_child_ptr.assign(N + 1, 0);
        for (uint32_t i = 0; i < N; ++i) m_child_ptr[i + 1] = m_child_ptr[i] + count[i + 1];
        m_child.assign(m_child_ptr[N], 0);
        std::vector<int32_t> fill(m_child_ptr.begin(), m_child_ptr.end() - 1);
        for (uint32_t id = 1; id < N; ++id) m_child[fill[m_parent[id]]++] = (int32_t)id;
        // a node without children must be a leaf, or a feature that reaches it could not go on
        for (uint32_t id = 0; id < N; ++id)
            if (m_child_ptr[id + 1] == m_child_ptr[id] && !(id > 0 && m_leaf[id]) && N > 1) { clear(); return false; }
        return true;
    }

    // the file TemplatedVocabulary::saveToBinaryFile (:1526-1546) writes
    bool saveToBinaryFile(const std::string& filename) const {
        std::FILE* f = std::fopen(filename.c_str(), "wb");
        if (!f) return false;
        const uint32_t nb_nodes = nodes(), size_node = 4 + kDescBytes + 4 + 1;
        const int32_t k = m_k, L = m_L, scoring = (int32_t)m_scoring, weighting = (int32_t)m_weighting;
        std::fwrite(&nb_nodes, 4, 1, f); std::fwrite(&size_node, 4, 1, f); std::fwrite(&k, 4, 1, f); std::fwrite(&L, 4, 1, f);
        std::fwrite(&scoring, 4, 1, f); std::fwrite(&weighting, 4, 1, f);
        for (uint32_t id = 1; id < nb_nodes; ++id) {
            const int32_t parent = m_parent[id];
            const float w = (float)m_weight[id];
            const uint8_t leaf = m_leaf[id];
            std::fwrite(&parent, 4, 1, f); std::fwrite(&m_desc[(size_t)id * kDescBytes], 1, kDescBytes, f);
            std::fwrite(&w, 4, 1, f); std::fwrite(&leaf, 1, 1, f);
        }
        return std::fclose(f) == 0;
    }

    // FORB::distance (FORB.cpp:82-102): bits that differ
    static int distance(const uint8_t* a, const uint8_t* b) {
        int d = 0;
        for (int i = 0; i < kDescBytes; i += 8) {
            uint64_t x, y;
            std::memcpy(&x, a + i, 8); std::memcpy(&y, b + i, 8);
            d += __builtin_popcountll(x ^ y);
        }
        return d;
    }

    // transform(feature, word_id, weight, &nid, levelsup)   (TemplatedVocabulary.h:1241-1280)
    void transform(const uint8_t* feature, WordId& word_id, WordValue& weight, NodeId* nid = nullptr, int levelsup = 0) const {
        const int nid_level = m_L - levelsup;
        bool nid_set = false;
        if (nid_level <= 0 && nid) { *nid = 0; nid_set = true; }
        int32_t final_id = 0, current_level = 0;
        do {
            ++current_level;
            const int32_t c0 = m_child_ptr[final_id], c1 = m_child_ptr[final_id + 1];
            final_id = m_child[c0];
            int best = distance(feature, &m_desc[(size_t)final_id * kDescBytes]);
            for (int32_t c = c0 + 1; c < c1; ++c) {
                const int32_t id = m_child[c];
                const int d = distance(feature, &m_desc[(size_t)id * kDescBytes]);
                if (d < best) { best = d; final_id = id; }
            }
            if (nid && current_level == nid_level) { *nid = (NodeId)final_id; nid_set = true; }
        } while (m_child_ptr[final_id + 1] > m_child_ptr[final_id]);   // Node::isLeaf() = children.empty()
        if (nid && !nid_set) *nid = (NodeId)final_id;
        word_id = (WordId)m_word[final_id];
        weight = m_weight[final_id];
    }

    // transform(features, v, fv, levelsup)   (TemplatedVocabulary.h:1150-1216); descriptors: n x 32 bytes, row-major
    void transform(const uint8_t* descriptors, int n, BowVector& v, FeatureVectorCSR& fv, int levelsup) const {
        v.clear();
        fv.clear();
        if (empty()) return;
        std::vector<WordValue> acc(m_words.size(), 0.0);
        std::vector<uint8_t> seen(m_words.size(), 0);
        std::vector<int32_t> node_of(n, -1);
        const bool once = m_weighting == IDF || m_weighting == BINARY;   // addIfNotExist instead of addWeight
        for (int i = 0; i < n; ++i) {
            WordId id; NodeId nid; WordValue w;
            transform(descriptors + (size_t)i * kDescBytes, id, w, &nid, levelsup);
            if (!(w > 0)) continue;   // stopped
            if (!seen[id]) { seen[id] = 1; acc[id] = w; }
            else if (!once) acc[id] += w;
            node_of[i] = (int32_t)nid;
        }
        for (size_t id = 0; id < acc.size(); ++id)
            if (seen[id]) { v.word.push_back((WordId)id); v.value.push_back(acc[id]); }
        const bool must = m_scoring != DOT_PRODUCT;
        if (!once && !v.empty() && !must) {
            const double nd = (double)v.size();
            for (auto& x : v.value) x /= nd;
        }
        if (must) {   // BowVector::normalize (BowVector.cpp:62-84)
            double norm = 0.0;
            if (m_scoring == L2_NORM) { for (double x : v.value) norm += x * x; norm = std::sqrt(norm); }
            else for (double x : v.value) norm += std::fabs(x);
            if (norm > 0.0) for (auto& x : v.value) x /= norm;
        }
        // FeatureVector: node -> features, both ascending (counting sort over the node ids that occur)
        std::vector<int32_t> cnt(nodes() + 1, 0);
        for (int i = 0; i < n; ++i) if (node_of[i] >= 0) ++cnt[node_of[i] + 1];
        std::vector<int32_t> start(nodes(), -1);
        fv.ptr.push_back(0);
        for (uint32_t node = 0; node < nodes(); ++node)
            if (cnt[node + 1]) {
                start[node] = fv.ptr.back();
                fv.nodes.push_back((int32_t)node);
                fv.ptr.push_back(fv.ptr.back() + cnt[node + 1]);
            }
        fv.idx.assign(fv.ptr.back(), 0);
        for (int i = 0; i < n; ++i) if (node_of[i] >= 0) fv.idx[start[node_of[i]]++] = i;
    }

    // the reference's call line unchanged - `_pVoc->transform(vCurrentDesc, mBowVec, mFeatVec, 4)` (KeyFrame.cpp:248-251) with
    // vCurrentDesc = toDescriptorVector(descriptors), a std::vector<cv::Mat> of 1 x 32 rows: any row type with a `data`
    // member pointing at the 32 descriptor bytes
    template <class Row>
    void transform(const std::vector<Row>& features, BowVector& v, FeatureVectorCSR& fv, int levelsup) const {
        std::vector<uint8_t> rows(features.size() * (size_t)kDescBytes);
        for (size_t i = 0; i < features.size(); ++i) std::memcpy(&rows[i * kDescBytes], features[i].data, kDescBytes);
        transform(rows.data(), (int)features.size(), v, fv, levelsup);
    }

    // score(a, b): the vectors are sorted and normalised as transform() leaves them   (ScoringObject.cpp)
    double score(const BowVector& a, const BowVector& b) const {
        double s = 0.0;
        size_t i = 0, j = 0;
        while (i < a.size() && j < b.size()) {
            if (a.word[i] == b.word[j]) {
                const double vi = a.value[i], wi = b.value[j];
                switch (m_scoring) {
                    case L1_NORM: s += std::fabs(vi - wi) - std::fabs(vi) - std::fabs(wi); break;
                    case L2_NORM: case DOT_PRODUCT: s += vi * wi; break;
                    case CHI_SQUARE: if (vi + wi != 0.0) s += vi * wi / (vi + wi); break;
                    case BHATTACHARYYA: s += std::sqrt(vi * wi); break;
                    case KL: break;   // needs the words only one vector has as well: below
                }
                ++i; ++j;
            } else if (a.word[i] < b.word[j]) ++i;
            else ++j;
        }
        switch (m_scoring) {
            case L1_NORM: return -s / 2.0;                                      // 1 - 0.5 ||v - w||_1, in [0, 1]
            case L2_NORM: return s >= 1.0 ? 1.0 : 1.0 - std::sqrt(1.0 - s);
            case CHI_SQUARE: return 2.0 * s;
            case KL: return scoreKL(a, b);
            default: return s;
        }
    }

    void clear() {
        m_k = 0; m_L = 0; m_scoring = L1_NORM; m_weighting = TF_IDF;
        m_parent.clear(); m_child_ptr.clear(); m_child.clear(); m_desc.clear(); m_weight.clear(); m_word.clear(); m_leaf.clear();
        m_words.clear();
    }

    // flat read access (tests, device upload of the tree in a later round)
    const std::vector<int32_t>& parents() const { return m_parent; }
    const std::vector<int32_t>& childPtr() const { return m_child_ptr; }
    const std::vector<int32_t>& children() const { return m_child; }
    const std::vector<uint8_t>& descriptors() const { return m_desc; }
    const std::vector<WordValue>& weights() const { return m_weight; }
    const std::vector<int32_t>& wordOfNode() const { return m_word; }

private:
    // KLScoring::score (ScoringObject.cpp:175-222): sum over the words of a; a word b lacks counts with log(eps)
    double scoreKL(const BowVector& a, const BowVector& b) const {
        const double log_eps = std::log(2.220446049250313e-16);   // GeneralScoring::LOG_EPS = log(DBL_EPSILON)
        double s = 0.0;
        size_t j = 0;
        for (size_t i = 0; i < a.size(); ++i) {
            while (j < b.size() && b.word[j] < a.word[i]) ++j;
            const double vi = a.value[i];
            if (j < b.size() && b.word[j] == a.word[i]) {
                const double wi = b.value[j];
                if (vi != 0.0 && wi != 0.0) s += vi * std::log(vi / wi);
            } else if (vi != 0.0) {
                s += vi * (std::log(vi) - log_eps);
            }
        }
        return s;
    }

    int m_k = 0, m_L = 0;
    ScoringType m_scoring = L1_NORM;
    WeightingType m_weighting = TF_IDF;
    std::vector<int32_t> m_parent, m_child_ptr, m_child, m_word;
    std::vector<uint8_t> m_desc, m_leaf;
    std::vector<WordValue> m_weight;
    std::vector<uint32_t> m_words;   // word id -> node id
};

}  // namespace se2lam_amd
