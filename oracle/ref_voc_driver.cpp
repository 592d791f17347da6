// ORACLE - TEST INFRASTRUCTURE ONLY (oracle/_ref/libse2lam_ref_map.so).  Never linked, imported or called by the product path.
//
// C entry points over the vocabulary the reference uses: se2lam::ORBVocabulary = DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>
// (include/se2lam/ORBVocabulary.h; Thirdparty/DBoW2 is vendored in the reference's tree and compiled from there, unmodified):
//   loadFromBinaryFile      Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h  (OdoSLAM.cpp:45)
//   transform(.., levelsup) the BowVector / FeatureVector of a key frame   (KeyFrame::ComputeBoW, src/KeyFrame.cpp:244-254)
//   score                   two BowVectors                                 (GlobalMapper::DetectLoopClose, src/GlobalMapper.cpp:237)
// What include/se2lam_amd/ORBVocabulary.h - the library's host-side mirror - is held to (tests/test_ref_compiled.py).
#include <cstdint>
#include <string>
#include <vector>

#include "ORBVocabulary.h"

using se2lam::ORBVocabulary;

extern "C" {

// info6 = {branching factor, depth levels, words, scoring type, weighting type, loaded}
void* ref_voc_load(const char* path, int32_t* info6) {
    ORBVocabulary* v = new ORBVocabulary();
    const bool ok = v->loadFromBinaryFile(path);
    info6[0] = v->getBranchingFactor(); info6[1] = v->getDepthLevels(); info6[2] = (int32_t)v->size();
    info6[3] = (int32_t)v->getScoringType(); info6[4] = (int32_t)v->getWeightingType(); info6[5] = ok ? 1 : 0;
    return v;
}
void ref_voc_free(void* h) { delete static_cast<ORBVocabulary*>(h); }

// transform(features, bow, fv, levelsup) on n descriptors of 32 bytes (rows of a CV_8U matrix, as toDescriptorVector hands them over).
// BowVector in word order (std::map), FeatureVector as CSR in node order.  Returns 0, or -1 when a capacity is too small.
int ref_voc_transform(void* h, const uint8_t* desc, int n, int levelsup, int cap_bow, uint32_t* bow_id, double* bow_val, int32_t* n_bow, int cap_nodes,
                      int32_t* fv_node, int32_t* fv_ptr, int cap_idx, int32_t* fv_idx, int32_t* n_nodes) {
    const ORBVocabulary* v = static_cast<const ORBVocabulary*>(h);
    cv::Mat D(std::max(n, 1), 32, CV_8UC1);
    for (int i = 0; i < n; ++i) for (int b = 0; b < 32; ++b) D.at<unsigned char>(i, b) = desc[32 * (size_t)i + b];
    std::vector<cv::Mat> features;
    for (int i = 0; i < n; ++i) features.push_back(D.row(i));
    DBoW2::BowVector bow;
    DBoW2::FeatureVector fv;
    v->transform(features, bow, fv, levelsup);
    *n_bow = (int32_t)bow.size();
    *n_nodes = (int32_t)fv.size();
    size_t total = 0;
    for (const auto& kv : fv) total += kv.second.size();
    if ((int)bow.size() > cap_bow || (int)fv.size() > cap_nodes || (int)total > cap_idx) return -1;
    int k = 0;
    for (const auto& kv : bow) { bow_id[k] = kv.first; bow_val[k] = kv.second; ++k; }
    k = 0;
    int t = 0;
    for (const auto& kv : fv) {
        fv_node[k] = (int32_t)kv.first;
        fv_ptr[k] = t;
        for (unsigned int idx : kv.second) fv_idx[t++] = (int32_t)idx;
        ++k;
    }
    fv_ptr[k] = t;
    return 0;
}
double ref_voc_score(void* h, int n1, const uint32_t* id1, const double* val1, int n2, const uint32_t* id2, const double* val2) {
    const ORBVocabulary* v = static_cast<const ORBVocabulary*>(h);
    DBoW2::BowVector a, b;
    for (int i = 0; i < n1; ++i) a.insert(a.end(), std::make_pair(id1[i], val1[i]));
    for (int i = 0; i < n2; ++i) b.insert(b.end(), std::make_pair(id2[i], val2[i]));
    return v->score(a, b);
}

}  // extern "C"
