// Driver of include/se2lam_amd/ORBVocabulary.h for tests/test_vocabulary.py (host only, no device):
//   cpp_vocabulary <voc.bin> <desc_a.bin> <desc_b.bin> <levelsup> <resaved.bin>
// prints the BowVector and FeatureVector of both descriptor sets (KeyFrame::ComputeBoW, KeyFrame.cpp:244-254), their score
// (GlobalMapper.cpp:237) and writes the vocabulary back (saveToBinaryFile).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <vector>

#include "se2lam_amd/ORBVocabulary.h"

static std::vector<uint8_t> slurp(const char* path) {
    std::ifstream f(path, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv) {
    if (argc < 6) return 2;
    se2lam_amd::ORBVocabulary voc;
    if (!voc.loadFromBinaryFile(argv[1])) { std::printf("LOAD failed\n"); return 1; }
    std::printf("VOC %d %d %u %u %d %d\n", voc.getBranchingFactor(), voc.getDepthLevels(), voc.nodes(), voc.size(),
                (int)voc.getScoringType(), (int)voc.getWeightingType());
    const int levelsup = std::atoi(argv[4]);
    se2lam_amd::BowVector bow[2];
    for (int s = 0; s < 2; ++s) {
        const std::vector<uint8_t> d = slurp(argv[2 + s]);
        const int n = (int)(d.size() / 32);
        se2lam_amd::FeatureVectorCSR fv;
        voc.transform(d.data(), n, bow[s], fv, levelsup);
        {   // the reference's call line with a vector of row objects (cv::Mat rows of toDescriptorVector, KeyFrame.cpp:248-251)
            struct Row { const uint8_t* data; };
            std::vector<Row> vCurrentDesc;
            for (int i = 0; i < n; ++i) vCurrentDesc.push_back(Row{d.data() + 32 * (size_t)i});
            se2lam_amd::BowVector mBowVec; se2lam_amd::FeatureVectorCSR mFeatVec;
            const se2lam_amd::ORBVocabulary* _pVoc = &voc;
            _pVoc->transform(vCurrentDesc, mBowVec, mFeatVec, levelsup);
            if (mBowVec.word != bow[s].word || mBowVec.value != bow[s].value || mFeatVec.nodes != fv.nodes ||
                mFeatVec.ptr != fv.ptr || mFeatVec.idx != fv.idx) { std::printf("ROWS differ\n"); return 4; }
        }
        std::printf("BOW%d", s);
        for (size_t i = 0; i < bow[s].size(); ++i) std::printf(" %u:%.17g", bow[s].word[i], bow[s].value[i]);
        std::printf("\nFV%d", s);
        const se2lam_amd::FeatureVectorView v = fv.view();
        for (int k = 0; k < v.numNodes; ++k) {
            std::printf(" %d:", v.nodes[k]);
            for (int t = v.ptr[k]; t < v.ptr[k + 1]; ++t) std::printf("%s%d", t == v.ptr[k] ? "" : ",", v.idx[t]);
        }
        std::printf("\n");
    }
    std::printf("SCORE %.17g %.17g\n", voc.score(bow[0], bow[1]), voc.score(bow[0], bow[0]));
    return voc.saveToBinaryFile(argv[5]) ? 0 : 3;
}
