// DROP-IN BINDING - this file takes the place of /root/reference/src/ORBmatcher.cpp in a build of the
// reference: the class is the reference's own, declared by its own, unmodified header (include/se2lam/ORBmatcher.h:42-80); the
// member definitions flatten what the reference's loops read from Frame / KeyFrame / MapPoint into the POD views of the
// mirror (include/se2lam_amd/ORBmatcher.h) and go to libse2gpu (se2gpu_match_window / _projection / se2gpu_search_by_bow).
// Compiled by `make -C oracle pipeline` with the reference's own Frame.cpp, Track.cpp, LocalMapper.cpp, Map.cpp, KeyFrame.cpp,
// MapPoint.cpp, GlobalMapper.cpp, Localizer.cpp ... where they lie; run by tests/test_dropin_pipeline.py.
//
// The reference constructs a matcher on the stack of every call (src/Track.cpp:131, src/LocalMapper.cpp:117) and its header
// has no member for a device handle: every calling thread keeps one mirror object (stream + device workspace) and the
// reference's two public fields are copied onto it per call.
#include "ORBmatcher.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "se2lam_amd/ORBmatcher.h"

#ifndef SE2LAM_AMD_HAVE_OPENCV
#error "the cv:: conversions are needed here (include/se2lam_amd/conversions.h did not find <opencv2/core.hpp>)"
#endif

namespace se2lam {

using std::vector;

const int ORBmatcher::TH_HIGH = 100;       // src/ORBmatcher.cpp:45-47
const int ORBmatcher::TH_LOW = 75;
const int ORBmatcher::HISTO_LENGTH = 30;

namespace {
se2lam_amd::ORBmatcher& device_matcher(float nnratio, bool checkOri) {
    thread_local se2lam_amd::ORBmatcher m;
    m.mfNNratio = nnratio;
    m.mbCheckOrientation = checkOri;
    return m;
}
const uint8_t* rows32(const cv::Mat& d, vector<uint8_t>& scratch) {   // N x 32 CV_8U, continuous as Frame::descriptors always is
    if (d.empty()) return nullptr;
    if (d.isContinuous()) return d.ptr<uint8_t>(0);
    scratch.resize((size_t)d.rows * 32);
    for (int r = 0; r < d.rows; ++r) std::memcpy(&scratch[(size_t)r * 32], d.ptr<uint8_t>(r), 32);
    return scratch.data();
}
se2lam_amd::FrameView view_of(const Frame& f, vector<uint8_t>& scratch) {
    se2lam_amd::FrameView v;
    v.keyPointsUn = se2lam_amd::mirror(f.keyPointsUn);
    v.descriptors = rows32(f.descriptors, scratch);
    v.N = f.N;
    v.minXUn = Frame::minXUn; v.minYUn = Frame::minYUn; v.maxXUn = Frame::maxXUn; v.maxYUn = Frame::maxYUn;   // src/Frame.cpp:37-44
    return v;
}
struct FlatFeatureVector {   // DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned>>) as CSR
    vector<int32_t> nodes, ptr, idx;
    vector<uint8_t> hasMP;
    se2lam_amd::FeatureVectorView view() const {
        se2lam_amd::FeatureVectorView v;
        v.nodes = nodes.data(); v.ptr = ptr.data(); v.idx = idx.data(); v.numNodes = (int)nodes.size(); v.hasMapPoint = hasMP.data();
        return v;
    }
};
FlatFeatureVector flatten(const DBoW2::FeatureVector& fv, const vector<PtrMapPoint>& mps) {
    FlatFeatureVector o;
    o.ptr.push_back(0);
    for (const auto& kv : fv) {
        o.nodes.push_back((int32_t)kv.first);
        for (unsigned i : kv.second) o.idx.push_back((int32_t)i);
        o.ptr.push_back((int32_t)o.idx.size());
    }
    o.hasMP.resize(mps.size() ? mps.size() : 1, 0);
    for (size_t i = 0; i < mps.size(); ++i) o.hasMP[i] = (mps[i] && !mps[i]->isNull()) ? 1 : 0;   // src/ORBmatcher.cpp:176-181
    return o;
}
}  // namespace

ORBmatcher::ORBmatcher(float nnratio, bool checkOri) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

// ORBmatcher.h:49, src/ORBmatcher.cpp:110-126
int ORBmatcher::DescriptorDistance(const cv::Mat& a, const cv::Mat& b) {
    return se2lam_amd::ORBmatcher::DescriptorDistance(a.ptr<uint8_t>(0), b.ptr<uint8_t>(0));
}

// ORBmatcher.h:57, src/ORBmatcher.cpp:64-105
void ORBmatcher::ComputeThreeMaxima(vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3) {
    device_matcher(mfNNratio, mbCheckOrientation).ComputeThreeMaxima(histo, L, ind1, ind2, ind3);
}

// ORBmatcher.h:68-71, src/ORBmatcher.cpp:278-381
int ORBmatcher::MatchByWindow(const Frame& frame1, Frame& frame2, vector<cv::Point2f>& vbPrevMatched, const int winSize,
                              vector<int>& vnMatches12, const int levelOffset, const int minLevel, const int maxLevel) {
    static_assert(sizeof(cv::Point2f) == sizeof(se2lam_amd::Point2f), "vbPrevMatched is updated in place");
    vector<uint8_t> s1, s2;
    const se2lam_amd::FrameView v1 = view_of(frame1, s1), v2 = view_of(frame2, s2);
    vector<se2lam_amd::Point2f> prev(vbPrevMatched.size());
    if (!prev.empty()) std::memcpy(prev.data(), vbPrevMatched.data(), prev.size() * sizeof(se2lam_amd::Point2f));
    const int n = device_matcher(mfNNratio, mbCheckOrientation).MatchByWindow(v1, v2, prev, winSize, vnMatches12, levelOffset, minLevel, maxLevel);
    if (!prev.empty()) std::memcpy(vbPrevMatched.data(), prev.data(), prev.size() * sizeof(se2lam_amd::Point2f));   // src/ORBmatcher.cpp:375-377
    return n;
}

// ORBmatcher.h:73-74, src/ORBmatcher.cpp:383-454
int ORBmatcher::MatchByProjection(PtrKeyFrame& pNewKF, vector<PtrMapPoint>& localMPs, const int winSize, const int levelOffset,
                                  vector<int>& vMatchesIdxMP) {
    const int M = (int)localMPs.size(), N = pNewKF->N;
    vector<float> pos((size_t)M * 3 + 3);
    vector<uint8_t> desc((size_t)M * 32 + 32, 0), skip((size_t)M + 1, 0), observed((size_t)N + 1, 0), scratch;
    vector<int32_t> octave((size_t)M + 1, 0);
    for (int i = 0; i < M; ++i) {
        const PtrMapPoint& pMP = localMPs[i];
        if (pMP->isNull() || !pMP->isGoodPrl() || pNewKF->hasObservation(pMP)) { skip[i] = 1; continue; }   // src/ORBmatcher.cpp:392-395
        const cv::Point3f p = pMP->getPos();
        pos[3 * i] = p.x; pos[3 * i + 1] = p.y; pos[3 * i + 2] = p.z;
        if (pMP->mMainDescriptor.empty()) { skip[i] = 1; continue; }
        std::memcpy(&desc[(size_t)i * 32], pMP->mMainDescriptor.ptr<uint8_t>(0), 32);
        octave[i] = pMP->mMainOctave;
    }
    for (int i = 0; i < N; ++i) observed[i] = pNewKF->hasObservation(i) ? 1 : 0;   // src/ORBmatcher.cpp:417
    float Tcw[12];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) Tcw[4 * r + c] = pNewKF->Tcw.at<float>(r, c);
    se2lam_amd::FrameView kf = view_of(*pNewKF, scratch);
    kf.observed = observed.data();
    kf.Tcw = Tcw;
    kf.fx = Config::Kcam.at<float>(0, 0); kf.fy = Config::Kcam.at<float>(1, 1);   // cvu::camprjc(Config::Kcam, ...), src/ORBmatcher.cpp:397
    kf.cx = Config::Kcam.at<float>(0, 2); kf.cy = Config::Kcam.at<float>(1, 2);
    se2lam_amd::MapPointView mps;
    mps.pos = pos.data(); mps.mainDescriptor = desc.data(); mps.mainOctave = octave.data(); mps.skip = skip.data(); mps.M = M;
    const int nm = device_matcher(mfNNratio, mbCheckOrientation).MatchByProjection(kf, mps, winSize, levelOffset, vMatchesIdxMP);
    if (const char* dump = std::getenv("SE2_DROPIN_DUMP")) {   // debugging aid: the flattened call, for a replay through the other implementations
        static int call = 0;
        char name[512];
        std::snprintf(name, sizeof(name), "%s/match_projection_%d.bin", dump, call++);
        if (FILE* f = std::fopen(name, "wb")) {
            const int32_t hdr[6] = {M, N, winSize, levelOffset, nm, 0};
            std::fwrite(hdr, 4, 6, f);
            const float k4[5] = {kf.fx, kf.fy, kf.cx, kf.cy, mfNNratio};
            std::fwrite(k4, 4, 5, f);
            const float bounds[4] = {kf.minXUn, kf.minYUn, kf.maxXUn, kf.maxYUn};
            std::fwrite(bounds, 4, 4, f);
            std::fwrite(Tcw, 4, 12, f);
            std::fwrite(pos.data(), 4, (size_t)M * 3, f);
            std::fwrite(desc.data(), 1, (size_t)M * 32, f);
            std::fwrite(octave.data(), 4, (size_t)M, f);
            std::fwrite(skip.data(), 1, (size_t)M, f);
            std::fwrite(kf.keyPointsUn, sizeof(se2lam_amd::KeyPoint), (size_t)N, f);
            std::fwrite(kf.descriptors, 1, (size_t)N * 32, f);
            std::fwrite(observed.data(), 1, (size_t)N, f);
            std::fwrite(vMatchesIdxMP.data(), 4, (size_t)N, f);
            std::fclose(f);
        }
    }
    return nm;
}

// ORBmatcher.h:55, src/ORBmatcher.cpp:128-276
int ORBmatcher::SearchByBoW(PtrKeyFrame pKF1, PtrKeyFrame pKF2, std::map<int, int>& mapMatches12, bool bIfMPOnly) {
    mapMatches12.clear();
    if (pKF1 == NULL || pKF1->isNull() || pKF2 == NULL || pKF2->isNull()) return 0;   // src/ORBmatcher.cpp:133-135
    const FlatFeatureVector f1 = flatten(pKF1->GetFeatureVector(), pKF1->GetMapPointMatches());
    const FlatFeatureVector f2 = flatten(pKF2->GetFeatureVector(), pKF2->GetMapPointMatches());
    vector<uint8_t> s1, s2;
    se2lam_amd::FrameView v1 = view_of(*pKF1, s1), v2 = view_of(*pKF2, s2);
    const se2lam_amd::FeatureVectorView fv1 = f1.view(), fv2 = f2.view();
    v1.bow = &fv1; v2.bow = &fv2;
    return device_matcher(mfNNratio, mbCheckOrientation).SearchByBoW(v1, v2, mapMatches12, bIfMPOnly);
}

// ORBmatcher.h:77, src/ORBmatcher.cpp:54-60 (no caller in the reference)
float ORBmatcher::RadiusByViewingCos(const float& viewCos) { return viewCos > 0.998 ? 2.5 : 4.0; }

}  // namespace se2lam
