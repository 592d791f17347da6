// Drop-in mirror of se2lam::ORBmatcher (/root/reference/include/se2lam/ORBmatcher.h:42-80) over libse2gpu.
//   ORBmatcher(nnratio=0.6, checkOri=true)                                                        (:46)
//   static int DescriptorDistance(a, b)                                                           (:49)
//   int MatchByWindow(frame1, frame2, vbPrevMatched, winSize, vnMatches12, levelOffset=1, minLevel=0, maxLevel=8) (:68-71)
//   int MatchByProjection(pNewKF, localMPs, winSize, levelOffset, vMatchesIdxMP)                  (:73-74)
//   int SearchByBoW(pKF1, pKF2, mapIdxMatches12, bIfMPOnly=true)                                 (:55)
//   void ComputeThreeMaxima(histo, L, ind1, ind2, ind3)                                          (:57)
// Frame / KeyFrame / MapPoint are pointer-graph classes of the reference's data model (out of scope); the adapters
// take the POD content the matchers actually read from them (FrameView / MapPointView below) - INTEGRATION.md shows
// the three-line glue that fills the views from the reference's classes.
#pragma once
#include <map>
#include <vector>

#include "types.h"

namespace se2lam_amd {

struct FrameView {                       // what MatchByWindow / MatchByProjection read from a (Key)Frame
    const KeyPoint* keyPointsUn = nullptr;   // Frame::keyPointsUn
    const uint8_t* descriptors = nullptr;    // Frame::descriptors (N x 32, CV_8U, continuous)
    int N = 0;
    float minXUn = 0, minYUn = 0, maxXUn = 640, maxYUn = 480;  // Frame::minXUn.. (Frame.cpp:183-200)
    const uint8_t* observed = nullptr;       // KeyFrame::hasObservation(idx) per feature (MatchByProjection only)
    const float* Tcw = nullptr;              // KeyFrame::Tcw rows 0..2 (3x4 row-major float)
    // Config::fxCam, fyCam, cxCam, cyCam: the reference's matcher reads them from the global Config (ORBmatcher.cpp:400-401)
    float fx = 0, fy = 0, cx = 0, cy = 0;
    const struct FeatureVectorView* bow = nullptr;   // KeyFrame::GetFeatureVector() (SearchByBoW only)
};

struct FeatureVectorView {               // DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned>>) flattened to CSR
    const int32_t* nodes = nullptr;          // ascending node ids
    const int32_t* ptr = nullptr;            // numNodes + 1 offsets into idx
    const int32_t* idx = nullptr;            // feature indices
    int numNodes = 0;
    const uint8_t* hasMapPoint = nullptr;    // per feature: GetMapPointMatches()[i] non-null (read only if bIfMPOnly)
};

struct MapPointView {                    // what MatchByProjection reads from localMPs[i]
    const float* pos = nullptr;              // M x 3, MapPoint::getPos()
    const uint8_t* mainDescriptor = nullptr; // M x 32, MapPoint::mMainDescriptor
    const int32_t* mainOctave = nullptr;     // M, MapPoint::mMainOctave
    const uint8_t* skip = nullptr;           // M, 1 = isNull() || !isGoodPrl() || pNewKF->hasObservation(pMP)
    int M = 0;
};

class ORBmatcher {
public:
    static const int TH_HIGH = 100, TH_LOW = 75, HISTO_LENGTH = 30;   // ORBmatcher.cpp:45-47

    // ORBmatcher(float nnratio = 0.6, bool checkOri = true)  (ORBmatcher.h:46); the device workspace grows on demand
    ORBmatcher(float nnratio = 0.6f, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {
        check(se2gpu_matcher_create(4096, 1, &h_), "ORBmatcher");
    }
    ~ORBmatcher() { se2gpu_matcher_destroy(h_); }
    ORBmatcher(const ORBmatcher&) = delete;
    ORBmatcher& operator=(const ORBmatcher&) = delete;

    static int DescriptorDistance(const uint8_t* a, const uint8_t* b) { return se2gpu_hamming(a, b); }

    int MatchByWindow(const FrameView& frame1, const FrameView& frame2, std::vector<Point2f>& vbPrevMatched,
                      const int winSize, std::vector<int>& vnMatches12, const int levelOffset = 1,
                      const int minLevel = 0, const int maxLevel = 8) {
        vnMatches12.assign(frame1.N, -1);
        se2gpu_frame_bounds b{frame2.minXUn, frame2.minYUn, frame2.maxXUn, frame2.maxYUn};
        int nmatches = 0;
        check(se2gpu_match_window(h_, &b, reinterpret_cast<const se2gpu_keypoint*>(frame1.keyPointsUn), frame1.descriptors,
                                  frame1.N, reinterpret_cast<const se2gpu_keypoint*>(frame2.keyPointsUn),
                                  frame2.descriptors, frame2.N, reinterpret_cast<float*>(vbPrevMatched.data()), winSize,
                                  levelOffset, minLevel, maxLevel, mfNNratio, vnMatches12.data(), &nmatches),
              "ORBmatcher::MatchByWindow");
        return nmatches;
    }

    // int MatchByProjection(PtrKeyFrame& pNewKF, std::vector<PtrMapPoint>& localMPs, winSize, levelOffset, vMatchesIdxMP)
    // (ORBmatcher.h:73-74): the key-frame view carries Tcw and the intrinsics the reference reads from Config
    int MatchByProjection(const FrameView& newKF, const MapPointView& localMPs, const int winSize, const int levelOffset,
                          std::vector<int>& vMatchesIdxMP) {
        const float fx = newKF.fx, fy = newKF.fy, cx = newKF.cx, cy = newKF.cy;
        vMatchesIdxMP.assign(newKF.N, -1);
        se2gpu_frame_bounds b{newKF.minXUn, newKF.minYUn, newKF.maxXUn, newKF.maxYUn};
        int nmatches = 0;
        check(se2gpu_match_projection(h_, &b, localMPs.pos, localMPs.mainDescriptor, localMPs.mainOctave, localMPs.skip,
                                      localMPs.M, newKF.Tcw, fx, fy, cx, cy,
                                      reinterpret_cast<const se2gpu_keypoint*>(newKF.keyPointsUn), newKF.descriptors,
                                      newKF.observed, newKF.N, winSize, levelOffset, mfNNratio, vMatchesIdxMP.data(),
                                      &nmatches),
              "ORBmatcher::MatchByProjection");
        return nmatches;
    }

    // SearchByBoW(pKF1, pKF2, mapIdxMatches12, bIfMPOnly) (ORBmatcher.h:55, ORBmatcher.cpp:128-276); matches as a dense
    // array (index into KF2 or -1) instead of std::map<int,int>
    int SearchByBoW(const FrameView& kf1, const FeatureVectorView& fv1, const FrameView& kf2, const FeatureVectorView& fv2,
                    std::vector<int>& matches12, bool bIfMPOnly = true) {
        matches12.assign(kf1.N, -1);
        int nmatches = 0;
        check(se2gpu_search_by_bow(h_, reinterpret_cast<const se2gpu_keypoint*>(kf1.keyPointsUn), kf1.descriptors, kf1.N,
                                   fv1.nodes, fv1.ptr, fv1.idx, fv1.numNodes, fv1.hasMapPoint,
                                   reinterpret_cast<const se2gpu_keypoint*>(kf2.keyPointsUn), kf2.descriptors, kf2.N,
                                   fv2.nodes, fv2.ptr, fv2.idx, fv2.numNodes, fv2.hasMapPoint, bIfMPOnly ? 1 : 0, mfNNratio,
                                   mbCheckOrientation ? 1 : 0, matches12.data(), &nmatches),
              "ORBmatcher::SearchByBoW");
        return nmatches;
    }

    // int SearchByBoW(PtrKeyFrame pKF1, PtrKeyFrame pKF2, std::map<int, int>& mapIdxMatches12, bool bIfMPOnly = true)
    // (ORBmatcher.h:55): the key-frame views carry their feature vectors (FrameView::bow)
    int SearchByBoW(const FrameView& kf1, const FrameView& kf2, std::map<int, int>& mapIdxMatches12, bool bIfMPOnly = true) {
        if (!kf1.bow || !kf2.bow) throw std::invalid_argument("ORBmatcher::SearchByBoW: FrameView::bow is not set");
        std::vector<int> dense;
        const int n = SearchByBoW(kf1, *kf1.bow, kf2, *kf2.bow, dense, bIfMPOnly);
        mapIdxMatches12.clear();
        for (int i = 0; i < (int)dense.size(); ++i)
            if (dense[i] >= 0) mapIdxMatches12[i] = dense[i];
        return n;
    }

    // void ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3) (ORBmatcher.h:57): a
    // public member in the reference; the device resolve kernels run the same function (se2gpu_three_maxima)
    void ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3) {
        std::vector<int32_t> counts(L > 0 ? L : 0);
        for (int i = 0; i < L; ++i) counts[i] = (int32_t)histo[i].size();
        check(se2gpu_three_maxima(counts.data(), L, &ind1, &ind2, &ind3), "ORBmatcher::ComputeThreeMaxima");
    }

    float mfNNratio;             // public in the reference (ORBmatcher.h:65-66)
    bool mbCheckOrientation;

protected:
    se2gpu_matcher* h_ = nullptr;
};

}  // namespace se2lam_amd
