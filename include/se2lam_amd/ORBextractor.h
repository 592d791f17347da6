// Drop-in mirror of se2lam::ORBextractor (/root/reference/include/se2lam/ORBextractor.h:38-83) over libse2gpu.
//   ORBextractor(nfeatures=1000, scaleFactor=1.2f, nlevels=8, scoreType=FAST_SCORE, fastTh=20)   (:44)
//   void operator()(image, mask, keypoints, descriptors)                                           (:49-51)
//   int GetLevels()  float GetScaleFactor()                                                        (:53-57)
// Callers in the reference: Frame::Frame (src/Frame.cpp:25, empty mask), Track (src/Track.cpp:34), Localizer.
// One instance per calling thread (the reference's instance is stateful too: mvImagePyramid).
#pragma once
#include <algorithm>
#include <cstring>

#include "types.h"

namespace se2lam_amd {

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    // the reference's five arguments (ORBextractor.h:44); the device buffers follow the image size they meet
    ORBextractor(int nfeatures = 1000, float scaleFactor = 1.2f, int nlevels = 8, int scoreType = FAST_SCORE,
                 int fastTh = 20)
        : nfeatures_(nfeatures) {
        se2gpu_orb_params p{};
        p.nfeatures = nfeatures; p.scale_factor = scaleFactor; p.nlevels = nlevels;
        p.score_type = scoreType; p.fast_th = fastTh; p.max_rows = 0; p.max_cols = 0; p.max_batch = 1;
        check(se2gpu_orb_create(&p, &h_), "ORBextractor");
    }
    ~ORBextractor() { se2gpu_orb_destroy(h_); }
    ORBextractor(const ORBextractor&) = delete;
    ORBextractor& operator=(const ORBextractor&) = delete;

    // Compute the ORB features and descriptors on an image.  A non-empty mask is accepted and ignored, as in the reference
    // (ORBextractor.cpp:797-828 builds a mask pyramid that cv::FAST at :616 / :622 is never given)
    void operator()(const Mat8U& image, const Mat8U& mask, std::vector<KeyPoint>& keypoints, Mat8U& descriptors) {
        if (image.empty()) return;                               // ORBextractor.cpp:730-731
        const int cap = 2 * nfeatures_;
        keypoints.assign(cap, KeyPoint());
        std::vector<uint8_t> desc((size_t)cap * 32);
        int n = 0;
        check(se2gpu_orb_extract(h_, image.data, image.rows, image.cols, image.step, mask.empty() ? nullptr : mask.data,
                                 reinterpret_cast<se2gpu_keypoint*>(keypoints.data()), desc.data(), cap, &n),
              "ORBextractor::operator()");
        keypoints.resize(n);
        if (n == 0) { descriptors = Mat8U(); return; }           // _descriptors.release()
        descriptors.create(n, 32);                               // _descriptors.create(nkeypoints, 32, CV_8U)
        std::copy(desc.begin(), desc.begin() + (size_t)n * 32, descriptors.owned.begin());
    }

#ifdef SE2LAM_AMD_HAVE_OPENCV
    // void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint>&, cv::OutputArray descriptors)
    void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint>& keypoints, cv::OutputArray descriptors) {
        std::vector<KeyPoint> kps;
        Mat8U d;
        (*this)(view8U(image.getMat()), mask.empty() ? Mat8U() : view8U(mask.getMat()), kps, d);
        keypoints.resize(kps.size());
        if (!kps.empty()) std::memcpy(keypoints.data(), kps.data(), kps.size() * sizeof(KeyPoint));   // layout-identical
        if (kps.empty()) { descriptors.release(); return; }
        descriptors.create((int)kps.size(), 32, CV_8U);
        cv::Mat out = descriptors.getMat();
        for (int r = 0; r < d.rows; ++r) std::memcpy(out.ptr<uint8_t>(r), d.ptr(r), 32);
    }
#endif

    int GetLevels() { return se2gpu_orb_levels(h_); }
    float GetScaleFactor() { return se2gpu_orb_scale_factor(h_); }

private:
    se2gpu_orb* h_ = nullptr;
    int nfeatures_;
};

}  // namespace se2lam_amd
