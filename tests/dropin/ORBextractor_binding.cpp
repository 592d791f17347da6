// DROP-IN BINDING - this file takes the place of /root/reference/src/ORBextractor.cpp in a build of the
// reference: the class is the reference's own, declared by its own, unmodified header (include/se2lam/ORBextractor.h:38-83);
// only the member definitions are new and go to libse2gpu through the C++ mirror (include/se2lam_amd/ORBextractor.h ->
// se2gpu_orb_create / se2gpu_orb_extract).  Compiled by `make -C oracle pipeline` together with the reference's Frame.cpp,
// Track.cpp, LocalMapper.cpp, Map.cpp ... where they lie (oracle/Makefile), and run by tests/test_dropin_pipeline.py.
//
// The reference's header has no member to keep a device handle in and stays as it is, so the handle lives in a table keyed by
// the object's address (an extractor is constructed once per thread and never destroyed: src/Track.cpp:34, src/Localizer.cpp:21;
// the header's inline destructor is empty).
#include "ORBextractor.h"

#include <memory>
#include <mutex>
#include <unordered_map>

#include "se2lam_amd/ORBextractor.h"   // the mirror; types.h pulls conversions.h, which sees <opencv2/core.hpp>

#ifndef SE2LAM_AMD_HAVE_OPENCV
#error "the cv:: overloads of the mirror are needed here (include/se2lam_amd/conversions.h did not find <opencv2/core.hpp>)"
#endif

namespace se2lam {

namespace {
std::mutex g_mutex;
std::unordered_map<const ORBextractor*, std::unique_ptr<se2lam_amd::ORBextractor>>& table() {
    static std::unordered_map<const ORBextractor*, std::unique_ptr<se2lam_amd::ORBextractor>> t;
    return t;
}
se2lam_amd::ORBextractor& impl_of(const ORBextractor* self) {
    std::lock_guard<std::mutex> lock(g_mutex);
    return *table().at(self);
}
}  // namespace

// ORBextractor.h:44 - the five arguments of the reference; the members Frame::Frame reads through GetLevels() / GetScaleFactor()
// (src/Frame.cpp:47-48) are filled as the reference's constructor fills them (src/ORBextractor.cpp:463-468)
ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _scoreType, int _fastTh)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), scoreType(_scoreType), fastTh(_fastTh) {
    std::unique_ptr<se2lam_amd::ORBextractor> impl(new se2lam_amd::ORBextractor(_nfeatures, _scaleFactor, _nlevels, _scoreType, _fastTh));
    std::lock_guard<std::mutex> lock(g_mutex);
    table()[this] = std::move(impl);
}

// ORBextractor.h:49-51, src/ORBextractor.cpp:727-788
void ORBextractor::operator()(cv::InputArray _image, cv::InputArray _mask, std::vector<cv::KeyPoint>& _keypoints, cv::OutputArray _descriptors) {
    if (_image.empty()) return;   // src/ORBextractor.cpp:730-731
    impl_of(this)(_image, _mask, _keypoints, _descriptors);
}

// declared by the header (ORBextractor.h:62-63), called from nowhere outside the file this one replaces
void ORBextractor::ComputePyramid(cv::Mat, cv::Mat) {}
void ORBextractor::ComputeKeyPoints(std::vector<std::vector<cv::KeyPoint>>&) {}

}  // namespace se2lam
