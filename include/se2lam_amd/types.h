// POD mirrors of the OpenCV / g2o value types that cross the reference's hot-path interfaces.
// When OpenCV / g2o headers are available a maintainer replaces these with the real types (the layouts are
// identical: see INTEGRATION.md); this image has neither, so the adapters are written against the mirrors.
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../se2gpu.h"

namespace se2lam_amd {

struct Point2f {          // cv::Point2f
    float x = 0, y = 0;
};

struct KeyPoint {         // cv::KeyPoint (28 bytes; layout == se2gpu_keypoint)
    Point2f pt;
    float size = 0, angle = -1, response = 0;
    int octave = 0, class_id = -1;
};
static_assert(sizeof(KeyPoint) == sizeof(se2gpu_keypoint), "cv::KeyPoint layout");

struct Mat8U {            // a CV_8UC1 cv::Mat view (rows x cols, row pitch `step`), or an owning n x 32 descriptor matrix
    int rows = 0, cols = 0;
    size_t step = 0;
    const uint8_t* data = nullptr;
    std::vector<uint8_t> owned;
    bool empty() const { return rows == 0 || cols == 0 || data == nullptr; }
    void create(int r, int c) {
        owned.assign((size_t)r * c, 0);
        rows = r; cols = c; step = (size_t)c; data = owned.data();
    }
    uint8_t* ptr(int r) { return owned.data() + (size_t)r * step; }
    const uint8_t* ptr(int r) const { return data + (size_t)r * step; }
};

struct SE2 { double x = 0, y = 0, theta = 0; };              // g2o::SE2 (translation + angle)
struct Vector2D { double v[2]; };                           // g2o::Vector2D
struct Vector3D { double v[3]; };                           // g2o::Vector3D
struct Matrix2D { double m[4]; };                           // row-major 2x2
struct Matrix3D { double m[9]; };                           // row-major 3x3
struct SE3Quat { double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; double t[3] = {0, 0, 0}; };  // rotation matrix + translation

// The reference has no error channel on these surfaces (asserts only); the adapters throw on a library error so a
// failure cannot pass silently.  Nothing is thrown across the C ABI itself.
inline void check(int rc, const char* what) {
    if (rc != SE2GPU_OK) throw std::runtime_error(std::string(what) + ": " + se2gpu_last_error());
}

}  // namespace se2lam_amd
