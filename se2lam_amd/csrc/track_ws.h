// Device workspace of one tracking thread, shared by ransac.hip (removeOutliers) and triangulate.hip (doTriangulate).
#pragma once
#include "common.h"

struct se2gpu_track {
    hipStream_t stream = nullptr;
    se2gpu::PinBuf<uint8_t> h_in, h_out;
    se2gpu::DevBuf<uint8_t> d_in, d_out;
    se2gpu::DevBuf<double> d_F, d_score;
    se2gpu::DevBuf<int> d_nm;
    std::vector<int> subsets;   // RANSAC sample indices; depend on the point count only
    int subsets_n = -1;
    std::vector<float> pt1, pt2;
    std::vector<int> idx;
    std::vector<uint8_t> mask;
    int last_info[4] = {0, -1, -1, 0};
};
