// ORACLE - TEST INFRASTRUCTURE ONLY (oracle/_ref).  The map the two drivers of libse2lam_ref_map.so share.
#pragma once
#include <vector>
#include "Map.h"
struct RefMap {
    se2lam::Map map;
    std::vector<se2lam::PtrKeyFrame> kfs;
    std::vector<se2lam::PtrMapPoint> mps;
};
