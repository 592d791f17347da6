// ORACLE - TEST INFRASTRUCTURE ONLY (oracle/_ref, second library: libse2lam_ref_map.so).  Pre-included (-include) in front of
// every reference source file of the MAP build, where src/Map.cpp, src/KeyFrame.cpp and src/MapPoint.cpp are compiled with
// their own headers, and with them the threads - src/Track.cpp, src/LocalMapper.cpp, src/GlobalMapper.cpp, src/Localizer.cpp,
// src/Sensors.cpp -, src/sparsifier.cpp and the vendored DBoW2 (Thirdparty/DBoW2: the vocabulary template, FORB, the scoring
// objects), src/MapStorage.cpp and src/OdoSLAM.cpp.  What is cut off: ROS (ros::Rate / ros::ok in the run() loops, never entered) and the two
// publishers (FramePublish.cpp, MapPublish.cpp: their members that OdoSLAM.cpp names are empty functions in ref_system_driver.cpp).
#pragma once
#include <climits>
#include <deque>
#include <fstream>
#include <map>
#include <memory>
#include <set>
#include <vector>

#include <opencv2/core/core.hpp>
#include <g2o_shim.hpp>

#include "Config.h"
#include "Thirdparty/DBoW2/DBoW2/BowVector.h"
#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"

using namespace std;   // Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:38 does this at global scope, and KeyFrame.h relies on it
