// ORACLE - TEST INFRASTRUCTURE ONLY (oracle/_ref).  Never linked, imported or called by the product path.
//
// /root/reference/src/OdoSLAM.cpp compiled unmodified (part of libse2lam_ref_map.so) for ONE of its functions: OdoSLAM::saveMap, whose
// second half writes the key-frame trajectory file (`id x y z yaw` per key frame that is not null, :198-212) - the text
// include/se2lam_amd/MapStorage.h's trajectoryText mirrors.  The publishers it owns (FramePublish, MapPublish: ROS) are cut off:
// their members that OdoSLAM.cpp names are defined here as empty functions; start() / wait() are compiled and never called.
// `private` is spelled `public` in this translation unit only, to reach OdoSLAM::mpMap and saveMap.
#include <cstring>
#include <string>

#define private public
#include "OdoSLAM.h"
#undef private
#include "ref_map_state.h"

namespace se2lam {
FramePublish::FramePublish() {}
FramePublish::FramePublish(Track*, GlobalMapper*) {}
FramePublish::~FramePublish() {}
void FramePublish::run() {}
void FramePublish::setLocalizer(Localizer*) {}
MapPublish::MapPublish(Map* pMap) : mpMap(pMap) {}
MapPublish::~MapPublish() {}
void MapPublish::run() {}
void MapPublish::setFramePub(FramePublish*) {}
void MapPublish::RequestFinish() {}
bool MapPublish::isFinished() { return true; }
}  // namespace se2lam

using namespace se2lam;

extern "C" {

// KeyFrame::id (the frame id the trajectory file prints; MapStorage does not store it)
void ref_system_kf_set_id(void* h, int kf, int id) { static_cast<RefMap*>(h)->kfs[kf]->id = id; }

// OdoSLAM::saveMap() with Config::SAVE_NEW_MAP off: the trajectory of the handle's map into <dir>/se2lam_kf_trajectory.txt
int ref_system_save_trajectory(void* h, const float* bTc16, const char* dir) {
    try {
        RefMap* m = static_cast<RefMap*>(h);
        Config::SAVE_NEW_MAP = false;
        Config::WRITE_MAP_FILE_PATH = dir;
        cv::Mat T(4, 4, CV_32FC1);
        std::memcpy(T.data, bTc16, 16 * sizeof(float));
        Config::bTc = T;
        OdoSLAM sys;
        sys.mpMap = &m->map;
        sys.mpMapStorage = nullptr; sys.mpMapPub = nullptr; sys.mpLocalizer = nullptr; sys.mpTrack = nullptr; sys.mpLocalMapper = nullptr;
        sys.mpGlobalMapper = nullptr; sys.mpFramePub = nullptr; sys.mpSensors = nullptr;
        sys.saveMap();
        sys.mpMap = nullptr;          // (the destructor deletes what it owns: nothing here)
        return 0;
    } catch (...) { return -1; }
}

}  // extern "C"
