// ORACLE - TEST INFRASTRUCTURE ONLY (oracle/_ref/libse2lam_pipeline_{cpu,dropin}.so).  Never linked, imported or called by the product path.
//
// BASELINE.json configs[0]: "synthetic 640x480 frames + SE(2) odometry through Track -> LocalMapper -> optimizer (ROS-stubbed)".
// C entry points that feed frames to the REFERENCE's own Track and LocalMapper, compiled from /root/reference where the sources lie
// (oracle/Makefile, target `pipeline`; nothing of the reference is copied into this repository).  Two builds share this file:
//   libse2lam_pipeline_cpu.so     every reference source, its own ORBextractor.cpp / ORBmatcher.cpp included; g2o's optimize() and
//                                 cv::findFundamentalMat - third-party libraries the image lacks - from oracle/pipeline_cpu_solver.cpp
//   libse2lam_pipeline_dropin.so  the same reference sources EXCEPT ORBextractor.cpp / ORBmatcher.cpp, whose place the bindings over
//                                 libse2gpu take (tests/dropin/ORBextractor_binding.cpp, ORBmatcher_binding.cpp); optimize() and findFundamentalMat go
//                                 to libse2gpu too (tests/dropin/g2o_forward.cpp).  No oracle restatement is linked into it.
// The reference's threads are loops around a few member calls (Track::run, src/Track.cpp:56-103; LocalMapper::run,
// src/LocalMapper.cpp:304-364) that poll ros::ok() and a mailbox; here the two loop bodies are called in turn for every frame,
// on one thread - the tracking step, then the mapper step if the tracker inserted a key frame - which makes a run deterministic
// and the two builds comparable frame by frame.  What runs inside is the reference's code: Frame::Frame, Track::mCreateFrame /
// mTrack (MatchByWindow, removeOutliers, updateFramePose, doTriangulate, needNewKF), LocalMapper::addNewKF / findCorrespd
// (MatchByProjection), Map::updateLocalGraph / pruneRedundantKF / loadLocalGraph / optimizeLocalGraph, LocalMapper::localBA.
// The members these functions work on are private in the reference's headers; this translation unit reads the headers with
// `private` / `protected` spelled `public` (no reference source is touched).
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <new>
#include <cstring>
#include <deque>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include <opencv2/core/core.hpp>
#include <g2o_shim.hpp>

#define private public
#define protected public
#include "KeyFrame.h"
#include "MapPoint.h"
#include "Map.h"
#include "Track.h"
#include "LocalMapper.h"
#undef private
#undef protected
#include "ORBmatcher.h"
#include "converter.h"
#include "cvutil.h"

using namespace se2lam;

// ---- creation order = address order for the map's objects.  The reference orders a map point's observations by the ADDRESS of the
// key frame (std::map<PtrKeyFrame, int>, include/se2lam/MapPoint.h:86; likewise std::map<PtrMapPoint, int> in KeyFrame.h and the
// std::set<PtrMapPoint> / std::set<PtrKeyFrame> its functions return), and MapPoint::updateMainKFandDescriptor
// (src/MapPoint.cpp:228-292) breaks the ties of its least-median rule by that order - with two observations, always.  Which key
// frame lies lower in memory is the allocator's business: the CPU build, which allocates image pyramids in between, and the drop-in
// build, which does not, get different orders from malloc and would then disagree about main key frames, main descriptors and, a few
// frames later, map points (observed: key-frame address ranks [0 2 1] against [0 1 2], the first difference in the map at the third
// key frame).  Both builds therefore take the storage of key frames and map points (std::make_shared: the object plus a control
// block) from a bump arena that is never reused: a younger object always lies higher, in every run.  This pins an
// implementation-defined behaviour of the reference for the comparison, as oracle/stl_nth.h pins std::nth_element; nothing of the
// library under test is involved.
namespace {
struct BumpArena {
    char* cur = nullptr;
    char* end = nullptr;
    std::vector<std::pair<char*, char*>> chunks;
    std::mutex mu;
    void* take(std::size_t n) {
        std::lock_guard<std::mutex> lk(mu);
        n = (n + 63) & ~std::size_t(63);
        if (!cur || cur + n > end) {
            const std::size_t sz = std::max<std::size_t>(n, std::size_t(32) << 20);
            cur = static_cast<char*>(std::malloc(sz));
            if (!cur) throw std::bad_alloc();
            end = cur + sz;
            // (chunks come from malloc in ascending order or not - objects of one run stay ordered as long as they share a chunk; a
            // chunk holds ~400 key frames, far more than any run here creates)
            chunks.emplace_back(cur, end);
        }
        void* p = cur;
        cur += n;
        return p;
    }
    bool owns(const void* p) {
        std::lock_guard<std::mutex> lk(mu);
        for (const auto& c : chunks)
            if (p >= c.first && p < c.second) return true;
        return false;
    }
};
BumpArena& arena() { static BumpArena* a = new (std::malloc(sizeof(BumpArena))) BumpArena; return *a; }
inline bool map_object_size(std::size_t n) {
    return (n >= sizeof(KeyFrame) && n <= sizeof(KeyFrame) + 64) || (n >= sizeof(MapPoint) && n <= sizeof(MapPoint) + 64);
}
}  // namespace
void* operator new(std::size_t n) {
    if (map_object_size(n)) return arena().take(n);
    void* p = std::malloc(n ? n : 1);
    if (!p) throw std::bad_alloc();
    return p;
}
void operator delete(void* p) noexcept {
    if (p && !arena().owns(p)) std::free(p);
}
void operator delete(void* p, std::size_t) noexcept {
    if (p && !arena().owns(p)) std::free(p);
}

extern "C" {
const char* pipeline_kind(void);          // oracle/pipeline_cpu_solver.cpp or tests/dropin/g2o_forward.cpp
void pipeline_install_hooks(void);
void pipeline_last_ba(double out[10]);
}

namespace {
thread_local std::string g_error;

struct Pipe {
    Map map;
    Track* track = nullptr;           // Track's extractor is never freed by the reference either (src/Track.cpp:34)
    LocalMapper mapper;
    int frames = 0, bas = 0;
};

uint64_t fnv1a(const void* p, size_t n, uint64_t h = 1469598103934665603ull) {
    const uint8_t* b = static_cast<const uint8_t*>(p);
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}
double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

cv::Mat mat_of(const float* v, int rows, int cols) {
    cv::Mat m(rows, cols, CV_32FC1);
    for (int r = 0; r < rows; ++r) for (int c = 0; c < cols; ++c) m.at<float>(r, c) = v[r * cols + c];
    return m;
}
}  // namespace

extern "C" {

struct ref_pipe_config {   // what Config::readConfig (src/Config.cpp:83-186) would have read from CamConfig.yml / Settings.yml
    float K[9];            // camera matrix
    float bTc[16];         // extrinsic, body <- camera
    float upper_depth, lower_depth;
    float scale_factor; int32_t max_level, max_features;
    float odo_noise[3], odo_uncertain[3];
    float planemotion_z_info, planemotion_xrot_info, planemotion_yrot_info;
    float th_huber2; int32_t local_iter; int32_t fps;
};

struct ref_pipe_frame {    // what one call of ref_pipe_feed did
    int32_t frame_id, n_keypoints, n_raw_matches, n_matches, new_kf, local_ba, n_kfs, n_mps, n_good_prl, n_local_kfs, n_local_mps, n_ref_kfs;
    int32_t n_match_entries, n_raw_entries;   // entries written to match_idx / raw_matches (the reference frame's key-point count; 0 after a key-frame insertion / none)
    uint64_t kp_hash, desc_hash;
    double ms_track, ms_mapper;
    double ba[10];         // {P, L, E, O, chi2 at the start, at the end, iterations, trials, lambda, stopped} of this frame's localBA
    float Twb[3], Tcw[16];
};

const char* ref_pipe_last_error(void) { return g_error.c_str(); }
const char* ref_pipe_kind(void) { return pipeline_kind(); }
void ref_pipe_shim_calls(long long out[4]) { std::memcpy(out, cv::shim_call_counts(), 4 * sizeof(long long)); }

void* ref_pipe_create(const ref_pipe_config* c) {
    try {
        Config::Kcam = mat_of(c->K, 3, 3);
        Config::fxCam = c->K[0];
        Config::fyCam = c->K[4];
        Config::Dcam = cv::Mat::zeros(4, 1, CV_32FC1);
        Config::bTc = mat_of(c->bTc, 4, 4);
        Config::cTb = cvu::inv(Config::bTc);
        Config::PrjMtrxEye = Config::Kcam * cv::Mat::eye(3, 4, CV_32FC1);       // src/Config.cpp:123
        Config::ImgSize = cv::Size(640, 480);
        Config::UPPER_DEPTH = c->upper_depth;
        Config::LOWER_DEPTH = c->lower_depth;
        Config::ScaleFactor = c->scale_factor;
        Config::MaxLevel = c->max_level;
        Config::MaxFtrNumber = c->max_features;
        Config::ODO_X_NOISE = c->odo_noise[0]; Config::ODO_Y_NOISE = c->odo_noise[1]; Config::ODO_T_NOISE = c->odo_noise[2];
        Config::ODO_X_UNCERTAIN = c->odo_uncertain[0]; Config::ODO_Y_UNCERTAIN = c->odo_uncertain[1]; Config::ODO_T_UNCERTAIN = c->odo_uncertain[2];
        Config::PLANEMOTION_Z_INFO = c->planemotion_z_info;
        Config::PLANEMOTION_XROT_INFO = c->planemotion_xrot_info;
        Config::PLANEMOTION_YROT_INFO = c->planemotion_yrot_info;
        Config::TH_HUBER = std::sqrt(c->th_huber2);                               // src/Config.cpp:155
        Config::LOCAL_ITER = c->local_iter;
        Config::LOCAL_VERBOSE = false;
        Config::LOCAL_PRINT = false;
        Config::FPS = c->fps;
        Config::LOCALIZATION_ONLY = false;
        Config::USE_PREV_MAP = false;
        // the process-wide counters of the data model start over (a second pipeline in one process is a fresh run)
        Frame::nextId = 0;
        Frame::mbInitialComputations = true;
        KeyFrame::mNextIdKF = 0;
        MapPoint::mNextId = 0;
        pipeline_install_hooks();
        Pipe* p = new Pipe;
        p->track = new Track;                                                      // OdoSLAM::start, src/OdoSLAM.cpp:99-117
        p->track->setMap(&p->map);
        p->track->setLocalMapper(&p->mapper);
        p->mapper.setMap(&p->map);
        return p;
    } catch (const std::exception& e) {
        g_error = e.what();
        return nullptr;
    }
}

void ref_pipe_destroy(void* h) {
    Pipe* p = static_cast<Pipe*>(h);
    if (!p) return;
    delete p->track;
    delete p;
}

// One image + one odometry reading.  match_idx (cap entries): Track::mMatchIdx after the frame (MatchByWindow, then the epipolar
// filter, then the depth gate of doTriangulate); raw_matches (cap entries, may be NULL): what MatchByWindow itself returned for the
// frame - the call is repeated on copies of (mRefFrame, mPrevMatched) taken before the frame, outside the timed section.
// Returns 0, or -1 on an exception (ref_pipe_last_error).
int ref_pipe_feed(void* h, const uint8_t* img, int rows, int cols, const float* odo3, ref_pipe_frame* out, int32_t* match_idx,
                  int32_t* raw_matches, int cap) {
    Pipe* p = static_cast<Pipe*>(h);
    try {
        Track& t = *p->track;
        cv::Mat image(rows, cols, CV_8UC1);
        for (int r = 0; r < rows; ++r) std::memcpy(image.ptr<uint8_t>(r), img + (size_t)r * cols, (size_t)cols);
        const Se2 odo(odo3[0], odo3[1], odo3[2]);
        std::memset(out, 0, sizeof(*out));
        const bool first = !(Frame::nextId);
        Frame ref_before;
        std::vector<cv::Point2f> prev_before;
        if (!first && raw_matches) { ref_before = t.mRefFrame; prev_before = t.mPrevMatched; }
        const size_t kfs_before = p->map.countKFs();

        // ---- the body of Track::run's loop (src/Track.cpp:73-89)
        const double t0 = now_ms();
        {
            std::lock_guard<std::mutex> lock(t.mMutexForPub);
            if (first) t.mCreateFrame(image, odo);
            else t.mTrack(image, odo);
        }
        p->map.setCurrentFramePose(t.mFrame.Tcw);
        t.lastOdom = odo;
        const double t1 = now_ms();

        // ---- the body of LocalMapper::run's loop (src/LocalMapper.cpp:320-352); the global mapper's thread is not started
        // (mpGlobalMapper->waitIfBusy() would return at once)
        LocalMapper& m = p->mapper;
        if (m.mbUpdated) {
            m.updateLocalGraphInMap();
            m.pruneRedundantKfInMap();
            m.updateLocalGraphInMap();
            m.localBA();
            m.mbUpdated = false;
            m.updateLocalGraphInMap();
            out->local_ba = 1;
            pipeline_last_ba(out->ba);
            ++p->bas;
        }
        m.mbAcceptNewKF = true;
        const double t2 = now_ms();

        out->ms_track = t1 - t0;
        out->ms_mapper = t2 - t1;
        out->frame_id = t.mFrame.id;
        out->n_keypoints = t.mFrame.N;
        out->kp_hash = t.mFrame.keyPoints.empty() ? 0 : fnv1a(t.mFrame.keyPoints.data(), t.mFrame.keyPoints.size() * sizeof(cv::KeyPoint));
        uint64_t dh = 1469598103934665603ull;
        for (int r = 0; r < t.mFrame.descriptors.rows; ++r) dh = fnv1a(t.mFrame.descriptors.ptr<uint8_t>(r), 32, dh);
        out->desc_hash = dh;
        out->new_kf = p->map.countKFs() != kfs_before || (first && p->map.countKFs() > 0) ? 1 : 0;
        out->n_kfs = (int)p->map.countKFs();
        out->n_mps = (int)p->map.countMPs();
        out->n_good_prl = t.mnGoodPrl;
        out->n_local_kfs = p->map.countLocalKFs();
        out->n_local_mps = p->map.countLocalMPs();
        out->n_ref_kfs = (int)p->map.getRefKFs().size();
        out->Twb[0] = t.mFrame.Twb.x; out->Twb[1] = t.mFrame.Twb.y; out->Twb[2] = t.mFrame.Twb.theta;
        if (!t.mFrame.Tcw.empty()) for (int i = 0; i < 16; ++i) out->Tcw[i] = t.mFrame.Tcw.at<float>(i / 4, i % 4);
        if (!first) {
            // a key frame inserted by this frame has already reset the local track (mMatchIdx cleared, src/Track.cpp:196)
            const int n = (int)t.mMatchIdx.size();
            out->n_match_entries = n < cap ? n : cap;
            for (int i = 0; i < out->n_match_entries; ++i) { match_idx[i] = t.mMatchIdx[i]; if (t.mMatchIdx[i] >= 0) ++out->n_matches; }
            if (raw_matches) {
                std::vector<int> raw;
                ORBmatcher matcher(0.9);                                            // src/Track.cpp:131-132
                out->n_raw_matches = matcher.MatchByWindow(ref_before, t.mFrame, prev_before, 20, raw);
                out->n_raw_entries = (int)raw.size() < cap ? (int)raw.size() : cap;
                for (int i = 0; i < out->n_raw_entries; ++i) raw_matches[i] = raw[i];
            }
        }
        ++p->frames;
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return -1;
    }
}

// the map as it stands: key frames (id, mIdKF, body pose Twb as the float Se2 the reference keeps, Tcw, observations) ...
int ref_pipe_keyframes(void* h, int cap, int32_t* id, int32_t* id_kf, float* Twb3, float* Tcw16, int32_t* n_obs) {
    Pipe* p = static_cast<Pipe*>(h);
    const std::vector<PtrKeyFrame> kfs = p->map.getAllKF();
    int n = 0;
    for (const PtrKeyFrame& k : kfs) {
        if (n < cap) {
            id[n] = k->id; id_kf[n] = k->mIdKF;
            Twb3[3 * n] = k->Twb.x; Twb3[3 * n + 1] = k->Twb.y; Twb3[3 * n + 2] = k->Twb.theta;
            for (int i = 0; i < 16; ++i) Tcw16[16 * n + i] = k->Tcw.at<float>(i / 4, i % 4);
            n_obs[n] = k->getSizeObsMP();
        }
        ++n;
    }
    return n;
}
// ... and map points (id, world position, observation count, good-parallax flag)
int ref_pipe_mappoints(void* h, int cap, int32_t* id, float* pos3, int32_t* n_obs, uint8_t* good_prl) {
    Pipe* p = static_cast<Pipe*>(h);
    const std::vector<PtrMapPoint> mps = p->map.getAllMP();
    int n = 0;
    for (const PtrMapPoint& m : mps) {
        if (n < cap) {
            id[n] = m->mId;
            const cv::Point3f x = m->getPos();
            pos3[3 * n] = x.x; pos3[3 * n + 1] = x.y; pos3[3 * n + 2] = x.z;
            n_obs[n] = (int)m->countObservation();
            good_prl[n] = m->isGoodPrl() ? 1 : 0;
        }
        ++n;
    }
    return n;
}

// the heap addresses of the key-frame objects (ref_pipe_keyframes' order).  The reference keeps a map point's observations in a
// std::map<PtrKeyFrame, int> (include/se2lam/MapPoint.h:86), i.e. ordered by ADDRESS, and MapPoint::updateMainKFandDescriptor
// (src/MapPoint.cpp:228-292) breaks the ties of its least-median rule by that order: with two observations both medians are 0 and the
// key frame at the lower address becomes the main one.  Two runs agree on such ties only if their allocators place the key frames in
// the same relative order - the test compares this order before it compares anything that depends on it.
int ref_pipe_keyframe_addresses(void* h, int cap, uint64_t* addr) {
    Pipe* p = static_cast<Pipe*>(h);
    const std::vector<PtrKeyFrame> kfs = p->map.getAllKF();
    int n = 0;
    for (const PtrKeyFrame& k : kfs) {
        if (n < cap) addr[n] = (uint64_t)(uintptr_t)k.get();
        ++n;
    }
    return n;
}

// the observations of one key frame (position in ref_pipe_keyframes' order): feature index -> map-point id, ascending feature index
int ref_pipe_observations(void* h, int kf_pos, int cap, int32_t* ftr, int32_t* mp_id) {
    Pipe* p = static_cast<Pipe*>(h);
    const std::vector<PtrKeyFrame> kfs = p->map.getAllKF();
    if (kf_pos < 0 || kf_pos >= (int)kfs.size()) return -1;
    int n = 0;
    for (const auto& kv : kfs[kf_pos]->mDualObservations) {
        if (n < cap) { ftr[n] = kv.first; mp_id[n] = kv.second ? kv.second->mId : -1; }
        ++n;
    }
    return n;
}

}  // extern "C"
