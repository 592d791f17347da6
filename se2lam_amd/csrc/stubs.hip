// TEMPORARY: entry points not implemented yet (replaced by orb.hip / match.hip).
#include "common.h"
using namespace se2gpu;
#define NOTYET(name) do { set_error(name ": not implemented yet"); return SE2GPU_ERR_STATE; } while (0)
extern "C" {
int se2gpu_orb_create(const se2gpu_orb_params*, se2gpu_orb**) { NOTYET("orb_create"); }
void se2gpu_orb_destroy(se2gpu_orb*) {}
int se2gpu_orb_levels(const se2gpu_orb*) { return 0; }
float se2gpu_orb_scale_factor(const se2gpu_orb*) { return 0; }
int se2gpu_orb_extract(se2gpu_orb*, const uint8_t*, int, int, size_t, const uint8_t*, se2gpu_keypoint*, uint8_t*, int, int*) { NOTYET("orb_extract"); }
int se2gpu_orb_extract_batch_device(se2gpu_orb*, const uint8_t*, int, int, int, se2gpu_keypoint*, uint8_t*, int32_t*, int) { NOTYET("orb_extract_batch"); }
int se2gpu_orb_sync(se2gpu_orb*) { NOTYET("orb_sync"); }
int se2gpu_orb_set_stream(se2gpu_orb*, void*) { NOTYET("orb_set_stream"); }
int se2gpu_orb_debug_level(se2gpu_orb*, int, int, int, uint8_t*, size_t, int*, int*) { NOTYET("orb_debug_level"); }
int se2gpu_orb_debug_score(se2gpu_orb*, int, int, uint8_t*, size_t, int*, int*) { NOTYET("orb_debug_score"); }
void* se2gpu_orb_stream(se2gpu_orb*) { return nullptr; }
int se2gpu_orb_profile(se2gpu_orb*, int) { NOTYET("orb_profile"); }
int se2gpu_orb_profile_get(se2gpu_orb*, int, const char**, double*, int64_t*) { return SE2GPU_ERR_INVALID; }
int se2gpu_matcher_create(int, int, se2gpu_matcher**) { NOTYET("matcher_create"); }
void se2gpu_matcher_destroy(se2gpu_matcher*) {}
int se2gpu_matcher_set_stream(se2gpu_matcher*, void*) { NOTYET("matcher"); }
int se2gpu_matcher_sync(se2gpu_matcher*) { NOTYET("matcher"); }
void* se2gpu_matcher_stream(se2gpu_matcher*) { return nullptr; }
int se2gpu_match_window(se2gpu_matcher*, const se2gpu_frame_bounds*, const se2gpu_keypoint*, const uint8_t*, int, const se2gpu_keypoint*, const uint8_t*, int, float*, int, int, int, int, float, int32_t*, int*) { NOTYET("match_window"); }
int se2gpu_match_window_batch_device(se2gpu_matcher*, const se2gpu_frame_bounds*, const se2gpu_keypoint*, const uint8_t*, const int32_t*, int, const int32_t*, const int32_t*, int, int, int, int, int, float, int32_t*, int32_t*) { NOTYET("match_window_batch"); }
int se2gpu_match_projection(se2gpu_matcher*, const se2gpu_frame_bounds*, const float*, const uint8_t*, const int32_t*, const uint8_t*, int, const float*, float, float, float, float, const se2gpu_keypoint*, const uint8_t*, const uint8_t*, int, int, int, float, int32_t*, int*) { NOTYET("match_projection"); }
}
