// TEMPORARY: entry points not implemented yet (replaced by orb.hip / match.hip).
#include "common.h"
using namespace se2gpu;
#define NOTYET(name) do { set_error(name ": not implemented yet"); return SE2GPU_ERR_STATE; } while (0)
extern "C" {
int se2gpu_matcher_create(int, int, se2gpu_matcher**) { NOTYET("matcher_create"); }
void se2gpu_matcher_destroy(se2gpu_matcher*) {}
int se2gpu_matcher_set_stream(se2gpu_matcher*, void*) { NOTYET("matcher"); }
int se2gpu_matcher_sync(se2gpu_matcher*) { NOTYET("matcher"); }
void* se2gpu_matcher_stream(se2gpu_matcher*) { return nullptr; }
int se2gpu_match_window(se2gpu_matcher*, const se2gpu_frame_bounds*, const se2gpu_keypoint*, const uint8_t*, int, const se2gpu_keypoint*, const uint8_t*, int, float*, int, int, int, int, float, int32_t*, int*) { NOTYET("match_window"); }
int se2gpu_match_window_batch_device(se2gpu_matcher*, const se2gpu_frame_bounds*, const se2gpu_keypoint*, const uint8_t*, const int32_t*, int, const int32_t*, const int32_t*, int, int, int, int, int, float, int32_t*, int32_t*) { NOTYET("match_window_batch"); }
int se2gpu_match_projection(se2gpu_matcher*, const se2gpu_frame_bounds*, const float*, const uint8_t*, const int32_t*, const uint8_t*, int, const float*, float, float, float, float, const se2gpu_keypoint*, const uint8_t*, const uint8_t*, int, int, int, float, int32_t*, int*) { NOTYET("match_projection"); }
}
