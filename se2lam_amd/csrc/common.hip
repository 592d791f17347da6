// libse2gpu: error channel, device helpers, event timers.
#include "common.h"

namespace se2gpu {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

bool have_device() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return n > 0;
}

}  // namespace se2gpu

using namespace se2gpu;

extern "C" {

const char* se2gpu_last_error(void) { return g_err.c_str(); }

int se2gpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

const char* se2gpu_version(void) { return "se2gpu 0.1 (gfx950, HIP)"; }

int se2gpu_hamming(const uint8_t* a, const uint8_t* b) {
    if (!a || !b) return SE2GPU_ERR_INVALID;
    int d = 0;
    for (int i = 0; i < 4; ++i) {
        uint64_t x, y;
        std::memcpy(&x, a + 8 * i, 8);
        std::memcpy(&y, b + 8 * i, 8);
        d += __builtin_popcountll(x ^ y);
    }
    return d;
}

struct se2gpu_timer {
    hipEvent_t e0 = nullptr, e1 = nullptr;
};

int se2gpu_timer_create(se2gpu_timer** out) {
    SE2_REQUIRE(out, SE2GPU_ERR_INVALID, "timer_create: out is NULL");
    SE2_REQUIRE(have_device(), SE2GPU_ERR_NO_DEVICE, "no HIP device visible");
    se2gpu_timer* t = new se2gpu_timer;
    if (hipEventCreate(&t->e0) != hipSuccess || hipEventCreate(&t->e1) != hipSuccess) {
        delete t;
        set_error("hipEventCreate failed");
        return SE2GPU_ERR_HIP;
    }
    *out = t;
    return SE2GPU_OK;
}
void se2gpu_timer_destroy(se2gpu_timer* t) {
    if (!t) return;
    if (t->e0) (void)hipEventDestroy(t->e0);
    if (t->e1) (void)hipEventDestroy(t->e1);
    delete t;
}
int se2gpu_timer_start(se2gpu_timer* t, void* s) {
    SE2_REQUIRE(t, SE2GPU_ERR_INVALID, "timer is NULL");
    SE2_HIP(hipEventRecord(t->e0, (hipStream_t)s));
    return SE2GPU_OK;
}
int se2gpu_timer_stop(se2gpu_timer* t, void* s) {
    SE2_REQUIRE(t, SE2GPU_ERR_INVALID, "timer is NULL");
    SE2_HIP(hipEventRecord(t->e1, (hipStream_t)s));
    return SE2GPU_OK;
}
int se2gpu_timer_elapsed_ms(se2gpu_timer* t, float* ms) {
    SE2_REQUIRE(t && ms, SE2GPU_ERR_INVALID, "timer/ms is NULL");
    SE2_HIP(hipEventSynchronize(t->e1));
    SE2_HIP(hipEventElapsedTime(ms, t->e0, t->e1));
    return SE2GPU_OK;
}

int se2gpu_malloc(void** p, size_t bytes) {
    SE2_REQUIRE(p, SE2GPU_ERR_INVALID, "malloc: p is NULL");
    SE2_REQUIRE(have_device(), SE2GPU_ERR_NO_DEVICE, "no HIP device visible");
    SE2_HIP(hipMalloc(p, bytes ? bytes : 1));
    return SE2GPU_OK;
}
int se2gpu_free(void* p) {
    if (p) SE2_HIP(hipFree(p));
    return SE2GPU_OK;
}
int se2gpu_memcpy_h2d(void* dst, const void* src, size_t bytes) {
    SE2_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return SE2GPU_OK;
}
int se2gpu_memcpy_d2h(void* dst, const void* src, size_t bytes) {
    SE2_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return SE2GPU_OK;
}
int se2gpu_host_alloc(void** p, size_t bytes) {
    SE2_REQUIRE(p, SE2GPU_ERR_INVALID, "host_alloc: p is NULL");
    SE2_REQUIRE(have_device(), SE2GPU_ERR_NO_DEVICE, "no HIP device visible");
    SE2_HIP(hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocDefault));
    return SE2GPU_OK;
}
int se2gpu_host_free(void* p) {
    if (p) SE2_HIP(hipHostFree(p));
    return SE2GPU_OK;
}
int se2gpu_memcpy_h2d_async(void* dst, const void* src, size_t bytes, void* s) {
    SE2_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)s));
    return SE2GPU_OK;
}
int se2gpu_memcpy_d2h_async(void* dst, const void* src, size_t bytes, void* s) {
    SE2_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)s));
    return SE2GPU_OK;
}
int se2gpu_stream_create(void** out) {
    SE2_REQUIRE(out, SE2GPU_ERR_INVALID, "stream_create: out is NULL");
    SE2_REQUIRE(have_device(), SE2GPU_ERR_NO_DEVICE, "no HIP device visible");
    hipStream_t s = nullptr;
    SE2_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out = (void*)s;
    return SE2GPU_OK;
}
int se2gpu_stream_destroy(void* s) {
    if (s) SE2_HIP(hipStreamDestroy((hipStream_t)s));
    return SE2GPU_OK;
}
int se2gpu_stream_synchronize(void* s) {
    SE2_HIP(hipStreamSynchronize((hipStream_t)s));
    return SE2GPU_OK;
}
int se2gpu_timer_stream_wait(se2gpu_timer* t, void* s) {
    SE2_REQUIRE(t, SE2GPU_ERR_INVALID, "timer is NULL");
    SE2_HIP(hipStreamWaitEvent((hipStream_t)s, t->e1, 0));
    return SE2GPU_OK;
}
int se2gpu_device_synchronize(void) {
    SE2_HIP(hipDeviceSynchronize());
    return SE2GPU_OK;
}
int se2gpu_set_device(int ordinal) {
    SE2_REQUIRE(have_device(), SE2GPU_ERR_NO_DEVICE, "no HIP device visible");
    SE2_HIP(hipSetDevice(ordinal));
    return SE2GPU_OK;
}

}  // extern "C"
