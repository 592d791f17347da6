// Shared host-side utilities of libse2gpu (error channel, device buffers, launch profiling).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/se2gpu.h"

namespace se2gpu {

void set_error(const char* fmt, ...);
bool have_device();
// in-place sum all-reduce of `count` doubles through a native RCCL communicator (comm.hip); 0 on success
int comm_allreduce(se2gpu_comm* c, void* dev_ptr, size_t count, void* hip_stream);
int comm_rank(const se2gpu_comm* c);
int comm_world(const se2gpu_comm* c);

#define SE2_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            ::se2gpu::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return SE2GPU_ERR_HIP;                                                             \
        }                                                                                      \
    } while (0)

#define SE2_CHECK(expr)                 \
    do {                                \
        int _rc = (expr);               \
        if (_rc != SE2GPU_OK) return _rc; \
    } while (0)

#define SE2_REQUIRE(cond, code, ...)          \
    do {                                      \
        if (!(cond)) {                        \
            ::se2gpu::set_error(__VA_ARGS__); \
            return (code);                    \
        }                                     \
    } while (0)

// Growable device allocation (never shrinks).  Not copyable.
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;  // elements
    bool owned = true;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p && owned) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        owned = true;
    }
    // view into somebody else's allocation (an arena): never freed here, replaced by the next alias() / reserve()
    void alias(T* q) {
        release();
        p = q;
        owned = false;
    }
    int reserve(size_t n) {
        if (n <= cap) return SE2GPU_OK;
        release();
        if (n == 0) return SE2GPU_OK;
        SE2_HIP(hipMalloc((void**)&p, n * sizeof(T)));
        cap = n;
        return SE2GPU_OK;
    }
    int upload(const T* src, size_t n, hipStream_t s) {
        SE2_CHECK(reserve(n));
        if (n) SE2_HIP(hipMemcpyAsync(p, src, n * sizeof(T), hipMemcpyHostToDevice, s));
        return SE2GPU_OK;
    }
    int upload(const std::vector<T>& v, hipStream_t s) { return upload(v.data(), v.size(), s); }
};

// Pinned host staging buffer.
template <typename T>
struct PinBuf {
    T* p = nullptr;
    size_t cap = 0;
    PinBuf() = default;
    PinBuf(const PinBuf&) = delete;
    PinBuf& operator=(const PinBuf&) = delete;
    ~PinBuf() {
        if (p) (void)hipHostFree(p);
    }
    int reserve(size_t n) {
        if (n <= cap) return SE2GPU_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        SE2_HIP(hipHostMalloc((void**)&p, n * sizeof(T), hipHostMallocDefault));
        cap = n;
        return SE2GPU_OK;
    }
};

// Optional per-kernel timing with HIP events around every launch (serialises; for bench.py's
// `roofline.achieved`, which must be measured on the stream the kernels are launched on).
struct LaunchProfile {
    struct Slot {
        const char* name;
        double ms = 0;
        int64_t launches = 0;
    };
    bool enabled = false;
    std::vector<Slot> slots;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~LaunchProfile() {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    }
    int slot(const char* name) {
        for (size_t i = 0; i < slots.size(); ++i)
            if (slots[i].name == name || std::strcmp(slots[i].name, name) == 0) return (int)i;
        slots.push_back(Slot{name});
        return (int)slots.size() - 1;
    }
    void begin(hipStream_t s) {
        if (!enabled) return;
        if (!e0) {
            (void)hipEventCreate(&e0);
            (void)hipEventCreate(&e1);
        }
        (void)hipEventRecord(e0, s);
    }
    void end(hipStream_t s, const char* name) {
        if (!enabled) return;
        (void)hipEventRecord(e1, s);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        Slot& sl = slots[slot(name)];
        sl.ms += ms;
        sl.launches += 1;
    }
    void reset() {
        for (auto& s : slots) {
            s.ms = 0;
            s.launches = 0;
        }
    }
};

#define SE2_LAUNCH(prof, stream, name, kernel, grid, block, shmem, ...)         \
    do {                                                                        \
        (prof).begin(stream);                                                   \
        hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);    \
        (prof).end(stream, name);                                               \
    } while (0)

}  // namespace se2gpu
