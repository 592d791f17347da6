// libse2gpu - native RCCL communicator (xGMI) for the landmark-sharded bundle adjustment.
// librccl.so.1 is loaded with dlopen on first use so that single-GPU users do not depend on it.
#include <dlfcn.h>

#include "common.h"

using namespace se2gpu;

namespace {

typedef struct { char internal[128]; } nccl_unique_id;   // ncclUniqueId (rccl.h:43)
typedef int (*fn_get_unique_id)(nccl_unique_id*);
typedef int (*fn_comm_init_rank)(void**, int, nccl_unique_id, int);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_comm_destroy)(void*);
typedef int (*fn_comm_count)(const void*, int*);
typedef const char* (*fn_error_string)(int);

struct Rccl {
    void* lib = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_comm_count comm_count = nullptr;
    fn_error_string error_string = nullptr;
};

Rccl* rccl() {
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1"};  // never an unversioned name: a host
        // process may already hold a different RCCL build under "librccl.so" (PyTorch bundles one)
        for (const char* n : names) {
            r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (r.lib) {
            r.get_unique_id = (fn_get_unique_id)dlsym(r.lib, "ncclGetUniqueId");
            r.comm_init_rank = (fn_comm_init_rank)dlsym(r.lib, "ncclCommInitRank");
            r.all_reduce = (fn_all_reduce)dlsym(r.lib, "ncclAllReduce");
            r.comm_destroy = (fn_comm_destroy)dlsym(r.lib, "ncclCommDestroy");
            r.error_string = (fn_error_string)dlsym(r.lib, "ncclGetErrorString");
            r.comm_count = (fn_comm_count)dlsym(r.lib, "ncclCommCount");
        }
    }
    if (!r.lib || !r.get_unique_id || !r.comm_init_rank || !r.all_reduce || !r.comm_destroy) return nullptr;
    return &r;
}

constexpr int kNcclDouble = 8;  // ncclFloat64 / ncclDouble (rccl.h:467)
constexpr int kNcclSum = 0;     // ncclSum (rccl.h:448)

}  // namespace

struct se2gpu_comm {
    void* comm = nullptr;
    int rank = 0, world = 1;
};

namespace se2gpu {

int comm_allreduce(se2gpu_comm* c, void* dev_ptr, size_t count, void* hip_stream) {
    Rccl* r = rccl();
    if (!c || !r) return -1;
    const int rc = r->all_reduce(dev_ptr, dev_ptr, count, kNcclDouble, kNcclSum, c->comm, (hipStream_t)hip_stream);
    if (rc != 0) set_error("ncclAllReduce failed: %s", r->error_string ? r->error_string(rc) : "?");
    return rc;
}
int comm_rank(const se2gpu_comm* c) { return c ? c->rank : 0; }
int comm_world(const se2gpu_comm* c) { return c ? c->world : 1; }

}  // namespace se2gpu

extern "C" {

int se2gpu_comm_unique_id(uint8_t id_out[128]) {
    SE2_REQUIRE(id_out, SE2GPU_ERR_INVALID, "comm_unique_id: NULL argument");
    Rccl* r = rccl();
    SE2_REQUIRE(r, SE2GPU_ERR_STATE, "librccl.so.1 could not be loaded: %s", dlerror() ? dlerror() : "symbol missing");
    nccl_unique_id id;
    const int rc = r->get_unique_id(&id);
    SE2_REQUIRE(rc == 0, SE2GPU_ERR_HIP, "ncclGetUniqueId failed: %s", r->error_string ? r->error_string(rc) : "?");
    std::memcpy(id_out, id.internal, 128);
    return SE2GPU_OK;
}

int se2gpu_comm_create(const uint8_t id[128], int rank, int world, se2gpu_comm** out) {
    SE2_REQUIRE(id && out && world >= 1 && rank >= 0 && rank < world, SE2GPU_ERR_INVALID, "comm_create: bad argument");
    SE2_REQUIRE(have_device(), SE2GPU_ERR_NO_DEVICE, "no HIP device visible");
    Rccl* r = rccl();
    SE2_REQUIRE(r, SE2GPU_ERR_STATE, "librccl.so.1 could not be loaded");
    nccl_unique_id uid;
    std::memcpy(uid.internal, id, 128);
    se2gpu_comm* c = new se2gpu_comm;
    c->rank = rank;
    c->world = world;
    const int rc = r->comm_init_rank(&c->comm, world, uid, rank);
    if (rc != 0) {
        set_error("ncclCommInitRank failed: %s", r->error_string ? r->error_string(rc) : "?");
        delete c;
        return SE2GPU_ERR_HIP;
    }
    *out = c;
    return SE2GPU_OK;
}

// ncclCommCount: the number of ranks RCCL itself says the communicator spans (bench.py checks it against WORLD_SIZE)
int se2gpu_comm_count(se2gpu_comm* c, int* nranks) {
    SE2_REQUIRE(c && nranks, SE2GPU_ERR_INVALID, "comm_count: NULL argument");
    Rccl* r = rccl();
    SE2_REQUIRE(r && r->comm_count, SE2GPU_ERR_STATE, "ncclCommCount is not available");
    const int rc = r->comm_count(c->comm, nranks);
    SE2_REQUIRE(rc == 0, SE2GPU_ERR_HIP, "ncclCommCount failed: %s", r->error_string ? r->error_string(rc) : "?");
    return SE2GPU_OK;
}

void se2gpu_comm_destroy(se2gpu_comm* c) {
    if (!c) return;
    Rccl* r = rccl();
    if (r && c->comm) (void)r->comm_destroy(c->comm);
    delete c;
}

int se2gpu_comm_allreduce_sum_f64(se2gpu_comm* c, void* dev_ptr, size_t count, void* hip_stream) {
    SE2_REQUIRE(c && dev_ptr, SE2GPU_ERR_INVALID, "comm_allreduce: NULL argument");
    return comm_allreduce(c, dev_ptr, count, hip_stream) == 0 ? SE2GPU_OK : SE2GPU_ERR_HIP;
}

}  // extern "C"
