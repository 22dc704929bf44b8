// RCCL directly behind the C ABI (sehip_comm_init / sehip_allreduce_f32): the data-parallel gradient exchange of
// the train step without torch.distributed in the data path.  The reference's only multi-GPU mechanism is single-process
// nn.DataParallel (src/solver.py:144-145); here it is one process per GPU and ONE in-place SUM all-reduce of the flat fp32
// gradient buffer (ranges of it, see sehip/solver.py) over xGMI.  librccl is loaded on first use (dlopen), so that libsehip itself
// loads on a box without it; a missing library or symbol is an error of the call, never a silent fallback.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "common.h"

namespace {
typedef struct { char internal[128]; } rccl_unique_id;            // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* rccl_comm;
typedef int (*get_unique_id_fn)(rccl_unique_id*);
typedef int (*comm_init_rank_fn)(rccl_comm*, int, rccl_unique_id, int);
typedef int (*all_reduce_fn)(const void*, void*, size_t, int /*dtype*/, int /*op*/, rccl_comm, hipStream_t);
typedef int (*comm_destroy_fn)(rccl_comm);
typedef const char* (*get_error_fn)(int);
typedef int (*comm_query_fn)(const rccl_comm, int*);
struct Rccl {
    void* h = nullptr;
    get_unique_id_fn get_unique_id = nullptr;
    comm_init_rank_fn comm_init_rank = nullptr;
    all_reduce_fn all_reduce = nullptr;
    comm_destroy_fn comm_destroy = nullptr;
    get_error_fn get_error = nullptr;
    comm_query_fn comm_count = nullptr, comm_user_rank = nullptr, comm_device = nullptr;
};
Rccl g_rccl;

int load_rccl(const char* who) {
    if (g_rccl.h) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return sehip_set_error(-3, "%s: librccl.so.1 cannot be loaded: %s", who, dlerror());
    Rccl r;
    r.h = h;
    r.get_unique_id = (get_unique_id_fn)dlsym(h, "ncclGetUniqueId");
    r.comm_init_rank = (comm_init_rank_fn)dlsym(h, "ncclCommInitRank");
    r.all_reduce = (all_reduce_fn)dlsym(h, "ncclAllReduce");
    r.comm_destroy = (comm_destroy_fn)dlsym(h, "ncclCommDestroy");
    r.get_error = (get_error_fn)dlsym(h, "ncclGetErrorString");
    r.comm_count = (comm_query_fn)dlsym(h, "ncclCommCount");
    r.comm_user_rank = (comm_query_fn)dlsym(h, "ncclCommUserRank");
    r.comm_device = (comm_query_fn)dlsym(h, "ncclCommCuDevice");
    if (!r.get_unique_id || !r.comm_init_rank || !r.all_reduce || !r.comm_destroy)
        return sehip_set_error(-3, "%s: librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy", who);
    g_rccl = r;
    return 0;
}
const char* err_text(int e) { return g_rccl.get_error ? g_rccl.get_error(e) : "rccl error"; }
}  // namespace

extern "C" int sehip_comm_unique_id(void* id128) {
    SEHIP_REQUIRE(id128 != nullptr, "comm_unique_id: null buffer");
    if (int e = load_rccl("comm_unique_id")) return e;
    rccl_unique_id id;
    const int r = g_rccl.get_unique_id(&id);
    SEHIP_REQUIRE(r == 0, "comm_unique_id: ncclGetUniqueId: %s", err_text(r));
    memcpy(id128, &id, sizeof(id));
    return 0;
}

extern "C" int sehip_comm_init(const void* id128, int world, int rank, void** comm_out) {
    SEHIP_REQUIRE(id128 && comm_out && world >= 1 && rank >= 0 && rank < world, "comm_init: bad arguments (world=%d rank=%d)", world, rank);
    if (int e = load_rccl("comm_init")) return e;
    rccl_unique_id id;
    memcpy(&id, id128, sizeof(id));
    rccl_comm c = nullptr;
    const int r = g_rccl.comm_init_rank(&c, world, id, rank);     // on the CURRENT HIP device (one process per GPU)
    SEHIP_REQUIRE(r == 0 && c, "comm_init: ncclCommInitRank(world=%d, rank=%d): %s", world, rank, err_text(r));
    *comm_out = c;
    return 0;
}

// in-place SUM over the ranks of buf[0 .. n) (fp32), enqueued on `stream`
extern "C" int sehip_allreduce_f32(void* comm, float* buf, long n, void* stream) {
    SEHIP_REQUIRE(comm && buf && n >= 0, "allreduce_f32: bad arguments (n=%ld)", n);
    if (int e = load_rccl("allreduce_f32")) return e;
    if (n == 0) return 0;
    const int r = g_rccl.all_reduce(buf, buf, (size_t)n, 7 /*ncclFloat32*/, 0 /*ncclSum*/, (rccl_comm)comm, (hipStream_t)stream);
    SEHIP_REQUIRE(r == 0, "allreduce_f32: ncclAllReduce(n=%ld): %s", n, err_text(r));
    return 0;
}

// in-place MAX over the ranks of buf[0 .. n) (int32): the data-parallel step guard (a device-side failure word such as Demucs'
// hand-off time-out must stop the optimizer step of EVERY replica, or the replicas diverge)
extern "C" int sehip_allreduce_i32_max(void* comm, int* buf, long n, void* stream) {
    SEHIP_REQUIRE(comm && buf && n >= 0, "allreduce_i32_max: bad arguments (n=%ld)", n);
    if (int e = load_rccl("allreduce_i32_max")) return e;
    if (n == 0) return 0;
    const int r = g_rccl.all_reduce(buf, buf, (size_t)n, 2 /*ncclInt32*/, 2 /*ncclMax*/, (rccl_comm)comm, (hipStream_t)stream);
    SEHIP_REQUIRE(r == 0, "allreduce_i32_max: ncclAllReduce(n=%ld): %s", n, err_text(r));
    return 0;
}

// what the LIVE communicator says about itself (ncclCommCount / ncclCommUserRank / ncclCommCuDevice): bench.py's N > 1 line carries it,
// so that the line proves N ranks on N devices instead of repeating what the launcher was asked for
extern "C" int sehip_comm_info(void* comm, int* nranks, int* rank, int* device) {
    SEHIP_REQUIRE(comm && nranks && rank && device, "comm_info: null argument");
    if (int e = load_rccl("comm_info")) return e;
    SEHIP_REQUIRE(g_rccl.comm_count && g_rccl.comm_user_rank && g_rccl.comm_device, "comm_info: librccl lacks ncclCommCount / ncclCommUserRank / ncclCommCuDevice");
    int r = g_rccl.comm_count((rccl_comm)comm, nranks);
    SEHIP_REQUIRE(r == 0, "comm_info: ncclCommCount: %s", err_text(r));
    r = g_rccl.comm_user_rank((rccl_comm)comm, rank);
    SEHIP_REQUIRE(r == 0, "comm_info: ncclCommUserRank: %s", err_text(r));
    r = g_rccl.comm_device((rccl_comm)comm, device);
    SEHIP_REQUIRE(r == 0, "comm_info: ncclCommCuDevice: %s", err_text(r));
    return 0;
}

extern "C" int sehip_comm_destroy(void* comm) {
    if (!comm) return 0;
    if (int e = load_rccl("comm_destroy")) return e;
    const int r = g_rccl.comm_destroy((rccl_comm)comm);
    SEHIP_REQUIRE(r == 0, "comm_destroy: ncclCommDestroy: %s", err_text(r));
    return 0;
}
