// Data-parallel collective of the C ABI (allreduce_bucket(ptr, n, dtype, comm, stream)): a SUM all-reduce of a
// device buffer over RCCL, for binders of include/dm_amd.h that do not bring torch.distributed.  The reference has no distributed code;
// the semantics are those of its gradient accumulation (new_scripy.py:786, 795-803): every rank = one micro-batch, gradients summed,
// 1/world applied by the optimiser (dm_adamw's gradient scale).
//
// librccl.so is bound at run time (dlopen / dlsym), not at link time: libdm_amd.so must load on a box without RCCL, and inside a
// torch process it must use the librccl instance torch has already mapped (two instances would each keep their own communicator
// state).  Search order: the path given to dm_comm_load; the environment variable DM_RCCL_LIB; a librccl already mapped into the
// process (RTLD_NOLOAD); the loader's default search for "librccl.so.1" / "librccl.so".
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include "common.h"

namespace {
struct UniqueId { char internal[128]; };                  // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128), passed BY VALUE to ncclCommInitRank
typedef void* Comm;
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef int (*CommDestroyFn)(Comm);
typedef const char* (*GetErrorStringFn)(int);
typedef int (*GroupFn)(void);

struct Rccl {
    void* h = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    AllReduceFn all_reduce = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    GetErrorStringFn error_string = nullptr;
    GroupFn group_start = nullptr, group_end = nullptr;
} g_rccl;

constexpr int NCCL_SUM = 0, NCCL_F16 = 6, NCCL_F32 = 7, NCCL_BF16 = 9;      // ncclRedOp_t / ncclDataType_t values of nccl.h (RCCL keeps them)

int bind(void* h, const char* from) {
    Rccl r;
    r.h = h;
    r.get_unique_id = (GetUniqueIdFn)dlsym(h, "ncclGetUniqueId");
    r.comm_init_rank = (CommInitRankFn)dlsym(h, "ncclCommInitRank");
    r.all_reduce = (AllReduceFn)dlsym(h, "ncclAllReduce");
    r.comm_destroy = (CommDestroyFn)dlsym(h, "ncclCommDestroy");
    r.error_string = (GetErrorStringFn)dlsym(h, "ncclGetErrorString");
    r.group_start = (GroupFn)dlsym(h, "ncclGroupStart");
    r.group_end = (GroupFn)dlsym(h, "ncclGroupEnd");
    if (!r.get_unique_id || !r.comm_init_rank || !r.all_reduce || !r.comm_destroy || !r.error_string) {
        dm_set_error("dm_comm: %s does not export the nccl* entry points", from);
        return DM_EUNSUPPORTED;
    }
    g_rccl = r;
    return DM_OK;
}

int ensure_loaded() {
    if (g_rccl.h) return DM_OK;
    const char* env = getenv("DM_RCCL_LIB");
    if (env && *env) {
        void* h = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
        if (!h) { dm_set_error("dm_comm: DM_RCCL_LIB=%s: %s", env, dlerror()); return DM_EUNSUPPORTED; }
        return bind(h, env);
    }
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* nm : names) {                           // the instance the process already has (torch's), if any
        void* h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
        if (h) return bind(h, nm);
    }
    for (const char* nm : {"librccl.so.1", "librccl.so"}) {
        void* h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (h) return bind(h, nm);
    }
    dm_set_error("dm_comm: librccl.so not found (set DM_RCCL_LIB or call dm_comm_load with its path): %s", dlerror());
    return DM_EUNSUPPORTED;
}

int check(int rc, const char* what) {
    if (rc == 0) return DM_OK;
    dm_set_error("%s: %s", what, g_rccl.error_string ? g_rccl.error_string(rc) : "RCCL error");
    return rc > 0 ? -1000 - rc : rc;
}
}  // namespace

extern "C" int dm_comm_load(const char* path) {
    DM_CHECK_ARG(path && *path, "dm_comm_load: need the path of librccl.so");
    void* h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    if (!h) { dm_set_error("dm_comm_load: %s", dlerror()); return DM_EUNSUPPORTED; }
    return bind(h, path);
}

extern "C" int dm_comm_unique_id(void* id128) {
    DM_CHECK_ARG(id128, "dm_comm_unique_id: need 128 bytes");
    int rc = ensure_loaded();
    if (rc) return rc;
    return check(g_rccl.get_unique_id((UniqueId*)id128), "ncclGetUniqueId");
}

extern "C" int dm_comm_init(void** comm_out, int world, int rank, const void* id128) {
    DM_CHECK_ARG(comm_out && id128 && world >= 1 && rank >= 0 && rank < world, "dm_comm_init: bad arguments");
    int rc = ensure_loaded();
    if (rc) return rc;
    UniqueId id;
    memcpy(&id, id128, sizeof(id));
    Comm c = nullptr;
    rc = check(g_rccl.comm_init_rank(&c, world, id, rank), "ncclCommInitRank");
    if (rc) return rc;
    *comm_out = c;
    return DM_OK;
}

extern "C" int dm_allreduce_bucket(void* ptr, int64_t n, int dtype, void* comm, dm_stream_t s) {
    DM_CHECK_ARG(ptr && n > 0 && comm, "dm_allreduce_bucket: bad arguments");
    DM_CHECK_ARG(dtype == DM_F32 || dtype == DM_BF16 || dtype == DM_F16, "dm_allreduce_bucket: dtype is DM_F32, DM_BF16 or DM_F16");
    DM_CHECK_ARG(g_rccl.h, "dm_allreduce_bucket: no communicator was made through dm_comm_init");
    const int dt = dtype == DM_F32 ? NCCL_F32 : (dtype == DM_BF16 ? NCCL_BF16 : NCCL_F16);
    return check(g_rccl.all_reduce(ptr, ptr, (size_t)n, dt, NCCL_SUM, (Comm)comm, (hipStream_t)s), "ncclAllReduce");
}

extern "C" int dm_comm_destroy(void* comm) {
    if (!comm) return DM_OK;
    DM_CHECK_ARG(g_rccl.h, "dm_comm_destroy: RCCL is not loaded");
    return check(g_rccl.comm_destroy((Comm)comm), "ncclCommDestroy");
}
