// comm.hip -- the one real exchange of the multi-GPU path: every rank registers its own scan against its
// replica of the map; the points each rank accepted are all-gathered over RCCL (xGMI on an MI355X node) so that every
// replica appends the same set.  RCCL is loaded lazily (dlopen): a process that never creates a communicator does not
// need librccl.so at all, and the library binds to whichever copy the process already holds (PyTorch ships one).
#include "common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl& rccl()
{
    static Rccl r;
    static bool tried = false;
    if (tried) return r;
    tried = true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.lib) break;
    }
    if (!r.lib) { r.error = std::string("cannot load librccl.so: ") + dlerror(); return r; }
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GetErrorString) {
        r.error = "librccl.so lacks an expected symbol";
        dlclose(r.lib); r.lib = nullptr;
    }
    return r;
}

} // namespace

#define RCCL_TRY(ctx, expr)                                                                        \
    do {                                                                                           \
        ncclResult_t r_ = (expr);                                                                  \
        if (r_ != ncclSuccess) {                                                                   \
            (ctx)->last_error = std::string(#expr) + ": " + rccl().GetErrorString(r_);             \
            return ICPMI_ERR_HIP;                                                                  \
        }                                                                                          \
    } while (0)

icpmi_status comm_unique_id(icpmi_comm_id* id, std::string& err)
{
    static_assert(sizeof(icpmi_comm_id) == sizeof(ncclUniqueId), "icpmi_comm_id must hold an ncclUniqueId");
    Rccl& r = rccl();
    if (!r.lib) { err = r.error; return ICPMI_ERR_HIP; }
    ncclUniqueId u;
    const ncclResult_t rc = r.GetUniqueId(&u);
    if (rc != ncclSuccess) { err = std::string("ncclGetUniqueId: ") + r.GetErrorString(rc); return ICPMI_ERR_HIP; }
    memcpy(id, &u, sizeof u);
    return ICPMI_OK;
}

icpmi_status comm_init(icpmi_ctx* c, const icpmi_comm_id* id, int n_ranks, int rank)
{
    Rccl& r = rccl();
    if (!r.lib) { c->last_error = r.error; return ICPMI_ERR_HIP; }
    if (c->comm) { RCCL_TRY(c, r.CommDestroy((ncclComm_t)c->comm)); c->comm = nullptr; }
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclComm_t comm = nullptr;
    RCCL_TRY(c, r.CommInitRank(&comm, n_ranks, u, rank));
    c->comm = comm; c->comm_ranks = n_ranks; c->comm_rank = rank;
    return ICPMI_OK;
}

icpmi_status comm_destroy(icpmi_ctx* c)
{
    if (!c->comm) return ICPMI_OK;
    Rccl& r = rccl();
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (r.lib) (void)r.CommDestroy((ncclComm_t)c->comm);
    c->comm = nullptr; c->comm_ranks = 1; c->comm_rank = 0;
    return ICPMI_OK;
}

// all-gather of `count` elements per rank on the handle's stream (device buffers); no communicator = one rank = a copy
icpmi_status comm_allgather(icpmi_ctx* c, const void* d_send, void* d_recv, size_t count, bool is_float)
{
    if (!c->comm) {
        const size_t bytes = count * (is_float ? sizeof(float) : sizeof(long long));
        if (d_send != d_recv) HIP_TRY(c, hipMemcpyAsync(d_recv, d_send, bytes, hipMemcpyDeviceToDevice, c->stream));
        return ICPMI_OK;
    }
    RCCL_TRY(c, rccl().AllGather(d_send, d_recv, count, is_float ? ncclFloat : ncclInt64, (ncclComm_t)c->comm, c->stream));
    return ICPMI_OK;
}
