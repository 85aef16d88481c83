// atomsmm_amd/csrc/comm.hip -- RCCL collectives issued by the library itself (host code only).
//
// Atom decomposition: every rank evaluates the pair forces of its slice of the cell-sorted atoms into a
// full-size buffer (zeros elsewhere) and the ranks all-reduce that buffer after each evaluation of a sliced group.
// When the host drives this through torch.distributed, every outer step costs ~10 python -> C round trips
// (measured: 480 us of host time per step, more than the GPU time of a step on 8 ranks).  With a communicator of its
// own the library runs whole step programs (AMM_OP_ALLREDUCE inside amm_run_ops) without returning to the host:
// ncclAllReduce is enqueued on the context's stream like any kernel.
//
// RCCL is bound at run time (dlopen) rather than at link time: a torch process already carries one librccl, and two
// copies of the library in one process must not be mixed; the caller names the file (or NULL: the loader's default).
#include "amm_ctx.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>

namespace {
struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
} g_rccl;

int rccl_load(const char *path) {
    if (g_rccl.handle) return 0;
    const char *candidates[] = {path, "librccl.so.1", "librccl.so"};
    void *h = nullptr;
    std::string tried;
    for (const char *name : candidates) {
        if (!name || !*name) continue;
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
        tried += std::string(tried.empty() ? "" : "; ") + dlerror();
    }
    if (!h) {
        amm_set_error(("amm_comm: cannot load RCCL: " + tried).c_str());
        return 1;
    }
    RcclApi api;
    api.handle = h;
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(h, "ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(h, "ncclCommDestroy");
    api.AllReduce = (decltype(api.AllReduce))dlsym(h, "ncclAllReduce");
    api.AllGather = (decltype(api.AllGather))dlsym(h, "ncclAllGather");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.AllGather || !api.GetErrorString) {
        amm_set_error("amm_comm: the RCCL library lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce");
        dlclose(h);
        return 1;
    }
    g_rccl = api;
    return 0;
}

int rccl_fail(const char *what, ncclResult_t r) {
    amm_set_error((std::string(what) + ": " + g_rccl.GetErrorString(r)).c_str());
    return 1;
}
}  // namespace

int amm_comm_unique_id_impl(const char *rccl_path, unsigned char *out) {
    if (rccl_load(rccl_path)) return 1;
    ncclUniqueId id;
    ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return rccl_fail("ncclGetUniqueId", r);
    static_assert(sizeof(id) == AMM_COMM_ID_BYTES, "ncclUniqueId size");
    std::memcpy(out, &id, sizeof(id));
    return 0;
}

int amm_comm_init_impl(amm_ctx *ctx, const char *rccl_path, const unsigned char *id_bytes, int rank, int world) {
    if (ctx->comm) {
        amm_set_error("amm_comm_init: the context already has a communicator");
        return 1;
    }
    if (rank != ctx->rank || world != ctx->world) {
        amm_set_error("amm_comm_init: rank / world differ from amm_set_slice");
        return 1;
    }
    if (rccl_load(rccl_path)) return 1;
    AMM_HIP(hipSetDevice(ctx->device));
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof(id));
    ncclComm_t comm = nullptr;
    ncclResult_t r = g_rccl.CommInitRank(&comm, world, id, rank);
    if (r != ncclSuccess) return rccl_fail("ncclCommInitRank", r);
    ctx->comm = comm;
    return 0;
}

int amm_comm_destroy_impl(amm_ctx *ctx) {
    if (ctx->comm) {
        (void)hipStreamSynchronize(ctx->stream);
        g_rccl.CommDestroy((ncclComm_t)ctx->comm);
        ctx->comm = nullptr;
    }
    return 0;
}

// in-place sum over ranks of `count` doubles, enqueued on the context's stream
int amm_comm_allreduce_impl(amm_ctx *ctx, double *d_buf, size_t count) {
    if (!ctx->comm) {
        amm_set_error("all-reduce without a communicator (amm_comm_init)");
        return 1;
    }
    ncclResult_t r = g_rccl.AllReduce(d_buf, d_buf, count, ncclDouble, ncclSum, (ncclComm_t)ctx->comm, ctx->stream);
    if (r != ncclSuccess) return rccl_fail("ncclAllReduce", r);
    ctx->comm_calls++;
    ctx->comm_doubles += (long long)count;
    return 0;
}

// in-place all-gather: chunk `rank` of d_buf (count_per_rank doubles) goes to every rank's d_buf, on the context's stream
int amm_comm_allgather_impl(amm_ctx *ctx, double *d_buf, size_t count_per_rank) {
    if (!ctx->comm) {
        amm_set_error("all-gather without a communicator (amm_comm_init)");
        return 1;
    }
    ncclResult_t r = g_rccl.AllGather(d_buf + (size_t)ctx->rank * count_per_rank, d_buf, count_per_rank, ncclDouble,
                                      (ncclComm_t)ctx->comm, ctx->stream);
    if (r != ncclSuccess) return rccl_fail("ncclAllGather", r);
    ctx->comm_calls++;
    ctx->comm_doubles += (long long)count_per_rank;
    return 0;
}
