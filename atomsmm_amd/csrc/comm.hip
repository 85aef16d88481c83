// atomsmm_amd/csrc/comm.hip -- RCCL collectives issued by the library itself (host code only).
//
// Atom decomposition: every rank evaluates the pair-force rows of its slice of the cell-sorted order; groups that
// hold one pair force exchange by ALL-GATHER of the owner-computed slices (the kernel leaves its rows in chunk `rank` of the
// caller's exchange buffer, amm_comm_allgather_impl brings the other chunks, k_unsort spreads them in atom order); groups that
// also hold sliced bond-list or reciprocal-space terms all-reduce(sum) their 3 N-double buffer.  When the host drives the
// exchange through torch.distributed, every outer step costs ~10 python -> C round trips (measured: 480 us of host time per step,
// more than the GPU time of a step on 8 ranks).  With a communicator of its own the library runs whole step programs
// (collectives inside amm_run_ops) without returning to the host: they are enqueued on the context's stream like any kernel.
//
// Errors (failure detection): an enqueue that fails returns at once; what fails LATER -- a peer that died, a
// link error -- is an asynchronous error of the communicator: polled (ncclCommGetAsyncError) after every enqueue and wherever the
// library waits for the stream (amm_check, amm_synchronize, amm_comm_destroy) -- an event-query loop, not hipStreamSynchronize, so
// that the error is seen while the stream is stuck -- and those waits can be bounded: with the option `comm_timeout` > 0 (seconds;
// default 0 = no deadline: a stream may legitimately hold minutes of queued steps) a stream that does not drain in time has the
// communicator aborted (ncclCommAbort) and the call returns non-zero with the reason in amm_last_error.
//
// RCCL is bound at run time (dlopen) rather than at link time: a torch process already carries one librccl, and two
// copies of the library in one process must not be mixed; the caller names the file (or NULL: the loader's default).
#include "amm_ctx.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstring>
#include <string>
#include <thread>

namespace {
struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr;      // (optional: older libraries lack them)
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
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
    api.CommGetAsyncError = (decltype(api.CommGetAsyncError))dlsym(h, "ncclCommGetAsyncError");
    api.CommAbort = (decltype(api.CommAbort))dlsym(h, "ncclCommAbort");
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

// asynchronous error of the communicator, if any: non-zero + message (the communicator is aborted: nothing more can run on it)
static void comm_abort(amm_ctx *ctx) {
    if (!ctx->comm) return;
    if (g_rccl.CommAbort) g_rccl.CommAbort((ncclComm_t)ctx->comm);
    else g_rccl.CommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_failed = true;
}

int amm_comm_poll_impl(amm_ctx *ctx) {
    if (!ctx->comm) {
        if (ctx->comm_failed) {
            amm_set_error("the context's RCCL communicator was aborted after an error; create a new context");
            return 1;
        }
        return 0;
    }
    if (!g_rccl.CommGetAsyncError) return 0;
    ncclResult_t err = ncclSuccess;
    const ncclResult_t r = g_rccl.CommGetAsyncError((ncclComm_t)ctx->comm, &err);
    if (r != ncclSuccess) err = r;
    if (err == ncclSuccess || err == ncclInProgress) return 0;
    const std::string why = g_rccl.GetErrorString(err);
    comm_abort(ctx);
    amm_set_error("RCCL asynchronous error (a peer rank died or a link failed): " + why + "; the communicator was aborted");
    return 1;
}

// wait for the context's stream, but not for ever when collectives may be stuck in it: event + query loop with the asynchronous
// error polled on the way; on a time-out the communicator is aborted (which releases its kernels) and the call fails
int amm_comm_wait_impl(amm_ctx *ctx, const char *who) {
    if (!ctx->comm) {
        AMM_HIP(hipStreamSynchronize(ctx->stream));
        return amm_comm_poll_impl(ctx);
    }
    hipEvent_t ev;
    AMM_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, ctx->stream);
    if (e != hipSuccess) {
        (void)hipEventDestroy(ev);
        amm_set_error(std::string("hipEventRecord: ") + hipGetErrorString(e));
        return 1;
    }
    const auto t0 = std::chrono::steady_clock::now();
    int rc = 0;
    for (long spins = 0;; ++spins) {
        e = hipEventQuery(ev);
        if (e == hipSuccess) break;
        if (e != hipErrorNotReady) {
            amm_set_error(std::string("hipEventQuery: ") + hipGetErrorString(e));
            rc = 1;
            break;
        }
        if ((spins & 63) == 63) {
            if (amm_comm_poll_impl(ctx)) {
                rc = 1;
                break;
            }
            const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (ctx->opt_comm_timeout > 0 && waited > ctx->opt_comm_timeout) {
                comm_abort(ctx);
                amm_set_error(std::string(who) + ": the stream did not drain within " + std::to_string((int)ctx->opt_comm_timeout) +
                              " s with collectives in flight (a peer rank that died or never arrived?); the RCCL communicator was aborted");
                rc = 1;
                break;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    }
    (void)hipEventDestroy(ev);
    if (rc) return rc;
    return amm_comm_poll_impl(ctx);
}

int amm_comm_destroy_impl(amm_ctx *ctx) {
    int rc = 0;
    if (ctx->comm) {
        rc = amm_comm_wait_impl(ctx, "amm_comm_destroy");      // (aborts the communicator itself when the wait fails)
        if (ctx->comm) {
            g_rccl.CommDestroy((ncclComm_t)ctx->comm);
            ctx->comm = nullptr;
        }
    }
    return rc;
}

// in-place sum over ranks of `count` doubles, enqueued on the context's stream
int amm_comm_allreduce_impl(amm_ctx *ctx, double *d_buf, size_t count) {
    if (!ctx->comm) {
        if (ctx->comm_failed) return amm_comm_poll_impl(ctx);
        amm_set_error("all-reduce without a communicator (amm_comm_init)");
        return 1;
    }
    ncclResult_t r = g_rccl.AllReduce(d_buf, d_buf, count, ncclDouble, ncclSum, (ncclComm_t)ctx->comm, ctx->stream);
    if (r != ncclSuccess) return rccl_fail("ncclAllReduce", r);
    ctx->comm_calls++;
    ctx->comm_doubles += (long long)count;
    return amm_comm_poll_impl(ctx);
}

// in-place all-gather: chunk `rank` of d_buf (count_per_rank doubles) goes to every rank's d_buf, on the context's stream
int amm_comm_allgather_impl(amm_ctx *ctx, double *d_buf, size_t count_per_rank) {
    if (!ctx->comm) {
        if (ctx->comm_failed) return amm_comm_poll_impl(ctx);
        amm_set_error("all-gather without a communicator (amm_comm_init)");
        return 1;
    }
    ncclResult_t r = g_rccl.AllGather(d_buf + (size_t)ctx->rank * count_per_rank, d_buf, count_per_rank, ncclDouble,
                                      (ncclComm_t)ctx->comm, ctx->stream);
    if (r != ncclSuccess) return rccl_fail("ncclAllGather", r);
    ctx->comm_calls++;
    ctx->comm_doubles += (long long)count_per_rank;
    return amm_comm_poll_impl(ctx);
}
