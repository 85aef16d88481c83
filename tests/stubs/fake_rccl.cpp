// tests/stubs/fake_rccl.cpp -- a stand-in for librccl.so with the eight entry points csrc/comm.hip binds, for the tests of the
// library's error path (tests/test_gpu_comm_errors.py builds it with hipcc and hands its path to amm_comm_init).  One rank only:
// the collectives move nothing.  Behaviour from the environment, read at every call:
//   FAKE_RCCL_ASYNC_ERROR_AFTER=n   ncclCommGetAsyncError reports ncclRemoteError once n collectives were enqueued
//   FAKE_RCCL_STALL_MS=t            every collective parks the stream for t milliseconds (a peer that never arrives) -- until
//                                   ncclCommAbort releases it
//   FAKE_RCCL_LOG=path              one line per ncclCommAbort / ncclCommDestroy
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

extern "C" {
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
               ncclInvalidUsage = 5, ncclRemoteError = 6, ncclInProgress = 7 } ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef struct FakeComm *ncclComm_t;
struct FakeComm {
    std::atomic<int> collectives{0};
    std::atomic<int> aborted{0};
};

static int env_int(const char *name, int fallback) {
    const char *v = std::getenv(name);
    return v && *v ? std::atoi(v) : fallback;
}
static void log_line(const char *what) {
    const char *path = std::getenv("FAKE_RCCL_LOG");
    if (!path) return;
    if (FILE *f = std::fopen(path, "a")) {
        std::fprintf(f, "%s\n", what);
        std::fclose(f);
    }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    std::memset(id, 7, sizeof(*id));
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int world, ncclUniqueId, int rank) {
    if (world != 1 || rank != 0) return ncclInvalidArgument;
    *comm = new FakeComm();
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    log_line("destroy");
    delete comm;
    return ncclSuccess;
}
ncclResult_t ncclCommAbort(ncclComm_t comm) {
    log_line("abort");
    comm->aborted.store(1);                 // (the parked host functions see it and return; the object is leaked on purpose)
    return ncclSuccess;
}
static void park(void *p) {
    FakeComm *comm = static_cast<FakeComm *>(p);
    const int ms = env_int("FAKE_RCCL_STALL_MS", 0);
    const auto t0 = std::chrono::steady_clock::now();
    while (!comm->aborted.load() && std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() < ms)
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
}
static ncclResult_t collective(ncclComm_t comm, hipStream_t stream) {
    comm->collectives.fetch_add(1);
    if (env_int("FAKE_RCCL_STALL_MS", 0) > 0 && hipLaunchHostFunc(stream, park, comm) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}
ncclResult_t ncclAllReduce(const void *, void *, size_t, int, int, ncclComm_t comm, hipStream_t stream) { return collective(comm, stream); }
ncclResult_t ncclAllGather(const void *, void *, size_t, int, ncclComm_t comm, hipStream_t stream) { return collective(comm, stream); }
ncclResult_t ncclCommGetAsyncError(ncclComm_t comm, ncclResult_t *err) {
    const int after = env_int("FAKE_RCCL_ASYNC_ERROR_AFTER", -1);
    *err = (after >= 0 && comm->collectives.load() >= after) ? ncclRemoteError : ncclSuccess;
    return ncclSuccess;
}
const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclRemoteError: return "remote process exited or there was a network error";
    case ncclInvalidArgument: return "invalid argument";
    default: return "unhandled error";
    }
}
}
