// RCCL side of the shared-parameter run (the reference's DDP + SyncBatchNorm on NCCL: src/tta_main.py:101-111, 326-354;
// DDP gradient all-reduce src/msg_chn_model_adapt.py:476-480; SyncBatchNorm src/msg_chn_model_adapt.py:547-556).
//
// The library owns an RCCL communicator (one per process = one per GPU) and enqueues its collectives itself, on the stream the
// step is being enqueued on: no Python frame per BatchNorm, and the collective is an ordinary stream operation, so the MSG_CHN
// step keeps its hipGraph while the exchange is on.  RCCL's C API is resolved at run time from the librccl the process already
// holds (PyTorch-ROCm ships one) or from /opt/rocm: libptta_hip.so has no link-time dependency on it, and two RCCL instances in
// one process are avoided.
//   rank 0: ptta_rccl_unique_id(id)  ->  the caller broadcasts the 128 bytes (any transport: torch.distributed, a file, MPI)
//   every rank: ptta_rccl_comm_create(id, rank, world, &comm)  ->  ptta_set_stat_sync_rccl(handle, comm, buf, capacity, world)
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <rccl/rccl.h>

#include <mutex>

#include "../../include/ptta.h"
#include "ptta_kernels.h"

namespace {
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
RcclApi g_api;
std::once_flag g_once;
thread_local char g_err[256] = "";
char g_load_err[192] = "";              // dlerror() of the last failed dlopen, captured right behind it (a later dl* call resets it)

void load_api() {
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names) if (!g_api.lib) g_api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);      // the process's own copy first
    const char* paths[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : paths)
        if (!g_api.lib) {
            g_api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (!g_api.lib) { const char* e = dlerror(); snprintf(g_load_err, sizeof(g_load_err), "%s", e ? e : "dlopen failed"); }
        }
    if (!g_api.lib) return;
#define SYM(field, name) g_api.field = (decltype(g_api.field))dlsym(g_api.lib, name)
    SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllReduce, "ncclAllReduce"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_api.ok = g_api.GetUniqueId && g_api.CommInitRank && g_api.CommDestroy && g_api.AllReduce;
}
int api() { std::call_once(g_once, load_api); if (!g_api.ok) { snprintf(g_err, sizeof(g_err), "librccl could not be loaded: %s", g_api.lib ? "symbols missing" : g_load_err); return -38; } return 0; }
int chk(ncclResult_t r, const char* what) {
    if (r == ncclSuccess) return 0;
    snprintf(g_err, sizeof(g_err), "%s: %s", what, g_api.GetErrorString ? g_api.GetErrorString(r) : "RCCL error");
    return -5;
}
}  // namespace

// SUM of `count` doubles over the communicator, in place, ordered on `s` (gbn.hip ptta_stat_sync calls this when a communicator is set)
int ptta_rccl_sum_f64(void* comm, double* buf, long long count, hipStream_t s) {
    if (!comm || !buf || count < 0) return -22;
    if (api()) return -38;
    return chk(g_api.AllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, (ncclComm_t)comm, s), "ncclAllReduce(double, sum)");
}

extern "C" {

const char* ptta_rccl_last_error(void) { return g_err; }

int ptta_rccl_unique_id(void* id128) {
    if (!id128) return -22;
    if (api()) return -38;
    ncclUniqueId id;
    const int rc = chk(g_api.GetUniqueId(&id), "ncclGetUniqueId");
    if (rc) return rc;
    memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

int ptta_rccl_comm_create(const void* id128, int rank, int world_size, void** comm_out) {
    if (!id128 || !comm_out || world_size < 1 || rank < 0 || rank >= world_size) return -22;
    if (api()) return -38;
    ncclUniqueId id; memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t comm = nullptr;
    const int rc = chk(g_api.CommInitRank(&comm, world_size, id, rank), "ncclCommInitRank");
    if (rc) return rc;
    *comm_out = (void*)comm;
    return 0;
}

int ptta_rccl_comm_destroy(void* comm) {
    if (!comm) return 0;
    if (api()) return -38;
    return chk(g_api.CommDestroy((ncclComm_t)comm), "ncclCommDestroy");
}

// The ONE gradient collective of a shared-parameter step: mean over the ranks of a flat fp32 vector, in place, ordered on `s`
// (replaces DDP's bucketed all-reduce of all 1.46 M gradients by the adapted ones only: 37 KB for MSG_CHN 1layer)
int ptta_rccl_allreduce_mean_f32(void* comm, float* buf, int64_t count, ptta_stream s) {
    if (!comm || !buf || count < 0) return -22;
    if (api()) return -38;
    return chk(g_api.AllReduce(buf, buf, (size_t)count, ncclFloat32, ncclAvg, (ncclComm_t)comm, (hipStream_t)s), "ncclAllReduce(float, avg)");
}

}  // extern "C"
