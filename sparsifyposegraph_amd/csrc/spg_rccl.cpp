// csrc/spg_rccl.cpp — the one collective of the path: an in-place all-gather of a sharded round's output
// region (recovered edge records + per-blanket out records) over RCCL / xGMI.
//
// The reference has no counterpart (single process, no collective). librccl.so.1 is bound with dlopen when the
// first multi-rank context is created: single-GPU users of libspg_hip.so never load it, and in a process that
// already holds a copy (torch ships one under the same SONAME) the loader hands back that copy, so there is one
// RCCL per process. Prototypes come from <rccl/rccl.h>.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>

#include "spg_internal.h"

namespace {

struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

RcclApi g_api;
std::mutex g_mu;

int bind_rccl(char *err, size_t errlen) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_api.lib) return 0;
    void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) { snprintf(err, errlen, "cannot load librccl.so.1: %s", dlerror()); return SPG_ENODEV; }
    RcclApi a;
    a.lib = lib;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(lib, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(lib, "ncclCommDestroy");
    a.AllGather = (decltype(a.AllGather))dlsym(lib, "ncclAllGather");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather || !a.GetErrorString) {
        snprintf(err, errlen, "librccl.so.1 lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather");
        return SPG_ENODEV;
    }
    g_api = a;
    return 0;
}

struct Comm {
    ncclComm_t comm = nullptr;
    int device = 0, rank = 0, nranks = 1;
};

}  // namespace

namespace spg {

int rccl_get_unique_id(void *id_out, char *err, size_t errlen) {
    static_assert(sizeof(ncclUniqueId) == SPG_UNIQUE_ID_BYTES, "SPG_UNIQUE_ID_BYTES must match ncclUniqueId");
    if (int rc = bind_rccl(err, errlen)) return rc;
    ncclUniqueId id;
    ncclResult_t r = g_api.GetUniqueId(&id);
    if (r != ncclSuccess) { snprintf(err, errlen, "ncclGetUniqueId: %s", g_api.GetErrorString(r)); return SPG_EHIP; }
    memcpy(id_out, &id, sizeof id);
    return 0;
}

int rccl_comm_create(int device, int rank, int nranks, const void *unique_id, void **handle, char *err, size_t errlen) {
    if (int rc = bind_rccl(err, errlen)) return rc;
    if (hipSetDevice(device) != hipSuccess) { snprintf(err, errlen, "hipSetDevice(%d) failed", device); return SPG_EHIP; }
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    Comm *c = new Comm;
    c->device = device; c->rank = rank; c->nranks = nranks;
    ncclResult_t r = g_api.CommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        snprintf(err, errlen, "ncclCommInitRank(rank %d of %d, device %d): %s", rank, nranks, device, g_api.GetErrorString(r));
        delete c;
        return SPG_EHIP;
    }
    *handle = c;
    return 0;
}

// In-place all-gather: rank r's chunk already sits at arena[region_off + r * chunk_len); ncclAllGather's in-place
// form wants sendbuff == recvbuff + rank * sendcount, which is exactly that layout.
int rccl_allgather_f64(void *handle, void *arena, int64_t region_off, int64_t chunk_len, void *stream, char *err, size_t errlen) {
    Comm *c = (Comm *)handle;
    if (!c || !c->comm) return SPG_EINVAL;
    if (chunk_len == 0) return 0;
    double *recv = (double *)arena + region_off;
    const double *send = recv + (int64_t)c->rank * chunk_len;
    ncclResult_t r = g_api.AllGather(send, recv, (size_t)chunk_len, ncclDouble, c->comm, (hipStream_t)stream);
    if (r != ncclSuccess) { snprintf(err, errlen, "ncclAllGather(%lld doubles per rank): %s", (long long)chunk_len, g_api.GetErrorString(r)); return SPG_EHIP; }
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) { snprintf(err, errlen, "hipStreamSynchronize after ncclAllGather: %s", hipGetErrorString(e)); return SPG_EHIP; }
    return 0;
}

void rccl_comm_destroy(void *handle) {
    Comm *c = (Comm *)handle;
    if (!c) return;
    if (c->comm && g_api.CommDestroy) (void)g_api.CommDestroy(c->comm);
    delete c;
}

}  // namespace spg
