// rmcv_gather.hip -- the one exchange of the multi-GPU path, callable from a C++ host (BASELINE config 4: a batch of
// independent frames sharded over the GPUs of one node, RCCL over xGMI only for the final gather of the armour lists).
//
// One process per GPU.  Every rank runs the detection path on its own frames (no collective on the data path) and
// rmcv_batch_compact_armours leaves a fixed-size record [frame_offs | armours] in HBM; rmcv_gather brings the records of all
// ranks to the root as ONE group of point-to-point transfers (ncclGroupStart + ncclSend / ncclRecv + ncclGroupEnd): xGMI links
// are point to point, every peer has its own direct link to the root, and the payload is O(100 KB) -- latency-bound, nothing to
// push round a ring.  Asynchronous on the caller's stream, no host round trip.
//
// RCCL is bound at run time (dlopen "librccl.so.1"): the detection library has no link-time dependency on it, a single-GPU user
// never loads it, and a process that already carries an RCCL (PyTorch's) shares that copy.
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include <mutex>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include "../../include/rmcv_abi.h"

namespace {

struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    char err[256] = {0};
};
Rccl g_rccl;
std::once_flag g_once;

template <typename F> bool sym(F& f, const char* name)
{
    f = reinterpret_cast<F>(dlsym(g_rccl.h, name));
    if (!f) snprintf(g_rccl.err, sizeof(g_rccl.err), "librccl lacks %s", name);
    return f != nullptr;
}

const Rccl* rccl()
{
    std::call_once(g_once, [] {
        // first a copy the process already carries (PyTorch bundles one without a version in its name), then the system's
        for (const char* n : {"librccl.so.1", "librccl.so"}) {
            g_rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
            if (g_rccl.h) break;
        }
        for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (g_rccl.h) break;
            g_rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!g_rccl.h) {
            snprintf(g_rccl.err, sizeof(g_rccl.err), "cannot load librccl.so.1: %s", dlerror());
            return;
        }
        const bool ok = sym(g_rccl.GetUniqueId, "ncclGetUniqueId") && sym(g_rccl.CommInitRank, "ncclCommInitRank") &&
                        sym(g_rccl.CommDestroy, "ncclCommDestroy") && sym(g_rccl.CommCount, "ncclCommCount") &&
                        sym(g_rccl.GroupStart, "ncclGroupStart") && sym(g_rccl.GroupEnd, "ncclGroupEnd") && sym(g_rccl.Send, "ncclSend") &&
                        sym(g_rccl.Recv, "ncclRecv") && sym(g_rccl.GetErrorString, "ncclGetErrorString");
        if (!ok) g_rccl.h = nullptr;
    });
    return g_rccl.h ? &g_rccl : nullptr;
}

} // namespace

struct rmcv_comm {
    ncclComm_t comm = nullptr;
    int n_ranks = 0, rank = 0, device = 0;
    char err[256] = {0};
};

static int comm_fail(rmcv_comm* c, int code, const char* what, ncclResult_t r)
{
    if (c) snprintf(c->err, sizeof(c->err), "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "rccl error");
    return code;
}

extern "C" {

int rmcv_comm_unique_id(uint8_t id_out[RMCV_COMM_ID_BYTES])
{
    static_assert(RMCV_COMM_ID_BYTES == sizeof(ncclUniqueId), "the id blob is an ncclUniqueId");
    if (!id_out) return RMCV_ERR_BAD_ARG;
    const Rccl* R = rccl();
    if (!R) return RMCV_ERR_RCCL;
    ncclUniqueId id;
    if (R->GetUniqueId(&id) != ncclSuccess) return RMCV_ERR_RCCL;
    memcpy(id_out, &id, sizeof(id));
    return RMCV_OK;
}

int rmcv_comm_create(const uint8_t id[RMCV_COMM_ID_BYTES], int n_ranks, int rank, int device, rmcv_comm** out)
{
    if (!out) return RMCV_ERR_BAD_ARG;
    *out = nullptr;
    if (!id || n_ranks < 1 || rank < 0 || rank >= n_ranks || device < 0) return RMCV_ERR_BAD_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device >= ndev) return RMCV_ERR_NO_DEVICE;
    const Rccl* R = rccl();
    if (!R) return RMCV_ERR_RCCL;
    if (hipSetDevice(device) != hipSuccess) return RMCV_ERR_HIP;
    rmcv_comm* c = new (std::nothrow) rmcv_comm();
    if (!c) return RMCV_ERR_NOMEM;
    c->n_ranks = n_ranks;
    c->rank = rank;
    c->device = device;
    ncclUniqueId nid;
    memcpy(&nid, id, sizeof(nid));
    const ncclResult_t r = R->CommInitRank(&c->comm, n_ranks, nid, rank); // collective: every rank of the group calls it
    if (r != ncclSuccess) {
        fprintf(stderr, "rmcv_comm_create: ncclCommInitRank: %s\n", R->GetErrorString(r));
        delete c;
        return RMCV_ERR_RCCL;
    }
    int count = 0;
    if (R->CommCount(c->comm, &count) != ncclSuccess || count != n_ranks) {
        R->CommDestroy(c->comm);
        delete c;
        return RMCV_ERR_RCCL;
    }
    *out = c;
    return RMCV_OK;
}

void rmcv_comm_destroy(rmcv_comm* c)
{
    if (!c) return;
    if (c->comm && g_rccl.CommDestroy) {
        hipSetDevice(c->device);
        g_rccl.CommDestroy(c->comm);
    }
    delete c;
}

int rmcv_comm_info(const rmcv_comm* c, int32_t* n_ranks, int32_t* rank)
{
    if (!c) return RMCV_ERR_BAD_ARG;
    if (n_ranks) *n_ranks = c->n_ranks;
    if (rank) *rank = c->rank;
    return RMCV_OK;
}

const char* rmcv_comm_last_error(const rmcv_comm* c) { return c ? c->err : g_rccl.err; }

int rmcv_gather(rmcv_comm* c, const void* d_record, int64_t record_bytes, void* d_recv, int root, void* hip_stream)
{
    if (!c || !d_record || record_bytes <= 0 || root < 0 || root >= c->n_ranks) return RMCV_ERR_BAD_ARG;
    if (c->rank == root && !d_recv) return RMCV_ERR_BAD_ARG;
    const Rccl* R = rccl();
    if (!R) return RMCV_ERR_RCCL;
    if (hipSetDevice(c->device) != hipSuccess) return RMCV_ERR_HIP;
    hipStream_t s = (hipStream_t)hip_stream;
    ncclResult_t r = R->GroupStart();
    if (r != ncclSuccess) return comm_fail(c, RMCV_ERR_RCCL, "ncclGroupStart", r);
    if (c->rank == root) {
        for (int p = 0; p < c->n_ranks && r == ncclSuccess; p++) {
            uint8_t* dst = (uint8_t*)d_recv + (size_t)p * (size_t)record_bytes;
            if (p == root) continue; // the root's own record: a device copy, below
            r = R->Recv(dst, (size_t)record_bytes, ncclUint8, p, c->comm, s);
        }
    } else {
        r = R->Send(d_record, (size_t)record_bytes, ncclUint8, root, c->comm, s);
    }
    const ncclResult_t e = R->GroupEnd();
    if (r != ncclSuccess) return comm_fail(c, RMCV_ERR_RCCL, "ncclSend/ncclRecv", r);
    if (e != ncclSuccess) return comm_fail(c, RMCV_ERR_RCCL, "ncclGroupEnd", e);
    if (c->rank == root) {
        uint8_t* dst = (uint8_t*)d_recv + (size_t)root * (size_t)record_bytes;
        if (dst != d_record && hipMemcpyAsync(dst, d_record, (size_t)record_bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) {
            snprintf(c->err, sizeof(c->err), "hipMemcpyAsync of the root's own record failed");
            return RMCV_ERR_HIP;
        }
    }
    return RMCV_OK;
}

} // extern "C"
