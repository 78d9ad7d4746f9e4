// fake_rccl.cpp -- TEST INFRASTRUCTURE (tests/test_gpu_gather_world2.py): a stand-in for librccl.so.1 that moves ncclSend / ncclRecv
// payloads between PROCESSES ON ONE GPU through files, so that rmcv_gather and the pipeline's ordering of its gathers
// (rmcv_pipeline_set_gather: ev_gather, gather_pending) run at world size 2 on the one-GPU box this build has -- no multi-GPU node
// was available to any round.  Not RCCL, not part of the product; it implements exactly the nine entry points rmcv_gather.hip binds.
//
// Semantics kept: an operation takes effect when its STREAM reaches it (a host function on the stream); one communicator's operations
// pair up IN EXECUTION ORDER -- the k-th send a rank executes towards a peer meets the k-th receive the peer executes from it.  That
// is what makes RCCL require one order of a communicator's operations on every rank, and what the pipeline's event chain provides:
// were two gathers of different tickets (on different streams) to execute out of ticket order on one rank, the payloads here would
// land in the wrong tickets' buffers and the test would see another batch's list.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <string>
#include <vector>

extern "C" {
typedef enum { ncclSuccess = 0, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1 } ncclDataType_t;
typedef struct { char internal[128]; } ncclUniqueId;
struct ncclComm {
    int n = 0, rank = 0;
    std::string dir;
    std::atomic<uint64_t> send_exec[16], recv_exec[16]; // per peer: operations EXECUTED so far (not enqueued)
    std::vector<void*> staging;
};
typedef struct ncclComm* ncclComm_t;
}

namespace {
struct Op {
    ncclComm* c;
    int peer;
    bool send;
    void* staging;
    size_t bytes;
};
std::string path_of(const ncclComm* c, int from, int to, uint64_t seq)
{
    char b[512];
    snprintf(b, sizeof(b), "%s/msg_%d_to_%d_%llu", c->dir.c_str(), from, to, (unsigned long long)seq);
    return b;
}
void on_stream(void* user)
{
    Op* op = static_cast<Op*>(user);
    ncclComm* c = op->c;
    if (op->send) {
        const uint64_t seq = c->send_exec[op->peer].fetch_add(1);
        const std::string p = path_of(c, c->rank, op->peer, seq), tmp = p + ".tmp";
        FILE* f = fopen(tmp.c_str(), "wb");
        if (f) {
            fwrite(op->staging, 1, op->bytes, f);
            fclose(f);
            rename(tmp.c_str(), p.c_str());
        }
    } else {
        const uint64_t seq = c->recv_exec[op->peer].fetch_add(1);
        const std::string p = path_of(c, op->peer, c->rank, seq);
        for (int tries = 0; tries < 60000; tries++) { // up to 60 s
            FILE* f = fopen(p.c_str(), "rb");
            if (f) {
                const size_t got = fread(op->staging, 1, op->bytes, f);
                fclose(f);
                if (got == op->bytes) { unlink(p.c_str()); break; }
            }
            timespec nap = {0, 1000000};
            nanosleep(&nap, nullptr);
        }
    }
    delete op;
}
} // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    memset(id, 0, sizeof(*id));
    const char* d = getenv("FAKE_RCCL_DIR");
    if (!d) return ncclSystemError;
    snprintf(id->internal, sizeof(id->internal), "%s", d); // the "id" is the directory the ranks meet in
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* out, int n, ncclUniqueId id, int rank)
{
    if (n < 1 || n > 16 || rank < 0 || rank >= n) return ncclInvalidArgument;
    ncclComm* c = new ncclComm();
    c->n = n;
    c->rank = rank;
    c->dir = id.internal;
    for (int i = 0; i < 16; i++) { c->send_exec[i] = 0; c->recv_exec[i] = 0; }
    *out = c;
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (!c) return ncclSuccess;
    (void)hipDeviceSynchronize();
    for (void* p : c->staging) (void)hipHostFree(p);
    delete c;
    return ncclSuccess;
}
ncclResult_t ncclCommCount(const ncclComm_t c, int* n) { *n = c->n; return ncclSuccess; }
ncclResult_t ncclGroupStart() { return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return ncclSuccess; }
static ncclResult_t enqueue(bool send, void* buf, size_t count, int peer, ncclComm_t c, hipStream_t s)
{
    if (!c || peer < 0 || peer >= c->n || peer == c->rank) return ncclInvalidArgument;
    void* st = nullptr;
    if (hipHostMalloc(&st, count, hipHostMallocDefault) != hipSuccess) return ncclSystemError;
    c->staging.push_back(st);
    Op* op = new Op{c, peer, send, st, count};
    if (send) { // device -> staging on the stream, then the host function hands the bytes over
        if (hipMemcpyAsync(st, buf, count, hipMemcpyDeviceToHost, s) != hipSuccess) return ncclSystemError;
        if (hipLaunchHostFunc(s, on_stream, op) != hipSuccess) return ncclSystemError;
    } else {    // the host function waits for the bytes, then staging -> device on the stream
        if (hipLaunchHostFunc(s, on_stream, op) != hipSuccess) return ncclSystemError;
        if (hipMemcpyAsync(buf, st, count, hipMemcpyHostToDevice, s) != hipSuccess) return ncclSystemError;
    }
    return ncclSuccess;
}
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t, int peer, ncclComm_t c, hipStream_t s) { return enqueue(true, const_cast<void*>(buf), count, peer, c, s); }
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t, int peer, ncclComm_t c, hipStream_t s) { return enqueue(false, buf, count, peer, c, s); }
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "ok" : "fake rccl error"; }
}
