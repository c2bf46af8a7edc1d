// mock_rccl.cpp — TEST INFRASTRUCTURE.  A stand-in for librccl.so with the ten entry points the library's multi-GPU seam
// binds (cray_hip.hip, struct Rccl), so that the N > 1 code path — communicator set-up, scene broadcast, the grouped
// send / receive gather of Film tiles, barrier and all-reduce — can run with SEVERAL RANKS ON ONE GPU, which real RCCL refuses.
// Transport: a POSIX shared-memory segment named after the unique id, one 4 MiB mailbox per (source, destination) pair,
// device buffers staged through it with hipMemcpy.  It checks what the caller does, not what RCCL does: no xGMI, no
// collective algorithms.  Loaded through CRAY_RCCL_LIB=<this .so>; never shipped, never loaded by the product by itself.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

namespace {
constexpr size_t kBox = 4u << 20;
constexpr int kMaxRanks = 8;
struct Mailbox {
    std::atomic<uint64_t> sent, taken;   // chunks published / consumed
    uint64_t bytes;                      // size of the chunk in flight
    unsigned char data[kBox];
};
struct Shared {
    std::atomic<uint32_t> attached;
    Mailbox box[kMaxRanks][kMaxRanks];   // [src][dst]
};
struct Op { bool send; char* buf; size_t bytes, done; int peer; bool started; };
// Fault injection for the tests of the library's error paths:
//   MOCK_RCCL_FAIL_SEND=<rank>:<n>   the n-th ncclSend (1-based) the library itself calls on that rank returns an error and sends nothing
//   MOCK_RCCL_STUCK_MS=<ms>          how long a receive waits for a peer that never sends before it gives up (default 60 000)
int g_group_depth = 0;      // ncclGroupStart - ncclGroupEnd of this process: must be 0 when the communicator goes away
int g_sends = 0;            // ncclSend calls made by the library (not the sends inside this file's own collectives)
uint64_t stuck_limit() {
    const char* e = getenv("MOCK_RCCL_STUCK_MS");
    const long ms = e ? atol(e) : 60000;
    return (uint64_t)(ms > 0 ? ms : 1) * 10;   // polls of 100 us
}
bool send_should_fail(int rank) {
    const char* e = getenv("MOCK_RCCL_FAIL_SEND");
    int r = -1, n = -1;
    return e && sscanf(e, "%d:%d", &r, &n) == 2 && r == rank && n == g_sends;
}
}  // namespace

struct ncclComm {
    int rank, world;
    Shared* sh;
    char name[80];
    int group_depth;
    std::vector<Op> pending;
};

static size_t dtype_size(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        default: return 8;
    }
}
// one step of an operation; returns true when it made progress
static bool progress(ncclComm* c, Op& op) {
    if (op.done == op.bytes && op.started) return false;
    if (op.send) {
        Mailbox& m = c->sh->box[c->rank][op.peer];
        if (m.sent.load(std::memory_order_acquire) != m.taken.load(std::memory_order_acquire)) return false;   // box still full
        const size_t n = op.bytes - op.done < kBox ? op.bytes - op.done : kBox;
        if (n && hipMemcpy(m.data, op.buf + op.done, n, hipMemcpyDeviceToHost) != hipSuccess) { fprintf(stderr, "mock_rccl: D2H failed\n"); _exit(3); }
        m.bytes = n;
        m.sent.fetch_add(1, std::memory_order_release);
        op.done += n; op.started = true;
        return true;
    }
    Mailbox& m = c->sh->box[op.peer][c->rank];
    if (m.sent.load(std::memory_order_acquire) == m.taken.load(std::memory_order_acquire)) return false;       // nothing there yet
    const size_t n = m.bytes;
    if (n > op.bytes - op.done) { fprintf(stderr, "mock_rccl: rank %d got %zu bytes from %d, expected at most %zu\n", c->rank, n, op.peer, op.bytes - op.done); _exit(4); }
    if (n && hipMemcpy(op.buf + op.done, m.data, n, hipMemcpyHostToDevice) != hipSuccess) { fprintf(stderr, "mock_rccl: H2D failed\n"); _exit(3); }
    m.taken.fetch_add(1, std::memory_order_release);
    op.done += n; op.started = true;
    return true;
}
static ncclResult_t run(ncclComm* c, hipStream_t stream) {
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;   // the real thing is stream-ordered
    uint64_t idle = 0;
    for (;;) {
        bool all = true, moved = false;
        for (Op& op : c->pending) {
            if (!(op.started && op.done == op.bytes)) { all = false; moved |= progress(c, op); }
        }
        if (all) break;
        if (!moved) {
            if (++idle > stuck_limit()) {
                // give up on what never arrived; the mailboxes are untouched, so the communicator stays usable
                fprintf(stderr, "mock_rccl: rank %d gave up waiting for a peer\n", c->rank);
                c->pending.clear();
                return ncclInternalError;
            }
            usleep(100);
        } else idle = 0;
    }
    c->pending.clear();
    return ncclSuccess;
}
static ncclResult_t post(ncclComm* c, bool send, const void* buf, size_t bytes, int peer, hipStream_t stream) {
    if (peer < 0 || peer >= c->world || peer == c->rank) return ncclInvalidArgument;
    c->pending.push_back(Op{send, (char*)buf, bytes, 0, peer, false});
    return c->group_depth ? ncclSuccess : run(c, stream);
}

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    std::random_device rd;
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "/cray_mock_%08x%08x", rd(), rd());
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    ncclComm* c = new ncclComm();
    c->rank = rank; c->world = nranks; c->group_depth = 0;
    snprintf(c->name, sizeof(c->name), "%s", id.internal);
    int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(Shared)) != 0) return ncclSystemError;   // a fresh segment is zero-filled: all counters 0
    c->sh = (Shared*)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->sh == MAP_FAILED) return ncclSystemError;
    c->sh->attached.fetch_add(1);
    for (uint64_t spin = 0; c->sh->attached.load() < (uint32_t)nranks; spin++) {   // like the real call: returns when everybody is in
        if (spin > 600000) return ncclInternalError;
        usleep(100);
    }
    *comm = c;
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclSuccess;
    if (g_group_depth != 0) { fprintf(stderr, "mock_rccl: rank %d left %d group(s) open\n", c->rank, g_group_depth); _exit(5); }
    if (c->sh->attached.fetch_sub(1) == 1) shm_unlink(c->name);
    munmap(c->sh, sizeof(Shared));
    delete c;
    return ncclSuccess;
}
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "mock_rccl error"; }
ncclResult_t ncclGroupStart() { g_group_depth++; return ncclSuccess; }   // operations still run when they are posted (see below)
ncclResult_t ncclGroupEnd() { if (g_group_depth <= 0) return ncclInvalidUsage; g_group_depth--; return ncclSuccess; }
ncclResult_t ncclGetVersion(int* v) { if (v) *v = 0; return ncclSuccess; }   // 0: not a release of the real library
}

// The library brackets its gather with GroupStart / GroupEnd and posts only receives (rank 0) or one send (the others) inside,
// so executing each operation when it is posted cannot deadlock; the group calls are accepted and ignored.
extern "C" {
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) {
    g_sends++;
    if (send_should_fail(c->rank)) { fprintf(stderr, "mock_rccl: rank %d: injected failure of ncclSend #%d\n", c->rank, g_sends); return ncclSystemError; }
    return post(c, true, buf, count * dtype_size(t), peer, s);
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) { return post(c, false, buf, count * dtype_size(t), peer, s); }
ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t t, int root, ncclComm_t c, hipStream_t s) {
    const size_t bytes = count * dtype_size(t);
    if (c->rank == root) {
        if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
        if (send != recv && bytes && hipMemcpy(recv, send, bytes, hipMemcpyDeviceToDevice) != hipSuccess) return ncclUnhandledCudaError;
        for (int r = 0; r < c->world; r++)
            if (r != root) { ncclResult_t e = post(c, true, send, bytes, r, s); if (e != ncclSuccess) return e; }
        return ncclSuccess;
    }
    return post(c, false, recv, bytes, root, s);
}
ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t s) {
    if (t != ncclDouble && t != ncclFloat64) return ncclInvalidArgument;   // all the library reduces
    const size_t bytes = count * 8;
    if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
    std::vector<double> acc(count), tmp(count);
    if (bytes && hipMemcpy(acc.data(), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    double* scratch = nullptr;
    if (hipMalloc((void**)&scratch, bytes ? bytes : 8) != hipSuccess) return ncclUnhandledCudaError;
    ncclResult_t e = ncclSuccess;
    if (c->rank == 0) {   // reduce at rank 0 in rank order, then hand the result out
        for (int r = 1; r < c->world && e == ncclSuccess; r++) {
            e = post(c, false, scratch, bytes, r, s);
            if (e == ncclSuccess && bytes && hipMemcpy(tmp.data(), scratch, bytes, hipMemcpyDeviceToHost) != hipSuccess) e = ncclUnhandledCudaError;
            for (size_t i = 0; i < count; i++)
                acc[i] = op == ncclSum ? acc[i] + tmp[i] : (op == ncclMax ? (tmp[i] > acc[i] ? tmp[i] : acc[i]) : (tmp[i] < acc[i] ? tmp[i] : acc[i]));
        }
        if (e == ncclSuccess && bytes && hipMemcpy(recv, acc.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) e = ncclUnhandledCudaError;
        for (int r = 1; r < c->world && e == ncclSuccess; r++) e = post(c, true, recv, bytes, r, s);
    } else {
        if (bytes && hipMemcpy(scratch, acc.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) e = ncclUnhandledCudaError;
        if (e == ncclSuccess) e = post(c, true, scratch, bytes, 0, s);
        if (e == ncclSuccess) e = post(c, false, recv, bytes, 0, s);
    }
    (void)hipFree(scratch);
    return e;
}
}
