// rccl_shim.cpp -- TEST INFRASTRUCTURE, not product code: the handful of RCCL entry points rend3_amd/csrc/comm.h binds,
// implemented between PROCESSES THAT SHARE ONE GPU, so that the library's own multi-rank exchange (r3n_comm_init,
// r3n_render_frame's collectives) executes its N > 1 branches on a one-GPU box.  Real RCCL refuses two ranks on one device.
//
// How: a communicator is a POSIX shared-memory segment named by the unique id (header with a sense-reversing barrier + a data
// area).  Every collective is executed synchronously at enqueue time: wait for the caller's stream, copy the contribution to the
// segment (device -> host), barrier, read / reduce what the other ranks wrote (host -> device), barrier.  Stream order is kept
// (everything enqueued before the call has finished, everything after is enqueued after it returns), and because a call blocks
// until every rank has made the SAME call on the SAME communicator, a rank that issues its collectives in another order than
// its peers deadlocks here and trips the barrier's time-out -- which is the property real RCCL needs from the caller.
//
// R3N_SHIM_ASYNC=1 -- the asynchronous mode.  The synchronous mode above can never show a dependence on the ORDER
// in which a device executes the collectives of DIFFERENT communicators: every call completes before the next one is even issued.
// Here a call only ENQUEUES: device -> host copy of the contribution into the (pinned) segment, a host function on the stream
// (barrier, gather / reduce into a private pinned buffer, barrier), host -> device copy of the result -- and returns.  The collective
// runs when its STREAM reaches it, and the runtime runs host functions one at a time per process: the model of a device with room
// for ONE collective kernel, the worst case RCCL allows itself.  Two ranks whose streams reach the collectives of two communicators
// in different orders then wait for each other and trip the barrier's time-out (R3N_SHIM_TIMEOUT seconds, default 120): exactly the
// deadlock the library's comm_serial order (r3n.hip comm_order_begin) exists to exclude.
//
// Loaded through R3N_RCCL_LIB (comm.h).  Build: hipcc -shared -fPIC -o librccl_shim.so rccl_shim.cpp -lrt  (tests/rccl_shim.py).
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {
constexpr size_t kDataBytesSync = 1ull << 30;   // virtual: pages are touched on use
constexpr size_t kDataBytesAsync = 128ull << 20;  // pinned (hipHostRegister): the copies must really be asynchronous
const bool kAsync = [] { const char *e = std::getenv("R3N_SHIM_ASYNC"); return e && e[0] == '1'; }();
const size_t kDataBytes = kAsync ? kDataBytesAsync : kDataBytesSync;
const int kTimeoutSeconds = [] { const char *e = std::getenv("R3N_SHIM_TIMEOUT"); const int v = e ? std::atoi(e) : 0; return v > 0 ? v : 120; }();

struct Header {
    std::atomic<uint32_t> arrived;
    std::atomic<uint32_t> generation;
    std::atomic<uint32_t> failed;
    uint32_t _pad[13];
};
static_assert(sizeof(Header) == 64, "one cache line");
}  // namespace

struct ncclComm {
    int rank = 0, world = 1;
    Header *hdr = nullptr;
    char *data = nullptr;
    size_t mapped = 0;
    char name[64] = {};
    // asynchronous mode
    bool registered = false;
    char *result = nullptr;       // private pinned buffer: what the host function gathered / reduced, source of the copy back
    size_t result_bytes = 0;
    hipStream_t last = nullptr;   // the stream of the previous operation (operations of one communicator are kept in order)
    hipEvent_t hand_over = nullptr;
    std::atomic<int> async_failed{0};
};

namespace {
thread_local const char *g_last_error = "no error";

bool barrier(ncclComm *c) {
    Header *h = c->hdr;
    if (h->failed.load()) return false;
    const uint32_t gen = h->generation.load(std::memory_order_acquire);
    if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1u == (uint32_t)c->world) {
        h->arrived.store(0, std::memory_order_relaxed);
        h->generation.store(gen + 1u, std::memory_order_release);
        return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t spins = 0;
    while (h->generation.load(std::memory_order_acquire) == gen) {
        if (h->failed.load()) return false;
        if ((++spins & 1023u) == 0u) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(kTimeoutSeconds)) {
                h->failed.store(1);
                g_last_error = "rccl_shim: barrier timed out (a rank did not issue the same collective: order mismatch or a dead peer)";
                return false;
            }
            sched_yield();
        }
    }
    return true;
}

size_t type_bytes(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}

template <class T> void reduce_into(T *acc, const T *src, size_t n, ncclRedOp_t op) {
    switch (op) {
        case ncclMax: for (size_t i = 0; i < n; ++i) acc[i] = src[i] > acc[i] ? src[i] : acc[i]; break;
        case ncclMin: for (size_t i = 0; i < n; ++i) acc[i] = src[i] < acc[i] ? src[i] : acc[i]; break;
        case ncclSum: for (size_t i = 0; i < n; ++i) acc[i] = (T)(acc[i] + src[i]); break;
        case ncclProd: for (size_t i = 0; i < n; ++i) acc[i] = (T)(acc[i] * src[i]); break;
        default: break;
    }
}
bool reduce_typed(void *acc, const void *src, size_t n, ncclDataType_t t, ncclRedOp_t op) {
    switch (t) {
        case ncclInt8: reduce_into((int8_t *)acc, (const int8_t *)src, n, op); return true;
        case ncclUint8: reduce_into((uint8_t *)acc, (const uint8_t *)src, n, op); return true;
        case ncclInt32: reduce_into((int32_t *)acc, (const int32_t *)src, n, op); return true;
        case ncclUint32: reduce_into((uint32_t *)acc, (const uint32_t *)src, n, op); return true;
        case ncclInt64: reduce_into((int64_t *)acc, (const int64_t *)src, n, op); return true;
        case ncclUint64: reduce_into((uint64_t *)acc, (const uint64_t *)src, n, op); return true;
        case ncclFloat32: reduce_into((float *)acc, (const float *)src, n, op); return true;
        case ncclFloat64: reduce_into((double *)acc, (const double *)src, n, op); return true;
        default: return false;
    }
}

#define SHIM_HIP(expr)                                                          \
    do {                                                                        \
        if ((expr) != hipSuccess) { g_last_error = "rccl_shim: " #expr " failed"; c->hdr->failed.store(1); return ncclUnhandledCudaError; } \
    } while (0)
#define SHIM_BARRIER()                                       \
    do {                                                     \
        if (!barrier(c)) return ncclSystemError;             \
    } while (0)

// ---- asynchronous mode: what runs on the stream
struct Op {
    ncclComm *c;
    enum Kind { Gather, Bcast, Reduce } kind;
    size_t bytes = 0;        // Gather: per rank; Bcast: message; Reduce: bytes of one rank's whole contribution
    size_t first = 0, n_out = 0;
    ncclDataType_t t = ncclUint8;
    ncclRedOp_t op = ncclMax;
    int root = 0;
};
void run_op(void *p) {
    Op *o = static_cast<Op *>(p);
    ncclComm *c = o->c;
    bool ok = barrier(c);  // every rank's contribution is in the segment (its copy is in front of its host function)
    if (ok) {
        switch (o->kind) {
            case Op::Gather: std::memcpy(c->result, c->data, o->bytes * (size_t)c->world); break;
            case Op::Bcast: if (c->rank != o->root) std::memcpy(c->result, c->data, o->bytes); break;
            case Op::Reduce: {
                const size_t tb = type_bytes(o->t);
                std::memcpy(c->result, c->data + o->first * tb, o->n_out * tb);
                for (int r = 1; r < c->world; ++r) ok = ok && reduce_typed(c->result, c->data + (size_t)r * o->bytes + o->first * tb, o->n_out, o->t, o->op);
                break;
            }
        }
    }
    ok = barrier(c) && ok;  // everybody has read the segment: the next operation may overwrite it
    if (!ok) c->async_failed.store(1);
    delete o;
}
// operations of ONE communicator stay in order even when the caller moves it to another stream; the result buffer grows here
ncclResult_t async_begin(ncclComm *c, hipStream_t s, size_t result_bytes) {
    if (c->async_failed.load() || c->hdr->failed.load()) { g_last_error = "rccl_shim: an earlier asynchronous collective failed (barrier time-out: the ranks' streams reached the communicators' collectives in different orders, or a peer died)"; return ncclSystemError; }
    if (c->last && c->last != s) {
        if (hipEventRecord(c->hand_over, c->last) != hipSuccess || hipStreamWaitEvent(s, c->hand_over, 0) != hipSuccess) { g_last_error = "rccl_shim: stream hand-over"; return ncclUnhandledCudaError; }
    }
    if (result_bytes > c->result_bytes) {
        if (c->last && hipStreamSynchronize(c->last) != hipSuccess) { g_last_error = "rccl_shim: sync before growing the result buffer"; return ncclUnhandledCudaError; }
        (void)hipHostFree(c->result);
        c->result = nullptr;
        c->result_bytes = result_bytes + result_bytes / 2;
        if (hipHostMalloc((void **)&c->result, c->result_bytes, hipHostMallocDefault) != hipSuccess) { g_last_error = "rccl_shim: result buffer"; return ncclUnhandledCudaError; }
    }
    c->last = s;
    return ncclSuccess;
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    static std::atomic<uint32_t> counter{0};
    std::memset(id->internal, 0, sizeof id->internal);
    unsigned r = 0;
    if (FILE *f = std::fopen("/dev/urandom", "rb")) { (void)!std::fread(&r, sizeof r, 1, f); std::fclose(f); }
    std::snprintf(id->internal, sizeof id->internal, "/r3nshim-%d-%u-%08x", (int)getpid(), counter.fetch_add(1), r);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank) {
    if (!out || nranks < 1 || rank < 0 || rank >= nranks || id.internal[0] != '/') { g_last_error = "rccl_shim: bad init arguments"; return ncclInvalidArgument; }
    ncclComm *c = new ncclComm;
    c->rank = rank; c->world = nranks;
    std::snprintf(c->name, sizeof c->name, "%s", id.internal);
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) { g_last_error = "rccl_shim: shm_open failed"; delete c; return ncclSystemError; }
    c->mapped = sizeof(Header) + kDataBytes;
    if (ftruncate(fd, (off_t)c->mapped) != 0) { g_last_error = "rccl_shim: ftruncate failed"; close(fd); delete c; return ncclSystemError; }
    void *p = mmap(nullptr, c->mapped, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { g_last_error = "rccl_shim: mmap failed"; delete c; return ncclSystemError; }
    c->hdr = static_cast<Header *>(p);  // a fresh segment is zero-filled: counters start at 0
    c->data = static_cast<char *>(p) + sizeof(Header);
    if (kAsync) {
        if (hipHostRegister(p, c->mapped, hipHostRegisterPortable) != hipSuccess) { g_last_error = "rccl_shim: hipHostRegister of the segment failed"; munmap(p, c->mapped); delete c; return ncclUnhandledCudaError; }
        c->registered = true;
        c->result_bytes = 32ull << 20;
        if (hipHostMalloc((void **)&c->result, c->result_bytes, hipHostMallocDefault) != hipSuccess || hipEventCreateWithFlags(&c->hand_over, hipEventDisableTiming) != hipSuccess) {
            g_last_error = "rccl_shim: pinned result buffer"; (void)hipHostUnregister(p); munmap(p, c->mapped); delete c; return ncclUnhandledCudaError;
        }
    }
    if (!barrier(c)) { munmap(p, c->mapped); delete c; return ncclSystemError; }  // everybody is attached ...
    if (rank == 0) shm_unlink(c->name);                                         // ... so the name can go: the segment dies with its last user
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclSuccess;
    if (c->last) (void)hipStreamSynchronize(c->last);  // (operations still on the stream hold pointers into the communicator)
    if (c->registered) (void)hipHostUnregister(c->hdr);
    if (c->result) (void)hipHostFree(c->result);
    if (c->hand_over) (void)hipEventDestroy(c->hand_over);
    if (c->hdr) munmap(c->hdr, c->mapped);
    delete c;
    return ncclSuccess;
}
ncclResult_t ncclCommCount(const ncclComm_t c, int *n) { *n = c->world; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t c, int *r) { *r = c->rank; return ncclSuccess; }
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : g_last_error; }
const char *ncclGetLastError(ncclComm_t) { return g_last_error; }

// Grouped calls run one after the other: every rank issues a group's members in the same order.
ncclResult_t ncclGroupStart() { return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return ncclSuccess; }

ncclResult_t ncclAllGather(const void *send, void *recv, size_t sendcount, ncclDataType_t t, ncclComm_t c, hipStream_t s) {
    const size_t bytes = sendcount * type_bytes(t);
    if (!bytes) return ncclSuccess;
    if (bytes * (size_t)c->world > kDataBytes) { g_last_error = "rccl_shim: message beyond the staging area"; return ncclInvalidArgument; }
    if (kAsync) {
        if (ncclResult_t r = async_begin(c, s, bytes * (size_t)c->world)) return r;
        SHIM_HIP(hipMemcpyAsync(c->data + (size_t)c->rank * bytes, send, bytes, hipMemcpyDeviceToHost, s));
        Op *o = new Op{c, Op::Gather};
        o->bytes = bytes;
        SHIM_HIP(hipLaunchHostFunc(s, run_op, o));
        for (int r = 0; r < c->world; ++r) {
            char *dst = static_cast<char *>(recv) + (size_t)r * bytes;
            if (r == c->rank && dst == send) continue;
            SHIM_HIP(hipMemcpyAsync(dst, c->result + (size_t)r * bytes, bytes, hipMemcpyHostToDevice, s));
        }
        return ncclSuccess;
    }
    SHIM_HIP(hipMemcpyAsync(c->data + (size_t)c->rank * bytes, send, bytes, hipMemcpyDeviceToHost, s));
    SHIM_HIP(hipStreamSynchronize(s));
    SHIM_BARRIER();
    for (int r = 0; r < c->world; ++r) {
        char *dst = static_cast<char *>(recv) + (size_t)r * bytes;
        if (r == c->rank && dst == send) continue;  // in place: the own chunk is where it belongs
        SHIM_HIP(hipMemcpyAsync(dst, c->data + (size_t)r * bytes, bytes, hipMemcpyHostToDevice, s));
    }
    SHIM_HIP(hipStreamSynchronize(s));
    SHIM_BARRIER();  // the staging area is free again
    return ncclSuccess;
}

ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t t, int root, ncclComm_t c, hipStream_t s) {
    const size_t bytes = count * type_bytes(t);
    if (!bytes) return ncclSuccess;
    if (bytes > kDataBytes || root < 0 || root >= c->world) { g_last_error = "rccl_shim: bad broadcast"; return ncclInvalidArgument; }
    if (kAsync) {
        if (ncclResult_t r = async_begin(c, s, bytes)) return r;
        if (c->rank == root) {
            SHIM_HIP(hipMemcpyAsync(c->data, send, bytes, hipMemcpyDeviceToHost, s));
            if (recv != send) SHIM_HIP(hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, s));
        }
        Op *o = new Op{c, Op::Bcast};
        o->bytes = bytes; o->root = root;
        SHIM_HIP(hipLaunchHostFunc(s, run_op, o));
        if (c->rank != root) SHIM_HIP(hipMemcpyAsync(recv, c->result, bytes, hipMemcpyHostToDevice, s));
        return ncclSuccess;
    }
    if (c->rank == root) {
        SHIM_HIP(hipMemcpyAsync(c->data, send, bytes, hipMemcpyDeviceToHost, s));
        if (recv != send) SHIM_HIP(hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, s));
    }
    SHIM_HIP(hipStreamSynchronize(s));
    SHIM_BARRIER();
    if (c->rank != root) {
        SHIM_HIP(hipMemcpyAsync(recv, c->data, bytes, hipMemcpyHostToDevice, s));
        SHIM_HIP(hipStreamSynchronize(s));
    }
    SHIM_BARRIER();
    return ncclSuccess;
}
ncclResult_t ncclBcast(void *buff, size_t count, ncclDataType_t t, int root, ncclComm_t c, hipStream_t s) { return ncclBroadcast(buff, buff, count, t, root, c, s); }

// every rank's contribution side by side in the segment; each rank reduces the slice it needs on the host
static ncclResult_t reduce_common(const void *send, void *recv, size_t first, size_t n_out, size_t n_all, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t s) {
    const size_t tb = type_bytes(t), bytes_all = n_all * tb;
    if (!bytes_all) return ncclSuccess;
    if (!tb || bytes_all * (size_t)c->world > kDataBytes) { g_last_error = "rccl_shim: reduction type / size not supported"; return ncclInvalidArgument; }
    if (kAsync) {
        if (ncclResult_t r = async_begin(c, s, n_out * tb)) return r;
        SHIM_HIP(hipMemcpyAsync(c->data + (size_t)c->rank * bytes_all, send, bytes_all, hipMemcpyDeviceToHost, s));
        Op *o = new Op{c, Op::Reduce};
        o->bytes = bytes_all; o->first = first; o->n_out = n_out; o->t = t; o->op = op;
        SHIM_HIP(hipLaunchHostFunc(s, run_op, o));
        SHIM_HIP(hipMemcpyAsync(recv, c->result, n_out * tb, hipMemcpyHostToDevice, s));
        return ncclSuccess;
    }
    SHIM_HIP(hipMemcpyAsync(c->data + (size_t)c->rank * bytes_all, send, bytes_all, hipMemcpyDeviceToHost, s));
    SHIM_HIP(hipStreamSynchronize(s));
    SHIM_BARRIER();
    std::vector<char> acc(n_out * tb);
    std::memcpy(acc.data(), c->data + first * tb, n_out * tb);
    for (int r = 1; r < c->world; ++r)
        if (!reduce_typed(acc.data(), c->data + (size_t)r * bytes_all + first * tb, n_out, t, op)) { g_last_error = "rccl_shim: reduction type not supported"; return ncclInvalidArgument; }
    SHIM_BARRIER();  // everybody has read the segment
    SHIM_HIP(hipMemcpyAsync(recv, acc.data(), n_out * tb, hipMemcpyHostToDevice, s));
    SHIM_HIP(hipStreamSynchronize(s));  // `acc` is a temporary
    return ncclSuccess;
}
ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t s) {
    return reduce_common(send, recv, 0, count, count, t, op, c, s);
}
ncclResult_t ncclReduceScatter(const void *send, void *recv, size_t recvcount, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t s) {
    return reduce_common(send, recv, (size_t)c->rank * recvcount, recvcount, recvcount * (size_t)c->world, t, op, c, s);
}

}  // extern "C"
