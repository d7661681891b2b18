// fake_rccl.cpp — TEST-ONLY stand-in for librccl.so.1.  Never shipped, never linked into the product: the library dlopen()s
// "librccl.so.1" (csrc/zr_dist.cpp: load_rccl), and tests put THIS file's build (tests/fake_rccl/_build/librccl.so.1) ahead of ROCm's on
// LD_LIBRARY_PATH of child processes that do not hold PyTorch's copy.  It lets the native multi-GPU host (zr_dist_prepare / connect /
// frame: the stream and event choreography around the collectives) run with a world > 1 on a box with ONE GPU: every rank is a process
// (or a thread) with a context of its own on that one device, and the "wire" is a POSIX shared-memory segment.
//
// The six entry points the library binds, with RCCL's signatures and semantics as far as the library relies on them:
//   ncclGetUniqueId, ncclCommInitRank (a collective: returns when every rank has attached), ncclCommDestroy, ncclGetErrorString,
//   ncclAllGather, ncclAllReduce (ncclFloat32 + ncclMin only: the shadow map's depth test is a min) - both STREAM-ORDERED: they take their
//   input when the stream reaches them and later work on the stream sees their output.
//
// Two ways of being stream-ordered:
//   * default: D2H copy -> hipLaunchHostFunc (publish my part in the segment, wait for everyone's, assemble the result) -> H2D copy; the
//     call returns at once, like RCCL's.  The host function blocks a runtime thread of ITS process until the other ranks arrive, so
//     ranks must be separate processes;
//   * FAKE_RCCL_BLOCKING=1: the CALL synchronises the stream, exchanges, and enqueues the H2D copy.  Correct for any host that issues
//     its collectives in the same order on every rank (as RCCL demands), and usable with several ranks as threads of one process.
// A rank that waits longer than FAKE_RCCL_TIMEOUT_S (default 60) for the others gives up: the segment's error flag is set, the result is
// filled with 0xCD, later calls return ncclSystemError - a test fails, nothing hangs.
//
// One stream per communicator is assumed (the library uses its collective stream for every call).  Buffers of successive operations
// alternate between two halves of the segment: a rank can only be one operation ahead of the slowest one.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace {

constexpr int kMaxRanks = 64;
struct Segment {                                   // zero-filled by ftruncate
    std::atomic<uint32_t> attached, error;
    std::atomic<uint64_t> arrived[kMaxRanks];      // the last operation rank r has published its part of
    uint64_t pad[7];
};
constexpr size_t kHeader = 4096;
static_assert(sizeof(Segment) <= kHeader, "header");

double timeout_s() { const char* e = getenv("FAKE_RCCL_TIMEOUT_S"); return e ? atof(e) : 60.0; }
size_t slot_bytes() { const char* e = getenv("FAKE_RCCL_SLOT_MB"); return (size_t)(e ? atoi(e) : 40) << 20; }
bool blocking_mode() { const char* e = getenv("FAKE_RCCL_BLOCKING"); return e && atoi(e) != 0; }

}  // namespace

struct ncclComm {
    int rank = 0, n = 1;
    Segment* seg = nullptr; uint8_t* data = nullptr; size_t map_bytes = 0, slot = 0;
    char name[sizeof(ncclUniqueId) + 1] = {};
    uint64_t seq = 0;                              // operations issued on this communicator
    uint8_t* stage_send = nullptr; uint8_t* stage_recv = nullptr; size_t cap_send = 0, cap_recv = 0;
    bool blocking = false;
    uint8_t* part(uint64_t op, int r) const { return data + ((size_t)(op & 1u) * (size_t)n + (size_t)r) * slot; }
};

namespace {

struct Op { ncclComm* c; int kind; size_t bytes; uint64_t seq; };      // kind 0: all-gather, 1: all-reduce(min, f32)

bool wait_all(ncclComm* c, uint64_t seq)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        bool all = true;
        for (int r = 0; r < c->n; ++r) all &= c->seg->arrived[r].load(std::memory_order_acquire) >= seq;
        if (all) return true;
        if (c->seg->error.load()) return false;
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) {
            fprintf(stderr, "fake_rccl: rank %d gave up waiting for operation %llu\n", c->rank, (unsigned long long)seq);
            c->seg->error.store(1);
            return false;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
}

// publish this rank's part, wait for everyone's, assemble the result in stage_recv
void exchange(Op* op)
{
    ncclComm* c = op->c;
    memcpy(c->part(op->seq, c->rank), c->stage_send, op->bytes);
    c->seg->arrived[c->rank].store(op->seq, std::memory_order_release);
    if (!wait_all(c, op->seq)) { memset(c->stage_recv, 0xCD, op->kind == 0 ? op->bytes * (size_t)c->n : op->bytes); return; }
    if (op->kind == 0) {
        for (int r = 0; r < c->n; ++r) memcpy(c->stage_recv + (size_t)r * op->bytes, c->part(op->seq, r), op->bytes);
    } else {
        float* out = (float*)c->stage_recv;
        const size_t k = op->bytes / 4;
        memcpy(out, c->part(op->seq, 0), op->bytes);
        for (int r = 1; r < c->n; ++r) {
            const float* in = (const float*)c->part(op->seq, r);
            for (size_t i = 0; i < k; ++i) out[i] = in[i] < out[i] ? in[i] : out[i];
        }
    }
}

void host_fn(void* p)
{
    Op* op = (Op*)p;
    exchange(op);
    // FAKE_RCCL_CORRUPT=1: one bit of every result flipped - the negative control of the tests (their comparison must notice what the wire carries)
    static const bool corrupt = getenv("FAKE_RCCL_CORRUPT") && atoi(getenv("FAKE_RCCL_CORRUPT")) != 0;
    if (corrupt) op->c->stage_recv[(op->kind == 0 ? op->bytes * (size_t)op->c->n : op->bytes) / 3] ^= 0x40;
    delete op;
}

bool ensure_stage(ncclComm* c, size_t send, size_t recv, hipStream_t s)
{
    if (send <= c->cap_send && recv <= c->cap_recv) return true;
    if (hipStreamSynchronize(s) != hipSuccess) return false;          // nothing in flight may still use the old staging buffers
    if (send > c->cap_send) {
        if (c->stage_send) (void)hipHostFree(c->stage_send);
        if (hipHostMalloc((void**)&c->stage_send, send, hipHostMallocDefault) != hipSuccess) return false;
        c->cap_send = send;
    }
    if (recv > c->cap_recv) {
        if (c->stage_recv) (void)hipHostFree(c->stage_recv);
        if (hipHostMalloc((void**)&c->stage_recv, recv, hipHostMallocDefault) != hipSuccess) return false;
        c->cap_recv = recv;
    }
    return true;
}

ncclResult_t collective(ncclComm* c, int kind, const void* send, void* recv, size_t bytes, hipStream_t s)
{
    if (!c || !c->seg) return ncclInvalidArgument;
    if (c->seg->error.load()) return ncclSystemError;
    if (bytes > c->slot) { fprintf(stderr, "fake_rccl: %zu bytes per rank exceed the slot (FAKE_RCCL_SLOT_MB)\n", bytes); return ncclInvalidArgument; }
    const size_t out_bytes = kind == 0 ? bytes * (size_t)c->n : bytes;
    if (!ensure_stage(c, bytes, out_bytes, s)) return ncclUnhandledCudaError;
    Op* op = new Op{ c, kind, bytes, ++c->seq };
    if (hipMemcpyAsync(c->stage_send, send, bytes, hipMemcpyDeviceToHost, s) != hipSuccess) { delete op; return ncclUnhandledCudaError; }
    if (c->blocking) {
        if (hipStreamSynchronize(s) != hipSuccess) { delete op; return ncclUnhandledCudaError; }
        host_fn(op);
    } else if (hipLaunchHostFunc(s, host_fn, op) != hipSuccess) { delete op; return ncclUnhandledCudaError; }
    if (hipMemcpyAsync(recv, c->stage_recv, out_bytes, hipMemcpyHostToDevice, s) != hipSuccess) return ncclUnhandledCudaError;
    return c->seg->error.load() ? ncclSystemError : ncclSuccess;
}

size_t type_size(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

std::atomic<uint32_t> g_ids{0};

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof *id);
    const uint64_t t = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
    snprintf(id->internal, sizeof id->internal, "/fake_rccl_%d_%u_%llx", (int)getpid(), g_ids.fetch_add(1), (unsigned long long)t);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank)
{
    if (!out || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    if (id.internal[0] != '/' || strncmp(id.internal, "/fake_rccl_", 11) != 0) return ncclInvalidArgument;      // not an id of ours
    ncclComm* c = new ncclComm();
    c->rank = rank; c->n = nranks; c->slot = slot_bytes(); c->blocking = blocking_mode();
    snprintf(c->name, sizeof c->name, "%s", id.internal);
    c->map_bytes = kHeader + 2u * (size_t)nranks * c->slot;
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) { if (fd >= 0) close(fd); delete c; return ncclSystemError; }
    void* m = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);      // sparse: only what is written gets pages
    close(fd);
    if (m == MAP_FAILED) { delete c; return ncclSystemError; }
    c->seg = (Segment*)m; c->data = (uint8_t*)m + kHeader;
    c->seg->attached.fetch_add(1);
    const auto t0 = std::chrono::steady_clock::now();
    while (c->seg->attached.load() < (uint32_t)nranks) {                                     // the bring-up is a collective
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) {
            fprintf(stderr, "fake_rccl: rank %d of %d: the others never attached\n", rank, nranks);
            c->seg->error.store(1);
            munmap(m, c->map_bytes); shm_unlink(c->name); delete c;
            return ncclSystemError;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    fprintf(stderr, "fake_rccl: rank %d of %d attached to %s (%s)\n", rank, nranks, c->name, c->blocking ? "blocking calls" : "host functions");
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (!c) return ncclSuccess;
    if (c->stage_send) (void)hipHostFree(c->stage_send);
    if (c->stage_recv) (void)hipHostFree(c->stage_recv);
    if (c->seg) munmap((void*)c->seg, c->map_bytes);
    shm_unlink(c->name);                           // the first rank to leave removes the name; the mapping lives on for the others
    delete c;
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error (fake_rccl)";
    case ncclUnhandledCudaError: return "a HIP call failed (fake_rccl)";
    case ncclSystemError: return "shared-memory rendezvous failed or timed out (fake_rccl)";
    case ncclInvalidArgument: return "invalid argument (fake_rccl)";
    default: return "error (fake_rccl)";
    }
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t sendcount, ncclDataType_t type, ncclComm_t c, hipStream_t s)
{
    const size_t ts = type_size(type);
    if (!ts || !send || !recv) return ncclInvalidArgument;
    return collective(c, 0, send, recv, sendcount * ts, s);
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t c, hipStream_t s)
{
    if (type != ncclFloat32 || op != ncclMin || !send || !recv) return ncclInvalidArgument;      // all the library asks for
    return collective(c, 1, send, recv, count * 4, s);
}

}  // extern "C"
