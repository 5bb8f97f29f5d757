// fake_rccl.cpp -- TEST INFRASTRUCTURE, never linked or loaded by the product unless a test says HIPETS_RCCL_LIB=<this .so>.
//
// A stand-in for the five RCCL entry points libhipets binds through dlopen (hipets.hip rccl_load: ncclGetUniqueId,
// ncclCommInitRank, ncclAllGather, ncclCommDestroy, ncclGetErrorString, + the optional ncclCommCount / ncclCommUserRank), for
// N ranks that are N PROCESSES SHARING ONE GPU.  It exists so that hipets_plan_cem_sharded's world > 1 path -- uneven candidate
// shards, the padded all-gather, the unpad kernel, the rank-offset rollout seeds, the error path -- executes on the one-GPU
// boxes this repository is built and graded on; it says nothing about xGMI performance.
//
// Transport: a POSIX shared-memory segment named by the unique id.  ncclAllGather synchronises the caller's stream, copies the
// rank's shard device -> segment, meets the other ranks at a sense-reversing barrier, copies all shards segment -> device and
// meets them again (so nobody overwrites a slot a peer still reads).  Stream semantics: everything enqueued on `stream` before
// the call has completed when it returns and the gathered data is in place -- a (synchronous) special case of what RCCL
// guarantees, so the library's call sequence is exercised unchanged.
//
// FAKE_RCCL_HOST_BUFFERS=1: sendbuff / recvbuff are HOST pointers and `stream` is ignored (no HIP call is made): lets the CPU
// test suite run the barrier / gather logic itself with 8 processes and no GPU (tests/test_fake_rccl_host.py).
//
// Fault injection (same environment on every rank): FAKE_RCCL_FAIL_AT=k makes the k-th ncclAllGather of a communicator
// return ncclSystemError on every rank before any data moves.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

constexpr size_t kDataBytes = 8u << 20;  // room for world x count elements of one all-gather

struct Shared {
    std::atomic<int> arrived;
    std::atomic<int> generation;
    std::atomic<int> attached;
    int world;
    alignas(64) char data[kDataBytes];
};

struct Comm {
    Shared* sh;
    int rank, world;
    long calls;
    long fail_at;
    char name[128];
};

// all `world` ranks meet; false after 30 s without the others (a peer died: report instead of hanging the test)
bool barrier(Comm* c) {
    Shared* sh = c->sh;
    const int gen = sh->generation.load(std::memory_order_acquire);
    if (sh->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == c->world) {
        sh->arrived.store(0, std::memory_order_relaxed);
        sh->generation.fetch_add(1, std::memory_order_release);
        return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (sh->generation.load(std::memory_order_acquire) == gen) {
        std::this_thread::sleep_for(std::chrono::microseconds(20));
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) return false;
    }
    return true;
}

size_t dtype_bytes(int dt) {  // ncclDataType_t of NCCL 2.x: int8 0, uint8 1, int32 2, uint32 3, int64 4, uint64 5, half 6, float 7, double 8, bf16 9
    switch (dt) {
        case 0: case 1: return 1;
        case 6: case 9: return 2;
        case 2: case 3: case 7: return 4;
        case 4: case 5: case 8: return 8;
        default: return 0;
    }
}

}  // namespace

extern "C" {

typedef struct { char internal[128]; } ncclUniqueId;

int ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return 4;  // ncclInvalidArgument
    static std::atomic<int> counter{0};
    std::memset(id->internal, 0, sizeof(id->internal));
    const long long now = (long long)std::chrono::steady_clock::now().time_since_epoch().count();
    std::snprintf(id->internal, sizeof(id->internal), "/hipets_fake_rccl_%d_%d_%llx", (int)getpid(), counter.fetch_add(1), (unsigned long long)now);
    return 0;
}

int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks || id.internal[0] != '/') return 4;
    id.internal[sizeof(id.internal) - 1] = 0;
    const int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return 2;  // ncclSystemError
    if (ftruncate(fd, sizeof(Shared)) != 0) { close(fd); return 2; }
    void* p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return 2;
    Comm* c = new Comm();
    c->sh = static_cast<Shared*>(p);  // a fresh segment is zero-filled: every counter starts at 0
    c->rank = rank;
    c->world = nranks;
    c->calls = 0;
    const char* f = std::getenv("FAKE_RCCL_FAIL_AT");
    c->fail_at = f ? std::atol(f) : 0;
    std::strncpy(c->name, id.internal, sizeof(c->name) - 1);
    c->sh->world = nranks;
    c->sh->attached.fetch_add(1);
    if (!barrier(c)) {  // everybody has mapped the segment
        munmap(p, sizeof(Shared));
        delete c;
        return 2;
    }
    if (rank == 0) shm_unlink(c->name);  // the mappings keep it alive; nothing is left behind if a rank dies
    *comm = c;
    return 0;
}

int ncclCommDestroy(void* comm) {
    Comm* c = static_cast<Comm*>(comm);
    if (!c) return 4;
    munmap(c->sh, sizeof(Shared));
    delete c;
    return 0;
}

int ncclCommCount(void* comm, int* count) {
    if (!comm || !count) return 4;
    *count = static_cast<Comm*>(comm)->world;
    return 0;
}

int ncclCommUserRank(void* comm, int* rank) {
    if (!comm || !rank) return 4;
    *rank = static_cast<Comm*>(comm)->rank;
    return 0;
}

int ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, int datatype, void* comm, hipStream_t stream) {
    Comm* c = static_cast<Comm*>(comm);
    const size_t eb = dtype_bytes(datatype);
    if (!c || !sendbuff || !recvbuff || eb == 0) return 4;
    const size_t bytes = sendcount * eb;
    if (bytes * (size_t)c->world > kDataBytes) return 4;
    c->calls += 1;
    if (c->fail_at > 0 && c->calls == c->fail_at) return 2;  // injected on every rank alike, before anybody waits for anybody
    const char* hb = std::getenv("FAKE_RCCL_HOST_BUFFERS");
    const bool host = hb && hb[0] == '1';
    if (host) {
        std::memcpy(c->sh->data + (size_t)c->rank * bytes, sendbuff, bytes);
    } else {
        if (hipStreamSynchronize(stream) != hipSuccess) return 1;  // ncclUnhandledCudaError
        if (hipMemcpy(c->sh->data + (size_t)c->rank * bytes, sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    }
    if (!barrier(c)) return 2;
    if (host) std::memcpy(recvbuff, c->sh->data, bytes * (size_t)c->world);
    else if (hipMemcpy(recvbuff, c->sh->data, bytes * (size_t)c->world, hipMemcpyHostToDevice) != hipSuccess) return 1;
    if (!barrier(c)) return 2;
    return 0;
}

const char* ncclGetErrorString(int r) {
    switch (r) {
        case 0: return "no error";
        case 1: return "unhandled cuda error";
        case 2: return "unhandled system error";
        case 3: return "internal error";
        case 4: return "invalid argument";
        default: return "unknown result code";
    }
}

}  // extern "C"
