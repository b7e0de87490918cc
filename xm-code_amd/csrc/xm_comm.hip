// xm_comm.hip — row-partition communicator: RCCL over xGMI, one process per GPU.
//
// The reference is single-GPU (memory.h:54 gpu_id == 0, no NCCL anywhere), so this is new design:
// cameras are split in contiguous equal ranges; per Q*W product one in-place all-gather of the product input W
// (3*nloc*OP doubles per rank) and, per tCG iteration, two all-gathers of a few hundred partial sums — gathered rather
// than all-reduced so that every rank adds them in the same fixed order and takes bit-identical branch decisions.
// RCCL is dlopen()ed at run time (torch bundles its own librccl.so with the same soname; whichever the process already
// loaded is reused, otherwise /opt/rocm/lib/librccl.so).  No link-time dependency, no collective unless world > 1.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <vector>

#include <cstdlib>
#include <cstring>

#include "xm_solver.h"

namespace xm {

namespace {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
constexpr int kNcclFloat64 = 8;  // rccl.h: ncclFloat64 = 8

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclComm_t comm = nullptr;
};
Rccl g_rccl;
Comm g_comm;

// Test transport: the same collectives through a POSIX shared-memory segment (device -> host -> shm -> host -> device), so
// that several ranks can share ONE GPU on a 1-GPU box and the whole row-partitioned solver (partition, padding cameras,
// identical collective counts on every rank) can be exercised there.  Never used when RCCL is initialised.
struct ShmHeader {
    std::atomic<unsigned long long> arrive;   // monotonically increasing barrier counter
};
struct Shm {
    void *base = nullptr;
    size_t bytes = 0;
    unsigned long long barriers = 0;           // barriers completed by this rank
    std::string name;
    std::vector<double> host;
    bool async = false;                        // XM_SHM_ASYNC=1: stream-ordered exchange (host functions on the stream)
    double *stage = nullptr;                   // pinned staging buffer of the stream-ordered variant
    std::atomic<int> err{0};                   // set by a host function that timed out (it cannot throw)
    bool active() const { return base != nullptr; }
};
Shm g_shm;

void shm_barrier(int world) {
    ShmHeader *h = static_cast<ShmHeader *>(g_shm.base);
    h->arrive.fetch_add(1, std::memory_order_acq_rel);
    g_shm.barriers++;
    const unsigned long long target = g_shm.barriers * (unsigned long long)world;
    const auto t0 = std::chrono::steady_clock::now();
    while (h->arrive.load(std::memory_order_acquire) < target) {
        static const double limit = [] { const char *e = std::getenv("XM_SHM_TIMEOUT"); return e ? std::atof(e) : 120.0; }();
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit)
            throw Error(XM_ERR_COMM, "shared-memory communicator: rank " + std::to_string(g_comm.rank) + " waited too long in barrier #" +
                                         std::to_string(g_shm.barriers) + " (arrived " + std::to_string(h->arrive.load()) +
                                         "): ranks issued different collectives?");
    }
}

// barrier for code that must not throw (host functions executed by the stream); false on timeout
bool shm_barrier_nothrow(int world) {
    ShmHeader *h = static_cast<ShmHeader *>(g_shm.base);
    h->arrive.fetch_add(1, std::memory_order_acq_rel);
    g_shm.barriers++;
    const unsigned long long target = g_shm.barriers * (unsigned long long)world;
    static const double limit = [] { const char *e = std::getenv("XM_SHM_TIMEOUT"); return e ? std::atof(e) : 120.0; }();
    const auto t0 = std::chrono::steady_clock::now();
    while (h->arrive.load(std::memory_order_acquire) < target)
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) return false;
    return true;
}
// Stream-ordered variant of the test transport (XM_SHM_ASYNC=1): the exchange runs inside a host function that the STREAM
// executes between the device-to-host and host-to-device copies, so the calling thread never blocks — exactly like an RCCL
// collective.  The solver's enqueue-ahead logic is then exercised for real: a rank that enqueues a different number of
// collectives than its peers leaves a barrier unmatched (time-out -> error flag -> XM_ERR_COMM at the next collective).
struct ShmOp { size_t count; };
void shm_exchange_cb(void *p) {
    ShmOp *op = static_cast<ShmOp *>(p);
    const size_t count = op->count;
    delete op;
    if (g_shm.err.load()) return;
    double *data = reinterpret_cast<double *>(static_cast<char *>(g_shm.base) + sizeof(ShmHeader) + 64);
    std::memcpy(data + (size_t)g_comm.rank * count, g_shm.stage + (size_t)g_comm.rank * count, count * sizeof(double));
    if (!shm_barrier_nothrow(g_comm.world)) { g_shm.err.store(1); return; }
    std::memcpy(g_shm.stage, data, count * (size_t)g_comm.world * sizeof(double));
    if (!shm_barrier_nothrow(g_comm.world)) g_shm.err.store(1);
}

void load_rccl(const char *path) {
    if (g_rccl.handle) return;
    const char *cands[] = {path, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char *c : cands) {
        if (!c || !*c) continue;
        g_rccl.handle = dlopen(c, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.handle) break;
    }
    if (!g_rccl.handle) throw Error(XM_ERR_COMM, std::string("cannot load RCCL: ") + dlerror());
    auto sym = [&](const char *n) {
        void *p = dlsym(g_rccl.handle, n);
        if (!p) throw Error(XM_ERR_COMM, std::string("RCCL symbol missing: ") + n);
        return p;
    };
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))sym("ncclAllGather");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
}
void check(ncclResult_t r, const char *what) {
    if (r != 0) throw Error(XM_ERR_COMM, std::string("RCCL ") + what + " failed: " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"));
}
}  // namespace

Comm &global_comm() { return g_comm; }

// XM_COMM_TRACE=<prefix>: every collective / marker is appended to <prefix>.<rank> (debugging aid for rank divergence)
static FILE *g_trace = nullptr;
static void trace_open() {
    static bool tried = false;
    if (tried) return;
    tried = true;
    const char *e = std::getenv("XM_COMM_TRACE");
    if (e && *e) g_trace = std::fopen((std::string(e) + "." + std::to_string(g_comm.rank)).c_str(), "w");
}
void Comm::note(const char *what, double a, double b) {
    trace_open();
    if (g_trace) { std::fprintf(g_trace, "%s %.17g %.17g\n", what, a, b); std::fflush(g_trace); }
}

void comm_unique_id(unsigned char id[128]) {
    load_rccl(nullptr);
    ncclUniqueId u;
    check(g_rccl.GetUniqueId(&u), "ncclGetUniqueId");
    std::memcpy(id, u.internal, 128);
}

void comm_init(int rank, int world, int device, const unsigned char id[128], const char *lib_path) {
    if (world < 1 || rank < 0 || rank >= world) throw Error(XM_ERR_ARG, "bad rank/world");
    XM_HIP_CHECK(hipSetDevice(device));
    if (g_rccl.comm) comm_finalize();
    if (world > 1 && id == nullptr) throw Error(XM_ERR_ARG, "xm_comm_init: world > 1 needs the 128-byte unique id of rank 0");
    if (world > 1 || id != nullptr) {
        load_rccl(lib_path);
        ncclUniqueId u;
        std::memcpy(u.internal, id, 128);
        check(g_rccl.CommInitRank(&g_rccl.comm, world, u, rank), "ncclCommInitRank");
    }
    g_comm.rank = rank;
    g_comm.world = world;
    const char *f = std::getenv("XM_FORCE_COMM");
    g_comm.forced = (f && *f == '1' && g_rccl.comm != nullptr);
}

void comm_init_shm(int rank, int world, int device, const char *name, size_t bytes) {
    if (world < 1 || rank < 0 || rank >= world || !name) throw Error(XM_ERR_ARG, "bad rank/world/name");
    XM_HIP_CHECK(hipSetDevice(device));
    comm_finalize();
    const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) throw Error(XM_ERR_COMM, "shm_open failed");
    const size_t total = sizeof(ShmHeader) + 64 + bytes;
    if (ftruncate(fd, (off_t)total) != 0) { close(fd); throw Error(XM_ERR_COMM, "ftruncate failed"); }
    void *p = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) throw Error(XM_ERR_COMM, "mmap failed");
    g_shm.base = p; g_shm.bytes = total; g_shm.barriers = 0; g_shm.name = name;
    g_shm.err.store(0);
    const char *as = std::getenv("XM_SHM_ASYNC");
    g_shm.async = (as && *as == '1');
    if (g_shm.async) XM_HIP_CHECK(hipHostMalloc((void **)&g_shm.stage, bytes, hipHostMallocDefault));
    g_comm.rank = rank; g_comm.world = world; g_comm.forced = true;
    shm_barrier(world);   // everybody has the segment mapped (a fresh segment is zero-filled)
}

void comm_finalize() {
    if (g_shm.base) {
        munmap(g_shm.base, g_shm.bytes);
        if (g_comm.rank == 0) shm_unlink(g_shm.name.c_str());
        if (g_shm.stage) (void)hipHostFree(g_shm.stage);
        g_shm.base = nullptr; g_shm.bytes = 0; g_shm.barriers = 0; g_shm.name.clear(); g_shm.async = false; g_shm.stage = nullptr;
        g_shm.err.store(0);
    }
    if (g_rccl.comm) { g_rccl.CommDestroy(g_rccl.comm); g_rccl.comm = nullptr; }
    g_comm.rank = 0;
    g_comm.world = 1;
    g_comm.forced = false;
}

void Comm::allgather(double *buf, size_t count, hipStream_t st) {
    note("allgather", (double)count, 0.0);
    if (g_shm.active()) {
        const size_t cap = (g_shm.bytes - sizeof(ShmHeader) - 64) / sizeof(double);
        if (count * (size_t)world > cap) throw Error(XM_ERR_COMM, "shared-memory communicator: message too large");
        if (g_shm.err.load()) throw Error(XM_ERR_COMM, "shared-memory communicator: a stream-ordered exchange timed out (ranks issued different collectives?)");
        if (g_shm.async) {
            XM_HIP_CHECK(hipMemcpyAsync(g_shm.stage + (size_t)rank * count, buf + (size_t)rank * count, count * sizeof(double), hipMemcpyDeviceToHost, st));
            XM_HIP_CHECK(hipLaunchHostFunc(st, shm_exchange_cb, new ShmOp{count}));
            XM_HIP_CHECK(hipMemcpyAsync(buf, g_shm.stage, count * (size_t)world * sizeof(double), hipMemcpyHostToDevice, st));
            return;
        }
        double *data = reinterpret_cast<double *>(static_cast<char *>(g_shm.base) + sizeof(ShmHeader) + 64);
        XM_HIP_CHECK(hipMemcpyAsync(data + (size_t)rank * count, buf + (size_t)rank * count, count * sizeof(double), hipMemcpyDeviceToHost, st));
        XM_HIP_CHECK(hipStreamSynchronize(st));
        shm_barrier(world);                       // every chunk is in the segment
        XM_HIP_CHECK(hipMemcpyAsync(buf, data, count * (size_t)world * sizeof(double), hipMemcpyHostToDevice, st));
        XM_HIP_CHECK(hipStreamSynchronize(st));
        shm_barrier(world);                       // everybody has read it; the segment may be overwritten
        return;
    }
    if (!g_rccl.comm) return;
    check(g_rccl.AllGather(buf + (size_t)rank * count, buf, count, kNcclFloat64, g_rccl.comm, st), "ncclAllGather");
}

}  // namespace xm
