// xm_comm.hip — row-partition communicators (the reference is single-GPU: memory.h:54 gpu_id == 0, no NCCL anywhere).
//
// Cameras are split in contiguous ranges; per Q*W product outside the truncated CG one in-place all-gather of the product input W
// (3*nloc*OP doubles per rank) and, per tCG iteration, ONE exchange of [rows of the image of Hp | a few hundred partial sums] —
// gathered rather than all-reduced so that every rank adds them in the same fixed order and takes bit-identical branch decisions.
// Three transports behind one interface (xm_solver.h: Comm), each owned by the Context that uses it:
//   RcclComm  one process per GPU; RCCL is dlopen()ed at run time (torch bundles its own librccl.so with the same soname; whichever
//             the process already loaded is reused, otherwise /opt/rocm/lib/librccl.so).  No link-time dependency.
//   ShmComm   TEST transport: the same calls staged through a POSIX shared-memory segment, so several ranks can share ONE GPU.
//   PeerComm  direct peer writes (this file, bottom): single process, one host thread per GPU, no library call per collective.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <vector>

#include <cstdlib>
#include <cstring>

#include "xm_solver.h"

namespace xm {

using clk = std::chrono::steady_clock;
static double since(clk::time_point t0) { return std::chrono::duration<double>(clk::now() - t0).count(); }

// ------------------------------------------------------------------------------------------------------------------
// settings: the caller's xm_tuning_t, else the defaults.  Resolved once per context; no kernel or layout is selected through the
// environment (the only variable read here is XM_WATCHDOG_S, for callers of the file surface, which has no tuning argument).
// ------------------------------------------------------------------------------------------------------------------
Settings Settings::resolve(const xm_tuning_t *t) {
    Settings s;
    xm_tuning_t z;
    std::memset(&z, 0, sizeof(z));
    if (t) z = *t;
    auto tri = [](int field) { return field > 0 ? 1 : (field < 0 ? -1 : 0); };
    s.sym = tri(z.sym);
    s.sym_min_rows = z.sym_min_rows > 0 ? z.sym_min_rows : 0;   // 0: the measured default of the path that asks (Settings::sym_rows)
    s.sell = tri(z.sell);
    s.sell_slabs = z.sell_slabs > 0 ? z.sell_slabs : 4;
    s.sell_lmax = z.sell_lmax > 0 ? z.sell_lmax : 64;
    if (z.sell_gather < 0 || z.sell_gather > 2) throw Error(XM_ERR_ARG, "xm_tuning_t.sell_gather must be 0 (default), 1 (a record per lane) or 2 (LDS-transposed)");
    s.sell_gather = (z.sell_gather == 1) ? 0 : 1;   // the kernels' mode: 0 a record per lane | 1 LDS-transposed (default)
    if (z.sell_codec < 0 || z.sell_codec > 2) throw Error(XM_ERR_ARG, "xm_tuning_t.sell_codec must be 0, 1 or 2");
    s.sell_codec = z.sell_codec;
    s.sell_wpad = tri(z.sell_wpad);
    s.overlap = (z.overlap < 0) ? -1 : 0;
    s.overlap_min_mb = z.overlap_min_mb > 0 ? (double)z.overlap_min_mb : (z.overlap_min_mb < 0 ? 0.0 : 64.0);
    s.cert_dense_rows = z.cert_dense_rows > 0 ? z.cert_dense_rows : 384;
    s.lanczos_mmax = z.lanczos_mmax > 0 ? z.lanczos_mmax : 400;
    s.lanczos_restarts = z.lanczos_restarts > 0 ? z.lanczos_restarts : 12;
    if (z.watchdog_s > 0) s.watchdog_s = (double)z.watchdog_s;
    else if (const char *e = std::getenv("XM_WATCHDOG_S")) { const double v = std::atof(e); if (v > 0) s.watchdog_s = v; }
    s.balance = z.balance;
    if (z.exchange < 0 || z.exchange > 3) throw Error(XM_ERR_ARG, "xm_tuning_t.exchange must be 0..3");
    s.exchange = z.exchange;
    s.split_k = z.split_k;
    s.exchange_lite = z.exchange_fence ? 0 : 1;
    s.schur_host_assembly = z.schur_host_assembly != 0;
    s.schur_trace = z.schur_trace != 0;
    if (z.schur_solver < 0 || z.schur_solver > 2) throw Error(XM_ERR_ARG, "xm_tuning_t.schur_solver must be 0, 1 or 2");
    s.schur_solver = z.schur_solver;
    if (z.schur_dense_max > 0) s.schur_dense_max = z.schur_dense_max;
    s.schur_pcg_first = z.schur_pcg_first > 0 ? z.schur_pcg_first : 0;
    if (z.schur_pcg_hess_digits != 0 && (z.schur_pcg_hess_digits < 6 || z.schur_pcg_hess_digits > 13)) throw Error(XM_ERR_ARG, "xm_tuning_t.schur_pcg_hess_digits must be 0 or 6..13");
    s.schur_pcg_hess_digits = z.schur_pcg_hess_digits;
    s.debug_drop_finalize = z.debug_drop_finalize > 0 ? z.debug_drop_finalize : -1;
    s.debug_peer_mute = z.debug_peer_mute;
    return s;
}

// ------------------------------------------------------------------------------------------------------------------
// base class
// ------------------------------------------------------------------------------------------------------------------
Comm::~Comm() { if (trace_) std::fclose(trace_); }

// XM_COMM_TRACE=<prefix>: every collective / marker is appended to <prefix>.<rank> (debugging aid for rank divergence)
void Comm::note(const char *what, double a, double b) {
    if (!trace_tried_) {
        trace_tried_ = true;
        const char *e = std::getenv("XM_COMM_TRACE");
        if (e && *e) trace_ = std::fopen((std::string(e) + "." + std::to_string(rank)).c_str(), "w");
    }
    if (trace_) { std::fprintf(trace_, "%s %.17g %.17g\n", what, a, b); std::fflush(trace_); }
}

namespace {
// ------------------------------------------------------------------------------------------------------------------
// RCCL
// ------------------------------------------------------------------------------------------------------------------
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
constexpr int kNcclFloat64 = 8;  // rccl.h: ncclFloat64 = 8

struct Rccl {   // the library (process-wide: a dlopen handle and its entry points, no communicator state)
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::mutex g_mu;
std::shared_ptr<Comm> g_default;   // installed by xm_comm_init / xm_comm_init_shm for the multi-PROCESS launch (one rank per process)

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

struct RcclComm : Comm {
    ncclComm_t comm = nullptr;
    ~RcclComm() override { if (comm) g_rccl.CommDestroy(comm); }
    int kind() const override { return 1; }
    void allgather(double *buf, size_t count, hipStream_t st) override {
        note("allgather", (double)count, 0.0);
        if (!comm) return;
        check(g_rccl.AllGather(buf + (size_t)rank * count, buf, count, kNcclFloat64, comm, st), "ncclAllGather");
    }
};

// ------------------------------------------------------------------------------------------------------------------
// Test transport: the same collectives through a POSIX shared-memory segment (device -> host -> shm -> host -> device), so
// that several ranks can share ONE GPU on a 1-GPU box and the whole row-partitioned solver (partition, padding cameras,
// identical collective counts on every rank) can be exercised there.
// Invariant: every collective of a ShmComm goes through ONE stream (the owning Context's st_).  The stream-ordered variant keeps a
// single pinned staging buffer and a plain barrier counter that only the stream's host functions touch; they execute in stream
// order, so there is never more than one in flight.  finalize() drains the device before the buffer is freed.
// ------------------------------------------------------------------------------------------------------------------
struct ShmHeader {
    std::atomic<unsigned long long> arrive;   // monotonically increasing barrier counter
};
struct ShmComm;
struct ShmOp { ShmComm *c; size_t count; };
struct ShmComm : Comm {
    void *base = nullptr;
    size_t bytes = 0;
    std::atomic<unsigned long long> barriers{0};   // barriers completed by this rank (host thread or stream callbacks, never both at once)
    std::string name;
    bool async = false;                        // XM_SHM_ASYNC=1: stream-ordered exchange (host functions on the stream)
    double *stage = nullptr;                   // pinned staging buffer of the stream-ordered variant
    std::atomic<int> err{0};                   // set by a host function that timed out (it cannot throw)
    double limit = 120.0;
    hipStream_t bound = nullptr;               // the one stream this communicator is used on (asserted)
    bool bound_set = false;
    int kind() const override { return 2; }
    ~ShmComm() override {
        if (base) {
            (void)hipDeviceSynchronize();      // no host function may still be pending when the staging buffer goes away
            munmap(base, bytes);
            if (rank == 0) shm_unlink(name.c_str());
            if (stage) (void)hipHostFree(stage);
        }
    }
    bool barrier_nothrow() {
        ShmHeader *h = static_cast<ShmHeader *>(base);
        h->arrive.fetch_add(1, std::memory_order_acq_rel);
        const unsigned long long target = (barriers.fetch_add(1) + 1) * (unsigned long long)world;
        const auto t0 = clk::now();
        while (h->arrive.load(std::memory_order_acquire) < target)
            if (since(t0) > limit) return false;
        return true;
    }
    void barrier() {
        if (!barrier_nothrow()) {
            ShmHeader *h = static_cast<ShmHeader *>(base);
            throw Error(XM_ERR_COMM, "shared-memory communicator: rank " + std::to_string(rank) + " waited too long in barrier #" +
                                         std::to_string(barriers.load()) + " (arrived " + std::to_string(h->arrive.load()) +
                                         "): ranks issued different collectives?");
        }
    }
    double *data() const { return reinterpret_cast<double *>(static_cast<char *>(base) + sizeof(ShmHeader) + 64); }
    // Stream-ordered variant (XM_SHM_ASYNC=1): the exchange runs inside a host function that the STREAM executes between the
    // device-to-host and host-to-device copies, so the calling thread never blocks — exactly like an RCCL collective.  The solver's
    // enqueue-ahead logic is then exercised for real: a rank that enqueues a different number of collectives than its peers
    // leaves a barrier unmatched (time-out -> error flag -> XM_ERR_COMM at the next collective).
    static void exchange_cb(void *p) {
        ShmOp *op = static_cast<ShmOp *>(p);
        ShmComm *c = op->c;
        const size_t count = op->count;
        delete op;
        if (c->err.load()) return;
        double *d = c->data();
        std::memcpy(d + (size_t)c->rank * count, c->stage + (size_t)c->rank * count, count * sizeof(double));
        if (!c->barrier_nothrow()) { c->err.store(1); return; }
        std::memcpy(c->stage, d, count * (size_t)c->world * sizeof(double));
        if (!c->barrier_nothrow()) c->err.store(1);
    }
    void allgather(double *buf, size_t count, hipStream_t st) override {
        note("allgather", (double)count, 0.0);
        if (!bound_set) { bound = st; bound_set = true; }
        if (st != bound) {   // a new context took over the communicator: nothing of the previous stream may still be pending
            XM_HIP_CHECK(hipDeviceSynchronize());
            bound = st;
        }
        const size_t cap = (bytes - sizeof(ShmHeader) - 64) / sizeof(double);
        if (count * (size_t)world > cap) throw Error(XM_ERR_COMM, "shared-memory communicator: message too large");
        if (err.load()) throw Error(XM_ERR_COMM, "shared-memory communicator: a stream-ordered exchange timed out (ranks issued different collectives?)");
        if (async) {
            XM_HIP_CHECK(hipMemcpyAsync(stage + (size_t)rank * count, buf + (size_t)rank * count, count * sizeof(double), hipMemcpyDeviceToHost, st));
            XM_HIP_CHECK(hipLaunchHostFunc(st, exchange_cb, new ShmOp{this, count}));
            XM_HIP_CHECK(hipMemcpyAsync(buf, stage, count * (size_t)world * sizeof(double), hipMemcpyHostToDevice, st));
            return;
        }
        double *d = data();
        XM_HIP_CHECK(hipMemcpyAsync(d + (size_t)rank * count, buf + (size_t)rank * count, count * sizeof(double), hipMemcpyDeviceToHost, st));
        XM_HIP_CHECK(hipStreamSynchronize(st));
        barrier();                                // every chunk is in the segment
        XM_HIP_CHECK(hipMemcpyAsync(buf, d, count * (size_t)world * sizeof(double), hipMemcpyHostToDevice, st));
        XM_HIP_CHECK(hipStreamSynchronize(st));
        barrier();                                // everybody has read it; the segment may be overwritten
    }
    void host_barrier() override { if (!async) barrier(); }
};
}  // namespace

std::shared_ptr<Comm> default_comm() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_default) return g_default;
    return std::make_shared<Comm>();
}

void comm_unique_id(unsigned char id[128]) {
    std::lock_guard<std::mutex> lk(g_mu);
    load_rccl(nullptr);
    ncclUniqueId u;
    check(g_rccl.GetUniqueId(&u), "ncclGetUniqueId");
    std::memcpy(id, u.internal, 128);
}

static std::shared_ptr<Comm> ipc_upgrade(int rank, int world, int device, const unsigned char id[128], const std::shared_ptr<Comm> &keep);
void comm_init(int rank, int world, int device, const unsigned char id[128], const char *lib_path) {
    if (world < 1 || rank < 0 || rank >= world) throw Error(XM_ERR_ARG, "bad rank/world");
    XM_HIP_CHECK(hipSetDevice(device));
    comm_finalize();
    if (world > 1 && id == nullptr) throw Error(XM_ERR_ARG, "xm_comm_init: world > 1 needs the 128-byte unique id of rank 0");
    std::lock_guard<std::mutex> lk(g_mu);
    auto c = std::make_shared<RcclComm>();
    if (world > 1 || id != nullptr) {
        load_rccl(lib_path);
        ncclUniqueId u;
        std::memcpy(u.internal, id, 128);
        check(g_rccl.CommInitRank(&c->comm, world, u, rank), "ncclCommInitRank");
    }
    c->rank = rank;
    c->world = world;
    const char *f = std::getenv("XM_FORCE_COMM");
    c->forced = (f && *f == '1' && c->comm != nullptr);
    g_default = c;
    // Ranks of ONE node: the direct peer exchange over IPC-mapped buffers replaces the library all-gather (2-4 x the partitioned
    // product at Venice size) when every rank can map every peer and the transport passes its self-test; otherwise -- ranks on several
    // nodes, no peer access, XM_COMM_PEER=0 -- all ranks keep the RCCL communicator (the decision is collective).
    if (auto pc = ipc_upgrade(rank, world, device, id, c)) g_default = pc;
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
    auto c = std::make_shared<ShmComm>();
    c->base = p; c->bytes = total; c->name = name;
    { const char *e = std::getenv("XM_SHM_TIMEOUT"); c->limit = e ? std::atof(e) : 120.0; }
    const char *as = std::getenv("XM_SHM_ASYNC");
    c->async = (as && *as == '1');
    if (c->async) XM_HIP_CHECK(hipHostMalloc((void **)&c->stage, bytes, hipHostMallocDefault));
    c->rank = rank; c->world = world; c->forced = true;
    c->barrier();   // everybody has the segment mapped (a fresh segment is zero-filled)
    std::lock_guard<std::mutex> lk(g_mu);
    g_default = c;
}

void comm_finalize() {
    std::shared_ptr<Comm> old;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        old.swap(g_default);
    }
    old.reset();   // contexts that still hold it keep it alive; the process default is a single-rank Comm again
}

// ==================================================================================================================
// PeerComm: direct peer writes.
//
// Every rank owns an ARENA in fine-grained memory of its own device (hipExtMallocWithFlags(hipDeviceMallocFinegrained): remote
// stores and system-scope atomics are coherent without cache maintenance on either side) that every rank of the group can address:
//     flags   2 slots x kMaxPeers epoch words for the generic all-gather + 2 x kMaxPeers for the fused tCG exchange
//     ticket  local arrival counters (which workgroup of a pushing launch finished last)
//     stage   2 slots x capacity doubles: landing zone of the generic all-gather
// all-gather(buf, count) = push kernel (each workgroup copies a slice of the rank's chunk into slot s of EVERY peer's stage,
// system-scope fence, local ticket; the last workgroup stores the epoch into flags[s][me] of every peer) + wait kernel (bounded
// spin on the world-1 epoch words of slot s, then copies the peers' chunks from the own stage into buf).  Slots alternate; a rank
// can be at most one collective ahead of its slowest peer (it cannot pass a wait the peer has not pushed for), so two suffice.
// No host rendezvous, no library: 2 short launches per collective.  The truncated CG does not even pay those: cg_step_kernel pushes
// the rank's [rows of B | partial sums] chunk into the peers' exchange buffers and waits for theirs inside the launch it needs
// anyway (xm_kernels.hip, PeerXchg).
// Every device-side wait is bounded (spin_ticks of the 100 MHz wall clock): a dead or diverged peer turns into an error word that the
// host reports as XM_ERR_COMM instead of a hung GPU.
// ==================================================================================================================
struct PeerGroup {   // who the ranks of a peer communicator are and how they find each other's memory
    int world = 1;
    int device[kMaxPeers] = {};
    double spin_seconds = 20.0;
    virtual ~PeerGroup() {}
    virtual int kind() const = 0;                                             // 3 threads of one process | 4 processes (IPC handles)
    virtual void barrier(unsigned long long &mine, const char *what) = 0;     // reusable host barrier of the group (counting)
    virtual void abort() = 0;
    virtual bool aborted() const = 0;
    // collective: rank `me` publishes its allocation `p` under `key` (0 arena, 1 tCG exchange buffer) together with `meta`; on
    // return addr[r] / metas[r] hold every rank's, addr[r] being an address THIS process can store through
    virtual void share(int me, int key, void *p, size_t meta, void *addr[kMaxPeers], size_t metas[kMaxPeers], unsigned long long &hb) = 0;
    // before a shared allocation is freed: drop this process's view of the peers' allocations under `key`; collective (ends with a
    // barrier: the owner may free afterwards) unless `teardown`
    virtual void unshare(int me, int key, unsigned long long &hb, bool teardown) = 0;
    virtual int ranks_on_device_of(int me) const = 0;   // ranks whose kernels run on rank me's GPU (itself included)
};

namespace {
struct LocalPeerGroup : PeerGroup {   // single process, one host thread per rank: addresses are plain pointers
    void *tab[2][kMaxPeers] = {};
    size_t meta[2][kMaxPeers] = {};
    std::atomic<int> aborted_{0};
    std::atomic<unsigned long long> arrive{0};
    int kind() const override { return 3; }
    void abort() override { aborted_.store(1); }
    bool aborted() const override { return aborted_.load() != 0; }
    void barrier(unsigned long long &mine, const char *what) override {
        arrive.fetch_add(1, std::memory_order_acq_rel);
        const unsigned long long target = (++mine) * (unsigned long long)world;
        const auto t0 = clk::now();
        while (arrive.load(std::memory_order_acquire) < target) {
            if (aborted_.load()) throw Error(XM_ERR_COMM, std::string("peer group aborted while waiting in ") + what);
            if (since(t0) > 120.0) throw Error(XM_ERR_COMM, std::string("peer group: a rank did not reach ") + what);
            std::this_thread::yield();
        }
    }
    void share(int me, int key, void *p, size_t m, void *addr[kMaxPeers], size_t metas[kMaxPeers], unsigned long long &hb) override {
        tab[key][me] = p; meta[key][me] = m;
        barrier(hb, "publication of a shared buffer");
        for (int r = 0; r < world; ++r) { addr[r] = tab[key][r]; metas[r] = meta[key][r]; }
    }
    void unshare(int, int, unsigned long long &, bool) override {}
    int ranks_on_device_of(int me) const override {
        int c = 0;
        for (int r = 0; r < world; ++r) c += (device[r] == device[me]);
        return c;
    }
};

// One process per GPU: the directory lives in a POSIX shared-memory segment, allocations travel as hipIpcMemHandle_t (dmabuf export;
// HSA_ENABLE_IPC_MODE_LEGACY=0 on this pool) and are mapped into every peer process.  Same kernels, same protocol as the in-process
// group -- the launch `python -m torch.distributed.run` gets the fused exchange instead of a library collective per tCG iteration.
struct IpcShared {
    std::atomic<unsigned long long> arrive;
    std::atomic<int> aborted;
    std::atomic<int> failed;      // a rank could not export or map: all ranks give the transport up together
    struct Slot { hipIpcMemHandle_t h; unsigned long long meta; } slot[2][kMaxPeers];
    unsigned long long devid[kMaxPeers];   // identity of every rank's GPU (hash of its PCI bus id): which ranks share a device
    std::atomic<unsigned long long> stamp; // wall-clock nanoseconds at which rank 0 created THIS segment (0: not initialised yet)
};
struct IpcPeerGroup : PeerGroup {
    IpcShared *sh = nullptr;
    std::string name;
    int me = 0;
    double limit = 120.0;
    void *mapped[2][kMaxPeers] = {};
    unsigned long long seg_ino = 0, seg_dev = 0;   // identity of the mapped segment (ranks > 0: to notice that rank 0 has replaced the name since)
    int kind() const override { return 4; }
    ~IpcPeerGroup() override {
        for (int k = 0; k < 2; ++k)
            for (int r = 0; r < kMaxPeers; ++r)
                if (mapped[k][r]) (void)hipIpcCloseMemHandle(mapped[k][r]);
        if (sh) munmap(sh, sizeof(IpcShared));
    }
    void abort() override { if (sh) sh->aborted.store(1); }
    bool aborted() const override { return sh && sh->aborted.load() != 0; }
    bool barrier_nothrow(unsigned long long &mine) {
        sh->arrive.fetch_add(1, std::memory_order_acq_rel);
        const unsigned long long target = (++mine) * (unsigned long long)world;
        const auto t0 = clk::now();
        while (sh->arrive.load(std::memory_order_acquire) < target) {
            if (sh->aborted.load() || since(t0) > limit) return false;
            std::this_thread::yield();
        }
        return sh->aborted.load() == 0;   // also for the rank whose arrival completed the count: an abort is everybody's
    }
    // The FIRST barrier of a group.  A rank > 0 that opened the name before rank 0 had replaced a segment left by an earlier job of the same
    // name (its stamp still fresh: a quick re-run after a crash) sits in the OLD segment, where rank 0 never arrives.  While it waits it
    // looks the name up again: another inode under it means exactly that.  1 met | 0 failed / timed out | -1 stale segment mapped
    int rendezvous(unsigned long long &mine) {
        sh->arrive.fetch_add(1, std::memory_order_acq_rel);
        const unsigned long long target = (++mine) * (unsigned long long)world;
        const auto t0 = clk::now();
        auto t_look = t0;
        while (sh->arrive.load(std::memory_order_acquire) < target) {
            if (sh->aborted.load() || since(t0) > limit) return 0;
            if (me != 0 && since(t_look) > 0.02) {
                t_look = clk::now();
                const int fd = shm_open(name.c_str(), O_RDWR, 0600);
                if (fd >= 0) {
                    struct stat sb;
                    const bool other = fstat(fd, &sb) == 0 && ((unsigned long long)sb.st_ino != seg_ino || (unsigned long long)sb.st_dev != seg_dev);
                    close(fd);
                    if (other) return -1;
                }
            }
            std::this_thread::yield();
        }
        return sh->aborted.load() == 0 ? 1 : 0;
    }
    void barrier(unsigned long long &mine, const char *what) override {
        if (!barrier_nothrow(mine))
            throw Error(XM_ERR_COMM, std::string(sh->aborted.load() ? "peer group aborted while waiting in " : "peer group: a rank did not reach ") + what);
    }
    void share(int r0, int key, void *p, size_t m, void *addr[kMaxPeers], size_t metas[kMaxPeers], unsigned long long &hb) override {
        IpcShared::Slot &mine = sh->slot[key][r0];
        std::memset(&mine.h, 0, sizeof(mine.h));
        if (hipIpcGetMemHandle(&mine.h, p) != hipSuccess) { (void)hipGetLastError(); sh->failed.store(1); }
        mine.meta = m;
        barrier(hb, "publication of a shared buffer");
        for (int r = 0; r < world; ++r) {
            metas[r] = (size_t)sh->slot[key][r].meta;
            if (r == r0) { addr[r] = p; continue; }
            addr[r] = nullptr;
            if (sh->failed.load()) continue;
            void *q = nullptr;
            if (hipIpcOpenMemHandle(&q, sh->slot[key][r].h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); sh->failed.store(1); continue; }
            mapped[key][r] = q;
            addr[r] = q;
        }
        barrier(hb, "mapping of the peers' buffers");
        if (sh->failed.load()) throw Error(XM_ERR_COMM, "peer communicator: a rank could not export or map a buffer through hipIpcMemHandle");
    }
    void unshare(int, int key, unsigned long long &hb, bool teardown) override {
        for (int r = 0; r < world; ++r)
            if (mapped[key][r]) { (void)hipIpcCloseMemHandle(mapped[key][r]); mapped[key][r] = nullptr; }
        if (!teardown) barrier(hb, "release of the peers' buffers");
    }
    int ranks_on_device_of(int r0) const override {   // valid after the rendezvous barrier (every rank has published its devid)
        int c = 0;
        for (int r = 0; r < world; ++r) c += (sh->devid[r] == sh->devid[r0]);
        return std::max(c, 1);
    }
};
}  // namespace

namespace {
constexpr size_t kFlagWords = 4 * kMaxPeers;                        // [gen slot0 | gen slot1 | tcg par0 | tcg par1]
constexpr size_t kArenaHead = (kFlagWords + 8) * sizeof(unsigned long long) + 64;   // flags, 4 tickets, error word, pad

struct PeerPtrs {   // by value into the kernels
    double *stage[kMaxPeers];
    unsigned long long *flags[kMaxPeers];
    int world, rank, lite;
};

__device__ __forceinline__ unsigned long long ld_sys(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// chunk -> slot of every peer's stage; last workgroup publishes the epoch
__global__ __launch_bounds__(256) void peer_push_kernel(PeerPtrs pp, const double *__restrict__ chunk, size_t count, size_t slot_off,
                                                         int fslot, unsigned long long epoch, unsigned long long *ticket) {
    __shared__ int last;
    const size_t stride = (size_t)gridDim.x * 256;
    // Payload through write-through (system-scope) stores: each is acknowledged by its destination before s_waitcnt vmcnt(0) lets the
    // wave go on and leaves nothing in this device's L2, so the hand-off needs no release fence (a fence writes back the whole L2 of the
    // XCD; measured on the fused tCG exchange: 6 us per iteration).  pp.lite == 0 keeps the fence form (xm_tuning_t.exchange_fence).
    for (int p = 0; p < pp.world; ++p) {
        if (p == pp.rank) continue;
        double *dst = pp.stage[p] + slot_off + (size_t)pp.rank * count;
        if (pp.lite) for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) __hip_atomic_store(dst + i, chunk[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) dst[i] = chunk[i];
    }
    if (!pp.lite) __threadfence_system();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = pp.lite ? __hip_atomic_fetch_add(ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                             : __hip_atomic_fetch_add(ticket, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = (t + 1 == gridDim.x);
        if (last) __hip_atomic_store(ticket, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // one pushing launch at a time (stream order)
    }
    __syncthreads();
    if (last && threadIdx.x < pp.world && (int)threadIdx.x != pp.rank) {
        if (pp.lite) {
            __hip_atomic_store(pp.flags[threadIdx.x] + fslot * kMaxPeers + pp.rank, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
            __threadfence_system();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(pp.flags[threadIdx.x] + fslot * kMaxPeers + pp.rank, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// wait for the world-1 epochs of the slot, then stage -> buf for the peers' chunks
// plain != 0 (self-test only): the landed chunks are read with ordinary loads behind the acquire fence -- the way cg_step_kernel reads
// its exchange buffer -- instead of system-scope loads, so that a platform on which that is not enough (stale lines of the rank's own
// fine-grained buffer in its L2) fails the self-test instead of a solve
__global__ __launch_bounds__(256) void peer_wait_kernel(PeerPtrs pp, double *__restrict__ buf, size_t count, size_t slot_off, int fslot,
                                                         unsigned long long epoch, long long spin_ticks, unsigned long long *err, int plain) {
    __shared__ int ok;
    if (threadIdx.x == 0) {
        int good = 1;
        const long long t0 = wall_clock64();
        const unsigned long long *f = pp.flags[pp.rank] + fslot * kMaxPeers;
        for (int p = 0; p < pp.world && good; ++p) {
            if (p == pp.rank) continue;
            while (ld_sys(f + p) < epoch) {
                if (wall_clock64() - t0 > spin_ticks) { good = 0; break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        if (!good) __hip_atomic_store(err, epoch | (1ull << 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        ok = good;
        __atomic_thread_fence(__ATOMIC_ACQUIRE);   // system scope: the peers' payload stores precede their flag
    }
    __syncthreads();
    if (!ok) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    const double *src = pp.stage[pp.rank] + slot_off;
    const size_t total = count * (size_t)pp.world, stride = (size_t)gridDim.x * 256;
    const size_t own0 = (size_t)pp.rank * count, own1 = own0 + count;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride)
        if (i < own0 || i >= own1) buf[i] = plain ? src[i] : __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct PeerComm : Comm {
    std::shared_ptr<PeerGroup> g;
    char *arena = nullptr;
    size_t cap = 0;                    // doubles per stage slot
    unsigned long long seq = 0;        // collectives issued (identical on every rank)
    unsigned long long hb;             // host barriers of the group this rank has passed
    unsigned long long *herr = nullptr;   // host-mapped error word
    unsigned long long *herr_dev = nullptr;
    double *xbuf = nullptr;
    size_t xcap = 0;                   // doubles of the tCG exchange buffer (grow-only)
    void *parena[kMaxPeers] = {};      // every rank's arena / exchange buffer as THIS process addresses it
    void *pxbuf[kMaxPeers] = {};
    int plain_reads = 0;               // self-test: see peer_wait_kernel
    int lite = 1;                      // write-through payload stores instead of a release fence (set_exchange_fence: fence form)
    void set_exchange_fence(bool on) override { lite = on ? 0 : 1; }
    std::shared_ptr<Comm> keep;        // a library communicator created beside this one (xm_comm_init): destroyed with it
    int kind() const override { return g->kind(); }
    bool peer() const override { return true; }
    int ranks_on_my_device() const override { return g->ranks_on_device_of(rank); }
    bool device_waits() const override { return g->ranks_on_device_of(rank) <= 4; }
    unsigned long long *flags_of(int r) const { return reinterpret_cast<unsigned long long *>(parena[r]); }
    unsigned long long *tickets() const { return reinterpret_cast<unsigned long long *>(arena) + kFlagWords; }
    double *stage_of(int r) const { return reinterpret_cast<double *>(static_cast<char *>(parena[r]) + kArenaHead); }
    long long spin_ticks() const { return (long long)(g->spin_seconds * 1e8); }

    PeerComm(const std::shared_ptr<PeerGroup> &grp, int r, unsigned long long hb0 = 0) : g(grp), hb(hb0) {
        rank = r; world = grp->world; forced = true;
        XM_HIP_CHECK(hipHostMalloc((void **)&herr, 64, hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(herr, 0, 64);
        XM_HIP_CHECK(hipHostGetDevicePointer((void **)&herr_dev, herr, 0));
        if (world > 1 && g->kind() == 3) {   // peer access between distinct devices of this process (no-op for virtual devices; IPC mappings enable it lazily)
            for (int p = 0; p < world; ++p) {
                if (g->device[p] == g->device[rank]) continue;
                int can = 0;
                XM_HIP_CHECK(hipDeviceCanAccessPeer(&can, g->device[rank], g->device[p]));
                if (!can) throw Error(XM_ERR_COMM, "peer communicator: device " + std::to_string(g->device[rank]) + " cannot address device " + std::to_string(g->device[p]));
                const hipError_t e = hipDeviceEnablePeerAccess(g->device[p], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) XM_HIP_CHECK(e);
                (void)hipGetLastError();
            }
        }
        // Between processes the buffers are sized generously ONCE (2 x 32 MB of landing zone here, 64 MB of tCG exchange buffer in
        // xchg_setup: enough for 100 k cameras at the top rank): re-exporting a re-allocated buffer works (collective path below) but
        // hipIpcOpenMemHandle has been seen to refuse the handle of an allocation that took the address of one this process had
        // mapped before (bench flow after other GPU processes had run: "could not export or map"), so growth is kept for the rare case.
        alloc_arena(g->kind() == 4 ? ((size_t)1 << 22) : ((size_t)1 << 16));
    }
    ~PeerComm() override {
        (void)hipDeviceSynchronize();
        g->unshare(rank, 0, hb, true);
        g->unshare(rank, 1, hb, true);
        if (arena) (void)hipFree(arena);
        if (xbuf) (void)hipFree(xbuf);
        if (herr) (void)hipHostFree(herr);
    }
    // ranks on one device (virtual devices of a single process) can live with ordinary memory; anything that crosses a device or a
    // process boundary needs the fine-grained kind, and its absence is an error (xm_comm_init then keeps RCCL), never a silent downgrade
    bool strict_fine() const {
        if (g->kind() == 4) return true;
        for (int p = 0; p < world; ++p) if (g->device[p] != g->device[rank]) return true;
        return false;
    }
    void *fine_alloc(size_t bytes) const {
        void *p = nullptr;
        hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            if (strict_fine()) throw Error(XM_ERR_COMM, std::string("peer communicator: no fine-grained device memory (") + hipGetErrorString(e) + ")");
            XM_HIP_CHECK(hipMalloc(&p, bytes));
        }
        XM_HIP_CHECK(hipMemset(p, 0, bytes));
        XM_HIP_CHECK(hipDeviceSynchronize());
        return p;
    }
    // collective: every rank (re)allocates its arena with `doubles` per stage slot; nobody may be inside a collective
    void alloc_arena(size_t doubles) {
        if (arena) {
            XM_HIP_CHECK(hipDeviceSynchronize());
            g->barrier(hb, "arena re-allocation (drain)");   // every rank has drained: no remote store into the old arenas is in flight
            g->unshare(rank, 0, hb, false);
            (void)hipFree(arena);
        }
        cap = doubles;
        arena = static_cast<char *>(fine_alloc(kArenaHead + 2 * cap * sizeof(double)));
        seq = 0;   // fresh flag words everywhere
        size_t caps[kMaxPeers] = {};
        g->share(rank, 0, arena, cap, parena, caps, hb);
        for (int p = 0; p < world; ++p)
            if (caps[p] != cap) throw Error(XM_ERR_COMM, "peer communicator: ranks reserved different staging sizes");
    }
    void reserve(size_t doubles) override {
        if (doubles <= cap) { g->barrier(hb, "reserve"); return; }   // same decision on every rank (same argument)
        alloc_arena(doubles + 1024);
    }
    void host_barrier() override { g->barrier(hb, "barrier"); }
    void check_device_error() override {
        if (g->aborted()) throw Error(XM_ERR_COMM, "peer group aborted (another rank failed)");
        const unsigned long long e = *reinterpret_cast<volatile unsigned long long *>(herr);
        if (e != 0) {
            g->abort();
            throw Error(XM_ERR_COMM, "peer exchange: rank " + std::to_string(rank) + " waited more than " + std::to_string(g->spin_seconds) +
                                         " s for a peer (epoch " + std::to_string(e & ~(1ull << 63)) + "): peer failed, or the ranks diverged");
        }
    }
    PeerPtrs ptrs() const {
        PeerPtrs pp;
        std::memset(&pp, 0, sizeof(pp));
        pp.world = world; pp.rank = rank; pp.lite = lite;
        for (int p = 0; p < world; ++p) { pp.stage[p] = stage_of(p); pp.flags[p] = flags_of(p); }
        return pp;
    }
    void allgather(double *buf, size_t count, hipStream_t st) override {
        note("allgather", (double)count, 0.0);
        check_device_error();
        if (count * (size_t)world > cap) throw Error(XM_ERR_COMM, "peer communicator: message larger than the reserved staging area");
        if (count == 0) return;
        ++seq;
        const int slot = (int)(seq & 1);
        const size_t slot_off = (size_t)slot * cap;
        const PeerPtrs pp = ptrs();
        if (world > 1) {
            const int grid = (int)std::min<size_t>(64, (count + 2047) / 2048);
            hipLaunchKernelGGL(peer_push_kernel, dim3(grid), dim3(256), 0, st, pp, buf + (size_t)rank * count, count, slot_off, slot, seq, tickets() + slot);
            const bool host_sync = !device_waits();
            if (host_sync) {   // many ranks on one GPU: everybody's push has landed before anybody's second launch starts (it then finds its flags set)
                XM_HIP_CHECK(hipStreamSynchronize(st));
                g->barrier(hb, "all-gather (host-synchronised)");
            }
            const int wgrid = (int)std::min<size_t>(128, (count * world + 2047) / 2048);
            hipLaunchKernelGGL(peer_wait_kernel, dim3(wgrid), dim3(256), 0, st, pp, buf, count, slot_off, slot, seq, spin_ticks(), herr_dev, plain_reads);
            check_launch("peer_allgather");
            if (host_sync) {   // and everybody has read this slot before anybody pushes into it again
                XM_HIP_CHECK(hipStreamSynchronize(st));
                g->barrier(hb, "all-gather (host-synchronised, read)");
            }
        }
    }
    // collective (host-synchronised): tCG exchange buffer of at least `doubles` (grow-only; the same argument on every rank) with
    // fresh flag words
    void xchg_setup(size_t doubles, PeerXchg &out) override {
        XM_HIP_CHECK(hipDeviceSynchronize());
        g->barrier(hb, "exchange buffer (drain)");
        const size_t want = std::max<size_t>(doubles, 8);
        if (want > xcap) {
            if (xbuf) { g->unshare(rank, 1, hb, false); (void)hipFree(xbuf); }
            xcap = std::max<size_t>(want + want / 4, g->kind() == 4 ? ((size_t)1 << 23) : 0);
            xbuf = static_cast<double *>(fine_alloc(xcap * sizeof(double)));
            size_t caps[kMaxPeers] = {};
            g->share(rank, 1, xbuf, xcap, pxbuf, caps, hb);
        }
        // the tCG flag words restart from zero with every set-up: epochs are (run << 12 | iteration + 1), compared with >=
        XM_HIP_CHECK(hipMemset(flags_of(rank) + 2 * kMaxPeers, 0, 2 * kMaxPeers * sizeof(unsigned long long)));
        XM_HIP_CHECK(hipDeviceSynchronize());
        g->barrier(hb, "exchange buffer");
        out = PeerXchg();
        out.world = world; out.rank = rank;
        for (int p = 0; p < world; ++p) { out.buf[p] = static_cast<double *>(pxbuf[p]); out.flag[p] = flags_of(p) + 2 * kMaxPeers; }
        out.ticket = tickets() + 2;
        out.err = herr_dev;
        out.spin_ticks = spin_ticks();
    }
};
}  // namespace

std::shared_ptr<PeerGroup> peer_group_create(int world, const int *devices, double spin_seconds) {
    if (world < 1 || world > kMaxPeers) throw Error(XM_ERR_ARG, "n_gpus must be 1.." + std::to_string(kMaxPeers));
    auto g = std::make_shared<LocalPeerGroup>();
    g->world = world;
    for (int r = 0; r < world; ++r) g->device[r] = devices[r];
    if (spin_seconds > 0) g->spin_seconds = spin_seconds;
    return g;
}
std::shared_ptr<Comm> peer_comm_create(const std::shared_ptr<PeerGroup> &g, int rank) { return std::make_shared<PeerComm>(g, rank); }
void peer_group_abort(const std::shared_ptr<PeerGroup> &g) { if (g) g->abort(); }

// ------------------------------------------------------------------------------------------------------------------
// one process per GPU over IPC handles (xm_comm_init_ipc; xm_comm_init tries it before it settles for the library all-gather)
// ------------------------------------------------------------------------------------------------------------------
namespace {
std::shared_ptr<IpcPeerGroup> ipc_group_open(int rank, int world, int device, const char *name, double limit, double spin_seconds) {
    if (world > kMaxPeers) throw Error(XM_ERR_ARG, "direct peer exchange supports up to " + std::to_string(kMaxPeers) + " ranks");
    // A segment a crashed job left under the same name would hand this job non-zero counters.  Rank 0 removes whatever carries the name and
    // creates the segment exclusively (zero-filled) and stamps it with its creation time; the others open WITHOUT O_CREAT and accept only a
    // completely sized segment whose stamp is recent (a stale one is dropped and re-opened until rank 0 has replaced it).
    auto wall_ns = [] { return (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::system_clock::now().time_since_epoch()).count(); };
    void *p = MAP_FAILED;
    unsigned long long ino = 0, dev = 0;
    if (rank == 0) {
        (void)shm_unlink(name);
        const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0) throw Error(XM_ERR_COMM, "shm_open failed");
        if (ftruncate(fd, (off_t)sizeof(IpcShared)) != 0) { close(fd); throw Error(XM_ERR_COMM, "ftruncate failed"); }
        p = mmap(nullptr, sizeof(IpcShared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) throw Error(XM_ERR_COMM, "mmap failed");
        static_cast<IpcShared *>(p)->stamp.store(wall_ns(), std::memory_order_release);
    } else {
        const auto t0 = clk::now();
        const unsigned long long fresh = (unsigned long long)((limit + 60.0) * 1e9);
        for (;;) {
            const int fd = shm_open(name, O_RDWR, 0600);
            if (fd >= 0) {
                struct stat sb;
                if (fstat(fd, &sb) == 0 && (size_t)sb.st_size >= sizeof(IpcShared)) {
                    void *q = mmap(nullptr, sizeof(IpcShared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
                    if (q != MAP_FAILED) {
                        const unsigned long long st = static_cast<IpcShared *>(q)->stamp.load(std::memory_order_acquire), now = wall_ns();
                        if (st != 0 && now - st < fresh) { p = q; ino = (unsigned long long)sb.st_ino; dev = (unsigned long long)sb.st_dev; close(fd); break; }
                        munmap(q, sizeof(IpcShared));
                    }
                }
                close(fd);
            }
            if (since(t0) > limit) throw Error(XM_ERR_COMM, "peer rendezvous: rank 0 did not create the shared segment in time");
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
    }
    auto g = std::make_shared<IpcPeerGroup>();
    g->sh = static_cast<IpcShared *>(p);   // a fresh segment is zero-filled
    g->name = name; g->me = rank; g->world = world; g->limit = limit;
    g->seg_ino = ino; g->seg_dev = dev;
    for (int r = 0; r < world; ++r) g->device[r] = (r == rank) ? device : -1;
    if (spin_seconds > 0) g->spin_seconds = spin_seconds;
    {   // which physical GPU this rank drives (two processes of a 1-GPU test box share one)
        char bus[64] = {};
        unsigned long long h = 1469598103934665603ull;
        if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) { (void)hipGetLastError(); std::snprintf(bus, sizeof(bus), "device-%d", device); }
        for (const char *c = bus; *c; ++c) { h ^= (unsigned char)*c; h *= 1099511628211ull; }
        g->sh->devid[rank] = h | 1ull;
    }
    return g;
}

// All-gathers with known contents that change from round to round at the same addresses: the transport has to prove itself on this
// machine before the solver relies on it -- three rounds read through system-scope loads (what allgather does), six through ordinary
// loads behind the acquire fence (what the fused tCG exchange does; a stale line of an earlier round would show up as a mismatch).
bool peer_selftest(PeerComm &c) {
    const size_t cnt = 4096;
    bool ok = true;
    double *d = nullptr;
    hipStream_t st = nullptr;
    std::vector<double> h(cnt * (size_t)c.world);
    try {
        XM_HIP_CHECK(hipMalloc((void **)&d, h.size() * sizeof(double)));
        XM_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        {   // load the code object and run both kernels once on this rank alone (a group of one: nothing is pushed, nothing awaited), then
            // meet the peers: the bounded waits below must measure the transport, not a peer that is still loading its kernels
            PeerPtrs solo = c.ptrs();
            solo.world = 1; solo.rank = 0;
            solo.stage[0] = c.stage_of(c.rank); solo.flags[0] = c.flags_of(c.rank);
            hipLaunchKernelGGL(peer_push_kernel, dim3(1), dim3(256), 0, st, solo, d, cnt, (size_t)0, 0, 0ull, c.tickets() + 6);
            hipLaunchKernelGGL(peer_wait_kernel, dim3(1), dim3(256), 0, st, solo, d, cnt, (size_t)0, 0, 0ull, c.spin_ticks(), c.herr_dev, 0);
            check_launch("peer self-test warm-up");
            XM_HIP_CHECK(hipStreamSynchronize(st));
            c.g->barrier(c.hb, "transport self-test");
        }
        for (int it = 0; it < 9 && ok; ++it) {
            c.plain_reads = (it >= 3);
            std::fill(h.begin(), h.end(), -1.0);
            for (size_t i = 0; i < cnt; ++i) h[(size_t)c.rank * cnt + i] = 1e6 * it + 1e4 * c.rank + (double)i;
            XM_HIP_CHECK(hipMemcpy(d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
            c.allgather(d, cnt, st);
            XM_HIP_CHECK(hipStreamSynchronize(st));
            c.check_device_error();
            XM_HIP_CHECK(hipMemcpy(h.data(), d, h.size() * sizeof(double), hipMemcpyDeviceToHost));
            for (int r = 0; r < c.world && ok; ++r)
                for (size_t i = 0; i < cnt; ++i)
                    if (h[(size_t)r * cnt + i] != 1e6 * it + 1e4 * r + (double)i) { ok = false; break; }
        }
    } catch (const Error &) { ok = false; }
    c.plain_reads = 0;
    if (st) (void)hipStreamDestroy(st);
    if (d) (void)hipFree(d);
    return ok;
}

}  // namespace
bool peer_comm_selftest(Comm &c) {
    PeerComm *pc = dynamic_cast<PeerComm *>(&c);
    if (!pc) throw Error(XM_ERR_ARG, "peer_comm_selftest: not a peer communicator");
    return peer_selftest(*pc);
}
std::shared_ptr<Comm> rccl_comm_create(int rank, int world, const unsigned char id[128]) {
    if (world < 2 || rank < 0 || rank >= world || !id) throw Error(XM_ERR_ARG, "rccl_comm_create: bad rank/world/id");
    {
        std::lock_guard<std::mutex> lk(g_mu);
        load_rccl(nullptr);
    }
    auto c = std::make_shared<RcclComm>();
    ncclUniqueId u;
    std::memcpy(u.internal, id, 128);
    check(g_rccl.CommInitRank(&c->comm, world, u, rank), "ncclCommInitRank");   // blocks until every rank of the group has called it
    c->rank = rank;
    c->world = world;
    return c;
}
namespace {
// all ranks return the same answer: a communicator that passed the self-test everywhere, or nullptr (why -> *why)
std::shared_ptr<PeerComm> ipc_comm_try(int rank, int world, int device, const char *name, double limit, std::string *why) {
    std::shared_ptr<IpcPeerGroup> g;
    std::shared_ptr<PeerComm> c;
    unsigned long long hb = 0;
    // every way out of the probation marks the group failed AND aborted first: a rank that is still waiting in one of the barriers
    // below, or arrives at it late, then leaves with the same answer as this one (the decision is collective)
    auto give_up = [&](const std::string &what) -> std::shared_ptr<PeerComm> {
        if (why) *why = what;
        if (g && g->sh) { g->sh->failed.store(1); g->sh->aborted.store(1); }
        shm_unlink(name);   // whoever gets here first removes the name (a segment left behind would poison the next job of that name)
        return nullptr;
    };
    try {
        const auto t_open = clk::now();
        for (;;) {
            g = ipc_group_open(rank, world, device, name, limit, 10.0);   // shorter device-side waits while the transport is on probation
            hb = 0;
            const int met = g->rendezvous(hb);
            if (met == -1 && since(t_open) < limit) { g.reset(); continue; }   // a stale segment of an earlier job of this name: open the name again
            if (met != 1 || g->sh->failed.load()) return give_up("not every rank reached the rendezvous segment (ranks on several nodes?)");
            break;
        }
        c = std::make_shared<PeerComm>(g, rank, hb);
        // the tCG exchange buffer is exported and mapped HERE, once, at its full size: a handle that a peer cannot open must end in the
        // collective fallback below, not in the middle of a solve
        PeerXchg probe;
        c->xchg_setup((size_t)1 << 20, probe);
    } catch (const Error &e) {
        return give_up(e.what());
    }
    bool ok = false;
    try { ok = peer_selftest(*c); } catch (...) { ok = false; }
    if (!ok) g->sh->failed.store(1);
    const bool met = g->barrier_nothrow(c->hb);
    if (!met || g->sh->failed.load()) return give_up(ok ? "a peer failed the transport self-test" : "the transport self-test failed on this rank");
    // confirmation: nobody installs the communicator before everybody has seen `failed` clear behind the barrier above
    if (!g->barrier_nothrow(c->hb) || g->sh->failed.load()) return give_up("a peer gave the transport up after the self-test");
    if (rank == 0) shm_unlink(name);   // everybody has it mapped and agrees
    g->limit = 120.0;   // ranks reach the solver's set-up barriers seconds apart (each uploads its own strip of Q first)
    return c;
}

std::string ipc_name_of(const unsigned char id[128]) {   // every rank of a job derives the same segment name from rank 0's unique id
    unsigned long long h = 1469598103934665603ull;
    for (int i = 0; i < 128; ++i) { h ^= id[i]; h *= 1099511628211ull; }
    char buf[64];
    std::snprintf(buf, sizeof(buf), "/xm_ipc_%016llx", h);
    return buf;
}
}  // namespace

static std::shared_ptr<Comm> ipc_upgrade(int rank, int world, int device, const unsigned char id[128], const std::shared_ptr<Comm> &keep) {
    const char *pe = std::getenv("XM_COMM_PEER");
    if (world <= 1 || world > kMaxPeers || id == nullptr || (pe && *pe == '0')) return nullptr;
    {   // a launcher that says the job spans several nodes (torch.distributed.run: LOCAL_WORLD_SIZE < WORLD_SIZE): the rendezvous segment
        // is per node, nobody would ever complete it -- keep RCCL at once instead of after the rendezvous time-out
        const char *lw = std::getenv("LOCAL_WORLD_SIZE");
        if (lw && *lw && std::atoi(lw) != world) { keep->fallback_note = "direct peer exchange not used (ranks on several nodes): RCCL all-gather"; return nullptr; }
    }
    std::string why;
    auto pc = ipc_comm_try(rank, world, device, ipc_name_of(id).c_str(), 30.0, &why);
    if (!pc) {
        if (std::getenv("XM_COMM_TRACE")) std::fprintf(stderr, "xm_comm_init: rank %d keeps RCCL (%s)\n", rank, why.c_str());
        keep->fallback_note = "direct peer exchange not used (" + why + "): RCCL all-gather";
        return nullptr;
    }
    pc->g->spin_seconds = 20.0;
    pc->keep = keep;
    return pc;
}

void comm_init_ipc(int rank, int world, int device, const char *name, double spin_seconds) {
    if (world < 1 || rank < 0 || rank >= world || !name) throw Error(XM_ERR_ARG, "bad rank/world/name");
    XM_HIP_CHECK(hipSetDevice(device));
    comm_finalize();
    std::string why;
    auto c = ipc_comm_try(rank, world, device, name, 120.0, &why);
    if (!c) throw Error(XM_ERR_COMM, "direct peer exchange between processes is not available: " + why);
    c->g->spin_seconds = spin_seconds > 0 ? spin_seconds : 20.0;
    std::lock_guard<std::mutex> lk(g_mu);
    g_default = c;
}

}  // namespace xm
