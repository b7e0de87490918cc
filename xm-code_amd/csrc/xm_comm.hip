// xm_comm.hip — row-partition communicator: RCCL over xGMI, one process per GPU.
//
// The reference is single-GPU (memory.h:54 gpu_id == 0, no NCCL anywhere), so this is new design:
// cameras are split in contiguous equal ranges; per Q*W product one in-place all-gather of the product input W
// (3*nloc*OP doubles per rank) and, per tCG iteration, two all-gathers of a few hundred partial sums — gathered rather
// than all-reduced so that every rank adds them in the same fixed order and takes bit-identical branch decisions.
// RCCL is dlopen()ed at run time (torch bundles its own librccl.so with the same soname; whichever the process already
// loaded is reused, otherwise /opt/rocm/lib/librccl.so).  No link-time dependency, no collective unless world > 1.
#include <dlfcn.h>

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

void comm_finalize() {
    if (g_rccl.comm) { g_rccl.CommDestroy(g_rccl.comm); g_rccl.comm = nullptr; }
    g_comm.rank = 0;
    g_comm.world = 1;
    g_comm.forced = false;
}

void Comm::allgather(double *buf, size_t count, hipStream_t st) {
    if (!g_rccl.comm) return;
    check(g_rccl.AllGather(buf + (size_t)rank * count, buf, count, kNcclFloat64, g_rccl.comm, st), "ncclAllGather");
}

}  // namespace xm
