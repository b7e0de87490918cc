// xm_bench.hip -- timing hooks of the micro-benchmarks (include/xm_bench.h; scripts/kbench_*.py).  Not part of the product ABI
// (include/xm_amd.h does not declare them): HIP-event timings of single kernels through the same launchers the solver uses.
#include <algorithm>
#include <cstring>
#include <vector>

#include "../../include/xm_bench.h"
#include "xm_schur.h"
#include "xm_sell.h"
#include "xm_symw.h"
#include "xm_solver.h"

namespace {
thread_local std::string g_berr;
int bfail(const xm::Error &e) { g_berr = e.what(); return e.code; }
int bfail(const std::exception &e) { g_berr = e.what(); return XM_ERR_HIP; }
#define XM_TRY try {
#define XM_CATCH                                                       \
    }                                                                  \
    catch (const xm::Error &e) { return bfail(e); }                    \
    catch (const std::exception &e) { return bfail(e); }
void require_device() {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt < 1) throw xm::Error(XM_ERR_HIP, "no HIP device available");
}
xm::CamArgs plain_args(int64_t n, double *out) {
    xm::CamArgs a;
    std::memset(&a, 0, sizeof(a));
    a.nloc = (int)n;
    a.out = out;
    return a;
}
// average milliseconds of `reps` back-to-back calls of f() on the NULL stream after `warm` untimed ones
template <class F>
double time_launches(int warm, int reps, F f) {
    hipEvent_t e0, e1;
    XM_HIP_CHECK(hipEventCreate(&e0)); XM_HIP_CHECK(hipEventCreate(&e1));
    for (int i = 0; i < warm; ++i) f();
    XM_HIP_CHECK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < reps; ++i) f();
    XM_HIP_CHECK(hipEventRecord(e1, nullptr));
    XM_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0;
    XM_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return (double)ms / std::max(reps, 1);
}
}  // namespace

extern "C" {
const char *xm_bench_last_error(void) { return g_berr.c_str(); }

static int g_sym_alternate = 1;
int xm_qw_dense_time(const double *dq, int64_t n, int o, const double *dW, double *dOut, int reps, double *ms_avg) {
    XM_TRY
    xm::CamArgs a = plain_args(n, dOut);
    const int64_t ld = xm::dense_ld(n);
    int it = 0;   // consecutive products alternate the tile direction as the solver's do (xm_bench_symv_k(k, 0): always left to right)
    const double ms = time_launches(3, reps, [&] { a.rev = g_sym_alternate ? (it++ & 1) : 0; xm::launch_qw_dense(o, xm::EPI_PLAIN, dq, ld, dW, 1.0, a, nullptr); });
    if (ms_avg) *ms_avg = ms;
    return XM_OK;
    XM_CATCH
}
int xm_bench_dense_policy(int nt) {
    xm::qw_bench_nt(nt);
    return XM_OK;
}
int xm_qw_dense_sym_time(const double *dq, int64_t n, int o, const double *dW, double *dOut, int reps, double *ms_avg) {
    XM_TRY
    const int64_t ld = xm::dense_ld(n);
    xm::DevBuf<double> prow, pcol;
    prow.alloc(xm::sym_prow_count((int)n, ld, o));
    pcol.alloc(xm::sym_pcol_count((int)n, ld, o), false);
    const xm::CamArgs a = plain_args(n, dOut);
    int it = 0;   // consecutive products alternate the sweep direction as the solver's do (xm_bench_symv_k(k, 0): always top-down)
    const double ms = time_launches(3, reps, [&] { xm::launch_qw_sym(o, xm::EPI_PLAIN, dq, ld, dW, 1.0, a, prow.p, pcol.p, nullptr, g_sym_alternate ? (it++ & 1) : 0); });
    if (ms_avg) *ms_avg = ms;
    return XM_OK;
    XM_CATCH
}
int xm_bench_symv_k(int k, int alternate, int kf) {
    xm::symv_bench_k(k, kf);
    g_sym_alternate = alternate;
    return XM_OK;
}
int xm_qw_dense_sym_trace(const double *dq, int64_t n, int o, const double *dW, double *dOut, unsigned long long *trace_host, int64_t trace_cap,
                          int grid[2], int *slots) {
    XM_TRY
    const int64_t ld = xm::dense_ld(n);
    const xm::CamArgs a = plain_args(n, dOut);
    xm::launch_qw_sym_traced(o, nullptr, ld, nullptr, a, nullptr, nullptr, nullptr, grid, nullptr);     // grid only
    *slots = xm::symv_trace_slots();
    const int64_t need = (int64_t)grid[0] * grid[1] * 4 * *slots;
    if (trace_host == nullptr) return XM_OK;
    if (trace_cap < need) throw xm::Error(XM_ERR_ARG, "trace buffer too small");
    xm::DevBuf<double> prow, pcol;
    prow.alloc(xm::sym_prow_count((int)n, ld, o));
    pcol.alloc(xm::sym_pcol_count((int)n, ld, o), false);
    xm::DevBuf<unsigned long long> tr;
    tr.alloc((size_t)need);
    for (int i = 0; i < 3; ++i) xm::launch_qw_sym(o, xm::EPI_PLAIN, dq, ld, dW, 1.0, a, prow.p, pcol.p, nullptr);   // warm: caches in their steady state
    XM_HIP_CHECK(hipMemsetAsync(tr.p, 0, (size_t)need * 8, nullptr));
    xm::launch_qw_sym_traced(o, dq, ld, dW, a, prow.p, pcol.p, tr.p, grid, nullptr);
    XM_HIP_CHECK(hipDeviceSynchronize());
    XM_HIP_CHECK(hipMemcpy(trace_host, tr.p, (size_t)need * 8, hipMemcpyDeviceToHost));
    return XM_OK;
    XM_CATCH
}
int xm_qw_dense_strip_time(const double *dq, int64_t nloc, int64_t n, int o, const double *dW, double *dOut, int reps, double *ms_avg) {
    XM_TRY
    if (nloc < 1 || nloc > n) throw xm::Error(XM_ERR_ARG, "bad strip");
    const xm::CamArgs a = plain_args(nloc, dOut);
    const int64_t ld = xm::dense_ld(n);
    const double ms = time_launches(3, reps, [&] { xm::launch_qw_dense(o, xm::EPI_PLAIN, dq, ld, dW, 1.0, a, nullptr); });
    if (ms_avg) *ms_avg = ms;
    return XM_OK;
    XM_CATCH
}
int xm_qw_dense_strip_ks(const double *dq, int64_t nloc, int64_t n, int o, const double *dW, double *dOut, double alpha, int ks, int reps,
                         double *ms_avg, int *ks_used) {
    XM_TRY
    if (nloc < 1 || nloc > n || ks < 0 || ks > 8) throw xm::Error(XM_ERR_ARG, "bad argument");
    const int64_t ld = xm::dense_ld(n);
    if (ks == 0) ks = xm::qw_dense_split_k((int)nloc, ld);
    if (ks_used) *ks_used = ks;
    xm::CamArgs a = plain_args(nloc, dOut);
    xm::DevBuf<double> ksum;
    xm::DevBuf<unsigned int> kcount;
    if (ks > 1) {
        ksum.alloc((size_t)ks * nloc * 3 * xm::pitch_of(o));
        kcount.alloc((size_t)xm::qw_grid((int)nloc));
        a.ks = ks; a.ksum = ksum.p; a.kcount = kcount.p;
    }
    xm::launch_qw_dense(o, xm::EPI_PLAIN, dq, ld, dW, alpha, a, nullptr);
    XM_HIP_CHECK(hipDeviceSynchronize());
    if (reps > 0) {
        const double ms = time_launches(0, reps, [&] { xm::launch_qw_dense(o, xm::EPI_PLAIN, dq, ld, dW, alpha, a, nullptr); });
        if (ms_avg) *ms_avg = ms;
    }
    return XM_OK;
    XM_CATCH
}
static int g_bsr_binned = 1;
int xm_bench_bsr_binned(int on) { g_bsr_binned = on; return XM_OK; }
int xm_qw_bsr3_time(const int64_t *rp, const int32_t *ci, const double *bl, int64_t n, int o, const double *dW, double *dOut, int reps,
                    double *ms_avg) {
    XM_TRY
    const xm::CamArgs a = plain_args(n, dOut);
    // as in a solve: the block count decides the load policy of the block stream (xm_bench_dense_policy overrides it) and the rows are
    // binned by their number of windows (xm_bench_bsr_binned(0): camera order)
    std::vector<int64_t> rph((size_t)n + 1);
    XM_HIP_CHECK(hipMemcpy(rph.data(), rp, rph.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
    const int64_t nb = rph[(size_t)n];
    xm::DevBuf<int4> dri;
    if (g_bsr_binned) {
        std::vector<int4> ri;
        xm::bsr_build_rowinfo(rph.data(), (int)n, ri);
        dri.alloc(std::max<size_t>(ri.size(), 1));
        if (!ri.empty()) XM_HIP_CHECK(hipMemcpy(dri.p, ri.data(), ri.size() * sizeof(int4), hipMemcpyHostToDevice));
    }
    const double ms = time_launches(3, reps, [&] { xm::launch_qw_bsr3(o, xm::EPI_PLAIN, rp, ci, bl, dW, 1.0, a, nullptr, nb, g_bsr_binned ? dri.p : nullptr); });
    if (ms_avg) *ms_avg = ms;
    return XM_OK;
    XM_CATCH
}
#ifdef XM_BSR_TRACE
void xm_bsr_trace_set(unsigned long long *p);
// experiment builds only (-DXM_BSR_TRACE): one traced launch of the plain block-CSR product as a solve runs it; trace_host[grid * 4 wavefronts][8]
int xm_qw_bsr3_trace(const int64_t *rp, const int32_t *ci, const double *bl, int64_t n, int o, const double *dW, double *dOut,
                     unsigned long long *trace_host) {
    XM_TRY
    const xm::CamArgs a = plain_args(n, dOut);
    std::vector<int64_t> rph((size_t)n + 1);
    XM_HIP_CHECK(hipMemcpy(rph.data(), rp, rph.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
    std::vector<int4> ri;
    xm::bsr_build_rowinfo(rph.data(), (int)n, ri);
    xm::DevBuf<int4> dri;
    dri.alloc(std::max<size_t>(ri.size(), 1));
    if (!ri.empty()) XM_HIP_CHECK(hipMemcpy(dri.p, ri.data(), ri.size() * sizeof(int4), hipMemcpyHostToDevice));
    const int4 *rip = g_bsr_binned ? dri.p : nullptr;
    const size_t need = (size_t)((n + xm::kBsrRows - 1) / xm::kBsrRows) * 4 * 8;
    xm::DevBuf<unsigned long long> tr;
    tr.alloc(need);
    for (int i = 0; i < 3; ++i) xm::launch_qw_bsr3(o, xm::EPI_PLAIN, rp, ci, bl, dW, 1.0, a, nullptr, rph[(size_t)n], rip);
    XM_HIP_CHECK(hipDeviceSynchronize());
    xm_bsr_trace_set(tr.p);
    xm::launch_qw_bsr3(o, xm::EPI_PLAIN, rp, ci, bl, dW, 1.0, a, nullptr, rph[(size_t)n], rip);
    XM_HIP_CHECK(hipDeviceSynchronize());
    xm_bsr_trace_set(nullptr);
    XM_HIP_CHECK(hipMemcpy(trace_host, tr.p, need * 8, hipMemcpyDeviceToHost));
    return XM_OK;
    XM_CATCH
}
#endif
// handle: what xm_sell_create / xm_sell_create2 returned; dWpad16 may be NULL (include/xm_amd.h: xm_qw_sell_padded)
int xm_qw_sell_time(void *handle, int o, const double *dW, const double *dWpad16, double *dOut, int gather_mode, int reps, double *ms_avg) {
    XM_TRY
    if (!handle) throw xm::Error(XM_ERR_ARG, "null handle");
    if (gather_mode != 0 && gather_mode != 1) throw xm::Error(XM_ERR_ARG, "gather_mode must be 0 or 1");
    xm::SellMatrix &m = *static_cast<xm::SellMatrix *>(handle);
    const xm::CamArgs a = plain_args(m.nloc(), dOut);
    const double ms = time_launches(3, reps, [&] { xm::launch_qw_sell(o, xm::EPI_PLAIN, m, dW, 1.0, a, gather_mode, nullptr, dWpad16); });
    if (ms_avg) *ms_avg = ms;
    return XM_OK;
    XM_CATCH
}
// variant: 0 thread per camera (MGS-QR) | 1 polar | 2 MGS-QR with a quad of lanes per camera; *ms_avg (may be NULL: one untimed call) = HIP-event average
int xm_retract_variant(int64_t n, int o, const double *dR, const double *ds, const double *dD, const double *dds, double t, double *dRout,
                       double *dsout, int variant, int reps, double *ms_avg) {
    XM_TRY
    if (variant < 0 || variant > 2 || reps < 1) throw xm::Error(XM_ERR_ARG, "bad argument");
    xm::launch_retract(o, (int)n, 0, dR, ds, dD, dds, t, dRout, dsout, nullptr, nullptr, variant);
    if (!ms_avg) { XM_HIP_CHECK(hipDeviceSynchronize()); return XM_OK; }
    *ms_avg = time_launches(0, reps, [&] { xm::launch_retract(o, (int)n, 0, dR, ds, dD, dds, t, dRout, dsout, nullptr, nullptr, variant); });
    return XM_OK;
    XM_CATCH
}
// xm_recover_rotations with the projection kernel chosen -- variant 0: one thread per camera (the default), 1: one wavefront per camera with
// cross-lane reductions (north_star's form) -- and that launch timed over `reps` repetitions
int xm_recover_rotations_variant(int64_t n, int r, const double *R, const double *s, double *rot, double *scale, int *n_negative_det, int variant,
                                 int reps, double *ms_avg) {
    XM_TRY
    xm::recover_rotations(n, r, R, s, rot, scale, n_negative_det, variant, reps, ms_avg);
    return XM_OK;
    XM_CATCH
}
int xm_peer_allgather_bench(int world, int gpu_map, int64_t count, int reps, double *us_avg) {
    XM_TRY
    require_device();
    if (us_avg) *us_avg = xm::peer_allgather_bench(world, gpu_map, count, reps);
    return XM_OK;
    XM_CATCH
}
// ONE rank's share of the multi-rank symmetric window product (xm_symw.h) on this GPU: rank `cam0 / nloc` of `world`, its row strip filled
// with an arbitrary pattern (timing only).  ms[0] = sweep + column sums, ms[1] = per-camera sum + plain epilogue; bytes = what the sweep
// streams.  The all-gather between the two is not part of it.
int xm_qw_symw_time(int64_t ntot, int nloc, int cam0, int o, int world, int reps, double ms[2], int64_t *bytes) {
    XM_TRY
    require_device();
    if (ntot < 2 || nloc < 2 || world < 1 || reps < 1 || !ms) throw xm::Error(XM_ERR_ARG, "bad argument");
    if (cam0 < 0 || cam0 % nloc != 0 || cam0 / nloc >= world || (int64_t)nloc * world < ntot || !(o == 1 || (o >= 3 && o <= 5)))
        throw xm::Error(XM_ERR_ARG, "xm_qw_symw_time: cam0 must be rank * nloc with rank < world, nloc * world >= ntot, o in 1, 3..5");
    const int64_t ld = xm::dense_ld(ntot);
    xm::SymwProduct sp(ntot, nloc, cam0, ld, nullptr);
    sp.ensure(o, world);
    xm::DevBuf<double> Q, W, out;
    Q.alloc((size_t)3 * nloc * (size_t)ld, false);
    W.alloc((size_t)ld * xm::pitch_of(o) + 16);
    out.alloc((size_t)3 * nloc * xm::pitch_of(o));
    XM_HIP_CHECK(hipMemset(Q.p, 0x3c, (size_t)3 * nloc * (size_t)ld * sizeof(double)));   // finite pattern
    XM_HIP_CHECK(hipDeviceSynchronize());
    xm::CamArgs a = plain_args(nloc, out.p);
    a.cam0 = cam0;
    hipEvent_t e0, e1, e2;
    XM_HIP_CHECK(hipEventCreate(&e0)); XM_HIP_CHECK(hipEventCreate(&e1)); XM_HIP_CHECK(hipEventCreate(&e2));
    const int rank = cam0 / nloc;
    for (int i = 0; i < 2; ++i) { sp.sweep(o, Q.p, W.p, nullptr, rank, nullptr); sp.reduce(o, xm::EPI_PLAIN, 1.0, a, world, nullptr); }
    float t_sw = 0, t_rd = 0;
    for (int i = 0; i < reps; ++i) {
        XM_HIP_CHECK(hipEventRecord(e0, nullptr));
        sp.sweep(o, Q.p, W.p, nullptr, rank, nullptr);
        XM_HIP_CHECK(hipEventRecord(e1, nullptr));
        sp.reduce(o, xm::EPI_PLAIN, 1.0, a, world, nullptr);
        XM_HIP_CHECK(hipEventRecord(e2, nullptr));
        XM_HIP_CHECK(hipEventSynchronize(e2));
        float x = 0, y = 0;
        XM_HIP_CHECK(hipEventElapsedTime(&x, e0, e1)); XM_HIP_CHECK(hipEventElapsedTime(&y, e1, e2));
        t_sw += x; t_rd += y;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
    ms[0] = t_sw / reps; ms[1] = t_rd / reps;
    if (bytes) *bytes = sp.stream_bytes();
    return XM_OK;
    XM_CATCH
}

// What does a GRID-WIDE BARRIER inside one launch cost against a kernel boundary?  ("open": folding the step of a tCG iteration into
// the launch before it needs one.)  `blocks` workgroups of 256 threads, all resident; per round every thread writes a value, the workgroups meet
// (release fence, arrival counter, bounded spin, acquire fence) and every thread checks a value another workgroup wrote.  us[0] = microseconds per
// round inside ONE launch, us[1] = per round as `rounds` launches of the same work split at the barrier, us[2] = mismatches seen (must be 0).
int xm_bench_grid_barrier(int blocks, int rounds, int reps, double us[3]);

}  // extern "C"

namespace {
__global__ __launch_bounds__(256) void grid_barrier_kernel(unsigned int *counter, double *buf, int rounds, int round0, int do_barrier, long long spin_ticks,
                                                          unsigned int *bad) {
    __shared__ int ok;
    const int n = gridDim.x * 256, gid = blockIdx.x * 256 + threadIdx.x;
    for (int r = round0; r < round0 + rounds; ++r) {
        if (r > round0 || round0 > 0) {   // check what the round before left (another workgroup's element, 7 workgroups away)
            const int j = (gid + 7 * 256) % n;
            if (buf[(size_t)((r - 1) & 1) * n + j] != (double)(r - 1) * 0.5 + j) atomicAdd(bad, 1u);
        }
        buf[(size_t)(r & 1) * n + gid] = (double)r * 0.5 + gid;
        if (!do_barrier) continue;
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned int target = (unsigned int)(r - round0 + 1) * gridDim.x;
            const long long t0 = wall_clock64();
            int good = 1;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (wall_clock64() - t0 > spin_ticks) { good = 0; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            ok = good;
        }
        __syncthreads();
        if (!ok) { if (threadIdx.x == 0) atomicAdd(bad, 1u << 20); return; }
    }
}
}  // namespace

extern "C" int xm_bench_grid_barrier(int blocks, int rounds, int reps, double us[3]) {
    XM_TRY
    require_device();
    if (blocks < 1 || blocks > 1024 || rounds < 1 || reps < 1 || !us) throw xm::Error(XM_ERR_ARG, "bad argument");
    xm::DevBuf<double> buf;
    xm::DevBuf<unsigned int> cnt;
    buf.alloc((size_t)2 * blocks * 256);
    cnt.alloc(2);
    const long long spin = 100000000ll / 20;   // 50 ms at 100 MHz: a workgroup that is not resident would otherwise hang the launch
    auto fused = [&]() {
        XM_HIP_CHECK(hipMemsetAsync(cnt.p, 0, sizeof(unsigned int), nullptr));
        hipLaunchKernelGGL(grid_barrier_kernel, dim3(blocks), dim3(256), 0, nullptr, cnt.p, buf.p, rounds, 0, 1, spin, cnt.p + 1);
    };
    auto split = [&]() {
        for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(grid_barrier_kernel, dim3(blocks), dim3(256), 0, nullptr, cnt.p, buf.p, 1, r, 0, spin, cnt.p + 1);
    };
    const double ms_f = time_launches(2, reps, fused), ms_s = time_launches(2, reps, split);
    unsigned int bad = 0;
    XM_HIP_CHECK(hipMemcpy(&bad, cnt.p + 1, sizeof(bad), hipMemcpyDeviceToHost));
    us[0] = ms_f * 1e3 / rounds; us[1] = ms_s * 1e3 / rounds; us[2] = (double)bad;
    return XM_OK;
    XM_CATCH
}

extern "C" {
}  // extern "C"
