// xm_symw.hip — half-traffic product for a symmetric dense Q under the camera row partition: cyclic half window (rationale: xm_symw.h).
// The sweep is the vertical sweep of xm_kernels.hip:qw_symv_kernel with a work list instead of a triangular grid and the window
// predicate of xm_symw.h instead of "right of the diagonal".  Replaces cublasDgemm on the symmetric C (Dense/matmul.h:42-87, XM_main.cu:191).
#include "xm_symw.h"

#include <algorithm>
#include <cmath>
#include <type_traits>

#include "xm_device.h"

namespace xm {

// ------------------------------------------------------------------------------------------------------------------
// host: the plan
// ------------------------------------------------------------------------------------------------------------------
void symw_plan_build(int64_t ntot, int nloc, int cam0, int K, SymwPlan &out) {
    if (ntot < 2 || (ntot & 1) || nloc < 2 || (nloc & 1) || (cam0 & 1) || cam0 < 0 || (int64_t)cam0 + nloc > ntot)
        throw Error(-2, "symmetric window product: camera counts and offsets must be even");
    if (3 * ntot > 2000000000LL) throw Error(-2, "symmetric window product: too many cameras");
    out = SymwPlan();
    SymwGeom &g = out.g;
    g.T = (int)(ntot / 2);
    g.Th = (g.T + 1) / 2;
    g.tie = (g.T % 2 == 0) ? 1 : 0;
    g.t0 = cam0 / 2;
    g.nsteps = nloc / 2;
    g.nstrips = (int)(((int64_t)6 * g.T + kSwStrip - 1) / kSwStrip);
    // chunk length: as for the triangular sweep (xm_kernels.hip:symv_k), from the number of (strip, step) pairs of this rank
    int64_t total = 0;
    for (int s = 0; s < g.nstrips; ++s)
        for (int j = 0; j < g.nsteps; ++j) total += symw_any(g, g.t0 + j, s) ? 1 : 0;
    if (K <= 0) {
        int64_t k = (int64_t)(std::sqrt((double)total) / 13.0 + 0.5);
        K = (int)std::min<int64_t>(64, std::max<int64_t>(2, k));
    }
    out.K = K;
    out.strip_ptr.assign((size_t)g.nstrips + 1, 0);
    for (int s = 0; s < g.nstrips; ++s) {
        int j = 0;
        while (j < g.nsteps) {
            if (!symw_any(g, g.t0 + j, s)) { ++j; continue; }
            int e = j;
            while (e < g.nsteps && e - j < K && symw_any(g, g.t0 + e, s)) ++e;
            out.items.push_back(SymwItem{s, j, e, 0});
            j = e;
        }
        out.strip_ptr[(size_t)s + 1] = (int32_t)out.items.size();
    }
}
size_t symw_prow_count(const SymwPlan &p, int o) { return (size_t)p.g.nstrips * 6 * (size_t)p.g.nsteps * o; }
size_t symw_pcol_count(const SymwPlan &p, int o) { return std::max<size_t>(p.items.size(), 1) * kSwStrip * (size_t)o; }
size_t symw_csum_count(int64_t ntot, int o) { return (size_t)3 * (size_t)ntot * o; }

// ------------------------------------------------------------------------------------------------------------------
// device: the sweep.  One wavefront per work item (strip of 256 columns x up to K steps of 6 rows of this rank's strip of Q).
// ------------------------------------------------------------------------------------------------------------------
template <int O, bool NT>
__global__ __launch_bounds__(256) void qw_symw_kernel(const double *__restrict__ Q, int64_t ld, const double *__restrict__ W, SymwGeom g,
                                                       const SymwItem *__restrict__ items, int nitems, const TcgScal *__restrict__ scal,
                                                       double *__restrict__ Prow, double *__restrict__ Pcol) {
    constexpr int OP = pitch_of(O), V = 6 * O;
    if (scal != nullptr) {
        if (scal->status != 0) return;
    }
    __shared__ __attribute__((aligned(16))) double lds[4][V * 64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int it = blockIdx.x * 4 + wave;
    if (it >= nitems) return;                                    // wave-uniform: no workgroup barrier in this kernel
    const SymwItem item = items[it];
    const int s = item.s, jb = item.jb, je = item.je;
    const int64_t c0 = (int64_t)s * kSwStrip;
    const int64_t M = (int64_t)6 * g.T;
    const bool half1 = c0 + 128 < ld;                            // ld is a multiple of 128: the strip may end after its first half
    const int64_t R = (int64_t)6 * g.nsteps;
    const int64_t R0 = (int64_t)6 * g.t0;                        // global index of this rank's first row
    double *L = lds[wave];
    const int64_t cA = c0 + 2 * lane;
    // column step of each of the lane's two column pairs (a pair never straddles a step: both bounds are even); -1: padding column
    int ucol[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int64_t c = cA + 128 * h;
        ucol[h] = (c < M && (h == 0 || half1)) ? (int)(c / 6) : -1;
    }

    double wc[2][2][O], ca[2][2][O];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int k = 0; k < O; ++k) {
                wc[h][e][k] = (ucol[h] >= 0) ? W[(size_t)(cA + 128 * h + e) * OP + k] : 0.0;
                ca[h][e][k] = 0.0;
            }

    const int64_t cB = half1 ? cA + 128 : cA;
    auto load_q = [&](int j, double2 (&q)[6][2]) {
        const double *row0 = Q + (size_t)6 * j * (size_t)ld;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const double2 *qp = reinterpret_cast<const double2 *>(row0 + (size_t)r * ld + (h ? cB : cA));
                double2 v;
                if (NT) v = make_double2(__builtin_nontemporal_load(&qp->x), __builtin_nontemporal_load(&qp->y));
                else v = *qp;
                const bool keep = (h == 0 || half1);
                q[r][h] = make_double2(keep ? v.x : 0.0, keep ? v.y : 0.0);
            }
        }
    };
    auto step = [&](int j, const double2 (&q)[6][2], auto masked) {
        constexpr bool MASK = decltype(masked)::value;
        const int t = g.t0 + j;
        const int64_t r0 = (int64_t)6 * j;
        double mr[2] = {1.0, 1.0}, mc[2] = {1.0, 1.0};
        if constexpr (MASK) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bool real = ucol[h] >= 0;
                const bool use = real && ucol[h] != t && symw_use(g, t, ucol[h]);
                const bool diag = real && ucol[h] == t;
                mr[h] = (use || diag) ? 1.0 : 0.0;   // row direction: the used blocks and the whole diagonal block
                mc[h] = use ? 1.0 : 0.0;             // column direction: the used blocks only
            }
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            double wr[O];
#pragma unroll
            for (int k = 0; k < O; ++k) wr[k] = W[(size_t)(R0 + r0 + r) * OP + k];   // wave-uniform: scalar loads
            double qr[2][2], qc[2][2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                qr[h][0] = MASK ? q[r][h].x * mr[h] : q[r][h].x; qr[h][1] = MASK ? q[r][h].y * mr[h] : q[r][h].y;
                qc[h][0] = MASK ? q[r][h].x * mc[h] : q[r][h].x; qc[h][1] = MASK ? q[r][h].y * mc[h] : q[r][h].y;
            }
#pragma unroll
            for (int k = 0; k < O; ++k) {
                double tt = qr[0][0] * wc[0][0][k];
                tt = fma(qr[0][1], wc[0][1][k], tt);
                tt = fma(qr[1][0], wc[1][0][k], tt);
                tt = fma(qr[1][1], wc[1][1][k], tt);
                L[(r * O + k) * 64 + lane] = tt;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 2; ++e) ca[h][e][k] = fma(qc[h][e], wr[k], ca[h][e][k]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // 64 addends per value: a 16-lane row takes value v = 4 i + (lane / 16), each lane four addends, DPP row sum (as qw_symv_kernel)
        const int gq = lane >> 4, jl = lane & 15;
#pragma unroll
        for (int v0 = 0; v0 < V; v0 += 4) {
            const int v = v0 + gq;
            double tt = 0.0;
            if (v < V) {
                const double2 a = *reinterpret_cast<const double2 *>(L + v * 64 + 4 * jl), b = *reinterpret_cast<const double2 *>(L + v * 64 + 4 * jl + 2);
                tt = (a.x + a.y) + (b.x + b.y);
            }
            tt = group_sum<16>(tt);
            if (jl == 0 && v < V) Prow[((size_t)s * (size_t)R + (size_t)r0) * O + v] = tt;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    auto run = [&](int j, const double2 (&q)[6][2]) {
        if (symw_full(g, g.t0 + j, s)) step(j, q, std::false_type{});   // wave-uniform
        else step(j, q, std::true_type{});
    };

    double2 qA[6][2], qB[6][2];
    load_q(jb, qA);
    int j = jb;
    for (; j + 1 < je; j += 2) {
        load_q(j + 1, qB);
        run(j, qA);
        if (j + 2 < je) load_q(j + 2, qA);
        run(j + 1, qB);
    }
    if (j < je) run(j, qA);

    // this item's column sums: 256 columns x O, [column in strip][k]
    double *pc = Pcol + (size_t)it * kSwStrip * O;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int k = 0; k < O; ++k) pc[(size_t)(2 * lane + 128 * h + e) * O + k] = ca[h][e][k];
}

// this rank's column sums per column: the items of a strip added in item order (fixed) -> csum[column * O + k]; zero where nothing was swept
template <int O>
__global__ __launch_bounds__(256) void symw_colsum_kernel(SymwGeom g, const int32_t *__restrict__ strip_ptr, const double *__restrict__ Pcol,
                                                           const TcgScal *__restrict__ scal, double *__restrict__ csum) {
    if (scal != nullptr) {
        if (scal->status != 0) return;
    }
    const int64_t M = (int64_t)6 * g.T;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;   // (column, k)
    if (idx >= M * O) return;
    const int64_t c = idx / O;
    const int k = (int)(idx - c * O);
    const int s = (int)(c / kSwStrip), cs = (int)(c - (int64_t)s * kSwStrip);
    double t = 0.0;
    for (int i = strip_ptr[s]; i < strip_ptr[s + 1]; ++i) t += Pcol[((size_t)i * kSwStrip + cs) * O + k];
    csum[idx] = t;
}

// per camera: row-direction partial sums of the strips its step touched + the column sums of every rank (all-gathered), fixed order, then
// the fused epilogue.  cs_all: world vectors of 3 * ntot * O doubles
template <int O, int EPI>
__global__ __launch_bounds__(256) void symw_reduce_kernel(SymwGeom g, const double *__restrict__ Prow, const double *__restrict__ cs_all, int world,
                                                           double alpha, CamArgs a) {
    if (EPI == EPI_HESS) {
        if (a.scal->status != 0) return;
    }
    __shared__ double red[kQwWaves][3];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cam = blockIdx.x * kQwWaves + wave;
    const bool active = cam < a.nloc;
    EpiOps eops;
    epi_prefetch<O, EPI>(eops, cam, lane, active, a);
    double acc[3][O];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < O; ++k) acc[r][k] = 0.0;
    if (active) {
        const int64_t R = (int64_t)6 * g.nsteps, M = (int64_t)6 * g.T;
        const int t = g.t0 + (cam >> 1);
        const int64_t grow = (int64_t)3 * (a.cam0 + cam);
        for (int i = lane; i < g.nstrips + world; i += 64) {   // lane i: strip i, then rank i - nstrips (fixed assignment -> fixed order)
            if (i < g.nstrips) {
                if (symw_any(g, t, i)) {
                    const double *p = Prow + ((size_t)i * (size_t)R + (size_t)cam * 3) * O;
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int k = 0; k < O; ++k) acc[r][k] += p[r * O + k];
                }
            } else {
                const double *p = cs_all + ((size_t)(i - g.nstrips) * (size_t)M + (size_t)grow) * O;
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int k = 0; k < O; ++k) acc[r][k] += p[r * O + k];
            }
        }
    }
    qw_finish<O, EPI, 64, kQwWaves>(cam, lane, wave, active, acc, alpha, a, eops, red);
}

// ------------------------------------------------------------------------------------------------------------------
// host: the per-rank object
// ------------------------------------------------------------------------------------------------------------------
SymwProduct::SymwProduct(int64_t ntot, int nloc, int cam0, int64_t ld, hipStream_t st) : ntot_(ntot), ld_(ld), nloc_(nloc) {
    symw_plan_build(ntot, nloc, cam0, 0, plan_);
    if (ld < (int64_t)6 * plan_.g.T || (ld % 128) != 0) throw Error(-2, "symmetric window product: bad leading dimension");
    items_.alloc(std::max<size_t>(plan_.items.size(), 1), false);
    strip_ptr_.alloc(plan_.strip_ptr.size(), false);
    if (!plan_.items.empty())
        XM_HIP_CHECK(hipMemcpyAsync(items_.p, plan_.items.data(), plan_.items.size() * sizeof(SymwItem), hipMemcpyHostToDevice, st));
    XM_HIP_CHECK(hipMemcpyAsync(strip_ptr_.p, plan_.strip_ptr.data(), plan_.strip_ptr.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    XM_HIP_CHECK(hipStreamSynchronize(st));
}

void SymwProduct::ensure(int omax, int world) {
    if (omax <= omax_ && world == world_) return;
    omax = std::max(omax, omax_);
    prow_.alloc(symw_prow_count(plan_, omax));
    pcol_.alloc(symw_pcol_count(plan_, omax), false);
    csum_.alloc(symw_csum_count(ntot_, omax) * (size_t)world);
    omax_ = omax; world_ = world;
}

int64_t SymwProduct::stream_bytes() const {
    int64_t steps = 0;
    for (const SymwItem &it : plan_.items) steps += it.je - it.jb;
    return steps * 6 * kSwStrip * 8;
}

static bool symw_nt(int nloc, int64_t ld) { return (size_t)nloc * 3 * (size_t)ld * sizeof(double) / 2 > ((size_t)240 << 20); }

template <int O>
static void symw_sweep_o(const SymwPlan &pl, const double *Q, int64_t ld, const double *W, const TcgScal *scal, const SymwItem *items,
                         const int32_t *strip_ptr, double *prow, double *pcol, double *csum, int nloc, hipStream_t st) {
    const int nitems = (int)pl.items.size();
    if (nitems > 0) {
        const dim3 g((nitems + 3) / 4), b(256);
        if (symw_nt(nloc, ld)) hipLaunchKernelGGL((qw_symw_kernel<O, true>), g, b, 0, st, Q, ld, W, pl.g, items, nitems, scal, prow, pcol);
        else hipLaunchKernelGGL((qw_symw_kernel<O, false>), g, b, 0, st, Q, ld, W, pl.g, items, nitems, scal, prow, pcol);
    }
    const int64_t n = (int64_t)6 * pl.g.T * O;
    hipLaunchKernelGGL((symw_colsum_kernel<O>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pl.g, strip_ptr, pcol, scal, csum);
}

void SymwProduct::sweep(int o, const double *Q, const double *W, const TcgScal *scal, int rank, hipStream_t st) {
    if (o > omax_) throw Error(-2, "symmetric window product: ensure() was not called for this rank");
    double *cs = csum_.p + (size_t)rank * csum_count(o);
    switch (o) {
        case 1: symw_sweep_o<1>(plan_, Q, ld_, W, scal, items_.p, strip_ptr_.p, prow_.p, pcol_.p, cs, nloc_, st); break;
        case 3: symw_sweep_o<3>(plan_, Q, ld_, W, scal, items_.p, strip_ptr_.p, prow_.p, pcol_.p, cs, nloc_, st); break;
        case 4: symw_sweep_o<4>(plan_, Q, ld_, W, scal, items_.p, strip_ptr_.p, prow_.p, pcol_.p, cs, nloc_, st); break;
        case 5: symw_sweep_o<5>(plan_, Q, ld_, W, scal, items_.p, strip_ptr_.p, prow_.p, pcol_.p, cs, nloc_, st); break;
        default: throw Error(-2, "symmetric window product is instantiated for o = 1, 3, 4, 5");
    }
    check_launch("qw_symw");
}

template <int O>
static void symw_reduce_o(int epi, const SymwGeom &g, const double *prow, const double *cs, int world, double alpha, const CamArgs &a, hipStream_t st) {
    const dim3 grid(qw_grid(a.nloc)), b(256);
    switch (epi) {
        case EPI_PLAIN: hipLaunchKernelGGL((symw_reduce_kernel<O, EPI_PLAIN>), grid, b, 0, st, g, prow, cs, world, alpha, a); break;
        case EPI_GRAD: hipLaunchKernelGGL((symw_reduce_kernel<O, EPI_GRAD>), grid, b, 0, st, g, prow, cs, world, alpha, a); break;
        case EPI_HESS: hipLaunchKernelGGL((symw_reduce_kernel<O, EPI_HESS>), grid, b, 0, st, g, prow, cs, world, alpha, a); break;
        case EPI_CERT:
            if constexpr (O == 1) { hipLaunchKernelGGL((symw_reduce_kernel<1, EPI_CERT>), grid, b, 0, st, g, prow, cs, world, alpha, a); break; }
            throw Error(-2, "certificate operator needs o == 1");
        default: throw Error(-2, "bad epilogue");
    }
}

void SymwProduct::reduce(int o, int epi, double alpha, const CamArgs &a, int world, hipStream_t st) {
    if (o > omax_ || world != world_) throw Error(-2, "symmetric window product: ensure() was not called for this rank / world");
    switch (o) {
        case 1: symw_reduce_o<1>(epi, plan_.g, prow_.p, csum_.p, world, alpha, a, st); break;
        case 3: symw_reduce_o<3>(epi, plan_.g, prow_.p, csum_.p, world, alpha, a, st); break;
        case 4: symw_reduce_o<4>(epi, plan_.g, prow_.p, csum_.p, world, alpha, a, st); break;
        case 5: symw_reduce_o<5>(epi, plan_.g, prow_.p, csum_.p, world, alpha, a, st); break;
        default: throw Error(-2, "symmetric window product is instantiated for o = 1, 3, 4, 5");
    }
    check_launch("symw_reduce");
}

}  // namespace xm
