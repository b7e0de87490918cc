// xm_kernels.hip — hand-written CDNA4 (gfx950) kernels of the XM Burer-Monteiro solve.
//
//   qw_dense_kernel   out = alpha * Q * W for the reference's dense Q (replaces cublasDgemm via DnMatDnMat,
//                     Dense/matmul.h:42-87, call sites trustregion.h:165,187,237,553, checkeig.h:182).
//                     HBM-bound (<= 2.5 flop/B): one wavefront per camera (3 rows of Q) streams its rows with
//                     16-byte coalesced loads, the W tile is staged once per workgroup in LDS (odd row pitch ->
//                     conflict-free ds_read_b128), fp64 FMA, wave-shuffle reduction, and the per-camera algebra that
//                     the reference runs as ~40 extra launches (trustregion.h:227-295, 186-194, 307-317) is fused
//                     into the epilogue.
//   qw_bsr3_kernel    same product from 3x3-block CSR: a 16-lane DPP row per camera row, one lane per stored block; the block
//                     stream and the gathered records of W pass through LDS with 16-byte loads, software-pipelined over
//                     windows of 16 blocks; same fused epilogues.
//   qw_sym_kernel     half-traffic product for symmetric dense Q (upper block triangle only) + sym_reduce_kernel.
//   cg_step_kernel    one launch per tCG iteration next to the Hessian product: step decision from the epilogue's partial
//                     sums, all vector updates; with several ranks also the replicated recurrence W+ = beta W - A+ that
//                     replaces the second all-gather of a distributed CG iteration.
//   flat kernels      the tCG vector updates (trustregion.h:605-644) with device-side alpha/beta/tau and branch
//                     logic, so the host never reads a scalar inside the inner loop.
//   per-camera        MGS-QR retraction (Dense/batchedQR.h:42-67), scale retraction (trustregion.h:19-24),
//                     certificate multipliers (closed form of checkeig.h:56-220).
//
// No MFMA on this path (no dense contraction with reuse); everything is float64 like the reference
// (Optimization/optimization.h:9).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "xm_common.h"
#include "xm_device.h"

namespace xm {

// one 16-byte non-temporal load (global_load_dwordx4 ... nt): streams past the Infinity Cache without displacing what is meant to stay there
__device__ __forceinline__ double2 nt_load16(const double2 *p) {
    typedef double d2v __attribute__((ext_vector_type(2)));
    const d2v v = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(p));
    return make_double2(v.x, v.y);
}

// ----------------------------------------------------------------------------------------------------------------
// dense Q*W
// ----------------------------------------------------------------------------------------------------------------
// Software-pipelined: while a wavefront multiplies tile t (Q fragment in registers, W tile in LDS buffer t&1) the loads
// of tile t+1 (its 3 rows of Q and the workgroup's share of the next W tile) are already in flight; one barrier per tile.
// SPLIT (separate instantiations, the default ones are untouched): the tile loop runs over a sub-range of the column tiles
// (CamArgs.range_mode) so that the product can be done in two launches around the all-gather of W.
template <int O, int EPI, int NSUB, bool SPLIT = false>
__global__ __launch_bounds__(256) void qw_dense_kernel(const double *__restrict__ Q, int64_t ld,
                                                        const double *__restrict__ W, double alpha, CamArgs a) {
    constexpr int OP = pitch_of(O);
    int role = EPI;   // EPI_AUTO: EPI_HESS or EPI_GRAD, from the phase of the device-driven outer iteration
    constexpr int TILE = NSUB * 128;                 // columns per tile
    constexpr int TILE2 = TILE * OP / 2;             // double2 elements of one W tile
    constexpr int NST = (TILE2 + 255) / 256;         // staging registers (double2) per thread
    __shared__ __attribute__((aligned(16))) double wt[2][TILE * OP];
    __shared__ double red[kQwWaves][3];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cam = blockIdx.x * kQwWaves + wave;   // wave-uniform (scalar): per-camera scalars load through the scalar cache
    const bool active = cam < a.nloc;
    const bool nt = (int)(blockIdx.x * kQwWaves) >= a.nt_cam0;   // workgroup-uniform load policy of these cameras' rows
    const double *q0 = Q + (size_t)(active ? cam : 0) * 3 * (size_t)ld + 2 * lane;
    const int ntiles_all = (int)((ld + TILE - 1) / TILE);
    // tile sequence: i-th tile of this launch (identity unless SPLIT)
    const int span = SPLIT ? (a.t_hi - a.t_lo) : 0;
    const int ntiles = !SPLIT ? ntiles_all : (a.range_mode == 1 ? span : ntiles_all - span);
    auto tile_at = [&](int i) -> int {
        if (!SPLIT) return a.rev ? ntiles_all - 1 - i : i;
        return (a.range_mode == 1) ? a.t_lo + i : ((i < a.t_lo) ? i : i + span);
    };

    double acc[3][O];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < O; ++k) acc[r][k] = 0.0;
    if (SPLIT) {
        if (a.addend != nullptr && active && lane == 0) {   // raw sums of the tiles an earlier launch did
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int k = 0; k < O; ++k) acc[r][k] = a.addend[((size_t)cam * 3 + r) * OP + k];
        }
    }

    EpiOps eops;
    double2 qn[NSUB][3];   // Q fragment of the NEXT tile
    double2 ws[NST];       // this thread's share of the NEXT W tile
    auto load_q = [&](int t, auto ntag) {
        constexpr bool NT = decltype(ntag)::value;
        const int64_t c0 = (int64_t)t * TILE;
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
            const int64_t c = c0 + u * 128;
            if (active && c < ld) {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const double2 *qp = reinterpret_cast<const double2 *>(q0 + (size_t)r * ld + c);
                    // a Q larger than the 256 MB Infinity Cache: the rows of the first nt_cam0 cameras (a cache-sized prefix) keep the default
                    // policy and stay resident across products, the rest is a pure stream -- non-temporal loads keep it from evicting the
                    // prefix (a 349 MB matrix with the default policy everywhere: 78.9 us, the stream evicts its own head); a Q that fits
                    // stays cacheable as a whole (nt_cam0 = nloc).  The policy is a compile-time property of the LOOP (two copies of it
                    // under one workgroup-uniform branch): a load whose hint is chosen by a run-time branch next to it is merged by the
                    // compiler with its twin into one plain load -- the hint is metadata and does not survive the merge.
                    if (NT) qn[u][r] = nt_load16(qp);
                    else qn[u][r] = *qp;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 3; ++r) qn[u][r] = make_double2(0.0, 0.0);
            }
        }
    };
    auto load_w = [&](int t) {
        const int64_t c0 = (int64_t)t * TILE;
        const int64_t cols = (ld - c0 < TILE) ? (ld - c0) : TILE;
        const int n2 = (int)(cols * OP / 2);
        const double2 *src = reinterpret_cast<const double2 *>(W + (size_t)c0 * OP);
#pragma unroll
        for (int j = 0; j < NST; ++j) {
            const int idx = threadIdx.x + j * 256;
            ws[j] = (idx < n2) ? src[idx] : make_double2(0.0, 0.0);
        }
    };
    auto store_w = [&](int buf) {
        double2 *dst = reinterpret_cast<double2 *>(wt[buf]);
#pragma unroll
        for (int j = 0; j < NST; ++j) {
            const int idx = threadIdx.x + j * 256;
            if (idx < TILE2) dst[idx] = ws[j];
        }
    };

    if (SPLIT) {
        if (ntiles <= 0) {   // nothing to multiply in this launch (uniform): straight to the tail
            epi_prefetch<O, EPI>(eops, cam, lane, active, a);
            qw_finish<O, EPI, 64, kQwWaves>(cam, lane, wave, active, acc, alpha, a, eops, red);
            return;
        }
    }
    auto stream = [&](auto ntag) -> bool {
    load_q(tile_at(0), ntag);
    load_w(tile_at(0));
    if (EPI == EPI_HESS) {
        // tCG already terminated: the enqueued-ahead launch becomes a no-op.  Checked only after the first tile's loads are
        // in flight, so that a live launch does not start with an exposed dependent load (uniform over the grid).
        if (a.scal->status != 0) return false;
    }
    if (EPI == EPI_AUTO) {   // device-driven outer iteration: Hessian product of the running tCG, or cost / gradient at the candidate point
        const int ph = a.scal->phase;
        if (ph >= PH_STOP) return false;
        role = (ph == PH_CAND) ? EPI_GRAD : EPI_HESS;
    }
    store_w(0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        double2 q[NSUB][3];
#pragma unroll
        for (int u = 0; u < NSUB; ++u)
#pragma unroll
            for (int r = 0; r < 3; ++r) q[u][r] = qn[u][r];
        const bool more = (t + 1 < ntiles);
        if (more) {  // uniform
            load_q(tile_at(t + 1), ntag);
            load_w(tile_at(t + 1));
        } else if (EPI != EPI_AUTO) {
            epi_prefetch<O, EPI>(eops, cam, lane, active, a, (int)EPI);
        }
        const double2 *wbase = reinterpret_cast<const double2 *>(wt[t & 1]);
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
            const double2 *wp = wbase + (size_t)(u * 64 + lane) * OP;
            double wv[2 * OP];
#pragma unroll
            for (int j = 0; j < OP; ++j) {
                const double2 tt = wp[j];
                wv[2 * j] = tt.x;
                wv[2 * j + 1] = tt.y;
            }
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int k = 0; k < O; ++k) acc[r][k] += q[u][r].x * wv[k] + q[u][r].y * wv[OP + k];
        }
        if (more) store_w((t + 1) & 1);
        __syncthreads();
    }
    return true;
    };
    const bool alive = nt ? stream(std::true_type{}) : stream(std::false_type{});
    if (!alive) return;
    // (EPI_AUTO asks for its epilogue operands only here: requested inside the last tile, the two roles' operand sets cost the loop 60 VGPRs and a
    // wavefront per SIMD -- 36.9 against 32.5 us per product at Venice size)
    if (EPI == EPI_AUTO) epi_prefetch<O, EPI>(eops, cam, lane, active, a, role);
    qw_finish<O, EPI, 64, kQwWaves>(cam, lane, wave, active, acc, alpha, a, eops, red, (EPI == EPI_AUTO) ? role : (int)EPI);
}

// Column-split variant for SMALL ROW STRIPS (a rank of a multi-GPU run: Venice-1778 over 8 GPUs leaves 223 cameras = 56 workgroups for
// 256 CUs, each walking all 21 column tiles serially: 18.7 us for 28 MB, profiles/r03_kbench_multi.txt).  grid = (camera groups, KS):
// workgroup (b, y) multiplies the tiles of slice y only and stores its wavefronts' raw 3 x O sums; an arrival counter per camera group
// tells the last slice to arrive, which adds the KS partial results IN SLICE ORDER (fixed, so the result does not depend on who
// finishes) and runs the fused epilogue.  Same pipelined body as qw_dense_kernel.
template <int O, int EPI>
__global__ __launch_bounds__(256) void qw_dense_ks_kernel(const double *__restrict__ Q, int64_t ld, const double *__restrict__ W, double alpha, CamArgs a) {
    constexpr int OP = pitch_of(O);
    constexpr int NSUB = 2, TILE = NSUB * 128, TILE2 = TILE * OP / 2, NST = (TILE2 + 255) / 256;
    __shared__ __attribute__((aligned(16))) double wt[2][TILE * OP];
    __shared__ double red[kQwWaves][3];
    __shared__ int s_last;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cam = blockIdx.x * kQwWaves + wave;
    const bool active = cam < a.nloc;
    const double *q0 = Q + (size_t)(active ? cam : 0) * 3 * (size_t)ld + 2 * lane;
    const int ntiles_all = (int)((ld + TILE - 1) / TILE);
    const int KS = a.ks, y = blockIdx.y;
    const int t_lo = (int)((int64_t)ntiles_all * y / KS), t_hi = (int)((int64_t)ntiles_all * (y + 1) / KS);
    const int ntiles = t_hi - t_lo;
    if (EPI == EPI_HESS) {
        if (a.scal->status != 0) return;
    }
    double acc[3][O];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < O; ++k) acc[r][k] = 0.0;
    double2 qn[NSUB][3];
    double2 ws[NST];
    auto load_q = [&](int t) {
        const int64_t c0 = (int64_t)t * TILE;
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
            const int64_t c = c0 + u * 128;
            if (active && c < ld) {
#pragma unroll
                for (int r = 0; r < 3; ++r) qn[u][r] = *reinterpret_cast<const double2 *>(q0 + (size_t)r * ld + c);
            } else {
#pragma unroll
                for (int r = 0; r < 3; ++r) qn[u][r] = make_double2(0.0, 0.0);
            }
        }
    };
    auto load_w = [&](int t) {
        const int64_t c0 = (int64_t)t * TILE;
        const int64_t cols = (ld - c0 < TILE) ? (ld - c0) : TILE;
        const int n2 = (int)(cols * OP / 2);
        const double2 *src = reinterpret_cast<const double2 *>(W + (size_t)c0 * OP);
#pragma unroll
        for (int j = 0; j < NST; ++j) {
            const int idx = threadIdx.x + j * 256;
            ws[j] = (idx < n2) ? src[idx] : make_double2(0.0, 0.0);
        }
    };
    auto store_w = [&](int buf) {
        double2 *dst = reinterpret_cast<double2 *>(wt[buf]);
#pragma unroll
        for (int j = 0; j < NST; ++j) {
            const int idx = threadIdx.x + j * 256;
            if (idx < TILE2) dst[idx] = ws[j];
        }
    };
    if (ntiles > 0) {
        load_q(t_lo);
        load_w(t_lo);
        store_w(0);
        __syncthreads();
        for (int t = 0; t < ntiles; ++t) {
            double2 q[NSUB][3];
#pragma unroll
            for (int u = 0; u < NSUB; ++u)
#pragma unroll
                for (int r = 0; r < 3; ++r) q[u][r] = qn[u][r];
            const bool more = (t + 1 < ntiles);
            if (more) { load_q(t_lo + t + 1); load_w(t_lo + t + 1); }
            const double2 *wbase = reinterpret_cast<const double2 *>(wt[t & 1]);
#pragma unroll
            for (int u = 0; u < NSUB; ++u) {
                const double2 *wp = wbase + (size_t)(u * 64 + lane) * OP;
                double wv[2 * OP];
#pragma unroll
                for (int j = 0; j < OP; ++j) {
                    const double2 tt = wp[j];
                    wv[2 * j] = tt.x;
                    wv[2 * j + 1] = tt.y;
                }
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int k = 0; k < O; ++k) acc[r][k] += q[u][r].x * wv[k] + q[u][r].y * wv[OP + k];
            }
            if (more) store_w((t + 1) & 1);
            __syncthreads();
        }
    }
    // raw sums of this slice: column k of the camera's block in lane k
    Col3 h;
    h.v[0] = h.v[1] = h.v[2] = 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < O; ++k) {
            const double t = wave_sum(acc[r][k]);
            if (lane == k) h.v[r] = t;
        }
    // hand-off without cache-wide fences (a release fence writes back the whole L2 of the XCD and the acquire invalidates it: measured
    // 39 us instead of 22 for the N = 4 strip): the partial sums are agent-scope (write-through) stores, drained with s_waitcnt before
    // the arrival is counted, and the finisher reads them with agent-scope loads (MI355X_MICROARCH: "sc1 stores AND sc1 loads")
    if (active && lane < O) {
        double *ps = a.ksum + ((size_t)y * a.nloc + cam) * 3 * OP + lane;
#pragma unroll
        for (int r = 0; r < 3; ++r) __hip_atomic_store(ps + r * OP, h.v[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int old = __hip_atomic_fetch_add(a.kcount + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (old + 1 == (unsigned int)KS);
        if (s_last) __hip_atomic_store(a.kcount + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
    __syncthreads();
    if (!s_last) return;
    EpiOps eops;
    epi_prefetch<O, EPI>(eops, cam, lane, active, a);
    h.v[0] = h.v[1] = h.v[2] = 0.0;
    if (active && lane < O) {
        for (int s = 0; s < KS; ++s) {   // slice order: the sum does not depend on which slice arrived last
            const double *ps = a.ksum + ((size_t)s * a.nloc + cam) * 3 * OP + lane;
#pragma unroll
            for (int r = 0; r < 3; ++r) h.v[r] += __hip_atomic_load(ps + r * OP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) h.v[r] *= alpha;
    }
    // the partial sums of the epilogue are indexed by the camera group, as in the unsplit kernel (gridDim.x = camera groups)
    qw_tail<O, EPI, 64, kQwWaves>(cam, lane, wave, active, h, a, eops, red);
}

// ----------------------------------------------------------------------------------------------------------------
// Half-traffic product for SYMMETRIC dense Q (single GPU, o <= 5): only the upper block triangle is read, every Q fragment is used
// twice.  A WORKGROUP owns a strip of 256 columns and 4 K consecutive steps of it; each of its four wavefronts (lane: 2 + 2 adjacent
// columns, W of its columns in registers for the whole sweep) walks K steps of two cameras (6 rows x 256 columns = 12 KB per step):
//   column direction  y_cols += Q_step^T w_rows : per-lane accumulators that live in registers for the whole chunk; the four
//                     wavefronts' sums are added in LDS (wavefront order 0..3: fixed) and written ONCE per workgroup;
//   row direction     y_rows  = Q_step w_cols   : 6 * o per-lane partial sums per step, summed over the 64 lanes through LDS
//                     (transposed write, 16-lane DPP row sums) and written as 6 * o doubles per step.
// Element (r, c) of step j (rows [6j, 6j+6)): used both ways when c >= 6j + 6, in the row direction only when 6j <= c < 6j + 6
// (the 6 x 6 diagonal block is read in full), not at all when c < 6j (its mirror image serves it).
// Round 6 (profiles/r06_kbench_symv.txt; Venice size, o = 3 / 4, pair of launches, us): workgroup on four strips with one chunk each and a
// select behind every load 29.8 / 34.0 -> workgroup on one strip, column sums combined in LDS (Pcol / 4: ~35 instead of ~100 partial records
// per camera), loads without a select and a peeled loop so that the next step's twelve requests stay in flight while the current step is
// multiplied (s_waitcnt vmcnt(12), not 0) 28.1 / 32.5 -> alternating sweep direction (rev) 27.4 / 31.9 -> K = 6 (one residency round of
// ~420 workgroups) 26.5 / 29.6.  The per-wavefront timestamps (TRACE) say where the time of the launch goes: all wavefronts start within
// 1 us, the first step completes after ~5-6 us (every wavefront asks for 20 KB at once: 25 MB at the ~7 TB/s the fabric delivers), every
// further step 2.1-2.5 us (= 7 TB/s over all wavefronts: the loop runs at the chip's saturation), the median wavefront ends at 20 us, the last at 24.
// The loads carry no select: a row past the end re-reads the last row and meets w_row = 0 in the column direction (its row sums land in
// rows of Prow nobody reads), the absent second half of the last strip re-reads the first half and meets w_col = 0.
// TRACE (micro-benchmark only): 100 MHz timestamps per wavefront -- entry, after the status word, after every step, end.
// ----------------------------------------------------------------------------------------------------------------
constexpr int kSvStrip = 256;
constexpr int kSvTraceSlots = 24;
template <int O, bool TRACE = false>
__global__ __launch_bounds__(256) void qw_symv_kernel(const double *__restrict__ Q, int64_t ld, const double *__restrict__ W, int nloc, int Kc, int Kf, int ysplit, int nt_step0,
                                                        const TcgScal *__restrict__ scal, double *__restrict__ Prow,
                                                        double *__restrict__ Pcol, unsigned long long *__restrict__ trace, int rev, int by_phase) {
    constexpr int OP = pitch_of(O), V = 6 * O;
    __shared__ __attribute__((aligned(16))) double lds[4][V * 64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // The live (strip, group) pairs form a staircase -- strip s has ~ (s + 1) * 42.7 / (4 K) groups -- and block b runs on XCD b mod 8: a grid of
    // strips x groups is half empty and its live blocks land on the XCDs unevenly (K = 11 at 2 560 cameras: 180 ... 272 live wavefronts per XCD,
    // one XCD beyond its 64 resident workgroups, a second dispatch round: 66.7 us instead of 48).  FOLDED grid: row y holds strip y and, behind
    // it, strip S - 1 - y -- every row has about the same number of live blocks, (nearly) every block of the grid is live, and consecutive
    // blocks are consecutive XCDs.
    const int nsteps = (nloc + 1) >> 1, nrows = 3 * nloc;
    const int nstrips = (int)((ld + kSvStrip - 1) / kSvStrip);
    const int K = ((int)blockIdx.y >= ysplit) ? Kf : Kc;         // the rows dispatched last are cut finer: they are the launch's tail
    int s = blockIdx.y, sc = blockIdx.x;
    {
        int jA = (int)(((int64_t)s * kSvStrip + kSvStrip + 5) / 6);
        if (jA > nsteps) jA = nsteps;
        const int nA = (jA + 4 * K - 1) / (4 * K);
        if (sc >= nA) {
            if (nstrips - 1 - s == s) return;                    // the middle strip of an odd count has no partner (uniform over the workgroup)
            s = nstrips - 1 - s; sc -= nA;
        }
    }
    unsigned long long *tr = nullptr;
    if constexpr (TRACE) {
        tr = trace + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * kSvTraceSlots;
        if (lane == 0) {
            unsigned int hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            tr[0] = wall_clock64();
            tr[kSvTraceSlots - 1] = ((unsigned long long)xcc << 32) | hw;
        }
    }
    const int64_t c0 = (int64_t)s * kSvStrip;
    int jend = (int)((c0 + kSvStrip + 5) / 6);                   // steps whose rows start above the end of the strip
    if (jend > nsteps) jend = nsteps;
    if (sc * 4 * K >= jend) return;                              // uniform over the workgroup
    const int jb = (sc * 4 + wave) * K;                          // this wavefront's chunk [jb, je): may be empty at the foot of the strip
    const int je = (jb + K < jend) ? jb + K : jend;
    const int jfull = (int)(c0 / 6);                             // steps j < jfull lie entirely above the diagonal: no masks
    const bool half1 = c0 + 128 < ld;                            // ld is a multiple of 128: the strip may end after its first half
    const int64_t R = (int64_t)6 * nsteps;
    double *L = lds[wave];
    const int64_t cA = c0 + 2 * lane, cB = half1 ? cA + 128 : cA;

    double wc[2][2][O], ca[2][2][O];

    // chunks that start at a step >= nt_step0 stream non-temporally (a matrix beyond the Infinity Cache keeps its top rows resident:
    // symv_nt_step0).  The policy is a compile-time property of the sweep loop (two copies under one wave-uniform branch, see qw_dense_kernel).
    // W of the step's six rows travels with the step's Q: lane l requests element l of the 6 * OP contiguous doubles (one more request behind
    // the twelve), and the multiply reads w_row out of that register with v_readlane.  Scalar loads at the point of use (round 5) were waited
    // for one by one inside the step -- up to six exposed round trips to L2 per step at o = 4, where a row's four values are a load of their own.
    const int64_t wlim = (int64_t)nrows * OP - 1;
    auto load_q = [&](int j, double2 (&q)[6][2], double &wl, auto ntag) __attribute__((always_inline)) {
        constexpr bool NT = decltype(ntag)::value;
        const int64_t r0 = (int64_t)6 * j;
        {
            const int64_t wi = r0 * OP + (lane < 6 * OP ? lane : 0);
            wl = W[wi < wlim ? wi : wlim];
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int64_t rr = (r0 + r < nrows) ? r0 + r : nrows - 1;   // wave-uniform clamp (odd camera count: three rows of the last step)
            const double *row = Q + (size_t)rr * (size_t)ld;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const double2 *qp = reinterpret_cast<const double2 *>(row + (h ? cB : cA));
                if (NT) q[r][h] = nt_load16(qp);
                else q[r][h] = *qp;
            }
        }
    };
    auto step = [&](int j, const double2 (&q)[6][2], const double wl, auto masked) __attribute__((always_inline)) {
        constexpr bool MASK = decltype(masked)::value;
        const int64_t r0 = (int64_t)6 * j;
        double mr[2] = {1.0, 1.0}, mc[2] = {1.0, 1.0};
        if constexpr (MASK) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {   // both columns of a pair fall on the same side (all bounds are even)
                mr[h] = (cA + 128 * h >= r0) ? 1.0 : 0.0;
                mc[h] = (cA + 128 * h >= r0 + 6) ? 1.0 : 0.0;
            }
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            double wr[O];
            const bool ok = r0 + r < nrows;                              // wave-uniform: scalar select
#pragma unroll
            for (int k = 0; k < O; ++k) {
                const int lo = __builtin_amdgcn_readlane(__double2loint(wl), r * OP + k), hi = __builtin_amdgcn_readlane(__double2hiint(wl), r * OP + k);
                wr[k] = ok ? __hiloint2double(hi, lo) : 0.0;
            }
            double qr[2][2], qc[2][2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                qr[h][0] = MASK ? q[r][h].x * mr[h] : q[r][h].x; qr[h][1] = MASK ? q[r][h].y * mr[h] : q[r][h].y;
                qc[h][0] = MASK ? q[r][h].x * mc[h] : q[r][h].x; qc[h][1] = MASK ? q[r][h].y * mc[h] : q[r][h].y;
            }
#pragma unroll
            for (int k = 0; k < O; ++k) {
                double t = qr[0][0] * wc[0][0][k];
                t = fma(qr[0][1], wc[0][1][k], t);
                t = fma(qr[1][0], wc[1][0][k], t);
                t = fma(qr[1][1], wc[1][1][k], t);
                L[(r * O + k) * 64 + lane] = t;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 2; ++e) ca[h][e][k] = fma(qc[h][e], wr[k], ca[h][e][k]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // 64 addends per value: a 16-lane row takes value v = 4 i + (lane / 16), each lane four addends, DPP row sum
        const int g = lane >> 4, jl = lane & 15;
#pragma unroll
        for (int v0 = 0; v0 < V; v0 += 4) {
            const int v = v0 + g;
            double t = 0.0;
            if (v < V) {
                const double2 a = *reinterpret_cast<const double2 *>(L + v * 64 + 4 * jl), b = *reinterpret_cast<const double2 *>(L + v * 64 + 4 * jl + 2);
                t = (a.x + a.y) + (b.x + b.y);
            }
            t = group_sum<16>(t);
            if (jl == 0 && v < V) Prow[((size_t)s * (size_t)R + (size_t)r0) * O + v] = t;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    int nrun = 0;
    auto run = [&](int j, const double2 (&q)[6][2], const double wl) __attribute__((always_inline)) {
        if (j < jfull) step(j, q, wl, std::false_type{});
        else step(j, q, wl, std::true_type{});
        if constexpr (TRACE) {
            if (lane == 0 && 2 + nrun < kSvTraceSlots - 3) tr[2 + nrun] = wall_clock64();
            ++nrun;
        }
    };

    double2 qA[6][2], qB[6][2];
    double wA = 0.0, wB = 0.0;
    auto sweep = [&](auto ntag) __attribute__((always_inline)) -> bool {
    if (jb < je) load_q(rev ? je - 1 : jb, qA, wA, ntag);                // wave-uniform
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int k = 0; k < O; ++k) {
                const double t = W[(size_t)((h ? cB : cA) + e) * OP + k];   // unconditional request (a predicated one is a branch per element)
                wc[h][e][k] = (h == 0 || half1) ? t : 0.0;
                ca[h][e][k] = 0.0;
            }
    // the tCG's status word (written by the previous launch on another XCD: an L2 miss) is looked at only now, with the columns of W and the
    // first step of Q already requested: one round trip at the head of every wavefront instead of two
    if (scal != nullptr) {   // by_phase (device-driven outer iteration): live in the phases PH_TCG and PH_CAND, whatever the tCG's status says
        if (by_phase ? (scal->phase >= PH_STOP) : (scal->status != 0)) return false;
    }
    if constexpr (TRACE) { if (lane == 0) tr[1] = wall_clock64(); }
    // rev: the chunk is walked bottom-up (position i <-> step je - 1 - i).  Launches alternate the direction, so that a launch starts with
    // the steps the previous one ended with: they are still in this XCD's L2 (4 MB; block b runs on XCD b mod 8 in every launch)
    auto at = [&](int i) __attribute__((always_inline)) { return rev ? je - 1 - i : jb + i; };
    const int cnt = je - jb;
    if (cnt > 0) {
        int i = 0;
        while (i + 2 < cnt) {          // two more steps follow: both requests below are unconditional
            load_q(at(i + 1), qB, wB, ntag);
            run(at(i), qA, wA);
            load_q(at(i + 2), qA, wA, ntag);
            run(at(i + 1), qB, wB);
            i += 2;
        }
        if (i + 1 < cnt) {
            load_q(at(i + 1), qB, wB, ntag);
            run(at(i), qA, wA);
            run(at(i + 1), qB, wB);
        } else {
            run(at(i), qA, wA);
        }
    }
    return true;
    };
    const bool alive = (jb >= nt_step0) ? sweep(std::true_type{}) : sweep(std::false_type{});
    if (!alive) return;
    if constexpr (TRACE) { if (lane == 0) tr[kSvTraceSlots - 3] = wall_clock64(); }

    // column sums of the four chunks, added in wavefront order; wavefront h writes the h-th half of the strip (2 O contiguous doubles per lane)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int k = 0; k < O; ++k) L[((h * 2 + e) * O + k) * 64 + lane] = ca[h][e][k];
    __syncthreads();
    if (wave < 2 && (wave == 0 || half1)) {
        double *pc = Pcol + ((size_t)sc * (size_t)ld + (size_t)(cA + 128 * wave)) * O;
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int k = 0; k < O; ++k) {
                const int idx = ((wave * 2 + e) * O + k) * 64 + lane;
                pc[e * O + k] = ((lds[0][idx] + lds[1][idx]) + lds[2][idx]) + lds[3][idx];
            }
    }
    if constexpr (TRACE) { if (lane == 0) tr[kSvTraceSlots - 2] = wall_clock64(); }
}

// second half: y_cam = sum_{strips s >= s_lo} Prow[s][rows of cam] + sum_{groups above} Pcol[group][rows of cam] (fixed order), fused epilogue.
// One wavefront per camera.  A record is 3 * O contiguous doubles: SIXTEEN lanes read one record (lane e its element e), so that a load
// instruction of the wavefront covers four records = four short contiguous runs -- with a lane per record (round 5) every instruction touched
// 64 different cache lines, 9 .. 12 instructions per round, and the launch was bound by the address rate of the vector memory pipeline
// (7.7 us in the Venice solve for 13 MB).  Round t of a block of 64 records: lane group g = lane / 16 reads record 4 t + g; all sixteen
// requests of the first block go out before the tCG's status word is looked at (the launch is latency-bound: status word, partial sums and
// epilogue operands were all written by earlier launches on other XCDs -- one round trip for all of them).  Loads are unconditional (a lane
// without a record re-reads a valid one and selects zero): a predicated load becomes a branch the compiler drains the queue for.  Order of
// the sum: per lane group the rounds in sequence, then (g0 + g1) + (g2 + g3) -- fixed.
template <int O, int EPI>
__global__ __launch_bounds__(256) void symv_reduce_kernel(const double *__restrict__ Prow, const double *__restrict__ Pcol, int64_t ld,
                                                           int nstrips, int Kc, int Kf, int ysplit, double alpha, CamArgs a) {
    int role = (EPI == EPI_AUTO) ? (int)EPI_HESS : EPI;   // (see qw_dense_kernel)
    constexpr int NE = 3 * O;          // elements of a record (<= 15)
    static_assert(NE <= 16, "a record must fit a 16-lane group");
    __shared__ double red[kQwWaves][3];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cam = blockIdx.x * kQwWaves + wave;
    const bool active = cam < a.nloc;
    EpiOps eops;
    // EPI_AUTO: the Hessian role's operands are requested up front like everybody's (most launches of a solve are Hessian products); a launch
    // that finds itself in the gradient role asks again below
    epi_prefetch<O, EPI == EPI_AUTO ? EPI_HESS : EPI>(eops, cam, lane, active, a);
    const int e = lane & 15, g = lane >> 4;
    const bool eok = e < NE;
    const int camc = active ? cam : 0;
    const int64_t R = (int64_t)6 * ((a.nloc + 1) >> 1);
    const int s_lo = (6 * (camc >> 1)) / kSvStrip;
    const int nrow = nstrips - s_lo;
    // column-sum records: the three columns of a camera lie in one step, but they may lie in two STRIPS (256 is no multiple of 3) whose grid
    // rows are cut differently (Kc / Kf), so the count belongs to the column -- lane e owns column 3 cam + e / O
    auto ncol_of = [&](int c) -> int {
        const int sc_ = c / kSvStrip, row = min(sc_, nstrips - 1 - sc_);   // grid row of the strip that owns the column (folded grid of the sweep)
        const int K = (row >= ysplit) ? Kf : Kc;                          // steps per column-sum record there
        return (c >= 6) ? (c - 6) / (6 * K) + 1 : 0;
    };
    const int ncol_l = ncol_of(3 * camc + (eok ? e / O : 0));
    const int ncol = max(ncol_of(3 * camc), ncol_of(3 * camc + 2));   // wave-uniform (the count is monotone in the column)
    const int nrec = nrow + ncol;                                     // nrow >= 1: always a valid record
    const int tot = active ? nrec : 0, tot_l = active ? nrow + ncol_l : 0;
    const size_t eo = (size_t)camc * 3 * O + (eok ? e : 0);
    auto fetch = [&](int i) -> double {
        const int ic = min(i, nrec - 1);
        const double *p = (ic < nrow) ? Prow + (size_t)(s_lo + ic) * (size_t)R * O : Pcol + (size_t)(ic - nrow) * (size_t)ld * O;
        return p[eo];
    };
    double v[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = fetch(4 * t + g);
    if (EPI == EPI_HESS) {
        if (a.scal->status != 0) return;
    }
    if (EPI == EPI_AUTO) {
        const int ph = a.scal->phase;
        if (ph >= PH_STOP) return;
        if (ph == PH_CAND) { role = EPI_GRAD; epi_prefetch<O, EPI>(eops, cam, lane, active, a, (EPI == EPI_AUTO) ? role : (int)EPI); }
    }
    double acc = 0.0;
#pragma unroll
    for (int t = 0; t < 16; ++t) acc += (eok && 4 * t + g < tot_l) ? v[t] : 0.0;
    for (int base = 64; base < tot; base += 64) {                    // wave-uniform trip count
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = fetch(base + 4 * t + g);
#pragma unroll
        for (int t = 0; t < 16; ++t) acc += (eok && base + 4 * t + g < tot_l) ? v[t] : 0.0;
    }
    acc += __shfl_xor(acc, 16);        // g0 + g1 | g2 + g3 (a + b == b + a bit for bit: both lanes of a pair hold the same sum)
    acc += __shfl_xor(acc, 32);
    // element r * O + k of the sum sits in lane r * O + k (of every group): column k -> lane k, as the epilogues expect
    Col3 h;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double t = __shfl(acc, (lane < O) ? r * O + lane : 0);
        h.v[r] = (active && lane < O) ? alpha * t : 0.0;
    }
    qw_tail<O, EPI, 64, kQwWaves>(cam, lane, wave, active, h, a, eops, red, (EPI == EPI_AUTO) ? role : (int)EPI);
}

// max |Q[r][c] - Q[c][r]| and max |Q| over the device layout (decides whether the symmetric path may be used).  Tile pairs (ti <= tj)
// of 64 x 64: the upper tile goes through LDS and is compared, transposed, with the coalesced read of its mirror image (a
// row-against-column sweep read 64 cache lines per instruction: 23 ms for 13.5 GB; this one streams the matrix once).  A NaN anywhere
// makes the asymmetry NaN (sticky flag: a max written with comparisons alone drops a NaN at the next finite element).
__global__ __launch_bounds__(256) void asym_kernel(const double *__restrict__ Q, int64_t ld, int64_t m, double *out /* [2*grid] */) {
    __shared__ double T[64][65];
    __shared__ double sh[3][4];
    const int64_t nt = (m + 63) / 64;
    double da = 0.0, mx = 0.0, bad = 0.0;
    for (int64_t p = blockIdx.x; p < nt * nt; p += gridDim.x) {
        const int64_t ti = p / nt, tj = p - ti * nt;
        if (ti > tj) continue;   // block-uniform
#pragma unroll 4
        for (int e = threadIdx.x; e < 4096; e += 256) {
            const int r = e >> 6, c = e & 63;
            const int64_t gr = ti * 64 + r, gc = tj * 64 + c;
            const double v = (gr < m && gc < m) ? Q[gr * ld + gc] : 0.0;
            T[r][c] = v;
            mx = fmax(mx, fabs(v));
            if (v != v) bad = 1.0;
        }
        __syncthreads();
#pragma unroll 4
        for (int e = threadIdx.x; e < 4096; e += 256) {
            const int r = e >> 6, c = e & 63;
            const int64_t gr = tj * 64 + r, gc = ti * 64 + c;
            const double v = (gr < m && gc < m) ? Q[gr * ld + gc] : 0.0;
            da = fmax(da, fabs(v - T[c][r]));
            mx = fmax(mx, fabs(v));
            if (v != v) bad = 1.0;
        }
        __syncthreads();
    }
    for (int off = 32; off >= 1; off >>= 1) {
        da = fmax(da, __shfl_xor(da, off, 64)); mx = fmax(mx, __shfl_xor(mx, off, 64)); bad = fmax(bad, __shfl_xor(bad, off, 64));
    }
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = da; sh[1][threadIdx.x >> 6] = mx; sh[2][threadIdx.x >> 6] = bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double b4 = fmax(fmax(sh[2][0], sh[2][1]), fmax(sh[2][2], sh[2][3]));
        const double t = fmax(fmax(sh[0][0], sh[0][1]), fmax(sh[0][2], sh[0][3]));
        out[blockIdx.x] = (b4 > 0.0) ? __longlong_as_double(0x7ff8000000000000LL) : t;
        out[gridDim.x + blockIdx.x] = fmax(fmax(sh[1][0], sh[1][1]), fmax(sh[1][2], sh[1][3]));
    }
}

// ----------------------------------------------------------------------------------------------------------------
// 3x3-block CSR Q*W: ONE 16-LANE GROUP (a DPP row) PER CAMERA ROW, ONE LANE PER STORED BLOCK.  A wavefront carries four rows (16 per
// workgroup); a lane fetches its column index, then (independently, all in flight together) its 72-byte block and the 3 x O rows of W
// it multiplies and does the 9*O FMAs itself.  The blocks of a row are contiguous and pass through LDS so that the global loads are
// perfectly coalesced (VAR 1; direct 72-byte-strided loads are 1.3x slower).  The row sums and the whole fused epilogue run inside the
// 16-lane row with DPP steps.  What bounds it (round 6: per-wavefront timestamps, profiles/r06_kbench_bsr_trace.txt) is the CU's load
// path -- ~45 cycles per 64-lane 16-byte load, the windows' loads queue whatever order they are issued in -- together with instruction
// issue; rounds 1-5 had read it as a chain of dependent round trips (row pointers -> column indices -> gathered records).
// ----------------------------------------------------------------------------------------------------------------

#ifdef XM_BSR_TRACE
// experiment builds only (make EXTRA=-DXM_BSR_TRACE OBJDIR=obj_x LIBDIR=lib_x; scripts/kbench_bsr_trace.py): 100 MHz timestamps per wavefront --
// entry, row record in, the data of window 0 / 1 / >= 2 in, end of the loop, end.  What they showed at 13 682 cameras (profiles/r06_kbench_bsr_trace.txt):
// a wavefront lives ~10 us = 0.4 (row pointers) + 3.0 + 2.6 + 2.0 (one window's loads after the other) + 0.9 + 0.8, and requesting two windows
// at once moves the arrivals without shortening their sum: the windows' loads queue in the CU's load path (~45 cycles per 64-lane 16-byte load)
__device__ unsigned long long *g_bsr_trace = nullptr;
#define BSR_TS(slot_) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (g_bsr_trace && (threadIdx.x & 63) == 0) g_bsr_trace[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (slot_)] = wall_clock64(); } while (0)
#else
#define BSR_TS(slot_) do { } while (0)
#endif
template <int O, int EPI, int VAR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((VAR == 2 && O == 3) ? 4 : 1, 8))) void qw_bsr3_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
                                                       const double *__restrict__ blocks, const double *__restrict__ W,
                                                       const int4 *__restrict__ rowinfo, double alpha, CamArgs a) {
    constexpr int OP = pitch_of(O);
    int role = EPI;   // (see qw_dense_kernel)
    constexpr int REC = 3 * OP;                                   // doubles of one camera's rows of W
    constexpr int SW = (VAR == 2 && REC > 9) ? REC : 9;           // LDS doubles per lane (blocks and W share the window)
    __shared__ double red[kBsrRows][3];
    __shared__ double stage[(VAR >= 1) ? kBsrRows * 16 * SW : 1]; // VAR >= 1: blocks pass through LDS (coalesced loads)
    const int gl = threadIdx.x & 15;          // lane inside the group
    const int slot = threadIdx.x >> 4;        // group inside the workgroup (0..15)
    // rowinfo (may be nullptr: rows in camera order): the workgroup's 16 rows ORDERED BY THEIR NUMBER OF WINDOWS, longest first (bsr_build_rowinfo) --
    // entry = {first block (64 bit), row length, camera}.  The four groups of a wavefront loop together, so in camera order a wavefront runs as
    // many windows as its longest row needs: with ~31 +- 5 blocks per row 82 % of the wavefronts ran three windows where 35 % of the rows need
    // them (13.0 us against 11.2 for the same number of blocks in rows of equal length).  One 16-byte load instead of the two row pointers.
    const int rslot = blockIdx.x * kBsrRows + slot;
    const bool active = rslot < a.nloc;
    int cam = rslot;
    double acc[3][O];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < O; ++k) acc[r][k] = 0.0;
    int64_t b0 = 0, b1 = 0;
    BSR_TS(0);
    if (rowinfo) {
        if (active) {
            const int4 ri = rowinfo[rslot];
            b0 = (int64_t)(((unsigned long long)(unsigned)ri.y << 32) | (unsigned long long)(unsigned)ri.x);
            b1 = b0 + ri.z;
            cam = ri.w;
        }
    } else if (active) { b0 = rowptr[cam]; b1 = rowptr[cam + 1]; }
    if (EPI == EPI_HESS) {   // the tCG's status word (written by the previous cg_step on another XCD: an L2 miss) travels WITH the row pointers:
        if (a.scal->status != 0) return;   // the dependent chain of this latency-bound kernel is one round trip shorter
    }
    if (EPI == EPI_AUTO) {
        const int ph = a.scal->phase;
        if (ph >= PH_STOP) return;
        role = (ph == PH_CAND) ? EPI_GRAD : EPI_HESS;
    }
    // the four groups of a wavefront loop together (wave-level trip count = the longest of their rows)
    int64_t span = b1 - b0;
#ifdef XM_BSR_TRACE
    int widx = 0;
#endif
    BSR_TS(1);
    span = max(span, (int64_t)__shfl_xor((long long)span, 16, 64));
    span = max(span, (int64_t)__shfl_xor((long long)span, 32, 64));
    double *st = stage + slot * 16 * SW;
    if (VAR == 2) {
        // Software pipeline over windows of 16 blocks per group (the kernel is latency-, not bandwidth-limited).
        // The 16 records of W a group multiplies are fetched ELEMENT-per-lane: one load instruction of the group covers 128
        // consecutive bytes of the concatenated records, so every cache line is touched by one or two instructions instead of by
        // all 3*OP of a lane-per-record gather; LDS turns them back into lane-per-record.
        // The load path (TA) is the limiter (TA_BUSY 85 % with 8-byte loads), so every global load is 16 bytes wide: a window
        // of 16 blocks is 72 pairs of doubles (5 loads per lane instead of 9), a record of W is (REC+1)/2 pairs whose last one
        // starts one double early when REC is odd (overlapping instead of over-reading).
        typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
        constexpr int NP = (REC + 1) / 2;                  // pairs per record of W
        // every load is UNCONDITIONAL (idle lanes re-read a valid neighbour, clamped addresses): predicated loads become
        // exec-mask branches whose outstanding-load count the compiler cannot track, and it then drains the next window's loads
        // (s_waitcnt vmcnt(0)) right after issuing them.
        // Pipeline state at the top of window k: blocks(k) and W-records(k) in flight since window k-1, column indices of window k+1 in flight
        // since window k-1.  Window k stages its data in LDS, requests cols(k+2), blocks(k+1) and - with the indices that have had a whole
        // window to arrive - W-records(k+1), and multiplies: no load waits on a load issued in the same window.
        // The blocks' load policy is a compile-time property of the pipeline's copy (a hint chosen by a branch next to the load is merged
        // into one plain load by the compiler): below the Infinity Cache's size the blocks are read with the DEFAULT policy -- they stay in the
        // L2s / the Infinity Cache between products: 13.0 -> 11.9 us at 425 k blocks (34 MB), 25.7 -> 21.8 at 929 k, 36.5 -> 30.9 at 1.39 M --
        // beyond it non-temporally, a pure stream (402 MB: 127.4 against 136.5 us); profiles/r06_kbench_bsr_policy.txt.  CamArgs.nt_cam0
        // (0 or nloc: set by the launcher from the size rule of the dense kernel) is the camera where the stream starts.
        // ONE set of load registers: a window first moves what has arrived (its blocks, then its records of W -- they share the LDS window) out
        // of the registers and requests the next window into them; its own arithmetic follows.  The requests leave a few dozen cycles later
        // than with a second register set, which costs nothing: the windows' loads queue in the CU's load path anyway (per-wavefront timestamps,
        // profiles/r06_kbench_bsr_trace.txt: requesting two windows at once moved their arrivals, not the sum).  The lane's position inside its
        // group is made opaque once per window -- derived from one loop-invariant value, the two dozen clamped offsets, record / piece numbers and
        // LDS addresses were all kept in registers across the loop.  Together: 124 instead of 160 VGPRs at o = 3, a FOURTH wavefront per SIMD
        // (1024 resident workgroups: the 856 of a 13 682-camera product in one round), three instead of two at o = 4, 5, two instead of one at
        // o = 6: 11.6 -> 11.1 us at 13 682 cameras (deg 12: 7.4 -> 6.5), same sums in the same order.  Offsets inside a row are 32-bit.
        auto pipeline = [&](auto ntflag) {
        constexpr bool NT = decltype(ntflag)::value;
        int jn = 0, jnn = 0, nd = 0;
        d2u t[5], tw[NP];
        const int len = (int)(b1 - b0);
        const double *rowblocks = blocks + b0 * 9;
        const int32_t *rowcols = colidx + b0;
        const int wbase = threadIdx.x & 48;
        auto load_cols = [&](int glo, int off, int &jj) { jj = (len > 0) ? rowcols[max(min(off + glo, len - 1), 0)] : 0; };
        auto load_blocks = [&](int glo, int off, int &ndd) {
            ndd = max(min(len - off, 16), 0) * 9;
            const double *src = (ndd > 0) ? rowblocks + off * 9 : blocks;
#pragma unroll
            for (int i = 0; i < 5; ++i)
                t[i] = NT ? __builtin_nontemporal_load((const d2u *)(src + max(min(2 * (glo + 16 * i), ndd - 2), 0)))
                          : *(const d2u *)(src + max(min(2 * (glo + 16 * i), ndd - 2), 0));
        };
        auto load_w = [&](int glo, int jj) {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int g = glo + 16 * i;
                const int sl = g / NP;
                const int start = min(2 * (g - sl * NP), REC - 2);
                const int js = __shfl(jj, wbase + sl, 64);
                tw[i] = *(const d2u *)(W + (size_t)((unsigned)js * (unsigned)REC + (unsigned)start));
            }
        };
        auto window = [&](int off, auto prefetch) {
#ifdef XM_BSR_TRACE
            BSR_TS(2 + min(widx, 2)); widx++;   // (this window's blocks and records have arrived)
#endif
            int glo = gl;
            asm volatile("" : "+v"(glo));
            double q[9], w[3][O];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int pos = max(min(2 * (glo + 16 * i), nd - 2), 0);
                st[pos] = t[i].x; st[pos + 1] = t[i].y;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int e = 0; e < 9; ++e) q[e] = st[glo * 9 + e];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int g = glo + 16 * i;
                const int sl = g / NP;
                const int start = min(2 * (g - sl * NP), REC - 2);
                st[sl * REC + start] = tw[i].x;
                st[sl * REC + start + 1] = tw[i].y;
            }
            if constexpr (decltype(prefetch)::value) {
                load_cols(glo, off + 32, jnn);
                load_blocks(glo, off + 16, nd);
                load_w(glo, jn);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int k = 0; k < O; ++k) w[c][k] = st[glo * REC + c * OP + k];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const bool keep = off + glo < len;
#pragma unroll
            for (int e = 0; e < 9; ++e) q[e] = keep ? q[e] : 0.0;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int k = 0; k < O; ++k) acc[r][k] += q[3 * r] * w[0][k] + q[3 * r + 1] * w[1][k] + q[3 * r + 2] * w[2][k];
            if constexpr (decltype(prefetch)::value) jn = jnn;
        };
        const int wspan = (int)span;
        if (wspan > 0) {
            int j0;
            load_cols(gl, 0, j0);
            load_cols(gl, 16, jn);
            load_blocks(gl, 0, nd);
            load_w(gl, j0);
            int off = 0;
            for (; off + 16 < wspan; off += 16) window(off, std::true_type{});
            window(off, std::false_type{});
        }
        };
        if ((int)(blockIdx.x * kBsrRows) >= a.nt_cam0) pipeline(std::true_type{});
        else pipeline(std::false_type{});
    } else
    for (int64_t off = 0; off < span; off += 16) {
        const int64_t base = b0 + off;
        const int64_t b = base + gl;
        const bool has = b < b1;
        const int j = has ? colidx[b] : 0;
        double q[9], w[3][O];
        if (VAR >= 1) {
            const int64_t left = b1 - base;
            const int64_t nd = ((left < 16) ? ((left > 0) ? left : 0) : 16) * 9;   // doubles of this group's window
            const double *src = blocks + base * 9;
            double t[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) t[i] = (gl + 16 * i < nd) ? __builtin_nontemporal_load(src + gl + 16 * i) : 0.0;   // pure stream
#pragma unroll
            for (int i = 0; i < 9; ++i) st[gl + 16 * i] = t[i];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int e = 0; e < 9; ++e) q[e] = st[gl * 9 + e];
        } else {
            const double *qb = blocks + (has ? b : 0) * 9;
#pragma unroll
            for (int e = 0; e < 9; ++e) q[e] = qb[e];
        }
        {
            const double *wj = W + (size_t)j * REC;
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int k = 0; k < O; ++k) w[c][k] = wj[c * OP + k];
        }
        if (has) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int k = 0; k < O; ++k) acc[r][k] += q[3 * r] * w[0][k] + q[3 * r + 1] * w[1][k] + q[3 * r + 2] * w[2][k];
        }
        if (VAR >= 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    // epilogue operands are fetched only now: holding ~50 more VGPRs through the gather phase costs rows in flight -- also in the latency
    // regime (13 682 cameras, 856 workgroups): requested up front they save a round trip but drop the kernel from three to two wavefronts per
    // SIMD, 22.5 against 20.3 us per Hessian product and 31.3 against 28.8 ms per solve on one box (profiles/r05_ab_rome.txt, lib_x)
    EpiOps eops;
    BSR_TS(5);
    epi_prefetch<O, EPI>(eops, active ? cam : 0, gl, active, a, (EPI == EPI_AUTO) ? role : (int)EPI);
    qw_finish<O, EPI, 16, kBsrRows>(cam, gl, slot, active, acc, alpha, a, eops, red, (EPI == EPI_AUTO) ? role : (int)EPI);
    BSR_TS(6);
}
#ifdef XM_BSR_TRACE
extern "C" void xm_bsr_trace_set(unsigned long long *p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bsr_trace), &p, sizeof(p)); }
#endif

// ----------------------------------------------------------------------------------------------------------------
// layout helpers
// ----------------------------------------------------------------------------------------------------------------
// dst[r*ldd + c] = src[r + c*lds]  (column-major host layout -> row-major padded device layout), 32x32 LDS tiles
__global__ __launch_bounds__(256) void transpose_pad_kernel(const double *__restrict__ src, int64_t lds, int64_t rows,
                                                             int64_t cols, double *__restrict__ dst, int64_t ldd) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int64_t r0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int64_t r = r0 + tx, c = c0 + ty + j;
        tile[ty + j][tx] = (r < rows && c < cols) ? src[r + c * lds] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int64_t r = r0 + ty + j, c = c0 + tx;
        if (r < rows && c < ldd) dst[r * ldd + c] = (c < cols) ? tile[tx][ty + j] : 0.0;
    }
}

__global__ void dense_from_bsr_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
                                      const double *__restrict__ blocks, int64_t nloc, int64_t cam0, double *dst, int64_t ldd) {
    const int64_t cam = blockIdx.x;
    if (cam >= nloc) return;
    for (int64_t b = rowptr[cam0 + cam] + threadIdx.x / 9; b < rowptr[cam0 + cam + 1]; b += blockDim.x / 9) {
        const int e = threadIdx.x % 9;
        if (threadIdx.x / 9 < (int)(blockDim.x / 9))
            dst[(cam * 3 + e / 3) * ldd + (int64_t)colidx[b] * 3 + e % 3] = blocks[b * 9 + e];
    }
}

// ----------------------------------------------------------------------------------------------------------------
// flat kernels (grid-stride over the nloc*3*OP elements; the row -> camera map is idx / (3*OP))
// ----------------------------------------------------------------------------------------------------------------
// (floating-point contraction stays off through the truncated-CG, trust-region and retraction kernels -- see xm_device.h: the host-driven
// kernels and the device-driven outer iteration evaluate the same expressions in different kernels and must get the same bits)
#pragma clang fp contract(off)
__device__ __forceinline__ unsigned long long pack_stat(int seq, int iter, int status) {   // [seq : 32 | iter : 24 | status : 8]
    return ((unsigned long long)(unsigned)seq << 32) | ((unsigned long long)((unsigned)iter & 0xffffffu) << 8) | (unsigned long long)(unsigned)(status & 0xff);
}
__device__ __forceinline__ void publish_host(unsigned long long *hstat, unsigned long long v) {
    if (hstat) __hip_atomic_store(hstat, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // the word is its own payload
}

struct StepDecision {
    int mode;     // 0 CG step (alpha) | 1 negative curvature (tau) | 2 boundary (tau) | 5 stop, residual below 1e-15
    double step;
};
// trustregion.h:565-600, evaluated identically by every thread that needs it
__device__ __forceinline__ StepDecision tcg_decide(const TcgScal &sc, double pHp) {
    StepDecision d;
    const double alpha = sc.rr / pHp;
    if (sc.rr < 1e-15) { d.mode = 5; d.step = 0.0; return d; }
    const bool neg = alpha <= 0.0;
    if (neg || sc.vv + 2.0 * alpha * sc.vp + alpha * alpha * sc.pp > sc.delta * sc.delta) {
        const double sq = sqrt(sc.vp * sc.vp + sc.pp * (sc.delta * sc.delta - sc.vv));
        d.mode = neg ? 1 : 2;
        d.step = (-sc.vp + sq) / sc.pp;
        return d;
    }
    d.mode = 0; d.step = alpha;
    return d;
}

template <int O>
__global__ __launch_bounds__(256) void scale_rows_kernel(int nloc, const double *__restrict__ R, const double *__restrict__ s,
                                                          double *__restrict__ Wloc) {
    constexpr int OP = pitch_of(O);
    const int64_t total = (int64_t)nloc * 3 * OP;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
        Wloc[i] = R[i] * s[i / (3 * OP)];
}

// r = rg, p = -rg, v = Hv = 0, W = s.*p + ps.*R   (trustregion.h:454-458, 476-482 and the first half of ehess :229-234)
template <int O>
__global__ __launch_bounds__(256) void tcg_init_kernel(int nloc, const double *__restrict__ rgR, const double *__restrict__ rgs,
                                                        const double *__restrict__ R, const double *__restrict__ s,
                                                        double *rR, double *rs, double *pR, double *ps, double *vR, double *vs,
                                                        double *HvR, double *Hvs, double *Wloc, TcgScal *scal0, double rr,
                                                        double delta, unsigned long long *hstat, double *Wpad, int seq, const SpecCtl *spec) {
    constexpr int OP = pitch_of(O);
    const int64_t total = (int64_t)nloc * 3 * OP;
    if (spec != nullptr) {   // speculative start (enqueued behind outer_finalize_kernel before the host knew the outcome)
        if (!spec->go) {     // not this way: the product / cg_step launches queued behind find a dormant state and return
            if (blockIdx.x == 0 && threadIdx.x == 0) { TcgScal sc = *scal0; sc.status = 9; sc.seq = seq; *scal0 = sc; }
            return;
        }
        rr = spec->rr; delta = spec->delta;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int cam = (int)(i / (3 * OP));
        const double g = rgR[i], gs = rgs[cam];
        rR[i] = g; pR[i] = -g; vR[i] = 0.0; HvR[i] = 0.0;
        const double wv = s[cam] * (-g) + (-gs) * R[i];
        if (Wloc) Wloc[i] = wv;   // (nullptr: the products of this tCG read the padded copy only, Context::run_tcg)
        if (Wpad) Wpad[(size_t)cam * 16 + (i - (int64_t)cam * (3 * OP))] = wv;   // copy at a 128-byte record pitch for the sliced-ELL gather (xm_sell.h)
        if (i % (3 * OP) == 0) { rs[cam] = gs; ps[cam] = -gs; vs[cam] = 0.0; Hvs[cam] = 0.0; }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        TcgScal sc = {};
        sc.rr = rr; sc.vv = 0.0; sc.vp = 0.0; sc.pp = rr; sc.delta = delta; sc.gradnorm = sqrt(rr); sc.last_step = 0.0; sc.model = 0.0;
        sc.status = 0; sc.iter = 0; sc.seq = seq;
        *scal0 = sc;
        publish_host(hstat, pack_stat(seq, 0, 0));
    }
}

// One flat kernel per tCG iteration (trustregion.h:565-644): alpha (or tau) from the gathered partial sums and the
// device-resident scalar block, the branch logic of :572-600, v/r/Hv updates, the :627 exit test, beta, the new direction p
// and the next product input W = s.*p_R + p_s.*R.  The residual norm after the step is obtained from the three inner products
// the Hessian epilogue already delivered, <r,r> + 2 alpha <r,Hp> + alpha^2 <Hp,Hp>, so beta needs no second grid-wide
// reduction; the directly summed |r|^2 (partsB_out) replaces that estimate as <r,r> of the NEXT iteration, so no error
// accumulates.  Block 0 owns the scalar state (other parity buffer) and the host-mapped progress word.  The scale parts of
// p and r are ping-ponged (cur -> next): every element thread of a camera reads them while one thread rewrites them.
template <int O>
__global__ __launch_bounds__(256) void cg_step_kernel(int nloc, const TcgScal *__restrict__ scal_cur, TcgScal *scal_next,
                                                       const double *__restrict__ parts, int nA_loc, int nB_loc, int world,
                                                       const double *__restrict__ HpR,
                                                       const double *__restrict__ Hps, const double *__restrict__ R,
                                                       const double *__restrict__ s, double *pR, const double *__restrict__ ps_cur,
                                                       double *ps_next, double *vR, double *vs, double *HvR, double *Hvs, double *rR,
                                                       const double *__restrict__ rs_cur, double *rs_next, double *Wloc,
                                                       double *partsB_out, unsigned long long *hstat, int b_off, int64_t mat,
                                                       double *Afull, double *Wfull, int grp, PeerXchg x, double *Wpad) {
    constexpr int OP = pitch_of(O);
    __shared__ double sh[4];
    __shared__ double sh16[16];
    __shared__ int sh_flag;
    const int64_t total = (int64_t)nloc * 3 * OP;
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    // operands of the thread's first element are requested before the scalar block / partial sums are read
    const bool in0 = i < total;
    const int camf = (int)(i / (3 * OP));
    const bool own0 = in0 && (i % (3 * OP) == 0);
    double hp = 0, pv = 0, vv0 = 0, hv = 0, rv = 0, Rv = 0, hs = 0, psv = 0, rsv = 0, sv = 1, vsv = 0, hvs = 0;
    // XM_FLAG_MODEL_RECURRENCE (HvR == nullptr): H v is not accumulated -- the model value travels in the scalar block instead (below): 2 x 3 n OP
    // doubles less read and written per iteration (28.8 MB of the launch's 93 MB at 100 k cameras, o = 3)
    const bool keep_hv = HvR != nullptr;
    if (in0) { hp = HpR[i]; pv = pR[i]; vv0 = vR[i]; if (keep_hv) hv = HvR[i]; rv = rR[i]; Rv = R[i]; hs = Hps[camf]; psv = ps_cur[camf]; rsv = rs_cur[camf]; sv = s[camf]; }
    if (own0) { vsv = vs[camf]; if (keep_hv) hvs = Hvs[camf]; }
    // ... and so are the first rounds of rank 0's partial sums (single GPU: all of them up to 512 workgroups): scalar block and partial sums
    // were written by the previous launches on other XCDs -- one memory round trip for both instead of two in a row
    // (not with the fused peer exchange: there the peers' chunks arrive DURING this launch)
    const bool can_pre = x.world <= 1;
    PartialsPre pre;
    if (can_pre) sum_partials_prefetch(parts + b_off, parts + b_off + nA_loc, parts + b_off + 2 * nA_loc, nA_loc, parts + b_off + 3 * nA_loc, nB_loc, pre, grp);
    const TcgScal sc0 = *scal_cur;
    const bool lead = (blockIdx.x == 0 && threadIdx.x == 0);
    if (sc0.status != 0) {
        if (lead) *scal_next = sc0;
        return;
    }
    if (x.world > 1) {
        // DIRECT PEER EXCHANGE fused into this launch (peer communicators, xm_comm.hip): `parts` is this rank's exchange buffer for the
        // parity of this iteration; its own chunk [rows of B | 3 nA sums of this iteration's Hessian epilogue | nB sums of the previous
        // cg_step] is complete (earlier launches of this stream).  Every workgroup stores a slice of it into the SAME place of every
        // peer's buffer, fences at system scope and takes a ticket; the last one publishes the epoch to the peers' flag words.  Then
        // everybody waits (bounded) for the peers' epochs and goes on with identical data on every rank.  Launches that find the tCG
        // finished returned above without touching the exchange, so ranks may enqueue ahead freely.
        const int par = sc0.iter & 1;
        const size_t chunk_d = (size_t)b_off + 3 * (size_t)nA_loc + nB_loc;
        const size_t off = ((size_t)par * x.world + x.rank) * chunk_d;
        const unsigned long long epoch = x.epoch_base + (unsigned long long)sc0.iter + 1ull;
        const double *src = x.buf[x.rank] + off;
        if (x.lite) {
            // write-through form: every payload store is itself a system-scope (sc0 sc1) store, acknowledged by its destination
            // before s_waitcnt vmcnt(0) lets the wave go on -- nothing of it is left in this device's L2, so the hand-off needs no
            // release fence (which writes back the WHOLE L2 of the XCD: the column-split product measured 39 vs 14 us for that)
            for (int p = 0; p < x.world; ++p) {
                if (p == x.rank) continue;
                double *dst = x.buf[p] + off;
                for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; j < chunk_d; j += (size_t)stride)
                    __hip_atomic_store(dst + j, src[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) {
                const unsigned long long t = __hip_atomic_fetch_add(x.ticket + par, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sh_flag = (t + 1 == gridDim.x);
                if (sh_flag) __hip_atomic_store(x.ticket + par, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            if (sh_flag && !x.mute && threadIdx.x < x.world && (int)threadIdx.x != x.rank)
                __hip_atomic_store(x.flag[threadIdx.x] + par * kMaxPeers + x.rank, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
            for (int p = 0; p < x.world; ++p) {
                if (p == x.rank) continue;
                double *dst = x.buf[p] + off;
                for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; j < chunk_d; j += (size_t)stride) dst[j] = src[j];
            }
            __threadfence_system();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) {
                const unsigned long long t = __hip_atomic_fetch_add(x.ticket + par, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                sh_flag = (t + 1 == gridDim.x);
                if (sh_flag) __hip_atomic_store(x.ticket + par, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            if (sh_flag && !x.mute && threadIdx.x < x.world && (int)threadIdx.x != x.rank) {
                __threadfence_system();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(x.flag[threadIdx.x] + par * kMaxPeers + x.rank, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int good = 1;
            const long long t0 = wall_clock64();
            const unsigned long long *f = x.flag[x.rank] + par * kMaxPeers;
            for (int p = 0; p < x.world && good; ++p) {
                if (p == x.rank) continue;
                while (__hip_atomic_load(f + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
                    if (wall_clock64() - t0 > x.spin_ticks) { good = 0; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            if (!good) __hip_atomic_store(x.err, epoch | (1ull << 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            sh_flag = good;
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
        __syncthreads();
        if (!sh_flag) {   // a peer never arrived: end the tCG with an error status (the host reports XM_ERR_COMM)
            if (lead) {
                TcgScal nx = sc0; nx.status = 7;
                *scal_next = nx;
                publish_host(hstat, pack_stat(nx.seq, nx.iter, nx.status));
            }
            return;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
    // `parts` = the gathered per-rank chunks [ <p,Hp> | <r,Hp> | <Hp,Hp> (nA_loc each, from this iteration's Hessian epilogue)
    //                                          | |r|^2 partials of the PREVIOUS iteration's cg_step (nB_loc) ]
    // (multi-rank: each chunk starts with b_off doubles = that rank's rows of the image of Hp, see below)
    const int chunk = b_off + 3 * nA_loc + nB_loc;
    double pHp = 0.0, rHp = 0.0, HpHp = 0.0, rr_prev = 0.0;
    for (int r = 0; r < world; ++r) {
        const double *pa = parts + (size_t)r * chunk + b_off;
        double t[4];
        if (r == 0 && can_pre) sum_partials256_x4_pre(pa, pa + nA_loc, pa + 2 * nA_loc, nA_loc, pa + 3 * nA_loc, nB_loc, sc0.iter > 0, pre, sh16, t, grp);
        else sum_partials256_x4(pa, pa + nA_loc, pa + 2 * nA_loc, nA_loc, pa + 3 * nA_loc, (sc0.iter > 0) ? nB_loc : 0, sh16, t, grp);
        pHp += t[0]; rHp += t[1]; HpHp += t[2];
        if (sc0.iter > 0) rr_prev += t[3];
    }
    TcgScal sc = sc0;
    if (sc0.iter > 0) sc.rr = rr_prev;   // exact |r|^2 summed by the previous iteration
    const StepDecision d = tcg_decide(sc, pHp);
    if (d.mode == 5) {
        if (lead) {
            TcgScal nx = sc; nx.status = 5; nx.last_step = 0.0;
            *scal_next = nx;
            publish_host(hstat, pack_stat(nx.seq, nx.iter, nx.status));
        }
        return;
    }
    const double step = d.step;
    const bool cg = (d.mode == 0);
    double rr_est = sc.rr + 2.0 * step * rHp + step * step * HpHp;
    if (rr_est < 0.0) rr_est = 0.0;
    const bool conv = cg && (sqrt(rr_est) < sc.gradnorm * fmin(sc.gradnorm, 0.1));   // trustregion.h:627
    const double beta = rr_est / sc.rr;
    const bool newdir = cg && !conv;
    double acc = 0.0;
    if (in0) {
        vR[i] = vv0 + step * pv;
        if (keep_hv) HvR[i] = hv + step * hp;
        if (cg) {
            const double rn = rv + step * hp;
            const double rsn = rsv + step * hs;
            rR[i] = rn;
            acc += rn * rn;
            if (newdir) {
                const double pn = beta * pv - rn;
                const double psn = beta * psv - rsn;
                pR[i] = pn;
                if (!Afull) {
                    const double wv = sv * pn + psn * Rv;
                    if (Wloc) Wloc[i] = wv;
                    if (Wpad) Wpad[(size_t)camf * 16 + (i - (int64_t)camf * (3 * OP))] = wv;
                }
                if (own0) ps_next[camf] = psn;
            }
            if (own0) { rs_next[camf] = rsn; const double q = rsn / sv; acc += q * q; }
        }
        if (own0) { vs[camf] = vsv + step * psv; if (keep_hv) Hvs[camf] = hvs + step * hs; }
    }
    for (i += stride; i < total; i += stride) {
        const int cam = (int)(i / (3 * OP));
        const bool own = (i % (3 * OP) == 0);
        const double hpi = HpR[i], pi = pR[i], hsi = Hps[cam], psi = ps_cur[cam];
        vR[i] += step * pi;
        if (keep_hv) HvR[i] += step * hpi;
        if (cg) {
            const double rn = rR[i] + step * hpi;
            const double rsn = rs_cur[cam] + step * hsi;
            rR[i] = rn;
            acc += rn * rn;
            if (newdir) {
                const double pn = beta * pi - rn;
                const double psn = beta * psi - rsn;
                pR[i] = pn;
                if (!Afull) {
                    const double wv = s[cam] * pn + psn * R[i];
                    if (Wloc) Wloc[i] = wv;
                    if (Wpad) Wpad[(size_t)cam * 16 + (i - (int64_t)cam * (3 * OP))] = wv;
                }
                if (own) ps_next[cam] = psn;
            }
            if (own) { rs_next[cam] = rsn; const double q = rsn / s[cam]; acc += q * q; }
        }
        if (own) { vs[cam] += step * psi; if (keep_hv) Hvs[cam] += step * hsi; }
    }
    if (Afull && newdir) {
        // Multi-rank tCG with ONE exchange per iteration.  The next product input is W+ = s.*p+ + ps+.*R with p+ = beta p - r+,
        // r+ = r + alpha Hp, hence  W+ = beta W - A+,  A+ = A + alpha B  where A is the image of r and B the image of Hp under
        // (xR, xs) -> s.*xR + xs.*R.  Every rank received every rank's rows of B in the same all-gather as the partial sums,
        // keeps A and W replicated and advances both here for ALL cameras - identical arithmetic on identical data, so the
        // replicas stay bit-identical and no second all-gather (of W) is needed.  A0 = -W0 because p0 = -r0.
        const int64_t full = mat * world;
        const bool first = (sc0.iter == 0);
        for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < full; j += stride) {
            const int64_t r = j / mat;
            const double b = parts[(size_t)r * chunk + (size_t)(j - r * mat)];
            const double w = Wfull[j];
            const double an = (first ? -w : Afull[j]) + step * b;
            Afull[j] = an;
            Wfull[j] = beta * w - an;
        }
    }
    if (cg) {
        const double tot = block_sum256(acc, sh);
        if (threadIdx.x == 0) partsB_out[blockIdx.x] = tot;
    }
    if (lead) {
        TcgScal nx = sc;
        nx.last_step = step;
        // m(v + step p) - m(v) = step <p, r> + step^2 <p,Hp> / 2 with <p, r> = -<r, r> (CG: r is orthogonal to the previous direction), for the
        // interior step (step = rr / <p,Hp>: -step rr / 2) and for the boundary / negative-curvature step alike (trustregion.h:605-610, 667-668
        // compute the same number from the accumulated vectors)
        if (!keep_hv) nx.model = sc.model - step * sc.rr + 0.5 * step * step * pHp;
        if (!cg) {
            nx.status = d.mode;
        } else {
            nx.rr = rr_est;
            nx.vv = sc.vv + 2.0 * step * sc.vp + step * step * sc.pp;  // trustregion.h:642-644
            nx.vp = beta * (sc.vp + step * sc.pp);
            nx.pp = beta * beta * sc.pp + rr_est;
            if (conv) {
                nx.status = 3;
            } else {
                nx.iter = sc.iter + 1;
                nx.status = (nx.iter >= kMaxInner) ? 6 : 0;
            }
        }
        *scal_next = nx;
        publish_host(hstat, pack_stat(nx.seq, nx.iter, nx.status));
    }
}

// End of an outer iteration: one 256-thread block adds the gathered partial sums in their fixed order and hands
// {f_new, <g,g>_new, model value, tCG exit status, inner iterations} to the host through mapped memory; the sequence word
// is written last.  Replaces three device-to-host copies and a stream synchronisation per outer iteration.
__global__ __launch_bounds__(256) void outer_finalize_kernel(const double *__restrict__ partsA, int nA_loc, int world,
                                                              const double *__restrict__ partsM, int nM,
                                                              const TcgScal *__restrict__ scal, double *hres, unsigned long long seq, int grp,
                                                              OuterArgs oa, SpecCtl *spec_out) {
    __shared__ double sh[4];
    double f = 0.0, rr = 0.0;
    for (int r = 0; r < world; ++r) {   // same grouping as the host-side summation it replaces: rank by rank
        f += sum_partials256(partsA + (size_t)r * 2 * nA_loc, nA_loc, sh, grp);
        rr += sum_partials256(partsA + (size_t)r * 2 * nA_loc + nA_loc, nA_loc, sh, grp);
    }
    const double m_vec = (nM > 0) ? sum_partials256(partsM, nM, sh, grp) : 0.0;
    if (threadIdx.x == 0) {
        const TcgScal sc = *scal;
        const double m = (nM > 0) ? m_vec : sc.model;     // XM_FLAG_MODEL_RECURRENCE: no partial sums of the model were formed
        __hip_atomic_store(hres + 0, f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(hres + 1, rr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(hres + 2, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(hres + 3, (double)sc.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(hres + 4, (double)sc.iter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (spec_out != nullptr) {
            // the trust-region update of trustregion.h:680-708, with the formulas (and the order) of Context::trust_region
            const int endreason = (sc.status == 0) ? 6 : sc.status;
            double delta = oa.delta;
            int shrink = oa.shrink_count;
            bool go = (m < 0.0) && endreason != 7;
            if (go) {
                const double rou = (f - oa.loss) / m;
                if (rou < 0.25) { delta *= 0.25; shrink++; }
                else if (rou > 0.75 && endreason <= 2) { delta = fmin(delta * 2, oa.delta_bar); shrink = 0; }
                else shrink = 0;
                bool stop_delta = false;
                if (shrink > 3) { delta *= 1e-3; shrink = 0; if (delta < 1e-20) stop_delta = true; }
                const bool reject = (f > oa.loss || rou < 0.1);
                // continue only on the plain path: accepted, no stop test of the next iteration's top fires (:527-543)
                go = !stop_delta && !reject && endreason != 5 && !(sqrt(rr) < oa.gradtol) && !oa.last_iter;
            }
            SpecCtl sp;
            sp.go = go ? 1 : 0; sp.pad_ = 0; sp.rr = rr; sp.delta = delta;
            *spec_out = sp;
            __hip_atomic_store(hres + 6, go ? 1.0 : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(hres + 7, delta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __atomic_thread_fence(__ATOMIC_SEQ_CST);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(hres + 5), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ----------------------------------------------------------------------------------------------------------------
// per-camera kernels (one thread per camera; 3 x O block in registers)
// ----------------------------------------------------------------------------------------------------------------
// Rout_i = MGS_rows(R_i + t D_i)  (Dense/batchedQR.h:42-67),  sout = s exp(t ds / s)  (trustregion.h:19-24),
// Wloc_i = sout_i * Rout_i  (the next product's input, trustregion.h:677).  The anchor's scale stays untouched.
__device__ __forceinline__ double det3(const double (&M)[3][3]);
// Polar retraction (XM_RETRACT_POLAR): the orthogonal factor U = (M M^T)^{-1/2} M of the 3 x O block M = R_i + t D_i -- the closest point
// of St(3, O) -- by the scaled Newton iteration X <- (g X + (X X^T)^{-1} X / g) / 2 on the 3 x O block itself (every step needs only
// the 3x3 Gram matrix and its cofactors; quadratically convergent; g = (|(X X^T)^{-1} X|_F / |X|_F)^(1/2) removes the slow start of
// a long step).  The reference uses this projection only after the solve (utils/recoversolution.py:65-86, via SVD).
template <int O>
__device__ __forceinline__ void polar_rows(double (&X)[3][O]) {
    for (int it = 0; it < 40; ++it) {
        double G[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = a; b < 3; ++b) {
                double tt = 0.0;
#pragma unroll
                for (int k = 0; k < O; ++k) tt += X[a][k] * X[b][k];
                G[a][b] = tt; G[b][a] = tt;
            }
        const double d = det3(G);
        if (!(d > 0.0)) return;   // rank-deficient block: leave it (the caller's MGS would divide by zero as well)
        double C[3][3];           // adjugate of the symmetric G: G^{-1} = C / d
        C[0][0] = G[1][1] * G[2][2] - G[1][2] * G[2][1]; C[0][1] = G[0][2] * G[2][1] - G[0][1] * G[2][2]; C[0][2] = G[0][1] * G[1][2] - G[0][2] * G[1][1];
        C[1][1] = G[0][0] * G[2][2] - G[0][2] * G[2][0]; C[1][2] = G[0][2] * G[1][0] - G[0][0] * G[1][2]; C[2][2] = G[0][0] * G[1][1] - G[0][1] * G[1][0];
        C[1][0] = C[0][1]; C[2][0] = C[0][2]; C[2][1] = C[1][2];
        double Y[3][O], nx = 0.0, ny = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int k = 0; k < O; ++k) {
                const double y = (C[a][0] * X[0][k] + C[a][1] * X[1][k] + C[a][2] * X[2][k]) / d;
                Y[a][k] = y; nx += X[a][k] * X[a][k]; ny += y * y;
            }
        const double g = sqrt(sqrt(ny / nx));
        double delta = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int k = 0; k < O; ++k) {
                const double z = 0.5 * (g * X[a][k] + Y[a][k] / g);
                delta += (z - X[a][k]) * (z - X[a][k]);
                X[a][k] = z;
            }
        if (delta < 1e-31) return;
    }
}

// the rows of one camera's 3 x O block back onto the manifold: modified Gram-Schmidt (Dense/batchedQR.h:42-67) or the polar factor
template <int O, int POLAR>
__device__ __forceinline__ void retract_rows(double (&q)[3][O]) {
    if constexpr (POLAR) {
        polar_rows<O>(q);
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double qq = 0.0;
#pragma unroll
            for (int k = 0; k < O; ++k) qq += q[i][k] * q[i][k];
            qq = sqrt(qq);
#pragma unroll
            for (int k = 0; k < O; ++k) q[i][k] /= qq;
#pragma unroll
            for (int j = i + 1; j < 3; ++j) {
                double uu = 0.0;
#pragma unroll
                for (int k = 0; k < O; ++k) uu += q[i][k] * q[j][k];
#pragma unroll
                for (int k = 0; k < O; ++k) q[j][k] -= uu * q[i][k];
            }
        }
    }
}

// MV: the same launch also delivers the model decrease of the step D = v it retracts, m = <v,Hv>/2 + <v,rg> in the product metric
// (trustregion.h:667-668): this camera's share, summed per workgroup into mv.parts[blockIdx.x] (fixed order: threads, then the DPP tree).
// The step is in registers anyway; a launch of its own (model_value_kernel: 4.5-8 us per outer iteration) reads it a second time.
// Wpad: the product input also at the 128-byte record pitch of the sliced-ELL gather (xm_sell.h), as tcg_init / cg_step write it.
struct ModelArgs { const double *HvR, *Hvs, *rgR, *rgs; double *parts; };
template <int O, int POLAR, bool MV = false>
__global__ __launch_bounds__(256) void retract_kernel(int nloc, int cam0, const double *__restrict__ R, const double *__restrict__ s,
                                                       const double *__restrict__ D, const double *__restrict__ ds, double t,
                                                       double *Rout, double *sout, double *Wloc, double *Wpad, ModelArgs mv) {
    constexpr int OP = pitch_of(O);
    __shared__ double sh_mv[4];
    const int cam = blockIdx.x * 256 + threadIdx.x;
    const bool live = cam < nloc;
    if constexpr (!MV) {
        if (!live) return;
    }
    const size_t base = (size_t)(live ? cam : 0) * 3 * OP;
    double q[3][O];
    double macc = 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < O; ++k) {
            const double d = D[base + r * OP + k];
            q[r][k] = R[base + r * OP + k] + t * d;
            if constexpr (MV) macc += d * (0.5 * mv.HvR[base + r * OP + k] + mv.rgR[base + r * OP + k]);
        }
    if constexpr (MV) {
        const int c = live ? cam : 0;
        const double vsds = ds[c] / (s[c] * s[c]);
        macc += vsds * (0.5 * mv.Hvs[c] + mv.rgs[c]);
        const double tot = block_sum256(live ? macc : 0.0, sh_mv);
        if (threadIdx.x == 0) mv.parts[blockIdx.x] = tot;
        if (!live) return;
    }
    retract_rows<O, POLAR>(q);
    const double so = s[cam];
    const double sn = ((cam0 + cam) == 0 || ds == nullptr) ? so : so * exp(t * ds[cam] / so);
    if (sout) sout[cam] = sn;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int k = 0; k < O; ++k) {
            Rout[base + r * OP + k] = q[r][k];
            if (Wloc) Wloc[base + r * OP + k] = sn * q[r][k];
            if (Wpad) Wpad[(size_t)cam * 16 + r * OP + k] = sn * q[r][k];
        }
        // the pad column of an even rank (pitch o + 1) is written too, in every copy: no kernel relies on another having zeroed it
        if (OP > O) { Rout[base + r * OP + O] = 0.0; if (Wloc) Wloc[base + r * OP + O] = 0.0; if (Wpad) Wpad[(size_t)cam * 16 + r * OP + O] = 0.0; }
    }
}

// ----------------------------------------------------------------------------------------------------------------
// Device-driven outer iteration (Context::trust_region_device): the step launch of a (product, step) pair.
// The host enqueues the SAME two launches over and over, several pairs ahead, and only watches a progress word; what a pair does is decided
// on the device (TcgScal.phase):
//   PH_TCG   product = Hessian product of tCG iteration i (EPI_AUTO in the Hessian role); this launch = cg_step_kernel's arithmetic, and when
//            that iteration ENDS the truncated CG (trustregion.h:572-600, 627, 664) the retraction of the step and its model decrease
//            (:667-678) follow in the same launch, per camera, from v + step p still in registers  -> PH_CAND
//   PH_CAND  product = cost / gradient at the candidate (EPI_AUTO in the gradient role); this launch = the trust-region update of
//            trustregion.h:680-708 evaluated by every workgroup from the same partial sums (the formulas of Context::trust_region),
//            the candidate copied over the current point when it is accepted, the stop tests of the next iteration's top (:527-543), and
//            tcg_init_kernel's work for the next truncated CG                                       -> PH_TCG or PH_STOP
// No launch is ever a run-ahead no-op, nothing waits for the host between an outer iteration's pieces (round 5: the host noticed the end
// of a tCG, enqueued retraction / product / result kernel and confirmed a speculative start: 16 % of the span idle at Final-13682 size in
// block CSR, and two to four no-op launches per outer iteration).  Every sum keeps the order the host-driven kernels use.
// ----------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long pack_prog(unsigned int run, int slots, int phase) {
    return ((unsigned long long)run << 32) | ((unsigned long long)((unsigned)slots & 0xffffffu) << 8) | (unsigned long long)(unsigned)(phase & 0xff);
}
// two sums of `count` partials and one of `count3` partials, each in the order of sum_partials256 -- where entry i of the third list is itself
// (w[4i] + w[4i+1]) + (w[4i+2] + w[4i+3]) of per-wavefront partials w[0 .. nw) (absent ones count 0): what block_sum256 makes of four wavefronts
__device__ __forceinline__ void sum_partials256_x4q(const double *p0, const double *p1, int count, const double *w, int count3, int nw, double *sh16,
                                                    double (&out)[4], int grp) {
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < count; i += 256) { const int j = sum_perm(i, count, grp); v[0] += p0[j]; v[1] += p1[j]; }
    for (int i = threadIdx.x; i < count3; i += 256) {
        const int j = 4 * sum_perm(i, count3, grp);
        const double a = w[j], b = (j + 1 < nw) ? w[j + 1] : 0.0, c = (j + 2 < nw) ? w[j + 2] : 0.0, d = (j + 3 < nw) ? w[j + 3] : 0.0;
        v[3] += (a + b) + (c + d);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = wave_sum(v[k]);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) sh16[k * 4 + (threadIdx.x >> 6)] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = (sh16[k * 4] + sh16[k * 4 + 1]) + (sh16[k * 4 + 2] + sh16[k * 4 + 3]);
}

template <int O>
__device__ __forceinline__ void outer_decide(const OuterStepArgs &A, const TcgScal &sc0, const OuterScal &os0, bool lead, int time_up, double *sh16) {
    constexpr int OP = pitch_of(O);
    const int64_t total = (int64_t)A.nloc * 3 * OP;
    const int64_t stride = (int64_t)gridDim.x * 256;
    double loss = os0.loss, rr = os0.rr_point, delta = sc0.delta;
    int shrink = os0.shrink_count, k = os0.k, stop = 0;
    long long totalite = os0.totalite;
    bool accept = false, start = true;
    if (sc0.phase == PH_CAND) {
        // same grouping as outer_finalize_kernel (rank by rank, one rank here)
        // (the three sums with all their loads in flight together and one pair of barriers; each keeps the tree of sum_partials256)
        double f = 0.0, rr_new = 0.0, t[4];
        sum_partials256_x4q(A.partsA, A.partsA + A.nA, A.nA, A.partsM, (A.HvR != nullptr) ? A.nM : 0, (A.nloc + 63) >> 6, sh16, t, A.grp);
        f += t[0];
        rr_new += t[1];
        const double m = (A.HvR != nullptr) ? t[3] : sc0.model;
        const int endreason = (sc0.status == 0) ? 6 : sc0.status;
        const int inner_print = sc0.iter + 1;
        int trstatus = 4;
        totalite += sc0.iter + 1;
        if (m >= 0.0) {   // "loss_qu is larger than 0": the point stays
            stop = 12; start = false;
        } else {
            const double rou = (f - loss) / m;   // trustregion.h:680-701
            if (rou < 0.25) { delta *= 0.25; trstatus = 1; shrink++; }
            else if (rou > 0.75 && endreason <= 2) { delta = fmin(delta * 2, A.delta_bar); trstatus = 2; shrink = 0; }
            else shrink = 0;
            bool stop_delta = false;
            if (shrink > 3) { delta *= 1e-3; shrink = 0; if (delta < 1e-20) stop_delta = true; }
            const bool reject = (f > loss || rou < 0.1);   // trustregion.h:702
            accept = stop_delta || !reject;
            if (stop_delta) {   // the reference leaves the new point in place but reports loss[k]
                stop = 13; start = false;
            } else {
                if (!reject) { loss = f; rr = rr_new; } else trstatus = 3;
                k += 1;
                if (k >= A.max_outer) { stop = 14; start = false; }
                else {
                    if (lead && A.trace != nullptr && k < A.trace_cap) {
                        double *t = A.trace + (size_t)k * 6;
                        t[0] = loss; t[1] = sqrt(rr); t[2] = (double)inner_print; t[3] = (double)endreason; t[4] = (double)trstatus; t[5] = delta;
                    }
                    if (endreason == 5) stop = 5;
                    else if (sqrt(rr) < A.gradtol) stop = 10;
                    else if (os0.time_up) stop = 11;
                    start = (stop == 0);
                }
            }
        }
    }
    if (accept || start) {
        const double *srcR = accept ? A.Rc : A.R, *srcs = accept ? A.sc : A.s;
        const double *srcg = accept ? A.cand.rgR : A.cur.rgR, *srcgs = accept ? A.cand.rgs : A.cur.rgs;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
            const int cam = (int)(i / (3 * OP));
            const bool own = (i % (3 * OP) == 0);
            const double Rv = srcR[i], sv = srcs[cam], g = srcg[i], gs = srcgs[cam];
            if (accept) {   // the candidate becomes the current point: buffers keep their roles (no pointer travels back to the host)
                A.R[i] = Rv; A.cur.G[i] = A.cand.G[i]; A.cur.rgR[i] = g;
                if (own) {
                    A.s[cam] = sv; A.cur.egs[cam] = A.cand.egs[cam]; A.cur.rgs[cam] = gs;
#pragma unroll
                    for (int j = 0; j < 9; ++j) A.cur.S0[(size_t)cam * 9 + j] = A.cand.S0[(size_t)cam * 9 + j];
                }
            }
            if (start) {    // tcg_init_kernel: r = rg, p = -rg, v = Hv = 0, W = s.*p + ps.*R
                A.rR[i] = g; A.pR[i] = -g; A.vR[i] = 0.0;
                if (A.HvR) A.HvR[i] = 0.0;
                const double wv = sv * (-g) + (-gs) * Rv;
                if (A.Wloc) A.Wloc[i] = wv;
                if (A.Wpad) A.Wpad[(size_t)cam * 16 + (i - (int64_t)cam * (3 * OP))] = wv;
                if (own) { A.rs_next[cam] = gs; A.ps_next[cam] = -gs; A.vs[cam] = 0.0; if (A.Hvs) A.Hvs[cam] = 0.0; }
            }
        }
    }
    if (lead) {
        TcgScal nx = {};
        nx.rr = rr; nx.pp = rr; nx.delta = delta; nx.gradnorm = sqrt(rr);
        nx.seq = sc0.seq;
        nx.phase = start ? PH_TCG : PH_STOP;
        OuterScal on = {};
        on.loss = loss; on.rr_point = rr; on.totalite = totalite; on.shrink_count = shrink; on.k = k; on.stop_reason = stop;
        on.time_up = time_up;
        on.slots = os0.slots + ((sc0.phase == PH_CAND) ? 1 : 0);
        *A.scal_next = nx;
        *A.os_next = on;
        publish_host(A.hprog, pack_prog(A.run, A.slot + 1, nx.phase));
    }
}

template <int O, int POLAR>
__global__ __launch_bounds__(256) void outer_step_kernel(OuterStepArgs A) {
    constexpr int OP = pitch_of(O);
    __shared__ double sh[4];
    __shared__ double sh16[16];
    const int64_t total = (int64_t)A.nloc * 3 * OP;
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    // as cg_step_kernel: operands of the thread's first element and the first rounds of the partial sums are requested before the scalar block
    const bool in0 = i < total;
    const int camf = (int)(i / (3 * OP));
    const bool own0 = in0 && (i % (3 * OP) == 0);
    const bool keep_hv = A.HvR != nullptr;
    double hp = 0, pv = 0, vv0 = 0, hv = 0, rv = 0, Rv = 0, hs = 0, psv = 0, rsv = 0, sv = 1, vsv = 0, hvs = 0;
    if (in0) { hp = A.HpR[i]; pv = A.pR[i]; vv0 = A.vR[i]; if (keep_hv) hv = A.HvR[i]; rv = A.rR[i]; Rv = A.R[i]; hs = A.Hps[camf]; psv = A.ps_cur[camf]; rsv = A.rs_cur[camf]; sv = A.s[camf]; }
    if (own0) { vsv = A.vs[camf]; if (keep_hv) hvs = A.Hvs[camf]; }
    PartialsPre pre;
    sum_partials_prefetch(A.parts, A.parts + A.nA, A.parts + 2 * A.nA, A.nA, A.parts + 3 * A.nA, A.nB, pre, A.grp);
    const TcgScal sc0 = *A.scal_cur;
    const OuterScal os0 = *A.os_cur;
    const bool lead = (blockIdx.x == 0 && threadIdx.x == 0);
    int time_up = 0;
    if (lead) time_up = os0.time_up | *A.stop_req;
    if (sc0.phase == PH_STOP) {
        if (lead) { *A.scal_next = sc0; *A.os_next = os0; }
        return;
    }
    if (sc0.phase != PH_TCG) {   // PH_CAND, PH_INIT
        outer_decide<O>(A, sc0, os0, lead, time_up, sh16);
        return;
    }
    double pHp = 0.0, rHp = 0.0, HpHp = 0.0, rr_prev = 0.0;
    {
        double t[4];
        sum_partials256_x4_pre(A.parts, A.parts + A.nA, A.parts + 2 * A.nA, A.nA, A.parts + 3 * A.nA, A.nB, sc0.iter > 0, pre, sh16, t, A.grp);
        pHp += t[0]; rHp += t[1]; HpHp += t[2];
        if (sc0.iter > 0) rr_prev += t[3];
    }
    TcgScal sc = sc0;
    if (sc0.iter > 0) sc.rr = rr_prev;   // exact |r|^2 summed by the previous iteration
    const StepDecision d = tcg_decide(sc, pHp);
    const bool stop5 = (d.mode == 5);
    const double step = d.step;
    const bool cg = (d.mode == 0);
    double rr_est = sc.rr + 2.0 * step * rHp + step * step * HpHp;
    if (rr_est < 0.0) rr_est = 0.0;
    const bool conv = cg && (sqrt(rr_est) < sc.gradnorm * fmin(sc.gradnorm, 0.1));   // trustregion.h:627
    const double beta = rr_est / sc.rr;
    const bool last = stop5 || !cg || conv || (sc.iter + 1 >= kMaxInner);            // this iteration ends the truncated CG
    if (!last) {
        // cg_step_kernel's update with a new direction
        double acc = 0.0;
        if (in0) {
            A.vR[i] = vv0 + step * pv;
            if (keep_hv) A.HvR[i] = hv + step * hp;
            const double rn = rv + step * hp;
            const double rsn = rsv + step * hs;
            A.rR[i] = rn;
            acc += rn * rn;
            const double pn = beta * pv - rn;
            const double psn = beta * psv - rsn;
            A.pR[i] = pn;
            const double wv = sv * pn + psn * Rv;
            if (A.Wloc) A.Wloc[i] = wv;
            if (A.Wpad) A.Wpad[(size_t)camf * 16 + (i - (int64_t)camf * (3 * OP))] = wv;
            if (own0) { A.ps_next[camf] = psn; A.rs_next[camf] = rsn; const double q = rsn / sv; acc += q * q; A.vs[camf] = vsv + step * psv; if (keep_hv) A.Hvs[camf] = hvs + step * hs; }
        }
        for (i += stride; i < total; i += stride) {
            const int cam = (int)(i / (3 * OP));
            const bool own = (i % (3 * OP) == 0);
            const double hpi = A.HpR[i], pi = A.pR[i], hsi = A.Hps[cam], psi = A.ps_cur[cam];
            A.vR[i] += step * pi;
            if (keep_hv) A.HvR[i] += step * hpi;
            const double rn = A.rR[i] + step * hpi;
            const double rsn = A.rs_cur[cam] + step * hsi;
            A.rR[i] = rn;
            acc += rn * rn;
            const double pn = beta * pi - rn;
            const double psn = beta * psi - rsn;
            A.pR[i] = pn;
            const double wv = A.s[cam] * pn + psn * A.R[i];
            if (A.Wloc) A.Wloc[i] = wv;
            if (A.Wpad) A.Wpad[(size_t)cam * 16 + (i - (int64_t)cam * (3 * OP))] = wv;
            if (own) { A.ps_next[cam] = psn; A.rs_next[cam] = rsn; const double q = rsn / A.s[cam]; acc += q * q; A.vs[cam] += step * psi; if (keep_hv) A.Hvs[cam] += step * hsi; }
        }
        const double tot = block_sum256(acc, sh);
        if (threadIdx.x == 0) A.partsB_out[blockIdx.x] = tot;
        if (lead) {
            TcgScal nx = sc;
            nx.last_step = step;
            if (!keep_hv) nx.model = sc.model - step * sc.rr + 0.5 * step * step * pHp;
            nx.rr = rr_est;
            nx.vv = sc.vv + 2.0 * step * sc.vp + step * step * sc.pp;  // trustregion.h:642-644
            nx.vp = beta * (sc.vp + step * sc.pp);
            nx.pp = beta * beta * sc.pp + rr_est;
            nx.iter = sc.iter + 1;
            nx.status = 0;
            nx.phase = PH_TCG;
            OuterScal on = os0;
            on.slots = os0.slots + 1;
            on.time_up = time_up;
            *A.scal_next = nx;
            *A.os_next = on;
            publish_host(A.hprog, pack_prog(A.run, A.slot + 1, PH_TCG));
        }
        return;
    }
    // The truncated CG ends here: the step eta = v + step p (v itself when the residual was already below 1e-15) is retracted straight
    // away, one thread per camera with the expressions of retract_kernel<O, POLAR, MV>; the tCG's vectors are not written back -- nothing reads
    // them before the next tcg_init.  A thread's loads are 72 .. 240 bytes apart from its neighbour's: a wavefront's load touches ~40 cache
    // lines, and four wavefronts per CU on 54 CUs (retract_kernel's shape at 13 682 cameras) queue up behind the CU's address unit -- here
    // only the FIRST wavefront of a workgroup takes 64 cameras, so that the work spreads over four times as many CUs.  The model decrease goes
    // out as one partial sum per WAVEFRONT; outer_decide adds four of them in block_sum256's order before it sums the list, which gives
    // retract_kernel's partial sums bit for bit.
    const double stp = stop5 ? 0.0 : step;
    const int nwave = (A.nloc + 63) >> 6;
    if ((threadIdx.x >> 6) == 0) {
        for (int gw = blockIdx.x; gw < nwave; gw += gridDim.x) {
            const int cam = gw * 64 + (int)threadIdx.x;
            const bool live = cam < A.nloc;
            const int c = live ? cam : 0;
            const size_t base = (size_t)c * 3 * OP;
            double q[3][O];
            double macc = 0.0;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int k = 0; k < O; ++k) {
                    const size_t idx = base + r * OP + k;
                    const double dv = A.vR[idx] + stp * A.pR[idx];
                    q[r][k] = A.R[idx] + dv;
                    if (keep_hv) {
                        const double hvv = A.HvR[idx] + stp * A.HpR[idx];
                        macc += dv * (0.5 * hvv + A.cur.rgR[idx]);
                    }
                }
            const double so = A.s[c];
            const double dsv = A.vs[c] + stp * A.ps_cur[c];
            if (keep_hv) {
                const double hvsv = A.Hvs[c] + stp * A.Hps[c];
                const double vsds = dsv / (so * so);
                macc += vsds * (0.5 * hvsv + A.cur.rgs[c]);
                const double tot = wave_sum(live ? macc : 0.0);
                if (threadIdx.x == 0) A.partsM[gw] = tot;
            }
            if (live) {
                retract_rows<O, POLAR>(q);
                const double sn = ((A.cam0 + cam) == 0) ? so : so * exp(dsv / so);
                A.sc[cam] = sn;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int k = 0; k < O; ++k) {
                        A.Rc[base + r * OP + k] = q[r][k];
                        if (A.Wloc) A.Wloc[base + r * OP + k] = sn * q[r][k];
                        if (A.Wpad) A.Wpad[(size_t)cam * 16 + r * OP + k] = sn * q[r][k];
                    }
                    if (OP > O) { A.Rc[base + r * OP + O] = 0.0; if (A.Wloc) A.Wloc[base + r * OP + O] = 0.0; if (A.Wpad) A.Wpad[(size_t)cam * 16 + r * OP + O] = 0.0; }
                }
            }
        }
    }
    if (lead) {
        TcgScal nx = sc;
        nx.last_step = stop5 ? 0.0 : step;
        if (!keep_hv && !stop5) nx.model = sc.model - step * sc.rr + 0.5 * step * step * pHp;
        if (stop5) nx.status = 5;
        else if (!cg) nx.status = d.mode;
        else {
            nx.rr = rr_est;
            nx.vv = sc.vv + 2.0 * step * sc.vp + step * step * sc.pp;
            nx.vp = beta * (sc.vp + step * sc.pp);
            nx.pp = beta * beta * sc.pp + rr_est;
            if (conv) nx.status = 3;
            else { nx.iter = sc.iter + 1; nx.status = 6; }
        }
        nx.phase = PH_CAND;
        OuterScal on = os0;
        on.slots = os0.slots + 1;
        on.time_up = time_up;
        *A.scal_next = nx;
        *A.os_next = on;
        publish_host(A.hprog, pack_prog(A.run, A.slot + 1, PH_CAND));
    }
}

// The same retraction (MGS-QR form) with a QUAD of lanes per camera -- `north_star` asks for cross-lane reductions per camera; the
// thread-per-camera kernel above is the default because it measures faster (scripts/kbench_retract.py, profiles/r04_kbench_retract.txt).
// Lane g of the quad owns columns g, g + 4, g + 8 of the camera's 3 x O block; the dot products of Gram-Schmidt are quad sums (two DPP
// butterflies).  Kept as the measured alternative (launch_retract variant 1); same results up to the summation order of the dot products.
template <int O>
__global__ __launch_bounds__(256) void retract_quad_kernel(int nloc, int cam0, const double *__restrict__ R, const double *__restrict__ s,
                                                            const double *__restrict__ D, const double *__restrict__ ds, double t,
                                                            double *Rout, double *sout, double *Wloc) {
    constexpr int OP = pitch_of(O), NC = (O + 3) / 4;
    const int g = threadIdx.x & 3;
    const int cam = blockIdx.x * 64 + (threadIdx.x >> 2);
    const bool active = cam < nloc;
    const size_t base = (size_t)(active ? cam : 0) * 3 * OP;
    double q[3][NC];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int k = g + 4 * c;
            q[r][c] = (active && k < O) ? R[base + r * OP + k] + t * D[base + r * OP + k] : 0.0;
        }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double qq = 0.0;
#pragma unroll
        for (int c = 0; c < NC; ++c) qq += q[i][c] * q[i][c];
        qq = sqrt(group_sum<4>(qq));
        if (!active) qq = 1.0;
#pragma unroll
        for (int c = 0; c < NC; ++c) q[i][c] /= qq;
#pragma unroll
        for (int j = i + 1; j < 3; ++j) {
            double uu = 0.0;
#pragma unroll
            for (int c = 0; c < NC; ++c) uu += q[i][c] * q[j][c];
            uu = group_sum<4>(uu);
#pragma unroll
            for (int c = 0; c < NC; ++c) q[j][c] -= uu * q[i][c];
        }
    }
    if (!active) return;
    const double so = s[cam];
    const double sn = ((cam0 + cam) == 0 || ds == nullptr) ? so : so * exp(t * ds[cam] / so);
    if (sout && g == 0) sout[cam] = sn;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int k = g + 4 * c;
            if (k < O) { Rout[base + r * OP + k] = q[r][c]; if (Wloc) Wloc[base + r * OP + k] = sn * q[r][c]; }
        }
        if (OP > O && g == 0) { Rout[base + r * OP + O] = 0.0; if (Wloc) Wloc[base + r * OP + O] = 0.0; }
    }
}

// Certificate multipliers, closed form per camera of the least-squares problem that the reference hands to Eigen's LSCG
// (checkeig.h:56-220): generators of camera 0 are the six symmetric unit matrices, of camera i>=1 the five
// traceless / off-diagonal ones.  Output: Lam_i = sum_k y_k A_k (row-major 3x3), dz_i = 2 lam (|sR row 3i|^2 - 1)
// (checkeig.h:30-40), partials {tr(Lam_0) [dual], lam*(1 - xii^2) [checkeig.h:324-332]}.
__device__ __forceinline__ void gen_matrix(int cam_is_anchor, int g, double (&A)[3][3]) {
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) A[a][b] = 0.0;
    if (cam_is_anchor) {
        // order (0,0),(0,1),(0,2),(1,1),(1,2),(2,2)   checkeig.h:71-98
        const int ii[6] = {0, 0, 0, 1, 1, 2}, jj[6] = {0, 1, 2, 1, 2, 2};
        const int i = ii[g], j = jj[g];
        if (i == j) A[i][i] = 1.0; else { A[i][j] = 0.5; A[j][i] = 0.5; }
    } else {
        // 0.5(E00-E11), 0.5(E11-E22), 0.5(E01+E10), 0.5(E02+E20), 0.5(E12+E21)   checkeig.h:100-161
        if (g == 0) { A[0][0] = 0.5; A[1][1] = -0.5; }
        else if (g == 1) { A[1][1] = 0.5; A[2][2] = -0.5; }
        else if (g == 2) { A[0][1] = 0.5; A[1][0] = 0.5; }
        else if (g == 3) { A[0][2] = 0.5; A[2][0] = 0.5; }
        else { A[1][2] = 0.5; A[2][1] = 0.5; }
    }
}

#pragma clang fp contract(fast)
template <int O>
__global__ __launch_bounds__(256) void cert_prepare_kernel(int nloc, int cam0, double lam, const double *__restrict__ QsR,
                                                            const double *__restrict__ R, const double *__restrict__ s,
                                                            double *Lam, double *dz, double *parts) {
    constexpr int OP = pitch_of(O);
    __shared__ double sh[4];
    const int cam = blockIdx.x * 256 + threadIdx.x;
    double d0 = 0.0, d1 = 0.0;
    if (cam < nloc) {
        const size_t base = (size_t)cam * 3 * OP;
        const bool anchor = (cam0 + cam) == 0;
        const double sc = s[cam];
        double B[3][O], Rt[3][O];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int k = 0; k < O; ++k) { B[r][k] = sc * R[base + r * OP + k]; Rt[r][k] = QsR[base + r * OP + k]; }
        double xii = 0.0;
#pragma unroll
        for (int k = 0; k < O; ++k) xii += B[0][k] * B[0][k];
        const double dzi = 2.0 * lam * (xii - 1.0);
#pragma unroll
        for (int k = 0; k < O; ++k) Rt[0][k] += dzi * B[0][k];  // Right = Z * sR with Z = C + diag term
        // P = B B^T, M = sym(Right B^T)
        double P[3][3], M[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                double tp = 0.0, tm = 0.0;
#pragma unroll
                for (int k = 0; k < O; ++k) { tp += B[a][k] * B[b][k]; tm += Rt[a][k] * B[b][k]; }
                P[a][b] = tp; M[a][b] = tm;
            }
        const int ng = anchor ? 6 : 5;
        double N[6][6], rhs[6];
        for (int g = 0; g < ng; ++g) {
            double A[3][3];
            gen_matrix(anchor, g, A);
            // rhs_g = <A_g B, Right> = <A_g, M>_F ;  AP = A_g P
            double t = 0.0, AP[3][3];
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) {
                    t += A[a][b] * M[a][b];
                    AP[a][b] = A[a][0] * P[0][b] + A[a][1] * P[1][b] + A[a][2] * P[2][b];
                }
            rhs[g] = t;
            for (int h = 0; h < ng; ++h) {
                double A2[3][3];
                gen_matrix(anchor, h, A2);
                double u = 0.0;  // <A_g B, A_h B> = tr(A_g P A_h^T)
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) u += AP[a][b] * A2[a][b];
                N[g][h] = u;
            }
        }
        // Gaussian elimination with partial pivoting (ng <= 6)
        for (int c = 0; c < ng; ++c) {
            int piv = c; double best = fabs(N[c][c]);
            for (int r = c + 1; r < ng; ++r) if (fabs(N[r][c]) > best) { best = fabs(N[r][c]); piv = r; }
            if (piv != c) {
                for (int j = 0; j < ng; ++j) { const double tt = N[c][j]; N[c][j] = N[piv][j]; N[piv][j] = tt; }
                const double tt = rhs[c]; rhs[c] = rhs[piv]; rhs[piv] = tt;
            }
            const double dd = N[c][c];
            if (dd != 0.0)
                for (int r = c + 1; r < ng; ++r) {
                    const double f = N[r][c] / dd;
                    for (int j = c; j < ng; ++j) N[r][j] -= f * N[c][j];
                    rhs[r] -= f * rhs[c];
                }
        }
        double y[6];
        for (int c = ng - 1; c >= 0; --c) {
            double acc = rhs[c];
            for (int j = c + 1; j < ng; ++j) acc -= N[c][j] * y[j];
            y[c] = (N[c][c] != 0.0) ? acc / N[c][c] : 0.0;
        }
        double L[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        for (int g = 0; g < ng; ++g) {
            double A[3][3];
            gen_matrix(anchor, g, A);
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) L[a][b] += y[g] * A[a][b];
        }
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) Lam[(size_t)cam * 9 + a * 3 + b] = L[a][b];
        dz[cam] = dzi;
        d0 = anchor ? (y[0] + y[3] + y[5]) : 0.0;   // checkeig.h:322
        d1 = (1.0 - xii * xii) * lam;               // checkeig.h:331
    }
    const double t0 = block_sum256(d0, sh);
    const double t1 = block_sum256(d1, sh);
    if (threadIdx.x == 0) { parts[blockIdx.x] = t0; parts[gridDim.x + blockIdx.x] = t1; }
}

// ----------------------------------------------------------------------------------------------------------------
// small vector kernels for the Lanczos eigen-solver of the certificate
// ----------------------------------------------------------------------------------------------------------------
// c[j] = V(:,j) . w   one block per column
__global__ __launch_bounds__(256) void dots_multi_kernel(const double *__restrict__ V, int64_t ldv, const double *__restrict__ w,
                                                          int64_t len, double *c) {
    __shared__ double sh[4];
    const double *v = V + (size_t)blockIdx.x * ldv;
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < len; i += 256) acc += v[i] * w[i];
    const double tot = block_sum256(acc, sh);
    if (threadIdx.x == 0) c[blockIdx.x] = tot;
}
// the same for long vectors (100 k cameras: one block per column walked 300 k elements alone, 295 us per call and three calls per
// Lanczos step): block (j, ch) sums one segment, the partial sums of a column are added in segment order by the second kernel
__global__ __launch_bounds__(256) void dots_multi_seg_kernel(const double *__restrict__ V, int64_t ldv, const double *__restrict__ w,
                                                              int64_t len, int64_t seg, double *__restrict__ part) {
    __shared__ double sh[4];
    const double *v = V + (size_t)blockIdx.x * ldv;
    const int64_t i0 = (int64_t)blockIdx.y * seg, i1 = (i0 + seg < len) ? i0 + seg : len;
    double acc = 0.0;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) acc += v[i] * w[i];
    const double tot = block_sum256(acc, sh);
    if (threadIdx.x == 0) part[(size_t)blockIdx.x * gridDim.y + blockIdx.y] = tot;
}
__global__ __launch_bounds__(256) void dots_multi_fin_kernel(const double *__restrict__ part, int m, int nseg, double *__restrict__ c) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    double t = 0.0;
    for (int q = 0; q < nseg; ++q) t += part[(size_t)j * nseg + q];
    c[j] = t;
}
// w -= V c
__global__ __launch_bounds__(256) void sub_vc_kernel(double *w, const double *__restrict__ V, int64_t ldv,
                                                      const double *__restrict__ c, int m, int64_t len) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < len; i += (int64_t)gridDim.x * 256) {
        double acc = 0.0;
        for (int j = 0; j < m; ++j) acc += V[(size_t)j * ldv + i] * c[j];
        w[i] -= acc;
    }
}
// One Gram-Schmidt pass of a Lanczos step in TWO launches instead of three (+ the alpha kernel): dots_multi_seg_kernel leaves the per-segment partial
// dots; this kernel adds them in segment order -- every workgroup for itself, into LDS: the sums of dots_multi_fin_kernel, bit for bit -- and
// subtracts V c from w.  Workgroup 0 keeps the coefficients (c_out) and, in the second pass (c_prev != nullptr), writes alpha_j = c_prev[j] + c[j].
// The Final-13682 certificate is 32 Lanczos steps of eleven launches at ~4-6 us each: launch-bound (profiles/r06_trace_summary_rome_bsr.txt).
constexpr int kLzMaxCols = 1024;
__global__ __launch_bounds__(256) void sub_vc_fin_kernel(double *w, const double *__restrict__ V, int64_t ldv, const double *__restrict__ part, int nseg,
                                                          int m, int64_t len, double *c_out, const double *c_prev, double *alpha_j) {
    __shared__ double cs[kLzMaxCols];
    for (int j = threadIdx.x; j < m; j += 256) {
        double t = 0.0;
        for (int q = 0; q < nseg; ++q) t += part[(size_t)j * nseg + q];
        cs[j] = t;
        if (blockIdx.x == 0) {
            c_out[j] = t;
            if (c_prev != nullptr && j == m - 1) *alpha_j = c_prev[j] + t;
        }
    }
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < len; i += (int64_t)gridDim.x * 256) {
        double acc = 0.0;
        for (int j = 0; j < m; ++j) acc += V[(size_t)j * ldv + i] * cs[j];
        w[i] -= acc;
    }
}
// beta_j = sqrt(<w,w>) from the per-segment partial sums (added in segment order, as dots_multi_fin_kernel does), v_{j+1} = w / beta_j
__global__ __launch_bounds__(256) void lz_next_fin_kernel(double *dst, const double *__restrict__ w, const double *__restrict__ part, int nseg, double *beta_j,
                                                           int64_t len) {
    double ww = 0.0;
    for (int q = 0; q < nseg; ++q) ww += part[q];
    const double beta = sqrt(fmax(ww, 0.0));
    if (blockIdx.x == 0 && threadIdx.x == 0) *beta_j = beta;
    const double inv = (beta > 0.0) ? 1.0 / beta : 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < len; i += (int64_t)gridDim.x * 256) dst[i] = w[i] * inv;
}
__global__ __launch_bounds__(256) void gemv_n_kernel(double *y, const double *__restrict__ V, int64_t ldv,
                                                      const double *__restrict__ c, int m, int64_t len) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < len; i += (int64_t)gridDim.x * 256) {
        double acc = 0.0;
        for (int j = 0; j < m; ++j) acc += V[(size_t)j * ldv + i] * c[j];
        y[i] = acc;
    }
}
// Lanczos bookkeeping on the device (no host round trip per step): alpha_j = c1[j] + c2[j];  beta_j = sqrt(<w,w>);
// v_{j+1} = w / beta_j
__global__ void lz_alpha_kernel(const double *__restrict__ c1j, const double *__restrict__ c2j, double *alpha_j) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *alpha_j = *c1j + *c2j;
}
__global__ __launch_bounds__(256) void lz_next_kernel(double *dst, const double *__restrict__ w, const double *__restrict__ ww, double *beta_j,
                                                       int64_t len) {
    const double beta = sqrt(fmax(*ww, 0.0));
    if (blockIdx.x == 0 && threadIdx.x == 0) *beta_j = beta;
    const double inv = (beta > 0.0) ? 1.0 / beta : 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < len; i += (int64_t)gridDim.x * 256) dst[i] = w[i] * inv;
}
__global__ __launch_bounds__(256) void scale_copy_kernel(double *dst, const double *__restrict__ src, double a, int64_t len) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < len; i += (int64_t)gridDim.x * 256) dst[i] = a * src[i];
}

// ----------------------------------------------------------------------------------------------------------------
// Solution recovery (reference utils/recoversolution.py:22-86): rank-r -> 3 projection, per-camera scale,
// anchoring to camera 0 and projection of every 3x3 block onto O(3) (the polar factor U V^T of its SVD).
// One thread per camera: the whole per-camera problem is 9 doubles, so a 64-lane wavefront per camera would idle 55 lanes;
// the polar factor is computed by the scaled Newton iteration X <- (g X + X^{-T}/g)/2 (quadratically convergent; the blocks
// are already within round-off of a rotation times a scale, so 3-4 steps reach 1e-16), which equals U V^T of the SVD.
// ----------------------------------------------------------------------------------------------------------------
// partial Gram matrix G = sR^T sR (r x r, r <= 10): block partial sums, summed on the host
__global__ __launch_bounds__(256) void recover_gram_kernel(int64_t n, int r, const double *__restrict__ R /* 3n x r col-major */,
                                                            const double *__restrict__ s, double *parts /* grid x r*r */) {
    __shared__ double sh[4];
    const int64_t m = 3 * n;
    for (int a = 0; a < r; ++a)
        for (int b = a; b < r; ++b) {
            double acc = 0.0;
            for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < m; row += (int64_t)gridDim.x * 256) {
                const double sc = s[row / 3];
                acc += (sc * R[row + a * m]) * (sc * R[row + b * m]);
            }
            const double t = block_sum256(acc, sh);
            if (threadIdx.x == 0) { parts[(size_t)blockIdx.x * r * r + a * r + b] = t; parts[(size_t)blockIdx.x * r * r + b * r + a] = t; }
        }
}

__device__ __forceinline__ double det3(const double (&M)[3][3]) {
    return M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
           M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
}
// Orthogonal polar factor U V^T of ANY 3x3 matrix through its SVD by one-sided (Hestenes) Jacobi rotations: the columns of A = X V are
// made orthogonal, their norms are the singular values, U = the normalised columns.  A rank-deficient block (a camera whose rows of
// the recovered factor are dependent or zero -- the reference's numpy SVD, utils/recoversolution.py:65-86, still returns an orthogonal
// U V^T there) gets its missing left vectors from cross products / the coordinate axes, so the result is always orthogonal; the zero
// matrix maps to the identity.
__device__ __forceinline__ void polar3_svd(double (&X)[3][3]) {
    double A[3][3], V[3][3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) { A[a][b] = X[a][b]; V[a][b] = (a == b) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double al = 0.0, be = 0.0, ga = 0.0;
                for (int k = 0; k < 3; ++k) { al += A[k][p] * A[k][p]; be += A[k][q] * A[k][q]; ga += A[k][p] * A[k][q]; }
                if (ga == 0.0 || fabs(ga) <= 1e-17 * sqrt(al * be)) continue;
                off += fabs(ga);
                const double zeta = (be - al) / (2.0 * ga);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int k = 0; k < 3; ++k) {
                    const double ap = A[k][p], aq = A[k][q];
                    A[k][p] = c * ap - sn * aq; A[k][q] = sn * ap + c * aq;
                    const double vp = V[k][p], vq = V[k][q];
                    V[k][p] = c * vp - sn * vq; V[k][q] = sn * vp + c * vq;
                }
            }
        if (off == 0.0) break;
    }
    double sg[3], smax = 0.0;
    for (int j = 0; j < 3; ++j) { sg[j] = sqrt(A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j]); smax = fmax(smax, sg[j]); }
    double U[3][3];
    bool have[3];
    int nh = 0;
    for (int j = 0; j < 3; ++j) {
        have[j] = smax > 0.0 && sg[j] > 1e-13 * smax;
        if (have[j]) { for (int k = 0; k < 3; ++k) U[k][j] = A[k][j] / sg[j]; ++nh; }
    }
    if (nh == 0) { for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) X[a][b] = (a == b) ? 1.0 : 0.0; return; }
    if (nh == 1) {   // one direction known: a second one orthogonal to it from the coordinate axis it is least aligned with
        int j0 = have[0] ? 0 : (have[1] ? 1 : 2), j1 = (j0 + 1) % 3;
        int ax = 0;
        for (int k = 1; k < 3; ++k) if (fabs(U[k][j0]) < fabs(U[ax][j0])) ax = k;
        double e[3] = {0.0, 0.0, 0.0};
        e[ax] = 1.0;
        const double d = U[ax][j0];
        double nn = 0.0;
        for (int k = 0; k < 3; ++k) { e[k] -= d * U[k][j0]; nn += e[k] * e[k]; }
        nn = sqrt(nn);
        for (int k = 0; k < 3; ++k) U[k][j1] = e[k] / nn;
        have[j1] = true;
    }
    for (int j = 0; j < 3; ++j)
        if (!have[j]) {   // the third direction: cross product of the other two, oriented so that det(U) = det(V) (U V^T is then a rotation)
            const int p = (j + 1) % 3, q = (j + 2) % 3;
            U[0][j] = U[1][p] * U[2][q] - U[2][p] * U[1][q];
            U[1][j] = U[2][p] * U[0][q] - U[0][p] * U[2][q];
            U[2][j] = U[0][p] * U[1][q] - U[1][p] * U[0][q];
            if (det3(V) < 0.0) for (int k = 0; k < 3; ++k) U[k][j] = -U[k][j];
        }
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) X[a][b] = U[a][0] * V[b][0] + U[a][1] * V[b][1] + U[a][2] * V[b][2];
}
// The blocks of a recovered solution are within round-off of a rotation times a scale: the scaled Newton iteration X <- (g X + X^{-T}/g)/2
// reaches the polar factor in 3-4 steps.  Anything it cannot handle -- a singular or numerically rank-deficient block, a non-finite
// cofactor, no convergence -- goes through the SVD above.
__device__ __forceinline__ void polar3(double (&X)[3][3]) {
    double X0[3][3];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) X0[a][b] = X[a][b];
    bool ok = false;
    for (int it = 0; it < 30; ++it) {
        const double d = det3(X);
        double nx = 0.0;
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) nx += X[a][b] * X[a][b];
        if (!(fabs(d) > 1e-10 * nx * sqrt(nx))) break;   // singular / rank-deficient (or NaN): not Newton's business
        double C[3][3];   // cofactor matrix: X^{-T} = C / det
        C[0][0] = X[1][1] * X[2][2] - X[1][2] * X[2][1]; C[0][1] = X[1][2] * X[2][0] - X[1][0] * X[2][2]; C[0][2] = X[1][0] * X[2][1] - X[1][1] * X[2][0];
        C[1][0] = X[0][2] * X[2][1] - X[0][1] * X[2][2]; C[1][1] = X[0][0] * X[2][2] - X[0][2] * X[2][0]; C[1][2] = X[0][1] * X[2][0] - X[0][0] * X[2][1];
        C[2][0] = X[0][1] * X[1][2] - X[0][2] * X[1][1]; C[2][1] = X[0][2] * X[1][0] - X[0][0] * X[1][2]; C[2][2] = X[0][0] * X[1][1] - X[0][1] * X[1][0];
        double nc = 0.0;
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) nc += C[a][b] * C[a][b];
        const double g = sqrt(sqrt(nc) / fabs(d) / sqrt(nx));   // Frobenius scaling: g = (|X^{-1}|_F / |X|_F)^(1/2)
        double delta = 0.0;
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                const double y = 0.5 * (g * X[a][b] + C[a][b] / (d * g));
                delta += (y - X[a][b]) * (y - X[a][b]);
                X[a][b] = y;
            }
        if (delta < 1e-31) { ok = true; break; }
    }
    if (!ok) {
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) { const double v = X0[a][b]; X[a][b] = (v == v && fabs(v) < 1e300) ? v : 0.0; }
        polar3_svd(X);
    }
}

// V: r x 3 (col-major) basis of the top-3 eigenspace of sR^T sR (identity when r == 3).  Outputs: rot (3 x 3n col-major,
// block i = columns 3i..3i+2, like the reference's R_real), scale (n), negdet partial counts.
__global__ __launch_bounds__(256) void recover_project_kernel(int64_t n, int r, const double *__restrict__ R, const double *__restrict__ s,
                                                               const double *__restrict__ V, double *rot, double *scale, int *negcount) {
    const int64_t m = 3 * n;
    const int64_t cam = (int64_t)blockIdx.x * 256 + threadIdx.x;
    // anchor block (camera 0): A = Rt_0^T with Rt_0 = B_0^T / s_0  ->  A = B_0 / s_0
    double B0[3][3], n0 = 0.0;
    for (int a = 0; a < 3; ++a)
        for (int c = 0; c < 3; ++c) {
            double t = 0.0;
            for (int k = 0; k < r; ++k) t += s[0] * R[a + (int64_t)k * m] * V[k + c * r];
            B0[a][c] = t; n0 += t * t;
        }
    const double s0 = sqrt(n0) / sqrt(3.0);
    if (cam >= n) return;
    double B[3][3], nb = 0.0;
    for (int a = 0; a < 3; ++a)
        for (int c = 0; c < 3; ++c) {
            double t = 0.0;
            for (int k = 0; k < r; ++k) t += s[cam] * R[3 * cam + a + (int64_t)k * m] * V[k + c * r];
            B[a][c] = t; nb += t * t;
        }
    const double sc = sqrt(nb) / sqrt(3.0);           // recoversolution.py:40
    double X[3][3];                                     // X = (B_0/s_0) * (B_i^T / s_i)   (recoversolution.py:41-47)
    const double den = s0 * sc;                         // a camera whose block vanished (scale 0): X = 0, the projection returns the identity
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) X[a][b] = (den > 0.0) ? (B0[a][0] * B[b][0] + B0[a][1] * B[b][1] + B0[a][2] * B[b][2]) / den : 0.0;
    polar3(X);
    const bool neg = det3(X) < 0.0;
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) rot[a + 3 * (3 * cam + b)] = X[a][b];
    scale[cam] = sc;
    if (neg) atomicAdd(negcount, 1);
}
// The same projection with ONE WAVEFRONT PER CAMERA and cross-lane reductions -- the form BASELINE.json's north_star names for the 3x3
// SVD ("one wavefront per camera ... with warp-shuffle reductions").  Lane e < 9 owns entry (e / 3, e % 3) of every 3x3 matrix; products
// across entries come through shuffles (ds_bpermute), sums over entries through the DPP wave reduction.  55 of 64 lanes carry no work:
// measured beside the thread-per-camera kernel by scripts/kbench_recover.py (profiles/r05_kbench_recover.txt), which is why the thread
// form stays the default.  Rank-deficient blocks take the same polar3 fallback (every lane gathers the block and runs it redundantly).
__global__ __launch_bounds__(256) void recover_project_wave_kernel(int64_t n, int r, const double *__restrict__ R, const double *__restrict__ s,
                                                                    const double *__restrict__ V, double *rot, double *scale, int *negcount) {
    const int64_t m = 3 * n;
    const int lane = threadIdx.x & 63;
    const int64_t cam = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cam >= n) return;   // wave-uniform
    const int e = lane % 9, a = e / 3, c = e % 3;
    const bool on = lane < 9;
    double b0 = 0.0, bi = 0.0;   // entry (a, c) of B_0 and of B_cam
    for (int k = 0; k < r; ++k) {
        const double v = V[k + c * r];
        b0 += s[0] * R[a + (int64_t)k * m] * v;
        bi += s[cam] * R[3 * cam + a + (int64_t)k * m] * v;
    }
    const double s0 = sqrt(wave_sum(on ? b0 * b0 : 0.0)) / sqrt(3.0);
    const double sc = sqrt(wave_sum(on ? bi * bi : 0.0)) / sqrt(3.0);
    const double den = s0 * sc;
    // X[a][c] = sum_j B0[a][j] B[c][j] / den : row a of B_0 (lanes 3a..3a+2), row c of B (lanes 3c..3c+2)
    double x = 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) x += __shfl(b0, 3 * a + j, 64) * __shfl(bi, 3 * c + j, 64);
    x = (den > 0.0) ? x / den : 0.0;
    const int a1 = (a + 1) % 3, a2 = (a + 2) % 3, c1 = (c + 1) % 3, c2 = (c + 2) % 3;
    bool ok = false;
    const double x_in = x;
    for (int it = 0; it < 30; ++it) {
        // cofactor of entry (a, c) by the cyclic formula; det = sum over row 0 of X .* C; norms over the nine entries
        const double cf = __shfl(x, 3 * a1 + c1, 64) * __shfl(x, 3 * a2 + c2, 64) - __shfl(x, 3 * a1 + c2, 64) * __shfl(x, 3 * a2 + c1, 64);
        const double d = wave_sum((lane < 3) ? x * cf : 0.0);
        const double nx = wave_sum(on ? x * x : 0.0), nc = wave_sum(on ? cf * cf : 0.0);
        if (!(fabs(d) > 1e-10 * nx * sqrt(nx))) break;
        const double g = sqrt(sqrt(nc) / fabs(d) / sqrt(nx));
        const double y = 0.5 * (g * x + cf / (d * g));
        const double delta = wave_sum(on ? (y - x) * (y - x) : 0.0);
        x = y;
        if (delta < 1e-31) { ok = true; break; }
    }
    if (!ok) {   // wave-uniform: rank-deficient block -> SVD, run redundantly by every lane on the gathered block
        double X[3][3];
#pragma unroll
        for (int q = 0; q < 9; ++q) { const double v = __shfl(x_in, q, 64); X[q / 3][q % 3] = (v == v && fabs(v) < 1e300) ? v : 0.0; }
        polar3_svd(X);
        x = X[a][c];
    }
    const double cf = __shfl(x, 3 * a1 + c1, 64) * __shfl(x, 3 * a2 + c2, 64) - __shfl(x, 3 * a1 + c2, 64) * __shfl(x, 3 * a2 + c1, 64);
    const double d = wave_sum((lane < 3) ? x * cf : 0.0);
    if (on) rot[a + 3 * (3 * cam + c)] = x;
    if (lane == 0) { scale[cam] = sc; if (d < 0.0) atomicAdd(negcount, 1); }
}
__global__ __launch_bounds__(256) void negate_kernel(double *x, int64_t len) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < len; i += (int64_t)gridDim.x * 256) x[i] = -x[i];
}


// ----------------------------------------------------------------------------------------------------------------
// XM^2 re-weighting on a resident context (reference loop 3_test_colmap_glomap.py:299-351: residual per
// observation -> 90-percentile filter -> rebuild Q -> solve again).  For a view-graph Q = sum_e w_e G_e (connection Laplacian:
// Q_ii += w_e I, Q_jj += w_e I, Q_ij = -w_e M_e, Q_ji = Q_ij^T) the rebuild is linear in the weights and runs on the device.
// ----------------------------------------------------------------------------------------------------------------
// position of block (row, col) in the local CSR arrays (-1: row not local or block not stored); rows need not be sorted
__global__ __launch_bounds__(256) void edge_locate_kernel(int64_t ne, const int32_t *__restrict__ ei, const int32_t *__restrict__ ej,
                                                           int cam0, int nloc, const int64_t *__restrict__ rowptr,
                                                           const int32_t *__restrict__ colidx, int64_t *pos_ij, int64_t *pos_ji) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= ne) return;
    auto find = [&](int r, int c) -> int64_t {
        const int lr = r - cam0;
        if (lr < 0 || lr >= nloc) return -1;
        for (int64_t b = rowptr[lr]; b < rowptr[lr + 1]; ++b)
            if (colidx[b] == c) return b;
        return -2;   // local row, block missing: the edge list does not match the stored pattern
    };
    pos_ij[e] = find(ei[e], ej[e]);
    pos_ji[e] = find(ej[e], ei[e]);
}
__global__ __launch_bounds__(256) void diag_locate_kernel(int nloc, int cam0, const int64_t *__restrict__ rowptr,
                                                           const int32_t *__restrict__ colidx, int64_t *pos_d) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= nloc) return;
    int64_t f = -2;
    for (int64_t b = rowptr[c]; b < rowptr[c + 1]; ++b)
        if (colidx[b] == cam0 + c) { f = b; break; }
    pos_d[c] = f;
}
// off-diagonal blocks of every edge: -w M_e (row i) and -w M_e^T (row j); DENSE: Q row-major with leading dimension ld, local rows
template <bool DENSE>
__global__ __launch_bounds__(256) void edge_write_kernel(int64_t ne, const int32_t *__restrict__ ei, const int32_t *__restrict__ ej,
                                                          const double *__restrict__ M, const double *__restrict__ w, int cam0, int nloc,
                                                          const int64_t *__restrict__ pos_ij, const int64_t *__restrict__ pos_ji,
                                                          double *blocks, double *Q, int64_t ld) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= ne) return;
    const double we = w[e];
    const int i = ei[e], j = ej[e];
    double m[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) m[k] = -we * M[e * 9 + k];
    if (DENSE) {
        if (i >= cam0 && i < cam0 + nloc)
            for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) Q[(size_t)(3 * (i - cam0) + a) * ld + 3 * j + c] = m[3 * a + c];
        if (j >= cam0 && j < cam0 + nloc)
            for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) Q[(size_t)(3 * (j - cam0) + a) * ld + 3 * i + c] = m[3 * c + a];
    } else {
        if (pos_ij[e] >= 0) for (int k = 0; k < 9; ++k) blocks[pos_ij[e] * 9 + k] = m[k];
        if (pos_ji[e] >= 0) for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) blocks[pos_ji[e] * 9 + 3 * a + c] = m[3 * c + a];
    }
}
// diagonal blocks: (sum of the incident weights, added in the fixed order of the incidence list) * I
template <bool DENSE>
__global__ __launch_bounds__(256) void diag_write_kernel(int nloc, int cam0, const int64_t *__restrict__ inc_ptr, const int32_t *__restrict__ inc_edge,
                                                          const double *__restrict__ w, const int64_t *__restrict__ pos_d, double *blocks,
                                                          double *Q, int64_t ld) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= nloc) return;
    double d = 0.0;
    for (int64_t k = inc_ptr[cam0 + c]; k < inc_ptr[cam0 + c + 1]; ++k) d += w[inc_edge[k]];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            const double v = (a == b) ? d : 0.0;
            if (DENSE) Q[(size_t)(3 * c + a) * ld + 3 * (cam0 + c) + b] = v;
            else if (pos_d[c] >= 0) blocks[pos_d[c] * 9 + 3 * a + b] = v;
        }
}
// residual of every edge at the point whose scaled rows Y = s.*R are in Y (all cameras, pitch OP): |Y_i - M_e Y_j|_F^2, i.e. the
// edge's term of <Q, Y Y^T> per unit weight
__global__ __launch_bounds__(256) void edge_residual_kernel(int64_t ne, const int32_t *__restrict__ ei, const int32_t *__restrict__ ej,
                                                             const double *__restrict__ M, const double *__restrict__ Y, int o, int OP, double *res) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= ne) return;
    const double *yi = Y + (size_t)ei[e] * 3 * OP, *yj = Y + (size_t)ej[e] * 3 * OP, *m = M + e * 9;
    double acc = 0.0;
    for (int k = 0; k < o; ++k)
        for (int a = 0; a < 3; ++a) {
            const double d = yi[a * OP + k] - (m[3 * a] * yj[k] + m[3 * a + 1] * yj[OP + k] + m[3 * a + 2] * yj[2 * OP + k]);
            acc += d * d;
        }
    res[e] = acc;
}

// ---- XM^2 outlier filter on the device (reference: np.percentile(error, 90) and the index filter, 3_test_colmap_glomap.py:316-324) ----
// err = w .* res
__global__ __launch_bounds__(256) void xm2_error_kernel(int64_t n, const double *__restrict__ w, const double *__restrict__ res, double *__restrict__ err) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < n) err[e] = w[e] * res[e];
}
// One pass of a most-significant-digit radix SELECT over non-negative doubles (their bit patterns order like unsigned integers):
// histogram of the 16 bits at `shift` over the elements whose higher bits equal `prefix`.  Integer atomics: exact and order-free.
__global__ __launch_bounds__(256) void radix_hist_kernel(int64_t n, const double *__restrict__ x, int shift, unsigned long long prefix,
                                                          unsigned int *__restrict__ hist) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const unsigned long long k = (unsigned long long)__double_as_longlong(x[e]);
        if (shift + 16 < 64 && (k >> (shift + 16)) != prefix) continue;
        atomicAdd(hist + ((k >> shift) & 0xffffull), 1u);
    }
}
// wout = err > thr ? 0 : w ; per-block counts of the removed entries (integers)
__global__ __launch_bounds__(256) void xm2_filter_kernel(int64_t n, const double *__restrict__ err, const double *__restrict__ w, double thr,
                                                          double *__restrict__ wout, unsigned int *__restrict__ removed) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const bool rm = err[e] > thr;
    wout[e] = rm ? 0.0 : w[e];
    if (rm && w[e] != 0.0) atomicAdd(removed, 1u);
}

// ----------------------------------------------------------------------------------------------------------------
// launchers (dispatch on the rank o)
// ----------------------------------------------------------------------------------------------------------------
void launch_xm2_error(int64_t n, const double *w, const double *res, double *err, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(xm2_error_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, w, res, err);
    check_launch("xm2_error");
}
void launch_radix_hist(int64_t n, const double *x, int shift, unsigned long long prefix, unsigned int *hist, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(radix_hist_kernel, dim3((unsigned)std::min<int64_t>(1024, (n + 255) / 256)), dim3(256), 0, st, n, x, shift, prefix, hist);
    check_launch("radix_hist");
}
void launch_xm2_filter(int64_t n, const double *err, const double *w, double thr, double *wout, unsigned int *removed, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(xm2_filter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, err, w, thr, wout, removed);
    check_launch("xm2_filter");
}
int qw_grid(int nloc) { return (nloc + kQwWaves - 1) / kQwWaves; }
int bsr_grid(int nloc) { return (nloc + kBsrRows - 1) / kBsrRows; }
int flat_grid(int64_t elems) {
    int64_t g = (elems + 255) / 256;
    if (g < 1) g = 1;
    if (g > 1024) g = 1024;
    return (int)g;
}


// Load policy of the dense stream by the bytes of Q this GPU streams per product.  Up to 310 MB everything stays cacheable: the matrix lives
// in the 256 MB Infinity Cache (+ 32 MB of L2) between products, and with the alternating direction a launch starts where the last one
// ended (non-temporal loads cost 31.5 -> 36.1 us at Venice size, 44.1 -> 48.1 at 302 MB; at 354 MB the default policy collapses: the stream
// evicts its own head).  Beyond, a prefix of kQwResidentMB stays cacheable and the rest streams non-temporally -- cameras >= the returned
// index in the general kernel, chunks from symv_nt_step0 on in the symmetric sweep (profiles/r06_kbench_dense_prefix.txt, general kernel, us:
// 472 MB 81.4 all non-temporal / 80.5 all cacheable / 74.3 prefix; 680 MB 106.9 / 106.1 / 96.0; 1.2 GB 179.4 / 192.6 / 175.2; 13.5 GB 2 003 / - / 1 998).
constexpr int kQwResidentMB = 220;
static int g_qw_nt_override = -1;
void qw_bench_nt(int nt) { g_qw_nt_override = nt; }
static int64_t qw_resident_bytes(size_t total_bytes) {      // bytes of the stream that stay cacheable; >= total: all of it
    if (g_qw_nt_override == 0) return INT64_MAX;
    if (g_qw_nt_override == 1) return 0;
    if (g_qw_nt_override >= 2) return (int64_t)g_qw_nt_override << 20;
    if (g_qw_nt_override <= -2) return (int64_t)(-g_qw_nt_override) << 10;   // (KB: mixed policies on the small matrices of the tests)
    return total_bytes <= ((size_t)310 << 20) ? INT64_MAX : (int64_t)kQwResidentMB << 20;
}
// sliced-ELL stream (xm_sell.hip): bytes of its prefix that stay cacheable when an iteration moves `other` bytes besides it
constexpr int64_t kSellCacheBudgetMB = 230;
int64_t sell_resident_bytes(int64_t stream, int64_t other) {
    if (g_qw_nt_override == 0) return INT64_MAX;
    if (g_qw_nt_override == 1) return 0;
    if (g_qw_nt_override >= 2) return (int64_t)g_qw_nt_override << 20;
    if (g_qw_nt_override <= -2) return (int64_t)(-g_qw_nt_override) << 10;
    return std::max<int64_t>(0, (kSellCacheBudgetMB << 20) - other);
}
static int qw_nt_cam0(int nloc, int64_t ld) {
    const size_t row3 = (size_t)3 * (size_t)ld * sizeof(double);
    const int64_t res = qw_resident_bytes((size_t)nloc * row3);
    return (int)std::min<int64_t>(nloc, res / (int64_t)row3);
}
template <int O, int NSUB>
static void qw_dense_epi(int epi, const double *Q, int64_t ld, const double *W, double alpha, const CamArgs &a0, hipStream_t st) {
    const dim3 g(qw_grid(a0.nloc)), b(256);
    CamArgs a = a0;
    a.nt_cam0 = qw_nt_cam0(a.nloc, ld);
    switch (epi) {
        case EPI_PLAIN: hipLaunchKernelGGL((qw_dense_kernel<O, EPI_PLAIN, NSUB>), g, b, 0, st, Q, ld, W, alpha, a); break;
        case EPI_GRAD: hipLaunchKernelGGL((qw_dense_kernel<O, EPI_GRAD, NSUB>), g, b, 0, st, Q, ld, W, alpha, a); break;
        case EPI_HESS: hipLaunchKernelGGL((qw_dense_kernel<O, EPI_HESS, NSUB>), g, b, 0, st, Q, ld, W, alpha, a); break;
        case EPI_AUTO:
            if constexpr (O >= 3) { hipLaunchKernelGGL((qw_dense_kernel<O, EPI_AUTO, NSUB>), g, b, 0, st, Q, ld, W, alpha, a); break; }
            throw Error(-2, "bad epilogue");
        default: throw Error(-2, "bad epilogue");
    }
}
// split launches (range_mode 1 / 2): plain or gradient epilogue, default tile width, default load policy by size
template <int O>
static void qw_dense_split_o(int epi, const double *Q, int64_t ld, const double *W, double alpha, const CamArgs &a0, hipStream_t st) {
    const dim3 g(qw_grid(a0.nloc)), b(256);
    CamArgs a = a0;
    a.nt_cam0 = qw_nt_cam0(a.nloc, ld);
    if (epi == EPI_PLAIN) hipLaunchKernelGGL((qw_dense_kernel<O, EPI_PLAIN, 2, true>), g, b, 0, st, Q, ld, W, alpha, a);
    else if (epi == EPI_GRAD) hipLaunchKernelGGL((qw_dense_kernel<O, EPI_GRAD, 2, true>), g, b, 0, st, Q, ld, W, alpha, a);
    else throw Error(-2, "split dense product: plain or gradient epilogue only");
}
void launch_qw_dense_split(int o, int epi, const double *Q, int64_t ld, const double *W, double alpha, const CamArgs &a, hipStream_t st) {
    if (a.nloc <= 0) return;
    XM_DISPATCH_O(o, (qw_dense_split_o<O_>(epi, Q, ld, W, alpha, a, st)));
    check_launch("qw_dense_split");
}
int qw_dense_tile_cols() { return 2 * 128; }   // column tile of the split launches

// Small-strip policy: split the columns so that about two workgroups per CU exist, every slice keeping at least two tiles
// (measured at Venice size, profiles/r03_kbench_multi.txt: 223 cameras x6 9.3 us against 19.0 unsplit, 445 cameras x3 14.2 against 22.0)
int qw_dense_split_k(int nloc, int64_t ld) {
    const int g = qw_grid(nloc), ntiles = (int)((ld + 255) / 256);
    int ks = 512 / (g > 0 ? g : 1);
    if (ks > 8) ks = 8;
    if (ks > ntiles / 2) ks = ntiles / 2;
    return ks < 2 ? 1 : ks;
}
template <int O>
static void qw_dense_ks_o(int epi, const double *Q, int64_t ld, const double *W, double alpha, const CamArgs &a, hipStream_t st) {
    const dim3 g(qw_grid(a.nloc), a.ks), b(256);
    switch (epi) {
        case EPI_PLAIN: hipLaunchKernelGGL((qw_dense_ks_kernel<O, EPI_PLAIN>), g, b, 0, st, Q, ld, W, alpha, a); break;
        case EPI_GRAD: hipLaunchKernelGGL((qw_dense_ks_kernel<O, EPI_GRAD>), g, b, 0, st, Q, ld, W, alpha, a); break;
        case EPI_HESS: hipLaunchKernelGGL((qw_dense_ks_kernel<O, EPI_HESS>), g, b, 0, st, Q, ld, W, alpha, a); break;
        case EPI_CERT: if constexpr (O == 1) { hipLaunchKernelGGL((qw_dense_ks_kernel<1, EPI_CERT>), g, b, 0, st, Q, ld, W, alpha, a); break; }
        default: throw Error(-2, "bad epilogue");
    }
}
void launch_qw_dense(int o, int epi, const double *Q, int64_t ld, const double *W, double alpha, const CamArgs &a, hipStream_t st) {
    if (a.nloc <= 0) return;
    if (a.ks > 1 && a.ksum && a.kcount) {
        XM_DISPATCH_O(o, (qw_dense_ks_o<O_>(epi, Q, ld, W, alpha, a, st)));
        check_launch("qw_dense_ks");
        return;
    }
    if (epi == EPI_CERT) {
        if (o != 1) throw Error(-2, "certificate operator needs o == 1");
        CamArgs ac = a;
        ac.nt_cam0 = qw_nt_cam0(a.nloc, ld);
        hipLaunchKernelGGL((qw_dense_kernel<1, EPI_CERT, 2>), dim3(qw_grid(a.nloc)), dim3(256), 0, st, Q, ld, W, alpha, ac);
    } else {
        XM_DISPATCH_O(o, (qw_dense_epi<O_, kQwNsub>(epi, Q, ld, W, alpha, a, st)));
    }
    check_launch("qw_dense");
}

// vertical sweep: steps (two cameras) per chunk K; a workgroup = 4 K steps of one strip.
// SMALL triangles run in ONE residency round -- every workgroup on the chip at once, 2 per CU x 256 CUs = 512 -- and the launch lasts as long as
// its longest wavefront: the shortest chunk whose live workgroups still fit wins, and a plan just beyond one round loses a third
// (profiles/r06_kbench_symv.txt, us per pair at o = 3, folded grid: Venice size K = 5 (503 workgroups) 26.5, 6 26.9, 7 26.7; n = 2 560: K = 9 (565)
// 59.6, 10 49.1, 11 48.7, 12 50.1; n = 3 072: 14 77.8, 15 64.7, 16 64.5, 18 67.8; n = 4 096: 17 121.3, 24 138.7, 26 138.0, 28 (one round) 116.7, 32 117.3).
// LARGE triangles take several rounds: a long chunk amortises the column flush and shortens the reducer's lists, a short one evens out the last
// round; the optimum grows like the square root of the triangle's step count T (n = 8 192, T = 197 k: K = 24 402 us, 34 437, 48 447; n = 13 682,
// T = 550 k: 48 1 123, 64 1 149) -> K = sqrt(T) / 18, 2 <= K <= 64, with the rows dispatched last cut four times finer (symv_plan).
static int64_t symv_live_groups(int64_t nsteps, int64_t nstrips, int64_t k) {
    int64_t live = 0;
    for (int64_t s = 0; s < nstrips; ++s) live += (std::min<int64_t>(nsteps, (s * kSvStrip + kSvStrip + 5) / 6) + 4 * k - 1) / (4 * k);
    return live;
}
int symv_k(int nloc, int64_t ld) {
    const int64_t nsteps = (nloc + 1) / 2, nstrips = (ld + kSvStrip - 1) / kSvStrip;
    int64_t total = 0;
    for (int64_t s = 0; s < nstrips; ++s) total += std::min<int64_t>(nsteps, (s * kSvStrip + kSvStrip + 5) / 6);
    int64_t k = (int64_t)(std::sqrt((double)total) / 18.0 + 0.5);
    k = std::min<int64_t>(64, std::max<int64_t>(2, k));
    if (symv_live_groups(nsteps, nstrips, k) <= 1100) {         // about two rounds or less at the square-root length: make it ONE round
        int64_t ks = 4;
        while (symv_live_groups(nsteps, nstrips, ks) > 504) ++ks;
        k = ks;
    }
    return (int)k;
}
size_t sym_prow_count(int nloc, int64_t ld, int o) { return (size_t)((ld + kSvStrip - 1) / kSvStrip) * 6 * (size_t)((nloc + 1) / 2) * o; }
// The launch's tail: workgroups are dispatched strip group by strip group (left to right) and a chunk of K = 64 steps is a quarter
// of a millisecond at 13.5 GB with only ~4.5 chunks per resident workgroup, so with equal chunks the last "round" runs part empty.
// The strip groups dispatched last (the rightmost 13 %, a quarter of the work) are cut four times finer -- guided self-scheduling by
// construction, no atomics: n = 8192 494 -> 445 us, 13.5 GB at K = 72 1 213 -> 1 144 us (K = 64: 1 145 either way).  A dynamic
// work counter instead (workgroups pulling item numbers) was measured too: no steadier at 13.5 GB and 1.5 x slower at Venice size
// (a barrier and an atomic per item).  The remaining +-6 % between MI355X boxes for one K (1 145 / 1 290 us) is not scheduling noise
// of this kind: it repeats on a box.
static int g_symv_k = 0, g_symv_kf = 0;                   // micro-benchmark / test override of the chunk lengths (xm_bench.h: xm_bench_symv_k)
void symv_bench_k(int k, int kf) { g_symv_k = k; g_symv_kf = kf; }
// K: steps per chunk (a wavefront); Kf: the same in the grid rows >= ysplit (dispatched last: cut finer when the sweep takes several residency
// rounds); nchunks: upper bound of column-sum records per column (sizes Pcol); gx, gy: the folded grid
struct SymvPlan { int K, Kf, ysplit, nchunks, gx, gy; };
static SymvPlan symv_plan(int nloc, int64_t ld) {
    SymvPlan p;
    p.K = g_symv_k > 0 ? g_symv_k : symv_k(nloc, ld);
    const int nsteps = (nloc + 1) / 2, nstrips = (int)((ld + kSvStrip - 1) / kSvStrip);
    p.gy = (nstrips + 1) / 2;
    if (g_symv_k > 0 && g_symv_kf > 0) {                               // forced finer cut of the last rows (tests of the index arithmetic at small sizes)
        p.Kf = std::min(g_symv_kf, p.K);
        p.ysplit = (int)(0.75 * p.gy);
    } else if (p.K >= 8 && symv_live_groups(nsteps, nstrips, p.K) > 1100) {   // several residency rounds
        p.Kf = p.K / 4;
        p.ysplit = (int)(0.75 * p.gy);          // every row of the folded grid holds the same work: the last quarter of it is cut four times finer
    } else {
        p.Kf = p.K; p.ysplit = p.gy;
    }
    auto groups = [&](int s, int k) { return (int)((std::min<int64_t>(nsteps, ((int64_t)s * kSvStrip + kSvStrip + 5) / 6) + 4 * k - 1) / (4 * k)); };
    p.gx = 1;
    for (int y = 0; y < p.gy; ++y) {
        const int k = (y >= p.ysplit) ? p.Kf : p.K, sb = nstrips - 1 - y;
        p.gx = std::max(p.gx, groups(y, k) + (sb != y ? groups(sb, k) : 0));
    }
    p.nchunks = (nsteps + 4 * p.Kf - 1) / (4 * p.Kf);
    return p;
}
void symv_plan_get(int nloc, int64_t ld, int out[4]) {   // host-only view of the plan (CPU test of the index arithmetic)
    const SymvPlan p = symv_plan(nloc, ld);
    out[0] = p.K; out[1] = p.Kf; out[2] = p.ysplit; out[3] = p.nchunks;
}
size_t sym_pcol_count(int nloc, int64_t ld, int o) { return (size_t)symv_plan(nloc, ld).nchunks * (size_t)ld * o; }

// first step (two cameras, six rows) of the symmetric sweep that streams non-temporally: the upper triangle's rows above it hold the resident bytes
static int symv_nt_step0(int nloc, int64_t ld) {
    const double m = 3.0 * nloc;
    const double tri = 8.0 * m * (m + 6.0) / 2.0;            // bytes the sweep streams (triangle incl. the 6-wide diagonal blocks)
    const int64_t res = qw_resident_bytes((size_t)tri);
    const int nsteps = (nloc + 1) / 2;
    if ((double)res >= tri) return nsteps + 1;
    // rows [0, r) of the triangle hold 8 (r m - r^2 / 2) bytes
    const double disc = m * m - 2.0 * (double)res / 8.0;
    const double r = m - std::sqrt(std::max(0.0, disc));
    return (int)std::min<double>(nsteps + 1, std::max(0.0, r / 6.0));
}
template <int O>
static void qw_symv_epi(int epi, const double *Q, int64_t ld, const double *W, double alpha, const CamArgs &a, double *Prow, double *Pcol,
                        hipStream_t st, int rev, unsigned long long *trace = nullptr) {
    const SymvPlan pl = symv_plan(a.nloc, ld);
    const int nstrips = (int)((ld + kSvStrip - 1) / kSvStrip);
    const TcgScal *sc = (epi == EPI_HESS || epi == EPI_AUTO) ? a.scal : (const TcgScal *)nullptr;
    const int by_phase = (epi == EPI_AUTO) ? 1 : 0;
    // the per-camera sum needs: steps per column-sum record (4 K: one record per workgroup), grid row from which the finer cut applies
    const int ys = pl.ysplit, rK = 4 * pl.K, rKf = 4 * pl.Kf, rys = ys;
    const dim3 gs(pl.gx, pl.gy);
    const int nt0 = symv_nt_step0(a.nloc, ld);
    if (trace) {
        if constexpr (O == 3 || O == 4) hipLaunchKernelGGL((qw_symv_kernel<O, true>), gs, dim3(256), 0, st, Q, ld, W, a.nloc, pl.K, pl.Kf, ys, nt0, sc, Prow, Pcol, trace, rev, by_phase);
    } else hipLaunchKernelGGL((qw_symv_kernel<O>), gs, dim3(256), 0, st, Q, ld, W, a.nloc, pl.K, pl.Kf, ys, nt0, sc, Prow, Pcol, trace, rev, by_phase);
    const dim3 g(qw_grid(a.nloc)), b(256);
    switch (epi) {
        case EPI_PLAIN: hipLaunchKernelGGL((symv_reduce_kernel<O, EPI_PLAIN>), g, b, 0, st, Prow, Pcol, ld, nstrips, rK, rKf, rys, alpha, a); break;
        case EPI_GRAD: hipLaunchKernelGGL((symv_reduce_kernel<O, EPI_GRAD>), g, b, 0, st, Prow, Pcol, ld, nstrips, rK, rKf, rys, alpha, a); break;
        case EPI_HESS: hipLaunchKernelGGL((symv_reduce_kernel<O, EPI_HESS>), g, b, 0, st, Prow, Pcol, ld, nstrips, rK, rKf, rys, alpha, a); break;
        case EPI_AUTO:
            if constexpr (O >= 3) { hipLaunchKernelGGL((symv_reduce_kernel<O, EPI_AUTO>), g, b, 0, st, Prow, Pcol, ld, nstrips, rK, rKf, rys, alpha, a); break; }
            throw Error(-2, "bad epilogue");
        case EPI_CERT:   // certificate operator (rank-1 input): the Lanczos products of a large dense Q at half the traffic too
            if constexpr (O == 1) { hipLaunchKernelGGL((symv_reduce_kernel<1, EPI_CERT>), g, b, 0, st, Prow, Pcol, ld, nstrips, rK, rKf, rys, alpha, a); break; }
            throw Error(-2, "certificate operator needs o == 1");
        default: throw Error(-2, "bad epilogue");
    }
}
int symv_trace_slots() { return kSvTraceSlots; }
// micro-benchmark: one traced product (o = 3 or 4); trace = [grid.y * grid.x * 4 wavefronts][kSvTraceSlots] timestamps, grid returned
void launch_qw_sym_traced(int o, const double *Q, int64_t ld, const double *W, const CamArgs &a, double *Prow, double *Pcol, unsigned long long *trace,
                          int grid[2], hipStream_t st) {
    const SymvPlan pl = symv_plan(a.nloc, ld);
    grid[0] = pl.gx; grid[1] = pl.gy;
    if (trace == nullptr) return;
    if (o == 3) qw_symv_epi<3>(EPI_PLAIN, Q, ld, W, 1.0, a, Prow, Pcol, st, 0, trace);
    else if (o == 4) qw_symv_epi<4>(EPI_PLAIN, Q, ld, W, 1.0, a, Prow, Pcol, st, 0, trace);
    else throw Error(-2, "traced symmetric product: o = 3 or 4");
    check_launch("qw_symv traced");
}

// symmetric half-traffic product (o in 1, 3..5); Prow: sym_prow_count() doubles, Pcol: sym_pcol_count() doubles.  rev (0 / 1): the direction
// of the sweep inside every chunk.  A launch that starts with the steps the previous launch ended with finds them in the XCDs' L2 (4 MB each:
// a quarter of a Venice-size sweep; block b runs on XCD b mod 8 in every launch), so callers alternate it between consecutive products --
// by a number both runs of the same solve agree on (the tCG iteration, the Lanczos step), never by a launch count that run-ahead no-ops would shift:
// the direction changes the order of the column sums, i.e. the last bits.
void launch_qw_sym(int o, int epi, const double *Q, int64_t ld, const double *W, double alpha, const CamArgs &a, double *Prow,
                   double *Pcol, hipStream_t st, int rev) {
    if (a.nloc <= 0) return;
    switch (o) {
        case 1: qw_symv_epi<1>(epi, Q, ld, W, alpha, a, Prow, Pcol, st, rev & 1); break;
        case 3: qw_symv_epi<3>(epi, Q, ld, W, alpha, a, Prow, Pcol, st, rev & 1); break;
        case 4: qw_symv_epi<4>(epi, Q, ld, W, alpha, a, Prow, Pcol, st, rev & 1); break;
        case 5: qw_symv_epi<5>(epi, Q, ld, W, alpha, a, Prow, Pcol, st, rev & 1); break;
        default: throw Error(-2, "symmetric product is instantiated for o = 1, 3..5");
    }
    check_launch("qw_symv");
}
// EXACT symmetry of a matrix that is spread over several ranks as row strips (no rank sees an entry and its mirror image): every rank sums
//     g(i, j) * f(i, j) * bits(Q[i][j])   modulo 2^64 over its strip,   g = +1 above the diagonal, -1 below, 0 on it,
// with f a 64-bit hash of the UNORDERED pair {i, j} and bits() the IEEE pattern (-0 counted as +0).  Integer arithmetic modulo 2^64 is
// associative and commutative, so the strips' sums add up -- in any order, exactly -- to sum_{i<j} f(i,j) (bits(Q_ij) - bits(Q_ji)): zero for
// a symmetric matrix, non-zero for any other one except with probability ~2^-64 per differing pair.  out[2 b] = workgroup b's sum,
// out[2 b + 1] = 1 when it met a NaN (never symmetric, like launch_asym).
__global__ __launch_bounds__(256) void symhash_kernel(const double *__restrict__ Q, int64_t ld, int64_t row0, int64_t nrows, int64_t m,
                                                       unsigned long long *out) {
    __shared__ unsigned long long sh[2][4];
    unsigned long long acc = 0ull, bad = 0ull;
    const int64_t total = nrows * m;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t lr = e / m, c = e - lr * m, r = row0 + lr;
        if (r >= m || r == c) continue;
        const double v = Q[lr * ld + c];
        if (v != v) bad = 1ull;
        unsigned long long b = (v == 0.0) ? 0ull : (unsigned long long)__double_as_longlong(v);
        unsigned long long z = (unsigned long long)((r < c) ? r : c) * 0x9E3779B97F4A7C15ull + (unsigned long long)((r < c) ? c : r);   // splitmix64 of the pair
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        const unsigned long long t = (z | 1ull) * b;
        acc += (r < c) ? t : (0ull - t);
    }
    // wavefront / workgroup sums in integer arithmetic: order-independent
    for (int off = 32; off > 0; off >>= 1) { acc += __shfl_xor(acc, off, 64); bad |= __shfl_xor(bad, off, 64); }
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = acc; sh[1][threadIdx.x >> 6] = bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
        out[2 * blockIdx.x + 1] = sh[1][0] | sh[1][1] | sh[1][2] | sh[1][3];
    }
}
void launch_symhash(const double *Q, int64_t ld, int64_t row0, int64_t nrows, int64_t m, unsigned long long *out, int grid, hipStream_t st) {
    hipLaunchKernelGGL(symhash_kernel, dim3(grid), dim3(256), 0, st, Q, ld, row0, nrows, m, out);
    check_launch("symhash");
}
void launch_asym(const double *Q, int64_t ld, int64_t m, double *out, int grid, hipStream_t st) {
    hipLaunchKernelGGL(asym_kernel, dim3(grid), dim3(256), 0, st, Q, ld, m, out);
    check_launch("asym");
}

template <int O, int VAR>
static void qw_bsr3_epi(int epi, const int64_t *rp, const int32_t *ci, const double *bl, const double *W, const int4 *ri, double alpha,
                        const CamArgs &a, hipStream_t st) {
    const dim3 g((a.nloc + kBsrRows - 1) / kBsrRows), b(256);
    switch (epi) {
        case EPI_PLAIN: hipLaunchKernelGGL((qw_bsr3_kernel<O, EPI_PLAIN, VAR>), g, b, 0, st, rp, ci, bl, W, ri, alpha, a); break;
        case EPI_GRAD: hipLaunchKernelGGL((qw_bsr3_kernel<O, EPI_GRAD, VAR>), g, b, 0, st, rp, ci, bl, W, ri, alpha, a); break;
        case EPI_HESS: hipLaunchKernelGGL((qw_bsr3_kernel<O, EPI_HESS, VAR>), g, b, 0, st, rp, ci, bl, W, ri, alpha, a); break;
        case EPI_AUTO:
            if constexpr (O >= 3) { hipLaunchKernelGGL((qw_bsr3_kernel<O, EPI_AUTO, VAR>), g, b, 0, st, rp, ci, bl, W, ri, alpha, a); break; }
            throw Error(-2, "bad epilogue");
        default: throw Error(-2, "bad epilogue");
    }
}
// nb: stored blocks of these nloc rows, or 0 when the caller does not know -- then, and up to the size the dense kernel's rule keeps cacheable, the
// blocks are read with the default policy; a larger matrix streams all of them non-temporally (a cacheable prefix as in the dense kernel
// loses here: 402 MB 133.5 us with a 220 MB prefix, 127.4 all non-temporal, 136.5 all cacheable)
static int bsr_nt_cam0(int nloc, int64_t nb) {
    const size_t bytes = (size_t)std::max<int64_t>(nb, 0) * 76;
    return qw_resident_bytes(bytes) >= (int64_t)bytes ? nloc : 0;
}
// The 16 rows of every workgroup (cameras 16 b .. 16 b + 15) ordered by their number of 16-block windows, most windows first, camera order among
// equals (stable): entry = {first block lo, hi, length, camera}.  A wavefront then carries four rows of similar length out of its workgroup's
// sixteen, and the workgroup still owns sixteen CONSECUTIVE cameras: the epilogue's vectors stay contiguous per workgroup (binning over the whole
// matrix is 0.9 us faster in the bare product at 13 682 cameras and loses all of it again in the Hessian epilogue, whose five vectors it scatters)
void bsr_build_rowinfo(const int64_t *rowptr_host, int nloc, std::vector<int4> &out) {
    out.resize((size_t)std::max(nloc, 0));
    for (int c0 = 0; c0 < nloc; c0 += kBsrRows) {
        const int m = std::min(kBsrRows, nloc - c0);
        int order[kBsrRows];
        for (int j = 0; j < m; ++j) order[j] = c0 + j;
        std::stable_sort(order, order + m, [&](int x, int y) {
            return (rowptr_host[x + 1] - rowptr_host[x] + 15) / 16 > (rowptr_host[y + 1] - rowptr_host[y] + 15) / 16;
        });
        for (int j = 0; j < m; ++j) {
            const int i = order[j];
            const int64_t b0 = rowptr_host[i], len = rowptr_host[i + 1] - b0;
            if (len >= ((int64_t)1 << 31)) throw Error(-2, "block-CSR row with 2^31 or more blocks");
            out[(size_t)c0 + j] = make_int4((int)(unsigned)(b0 & 0xffffffffll), (int)(unsigned)((unsigned long long)b0 >> 32), (int)len, i);
        }
    }
}
void launch_qw_bsr3(int o, int epi, const int64_t *rp, const int32_t *ci, const double *bl, const double *W, double alpha,
                    const CamArgs &a0, hipStream_t st, int64_t nb, const int4 *ri) {
    if (a0.nloc <= 0) return;
    CamArgs a = a0;
    a.nt_cam0 = bsr_nt_cam0(a.nloc, nb);
    if (epi == EPI_CERT) {
        if (o != 1) throw Error(-2, "certificate operator needs o == 1");
        hipLaunchKernelGGL((qw_bsr3_kernel<1, EPI_CERT, 0>), dim3((a.nloc + kBsrRows - 1) / kBsrRows), dim3(256), 0, st, rp, ci, bl, W, ri, alpha, a);
    } else {
        // o <= 6: blocks AND the gathered records of W staged through LDS (VAR 2); above that the records no longer fit: blocks only (VAR 1)
        switch (o) {
            case 3: qw_bsr3_epi<3, 2>(epi, rp, ci, bl, W, ri, alpha, a, st); break;
            case 4: qw_bsr3_epi<4, 2>(epi, rp, ci, bl, W, ri, alpha, a, st); break;
            case 5: qw_bsr3_epi<5, 2>(epi, rp, ci, bl, W, ri, alpha, a, st); break;
            case 6: qw_bsr3_epi<6, 2>(epi, rp, ci, bl, W, ri, alpha, a, st); break;
            case 1: qw_bsr3_epi<1, 1>(epi, rp, ci, bl, W, ri, alpha, a, st); break;
            case 7: qw_bsr3_epi<7, 1>(epi, rp, ci, bl, W, ri, alpha, a, st); break;
            case 8: qw_bsr3_epi<8, 1>(epi, rp, ci, bl, W, ri, alpha, a, st); break;
            case 9: qw_bsr3_epi<9, 1>(epi, rp, ci, bl, W, ri, alpha, a, st); break;
            case 10: qw_bsr3_epi<10, 1>(epi, rp, ci, bl, W, ri, alpha, a, st); break;
            default: throw Error(-2, "rank o must be 1 or 3..10, got " + std::to_string(o));
        }
    }
    check_launch("qw_bsr3");
}

void launch_transpose_pad(const double *src, int64_t lds, int64_t rows, int64_t cols, double *dst, int64_t ldd, hipStream_t st) {
    const dim3 g((unsigned)((ldd + 31) / 32), (unsigned)((rows + 31) / 32));
    hipLaunchKernelGGL(transpose_pad_kernel, g, dim3(256), 0, st, src, lds, rows, cols, dst, ldd);
    check_launch("transpose_pad");
}
void launch_dense_from_bsr(const int64_t *rp, const int32_t *ci, const double *bl, int64_t nloc, int64_t cam0, double *dst,
                           int64_t ldd, hipStream_t st) {
    hipLaunchKernelGGL(dense_from_bsr_kernel, dim3((unsigned)nloc), dim3(252), 0, st, rp, ci, bl, nloc, cam0, dst, ldd);
    check_launch("dense_from_bsr");
}

void launch_scale_rows(int o, int nloc, const double *R, const double *s, double *Wloc, hipStream_t st) {
    XM_DISPATCH_O(o, hipLaunchKernelGGL((scale_rows_kernel<O_>), dim3(flat_grid((int64_t)nloc * 3 * pitch_of(O_))), dim3(256), 0, st,
                                        nloc, R, s, Wloc));
    check_launch("scale_rows");
}
void launch_tcg_init(int o, int nloc, const double *rgR, const double *rgs, const double *R, const double *s, double *rR, double *rs,
                     double *pR, double *ps, double *vR, double *vs, double *HvR, double *Hvs, double *Wloc, TcgScal *scal0,
                     double rr, double delta, unsigned long long *hstat, hipStream_t st, double *Wpad, int seq, const SpecCtl *spec) {
    XM_DISPATCH_O(o, hipLaunchKernelGGL((tcg_init_kernel<O_>), dim3(flat_grid((int64_t)nloc * 3 * pitch_of(O_))), dim3(256), 0, st,
                                        nloc, rgR, rgs, R, s, rR, rs, pR, ps, vR, vs, HvR, Hvs, Wloc, scal0, rr, delta, hstat, Wpad, seq, spec));
    check_launch("tcg_init");
}
void launch_cg_step(int o, int nloc, const TcgScal *scal_cur, TcgScal *scal_next, const double *parts, int nA_loc, int nB_loc, int world,
                    const double *HpR, const double *Hps, const double *R, const double *s, double *pR,
                    const double *ps_cur, double *ps_next, double *vR, double *vs, double *HvR, double *Hvs, double *rR, const double *rs_cur,
                    double *rs_next, double *Wloc, double *partsB_out, unsigned long long *hstat, int b_off, int64_t mat, double *Afull,
                    double *Wfull, int grouping, const PeerXchg &xchg, hipStream_t st, double *Wpad) {
    // grid = nB_loc: one |r|^2 partial sum per workgroup (Context::tcg_blocks: capped when the launch waits for its peers inside the kernel)
    XM_DISPATCH_O(o, hipLaunchKernelGGL((cg_step_kernel<O_>), dim3(nB_loc), dim3(256), 0, st, nloc,
                                        scal_cur, scal_next, parts, nA_loc, nB_loc, world, HpR, Hps, R, s, pR, ps_cur, ps_next, vR,
                                        vs, HvR, Hvs, rR, rs_cur, rs_next, Wloc, partsB_out, hstat, b_off, mat, Afull, Wfull, grouping, xchg, Wpad));
    check_launch("cg_step");
}
void launch_outer_finalize(const double *partsA, int nA_loc, int world, const double *partsM, int nM, const TcgScal *scal, double *hres,
                           unsigned long long seq, int grouping, hipStream_t st, const OuterArgs *oa, SpecCtl *spec_out) {
    OuterArgs z;
    std::memset(&z, 0, sizeof(z));
    hipLaunchKernelGGL(outer_finalize_kernel, dim3(1), dim3(256), 0, st, partsA, nA_loc, world, partsM, nM, scal, hres, seq, grouping, oa ? *oa : z,
                       (oa != nullptr) ? spec_out : (SpecCtl *)nullptr);
    check_launch("outer_finalize");
}
void launch_outer_step(int o, int polar, const OuterStepArgs &A, int grid, hipStream_t st) {
    switch (o) {
#define XM_OS_CASE(OO) case OO: if (polar) hipLaunchKernelGGL((outer_step_kernel<OO, 1>), dim3(grid), dim3(256), 0, st, A); \
                                else hipLaunchKernelGGL((outer_step_kernel<OO, 0>), dim3(grid), dim3(256), 0, st, A); break;
        XM_OS_CASE(3) XM_OS_CASE(4) XM_OS_CASE(5) XM_OS_CASE(6) XM_OS_CASE(7) XM_OS_CASE(8) XM_OS_CASE(9) XM_OS_CASE(10)
#undef XM_OS_CASE
        default: throw Error(-2, "device-driven outer iteration: rank o must be 3..10, got " + std::to_string(o));
    }
    check_launch("outer_step");
}
int retract_grid(int nloc) { return (nloc + 255) / 256; }
// the outer iteration's retraction: + model decrease of the step (partial sums parts[retract_grid(nloc)]) + the padded copy of the product input
void launch_retract_model(int o, int nloc, int cam0, const double *R, const double *s, const double *vR, const double *vs, double *Rout, double *sout,
                          double *Wloc, double *Wpad, const double *HvR, const double *Hvs, const double *rgR, const double *rgs, double *parts,
                          hipStream_t st, int polar) {
    const ModelArgs mv = {HvR, Hvs, rgR, rgs, parts};
    const dim3 g(retract_grid(nloc)), b(256);
    if (HvR == nullptr) {   // XM_FLAG_MODEL_RECURRENCE: the model value is in the tCG's scalar block
        if (polar) { XM_DISPATCH_O(o, hipLaunchKernelGGL((retract_kernel<O_, 1, false>), g, b, 0, st, nloc, cam0, R, s, vR, vs, 1.0, Rout, sout, Wloc, Wpad, mv)); }
        else { XM_DISPATCH_O(o, hipLaunchKernelGGL((retract_kernel<O_, 0, false>), g, b, 0, st, nloc, cam0, R, s, vR, vs, 1.0, Rout, sout, Wloc, Wpad, mv)); }
        check_launch("retract");
        return;
    }
    if (polar) { XM_DISPATCH_O(o, hipLaunchKernelGGL((retract_kernel<O_, 1, true>), g, b, 0, st, nloc, cam0, R, s, vR, vs, 1.0, Rout, sout, Wloc, Wpad, mv)); }
    else { XM_DISPATCH_O(o, hipLaunchKernelGGL((retract_kernel<O_, 0, true>), g, b, 0, st, nloc, cam0, R, s, vR, vs, 1.0, Rout, sout, Wloc, Wpad, mv)); }
    check_launch("retract_model");
}
void launch_retract(int o, int nloc, int cam0, const double *R, const double *s, const double *D, const double *ds, double t,
                    double *Rout, double *sout, double *Wloc, hipStream_t st, int polar) {
    const ModelArgs mv = {nullptr, nullptr, nullptr, nullptr, nullptr};
    if (polar == 2) {   // the quad-per-camera form of the MGS-QR retraction (measured alternative)
        XM_DISPATCH_O(o, hipLaunchKernelGGL((retract_quad_kernel<O_>), dim3((nloc + 63) / 64), dim3(256), 0, st, nloc, cam0, R, s, D, ds, t,
                                            Rout, sout, Wloc));
    } else if (polar) {
        XM_DISPATCH_O(o, hipLaunchKernelGGL((retract_kernel<O_, 1>), dim3((nloc + 255) / 256), dim3(256), 0, st, nloc, cam0, R, s, D, ds, t,
                                            Rout, sout, Wloc, (double *)nullptr, mv));
    } else {
        XM_DISPATCH_O(o, hipLaunchKernelGGL((retract_kernel<O_, 0>), dim3((nloc + 255) / 256), dim3(256), 0, st, nloc, cam0, R, s, D, ds, t,
                                            Rout, sout, Wloc, (double *)nullptr, mv));
    }
    check_launch("retract");
}
void launch_cert_prepare(int o, int nloc, int cam0, double lam, const double *QsR, const double *R, const double *s, double *Lam,
                         double *dz, double *parts, hipStream_t st) {
    XM_DISPATCH_O(o, hipLaunchKernelGGL((cert_prepare_kernel<O_>), dim3((nloc + 255) / 256), dim3(256), 0, st, nloc, cam0, lam, QsR, R,
                                        s, Lam, dz, parts));
    check_launch("cert_prepare");
}
void launch_recover_gram(int64_t n, int r, const double *R, const double *s, double *parts, int grid, hipStream_t st) {
    hipLaunchKernelGGL(recover_gram_kernel, dim3(grid), dim3(256), 0, st, n, r, R, s, parts);
    check_launch("recover_gram");
}
void launch_recover_project(int64_t n, int r, const double *R, const double *s, const double *V, double *rot, double *scale, int *negcount,
                            hipStream_t st, int variant) {
    if (variant == 1) hipLaunchKernelGGL(recover_project_wave_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, n, r, R, s, V, rot, scale, negcount);
    else hipLaunchKernelGGL(recover_project_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, r, R, s, V, rot, scale, negcount);
    check_launch("recover_project");
}
void launch_negate(double *x, int64_t len, hipStream_t st) {
    hipLaunchKernelGGL(negate_kernel, dim3(flat_grid(len)), dim3(256), 0, st, x, len);
    check_launch("negate");
}
// segments of ~4096 elements (16 strided steps per thread): with 16384 the 41 k-element Lanczos vectors of the Final-13682 certificate were two
// segments per column -- 66 workgroups walking 80 steps each, 21 us per call, three calls per Lanczos step (profiles/r05_trace_summary_rome_bsr.txt)
int dots_multi_segments(int64_t len) { return (int)std::min<int64_t>(64, std::max<int64_t>(1, len / 4096)); }
// scratch: m * dots_multi_segments(len) doubles (may be null when that is 1 segment: short vectors keep the single-kernel sum)
void launch_dots_multi(const double *V, int64_t ldv, int m, const double *w, int64_t len, double *c, double *scratch, hipStream_t st) {
    if (m <= 0) return;
    const int nseg = dots_multi_segments(len);
    if (nseg <= 1 || !scratch) {
        hipLaunchKernelGGL(dots_multi_kernel, dim3(m), dim3(256), 0, st, V, ldv, w, len, c);
    } else {
        const int64_t seg = ((len + nseg - 1) / nseg + 255) / 256 * 256;
        hipLaunchKernelGGL(dots_multi_seg_kernel, dim3(m, nseg), dim3(256), 0, st, V, ldv, w, len, seg, scratch);
        hipLaunchKernelGGL(dots_multi_fin_kernel, dim3((m + 255) / 256), dim3(256), 0, st, scratch, m, nseg, c);
    }
    check_launch("dots_multi");
}
// the per-segment partial dots only (scratch[m][nseg], nseg = dots_multi_segments(len)); the sums are taken by the kernels that use them
void launch_dots_multi_parts(const double *V, int64_t ldv, int m, const double *w, int64_t len, double *scratch, hipStream_t st) {
    if (m <= 0) return;
    const int nseg = dots_multi_segments(len);
    const int64_t seg = ((len + nseg - 1) / nseg + 255) / 256 * 256;
    hipLaunchKernelGGL(dots_multi_seg_kernel, dim3(m, nseg), dim3(256), 0, st, V, ldv, w, len, seg, scratch);
    check_launch("dots_multi_parts");
}
bool lz_fused_ok(int m) { return m <= kLzMaxCols; }
// w -= V c with c = the sums of `parts` (launch_dots_multi_parts); c_out[0..m) keeps them; second pass: alpha_j = c_prev[m-1] + c[m-1]
void launch_sub_vc_fin(double *w, const double *V, int64_t ldv, const double *parts, int m, int64_t len, double *c_out, const double *c_prev,
                       double *alpha_j, hipStream_t st) {
    if (m <= 0 || m > kLzMaxCols) throw Error(-2, "launch_sub_vc_fin: bad column count");
    hipLaunchKernelGGL(sub_vc_fin_kernel, dim3(flat_grid(len)), dim3(256), 0, st, w, V, ldv, parts, dots_multi_segments(len), m, len, c_out, c_prev, alpha_j);
    check_launch("sub_vc_fin");
}
void launch_lz_next_fin(double *dst, const double *w, const double *parts, double *beta_j, int64_t len, hipStream_t st) {
    hipLaunchKernelGGL(lz_next_fin_kernel, dim3(flat_grid(len)), dim3(256), 0, st, dst, w, parts, dots_multi_segments(len), beta_j, len);
    check_launch("lz_next_fin");
}
void launch_sub_vc(double *w, const double *V, int64_t ldv, const double *c, int m, int64_t len, hipStream_t st) {
    if (m <= 0) return;
    hipLaunchKernelGGL(sub_vc_kernel, dim3(flat_grid(len)), dim3(256), 0, st, w, V, ldv, c, m, len);
    check_launch("sub_vc");
}
void launch_gemv_n(double *y, const double *V, int64_t ldv, const double *c, int m, int64_t len, hipStream_t st) {
    hipLaunchKernelGGL(gemv_n_kernel, dim3(flat_grid(len)), dim3(256), 0, st, y, V, ldv, c, m, len);
    check_launch("gemv_n");
}
void launch_lz_alpha(const double *c1j, const double *c2j, double *alpha_j, hipStream_t st) {
    hipLaunchKernelGGL(lz_alpha_kernel, dim3(1), dim3(64), 0, st, c1j, c2j, alpha_j);
    check_launch("lz_alpha");
}
void launch_lz_next(double *dst, const double *w, const double *ww, double *beta_j, int64_t len, hipStream_t st) {
    hipLaunchKernelGGL(lz_next_kernel, dim3(flat_grid(len)), dim3(256), 0, st, dst, w, ww, beta_j, len);
    check_launch("lz_next");
}
// plain copy in 4-byte words (staging through host-mapped memory without the copy engine, see Context::to_host)
__global__ __launch_bounds__(256) void copy_words_kernel(unsigned int *__restrict__ dst, const unsigned int *__restrict__ src, size_t nwords) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
void launch_copy_words(void *dst, const void *src, size_t bytes, hipStream_t st) {
    if (bytes == 0) return;
    if (bytes % 4 != 0) throw Error(-2, "launch_copy_words: size must be a multiple of 4 bytes");
    const size_t nw = bytes / 4;
    const int grid = (int)std::min<size_t>(1024, (nw + 255) / 256);
    hipLaunchKernelGGL(copy_words_kernel, dim3(grid), dim3(256), 0, st, static_cast<unsigned int *>(dst), static_cast<const unsigned int *>(src), nw);
    check_launch("copy_words");
}
void launch_scale_copy(double *dst, const double *src, double a, int64_t len, hipStream_t st) {
    hipLaunchKernelGGL(scale_copy_kernel, dim3(flat_grid(len)), dim3(256), 0, st, dst, src, a, len);
    check_launch("scale_copy");
}


void launch_edge_locate(int64_t ne, const int32_t *ei, const int32_t *ej, int cam0, int nloc, const int64_t *rowptr, const int32_t *colidx,
                        int64_t *pos_ij, int64_t *pos_ji, int64_t *pos_d, hipStream_t st) {
    if (ne > 0) hipLaunchKernelGGL(edge_locate_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st, ne, ei, ej, cam0, nloc, rowptr, colidx, pos_ij, pos_ji);
    hipLaunchKernelGGL(diag_locate_kernel, dim3((nloc + 255) / 256), dim3(256), 0, st, nloc, cam0, rowptr, colidx, pos_d);
    check_launch("edge_locate");
}
void launch_edge_write(bool dense, int64_t ne, const int32_t *ei, const int32_t *ej, const double *M, const double *w, int cam0, int nloc,
                       const int64_t *inc_ptr, const int32_t *inc_edge, const int64_t *pos_ij, const int64_t *pos_ji, const int64_t *pos_d,
                       double *blocks, double *Q, int64_t ld, hipStream_t st) {
    const dim3 ge((unsigned)((ne + 255) / 256)), gc((nloc + 255) / 256), b(256);
    if (dense) {
        if (ne > 0) hipLaunchKernelGGL((edge_write_kernel<true>), ge, b, 0, st, ne, ei, ej, M, w, cam0, nloc, pos_ij, pos_ji, blocks, Q, ld);
        hipLaunchKernelGGL((diag_write_kernel<true>), gc, b, 0, st, nloc, cam0, inc_ptr, inc_edge, w, pos_d, blocks, Q, ld);
    } else {
        if (ne > 0) hipLaunchKernelGGL((edge_write_kernel<false>), ge, b, 0, st, ne, ei, ej, M, w, cam0, nloc, pos_ij, pos_ji, blocks, Q, ld);
        hipLaunchKernelGGL((diag_write_kernel<false>), gc, b, 0, st, nloc, cam0, inc_ptr, inc_edge, w, pos_d, blocks, Q, ld);
    }
    check_launch("edge_write");
}
void launch_edge_residual(int64_t ne, const int32_t *ei, const int32_t *ej, const double *M, const double *Y, int o, int OP, double *res, hipStream_t st) {
    if (ne <= 0) return;
    hipLaunchKernelGGL(edge_residual_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st, ne, ei, ej, M, Y, o, OP, res);
    check_launch("edge_residual");
}

}  // namespace xm
