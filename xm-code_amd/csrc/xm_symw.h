// xm_symw.h — half-traffic product for a SYMMETRIC dense Q under the camera row partition (N > 1 GPUs).
//
// The single-GPU symmetric product (xm_kernels.hip: qw_symv_kernel) sweeps the upper triangle of the WHOLE matrix; a rank of a multi-GPU run
// holds a row strip, and the upper triangle cut into strips is hopelessly unbalanced (rank 0's trapezoid is almost its whole strip, the last
// rank's a small triangle: 15 : 1 at N = 8).  Here every row reads a CYCLIC HALF WINDOW instead: with the matrix cut into T steps of 6 rows /
// columns, row step t uses the 6 x 6 block (t, u) -- in both directions, y_t += B w_u and y_u += B^T w_t -- iff
//         0 < (u - t) mod T < Th        or        (u - t) mod T == Th  and  t < u   (the tie exists for even T only),    Th = ceil(T / 2),
// plus its own diagonal block (row direction only).  Of every pair of mirror blocks exactly one is used (symw_use below; CPU test
// tests/test_symw_plan.py), every row reads half of its columns whatever its position, so equal camera ranges give equal work: each rank
// streams HALF of its strip.  The column-direction sums belong to cameras of other ranks: every rank adds its own up per column
// (symw_colsum_kernel), the ranks all-gather those vectors (3 n o doubles each: 985 KB at 13 682 cameras, o = 3) and every rank adds, per
// camera, its row-direction partial sums and the ranks' column sums in a fixed order before the fused epilogue (symw_reduce_kernel).
// Replaces, like the other product kernels, cublasDgemm on the symmetric C of XM_main.cu:191 (Dense/matmul.h:42-87).
//
// Work decomposition: the vertical sweep of qw_symv_kernel (a wavefront owns a strip of 256 columns and walks down it in steps of 6 rows,
// column sums in registers, row sums through LDS).  Which (strip, step) pairs hold any used block is decided by ONE predicate (symw_any)
// shared by the host planner, the sweep (it takes the unmasked fast path when symw_full says so) and the reducer.
#pragma once

#include <cstdint>
#include <vector>

#include "xm_solver.h"

namespace xm {

constexpr int kSwStrip = 256;

struct SymwGeom {   // by value into the kernels
    int T;          // steps of the whole (padded) matrix: 3 * ntot / 6
    int Th;         // ceil(T / 2)
    int tie;        // 1: T even (a pair at distance exactly Th is used by the smaller step)
    int t0;         // global step of this rank's local step 0 (cam0 / 2)
    int nsteps;     // local steps (nloc / 2)
    int nstrips;    // strips of 256 columns that hold real columns: ceil(6 T / 256)
};

// block (t, u), t != u: used by row step t?
__host__ __device__ inline bool symw_use(const SymwGeom &g, int t, int u) {
    int d = u - t;
    if (d < 0) d += g.T;
    return (d > 0 && d < g.Th) || (g.tie && d == g.Th && t < u);
}
// column steps of strip s: [lo, hi] (hi < T); false: the strip holds no real column
__host__ __device__ inline bool symw_strip_steps(const SymwGeom &g, int s, int &lo, int &hi) {
    lo = (s * kSwStrip) / 6;
    hi = (s * kSwStrip + kSwStrip - 1) / 6;
    if (hi > g.T - 1) hi = g.T - 1;
    return lo <= g.T - 1;
}
// may row step t use anything of strip s?  (conservative: a true with nothing used only costs a step of zeros; host, sweep and reducer agree)
__host__ __device__ inline bool symw_any(const SymwGeom &g, int t, int s) {
    int lo, hi;
    if (!symw_strip_steps(g, s, lo, hi)) return false;
    int dlo = lo - t, dhi = hi - t;
    if (dlo < 0) dlo += g.T;
    if (dhi < 0) dhi += g.T;
    return (dlo > dhi) || dlo <= g.Th;   // the strip holds the diagonal block of t (or wraps around it), or starts inside the window
}
// is every column of strip s used by row step t in BOTH directions (no diagonal block, no window edge, no padding column)?
__host__ __device__ inline bool symw_full(const SymwGeom &g, int t, int s) {
    int lo, hi;
    if (!symw_strip_steps(g, s, lo, hi)) return false;
    if ((int64_t)s * kSwStrip + kSwStrip > (int64_t)6 * g.T) return false;
    int dlo = lo - t, dhi = hi - t;
    if (dlo < 0) dlo += g.T;
    if (dhi < 0) dhi += g.T;
    return dlo <= dhi && dlo > 0 && dhi < g.Th;
}

struct SymwItem { int s, jb, je, pad; };   // strip, local steps [jb, je): every one of them has symw_any

struct SymwPlan {
    SymwGeom g;
    int K = 0;                          // most steps of an item
    std::vector<SymwItem> items;        // sorted by strip, then by step: the items of a strip are consecutive
    std::vector<int32_t> strip_ptr;     // nstrips + 1: items of strip s are [strip_ptr[s], strip_ptr[s+1])
};
// ntot, nloc, cam0: cameras in the padded numbering (all even); K <= 0: automatic
void symw_plan_build(int64_t ntot, int nloc, int cam0, int K, SymwPlan &out);

size_t symw_prow_count(const SymwPlan &p, int o);    // doubles of the row-direction partial sums
size_t symw_pcol_count(const SymwPlan &p, int o);    // doubles of the per-item column sums
size_t symw_csum_count(int64_t ntot, int o);         // doubles of one rank's column-sum vector (the all-gathered message)

// device-resident plan + buffers of one rank
class SymwProduct {
public:
    SymwProduct(int64_t ntot, int nloc, int cam0, int64_t ld, hipStream_t st);
    void ensure(int omax, int world);                  // buffers for every rank o <= omax (grow-only: no free between two collectives)
    // main sweep + column sums of this rank -> csum_all() + rank * csum_count(o)
    void sweep(int o, const double *Q, const double *W, const TcgScal *scal, int rank, hipStream_t st);
    // per-camera sum (row partials + every rank's column sums, fixed order) + fused epilogue
    void reduce(int o, int epi, double alpha, const CamArgs &a, int world, hipStream_t st);
    double *csum_all() { return csum_.p; }
    size_t csum_count(int o) const { return symw_csum_count(ntot_, o); }
    const SymwPlan &plan() const { return plan_; }
    int64_t stream_bytes() const;                      // bytes of Q one product of this rank reads
private:
    SymwPlan plan_;
    int64_t ntot_ = 0, ld_ = 0;
    int nloc_ = 0, omax_ = 0, world_ = 0;
    DevBuf<SymwItem> items_;
    DevBuf<int32_t> strip_ptr_;
    DevBuf<double> prow_, pcol_, csum_;
};

}  // namespace xm
