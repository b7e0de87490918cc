// xm_common.h — shared declarations of the MI355X-native XM solver (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

namespace xm {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

#define XM_HIP_CHECK(expr)                                                                            \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess)                                                                         \
            throw ::xm::Error(-3, std::string("HIP error: ") + hipGetErrorString(_e) + " at " +       \
                                      __FILE__ + ":" + std::to_string(__LINE__) + " (" #expr ")");    \
    } while (0)

// Row pitch of every device "3n x o" matrix: o rounded up to an odd number.  An odd pitch makes the
// per-lane ds_read_b128 of two consecutive W rows bank-conflict free (lane stride 4*OP dwords, OP odd).
__host__ __device__ constexpr int pitch_of(int o) { return o | 1; }

constexpr int kQwWaves = 4;       // cameras (wavefronts) per workgroup in the Q*W kernels
constexpr int kQwNsub = 2;        // 128-column sub-chunks per LDS-staged W tile (tile = 256 columns)
constexpr int kBsrRows = 16;     // camera rows per workgroup of the BSR3 kernels (4 wavefronts x 4 lane groups of 16)
constexpr int kColPad = 128;      // dense leading dimension is a multiple of this (64 lanes x double2)
constexpr int kMaxRank = 10;      // template instantiations cover o = 1 and 3..10
constexpr int kMaxInner = 1000;   // trustregion.h:416
constexpr int kMaxOuter = 1000;   // trustregion.h:417
// XM_FLAG_PROFILE_QW: every kProfileStride-th tCG product is bracketed by a pair of HIP events.  Each record is a barrier packet of its own: the
// kernel trace shows ~6 us of idle queue in front of the product and again in front of the launch behind it (11.5 us per sample) -- with a
// stride of 8 that was 1.4 us per tCG iteration, 3-5 % of the solve the measurement is about; 32 was still 0.36 us (0.9 % of the headline's
// 41 us per iteration); 64 leaves 70 samples per headline solve and 11 per Final-13682 solve (bench.py averages over six solves)
constexpr int kProfileStride = 64;

inline int64_t dense_ld(int64_t n_total) { return ((3 * n_total + kColPad - 1) / kColPad) * kColPad; }

// Device-resident scalar state of the truncated CG (double-buffered by iteration parity; written only by
// block 0 of the p-update kernel, read by everybody in the following launches).
struct TcgScal {
    double rr;        // <r,r>_metric of the current residual               (rdotr[i],   trustregion.h:484,626)
    double vv, vp, pp;  // |v|^2, <v,p>, |p|^2 by recurrence                (trustregion.h:642-644)
    double delta;     // trust-region radius
    double gradnorm;  // sqrt(rdotr[0])                                     (trustregion.h:485)
    double last_step; // alpha or tau of the last iteration (diagnostics)
    double model;     // XM_FLAG_MODEL_RECURRENCE: model value m(v) = <g,v> + <v,Hv>/2 by recurrence (m -= step * rr - step^2 <p,Hp> / 2); else 0
    int32_t status;   // 0 running | 1 negative curvature | 2 boundary | 3 norm tolerance | 5 rdotr<1e-15 | 6 max iterations | 7 a peer never arrived
                      // | 9 dormant: a speculatively enqueued tCG whose outer iteration did not go the predicted way (SpecCtl)
    int32_t iter;     // inner iteration index i (== completed iterations)
    int32_t seq;      // which truncated-CG run of this context this is (travels in the host-mapped progress word: a word of an earlier run is stale)
    int32_t phase;    // device-driven outer iteration only (Phase below); 0 in every other mode
};
// Trust-region state of the DEVICE-DRIVEN outer iteration (Context::trust_region_device; trustregion.h:416-718), double-buffered by parity next to
// TcgScal: with it the launch that ends a truncated CG can retract, the next product can take the candidate's gradient and the launch behind it
// can accept or reject, update the radius and start the next truncated CG -- without the host in between.  (A block of its own: folded into
// TcgScal it made cg_step_kernel copy a 128-byte struct through private memory -- 56 bytes of scratch, 13 KB of LDS, and 18.7 instead of 5.7 us
// per launch at Venice size, on the host-driven path that never looks at these fields.)
struct OuterScal {
    double loss;          // f at the current point                                 (loss[k])
    double rr_point;      // <rg, rg> at the current point                          (rdotr[0] of the truncated CG that starts there)
    int64_t totalite;     // the reference's "Total iteration"
    int32_t shrink_count; // consecutive radius shrinks                             (trustregion.h:683-694)
    int32_t k;            // outer iteration index
    int32_t stop_reason;  // 5 | 10 | 11 | 12 | 13 | 14 as Context::trust_region reports them; 0 while running
    int32_t time_up;      // the host's time limit has expired (sticky; looked at where the reference looks at its clock: top of an outer iteration)
    int32_t slots;        // (product, step) launch pairs that did work so far
    int32_t pad_;
};
// what a (product, step) pair of the device-driven outer iteration does, decided by the step launch before it
enum Phase { PH_TCG = 0,    // product: Hessian product of tCG iteration `iter`; step: cg_step -- and, when that ends the tCG, the retraction of the step
             PH_CAND = 1,   // product: cost / gradient at the candidate point; step: trust-region update, accept / reject, stop tests, start of the next tCG
             PH_STOP = 2,   // the trust region has ended (stop_reason): both launches return at once
             PH_INIT = 3 }; // first launch of a run: start the first tCG from the state the host uploaded (product launches return)

// Hand-over between the end of an outer iteration and a SPECULATIVELY enqueued start of the next truncated CG (single GPU): outer_finalize_kernel
// evaluates the trust-region update (trustregion.h:680-708) with the same formulas the host uses and, when the step is accepted and no stop test
// fires, lets the tcg_init / product / cg_step launches that are already in the queue behind it run -- the GPU does not idle through the host
// round trip (result word over PCIe, host arithmetic, three launches).  The host stays the authority: it takes the same decision from the same
// numbers and ADOPTS the running tCG only if `go` was set and the radius agrees bit for bit; otherwise the speculative launches were no-ops
// (status 9) or are overwritten by a regular start.
struct SpecCtl {
    int32_t go;       // 1: accepted, continue -- the queued tcg_init may start from the candidate point
    int32_t pad_;
    double rr;        // <rg,rg> of the accepted point
    double delta;     // radius after the update
};

// Everything the fused epilogues / per-camera kernels need.  All pointers are device pointers; matrices are
// row-major with pitch OP = pitch_of(o); "local" arrays hold the cameras [cam0, cam0+nloc) owned by this GPU.
struct CamArgs {
    int nloc;            // local cameras
    int cam0;            // global index of local camera 0 (camera 0 globally is the scale anchor, s == 1)
    double lam;
    // current point
    const double *R;     // nloc*3*OP
    const double *s;     // nloc
    // point state produced by the gradient epilogue
    double *G;           // 2*Q*sR rows
    double *egs;         // Euclidean d f / d s   (0 for the anchor)
    double *S0;          // nloc*9: sym(R_i egR_i^T), row-major 3x3
    double *rgR;         // Riemannian gradient
    double *rgs;
    // tCG vectors
    const double *pR;
    const double *ps;
    const double *rR;    // residual (for the <r,Hp> partial of the Hessian epilogue)
    const double *rs;
    double *HpR;
    double *Hps;
    // generic
    const double *Wloc;  // rows of the product input that belong to the local cameras (f partial)
    double *out;         // plain epilogue output rows
    double *partials;    // per-workgroup partial sums
    // certificate operator  S x = Q x + dz.*x(row 3i) - Lam_i x_i
    const double *Lam;   // nloc*9 row-major symmetric
    const double *dz;    // nloc
    const TcgScal *scal; // tCG kernels return immediately when scal->status != 0
    double *Bout;        // multi-rank tCG only: rows of  s.*Hp_R + Hp_s.*R  (the product-input image of Hp), else nullptr
    // dense product split in two launches (overlap of the W all-gather with the local column strip, xm_solver.hip:gathered product):
    int range_mode;      // 0 whole matrix | 1 only the column tiles [t_lo, t_hi) | 2 all tiles except [t_lo, t_hi)
    int t_lo, t_hi;
    const double *addend; // range_mode 2: raw row sums of the tiles done earlier (3*nloc x OP, pitch OP), added before the epilogue
    // dense product with the COLUMNS split over `ks` workgroups per camera group (small row strips of a multi-GPU run: 223 cameras
    // per rank = 56 workgroups on 256 CUs otherwise): raw partial sums per (slice, camera) + an arrival counter per camera group; the
    // last slice to arrive adds them in slice order and runs the epilogue (one launch, no atomics on data, bit-reproducible)
    int ks;               // 0 / 1: off
    double *ksum;         // ks x nloc x 3 x OP
    unsigned int *kcount; // one per camera group, zero between launches
    int nt_cam0;          // dense product: cameras >= this stream their rows non-temporally (set by the launcher from the size rule, not by callers)
    int rev;              // dense product (unsplit): 1 = the column tiles are walked right to left.  Consecutive products alternate it so that a
                          // launch starts with the tiles the previous one ended with (still in the L2s / the Infinity Cache); see launch_qw_sym
    // EPI_AUTO (device-driven outer iteration): the launch takes its role from scal->phase -- PH_TCG: Hessian epilogue with the fields above;
    // PH_CAND: gradient epilogue at the CANDIDATE point, whose buffers are these (cand_args()); otherwise it returns
    struct Cand { const double *R, *s; double *G, *egs, *S0, *rgR, *rgs, *partials; } cand;
};

enum Epilogue { EPI_PLAIN = 0, EPI_GRAD = 1, EPI_HESS = 2, EPI_CERT = 3, EPI_AUTO = 4 };
// the arguments an EPI_AUTO launch works with in the gradient role
__host__ __device__ inline CamArgs cand_args(const CamArgs &a) {
    CamArgs b = a;
    b.R = a.cand.R; b.s = a.cand.s; b.G = a.cand.G; b.egs = a.cand.egs; b.S0 = a.cand.S0; b.rgR = a.cand.rgR; b.rgs = a.cand.rgs;
    b.partials = a.cand.partials;
    return b;
}

// Direct peer-write exchange of the truncated CG (peer communicators, xm_comm.hip): device-visible description, by value into
// cg_step_kernel.  world == 0: no exchange (single GPU, or a communicator that gathers between the launches).
constexpr int kMaxPeers = 8;
struct PeerXchg {
    int world = 0, rank = 0;
    double *buf[kMaxPeers] = {};                // every rank's exchange buffer (2 parities x world chunks); [rank] = this rank's own
    unsigned long long *flag[kMaxPeers] = {};   // every rank's tCG flag words: flag[r][parity * kMaxPeers + source]
    unsigned long long *ticket = nullptr;       // this rank's arrival counters (one per parity)
    unsigned long long *err = nullptr;          // this rank's error word (a bounded spin expired), host-mapped
    long long spin_ticks = 0;                   // bound of every device-side wait, in wall_clock64() ticks (100 MHz)
    unsigned long long epoch_base = 0;          // epochs of this tCG run are epoch_base + iteration + 1 (identical on all ranks)
    int mute = 0;                               // tests: this rank never publishes its epoch (a dead peer)
    int lite = 0;                               // 1: payload through write-through (system-scope) stores + s_waitcnt instead of a release fence
};

// Arguments of the step launch of the device-driven outer iteration (outer_step_kernel, xm_kernels.hip), by value.  Buffers with a "cur" and a
// "next" flavour are the two parity copies of the truncated CG (the host passes them by the parity of the SLOT, the launch pair's index).
struct PointPtrs { double *G, *egs, *S0, *rgR, *rgs; };
struct OuterStepArgs {
    int nloc, cam0;
    const TcgScal *scal_cur;
    TcgScal *scal_next;
    const OuterScal *os_cur;
    OuterScal *os_next;
    const double *parts;     // this slot's tCG partial sums [<p,Hp> | <r,Hp> | <Hp,Hp> (nA each, Hessian epilogue) | |r|^2 of the previous step (nB)]
    double *partsB_out;      // |r|^2 partial sums of this step, in the other parity's chunk
    int nA, nB;
    const double *HpR, *Hps;
    double *R, *s;           // current point (overwritten with the candidate when it is accepted: the buffers keep their roles, nothing is swapped)
    double *Rc, *sc;         // candidate point
    double *pR;
    const double *ps_cur;
    double *ps_next;
    double *vR, *vs, *HvR, *Hvs, *rR;   // HvR == nullptr: XM_FLAG_MODEL_RECURRENCE
    const double *rs_cur;
    double *rs_next;
    double *Wloc, *Wpad;
    PointPtrs cur, cand;     // what the gradient epilogue wrote for the current / the candidate point
    const double *partsA;    // [f | <rg,rg>] partial sums of the candidate's gradient epilogue (nA each)
    double *partsM;          // model-decrease partial sums of the retraction
    int nM;
    double delta_bar, gradtol;
    int max_outer;
    double *trace;           // device copy of the per-outer-iteration trace (6 doubles per record), record k written at the top of iteration k
    int trace_cap;
    const int *stop_req;     // device word the host sets when its time limit has expired
    unsigned long long *hprog;   // host-mapped progress word [run : 32 | slots done : 24 | phase : 8]
    unsigned int run;
    int slot, grp;
};
void launch_outer_step(int o, int polar, const OuterStepArgs &A, int grid, hipStream_t st);

// ---- launchers implemented in xm_kernels.hip -------------------------------------------------------------------
// Q*W products.  grid = ceil(nloc / kQwWaves).  Q rows are the local cameras' rows; W has `ld` rows (all cameras).
void launch_qw_dense(int o, int epi, const double *Q, int64_t ld, const double *W, double alpha, const CamArgs &a,
                     hipStream_t st);
// nb = stored blocks of the nloc rows (0: unknown): decides the load policy of the block stream (cacheable below the Infinity Cache's size)
// rowinfo (device, nloc entries, may be nullptr = camera order): the rows binned by their number of 16-block windows (bsr_build_rowinfo)
void launch_qw_bsr3(int o, int epi, const int64_t *rowptr, const int32_t *colidx, const double *blocks, const double *W,
                    double alpha, const CamArgs &a, hipStream_t st, int64_t nb = 0, const int4 *rowinfo = nullptr);
// sliced-ELL stream: bytes of its prefix that are read with the default cache policy (found in the Infinity Cache by the next product) when an
// iteration moves `other` bytes besides the matrix; the rest streams non-temporally (xm_bench_dense_policy overrides the rule)
int64_t sell_resident_bytes(int64_t stream, int64_t other);
void bsr_build_rowinfo(const int64_t *rowptr_host, int nloc, std::vector<int4> &out);
int qw_grid(int nloc);
// the same product restricted to a range of column tiles (CamArgs.range_mode / t_lo / t_hi / addend); plain or gradient epilogue
void launch_qw_dense_split(int o, int epi, const double *Q, int64_t ld, const double *W, double alpha, const CamArgs &a, hipStream_t st);
int qw_dense_tile_cols();
int qw_dense_split_k(int nloc, int64_t ld);   // column split the small-strip policy picks for `nloc` cameras (1 = none)
int bsr_grid(int nloc);   // workgroups (= partial sums per epilogue slot) of the BSR3 kernels
size_t sym_prow_count(int nloc, int64_t ld, int o);
size_t sym_pcol_count(int nloc, int64_t ld, int o);
void qw_bench_nt(int nt);                                  // micro-benchmark override of the load policy: -1 by size, 0 default, 1 non-temporal
void symv_bench_k(int k, int kf);                                  // micro-benchmark override of the chunk length (xm_bench.h)
int symv_trace_slots();
void launch_qw_sym_traced(int o, const double *Q, int64_t ld, const double *W, const CamArgs &a, double *Prow, double *Pcol, unsigned long long *trace,
                          int grid[2], hipStream_t st);
void symv_plan_get(int nloc, int64_t ld, int out[4]);   // K, Kf, ysplit, nchunks of the vertical-sweep symmetric product
void launch_qw_sym(int o, int epi, const double *Q, int64_t ld, const double *W, double alpha, const CamArgs &a, double *Prow,
                   double *Pcol, hipStream_t st, int rev = 0);   // rev: sweep direction, alternated by the caller between consecutive products
void launch_asym(const double *Q, int64_t ld, int64_t m, double *out, int grid, hipStream_t st);
// exact symmetry check of a row-partitioned matrix: this strip's (rows row0 .. row0 + nrows of the m x m matrix) share of a sum modulo 2^64
// that vanishes over all strips iff the matrix is symmetric (xm_kernels.hip: symhash_kernel); out: 2 * grid words
void launch_symhash(const double *Q, int64_t ld, int64_t row0, int64_t nrows, int64_t m, unsigned long long *out, int grid, hipStream_t st);

// layout helpers
void launch_transpose_pad(const double *src_colmajor, int64_t lds, int64_t rows, int64_t cols, double *dst, int64_t ldd,
                          hipStream_t st);  // dst[r*ldd + c] = src[r + c*lds]
void launch_dense_from_bsr(const int64_t *rowptr, const int32_t *colidx, const double *blocks, int64_t nloc, int64_t cam0,
                           double *dst, int64_t ldd, hipStream_t st);

// flat / per-camera kernels
int flat_grid(int64_t elems);
void launch_scale_rows(int o, int nloc, const double *R, const double *s, double *Wloc, hipStream_t st);
void launch_tcg_init(int o, int nloc, const double *rgR, const double *rgs, const double *R, const double *s, double *rR,
                     double *rs, double *pR, double *ps, double *vR, double *vs, double *HvR, double *Hvs, double *Wloc,
                     TcgScal *scal0, double rr, double delta, unsigned long long *hstat, hipStream_t st, double *Wpad = nullptr, int seq = 0,
                     const SpecCtl *spec = nullptr);   // spec: start only if spec->go, with spec->rr / spec->delta (else leave scal0 dormant)
void launch_cg_step(int o, int nloc, const TcgScal *scal_cur, TcgScal *scal_next, const double *parts, int nA_loc, int nB_loc, int world,
                    const double *HpR, const double *Hps, const double *R, const double *s, double *pR,
                    const double *ps_cur, double *ps_next, double *vR, double *vs, double *HvR, double *Hvs, double *rR, const double *rs_cur,
                    double *rs_next, double *Wloc, double *partsB_out, unsigned long long *hstat, int b_off, int64_t mat, double *Afull,
                    double *Wfull, int grouping, const struct PeerXchg &xchg, hipStream_t st, double *Wpad = nullptr);   // Wpad: single-rank only
// trust-region numbers of the iteration that ends (what the host holds when it enqueues this launch); spec_out: see SpecCtl (nullptr: none)
struct OuterArgs { double loss, delta, delta_bar, gradtol; int shrink_count, last_iter; };
void launch_outer_finalize(const double *partsA, int nA_loc, int world, const double *partsM, int nM, const TcgScal *scal, double *hres,
                           unsigned long long seq, int grouping, hipStream_t st, const OuterArgs *oa = nullptr, SpecCtl *spec_out = nullptr);
// polar: 0 the reference's Gram-Schmidt retraction, one thread per camera | 1 polar retraction (XM_RETRACT_POLAR) | 2 Gram-Schmidt with a quad
// of lanes per camera (measured alternative, scripts/kbench_retract.py)
void launch_retract(int o, int nloc, int cam0, const double *R, const double *s, const double *D, const double *ds, double t,
                    double *Rout, double *sout, double *Wloc, hipStream_t st, int polar = 0);
// the retraction of an outer iteration's step (vR, vs) with the model decrease of that step fused in (parts: retract_grid(nloc) partial sums, the
// sum trustregion.h:667-668 takes) and, if Wpad is given, the product input also at the 128-byte record pitch (xm_sell.h)
int retract_grid(int nloc);
void launch_retract_model(int o, int nloc, int cam0, const double *R, const double *s, const double *vR, const double *vs, double *Rout, double *sout,
                          double *Wloc, double *Wpad, const double *HvR, const double *Hvs, const double *rgR, const double *rgs, double *parts,
                          hipStream_t st, int polar);
void launch_cert_prepare(int o, int nloc, int cam0, double lam, const double *QsR, const double *R, const double *s,
                         double *Lam, double *dz, double *parts, hipStream_t st);
// solution recovery
void launch_recover_gram(int64_t n, int r, const double *R, const double *s, double *parts, int grid, hipStream_t st);
// variant 0: one thread per camera (default) | 1: one wavefront per camera, cross-lane reductions (north_star's form; measured alternative)
void launch_recover_project(int64_t n, int r, const double *R, const double *s, const double *V, double *rot, double *scale, int *negcount,
                            hipStream_t st, int variant = 0);
void launch_negate(double *x, int64_t len, hipStream_t st);
// small vector kernels used by Lanczos
int dots_multi_segments(int64_t len);
void launch_dots_multi(const double *V, int64_t ldv, int m, const double *w, int64_t len, double *c, double *scratch, hipStream_t st);
void launch_sub_vc(double *w, const double *V, int64_t ldv, const double *c, int m, int64_t len, hipStream_t st);
void launch_scale_copy(double *dst, const double *src, double a, int64_t len, hipStream_t st);
void launch_copy_words(void *dst, const void *src, size_t bytes, hipStream_t st);   // bytes % 4 == 0; either side may be host-mapped memory
void launch_lz_alpha(const double *c1j, const double *c2j, double *alpha_j, hipStream_t st);
// the same Lanczos step in seven launches instead of eleven (the sums of the partial dots are taken inside the kernels that use them: same bits)
void launch_dots_multi_parts(const double *V, int64_t ldv, int m, const double *w, int64_t len, double *scratch, hipStream_t st);
bool lz_fused_ok(int m);
void launch_sub_vc_fin(double *w, const double *V, int64_t ldv, const double *parts, int m, int64_t len, double *c_out, const double *c_prev,
                       double *alpha_j, hipStream_t st);
void launch_lz_next_fin(double *dst, const double *w, const double *parts, double *beta_j, int64_t len, hipStream_t st);
void launch_lz_next(double *dst, const double *w, const double *ww, double *beta_j, int64_t len, hipStream_t st);
void launch_gemv_n(double *y, const double *V, int64_t ldv, const double *c, int m, int64_t len, hipStream_t st);  // y = V c


// XM^2 re-weighting
void launch_edge_locate(int64_t ne, const int32_t *ei, const int32_t *ej, int cam0, int nloc, const int64_t *rowptr, const int32_t *colidx,
                        int64_t *pos_ij, int64_t *pos_ji, int64_t *pos_d, hipStream_t st);
void launch_edge_write(bool dense, int64_t ne, const int32_t *ei, const int32_t *ej, const double *M, const double *w, int cam0, int nloc,
                       const int64_t *inc_ptr, const int32_t *inc_edge, const int64_t *pos_ij, const int64_t *pos_ji, const int64_t *pos_d,
                       double *blocks, double *Q, int64_t ld, hipStream_t st);
void launch_edge_residual(int64_t ne, const int32_t *ei, const int32_t *ej, const double *M, const double *Y, int o, int OP, double *res,
                          hipStream_t st);

// XM^2 outlier filter (weighted residuals, radix select of an order statistic, weight filter)
void launch_xm2_error(int64_t n, const double *w, const double *res, double *err, hipStream_t st);
void launch_radix_hist(int64_t n, const double *x, int shift, unsigned long long prefix, unsigned int *hist, hipStream_t st);
void launch_xm2_filter(int64_t n, const double *err, const double *w, double thr, double *wout, unsigned int *removed, hipStream_t st);

// ---- host-side launch helpers shared by the kernel translation units ---------------------------------------------
#define XM_DISPATCH_O(o, CALL)                                                         \
    switch (o) {                                                                       \
        case 1: { constexpr int O_ = 1; CALL; } break;                                 \
        case 3: { constexpr int O_ = 3; CALL; } break;                                 \
        case 4: { constexpr int O_ = 4; CALL; } break;                                 \
        case 5: { constexpr int O_ = 5; CALL; } break;                                 \
        case 6: { constexpr int O_ = 6; CALL; } break;                                 \
        case 7: { constexpr int O_ = 7; CALL; } break;                                 \
        case 8: { constexpr int O_ = 8; CALL; } break;                                 \
        case 9: { constexpr int O_ = 9; CALL; } break;                                 \
        case 10: { constexpr int O_ = 10; CALL; } break;                               \
        default: throw Error(-2, "rank o must be 1 or 3..10, got " + std::to_string(o)); \
    }

inline void check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) throw Error(-3, std::string("kernel launch failed (") + what + "): " + hipGetErrorString(e));
}

}  // namespace xm
