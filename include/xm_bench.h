/*
 * xm_bench.h -- timing hooks of the micro-benchmarks (scripts/kbench_*.py, bench.py's roofline_hbm leg).  NOT part of the product ABI:
 * include/xm_amd.h does not include this file and a caller of the solver never needs it.  Every function times launches of a kernel
 * the solver uses, through the solver's own launchers, with HIP events; 0 == success, xm_bench_last_error() gives the message.
 */
#ifndef XM_BENCH_H
#define XM_BENCH_H

#include "xm_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

const char *xm_bench_last_error(void);
/* average milliseconds of `reps` back-to-back xm_qw_dense launches (dq from xm_dense_upload) */
int xm_qw_dense_time(const double *dq, int64_t n, int o, const double *dW, double *dOut, int reps, double *ms_avg);
/* the half-traffic symmetric product (xm_qw_dense_sym), scratch allocated once */
int xm_qw_dense_sym_time(const double *dq, int64_t n, int o, const double *dW, double *dOut, int reps, double *ms_avg);
/* micro-benchmark settings of the symmetric sweep: k > 0 overrides the chunk length of the plan (0 = the plan's own; set it BEFORE a context
 * or a timing call sizes its partial-result buffers); alternate = 0: xm_qw_dense_sym_time / xm_qw_dense_time walk the matrix in the same direction in every
 * launch instead of alternating it between consecutive products as the solver does; kf > 0 (with k > 0): the last quarter of the grid rows is cut
 * into chunks of kf steps whatever the size (the plan does that by itself only for sweeps of several residency rounds) */
int xm_bench_symv_k(int k, int alternate, int kf);
/* load policy of the matrix streams (dense, block-CSR, sliced ELL) in the micro-benchmarks and in solves of this process: -1 = by size (the
 * products' rules), 0 = default (cacheable), 1 = non-temporal, n >= 2 = a cacheable prefix of n MB and the rest non-temporal, n <= -2 = a prefix of
 * -n KB (mixed policies on the small matrices of the tests).  A cache hint only: results never depend on it (tests assert that) */
int xm_bench_dense_policy(int nt);
/* ONE traced launch of the sweep (o = 3 or 4, top-down): per wavefront `slots` 100 MHz timestamps -- [0] entry, [1] after the status word,
 * [2 + i] after step i, [slots - 3] loop done, [slots - 2] column sums written, [slots - 1] XCC_ID << 32 | HW_ID; trace_host = NULL: grid and slots only */
int xm_qw_dense_sym_trace(const double *dq, int64_t n, int o, const double *dW, double *dOut, unsigned long long *trace_host, int64_t trace_cap,
                          int grid[2], int *slots);
/* the same for a ROW STRIP of nloc cameras of an n-camera matrix (what one rank of an N-GPU row partition multiplies: dq = 3 nloc rows x
 * xm_dense_ld(n)): the expected per-iteration time of the partitioned solve from measured pieces */
int xm_qw_dense_strip_time(const double *dq, int64_t nloc, int64_t n, int o, const double *dW, double *dOut, int reps, double *ms_avg);
/* the strip product with its COLUMNS split over `ks` workgroups per camera group (ks = 0: the small-strip policy picks; 1: no split):
 * out = alpha * Q_strip * W (also a functional entry: tests compare it with the unsplit product); with reps > 0 also the average launch time */
int xm_qw_dense_strip_ks(const double *dq, int64_t nloc, int64_t n, int o, const double *dW, double *dOut, double alpha, int ks, int reps,
                         double *ms_avg, int *ks_used);
/* xm_qw_bsr3_time multiplies with every workgroup's 16 rows ordered by their number of 16-block windows, as a solve does (1, default), or in camera order (0) */
int xm_bench_bsr_binned(int on);
int xm_qw_bsr3_time(const int64_t *d_rowptr, const int32_t *d_colidx, const double *d_blocks, int64_t n, int o, const double *dW,
                    double *dOut, int reps, double *ms_avg);
/* sliced-ELL product (both launches); dWpad16 = the input at a record pitch of 16 doubles or NULL (xm_qw_sell_padded) */
int xm_qw_sell_time(void *handle, int o, const double *dW, const double *dWpad16, double *dOut, int gather_mode, int reps, double *ms_avg);
/* the retraction with the kernel form chosen -- variant 0: one thread per camera (the default), 1: polar retraction, 2: MGS-QR with a quad of
 * lanes per camera and DPP reductions (`north_star`'s cross-lane form; measured slower, kept as the recorded alternative) -- and, if ms_avg
 * is not NULL, timed over `reps` launches */
int xm_retract_variant(int64_t n, int o, const double *dR, const double *ds, const double *dD, const double *dds, double t, double *dRout,
                       double *dsout, int variant, int reps, double *ms_avg);
/* xm_recover_rotations (include/xm_amd.h) with the per-camera projection kernel chosen -- variant 0: one thread per camera (the default), 1: one
 * wavefront per camera with cross-lane reductions, the form BASELINE.json's north_star names for the 3x3 SVD -- and, with reps > 0, that launch
 * timed (ms_avg) */
int xm_recover_rotations_variant(int64_t n, int r, const double *R, const double *s, double *rot, double *scale, int *n_negative_det, int variant,
                                 int reps, double *ms_avg);
/* the direct peer-write all-gather (xm-code_amd/csrc/xm_comm.hip): `world` ranks (one host thread each; gpu_map 1 = all on device 0) gather
 * `count` doubles per rank `reps` times; us_avg = average time per collective on rank 0, stream time */
int xm_peer_allgather_bench(int world, int gpu_map, int64_t count, int reps, double *us_avg);
/* ONE rank's share of the multi-rank symmetric window product on this GPU (rank cam0 / nloc of `world`, an arbitrary row strip): ms[0] =
 * sweep + column sums, ms[1] = per-camera sum + plain epilogue; *bytes = bytes of Q the sweep streams.  XM_ERR_ARG unless cam0 is a
 * multiple of nloc, cam0 / nloc < world and o in 1, 3..5 */
int xm_qw_symw_time(int64_t ntot, int nloc, int cam0, int o, int world, int reps, double ms[2], int64_t *bytes);

/* a grid-wide barrier inside one launch against a kernel boundary (`blocks` resident workgroups, `rounds` rounds of write / meet / check): us[0] =
 * microseconds per round inside ONE launch, us[1] = per round as separate launches, us[2] = failed checks or expired waits (must be 0) */
int xm_bench_grid_barrier(int blocks, int rounds, int reps, double us[3]);

#ifdef __cplusplus
}
#endif
#endif /* XM_BENCH_H */
