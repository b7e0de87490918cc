/*
 * xm_amd.h — C ABI of the MI355X-native XM solver (libxm_amd.so).
 *
 * Drop-in boundary for the Burer-Monteiro / Riemannian-staircase SDP solve of
 * ComputationalRobotics/XM-code.  The reference exposes this path as three pybind11 functions
 * (XM/src/XM_main.cu:403-408):
 *      XM.solve(dataset_path, max_rank, tol, lam, max_time)           -> None   (XM_main.cu:180)
 *      XM.solve_rank3(dataset_path, max_rank, tol, lam, max_time)     -> None   (XM_main.cu:312)
 *      XM.solve_rebuttle(dataset_path, max_rank, tol, lam, max_time)  -> int    (XM_main.cu:35)
 * Section 1 below are exactly those entry points (same argument meaning, same Q.bin/R.bin/s.bin
 * files); the pybind11 module `XM` shipped in xm-code_amd/csrc/xm_pybind.cpp is a ~30-line shim
 * over them (INTEGRATION.md shows the stub).  Sections 2-4 are additive: an in-memory context API
 * (what bench.py times: Q already resident in HBM), kernel-level entry points on device pointers
 * (what the parity tests call), and the multi-GPU row-partition hooks.
 *
 * Conventions: plain pointers and sizes only, no exceptions cross the ABI, 0 == success,
 * negative == error (xm_last_error() gives the message).  All matrices float64.
 * Host-side matrices use the reference's layouts (column-major; R is 3n x r, camera i = rows
 * 3i..3i+2, XM_main.cu:231-237).  Device-side vectors use the solver's internal row-major
 * "camera-major" layout: element (row r, column k) of a 3n x o matrix at r*o + k.
 */
#ifndef XM_AMD_H
#define XM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ error codes */
#define XM_OK              0
#define XM_ERR_IO         -1   /* missing / short Q.bin etc. (reference: prints "cannot open file" and goes on, XM_main.cu:21-24) */
#define XM_ERR_ARG        -2
#define XM_ERR_HIP        -3   /* HIP runtime / no device / kernel failure */
#define XM_ERR_COMM       -4   /* RCCL failure */
#define XM_ERR_NOMEM      -5

/* status codes of the staircase (solve_rebuttle's return value, XM_main.cu:35-178) */
#define XM_STATUS_NONE          0
#define XM_STATUS_CERTIFIED     1
#define XM_STATUS_MAX_RANK      2
#define XM_STATUS_LS_FAILED    -2

const char *xm_last_error(void);
const char *xm_version(void);
/* ABI revision of the structs below.  Since revision 3 xm_problem_t, xm_options_t and xm_result_t START with `struct_size`, which the
 * caller sets to sizeof(its own struct).  The library copies min(struct_size, its own sizeof) bytes, treats what the caller did not
 * pass as zero and never writes past the caller's size, so a caller compiled against a shorter (older) header keeps working;
 * struct_size == 0 is rejected with XM_ERR_ARG (revisions 1-2 had no size field and are NOT binary compatible).  Revision 4 (this
 * one): xm_tuning_t re-laid (the fields that selected removed experiment kernels are gone, test / matrix-free switches added) -- a
 * caller that passes a NON-NULL xm_problem_t.tuning must be compiled against this header; timing hooks moved to xm_bench.h. */
#define XM_ABI_REVISION 4
int xm_abi_revision(void);

/* ================================================================== 1. file-based surface == the reference's pybind functions */
/* replaces XM_main.cu:180  solve(): reads <path>/Q.bin, writes <path>/R.bin and <path>/s.bin
 * (.bin = int32 rows, int32 cols, float64 column-major, XM_main.cu:18-33; Q.bin may instead carry the two 8-byte header
 * fields of utils/io.py:24-26, recognised by the file size) */
int xm_solve(const char *dataset_path, unsigned int max_rank, double tol, double lam, double max_time);
/* replaces XM_main.cu:312  solve_rank3() */
int xm_solve_rank3(const char *dataset_path, unsigned int max_rank, double tol, double lam, double max_time);
/* replaces XM_main.cu:35   solve_rebuttle(): also reads R_ini.bin / s_ini.bin; *status receives 1 / 2 / -2 / 0 */
int xm_solve_rebuttle(const char *dataset_path, unsigned int max_rank, double tol, double lam, double max_time,
                      int *status);

/* ================================================================== 2. in-memory context API */
typedef struct xm_ctx xm_ctx_t;

#define XM_STORAGE_DENSE 0     /* dense symmetric 3n x 3n, column-major (the reference format, XM_main.cu:18-33,191) */
#define XM_STORAGE_BSR3  1     /* 3x3-block CSR over view-graph edges, both triangles stored */
#define XM_STORAGE_BSR3_DENSE 2 /* described as BSR3 on the host (same fields), expanded to the dense layout on the device:
                                  each rank builds only its own camera rows (a >= 10k-camera Q never exists on the host) */
#define XM_STORAGE_VIEWGRAPH 4 /* the north_star workload described by its EDGE LIST: Q = sum_e w_e G_e over view-graph edges e = (i, j), Q_ii += w_e I,
                                  Q_jj += w_e I, Q_ij = -w_e M_e, Q_ji = Q_ij^T with M_e the measured relative rotation (what xm_ctx_attach_edges takes).
                                  Stored as 3x3-block CSR; with >= 2.2 M blocks per GPU (xm_tuning_t.sell) the products stream the compressed sliced-ELL copy
                                  (quaternion per off-diagonal block, one double per diagonal block: 36 B per stored block instead of 76,
                                  xm-code_amd/csrc/xm_sell.h).  The edges are attached for the XM^2 calls at creation. */
#define XM_STORAGE_SCHUR 3     /* MATRIX-FREE: Q is never formed.  The problem is the observation list the reference's
                                  utils/creatematrix.py:create_matrix(weight, edges, landmarks) takes (creatematrix.py:51); the product applies
                                  Q = Q1 - Vtp_bar Qtp_bar^{-1} Vtp_bar^T as a factor chain (xm-code_amd/csrc/xm_schur.h).  Single GPU. */

/* Context-creation settings that select kernels / layouts (all 0 = the defaults = automatic choice).  They are the ONLY way to select
 * a kernel or a layout: the library reads no environment variable for that (the variables it does read are listed in INTEGRATION.md:
 * XM_QUIET, XM_GPUS, XM_GPU_MAP, XM_RETRACTION, XM_WATCHDOG_S for callers of the file surface, which has no tuning argument, and
 * XM_COMM_PEER, XM_COMM_TRACE, XM_FORCE_COMM, XM_SHM_TIMEOUT, XM_SHM_ASYNC for the process-level communicator set-up of section 4). */
typedef struct {
    int32_t sym;               /* half-traffic symmetric dense product: 0 auto (3n >= sym_min_rows, exactly symmetric Q), 1 force (1e-9 asymmetry accepted), -1 off */
    int32_t sym_min_rows;      /* 0 = measured: 4096 on one GPU (also where the matrix-free storage starts applying its inverse with the symmetric kernel), 6144 for the multi-rank window */
    int32_t sell;              /* sliced-ELL copy of a block-sparse Q: 0 auto (by size, xm_solver.hip), 1 force, -1 off */
    int32_t sell_slabs;        /* 0 = 4 (1, 2, 4, 8) */
    int32_t sell_lmax;         /* 0 = 64 */
    int32_t sell_gather;       /* how the records of W are fetched: 0 = default (2), 1 = one record per lane, 2 = records fetched element-per-lane and
                                  transposed through LDS; anything else: XM_ERR_ARG */
    int32_t sell_codec;        /* 0 auto (view-graph storage: quaternion codec; BSR3: full blocks), 1 full blocks, 2 quaternion codec (XM_ERR_ARG if Q is not a view-graph matrix) */
    int32_t overlap;           /* split dense products outside the tCG around the all-gather of W: 0 auto (>= overlap_min_mb per rank), -1 off */
    int32_t overlap_min_mb;    /* 0 = 64; -1 = no minimum (tests) */
    int32_t cert_dense_rows;   /* certificate: complete tridiagonalisation for 3n <= this; 0 = 384 */
    int32_t lanczos_mmax;      /* 0 = 400 */
    int32_t lanczos_restarts;  /* 0 = 12 */
    int32_t watchdog_s;        /* host spin loops give up after this many seconds without progress; 0 = XM_WATCHDOG_S, else 600 */
    int32_t balance;           /* row partition of block-sparse storage: 0 = by stored blocks, 1 = equal camera ranges */
    int32_t exchange;          /* multi-GPU tCG exchange: 0 auto (direct peer writes fused into the tCG kernel when the ranks share this process or IPC
                                  is set up and the transport passes its self-test, else RCCL), 1 an all-gather between the launches (whatever the
                                  transport), 2 direct peer writes, 3 RCCL even where peer writes would work (single-process mode) */
    int32_t split_k;           /* dense product of a SMALL row strip with its columns split over several workgroups per camera group: 0 auto
                                  (multi-GPU runs whose strip has fewer than ~1.5 workgroups per CU), -1 off, 2..8 forced (also on one GPU) */
    int32_t sell_wpad;         /* sliced-ELL product inside the truncated CG (single GPU, rank 3..5): the kernels that write the product input also write
                                  a copy at a record pitch of 128 bytes, which the gather reads (one cache line per record instead of 1.4 / 1.9):
                                  0 auto (when the column pattern has no locality: random view graphs yes, banded ones no) | 1 always | -1 never */
    int32_t exchange_fence;    /* 1: the direct peer exchange publishes with plain stores + a system-scope release fence instead of write-through stores */
    int32_t schur_host_assembly; /* XM_STORAGE_SCHUR: 1 = assemble the reduced camera Laplacian on the host (the reference's route, utils/creatematrix.py:137-260) */
    int32_t schur_trace;       /* XM_STORAGE_SCHUR: 1 = set-up phase times on stderr */
    int32_t schur_solver;      /* XM_STORAGE_SCHUR: how the reduced camera Laplacian is applied inside the product: 0 auto (dense inverse up to
                                  schur_dense_max cameras, preconditioned CG above), 1 dense inverse, 2 preconditioned CG (xm-code_amd/csrc/xm_schur.h) */
    int32_t schur_dense_max;   /* 0 = 20000 */
    int32_t debug_drop_finalize; /* tests: the k-th outer iteration loses its result kernel (the host must come back with XM_ERR_HIP) */
    int32_t debug_peer_mute;   /* tests: rank 1 never publishes its tCG epoch (a dead peer: the bounded waits must expire) */
    int32_t schur_pcg_first;   /* XM_STORAGE_SCHUR, CG form: iterations enqueued in the first batch of the first product (0 = 26; tests force top-up batches) */
    int32_t schur_pcg_hess_digits; /* ... relative residual 10^-d of the inner solve inside Hessian products: 0 = 9, 6 .. 13 (13 = as tight as the
                                  gradient / cost / certificate products always are) */
    int32_t reserved[2];
} xm_tuning_t;

typedef struct {
    uint32_t struct_size;      /* sizeof(xm_problem_t) of the CALLER (XM_ABI_REVISION) */
    int64_t n;                 /* cameras */
    int32_t storage;           /* XM_STORAGE_* */
    int32_t q_on_device;       /* dense only: q is a DEVICE pointer already in the solver's padded row-major layout
                                  (xm_dense_ld(n) doubles per row); the context borrows it (no copy) */
    const double *q;           /* dense: host column-major (ldq >= 3n) unless q_on_device */
    int64_t ldq;
    int64_t nb;                /* BSR3: stored blocks */
    const int64_t *rowptr;     /* n+1 */
    const int32_t *colidx;     /* nb */
    const double *blocks;      /* nb x 9, each block ROW-major (b[3*a+c] = Q[3i+a, 3j+c]) */
    /* XM_STORAGE_SCHUR: nobs observations (camera obs_cam[e], landmark obs_lm[e], both 0-based; obs_p: nobs x 3 row-major point in the
     * camera frame, already normalised with the intrinsics; obs_w: weights) -- i.e. edges - 1, landmarks, weight of create_matrix */
    int64_t nobs, n_landmarks;
    const int32_t *obs_cam, *obs_lm;
    const double *obs_p, *obs_w;
    int64_t q_row0;            /* dense host q only: q holds the rows [q_row0, q_row0 + ldq) of Q (all 3n columns, column-major,
                                  leading dimension ldq); 0 with ldq >= 3n = the whole matrix.  Lets a rank of a multi-GPU run hand
                                  over just its own row strip (xm_solve reads only that strip of Q.bin) */
    /* XM_STORAGE_VIEWGRAPH: ne edges (edge_i[e], edge_j[e]), 0-based, i != j, no unordered pair twice; edge_w: ne weights;
     * edge_M: ne x 9 row-major rotations */
    int64_t ne;
    const int32_t *edge_i, *edge_j;
    const double *edge_w, *edge_M;
    /* SINGLE-PROCESS MULTI-GPU (SURVEY.md 8b "Threading"): n_gpus > 1 row-partitions the cameras over n_gpus devices driven by one host
     * thread each inside THIS process; every xm_ctx_* call fans out to them and returns the (identical) result of rank 0.  The
     * reference's callers (1_test_solve.py:42, 3_test_colmap_glomap.py:285) stay single-process scripts.  0 / 1 = one GPU.  The file
     * surface (xm_solve...) takes the count from the environment variable XM_GPUS. */
    int32_t n_gpus;
    int32_t gpu_map;           /* 0: rank g on device g;  1: every rank on device 0 with its own stream ("virtual devices": exercises the
                                  whole multi-GPU path, peer writes included, on a 1-GPU box) */
    const xm_tuning_t *tuning; /* NULL = all defaults */
} xm_problem_t;

#define XM_MODE_SOLVE    0     /* XM_main.cu:180 */
#define XM_MODE_RANK3    1     /* XM_main.cu:312 */
#define XM_MODE_REBUTTLE 2     /* XM_main.cu:35 (s_ini honoured, R_ini overwritten by the identity stack like the reference) */

#define XM_FLAG_VERBOSE        1u   /* reference-style progress lines on stdout (trustregion.h:504, checkeig.h:317-337) */
#define XM_FLAG_FIX_STALE_SR   2u   /* recompute sR after the escalation line search (reference does not, trustregion.h:394-422) */
#define XM_FLAG_PROFILE_QW     4u   /* time every 64th tCG Q*W launch with HIP events (result.qw_*) */
#define XM_FLAG_HOST_STEPPED   8u   /* debugging: synchronise after every tCG iteration instead of run-ahead polling */
#define XM_FLAG_MODEL_RECURRENCE 32u /* the model decrease of a truncated CG from its own recurrences (m -= step <r,r> - step^2 <p,Hp> / 2) instead of from the
                                     * accumulated vectors v, Hv as the reference forms it (trustregion.h:605-610, 667-668): the tCG neither reads nor writes
                                     * Hv (2 x 24 n o bytes per iteration -- it shows from ~50 k cameras on, where cg_step is bound by its bytes).  Equal in
                                     * exact arithmetic; the last bits of the model value, hence possibly the path, differ: default OFF */
#define XM_FLAG_HOST_OUTER    64u   /* keep the outer iteration of the trust region on the HOST (the form of rounds 1-5: the host notices the end of a truncated
                                     * CG, enqueues retraction / candidate gradient / result kernel and confirms a speculatively started next tCG).  Default on
                                     * one GPU with block-CSR products -- and with dense products when XM_FLAG_DEVICE_OUTER asks for it -- (not: sliced ELL,
                                     * matrix-free, several ranks, XM_FLAG_HOST_STEPPED; with XM_FLAG_VERBOSE a stage's progress lines appear when its trust region has ended)
                                     * is the DEVICE-driven form: everything trustregion.h:527-708 does between two truncated CGs (retraction, candidate's cost /
                                     * gradient, accept / reject, radius, stop tests, start of the next tCG) is decided on the device, the host enqueues one
                                     * repeating pair of launches ahead and watches a progress word.  Same decisions from the same numbers: bit-identical
                                     * paths in block-CSR storage; the dense products alternate their sweep direction by launch pair instead of by tCG
                                     * iteration, so dense paths differ in the last bits */
#define XM_FLAG_DEVICE_OUTER 128u   /* the device-driven outer iteration also for DENSE products (one GPU), where the host-driven form is the default because
                                     * it measures 1.5-2 % faster there (the role-switching product kernels are 0.3-0.8 us dearer per launch and the host-driven
                                     * loop's round trips are hidden behind the speculative start of the next tCG: 41.7 against 42.4 us per tCG iteration on
                                     * the Venice-1778-size problem); in block-CSR storage the device-driven form is the faster one (34.7 against 35.5 us at
                                     * 13 682 cameras) and the default.  XM_FLAG_HOST_OUTER wins over this flag */
#define XM_FLAG_WARM_R        16u   /* XM_MODE_REBUTTLE: start the rank-3 stage from opt.R_ini instead of the identity stack.  The reference
                                       reads R_ini.bin and then overwrites it with the identity (XM_main.cu:41,95-103); this flag honours it,
                                       which is what makes the second solve of the XM^2 loop cheap */

#define XM_RETRACT_QR    0     /* the reference's retraction: modified Gram-Schmidt of the 3 rows (Dense/batchedQR.h:9-69, trustregion.h:341-351) */
#define XM_RETRACT_POLAR 1     /* polar retraction R_i <- (M M^T)^{-1/2} M, M = R_i + xi_i (the orthogonal factor of the 3 x o block, via the 3x3 Gram
                                  matrix; BASELINE.json north_star names it, the reference uses it only in utils/recoversolution.py:65-86).  Same
                                  fixed points and certified optimum, different trajectory. */

typedef struct {
    uint32_t struct_size;      /* sizeof(xm_options_t) of the CALLER */
    uint32_t max_rank;
    double tol, lam, max_time;
    int32_t mode;
    uint32_t flags;
    const double *s_ini;       /* n values, XM_MODE_REBUTTLE only (may be NULL -> ones) */
    int32_t trace_cap;         /* optional per-outer-iteration trace: records of 6 doubles */
    double *trace;             /*   loss, gradnorm, inner_iters, endreason, trstatus, delta (same as the oracle) */
    const double *R_ini;       /* XM_FLAG_WARM_R only: 3n x 3 column-major starting point (rows are re-orthonormalised) */
    int32_t retraction;        /* XM_RETRACT_* */
    int32_t sum_grouping;      /* order in which the per-workgroup partial sums of the tCG / outer iteration are added: 0 ascending (default),
                                  1 descending, 2 even-then-odd.  Every grouping is fixed and bit-reproducible; they differ in the last bits, which
                                  is enough to change the iteration count at the saddle points of a staircase -- bench.py
                                  rotates through them so that the headline is not one draw of that lottery */
} xm_options_t;

typedef struct {
    uint32_t struct_size;      /* sizeof(xm_result_t) of the CALLER */
    double *R;                 /* caller-allocated 3n x max(max_rank,3)+1, column-major, leading dim 3n */
    double *s;                 /* caller-allocated n (s[0] == 1) */
    int32_t rank;              /* columns of R that are valid (what R.bin would hold) */
    int32_t status;            /* XM_STATUS_* */
    double primal, dual, min_eig, gap;
    int64_t tcg_iters;         /* sum over outer iterations of (i+1) — the reference's "Total iteration" */
    int64_t outer_iters;
    int64_t qw_products;       /* launches of the Q*W kernel (any epilogue) */
    int64_t lanczos_iters;
    double seconds;            /* wall clock of the whole solve (Q already resident) */
    double tr_seconds;         /* time inside the trust-region loops */
    double cert_seconds;       /* time inside the certificates */
    double qw_ms_sum;          /* XM_FLAG_PROFILE_QW: summed duration of the sampled Q*W launches (HIP events) */
    int64_t qw_ms_count;       /*   number of sampled launches */
    int64_t qw_bytes;          /* algorithmic bytes of ONE tCG Q*W launch at the final rank */
    int32_t trace_len;
    int32_t last_stop_reason;
    int32_t sym_product;       /* 1 when the half-traffic symmetric product was used (dense, single GPU, Q exactly symmetric) */
    int32_t cert_flags;        /* XM_CERT_* bits of the LAST certificate */
    double eig_residual;       /* Ritz residual |S x - theta x| of the last certificate's Lanczos run (reference: exact syevd, checkeig.h:303-318) */
    int32_t n_gpus;            /* ranks that took part in the solve */
    int32_t exchange;          /* multi-GPU tCG exchange used: 1 RCCL all-gather, 2 direct peer writes (0 single GPU) */
    int64_t qw_stream_bytes;   /* bytes of Q one tCG product actually streams (== the matrix part of qw_bytes unless a compressed or
                                  symmetric path is used) */
    int32_t outer_on_device;   /* trust regions (rank levels) of this solve whose outer iteration was driven by the device (0: all by the host --
                                  XM_FLAG_HOST_OUTER, dense products without XM_FLAG_DEVICE_OUTER, or a configuration the device-driven form does not
                                  cover); appended in round 6 */
    int32_t reserved_;
} xm_result_t;
#define XM_CERT_EIG_NOT_CONVERGED 1   /* Lanczos hit its iteration cap: min_eig is only an upper bound, the certificate was NOT accepted on it */
#define XM_CERT_EIG_EXACT 2           /* small problem (3n <= cert_dense_rows, default 384): the Krylov space of S was EXHAUSTED (3n steps, or an
                                       * invariant subspace was hit: beta below round-off under full re-orthogonalisation) and min_eig is the
                                       * smallest eigenvalue of the complete tridiagonal matrix -- the dense path of the reference
                                       * (Dense/eig.h:35-73 dsyevd, checkeig.h:303-318) with the reduction done matrix-free.  Set AFTER the run. */

#define XM_CERT_INEXACT_OPERATOR 4     /* matrix-free storage, CG form: an inner solve of the reduced camera system stopped at its iteration cap without
                                       * reaching the tolerance while the multipliers or the Lanczos products were formed -- S was applied inexactly and
                                       * the certificate was NOT accepted (xm_ctx_schur_info counts such products over the context's life) */

int xm_ctx_create(const xm_problem_t *prob, xm_ctx_t **out);          /* uploads / lays out Q on device 0 (or the rank's device) */
int xm_ctx_solve(xm_ctx_t *ctx, const xm_options_t *opt, xm_result_t *res);
void xm_ctx_destroy(xm_ctx_t *ctx);
/* out = alpha * Q * W through the storage the context holds (dense, BSR3 / sliced ELL, matrix-free; after a re-weighting: the
 * updated Q).  W, out: HOST, column-major 3n x o, o in 1, 3..10.  Diagnostic / test entry (single-rank contexts). */
int xm_ctx_qw(xm_ctx_t *ctx, int o, const double *W, double *out, double alpha);
int64_t xm_dense_ld(int64_t n);                                       /* padded leading dimension (doubles) of the device layout */

/* ---- XM^2 re-weighting on a resident context (reference loop 3_test_colmap_glomap.py:299-351: residual per
 * observation, 90-percentile filter at :321, rebuild Q, solve again).  For a view-graph Q = sum_e w_e G_e (Q_ii += w_e I,
 * Q_jj += w_e I, Q_ij = -w_e M_e, Q_ji = Q_ij^T) the rebuild is linear in the weights and happens on the device, in the
 * storage the context was created with (dense or BSR3, incl. its sliced-ELL copy): Q is never uploaded again.
 *   attach:    edges e = (ei[e], ej[e]), ei != ej, M: ne x 9 row-major; the stored pattern must hold both blocks of every edge and
 *              every diagonal block.  Does not change Q.
 *   residuals: res[e] = |Y_i - M_e Y_j|_F^2 with Y = s.*R of the LAST solve (the edge's share of <Q, Y Y^T> per unit weight)
 *   weights:   rewrites every edge block and every diagonal block from w (w[e] = 0 removes an observation).
 * XM_STORAGE_SCHUR contexts need no attach: their observations ARE the edges -- residuals: one per observation in input order,
 * |p^T U_i + t_i - P_l|^2 with the eliminated translations / landmarks of the last solution (what the reference computes from
 * recover_XM's p_est / t_est, 3_test_colmap_glomap.py:305-316); weights: one per observation (Q1, V1, Q3 and the reduced camera
 * Laplacian are rebuilt; the latter is re-inverted on the device). */
int xm_ctx_attach_edges(xm_ctx_t *ctx, int64_t ne, const int32_t *ei, const int32_t *ej, const double *M);
int xm_ctx_edge_residuals(xm_ctx_t *ctx, double *res);
int xm_ctx_set_edge_weights(xm_ctx_t *ctx, const double *w);
/* ---- the same loop with the REFERENCE's residual definition and sequencing (3_test_colmap_glomap.py:299-351).
 *   residuals_recovered: res[e] = squared distance of edge / observation e for the RECOVERED solution -- rot: 3 x 3n column-major anchored
 *              rotations and scale: n, as xm_recover_rotations returns them (the reference's R_real / s_real, :295-316).  Matrix-free
 *              contexts: |s_i R_i p + t_i - P_l|^2 with t, P eliminated (== the reference's landmarks_transformed vs p_est); view-graph
 *              contexts: |s_i R_i - M_e s_j R_j|_F^2.  Host array, input order.  Single-GPU contexts.
 *   xm2_filter: error = w .* res on the device, threshold = np.percentile(error, pct) (linear interpolation between two order
 *              statistics found by a device radix select, :321), weights of everything above it set to 0 and Q rebuilt (:323-338,
 *              without the reference's re-indexing of emptied landmarks / cameras: checklandmarks is upstream data cleaning).
 *   xm2_round: filter at `percentile` (0 -> 90), then the reference's second pass (:339-351): solve_rank3 at lam = 0, statistics of its
 *              scales; |mean(s[1:]) - 1| > 2 std(s[1:]) or more than 10 scales < 0.1  ->  lam = kept edges / n, else lam = 0; final solve
 *              with opt->max_rank / tol / max_time / flags (XM_FLAG_WARM_R in opt->flags: start it from the rank-3 result instead of the
 *              reference's cold start).  R (3n x r column-major), s (n): the solution the round starts from (what the first solve
 *              returned).  res: the final solve's result. */
typedef struct {
    uint32_t struct_size;
    double percentile;         /* in: 0 = the reference's 90 */
    double threshold;          /* out */
    int64_t removed;           /* out: edges newly removed by this round */
    double s_avg, s_std;       /* out: mean / standard deviation of the rank-3 scales s[1:] (:343-344) */
    int64_t n_small;           /* out: scales < 0.1 */
    int32_t regularised;       /* out: 1 = the final solve ran with lam = kept / n */
    int32_t rank3_status;
    double lam_used;
    int64_t rank3_tcg_iters;
} xm_xm2_info_t;
int xm_ctx_edge_residuals_recovered(xm_ctx_t *ctx, const double *rot, const double *scale, double *res);
int xm_ctx_xm2_filter(xm_ctx_t *ctx, const double *rot, const double *scale, double percentile, double *threshold, int64_t *removed,
                      double *w_out /* optional: the new weights */);
int xm_ctx_xm2_round(xm_ctx_t *ctx, const double *R, const double *s, int r, const xm_options_t *opt, xm_xm2_info_t *info,
                     xm_result_t *res);
/* Translations and landmarks of a solution: the last step of utils/recoversolution.py:recover_XM (lines 77-86,
 * ybar_est = Abar @ sR_real.T; t_est = [0 | first N-1 columns], p_est = the rest) for an XM_STORAGE_SCHUR context.  The reference
 * needs the dense (N-1+M) x 3N matrix Abar.bin that create_matrix writes (creatematrix.py:283-311; 80 GB at Final-13682 with 800 k
 * landmarks); here Abar = -Qtp_bar^-1 Vtp_bar^T is applied through the factor chain of the matrix-free product, O(observations +
 * N^2).  rot: 3 x 3n column-major and scale: n as returned by xm_recover_rotations; t: 3 x n column-major (t[:, 0] = 0, the
 * anchor), p: 3 x n_landmarks column-major.  Uses the context's CURRENT observation weights. */
int xm_ctx_recover_tp(xm_ctx_t *ctx, const double *rot, const double *scale, double *t, double *p);
/* XM_STORAGE_SCHUR: how the reduced camera Laplacian VT = Q2_bar - V3_bar Q3^-1 V3_bar^T (utils/creatematrix.py:150-166) is applied inside a
 * product -- *uses_cg = 0: through its dense inverse (set-up O(N^3), 8 (N-1)^2 bytes; up to xm_tuning_t.schur_dense_max cameras), 1: by
 * preconditioned CG on the matrix-free VT (no N^2 array) -- and, for the CG form, stats = {products so far, inner CG iterations
 * so far, products that stopped at the iteration cap instead of at the tolerance 1e-13} and the relative residual of the last product. */
int xm_ctx_schur_info(xm_ctx_t *ctx, int *uses_cg, int64_t stats[3], double *last_relres);

/* ================================================================== 3. kernel-level entry points (device pointers) */
/* device memory helpers so that callers need no other GPU runtime */
int xm_dev_count(int *count);
int xm_dev_alloc(void **ptr, size_t bytes);
int xm_dev_free(void *ptr);
int xm_dev_h2d(void *dst, const void *src, size_t bytes);
int xm_dev_d2h(void *dst, const void *src, size_t bytes);
int xm_dev_sync(void);

/* lay a host column-major symmetric Q out as the device row-major padded matrix (allocates *dq) */
int xm_dense_upload(const double *q_host, int64_t ldq, int64_t n, double **dq);
/* build the same device layout from a 3x3-block CSR description on the host (zero elsewhere) without ever forming
 * the dense matrix on the host — used to store a >= 10k-camera view-graph Q densely (13.5 GB at n = 13682) */
int xm_dense_from_bsr3(const int64_t *rowptr, const int32_t *colidx, const double *blocks, int64_t n, double **dq);
/* out = alpha * Q * W.  dq from xm_dense_upload; dW, dOut: device, row-major 3n x o (o in 1,3..10).
 * Replaces cublasDgemm via DnMatDnMat (Dense/matmul.h:42-87). stream: hipStream_t or NULL. */
int xm_qw_dense(const double *dq, int64_t n, int o, const double *dW, double *dOut, double alpha, void *stream);
/* the same product reading only the upper block triangle of a SYMMETRIC Q (o in 1, 3..5; allocates its scratch per call) */
int xm_qw_dense_sym(const double *dq, int64_t n, int o, const double *dW, double *dOut, double alpha, void *stream);
/* same product from 3x3-block CSR (device arrays; blocks row-major 9 doubles) */
int xm_qw_bsr3(const int64_t *d_rowptr, const int32_t *d_colidx, const double *d_blocks, int64_t n, int o,
               const double *dW, double *dOut, double alpha, void *stream);

/* A (host, column-major n x n, symmetric positive definite; lower triangle read) is overwritten by its inverse, computed on the device
 * (blocked Cholesky + triangular solves, xm-code_amd/csrc/xm_dense_la.hip): the set-up step of XM_STORAGE_SCHUR, which the reference
 * does on the host with scipy.linalg.solve (utils/creatematrix.py:260). */
int xm_spd_inverse(int64_t n, double *A);

/* Work decomposition of the half-traffic symmetric dense product (xm_qw_dense_sym, vertical sweep) for n cameras, host-only (CPU
 * test of the index arithmetic, tests/test_symv_layout.py): plan = { K steps (of two cameras) per chunk, Kf for the strip groups
 * dispatched last, first such group, chunks }. */
int xm_symv_plan(int64_t n, int32_t plan[4]);

/* Multi-rank symmetric dense product (xm-code_amd/csrc/xm_symw.h), host-only views for the CPU test tests/test_symw_plan.py: the work list of one
 * rank -- geom = {T, Th, tie, t0, nsteps, nstrips, K, number of items}, items: (strip, first step, end step) triples (NULL: sizes only) -- and the
 * predicate "row step t uses block (t, u)" of a matrix of T steps. */
int xm_symw_plan(int64_t ntot, int nloc, int cam0, int K, int32_t geom[8], int32_t *items);
int xm_symw_use(int T, int t, int u);
/* Large block-sparse Q: "sliced ELL over per-XCD column slabs" (xm-code_amd/csrc/xm_sell.h).  Same product as xm_qw_bsr3
 * (the reference has no sparse product: Dense/matmul.h:42-87 on a dense Q); the matrix is described on the HOST as 3x3-block CSR
 * (rows n, global columns in [0, ncols)) and re-laid on the device.  slabs in {1,2,4,8}; lmax = longest virtual row (hub
 * cameras are cut); gather_mode 0 (a record of W per lane) | 1 (LDS-transposed, the solver's default) selects how the rows of W are fetched.  xm_sell_layout is host-only (CPU tests): with
 * NULL arrays it fills sizes = {slices, steps, partial results, virtual rows, entries of the partial-result array}. */
int xm_sell_layout(const int64_t *rowptr, const int32_t *colidx, int64_t n, int64_t ncols, int slabs, int lmax, int64_t sizes[5],
                   int64_t *slice_off, int32_t *slab_start, uint8_t *kind, int64_t *src, int32_t *pslot, int64_t *pptr, int32_t *ridx);
int xm_sell_create(const int64_t *rowptr, const int32_t *colidx, const double *blocks, int64_t n, int64_t ncols, int slabs, int lmax,
                   void **handle);
/* the same with a block codec (xm-code_amd/csrc/xm_sell.h): codec 0 = 9 doubles per block, 1 = view-graph codec (off-diagonal blocks
 * -w * rotation stored as 4 doubles, diagonal blocks d * I as one double per camera; XM_ERR_ARG when the matrix is not of that form).
 * row0 = global camera index of row 0 (which column is "the diagonal"). */
int xm_sell_create2(const int64_t *rowptr, const int32_t *colidx, const double *blocks, int64_t n, int64_t ncols, int slabs, int lmax,
                    int codec, int64_t row0, void **handle);
/* host-only: distinct 128-byte lines of W the 64 lanes of a step touch, summed over every 4th step -- records of 72 bytes (o = 3), of 120 bytes
 * (o = 4, 5) at their native pitch, and at the 128-byte pitch: the figures behind the automatic choice of xm_tuning_t.sell_wpad */
int xm_sell_locality(const int64_t *rowptr, const int32_t *colidx, int64_t n, int64_t ncols, int slabs, int lmax, int64_t lines[3]);
/* host-only: block (row-major 3x3, -w * rotation) -> stored quaternion -> the block the product kernel rebuilds (CPU test of the codec) */
int xm_sell_quat_roundtrip(const double block[9], double quat[4], double rebuilt[9]);
void xm_sell_destroy(void *handle);
int xm_qw_sell(void *handle, int o, const double *dW, double *dOut, double alpha, int gather_mode, void *stream);
/* the same products with the input ALSO given at a record pitch of 16 doubles (dWpad16[cam * 16 + e] = dW[cam * 3 * pitch + e]; o = 3..5, layout 1;
 * NULL = not given): the gather reads one 128-byte line per camera.  Inside the solver the kernels that write the product input of the truncated
 * CG write that copy as well (xm_tuning_t.sell_wpad). */
int xm_qw_sell_padded(void *handle, int o, const double *dW, const double *dWpad16, double *dOut, double alpha, int gather_mode, void *stream);
/* per-camera kernels (device, row-major 3n x o; s: n):
 * Rout = MGS_rows(R + t*D) (Dense/batchedQR.h:42-67), sout = s*exp(t*ds/s) (trustregion.h:19-24), s[0] stays 1 */
int xm_retract(int64_t n, int o, const double *dR, const double *ds, const double *dD, const double *dds, double t,
               double *dRout, double *dsout, void *stream);
/* the same with the polar retraction (XM_RETRACT_POLAR): Rout_i = (M M^T)^{-1/2} M, M = R_i + t D_i */
int xm_retract_polar(int64_t n, int o, const double *dR, const double *ds, const double *dD, const double *dds, double t,
                     double *dRout, double *dsout, void *stream);
/* Solution recovery, the step right after the solve (replaces the rotation/scale part of
 * utils/recoversolution.py:recover_XM, lines 12-86): R (3n x r column-major) and s (n) as written to R.bin / s.bin ->
 * rot: 3 x 3n column-major, block i = the anchored orthogonal 3x3 of camera i (block 0 = identity), scale: n.
 * n_negative_det (optional) = cameras whose block had negative determinant before the majority sign flip. */
int xm_recover_rotations(int64_t n, int r, const double *R, const double *s, double *rot, double *scale, int *n_negative_det);

/* ================================================================== 4. multi-GPU row partition (one process per GPU) */
/* 128-byte unique id of the RCCL communicator: rank 0 calls xm_comm_unique_id and broadcasts the bytes
 * (bench.py does that with torch.distributed); every rank then calls xm_comm_init before xm_ctx_create. */
int xm_comm_unique_id(unsigned char id[128]);
int xm_comm_init(int rank, int world, int device, const unsigned char id[128], const char *rccl_path /* NULL = default search */);
/* Which transport joins the ranks of a context -- *kind: 0 none (one GPU), 1 RCCL all-gathers, 2 shared-memory TEST transport, 3 direct peer
 * writes between the host threads of this process (xm_problem_t.n_gpus), 4 direct peer writes between processes (IPC-mapped buffers) -- and,
 * in note (NUL-terminated, truncated to note_cap), why a faster transport was given up for it (empty: first choice).  The multi-GPU modes try
 * direct peer writes first (self-tested on the machine at context creation), then RCCL, and fail with XM_ERR_COMM naming both reasons. */
int xm_ctx_transport(xm_ctx_t *ctx, int *kind, char *note, size_t note_cap);
/* *on = 1 when the truncated CG of the last solved rank kept its product input at the 128-byte record pitch as well (xm_tuning_t.sell_wpad) */
int xm_ctx_sell_wpad(xm_ctx_t *ctx, int *on);
/* which product kernel serves the tCG of this context at rank o (3..10): XM_PRODUCT_* below */
#define XM_PRODUCT_DENSE       0   /* qw_dense_kernel */
#define XM_PRODUCT_DENSE_SYM   1   /* half-traffic symmetric dense product (one GPU: qw_symv_kernel; several: the cyclic half window) */
#define XM_PRODUCT_BSR3        2   /* qw_bsr3_kernel (3x3-block CSR, one launch) */
#define XM_PRODUCT_SELL        3   /* sliced ELL, 9 doubles per block (two launches) */
#define XM_PRODUCT_SELL_QUAT   4   /* sliced ELL, view-graph codec */
#define XM_PRODUCT_SCHUR       5   /* matrix-free factor chain */
int xm_ctx_product_kind(xm_ctx_t *ctx, int o, int *kind);

/* One process per GPU WITHOUT a collective library on the data path: every rank exports its exchange buffers as hipIpcMemHandle_t
 * through the POSIX shared-memory segment `name` (same string on every rank of the node, unique per job), maps the peers' and uses
 * the direct peer-write exchange of the single-process mode (fused into the tCG kernel).  xm_comm_init tries the same transport on
 * its own (segment name derived from the unique id) and keeps RCCL when any rank cannot map a peer, the transport's self-test
 * fails, the ranks span several nodes, or XM_COMM_PEER=0; this entry point has no fallback: XM_ERR_COMM instead.
 * spin_seconds <= 0: default bound (20 s) of the device-side waits.  Two ranks may share one GPU (tests). */
int xm_comm_init_ipc(int rank, int world, int device, const char *name, double spin_seconds);
/* TEST transport: the same collectives through a POSIX shared-memory segment, so several ranks can share one GPU on a
 * 1-GPU box (tests/test_gpu_parity.py::test_two_ranks_one_gpu); `bytes` = capacity of the exchange area */
int xm_comm_init_shm(int rank, int world, int device, const char *name, size_t bytes);
int xm_comm_finalize(void);
/* contiguous camera range [*c0, *c1) owned by `rank` for DENSE storage: equal ranges of ceil(n / world) cameras (the last rank is
 * padded with inert cameras inside the solver) */
int xm_partition(int64_t n, int world, int rank, int64_t *c0, int64_t *c1);
/* the same for block-sparse storage (XM_STORAGE_BSR3 / _VIEWGRAPH): ranges balanced by STORED BLOCKS (rowptr: n + 1 offsets), which
 * is what a context uses unless xm_tuning_t.balance == 1.  Host only. */
int xm_partition_blocks(int64_t n, const int64_t *rowptr, int world, int rank, int64_t *c0, int64_t *c1);

#ifdef __cplusplus
}
#endif
#endif /* XM_AMD_H */
