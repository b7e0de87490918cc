/*
 * xm_oracle.h — CPU ORACLE for the XM Burer-Monteiro / Riemannian-staircase solve.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The shipped HIP path never calls it.
 *
 * It is a plain-C restatement (double precision, column-major, same operation order
 * where that is cheap) of the reference's algorithm:
 *     XM/include/XM/trustregion.h   (RTR + Steihaug-Toint tCG)        -> xmo_trustregion
 *     XM/include/XM/checkeig.h      (dual certificate)                -> xmo_checkeig
 *     XM/src/XM_main.cu             (staircase drivers, .bin I/O)     -> xmo_solve*, xmo_*_path
 * The reference has no CPU implementation and cannot be built here (CUDA TU + cuBLAS /
 * cuSOLVER / cuSPARSE + Eigen3; none present), so there is no oracle/_ref binary.
 * Third-party arithmetic that is not under /root/reference is restated from its published
 * algorithm: cublasDgemm/Ddot/Daxpy (plain loops), cusolverDnXsyevd (Householder
 * tridiagonalisation + implicit QL), Eigen::LeastSquaresConjugateGradient (Eigen3,
 * version unpinned by the reference: find_package(Eigen3 REQUIRED), XM/CMakeLists.txt:26).
 *
 * PARITY PINNING: the reference ships no tests / golden outputs, so parity is
 * pinned by (a) the implementation-independent optimality certificate and (b) golden
 * fixtures under tests/golden generated in the build container (see tests/golden/README.md).
 */
#ifndef XM_ORACLE_H
#define XM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* per-outer-iteration trace record (6 doubles each) */
#define XMO_TRACE_STRIDE 6 /* loss, gradnorm, inner_iters(i+1), endreason, trstatus, delta */

typedef struct {
    int32_t outer_iters;   /* value of k when the outer loop was left                    */
    int32_t stop_reason;   /* 5 rdotr<1e-15 | 10 gradnorm<gradtol | 11 time | 12 loss_qu>=0
                              | 13 delta<1e-20 | 14 max outer | -1 line search failed     */
    int64_t tcg_iters;     /* reference's "Total iteration" (sum of i+1, tr.h:666)        */
    int64_t qw_products;   /* number of C*W products issued                               */
    double  seconds;       /* wall clock of the outer loop (tr.h:451,712)                 */
    double  qw_seconds;    /* wall clock spent inside the C*W products                    */
    int32_t trace_cap;     /* capacity of trace in records (0 = no trace)                 */
    int32_t trace_len;
    double *trace;         /* caller-allocated, trace_cap*XMO_TRACE_STRIDE doubles        */
} xmo_stats;

typedef struct {
    double min_eig;        /* W[0] of syevd (ce.h:317)                                    */
    double dual;           /* "new dual" (ce.h:322-333)                                   */
    double gap;            /* ce.h:336                                                    */
    double ls_residual;    /* ||Acell*y - vec(Z*sR)||_2 of the multiplier solve           */
    int32_t lscg_iters;
    int32_t accepted;
} xmo_cert;

/* flags */
#define XMO_VERBOSE        1u
#define XMO_FIX_STALE_SR   2u  /* recompute sR after the escalation line search (reference does not, tr.h:394-422) */
#define XMO_CLOSED_FORM_Y  4u  /* multipliers by the per-camera closed form instead of restated LSCG */

/* tr.h:77  XMtrustregion.  C: 3n x 3n col-major.  R0,R: 3n x o col-major.  s0_ex,s_ex: n (entry 0 == 1).
 * v: 3n (only read when linesearch_step != 0).  gradtol is in/out (tr.h:534).  Returns 0. */
/* test-only extension: multiply from a 3x3-block CSR description instead of the dense C (NULL, NULL, NULL to unset) */
void xmo_set_bsr(const int64_t *rowptr, const int32_t *colidx, const double *blocks);
int xmo_trustregion(int n, int o, const double *C, const double *R0, const double *s0_ex,
                    double *R, double *s_ex, double lam, double *gradtol, double linesearch_step,
                    const double *v, double *primal, double maxtime, xmo_stats *st, unsigned flags);

/* ce.h:42 checkeig.  sR: 3n x o col-major.  v (3n) receives eigvec of lambda_min.  Returns 1 accepted / 0. */
int xmo_checkeig(int n, int o, const double *C, const double *sR, double lam, double *v,
                 double primal, xmo_cert *cert, unsigned flags);

/* main.cu:180/312/35.  mode 0 solve, 1 solve_rank3, 2 solve_rebuttle (s_ini used, R_ini ignored like the reference).
 * R_out: 3n x max_rank col-major (leading dim 3n), s_out: n.  Returns the solve_rebuttle status code. */
int xmo_solve(int n, const double *C, unsigned max_rank, double tol, double lam, double max_time, int mode,
              const double *s_ini, double *R_out, double *s_out, int *rank_out,
              xmo_stats *st_total, xmo_cert *last_cert, unsigned flags);

/* file-based surface == the pybind functions (main.cu:403-408). Return <0 on I/O error. */
int xmo_solve_path(const char *dataset_path, unsigned max_rank, double tol, double lam, double max_time,
                   int mode, unsigned flags);

/* helpers exported for unit tests */
void xmo_qw(int n, int o, const double *C, const double *W, double *out, double alpha);        /* Dense/matmul.h:42 */
void xmo_mgs_rows(int n, int o, const double *A, double *Qm);                                   /* Dense/batchedQR.h:42-67 on 3n x o col-major */
/* eigen step of xmo_checkeig: NULL = the restated tred2 / tql2 (default), else the caller's routine with the contract of xmo_syev_lower */
typedef int (*xmo_syev_fn)(int m, double *A, double *w);
void xmo_set_syev(xmo_syev_fn f);
int  xmo_syev_lower(int m, double *A, double *w);                                               /* Dense/eig.h:35 (vectors overwrite A, ascending) */
int  xmo_read_bin(const char *file, double **data, int *rows, int *cols);                       /* main.cu:18 */
int  xmo_write_bin(const char *file, const double *data, int rows, int cols);                   /* main.cu:284-305 */
void xmo_free(void *p);
int  xmo_num_threads(void);
/* host-bandwidth mode of the dense product for bench.py's cpu_baseline (never used by the parity tests): a NUMA-distributed,
 * first-touch copy of C; while prepared, xmo_qw on a matrix of that size multiplies from it.  0 on success. */
int  xmo_numa_prepare(int n, const double *C);
void xmo_numa_release(void);

#ifdef __cplusplus
}
#endif
#endif
