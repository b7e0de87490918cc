// xm_schur.h — matrix-free Q for the XM solve (the reference's own "TODO: implement sparse Q matrix
// construction", utils/creatematrix.py:54-56).
//
// The reference forms the dense 3N x 3N Schur complement Q = Q1 - Vtp_bar Qtp_bar^{-1} Vtp_bar^T on the host
// (creatematrix.py:137-339: Q1, V1, V2, V3, the bipartite Laplacian Qtp, two solves with the reduced camera Laplacian
// VT = Q2_bar - V3_bar Q3^{-1} V3_bar^T) and the solver multiplies with it (72 N^2 bytes per product).  Here the factors stay
// factors: the context is created from the observation list itself (the arguments of create_matrix: camera, landmark,
// camera-frame point, weight) and a product Y = alpha * Q * W is the chain
//      h_l   = -(1/Q3_l) sum_{obs of l} w (p . W_i)                       per landmark      (V2^T W, landmarks eliminated)
//      r_i   =  c_i . W_i + sum_{obs of i} w h_l                          per camera i >= 1 (right-hand side of the camera system)
//      x_cam =  VT^{-1} r                                                 dense (N-1)^2, the existing dense Q*W kernel
//      x_l   =  h_l + (1/Q3_l) sum_{obs of l} w x_cam_i                   per landmark
//      Y_i   =  Q1_i W_i - c_i x_cam_i + sum_{obs of i} w p x_l           per camera, + the fused epilogue of the Q*W kernels
// i.e. O(observations) + (N-1)^2 bytes per product instead of 9 N^2: at N = 13682 cameras / 4.5 M observations 0.4 GB + 1.5 GB
// against 13.5 GB.  Single GPU in this version.  VT is assembled on the host from the co-visibility lists (O(sum of squared
// landmark degrees)) and inverted on the device at context creation (xm_dense_la.hip: blocked Cholesky + two blocked triangular
// solves, O(N^3): 2 s at N = 14 000).
#pragma once

#include <cstdint>
#include <vector>

#include "xm_solver.h"

namespace xm {

constexpr int64_t kSchurMaxCams = 40000;   // (N-1)^2 inverse + workspace = 3 x 8 N^2 bytes during set-up: 38 GB at the limit

// A (device, column-major n x n, lower triangle read) -> Cholesky factor; X <- the LOWER triangle of A^-1 (above the diagonal: scratch).
// false: not positive definite
bool spd_inverse_device(int n, double *A, double *X, hipStream_t st, bool trace = false);   // trace: phase times on stderr
// dst (row-major n x n, leading dimension ldd) <- the full symmetric inverse from X's lower triangle
void spd_inverse_layout(int n, const double *X, double *dst, int64_t ldd, hipStream_t st);
// A (device, column-major n x n) -= q * u u^T  (u: device, n doubles)
void rank1_sub_device(int n, double *A, const double *u, double q, hipStream_t st);

struct SchurSettings {          // from xm_tuning_t (Settings::resolve)
    bool host_assembly = false; // assemble the reduced camera Laplacian on the host (the reference's route, utils/creatematrix.py:137-260; tests)
    int64_t sym_min_rows = 4096; // VT^-1 is applied with the half-traffic symmetric kernel from this many rows on
    bool trace = false;         // set-up phase times on stderr (scripts/kbench_schur.py)
    int solver = 0;             // reduced camera system inside the product: 0 by size (dense inverse up to dense_max cameras, CG above) | 1 dense inverse | 2 preconditioned CG
    int64_t dense_max = 20000;  // (the dense inverse costs 8 (N-1)^2 bytes -- 3.2 GB here -- and an O(N^3) set-up; the CG form nothing but the observation lists)
    int pcg_first = 0;          // CG form: iterations of the first batch of a context's first product (0 = 26; tests force top-up batches with a small one)
    int pcg_hess_digits = 0;    // CG form: relative residual 10^-digits of the inner solve inside HESSIAN products (0 = 9; gradient, cost and certificate products: 13)
};
struct SchurLm;                 // xm_schur.hip

class SchurOp {
public:
    // cam / lm: 0-based indices of the nobs observations, p: nobs x 3 (row-major), w: nobs
    // comm (not owned; may be null): the communicator of a row-partitioned context -- every rank holds all the observations and repeats
    // the landmark kernels, multiplies its own rows of VT^-1 only and all-gathers x_cam; the last kernel runs for the rank's cameras
    // (CamArgs.nloc / cam0)
    SchurOp(int64_t n, int64_t n_landmarks, int64_t nobs, const int32_t *cam, const int32_t *lm, const double *p, const double *w,
            hipStream_t st, Comm *comm = nullptr, const SchurSettings &cfg = SchurSettings());
    // Y = alpha * Q * W for all n cameras (W: camera records of 3 * pitch_of(o) doubles), same CamArgs / epilogue contract and
    // per-workgroup partial sums (grid qw_grid(n)) as launch_qw_dense
    void product(int o, int epi, const double *W, double alpha, const CamArgs &a, hipStream_t st);
    // XM^2 loop on the reference's own Q: per-observation residuals at U = s.*R (input order of the observations) and new weights
    void residuals(int o, const double *U, double *res_host, const CamArgs &a, hipStream_t st);
    const double *residuals_device(int o, const double *U, const CamArgs &a, hipStream_t st);   // the same, left on the device
    void set_weights(const double *w, hipStream_t st);
    // translations t (3 x n column-major, camera 0 at the origin) and landmarks p (3 x m column-major) of a rank-3 solution given as
    // anchored rotations rot (3 x 3n column-major) and scales (n): the eliminated variables of the chain at U = (s.*R)^T, negated
    void recover_tp(const double *rot, const double *scale, double *t, double *p, hipStream_t st);
    int64_t n_landmarks() const { return m_; }
    int64_t nobs() const { return nobs_; }
    int64_t bytes_per_product(int o) const;
    // CG form of the reduced camera system: statistics since construction (products, inner iterations, products that hit the iteration cap)
    // and the relative residual of the last one
    bool uses_pcg() const { return pcg_; }
    void pcg_stats(int64_t out[3], double *relres) const { out[0] = pcg_products_; out[1] = pcg_iters_total_; out[2] = pcg_unconverged_; if (relres) *relres = pcg_last_relres_; }

private:
    int64_t n_ = 0, m_ = 0, nobs_ = 0, nred_ = 0, ldv_ = 0;   // nred = cameras of the padded (N-1) system / 3
    Comm *comm_ = nullptr;
    int world_ = 1, rank_ = 0;
    int64_t nred_loc_ = 0, nred_pad_ = 0;                     // pseudo-cameras per rank, padded total
    DevBuf<int64_t> cam_ptr_, lm_ptr_;
    DevBuf<int32_t> cam_lm_, lm_cam_;         // by camera: landmark of each observation; by landmark: camera
    // landmarks are numbered by degree (descending) on the device; the first nheavy_ (more than kSchurHeavy observations) keep contiguous
    // lists (lm_ptr_), the others are packed 64 to a group (gbase_, ldeg_): xm_schur.hip, SchurLm
    DevBuf<int64_t> gbase_;
    DevBuf<int32_t> ldeg_;
    int64_t nheavy_ = 0, ltotal_ = 1;
    std::vector<int32_t> slot_of_;            // landmark -> device number
    std::vector<int64_t> dpos_l_;             // observation -> position in the device's by-landmark arrays
    DevBuf<double> cam_w_, cam_p_, lm_w_, lm_p_;
    DevBuf<double> Q1_, c_, q3inv_;
    DevBuf<double> vtinv_;                     // (N-1)^2 inverse in the dense kernel's padded row-major layout
    DevBuf<double> h_, r_, xc_, xl_, res_;
    DevBuf<double> sym_prow_, sym_pcol_;       // partial sums of the half-traffic product with VT^-1 (large N only)
    bool vt_sym_ = false;
    DevBuf<int32_t> obs_cam_, obs_lm_;        // the observations in input order (residual kernel)
    DevBuf<double> obs_p_;
    std::vector<int32_t> hcam_, hlm_, lcam_;  // host copies of the structure for set_weights
    std::vector<double> hp_;
    std::vector<int64_t> cp_, lp_, pos_c_, pos_l_;
    std::vector<int64_t> cam_obs_;            // observation (input index) at each position of the by-camera lists
    int o_alloc_ = 0, o_last_ = 0;
    void ensure(int o);
    // weight-dependent factors assembled on the device (set_weights_device); the host version serves lists with duplicate (camera, landmark) pairs
    DevBuf<int64_t> pos_c_dev_, dpos_l_dev_;
    DevBuf<double> w_in_, q2_;
    bool dup_pairs_ = false;
    SchurSettings cfg_;
    std::vector<int64_t> hub_lm_, hub_obs_ptr_, hub_obs_;
    void set_weights_device(const double *w, hipStream_t st);
    // preconditioned CG on the matrix-free reduced camera Laplacian (cfg.solver; xm_schur.hip)
    bool pcg_ = false;
    DevBuf<double> pcg_dinv_, pcg_r_, pcg_p_, pcg_ap_, pcg_parts_;
    DevBuf<int32_t> pcg_state_;
    struct PcgState *pcg_host_ = nullptr;   // pinned copy of the device state word
    int pcg_grid_ = 1, pcg_last_iters_[2] = {24, 24}, pcg_max_iters_ = 1000;   // [0] products at the tight tolerance, [1] Hessian products
    double pcg_tol_[2] = {1e-13, 1e-9}, pcg_last_relres_ = 0.0;
    int64_t pcg_products_ = 0, pcg_iters_total_ = 0, pcg_unconverged_ = 0;
    template <int O> void pcg_solve(const SchurLm &L, const struct TcgScal *sc, hipStream_t st, int kind);
public:
    ~SchurOp();
    SchurOp(const SchurOp &) = delete;
    SchurOp &operator=(const SchurOp &) = delete;
};

}  // namespace xm
