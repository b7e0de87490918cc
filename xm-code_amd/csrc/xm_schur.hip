// xm_schur.hip — matrix-free Q*W from the observation list (design: xm_schur.h).  Replaces, for the reference's own Q
// (utils/creatematrix.py:51-339), the dense product Dense/matmul.h:42-87 by the factor chain; the fused epilogues are those of
// xm_device.h, so the solver above (trust region, certificate, Lanczos) is unchanged.
#include "xm_schur.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "xm_device.h"

namespace xm {

// ------------------------------------------------------------------------------------------------------------------
// device kernels
// ------------------------------------------------------------------------------------------------------------------
constexpr int kSchurHeavy = 64;   // landmarks with more observations get a workgroup of their own (a thread per landmark serialises
                                  // them: 963 us per product with three landmarks seen by all 1778 cameras, 122 us once split)

// h_l = -(1/Q3_l) sum_{obs of l} w (p . W_i)      HEAVY 0: one thread per landmark (heavy ones skipped) | 1: one 1024-thread
// workgroup per listed heavy landmark (a landmark seen by all 13 682 cameras: 14 strided steps instead of 13 682 serial ones;
// thread-strided order, DPP tree per wavefront, the 16 wavefront sums added in a fixed order)
constexpr int kSchurHeavyThreads = 1024;
template <int O>
__device__ __forceinline__ bool heavy_block_sum(double (&acc)[O]) {   // result valid in thread 0 (returns true there)
    __shared__ double part[kSchurHeavyThreads / 64][O];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < O; ++k) acc[k] = wave_sum(acc[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < O; ++k) part[wv][k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x != 0) return false;
#pragma unroll
    for (int k = 0; k < O; ++k) {
        double t = 0.0;
        for (int q = 0; q < kSchurHeavyThreads / 64; ++q) t += part[q][k];
        acc[k] = t;
    }
    return true;
}
template <int O, int HEAVY>
__global__ __launch_bounds__(HEAVY ? kSchurHeavyThreads : 256) void schur_lm_h_kernel(int64_t m, const int32_t *__restrict__ heavy, const int64_t *__restrict__ lm_ptr,
                                                          const int32_t *__restrict__ lm_cam, const double *__restrict__ lm_w,
                                                          const double *__restrict__ lm_p, const double *__restrict__ q3inv,
                                                          const double *__restrict__ W, const TcgScal *__restrict__ scal, double *__restrict__ h) {
    constexpr int OP = pitch_of(O);
    if (scal != nullptr) {
        if (scal->status != 0) return;
    }
    int64_t l = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (HEAVY) l = heavy[blockIdx.x];
    else if (l >= m || lm_ptr[l + 1] - lm_ptr[l] > kSchurHeavy) return;
    double acc[O];
#pragma unroll
    for (int k = 0; k < O; ++k) acc[k] = 0.0;
    for (int64_t e = lm_ptr[l] + (HEAVY ? (int)threadIdx.x : 0); e < lm_ptr[l + 1]; e += (HEAVY ? kSchurHeavyThreads : 1)) {
        const double *Wi = W + (size_t)lm_cam[e] * 3 * OP;
        const double w = lm_w[e], p0 = lm_p[3 * e], p1 = lm_p[3 * e + 1], p2 = lm_p[3 * e + 2];
#pragma unroll
        for (int k = 0; k < O; ++k) acc[k] += w * (p0 * Wi[k] + p1 * Wi[OP + k] + p2 * Wi[2 * OP + k]);
    }
    const double qi = q3inv[l];
    if (HEAVY) {
        if (!heavy_block_sum<O>(acc)) return;
    }
#pragma unroll
    for (int k = 0; k < O; ++k) h[(size_t)l * OP + k] = -acc[k] * qi;
}

// r_{i-1} = c_i . W_i + sum_{obs of i} w h_l      one wavefront per camera (a camera has hundreds of observations; fixed lane-strided
// order + DPP tree: bit-reproducible)
template <int O>
__global__ __launch_bounds__(256) void schur_cam_r_kernel(int n, const int64_t *__restrict__ cam_ptr, const int32_t *__restrict__ cam_lm,
                                                           const double *__restrict__ cam_w, const double *__restrict__ c,
                                                           const double *__restrict__ W, const double *__restrict__ h,
                                                           const TcgScal *__restrict__ scal, double *__restrict__ r) {
    constexpr int OP = pitch_of(O);
    if (scal != nullptr) {
        if (scal->status != 0) return;
    }
    const int gl = threadIdx.x & 63, cam = blockIdx.x * kQwWaves + (threadIdx.x >> 6);
    double acc[O];
#pragma unroll
    for (int k = 0; k < O; ++k) acc[k] = 0.0;
    if (cam < n && cam >= 1)
        for (int64_t e = cam_ptr[cam] + gl; e < cam_ptr[cam + 1]; e += 64) {
            const double w = cam_w[e];
            const double *hl = h + (size_t)cam_lm[e] * OP;
#pragma unroll
            for (int k = 0; k < O; ++k) acc[k] += w * hl[k];
        }
#pragma unroll
    for (int k = 0; k < O; ++k) acc[k] = wave_sum(acc[k]);
    if (cam < n && cam >= 1 && gl == 0) {
        const double *Wi = W + (size_t)cam * 3 * OP, *ci = c + (size_t)cam * 3;
#pragma unroll
        for (int k = 0; k < O; ++k) r[(size_t)(cam - 1) * OP + k] = acc[k] + ci[0] * Wi[k] + ci[1] * Wi[OP + k] + ci[2] * Wi[2 * OP + k];
    }
}

// x_l = h_l + (1/Q3_l) sum_{obs of l} w x_cam_i     (x_cam of the anchor camera 0 is 0: its translation is the gauge)
template <int O, int HEAVY>
__global__ __launch_bounds__(HEAVY ? kSchurHeavyThreads : 256) void schur_lm_x_kernel(int64_t m, const int32_t *__restrict__ heavy, const int64_t *__restrict__ lm_ptr,
                                                          const int32_t *__restrict__ lm_cam, const double *__restrict__ lm_w,
                                                          const double *__restrict__ q3inv, const double *__restrict__ h,
                                                          const double *__restrict__ xc, const TcgScal *__restrict__ scal,
                                                          double *__restrict__ xl) {
    constexpr int OP = pitch_of(O);
    if (scal != nullptr) {
        if (scal->status != 0) return;
    }
    int64_t l = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (HEAVY) l = heavy[blockIdx.x];
    else if (l >= m || lm_ptr[l + 1] - lm_ptr[l] > kSchurHeavy) return;
    double acc[O];
#pragma unroll
    for (int k = 0; k < O; ++k) acc[k] = 0.0;
    for (int64_t e = lm_ptr[l] + (HEAVY ? (int)threadIdx.x : 0); e < lm_ptr[l + 1]; e += (HEAVY ? kSchurHeavyThreads : 1)) {
        const int i = lm_cam[e];
        if (i == 0) continue;
        const double w = lm_w[e];
        const double *xi = xc + (size_t)(i - 1) * OP;
#pragma unroll
        for (int k = 0; k < O; ++k) acc[k] += w * xi[k];
    }
    const double qi = q3inv[l];
    if (HEAVY) {
        if (!heavy_block_sum<O>(acc)) return;
    }
#pragma unroll
    for (int k = 0; k < O; ++k) xl[(size_t)l * OP + k] = h[(size_t)l * OP + k] + acc[k] * qi;
}

// Y_i = Q1_i W_i - c_i x_cam_i + sum_{obs of i} w p x_l, then the common tail of the Q*W kernels (xm_device.h)
template <int O, int EPI>
__global__ __launch_bounds__(256) void schur_cam_y_kernel(const int64_t *__restrict__ cam_ptr, const int32_t *__restrict__ cam_lm,
                                                           const double *__restrict__ cam_w, const double *__restrict__ cam_p,
                                                           const double *__restrict__ Q1, const double *__restrict__ c,
                                                           const double *__restrict__ W, const double *__restrict__ xc,
                                                           const double *__restrict__ xl, double alpha, CamArgs a) {
    constexpr int OP = pitch_of(O);
    if (EPI == EPI_HESS) {
        if (a.scal->status != 0) return;
    }
    __shared__ double red[kQwWaves][3];
    const int gl = threadIdx.x & 63, slot = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cam = blockIdx.x * kQwWaves + slot;
    const bool active = cam < a.nloc;
    EpiOps eops;
    epi_prefetch<O, EPI>(eops, active ? cam : 0, gl, active, a);
    double acc[3][O];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < O; ++k) acc[r][k] = 0.0;
    if (active) {
        for (int64_t e = cam_ptr[cam] + gl; e < cam_ptr[cam + 1]; e += 64) {
            const double w = cam_w[e];
            const double *x = xl + (size_t)cam_lm[e] * OP;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double wp = w * cam_p[3 * e + r];
#pragma unroll
                for (int k = 0; k < O; ++k) acc[r][k] += wp * x[k];
            }
        }
        if (gl == 0) {
            const double *Wi = W + (size_t)cam * 3 * OP, *q = Q1 + (size_t)cam * 9, *ci = c + (size_t)cam * 3;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int k = 0; k < O; ++k) {
                    const double xk = (cam >= 1) ? xc[(size_t)(cam - 1) * OP + k] : 0.0;
                    acc[r][k] += q[3 * r] * Wi[k] + q[3 * r + 1] * Wi[OP + k] + q[3 * r + 2] * Wi[2 * OP + k] - ci[r] * xk;
                }
        }
    }
    qw_finish<O, EPI, 64, kQwWaves>(cam, gl, slot, active, acc, alpha, a, eops, red);
}

// ------------------------------------------------------------------------------------------------------------------
// host: factors from the observation list (utils/creatematrix.py:62-175, restated on the observation level)
// ------------------------------------------------------------------------------------------------------------------
SchurOp::SchurOp(int64_t n, int64_t n_landmarks, int64_t nobs, const int32_t *cam, const int32_t *lm, const double *p, const double *w,
                 hipStream_t st) {
    if (n < 1 || n_landmarks < 1 || nobs < 1 || !cam || !lm || !p || !w) throw Error(XM_ERR_ARG, "matrix-free Q: bad observation list");
    if (n > kSchurMaxCams) throw Error(XM_ERR_ARG, "matrix-free Q: more than " + std::to_string(kSchurMaxCams) + " cameras");
    n_ = n; m_ = n_landmarks; nobs_ = nobs;
    const int64_t N = n, M = n_landmarks;
    // ---- structure (fixed for the life of the context): observation lists by camera and by landmark, input order inside each
    // (fixed summation order); pos_*[e] = where observation e of the input sits in them (weights are re-scattered by set_weights)
    hcam_.assign(cam, cam + nobs); hlm_.assign(lm, lm + nobs); hp_.assign(p, p + 3 * nobs);
    cp_.assign((size_t)N + 1, 0); lp_.assign((size_t)M + 1, 0);
    for (int64_t e = 0; e < nobs; ++e) {
        const int64_t i = cam[e], l = lm[e];
        if (i < 0 || i >= N || l < 0 || l >= M) throw Error(XM_ERR_ARG, "matrix-free Q: observation index out of range");
        cp_[(size_t)i + 1]++; lp_[(size_t)l + 1]++;
    }
    for (int64_t i = 0; i < N; ++i) cp_[(size_t)i + 1] += cp_[(size_t)i];
    for (int64_t l = 0; l < M; ++l) lp_[(size_t)l + 1] += lp_[(size_t)l];
    std::vector<int32_t> c_lm((size_t)nobs);
    std::vector<double> c_p((size_t)nobs * 3), l_p((size_t)nobs * 3);
    lcam_.assign((size_t)nobs, 0);
    pos_c_.assign((size_t)nobs, 0); pos_l_.assign((size_t)nobs, 0);
    {
        std::vector<int64_t> nc(cp_.begin(), cp_.end() - 1), nl(lp_.begin(), lp_.end() - 1);
        for (int64_t e = 0; e < nobs; ++e) {
            const int64_t a2 = nc[(size_t)cam[e]]++, b2 = nl[(size_t)lm[e]]++;
            pos_c_[(size_t)e] = a2; pos_l_[(size_t)e] = b2;
            c_lm[(size_t)a2] = lm[e]; std::memcpy(&c_p[(size_t)a2 * 3], p + 3 * e, 24);
            lcam_[(size_t)b2] = cam[e]; std::memcpy(&l_p[(size_t)b2 * 3], p + 3 * e, 24);
        }
    }
    auto up = [&](auto &buf, const auto &v) {
        buf.alloc(std::max<size_t>(v.size(), 1), false);
        if (!v.empty()) XM_HIP_CHECK(hipMemcpy(buf.p, v.data(), v.size() * sizeof(v[0]), hipMemcpyHostToDevice));
    };
    up(cam_ptr_, cp_); up(lm_ptr_, lp_); up(cam_lm_, c_lm); up(lm_cam_, lcam_); up(cam_p_, c_p); up(lm_p_, l_p);
    up(obs_cam_, hcam_); up(obs_lm_, hlm_); up(obs_p_, hp_);
    std::vector<int32_t> heavy;
    for (int64_t l = 0; l < M; ++l)
        if (lp_[(size_t)l + 1] - lp_[(size_t)l] > kSchurHeavy) heavy.push_back((int32_t)l);
    nheavy_ = (int64_t)heavy.size();
    up(heavy_, heavy);
    cam_w_.alloc((size_t)nobs, false); lm_w_.alloc((size_t)nobs, false);
    Q1_.alloc((size_t)N * 9, false); c_.alloc((size_t)N * 3, false); q3inv_.alloc((size_t)M, false);
    const int64_t mr = N - 1;
    nred_ = std::max<int64_t>(1, (mr + 2) / 3);
    ldv_ = dense_ld(nred_);
    vtinv_.alloc((size_t)3 * nred_ * (size_t)ldv_);
    set_weights(w, st);
}

// Everything that depends on the weights (utils/creatematrix.py:62-175 restated on the observation level): Q1, c (= V1), Q3, the
// reduced camera Laplacian VT = Q2_bar - V3_bar Q3^{-1} V3_bar^T and its inverse.  Called by the constructor and by the XM^2 loop
// (observations filtered by weight 0).  A landmark whose observations all have weight 0 drops out (1/Q3 := 0); a camera without
// weight makes VT singular -> XM_ERR_ARG.
void SchurOp::set_weights(const double *w, hipStream_t st) {
    if (!w) throw Error(XM_ERR_ARG, "matrix-free Q: null weights");
    const int64_t N = n_, M = m_, nobs = nobs_;
    std::vector<double> Q1((size_t)N * 9, 0.0), c((size_t)N * 3, 0.0), Q2((size_t)N, 0.0), Q3((size_t)M, 0.0);
    std::vector<double> c_w((size_t)nobs), l_w((size_t)nobs);
    for (int64_t e = 0; e < nobs; ++e) {
        if (!(w[e] >= 0.0)) throw Error(XM_ERR_ARG, "matrix-free Q: negative or NaN weight");
        const int64_t i = hcam_[(size_t)e], l = hlm_[(size_t)e];
        const double *pe = &hp_[(size_t)e * 3];
        for (int a = 0; a < 3; ++a) {
            c[(size_t)i * 3 + a] += w[e] * pe[a];                                                 // V1 block (creatematrix.py:27)
            for (int b = 0; b < 3; ++b) Q1[(size_t)i * 9 + 3 * a + b] += w[e] * pe[a] * pe[b];    // Q1 block (:26)
        }
        Q2[(size_t)i] += w[e]; Q3[(size_t)l] += w[e];                                             // :68-69
        c_w[(size_t)pos_c_[(size_t)e]] = w[e]; l_w[(size_t)pos_l_[(size_t)e]] = w[e];
    }
    std::vector<double> q3inv((size_t)M);
    for (int64_t l = 0; l < M; ++l) q3inv[(size_t)l] = (Q3[(size_t)l] > 0.0) ? 1.0 / Q3[(size_t)l] : 0.0;
    const int64_t mr = N - 1;
    std::vector<double> VT((size_t)std::max<int64_t>(mr, 1) * (size_t)std::max<int64_t>(mr, 1), 0.0);
    for (int64_t i = 1; i < N; ++i) VT[(size_t)(i - 1) + (size_t)(i - 1) * mr] = Q2[(size_t)i];
    for (int64_t l = 0; l < M; ++l) {
        const double qi = q3inv[(size_t)l];
        if (qi == 0.0) continue;
        for (int64_t e1 = lp_[(size_t)l]; e1 < lp_[(size_t)l + 1]; ++e1) {
            const int64_t a2 = lcam_[(size_t)e1];
            if (a2 == 0 || l_w[(size_t)e1] == 0.0) continue;
            for (int64_t e2 = lp_[(size_t)l]; e2 < lp_[(size_t)l + 1]; ++e2) {
                const int64_t b2 = lcam_[(size_t)e2];
                if (b2 == 0) continue;
                VT[(size_t)(a2 - 1) + (size_t)(b2 - 1) * mr] -= l_w[(size_t)e1] * l_w[(size_t)e2] * qi;
            }
        }
    }
    auto put = [&](DevBuf<double> &buf, const std::vector<double> &v) {
        if (!v.empty()) XM_HIP_CHECK(hipMemcpy(buf.p, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice));
    };
    put(cam_w_, c_w); put(lm_w_, l_w); put(Q1_, Q1); put(c_, c); put(q3inv_, q3inv);
    if (mr > 0) {   // invert on the device (blocked Cholesky, xm_dense_la.hip), then lay the inverse out like a dense Q
        DevBuf<double> tmp, inv;
        tmp.alloc((size_t)mr * mr, false); inv.alloc((size_t)mr * mr, false);
        XM_HIP_CHECK(hipMemcpy(tmp.p, VT.data(), (size_t)mr * mr * sizeof(double), hipMemcpyHostToDevice));
        std::vector<double>().swap(VT);
        if (!spd_inverse_device((int)mr, tmp.p, inv.p, st))
            throw Error(XM_ERR_ARG, "matrix-free Q: the reduced camera Laplacian is not positive definite (observation graph not connected?)");
        launch_transpose_pad(inv.p, mr, mr, mr, vtinv_.p, ldv_, st);   // (symmetric: the transposition is immaterial)
        XM_HIP_CHECK(hipStreamSynchronize(st));
    }
}

// residual of every observation at the point whose scaled rows are U (camera records of 3 * pitch_of(o) doubles):
// |p^T U_i + t_i - P_l|^2 with the eliminated translations / landmarks [t; P] = -Qtp_bar^{-1} Vtp_bar^T U, i.e. the observation's share
// of <Q, U U^T> per unit weight (what the reference's XM^2 loop computes from recover_XM's p_est / t_est, 3_test_colmap_glomap.py:305-316)
template <int O>
__global__ __launch_bounds__(256) void schur_obs_residual_kernel(int64_t nobs, const int32_t *__restrict__ cam, const int32_t *__restrict__ lm,
                                                                  const double *__restrict__ p, const double *__restrict__ U,
                                                                  const double *__restrict__ xc, const double *__restrict__ xl,
                                                                  double *__restrict__ res) {
    constexpr int OP = pitch_of(O);
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= nobs) return;
    const int i = cam[e];
    const double *Ui = U + (size_t)i * 3 * OP, *x = xl + (size_t)lm[e] * OP;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < O; ++k) {
        const double xk = (i >= 1) ? xc[(size_t)(i - 1) * OP + k] : 0.0;
        const double d = p[3 * e] * Ui[k] + p[3 * e + 1] * Ui[OP + k] + p[3 * e + 2] * Ui[2 * OP + k] - xk + x[k];
        acc += d * d;
    }
    res[e] = acc;
}

void SchurOp::residuals(int o, const double *U, double *res_host, const CamArgs &a, hipStream_t st) {
    // the chain with the plain epilogue leaves x_cam / x_l of this U in the scratch buffers
    product(o, EPI_PLAIN, U, 1.0, a, st);
    res_.alloc((size_t)nobs_, false);
    XM_DISPATCH_O(o, hipLaunchKernelGGL((schur_obs_residual_kernel<O_>), dim3((unsigned)((nobs_ + 255) / 256)), dim3(256), 0, st, nobs_, obs_cam_.p,
                                        obs_lm_.p, obs_p_.p, U, xc_.p, xl_.p, res_.p));
    check_launch("schur_obs_residual");
    XM_HIP_CHECK(hipMemcpyAsync(res_host, res_.p, (size_t)nobs_ * sizeof(double), hipMemcpyDeviceToHost, st));
    XM_HIP_CHECK(hipStreamSynchronize(st));
}

// ybar_est = Abar @ sR_real^T of utils/recoversolution.py:77-86 without Abar (the dense (N-1+M) x 3N matrix of creatematrix.py:283-311):
// Abar = -Qtp_bar^-1 Vtp_bar^T, and the first four steps of the product chain at W = (sR_real)^T leave exactly Qtp_bar^-1 Vtp_bar^T W
// in x_cam (cameras 1..N-1) and x_l (landmarks).
void SchurOp::recover_tp(const double *rot, const double *scale, double *t, double *p, hipStream_t st) {
    if (!rot || !scale || !t || !p) throw Error(XM_ERR_ARG, "recover_tp: null argument");
    constexpr int OP = pitch_of(3);
    std::vector<double> hW((size_t)(n_ + kColPad) * 3 * OP, 0.0);
    for (int64_t i = 0; i < n_; ++i)
        for (int a = 0; a < 3; ++a)
            for (int k = 0; k < 3; ++k) hW[((size_t)i * 3 + a) * OP + k] = scale[i] * rot[(size_t)k + 3 * ((size_t)3 * i + a)];
    DevBuf<double> dW, dY;
    dW.alloc(hW.size(), false); dY.alloc((size_t)n_ * 3 * OP + 2);
    XM_HIP_CHECK(hipMemcpyAsync(dW.p, hW.data(), hW.size() * sizeof(double), hipMemcpyHostToDevice, st));
    CamArgs a;
    std::memset(&a, 0, sizeof(a));
    a.nloc = (int)n_; a.out = dY.p;
    product(3, EPI_PLAIN, dW.p, 1.0, a, st);
    std::vector<double> hx((size_t)std::max<int64_t>(n_ - 1, 0) * OP), hl((size_t)m_ * OP);
    if (!hx.empty()) XM_HIP_CHECK(hipMemcpyAsync(hx.data(), xc_.p, hx.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    XM_HIP_CHECK(hipMemcpyAsync(hl.data(), xl_.p, hl.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    XM_HIP_CHECK(hipStreamSynchronize(st));
    for (int k = 0; k < 3; ++k) t[k] = 0.0;                                   // the anchor
    for (int64_t i = 1; i < n_; ++i)
        for (int k = 0; k < 3; ++k) t[(size_t)3 * i + k] = -hx[(size_t)(i - 1) * OP + k];
    for (int64_t l = 0; l < m_; ++l)
        for (int k = 0; k < 3; ++k) p[(size_t)3 * l + k] = -hl[(size_t)l * OP + k];
}

void SchurOp::ensure(int o) {
    if (o <= o_alloc_) return;
    const size_t OP = (size_t)pitch_of(o);
    h_.alloc((size_t)m_ * OP); xl_.alloc((size_t)m_ * OP);
    r_.alloc((size_t)ldv_ * OP + 2);            // product input of the dense kernel: ldv rows, zero beyond N-1
    xc_.alloc((size_t)3 * nred_ * OP + 2);
    o_alloc_ = o;
}

int64_t SchurOp::bytes_per_product(int o) const {
    // observation arrays are streamed twice by camera (w, landmark index; w, p, landmark index) and twice by landmark, VT^{-1} once
    return nobs_ * (8 + 4) + nobs_ * (8 + 24 + 4) + nobs_ * (8 + 24 + 4) + nobs_ * (8 + 4) + 8 * (n_ - 1) * (n_ - 1) + 2LL * 8 * 3 * n_ * o;
}

template <int O>
static void schur_product_o(int epi, int64_t n, int64_t m, int64_t nheavy, const int32_t *heavy, const int64_t *cam_ptr, const int32_t *cam_lm, const double *cam_w, const double *cam_p,
                            const int64_t *lm_ptr, const int32_t *lm_cam, const double *lm_w, const double *lm_p, const double *Q1,
                            const double *c, const double *q3inv, const double *vtinv, int64_t nred, int64_t ldv, double *h, double *r,
                            double *xc, double *xl, const double *W, double alpha, const CamArgs &a, hipStream_t st) {
    const TcgScal *sc = (epi == EPI_HESS) ? a.scal : (const TcgScal *)nullptr;
    const dim3 b(256), gl((unsigned)((m + 255) / 256)), gc(qw_grid((int)n));
    const dim3 gh((unsigned)nheavy), bh(kSchurHeavyThreads);
    hipLaunchKernelGGL((schur_lm_h_kernel<O, 0>), gl, b, 0, st, m, heavy, lm_ptr, lm_cam, lm_w, lm_p, q3inv, W, sc, h);
    if (nheavy > 0) hipLaunchKernelGGL((schur_lm_h_kernel<O, 1>), gh, bh, 0, st, nheavy, heavy, lm_ptr, lm_cam, lm_w, lm_p, q3inv, W, sc, h);
    hipLaunchKernelGGL((schur_cam_r_kernel<O>), gc, b, 0, st, (int)n, cam_ptr, cam_lm, cam_w, c, W, h, sc, r);
    if (n > 1) {
        CamArgs pa;
        std::memset(&pa, 0, sizeof(pa));
        pa.nloc = (int)nred; pa.out = xc; pa.scal = a.scal;
        launch_qw_dense(O, EPI_PLAIN, vtinv, ldv, r, 1.0, pa, st);
    }
    hipLaunchKernelGGL((schur_lm_x_kernel<O, 0>), gl, b, 0, st, m, heavy, lm_ptr, lm_cam, lm_w, q3inv, h, xc, sc, xl);
    if (nheavy > 0) hipLaunchKernelGGL((schur_lm_x_kernel<O, 1>), gh, bh, 0, st, nheavy, heavy, lm_ptr, lm_cam, lm_w, q3inv, h, xc, sc, xl);
    switch (epi) {
        case EPI_PLAIN: hipLaunchKernelGGL((schur_cam_y_kernel<O, EPI_PLAIN>), gc, b, 0, st, cam_ptr, cam_lm, cam_w, cam_p, Q1, c, W, xc, xl, alpha, a); break;
        case EPI_GRAD: hipLaunchKernelGGL((schur_cam_y_kernel<O, EPI_GRAD>), gc, b, 0, st, cam_ptr, cam_lm, cam_w, cam_p, Q1, c, W, xc, xl, alpha, a); break;
        case EPI_HESS: hipLaunchKernelGGL((schur_cam_y_kernel<O, EPI_HESS>), gc, b, 0, st, cam_ptr, cam_lm, cam_w, cam_p, Q1, c, W, xc, xl, alpha, a); break;
        case EPI_CERT:
            if constexpr (O == 1) { hipLaunchKernelGGL((schur_cam_y_kernel<1, EPI_CERT>), gc, b, 0, st, cam_ptr, cam_lm, cam_w, cam_p, Q1, c, W, xc, xl, alpha, a); break; }
            throw Error(XM_ERR_ARG, "certificate operator needs o == 1");
        default: throw Error(XM_ERR_ARG, "bad epilogue");
    }
}

void SchurOp::product(int o, int epi, const double *W, double alpha, const CamArgs &a, hipStream_t st) {
    ensure(o);
    if (o != o_last_) {   // the row pitch of the scratch vectors changes with o: start from clean zero padding
        XM_HIP_CHECK(hipMemsetAsync(r_.p, 0, r_.count * sizeof(double), st));
        XM_HIP_CHECK(hipMemsetAsync(xc_.p, 0, xc_.count * sizeof(double), st));
        o_last_ = o;
    }
    XM_DISPATCH_O(o, (schur_product_o<O_>(epi, n_, m_, nheavy_, heavy_.p, cam_ptr_.p, cam_lm_.p, cam_w_.p, cam_p_.p, lm_ptr_.p, lm_cam_.p, lm_w_.p, lm_p_.p, Q1_.p,
                                         c_.p, q3inv_.p, vtinv_.p, nred_, ldv_, h_.p, r_.p, xc_.p, xl_.p, W, alpha, a, st)));
    check_launch("schur_product");
}

}  // namespace xm
