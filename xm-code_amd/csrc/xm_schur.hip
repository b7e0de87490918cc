// xm_schur.hip — matrix-free Q*W from the observation list (design: xm_schur.h).  Replaces, for the reference's own Q
// (utils/creatematrix.py:51-339), the dense product Dense/matmul.h:42-87 by the factor chain; the fused epilogues are those of
// xm_device.h, so the solver above (trust region, certificate, Lanczos) is unchanged.
#include "xm_schur.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#include "xm_device.h"

namespace xm {

// ------------------------------------------------------------------------------------------------------------------
// device kernels
// ------------------------------------------------------------------------------------------------------------------
constexpr int kSchurHeavy = 64;   // landmarks with more observations get a workgroup of their own (a thread per landmark serialises
                                  // them: 963 us per product with three landmarks seen by all 1778 cameras, 122 us once split)

// h_l = -(1/Q3_l) sum_{obs of l} w (p . W_i)      a light landmark: one thread | a heavy one: a 1024-thread workgroup (a landmark
// seen by all 13 682 cameras: 14 strided steps instead of 13 682 serial ones; thread-strided order, DPP tree per wavefront, the 16
// wavefront sums added in a fixed order)
constexpr int kSchurHeavyThreads = 1024;
// a gathered record of N doubles at 8-byte alignment with 16-byte loads (5 instead of 9 instructions for the 72-byte row block of W).
// Measured neutral at Final-13682 size: the scattered gathers are bound by cache LINES (one per lane), not by instructions.
typedef double schur_d2 __attribute__((ext_vector_type(2), aligned(8)));
template <int N>
__device__ __forceinline__ void load_rec(const double *__restrict__ p, double (&v)[N]) {
#pragma unroll
    for (int i = 0; i + 1 < N; i += 2) {
        const schur_d2 t = *reinterpret_cast<const schur_d2 *>(p + i);
        v[i] = t.x; v[i + 1] = t.y;
    }
    if (N & 1) v[N - 1] = p[N - 1];
}
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
// Landmark lists on the device (SchurLm): landmarks are numbered by degree, descending ("slot").  The first nheavy slots keep contiguous
// lists [ptr[s], ptr[s+1]) (one workgroup each); the others are packed 64 to a group in slot order -- observation k of the group's lane
// j sits at gbase[group] + 64 k + j, so the thread-per-landmark kernels read cam / w / p as coalesced 64-wide rows (a contiguous list
// per thread made every load instruction touch 64 cache lines: 180 us for h at 800 k landmarks / 6.4 M observations, 5 x its traffic).
// p is stored as three planes of `total` doubles.
struct SchurLm {
    int64_t m, nheavy, total;
    const int64_t *ptr;      // nheavy + 1
    const int64_t *gbase;    // per group of 64 light slots
    const int32_t *deg;      // per slot
    const int32_t *cam;
    const double *w, *p;
};
// ONE launch for both kinds: workgroups [0, nheavy) take a heavy landmark each (1024 threads stride its list), the others 1024 light
// landmarks each -- the few heavy workgroups (14 serial steps for a landmark seen by all 13 682 cameras) run beside the light ones
// instead of after them (32 + 14 us of the chain at Final-13682 size)
template <int O>
__global__ __launch_bounds__(kSchurHeavyThreads) void schur_lm_h_kernel(SchurLm L, const double *__restrict__ q3inv,
                                                          const double *__restrict__ W, const TcgScal *__restrict__ scal, double *__restrict__ h) {
    constexpr int OP = pitch_of(O);
    if (scal != nullptr) {
        if (scal->status != 0) return;
    }
    const bool heavy = (int64_t)blockIdx.x < L.nheavy;   // workgroup-uniform
    int64_t l, e, e_end, step;
    if (heavy) {
        l = blockIdx.x; e = L.ptr[l] + threadIdx.x; e_end = L.ptr[l + 1]; step = kSchurHeavyThreads;
    } else {
        const int64_t t = ((int64_t)blockIdx.x - L.nheavy) * kSchurHeavyThreads + threadIdx.x;
        l = L.nheavy + t;
        if (l >= L.m) return;
        e = L.gbase[t >> 6] + (t & 63); e_end = e + (int64_t)64 * L.deg[l]; step = 64;
    }
    double acc[O];
#pragma unroll
    for (int k = 0; k < O; ++k) acc[k] = 0.0;
    for (; e < e_end; e += step) {
        double Wi[3 * OP];
        load_rec<3 * OP>(W + (size_t)L.cam[e] * 3 * OP, Wi);
        const double w = L.w[e], p0 = L.p[e], p1 = L.p[L.total + e], p2 = L.p[2 * L.total + e];
#pragma unroll
        for (int k = 0; k < O; ++k) acc[k] += w * (p0 * Wi[k] + p1 * Wi[OP + k] + p2 * Wi[2 * OP + k]);
    }
    const double qi = q3inv[l];
    if (heavy) {
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
            double hl[O];
            load_rec<O>(h + (size_t)cam_lm[e] * OP, hl);
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
template <int O>
__global__ __launch_bounds__(kSchurHeavyThreads) void schur_lm_x_kernel(SchurLm L, const double *__restrict__ q3inv, const double *__restrict__ h,
                                                          const double *__restrict__ xc, const TcgScal *__restrict__ scal,
                                                          double *__restrict__ xl) {
    constexpr int OP = pitch_of(O);
    if (scal != nullptr) {
        if (scal->status != 0) return;
    }
    const bool heavy = (int64_t)blockIdx.x < L.nheavy;
    int64_t l, e, e_end, step;
    if (heavy) {
        l = blockIdx.x; e = L.ptr[l] + threadIdx.x; e_end = L.ptr[l + 1]; step = kSchurHeavyThreads;
    } else {
        const int64_t t = ((int64_t)blockIdx.x - L.nheavy) * kSchurHeavyThreads + threadIdx.x;
        l = L.nheavy + t;
        if (l >= L.m) return;
        e = L.gbase[t >> 6] + (t & 63); e_end = e + (int64_t)64 * L.deg[l]; step = 64;
    }
    double acc[O];
#pragma unroll
    for (int k = 0; k < O; ++k) acc[k] = 0.0;
    for (; e < e_end; e += step) {
        const int i = L.cam[e];
        if (i == 0) continue;
        const double w = L.w[e];
        double xi[O];
        load_rec<O>(xc + (size_t)(i - 1) * OP, xi);
#pragma unroll
        for (int k = 0; k < O; ++k) acc[k] += w * xi[k];
    }
    const double qi = q3inv[l];
    if (heavy) {
        if (!heavy_block_sum<O>(acc)) return;
    }
#pragma unroll
    for (int k = 0; k < O; ++k) xl[(size_t)l * OP + k] = h[(size_t)l * OP + k] + acc[k] * qi;
}

// Y_i = Q1_i W_i - c_i x_cam_i + sum_{obs of i} w p x_l, then the common tail of the Q*W kernels (xm_device.h)
template <int O, int EPI>
__global__ __launch_bounds__(256) void schur_cam_y_kernel(int64_t n, int64_t nobs, const int64_t *__restrict__ cam_ptr, const int32_t *__restrict__ cam_lm,
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
    const int lcam = blockIdx.x * kQwWaves + slot;   // camera of THIS rank (epilogue arrays); cam: its global index (observation lists, W, Q1, c)
    const bool active = lcam < a.nloc;
    const int64_t cam = (int64_t)a.cam0 + lcam;
    EpiOps eops;
    epi_prefetch<O, EPI>(eops, active ? lcam : 0, gl, active, a);
    double acc[3][O];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < O; ++k) acc[r][k] = 0.0;
    if (active && cam < n) {   // (cam >= n: an inert padding camera of the row partition -- zero row)
        for (int64_t e = cam_ptr[cam] + gl; e < cam_ptr[cam + 1]; e += 64) {
            const double w = cam_w[e];
            double x[O];
            load_rec<O>(xl + (size_t)cam_lm[e] * OP, x);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double wp = w * cam_p[(size_t)r * (size_t)nobs + e];   // three planes: coalesced
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
    qw_finish<O, EPI, 64, kQwWaves>(lcam, gl, slot, active, acc, alpha, a, eops, red);
}


// ------------------------------------------------------------------------------------------------------------------
// The reduced camera Laplacian WITHOUT its inverse ("a sparse Cholesky of the reduced Laplacian" -- here: no factorisation at
// all).  x_cam = VT^-1 r is solved per product by preconditioned CG on  VT = Q2_bar - V3_bar Q3^-1 V3_bar^T  applied matrix-free through the
// two observation lists (by landmark: y_l = (1/Q3_l) sum w x_cam; by camera: (VT x)_i = Q2_i x_i - sum w y_l -- the kernels of the chain
// itself), Jacobi preconditioner diag(VT), the o columns of the right-hand side advanced together with their own alpha / beta.  No
// (N-1)^2 array exists: memory and set-up are O(observations); the dense inverse costs 8 (N-1)^2 bytes (1.5 GB at 13 682 cameras, 80 GB
// at 100 k) and an O(N^3) factorisation.  Every scalar is a fixed-order sum of per-workgroup partial sums that each workgroup adds up
// itself (as cg_step_kernel does): no atomics, no grid barrier, bit-reproducible.  Launches per iteration: direction, landmark pass,
// camera pass (+ <p, VT p> partials), update (+ <r, z>, |r|^2 partials).
// ------------------------------------------------------------------------------------------------------------------
struct PcgState {             // device-resident; written by workgroup 0 of the kernel that decides
    int32_t done;             // 1: every column reached the tolerance (or the tCG launch this product belongs to is a no-op)
    int32_t iters;            // iterations performed
    double relres;            // max over the columns of |r| / |b| when it stopped
};
struct PcgArgs {
    int n1;                   // N - 1 unknowns per column
    int grid;                 // workgroups of the flat kernels == partial sums per column
    double tol2;              // (relative residual)^2 to reach
    const double *b;          // right-hand side r_ (pitch OP)
    const double *dinv;       // 1 / diag(VT), n1
    const double *q2;         // Q2 per camera (index cam)
    double *x, *r, *p, *Ap;   // pitch OP, n1 records
    double *prz[2], *prr, *pbb;   // per-workgroup partial sums of <r,z> (by iteration parity), |r|^2, |b|^2: grid * O each (column-major: [k][workgroup])
    double *ppap;             // the same of <p, VT p> from the camera pass: cam_grid * O
    int cam_grid;
    PcgState *st;
};
template <int O>
__device__ __forceinline__ void pcg_sums(const double *parts, int grid, double (&out)[O], double *sh) {
#pragma unroll
    for (int k = 0; k < O; ++k) out[k] = sum_partials256(parts + (size_t)k * grid, grid, sh);
}
template <int O>
__device__ __forceinline__ void pcg_store_partials(const double (&acc)[O], double *parts, int grid, double *sh) {
#pragma unroll
    for (int k = 0; k < O; ++k) {
        const double t = block_sum256(acc[k], sh);
        if (threadIdx.x == 0) parts[(size_t)k * grid + blockIdx.x] = t;
    }
}
// x = 0, r = b, p = z = dinv .* r; partials of <r, z> (parity 0) and |b|^2
template <int O>
__global__ __launch_bounds__(256) void pcg_init_kernel(PcgArgs a, const TcgScal *__restrict__ scal) {
    constexpr int OP = pitch_of(O);
    __shared__ double sh[4];
    if (scal != nullptr && scal->status != 0) {   // a launch enqueued past the end of the tCG: nothing to solve
        if (blockIdx.x == 0 && threadIdx.x == 0) { a.st->done = 1; a.st->iters = 0; a.st->relres = 0.0; }
        return;
    }
    double rz[O], bb[O];
#pragma unroll
    for (int k = 0; k < O; ++k) rz[k] = bb[k] = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < a.n1; i += gridDim.x * 256) {
        const double di = a.dinv[i];
#pragma unroll
        for (int k = 0; k < O; ++k) {
            const double v = a.b[(size_t)i * OP + k], z = di * v;
            a.x[(size_t)i * OP + k] = 0.0; a.r[(size_t)i * OP + k] = v; a.p[(size_t)i * OP + k] = z;
            rz[k] += v * z; bb[k] += v * v;
        }
    }
    pcg_store_partials<O>(rz, a.prz[0], a.grid, sh);
    pcg_store_partials<O>(bb, a.pbb, a.grid, sh);
    if (blockIdx.x == 0 && threadIdx.x == 0) { a.st->done = 0; a.st->iters = 0; a.st->relres = 1.0; }
}
// iteration it >= 1: beta = <r,z>_new / <r,z>_old per column, p = z + beta p; decides whether the previous update reached the tolerance
template <int O>
__global__ __launch_bounds__(256) void pcg_dir_kernel(PcgArgs a, int it) {
    constexpr int OP = pitch_of(O);
    __shared__ double sh[4];
    __shared__ int was_done;
    if (threadIdx.x == 0) was_done = a.st->done;   // ONE read per workgroup: workgroup 0 may set the flag while later workgroups start
    __syncthreads();
    if (was_done) return;
    double rzn[O], rzo[O], rr[O], bb[O];
    pcg_sums<O>(a.prz[it & 1], a.grid, rzn, sh);
    pcg_sums<O>(a.prz[(it & 1) ^ 1], a.grid, rzo, sh);
    pcg_sums<O>(a.prr, a.grid, rr, sh);
    pcg_sums<O>(a.pbb, a.grid, bb, sh);
    bool conv = true;
    double worst = 0.0;
#pragma unroll
    for (int k = 0; k < O; ++k) {
        const double q = (bb[k] > 0.0) ? rr[k] / bb[k] : 0.0;
        worst = fmax(worst, q);
        if (!(q <= a.tol2)) conv = false;
    }
    if (conv) {   // identical decision in every workgroup (identical sums); workgroup 0 publishes it for the launches behind this one.  The
                  // flag is written only AFTER every workgroup has read it above: all of them read st->done before this store can matter,
                  // because a workgroup that sees done == 1 here returns exactly as one that computes conv itself
        if (blockIdx.x == 0 && threadIdx.x == 0) { a.st->done = 1; a.st->iters = it; a.st->relres = sqrt(worst); }
        return;
    }
    double beta[O];
#pragma unroll
    for (int k = 0; k < O; ++k) beta[k] = (rzo[k] > 0.0) ? rzn[k] / rzo[k] : 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < a.n1; i += gridDim.x * 256) {
        const double di = a.dinv[i];
#pragma unroll
        for (int k = 0; k < O; ++k) a.p[(size_t)i * OP + k] = di * a.r[(size_t)i * OP + k] + beta[k] * a.p[(size_t)i * OP + k];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { a.st->iters = it; a.st->relres = sqrt(worst); }
}
// y_l = (1/Q3_l) sum_{obs of l} w p_cam   (the anchor camera 0 is not an unknown)
template <int O>
__global__ __launch_bounds__(kSchurHeavyThreads) void pcg_lm_kernel(SchurLm L, const double *__restrict__ q3inv, const double *__restrict__ pv,
                                                                  const PcgState *__restrict__ st, double *__restrict__ y) {
    constexpr int OP = pitch_of(O);
    if (st->done) return;
    const bool heavy = (int64_t)blockIdx.x < L.nheavy;
    int64_t l, e, e_end, step;
    if (heavy) {
        l = blockIdx.x; e = L.ptr[l] + threadIdx.x; e_end = L.ptr[l + 1]; step = kSchurHeavyThreads;
    } else {
        const int64_t t = ((int64_t)blockIdx.x - L.nheavy) * kSchurHeavyThreads + threadIdx.x;
        l = L.nheavy + t;
        if (l >= L.m) return;
        e = L.gbase[t >> 6] + (t & 63); e_end = e + (int64_t)64 * L.deg[l]; step = 64;
    }
    double acc[O];
#pragma unroll
    for (int k = 0; k < O; ++k) acc[k] = 0.0;
    for (; e < e_end; e += step) {
        const int i = L.cam[e];
        if (i == 0) continue;
        const double w = L.w[e];
        double xi[O];
        load_rec<O>(pv + (size_t)(i - 1) * OP, xi);
#pragma unroll
        for (int k = 0; k < O; ++k) acc[k] += w * xi[k];
    }
    const double qi = q3inv[l];
    if (heavy) {
        if (!heavy_block_sum<O>(acc)) return;
    }
#pragma unroll
    for (int k = 0; k < O; ++k) y[(size_t)l * OP + k] = acc[k] * qi;
}
// Ap_i = Q2_i p_i - sum_{obs of i} w y_l, one wavefront per camera 1..N-1; per-workgroup partials of <p, Ap>
template <int O>
__global__ __launch_bounds__(256) void pcg_cam_kernel(int n, const int64_t *__restrict__ cam_ptr, const int32_t *__restrict__ cam_lm,
                                                       const double *__restrict__ cam_w, const double *__restrict__ y, PcgArgs a) {
    constexpr int OP = pitch_of(O);
    __shared__ double red[kQwWaves][O];
    if (a.st->done) return;
    const int gl = threadIdx.x & 63, wv = threadIdx.x >> 6, cam = blockIdx.x * kQwWaves + wv;
    double acc[O];
#pragma unroll
    for (int k = 0; k < O; ++k) acc[k] = 0.0;
    const bool on = cam < n && cam >= 1;
    if (on)
        for (int64_t e = cam_ptr[cam] + gl; e < cam_ptr[cam + 1]; e += 64) {
            const double w = cam_w[e];
            double yl[O];
            load_rec<O>(y + (size_t)cam_lm[e] * OP, yl);
#pragma unroll
            for (int k = 0; k < O; ++k) acc[k] += w * yl[k];
        }
#pragma unroll
    for (int k = 0; k < O; ++k) acc[k] = wave_sum(acc[k]);
    if (gl == 0) {
#pragma unroll
        for (int k = 0; k < O; ++k) {
            double pap = 0.0;
            if (on) {
                const double pk = a.p[(size_t)(cam - 1) * OP + k];
                const double v = a.q2[cam] * pk - acc[k];
                a.Ap[(size_t)(cam - 1) * OP + k] = v;
                pap = pk * v;
            }
            red[wv][k] = pap;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < O; ++k) {
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < kQwWaves; ++q) t += red[q][k];
            a.ppap[(size_t)k * gridDim.x + blockIdx.x] = t;
        }
    }
}
// alpha = <r,z> / <p,Ap> per column; x += alpha p, r -= alpha Ap; partials of the new <r,z> (other parity) and |r|^2
template <int O>
__global__ __launch_bounds__(256) void pcg_upd_kernel(PcgArgs a, int it) {
    constexpr int OP = pitch_of(O);
    __shared__ double sh[4];
    if (a.st->done) return;
    double rz[O], pap[O], alpha[O];
    pcg_sums<O>(a.prz[it & 1], a.grid, rz, sh);
    pcg_sums<O>(a.ppap, a.cam_grid, pap, sh);
#pragma unroll
    for (int k = 0; k < O; ++k) alpha[k] = (pap[k] > 0.0 && rz[k] > 0.0) ? rz[k] / pap[k] : 0.0;
    double rzn[O], rr[O];
#pragma unroll
    for (int k = 0; k < O; ++k) rzn[k] = rr[k] = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < a.n1; i += gridDim.x * 256) {
        const double di = a.dinv[i];
#pragma unroll
        for (int k = 0; k < O; ++k) {
            const size_t j = (size_t)i * OP + k;
            a.x[j] += alpha[k] * a.p[j];
            const double rn = a.r[j] - alpha[k] * a.Ap[j];
            a.r[j] = rn;
            rzn[k] += rn * rn * di; rr[k] += rn * rn;
        }
    }
    pcg_store_partials<O>(rzn, a.prz[(it & 1) ^ 1], a.grid, sh);
    pcg_store_partials<O>(rr, a.prr, a.grid, sh);
}
// 1 / diag(VT):  diag_i = Q2_i - sum_{obs of i} w^2 / Q3_l   (one wavefront per camera; set-up)
__global__ __launch_bounds__(256) void pcg_diag_kernel(int n, const int64_t *__restrict__ cam_ptr, const int32_t *__restrict__ cam_lm,
                                                        const double *__restrict__ cam_w, const double *__restrict__ q3inv,
                                                        const double *__restrict__ q2, double *__restrict__ dinv) {
    const int gl = threadIdx.x & 63, cam = blockIdx.x * kQwWaves + (threadIdx.x >> 6);
    double acc = 0.0;
    if (cam < n && cam >= 1)
        for (int64_t e = cam_ptr[cam] + gl; e < cam_ptr[cam + 1]; e += 64) { const double w = cam_w[e]; acc += w * w * q3inv[cam_lm[e]]; }
    acc = wave_sum(acc);
    if (cam < n && cam >= 1 && gl == 0) {
        const double d = q2[cam] - acc;
        dinv[cam - 1] = (d > 0.0) ? 1.0 / d : 0.0;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// device: everything that depends on the weights (round 4; the host version below stays as the fall-back for observation lists that name
// a (camera, landmark) pair twice).  utils/creatematrix.py:62-175 restated on the observation level.
// ------------------------------------------------------------------------------------------------------------------
// weights in input order -> the by-camera and by-landmark arrays
__global__ __launch_bounds__(256) void schur_scatter_w_kernel(int64_t nobs, const double *__restrict__ w, const int64_t *__restrict__ pos_c,
                                                               const int64_t *__restrict__ dpos_l, double *__restrict__ cam_w, double *__restrict__ lm_w) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= nobs) return;
    const double x = w[e];
    cam_w[pos_c[e]] = x;
    lm_w[dpos_l[e]] = x;
}
// per camera: Q1 = sum w p p^T (creatematrix.py:26), c = V1 = sum w p (:27), Q2 = sum w (:68); one wavefront per camera, lane-strided
// list order + DPP tree (fixed order)
__global__ __launch_bounds__(256) void schur_cam_sums_kernel(int64_t n, int64_t nobs, const int64_t *__restrict__ cam_ptr, const double *__restrict__ cam_w,
                                                              const double *__restrict__ cam_p, double *__restrict__ Q1, double *__restrict__ c,
                                                              double *__restrict__ q2) {
    const int gl = threadIdx.x & 63;
    const int64_t cam = (int64_t)blockIdx.x * kQwWaves + (threadIdx.x >> 6);
    double a[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) a[k] = 0.0;
    if (cam < n)
        for (int64_t e = cam_ptr[cam] + gl; e < cam_ptr[cam + 1]; e += 64) {
            const double w = cam_w[e], p0 = cam_p[e], p1 = cam_p[nobs + e], p2 = cam_p[2 * nobs + e];
            const double pp[3] = {p0, p1, p2};
#pragma unroll
            for (int x = 0; x < 3; ++x) {
                a[9 + x] += w * pp[x];
#pragma unroll
                for (int y = 0; y < 3; ++y) a[3 * x + y] += w * pp[x] * pp[y];
            }
            a[12] += w;
        }
#pragma unroll
    for (int k = 0; k < 13; ++k) a[k] = wave_sum(a[k]);
    if (cam < n && gl == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) Q1[cam * 9 + k] = a[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) c[cam * 3 + k] = a[9 + k];
        q2[cam] = a[12];
    }
}
// per landmark slot: 1 / Q3, Q3 = sum w over its observations (:69); a landmark without weight drops out (0)
__global__ __launch_bounds__(kSchurHeavyThreads) void schur_lm_q3_kernel(SchurLm L, double *__restrict__ q3inv) {
    const bool heavy = (int64_t)blockIdx.x < L.nheavy;
    int64_t l, e, e_end, step;
    if (heavy) {
        l = blockIdx.x; e = L.ptr[l] + threadIdx.x; e_end = L.ptr[l + 1]; step = kSchurHeavyThreads;
    } else {
        const int64_t t = ((int64_t)blockIdx.x - L.nheavy) * kSchurHeavyThreads + threadIdx.x;
        l = L.nheavy + t;
        if (l >= L.m) return;
        e = L.gbase[t >> 6] + (t & 63); e_end = e + (int64_t)64 * L.deg[l]; step = 64;
    }
    double acc[1] = {0.0};
    for (; e < e_end; e += step) acc[0] += L.w[e];
    if (heavy) {
        if (!heavy_block_sum<1>(acc)) return;
    }
    q3inv[l] = (acc[0] > 0.0) ? 1.0 / acc[0] : 0.0;
}
// the reduced camera Laplacian VT = Q2_bar - V3_bar Q3^-1 V3_bar^T (:150-166) row by row: ONE wavefront per camera a, the row (columns
// [col0, col0 + ncol)) in LDS; the camera's observations are walked in list order and for each the landmark's observation list is spread
// over the lanes (a landmark names a camera once: no two lanes meet on one entry; the additions into an entry happen in the camera's
// observation order -- the order of the host assembly this replaces).  Landmarks with more than kSchurHeavy observations are full rank-1
// terms and are applied afterwards (rank1_sub_device).
__global__ __launch_bounds__(64) void schur_vt_rows_kernel(int64_t mr, int64_t col0, int ncol, const int64_t *__restrict__ cam_ptr,
                                                            const int32_t *__restrict__ cam_lm, const double *__restrict__ cam_w, SchurLm L,
                                                            const double *__restrict__ q3inv, const double *__restrict__ q2, double *__restrict__ VT) {
    extern __shared__ double row[];
    const int lane = threadIdx.x;
    const int64_t a = (int64_t)blockIdx.x + 1;
    for (int j = lane; j < ncol; j += 64) row[j] = 0.0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0 && a - 1 >= col0 && a - 1 < col0 + ncol) row[a - 1 - col0] = q2[a];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int64_t e = cam_ptr[a]; e < cam_ptr[a + 1]; ++e) {   // wave-uniform
        const int64_t sl = cam_lm[e];
        const double wa = cam_w[e];
        if (sl < L.nheavy || wa == 0.0) continue;
        const double qi = q3inv[sl];
        if (qi == 0.0) continue;
        const int64_t t = sl - L.nheavy;
        if (lane < L.deg[sl]) {
            const int64_t at = L.gbase[t >> 6] + (t & 63) + (int64_t)64 * lane;
            const int64_t b = L.cam[at];
            const int64_t j = b - 1 - col0;
            if (b != 0 && j >= 0 && j < ncol) row[j] -= wa * L.w[at] * qi;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    double *out = VT + (size_t)(a - 1) * (size_t)mr + (size_t)col0;   // row a-1 of the symmetric matrix = its column a-1 (column-major)
    for (int j = lane; j < ncol; j += 64) out[j] = row[j];
}

// ------------------------------------------------------------------------------------------------------------------
// host: factors from the observation list (utils/creatematrix.py:62-175, restated on the observation level)
// ------------------------------------------------------------------------------------------------------------------
SchurOp::SchurOp(int64_t n, int64_t n_landmarks, int64_t nobs, const int32_t *cam, const int32_t *lm, const double *p, const double *w,
                 hipStream_t st, Comm *comm, const SchurSettings &cfg) {
    cfg_ = cfg;
    comm_ = (comm && comm->active()) ? comm : nullptr;
    world_ = comm_ ? comm_->world : 1;
    rank_ = comm_ ? comm_->rank : 0;
    if (n < 1 || n_landmarks < 1 || nobs < 1 || !cam || !lm || !p || !w) throw Error(XM_ERR_ARG, "matrix-free Q: bad observation list");
    // how VT^-1 is applied: the dense inverse (fast products, 8 (N-1)^2 bytes and an O(N^3) set-up) up to cfg.dense_max cameras, above that
    // preconditioned CG inside every product (no (N-1)^2 array at all); cfg.solver forces one of the two
    pcg_ = cfg.solver == 2 || (cfg.solver == 0 && n > cfg.dense_max);
    if (pcg_ && comm_) throw Error(XM_ERR_ARG, "matrix-free Q: the CG form of the reduced camera system runs on one GPU (row-partitioned contexts use the dense inverse)");
    if (!pcg_ && n > kSchurMaxCams) throw Error(XM_ERR_ARG, "matrix-free Q: more than " + std::to_string(kSchurMaxCams) + " cameras need the CG form (xm_tuning_t.schur_solver)");
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
    // device numbering of the landmarks ("slot"): by degree, descending, stable
    std::vector<int32_t> order((size_t)M);
    for (int64_t l = 0; l < M; ++l) order[(size_t)l] = (int32_t)l;
    auto deg_of = [&](int64_t l) { return lp_[(size_t)l + 1] - lp_[(size_t)l]; };
    // ... and, among equals, by the first camera that sees them (smallest index): the landmarks of neighbouring cameras get neighbouring
    // numbers, so that a camera's gathers of landmark records (schur_cam_*, pcg_cam_kernel: one 128-byte line per 24 .. 40-byte record when
    // the numbers are scattered) fall into few lines wherever the scene has locality (sequential capture; a scene whose landmarks are seen by
    // cameras drawn at random has none to find)
    std::vector<int32_t> first_cam((size_t)M, INT32_MAX);
    for (int64_t e = 0; e < nobs; ++e) first_cam[(size_t)lm[e]] = std::min(first_cam[(size_t)lm[e]], cam[e]);
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) {
        const int64_t dx = deg_of(x), dy = deg_of(y);
        return dx != dy ? dx > dy : first_cam[(size_t)x] < first_cam[(size_t)y];
    });
    slot_of_.assign((size_t)M, 0);
    std::vector<int32_t> ldeg((size_t)M);
    nheavy_ = 0;
    for (int64_t sl = 0; sl < M; ++sl) {
        slot_of_[(size_t)order[(size_t)sl]] = (int32_t)sl;
        ldeg[(size_t)sl] = (int32_t)deg_of(order[(size_t)sl]);
        if (ldeg[(size_t)sl] > kSchurHeavy) nheavy_ = sl + 1;
    }
    std::vector<int64_t> hptr((size_t)nheavy_ + 1, 0);
    for (int64_t sl = 0; sl < nheavy_; ++sl) hptr[(size_t)sl + 1] = hptr[(size_t)sl] + ldeg[(size_t)sl];
    int64_t total = hptr[(size_t)nheavy_];
    const int64_t ngroups = (M - nheavy_ + 63) / 64;
    std::vector<int64_t> gbase((size_t)std::max<int64_t>(ngroups, 1), 0);
    for (int64_t g = 0; g < ngroups; ++g) {
        gbase[(size_t)g] = total;
        total += (int64_t)64 * ldeg[(size_t)(nheavy_ + 64 * g)];   // the group's first slot has its longest list
    }
    ltotal_ = std::max<int64_t>(total, 1);
    std::vector<int32_t> c_lm((size_t)nobs), d_lcam((size_t)ltotal_, 0), o_lm((size_t)nobs);
    std::vector<double> c_p((size_t)nobs * 3), d_lp((size_t)ltotal_ * 3, 0.0);
    lcam_.assign((size_t)nobs, 0);
    pos_c_.assign((size_t)nobs, 0); pos_l_.assign((size_t)nobs, 0); dpos_l_.assign((size_t)nobs, 0); cam_obs_.assign((size_t)nobs, 0);
    {
        // position of every observation in its camera's list: ascending landmark number (ties: input order), so that the lanes of a wavefront
        // that walk a camera's list gather neighbouring landmark records
        std::vector<int64_t> apos((size_t)nobs);
        {
            std::vector<int64_t> nc0(cp_.begin(), cp_.end() - 1), byc((size_t)nobs);
            for (int64_t e = 0; e < nobs; ++e) byc[(size_t)nc0[(size_t)cam[e]]++] = e;
            for (int64_t i = 0; i < N; ++i)
                std::stable_sort(byc.begin() + cp_[(size_t)i], byc.begin() + cp_[(size_t)i + 1],
                                 [&](int64_t x, int64_t y) { return slot_of_[(size_t)lm[x]] < slot_of_[(size_t)lm[y]]; });
            for (int64_t q = 0; q < nobs; ++q) apos[(size_t)byc[(size_t)q]] = q;
        }
        std::vector<int64_t> nl(lp_.begin(), lp_.end() - 1);
        for (int64_t e = 0; e < nobs; ++e) {
            const int64_t a2 = apos[(size_t)e], b2 = nl[(size_t)lm[e]]++;
            const int64_t sl = slot_of_[(size_t)lm[e]], k = b2 - lp_[(size_t)lm[e]];
            const int64_t d2 = (sl < nheavy_) ? hptr[(size_t)sl] + k : gbase[(size_t)((sl - nheavy_) >> 6)] + 64 * k + ((sl - nheavy_) & 63);
            pos_c_[(size_t)e] = a2; pos_l_[(size_t)e] = b2; dpos_l_[(size_t)e] = d2;
            cam_obs_[(size_t)a2] = e;
            c_lm[(size_t)a2] = (int32_t)sl; o_lm[(size_t)e] = (int32_t)sl;
            for (int a = 0; a < 3; ++a) c_p[(size_t)a * (size_t)nobs + (size_t)a2] = p[3 * e + a];   // three planes
            lcam_[(size_t)b2] = cam[e];                           // host copy (by landmark, contiguous): assembly of VT
            d_lcam[(size_t)d2] = cam[e];
            for (int a = 0; a < 3; ++a) d_lp[(size_t)a * (size_t)ltotal_ + (size_t)d2] = p[3 * e + a];
        }
    }
    auto up = [&](auto &buf, const auto &v) {
        buf.alloc(std::max<size_t>(v.size(), 1), false);
        if (!v.empty()) XM_HIP_CHECK(hipMemcpy(buf.p, v.data(), v.size() * sizeof(v[0]), hipMemcpyHostToDevice));
    };
    up(cam_ptr_, cp_); up(lm_ptr_, hptr); up(gbase_, gbase); up(ldeg_, ldeg); up(cam_lm_, c_lm); up(lm_cam_, d_lcam); up(cam_p_, c_p); up(lm_p_, d_lp);
    up(obs_cam_, hcam_); up(obs_lm_, o_lm); up(obs_p_, hp_);
    up(pos_c_dev_, pos_c_); up(dpos_l_dev_, dpos_l_);
    w_in_.alloc((size_t)nobs, false); q2_.alloc((size_t)N, false);
    {   // a (camera, landmark) pair named twice would put two lanes of the device assembly on one entry of VT: such a list keeps the host path
        std::vector<int64_t> seen((size_t)N, -1);
        dup_pairs_ = false;
        for (int64_t l = 0; l < M && !dup_pairs_; ++l)
            for (int64_t e2 = lp_[(size_t)l]; e2 < lp_[(size_t)l + 1]; ++e2) {
                const int64_t b = lcam_[(size_t)e2];
                if (seen[(size_t)b] == l) { dup_pairs_ = true; break; }
                seen[(size_t)b] = l;
            }
        if (cfg_.host_assembly) dup_pairs_ = true;
    }
    hub_lm_.clear(); hub_obs_ptr_.assign(1, 0); hub_obs_.clear();   // heavy landmarks: their observations (input indices), for the rank-1 terms
    {
        std::vector<std::vector<int64_t>> by((size_t)nheavy_);
        for (int64_t e = 0; e < nobs; ++e) {
            const int64_t sl = slot_of_[(size_t)lm[e]];
            if (sl < nheavy_) by[(size_t)sl].push_back(e);
        }
        for (int64_t sl = 0; sl < nheavy_; ++sl) {
            hub_lm_.push_back(sl);
            hub_obs_.insert(hub_obs_.end(), by[(size_t)sl].begin(), by[(size_t)sl].end());
            hub_obs_ptr_.push_back((int64_t)hub_obs_.size());
        }
    }
    cam_w_.alloc((size_t)nobs, false); lm_w_.alloc((size_t)ltotal_, false);
    Q1_.alloc((size_t)N * 9, false); c_.alloc((size_t)N * 3, false); q3inv_.alloc((size_t)M, false);
    const int64_t mr = N - 1;
    nred_ = std::max<int64_t>(1, (mr + 2) / 3);
    ldv_ = dense_ld(nred_);
    // several ranks: the rows of VT^-1 are dealt out in equal ranges of pseudo-cameras (3 rows each); every rank keeps the whole inverse
    // (the set-up is replicated) but multiplies only its own rows, and the ranks all-gather x_cam
    nred_loc_ = (nred_ + world_ - 1) / world_;
    nred_pad_ = nred_loc_ * world_;
    if (pcg_) {
        dup_pairs_ = false;   // (the host assembly exists for VT; a pair named twice only makes the Jacobi diagonal approximate)
        if (cfg_.pcg_first > 0) pcg_last_iters_[0] = pcg_last_iters_[1] = std::max(1, cfg_.pcg_first - 2);
        if (cfg_.pcg_hess_digits > 0) pcg_tol_[1] = std::pow(10.0, -(double)std::min(13, std::max(6, cfg_.pcg_hess_digits)));
        pcg_dinv_.alloc((size_t)std::max<int64_t>(mr, 1));
        XM_HIP_CHECK(hipHostMalloc((void **)&pcg_host_, sizeof(PcgState), hipHostMallocDefault));
        pcg_state_.alloc(sizeof(PcgState) / sizeof(int32_t) + 2);
    } else {
        vtinv_.alloc((size_t)3 * nred_pad_ * (size_t)ldv_);
    }
    set_weights(w, st);
}

SchurOp::~SchurOp() {
    if (pcg_host_) (void)hipHostFree(pcg_host_);
}

// Everything that depends on the weights (utils/creatematrix.py:62-175 restated on the observation level): Q1, c (= V1), Q3, the
// reduced camera Laplacian VT = Q2_bar - V3_bar Q3^{-1} V3_bar^T and its inverse.  Called by the constructor and by the XM^2 loop
// (observations filtered by weight 0).  A landmark whose observations all have weight 0 drops out (1/Q3 := 0); a camera without
// weight makes VT singular -> XM_ERR_ARG.
void SchurOp::set_weights(const double *w, hipStream_t st) {
    if (!w) throw Error(XM_ERR_ARG, "matrix-free Q: null weights");
    if (!dup_pairs_) { set_weights_device(w, st); return; }
    const int64_t N = n_, M = m_, nobs = nobs_;
    const bool trace = cfg_.trace;   // set-up phase times on stderr
    auto tp = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "schur set-up: %-28s %8.1f ms\n", what, std::chrono::duration<double>(now - tp).count() * 1e3);
        tp = now;
    };
    std::vector<double> Q1((size_t)N * 9, 0.0), c((size_t)N * 3, 0.0), Q2((size_t)N, 0.0), Q3((size_t)M, 0.0);
    std::vector<double> c_w((size_t)nobs), l_w((size_t)nobs), d_lw((size_t)ltotal_, 0.0);
    for (int64_t e = 0; e < nobs; ++e) {
        if (!(w[e] >= 0.0)) throw Error(XM_ERR_ARG, "matrix-free Q: negative or NaN weight");
        const int64_t i = hcam_[(size_t)e], l = hlm_[(size_t)e];
        const double *pe = &hp_[(size_t)e * 3];
        for (int a = 0; a < 3; ++a) {
            c[(size_t)i * 3 + a] += w[e] * pe[a];                                                 // V1 block (creatematrix.py:27)
            for (int b = 0; b < 3; ++b) Q1[(size_t)i * 9 + 3 * a + b] += w[e] * pe[a] * pe[b];    // Q1 block (:26)
        }
        Q2[(size_t)i] += w[e]; Q3[(size_t)l] += w[e];                                             // :68-69
        c_w[(size_t)pos_c_[(size_t)e]] = w[e]; l_w[(size_t)pos_l_[(size_t)e]] = w[e]; d_lw[(size_t)dpos_l_[(size_t)e]] = w[e];
    }
    lap("Q1, c, Q2, Q3 (host)");
    std::vector<double> q3inv((size_t)M), q3inv_slot((size_t)M);   // by landmark (host: VT) and by slot (device)
    for (int64_t l = 0; l < M; ++l) {
        q3inv[(size_t)l] = (Q3[(size_t)l] > 0.0) ? 1.0 / Q3[(size_t)l] : 0.0;
        q3inv_slot[(size_t)slot_of_[(size_t)l]] = q3inv[(size_t)l];
    }
    const int64_t mr = N - 1;
    std::vector<double> VT((size_t)std::max<int64_t>(mr, 1) * (size_t)std::max<int64_t>(mr, 1), 0.0);
    for (int64_t i = 1; i < N; ++i) VT[(size_t)(i - 1) + (size_t)(i - 1) * mr] = Q2[(size_t)i];
    // VT is assembled ROW BY ROW (camera a: its observations in list order, for each the cameras of that landmark): a row of N-1
    // doubles stays in the cache while its ~deg(a) * deg(l) updates land, rows are independent (threads own row ranges; the order of
    // the additions into an entry is the camera's observation order whatever the thread count), and the matrix is symmetric, so the
    // row-major image is the column-major one.  By landmark instead, the same 51 M updates at 13 682 cameras scatter over 1.5 GB: 1.7 s
    // against 0.1 s.  Landmarks with more than kSchurHeavy observations are full rank-1 terms and are applied on the device below.
    std::vector<int64_t> hubs;
    for (int64_t l = 0; l < M; ++l)
        if (q3inv[(size_t)l] != 0.0 && lp_[(size_t)l + 1] - lp_[(size_t)l] > kSchurHeavy && mr > 0) hubs.push_back(l);
    if (mr > 0) {
        auto rows = [&](int64_t a0, int64_t a1) {
            for (int64_t a = std::max<int64_t>(a0, 1); a < a1; ++a) {
                double *row = VT.data() + (size_t)(a - 1) * (size_t)mr;
                for (int64_t pc = cp_[(size_t)a]; pc < cp_[(size_t)a + 1]; ++pc) {
                    const int64_t e = cam_obs_[(size_t)pc], l = hlm_[(size_t)e];
                    const double qi = q3inv[(size_t)l], wa = w[e];
                    if (qi == 0.0 || wa == 0.0 || lp_[(size_t)l + 1] - lp_[(size_t)l] > kSchurHeavy) continue;
                    for (int64_t e2 = lp_[(size_t)l]; e2 < lp_[(size_t)l + 1]; ++e2) {
                        const int64_t b = lcam_[(size_t)e2];
                        if (b != 0) row[b - 1] -= wa * l_w[(size_t)e2] * qi;
                    }
                }
            }
        };
        const int nthr = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(16, (int64_t)std::thread::hardware_concurrency()), N / 256));
        if (nthr <= 1) rows(1, N);
        else {
            std::vector<std::thread> pool;
            for (int t = 0; t < nthr; ++t) pool.emplace_back(rows, N * t / nthr, N * (t + 1) / nthr);
            for (auto &th : pool) th.join();
        }
    }
    lap("VT rows (host threads)");
    auto put = [&](DevBuf<double> &buf, const std::vector<double> &v) {
        if (!v.empty()) XM_HIP_CHECK(hipMemcpy(buf.p, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice));
    };
    put(cam_w_, c_w); put(lm_w_, d_lw); put(Q1_, Q1); put(c_, c); put(q3inv_, q3inv_slot);
    if (mr > 0) {   // invert on the device (blocked Cholesky, xm_dense_la.hip), then lay the inverse out like a dense Q
        DevBuf<double> tmp, inv;
        tmp.alloc((size_t)mr * mr, false); inv.alloc((size_t)mr * mr, false);
        XM_HIP_CHECK(hipMemcpy(tmp.p, VT.data(), (size_t)mr * mr * sizeof(double), hipMemcpyHostToDevice));
        std::vector<double>().swap(VT);
        lap("VT to the device");
        if (!hubs.empty()) {
            DevBuf<double> du;
            du.alloc((size_t)mr, false);
            std::vector<double> u((size_t)mr);
            for (int64_t l : hubs) {   // in landmark order (fixed): VT -= (1/Q3_l) u u^T, u = the weights of l's observations by camera
                std::fill(u.begin(), u.end(), 0.0);
                for (int64_t e1 = lp_[(size_t)l]; e1 < lp_[(size_t)l + 1]; ++e1)
                    if (lcam_[(size_t)e1] != 0) u[(size_t)lcam_[(size_t)e1] - 1] += l_w[(size_t)e1];
                XM_HIP_CHECK(hipMemcpyAsync(du.p, u.data(), (size_t)mr * sizeof(double), hipMemcpyHostToDevice, st));
                rank1_sub_device((int)mr, tmp.p, du.p, q3inv[(size_t)l], st);
                XM_HIP_CHECK(hipStreamSynchronize(st));   // u is reused
            }
        }
        XM_HIP_CHECK(hipStreamSynchronize(st));
        lap("hub rank-1 terms (device)");
        if (!spd_inverse_device((int)mr, tmp.p, inv.p, st, cfg_.trace))
            throw Error(XM_ERR_ARG, "matrix-free Q: the reduced camera Laplacian is not positive definite (observation graph not connected?)");
        lap("SPD inverse (device)");
        spd_inverse_layout((int)mr, inv.p, vtinv_.p, ldv_, st);
        XM_HIP_CHECK(hipStreamSynchronize(st));
        lap("layout of the inverse");
    }
}

// The same on the device: weights uploaded once, per-camera / per-landmark sums, the reduced camera Laplacian row by row in LDS (the host
// needed 0.2 + 0.3 s for them at 13 682 cameras and 0.18 s to ship the 1.5 GB matrix), hub landmarks as rank-1 terms, inverse, layout.
void SchurOp::set_weights_device(const double *w, hipStream_t st) {
    const int64_t N = n_, M = m_, nobs = nobs_;
    const bool trace = cfg_.trace;
    auto tp = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        XM_HIP_CHECK(hipStreamSynchronize(st));
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "schur set-up: %-28s %8.1f ms\n", what, std::chrono::duration<double>(now - tp).count() * 1e3);
        tp = now;
    };
    for (int64_t e = 0; e < nobs; ++e)
        if (!(w[e] >= 0.0)) throw Error(XM_ERR_ARG, "matrix-free Q: negative or NaN weight");
    XM_HIP_CHECK(hipMemcpyAsync(w_in_.p, w, (size_t)nobs * sizeof(double), hipMemcpyHostToDevice, st));
    XM_HIP_CHECK(hipMemsetAsync(lm_w_.p, 0, (size_t)ltotal_ * sizeof(double), st));   // the padding entries of the packed landmark lists
    hipLaunchKernelGGL(schur_scatter_w_kernel, dim3((unsigned)((nobs + 255) / 256)), dim3(256), 0, st, nobs, w_in_.p, pos_c_dev_.p, dpos_l_dev_.p,
                       cam_w_.p, lm_w_.p);
    hipLaunchKernelGGL(schur_cam_sums_kernel, dim3((unsigned)((N + kQwWaves - 1) / kQwWaves)), dim3(256), 0, st, N, nobs, cam_ptr_.p, cam_w_.p, cam_p_.p,
                       Q1_.p, c_.p, q2_.p);
    SchurLm L;
    L.m = m_; L.nheavy = nheavy_; L.total = ltotal_; L.ptr = lm_ptr_.p; L.gbase = gbase_.p; L.deg = ldeg_.p; L.cam = lm_cam_.p; L.w = lm_w_.p; L.p = lm_p_.p;
    const int64_t nlight = M - nheavy_;
    hipLaunchKernelGGL(schur_lm_q3_kernel, dim3((unsigned)(nheavy_ + (nlight + kSchurHeavyThreads - 1) / kSchurHeavyThreads)), dim3(kSchurHeavyThreads), 0, st, L,
                       q3inv_.p);
    check_launch("schur set-up (sums)");
    lap("weights, Q1, c, Q2, 1/Q3 (device)");
    const int64_t mr = N - 1;
    if (mr <= 0) { XM_HIP_CHECK(hipStreamSynchronize(st)); return; }
    if (pcg_) {   // no VT, no inverse: the Jacobi diagonal is all the CG form needs
        hipLaunchKernelGGL(pcg_diag_kernel, dim3((unsigned)((N + kQwWaves - 1) / kQwWaves)), dim3(256), 0, st, (int)N, cam_ptr_.p, cam_lm_.p, cam_w_.p,
                           q3inv_.p, q2_.p, pcg_dinv_.p);
        check_launch("schur set-up (Jacobi diagonal)");
        std::vector<double> hd((size_t)mr);
        XM_HIP_CHECK(hipMemcpyAsync(hd.data(), pcg_dinv_.p, (size_t)mr * sizeof(double), hipMemcpyDeviceToHost, st));
        XM_HIP_CHECK(hipStreamSynchronize(st));
        for (double d : hd)
            if (!(d > 0.0) || !std::isfinite(d)) throw Error(XM_ERR_ARG, "matrix-free Q: a camera has no weight on the reduced camera Laplacian's diagonal (camera without observations?)");
        lap("Jacobi diagonal (device)");
        return;
    }
    DevBuf<double> tmp, inv;
    tmp.alloc((size_t)mr * mr, false); inv.alloc((size_t)mr * mr, false);
    constexpr int64_t kColsPerPass = 16384;   // 128 KB of LDS per wavefront (one workgroup per CU; gfx950 has 160 KB)
    {
        static const bool once = [] {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(schur_vt_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     (int)(kColsPerPass * sizeof(double)));
            (void)hipGetLastError();
            return e == hipSuccess;
        }();
        (void)once;
    }
    for (int64_t col0 = 0; col0 < mr; col0 += kColsPerPass) {
        const int ncol = (int)std::min<int64_t>(kColsPerPass, mr - col0);
        hipLaunchKernelGGL(schur_vt_rows_kernel, dim3((unsigned)mr), dim3(64), (size_t)ncol * sizeof(double), st, mr, col0, ncol, cam_ptr_.p, cam_lm_.p,
                           cam_w_.p, L, q3inv_.p, q2_.p, tmp.p);
    }
    check_launch("schur set-up (VT rows)");
    lap("VT rows (device)");
    if (nheavy_ > 0) {   // hub landmarks in slot order (fixed): VT -= (1/Q3_l) u u^T, u = the weights of l's observations by camera
        std::vector<double> q3h((size_t)nheavy_);
        XM_HIP_CHECK(hipMemcpyAsync(q3h.data(), q3inv_.p, (size_t)nheavy_ * sizeof(double), hipMemcpyDeviceToHost, st));
        XM_HIP_CHECK(hipStreamSynchronize(st));
        DevBuf<double> du;
        du.alloc((size_t)mr, false);
        std::vector<double> u((size_t)mr);
        for (int64_t h = 0; h < nheavy_; ++h) {
            if (q3h[(size_t)h] == 0.0) continue;
            std::fill(u.begin(), u.end(), 0.0);
            for (int64_t k = hub_obs_ptr_[(size_t)h]; k < hub_obs_ptr_[(size_t)h + 1]; ++k) {
                const int64_t e = hub_obs_[(size_t)k];
                if (hcam_[(size_t)e] != 0) u[(size_t)hcam_[(size_t)e] - 1] += w[e];
            }
            XM_HIP_CHECK(hipMemcpyAsync(du.p, u.data(), (size_t)mr * sizeof(double), hipMemcpyHostToDevice, st));
            rank1_sub_device((int)mr, tmp.p, du.p, q3h[(size_t)h], st);
            XM_HIP_CHECK(hipStreamSynchronize(st));   // u is reused
        }
    }
    lap("hub rank-1 terms (device)");
    if (!spd_inverse_device((int)mr, tmp.p, inv.p, st, cfg_.trace))
        throw Error(XM_ERR_ARG, "matrix-free Q: the reduced camera Laplacian is not positive definite (observation graph not connected?)");
    lap("SPD inverse (device)");
    spd_inverse_layout((int)mr, inv.p, vtinv_.p, ldv_, st);
    XM_HIP_CHECK(hipStreamSynchronize(st));
    lap("layout of the inverse");
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

const double *SchurOp::residuals_device(int o, const double *U, const CamArgs &a, hipStream_t st) {
    // the chain with the plain epilogue leaves x_cam / x_l of this U in the scratch buffers
    product(o, EPI_PLAIN, U, 1.0, a, st);
    if (res_.count < (size_t)nobs_) res_.alloc((size_t)nobs_, false);
    XM_DISPATCH_O(o, hipLaunchKernelGGL((schur_obs_residual_kernel<O_>), dim3((unsigned)((nobs_ + 255) / 256)), dim3(256), 0, st, nobs_, obs_cam_.p,
                                        obs_lm_.p, obs_p_.p, U, xc_.p, xl_.p, res_.p));
    check_launch("schur_obs_residual");
    return res_.p;
}
void SchurOp::residuals(int o, const double *U, double *res_host, const CamArgs &a, hipStream_t st) {
    const double *r = residuals_device(o, U, a, st);
    XM_HIP_CHECK(hipMemcpyAsync(res_host, r, (size_t)nobs_ * sizeof(double), hipMemcpyDeviceToHost, st));
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
        for (int k = 0; k < 3; ++k) p[(size_t)3 * l + k] = -hl[(size_t)slot_of_[(size_t)l] * OP + k];
}

void SchurOp::ensure(int o) {
    if (o <= o_alloc_) return;
    const size_t OP = (size_t)pitch_of(o);
    h_.alloc((size_t)m_ * OP); xl_.alloc((size_t)m_ * OP);
    r_.alloc((size_t)ldv_ * OP + 2);            // product input of the dense kernel: ldv rows, zero beyond N-1
    xc_.alloc((size_t)3 * nred_pad_ * OP + 2);
    if (pcg_) {
        const size_t v = (size_t)std::max<int64_t>(n_ - 1, 1) * OP + 2;
        pcg_r_.alloc(v); pcg_p_.alloc(v); pcg_ap_.alloc(v);
        pcg_grid_ = flat_grid(std::max<int64_t>(n_ - 1, 1));
        pcg_parts_.alloc(((size_t)4 * pcg_grid_ + (size_t)qw_grid((int)n_)) * (size_t)o + 16);
    }
    // VT^-1 is symmetric: from xm_tuning_t.sym_min_rows rows on (default 4096, the measured threshold of the dense solver) the chain applies it
    // with the half-traffic kernel (upper triangle only; o = 3, 4)
    vt_sym_ = !pcg_ && 3 * nred_ >= cfg_.sym_min_rows;
    if (vt_sym_ && o >= 3) {
        const int os = std::min(o, 4);
        sym_prow_.alloc(sym_prow_count((int)nred_, ldv_, os));
        sym_pcol_.alloc(sym_pcol_count((int)nred_, ldv_, os), false);
    }
    o_alloc_ = o;
}

int64_t SchurOp::bytes_per_product(int o) const {
    // observation arrays are streamed twice by camera (w, landmark index; w, p, landmark index) and twice by landmark, VT^{-1} once
    // CG form: every iteration streams both observation lists once more (w + index each way); counted with the iterations of the last product
    const int64_t inner = pcg_ ? (int64_t)std::max(1, pcg_last_iters_[1]) * (2 * nobs_ * (8 + 4)) : 8 * (n_ - 1) * (n_ - 1);
    return nobs_ * (8 + 4) + nobs_ * (8 + 24 + 4) + nobs_ * (8 + 24 + 4) + nobs_ * (8 + 4) + inner + 2LL * 8 * 3 * n_ * o;
}

template <int O>
static void schur_product_o(int epi, int64_t n, int64_t nobs, const SchurLm &L, const int64_t *cam_ptr, const int32_t *cam_lm, const double *cam_w, const double *cam_p,
                            const double *Q1,
                            const double *c, const double *q3inv, const double *vtinv, int64_t nred, int64_t ldv, double *h, double *r,
                            double *xc, double *xl, const double *W, double alpha, const CamArgs &a, double *sym_prow, double *sym_pcol, hipStream_t st,
                            Comm *comm, int64_t nred_loc, const std::function<void(const TcgScal *)> &solve_pcg) {
    const TcgScal *sc = (epi == EPI_HESS) ? a.scal : (const TcgScal *)nullptr;
    const int64_t nheavy = L.nheavy, nlight = L.m - L.nheavy;
    const dim3 b(256), gc(qw_grid((int)n)), gy(qw_grid(a.nloc));
    const dim3 glm((unsigned)(nheavy + (nlight + kSchurHeavyThreads - 1) / kSchurHeavyThreads)), blm(kSchurHeavyThreads);
    hipLaunchKernelGGL((schur_lm_h_kernel<O>), glm, blm, 0, st, L, q3inv, W, sc, h);
    hipLaunchKernelGGL((schur_cam_r_kernel<O>), gc, b, 0, st, (int)n, cam_ptr, cam_lm, cam_w, c, W, h, sc, r);
    if (n > 1 && solve_pcg) {
        solve_pcg(sc);   // x_cam by preconditioned CG on the matrix-free reduced camera Laplacian (SchurOp::pcg_solve)
    } else if (n > 1) {
        constexpr int OPc = pitch_of(O);
        CamArgs pa;
        std::memset(&pa, 0, sizeof(pa));
        pa.scal = a.scal;
        if (comm) {   // this rank's rows of VT^-1, then everybody's x_cam (equal chunks: the system is padded to world * nred_loc pseudo-cameras)
            const int64_t a0 = (int64_t)comm->rank * nred_loc;
            pa.nloc = (int)nred_loc; pa.out = xc + (size_t)3 * a0 * OPc;
            launch_qw_dense(O, EPI_PLAIN, vtinv + (size_t)3 * a0 * (size_t)ldv, ldv, r, 1.0, pa, st);
            comm->allgather(xc, (size_t)3 * nred_loc * OPc, st);
        } else {
            pa.nloc = (int)nred; pa.out = xc;
            if (sym_prow && (O == 3 || O == 4)) launch_qw_sym(O, EPI_PLAIN, vtinv, ldv, r, 1.0, pa, sym_prow, sym_pcol, st);
            else launch_qw_dense(O, EPI_PLAIN, vtinv, ldv, r, 1.0, pa, st);
        }
    }
    hipLaunchKernelGGL((schur_lm_x_kernel<O>), glm, blm, 0, st, L, q3inv, h, xc, sc, xl);
    switch (epi) {
        case EPI_PLAIN: hipLaunchKernelGGL((schur_cam_y_kernel<O, EPI_PLAIN>), gy, b, 0, st, n, nobs, cam_ptr, cam_lm, cam_w, cam_p, Q1, c, W, xc, xl, alpha, a); break;
        case EPI_GRAD: hipLaunchKernelGGL((schur_cam_y_kernel<O, EPI_GRAD>), gy, b, 0, st, n, nobs, cam_ptr, cam_lm, cam_w, cam_p, Q1, c, W, xc, xl, alpha, a); break;
        case EPI_HESS: hipLaunchKernelGGL((schur_cam_y_kernel<O, EPI_HESS>), gy, b, 0, st, n, nobs, cam_ptr, cam_lm, cam_w, cam_p, Q1, c, W, xc, xl, alpha, a); break;
        case EPI_CERT:
            if constexpr (O == 1) { hipLaunchKernelGGL((schur_cam_y_kernel<1, EPI_CERT>), gy, b, 0, st, n, nobs, cam_ptr, cam_lm, cam_w, cam_p, Q1, c, W, xc, xl, alpha, a); break; }
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
    SchurLm L;
    L.m = m_; L.nheavy = nheavy_; L.total = ltotal_; L.ptr = lm_ptr_.p; L.gbase = gbase_.p; L.deg = ldeg_.p; L.cam = lm_cam_.p; L.w = lm_w_.p; L.p = lm_p_.p;
    std::function<void(const TcgScal *)> pcg;
    // inner tolerance by the kind of product: the truncated CG tolerates a Hessian applied to 1e-9 (its own stop rule is a RELATIVE residual of
    // 1e-1 .. 1e-6, its recurrences never look at the true residual), the gradient that decides convergence and the certificate do not
    const int kind = (epi == EPI_HESS) ? 1 : 0;
    if (pcg_) pcg = [&](const TcgScal *sc) { XM_DISPATCH_O(o, (pcg_solve<O_>(L, sc, st, kind))); };
    XM_DISPATCH_O(o, (schur_product_o<O_>(epi, n_, nobs_, L, cam_ptr_.p, cam_lm_.p, cam_w_.p, cam_p_.p, Q1_.p,
                                         c_.p, q3inv_.p, vtinv_.p, nred_, ldv_, h_.p, r_.p, xc_.p, xl_.p, W, alpha, a, (vt_sym_ && !comm_) ? sym_prow_.p : (double *)nullptr, sym_pcol_.p, st,
                                         comm_, nred_loc_, pcg)));
    check_launch("schur_product");
}

// x_cam = VT^-1 r by preconditioned CG (kernels above).  The iterations are enqueued in batches without looking at the device: as many as
// the previous product needed (+ 2), then the host reads the state word and tops up in steps of 8 -- one host round trip per product in
// the steady state, where consecutive right-hand sides of a truncated CG need the same number of iterations to within a few.
template <int O>
void SchurOp::pcg_solve(const SchurLm &L, const TcgScal *sc, hipStream_t st, int kind) {
    constexpr int OP = pitch_of(O);
    const int n1 = (int)(n_ - 1);
    PcgArgs a;
    a.n1 = n1; a.grid = pcg_grid_; a.tol2 = pcg_tol_[kind] * pcg_tol_[kind];
    a.b = r_.p; a.dinv = pcg_dinv_.p; a.q2 = q2_.p;
    a.x = xc_.p; a.r = pcg_r_.p; a.p = pcg_p_.p; a.Ap = pcg_ap_.p;
    const size_t seg = (size_t)pcg_grid_ * O;
    a.prz[0] = pcg_parts_.p; a.prz[1] = pcg_parts_.p + seg; a.prr = pcg_parts_.p + 2 * seg; a.pbb = pcg_parts_.p + 3 * seg;
    a.ppap = pcg_parts_.p + 4 * seg; a.cam_grid = qw_grid((int)n_);
    a.st = reinterpret_cast<PcgState *>(pcg_state_.p);
    const dim3 b(256), gf(pcg_grid_), gc(a.cam_grid);
    const int64_t nlight = L.m - L.nheavy;
    const dim3 glm((unsigned)(L.nheavy + (nlight + kSchurHeavyThreads - 1) / kSchurHeavyThreads)), blm(kSchurHeavyThreads);
    hipLaunchKernelGGL((pcg_init_kernel<O>), gf, b, 0, st, a, sc);
    int it = 0, dir_applied = 0;   // dir_applied: the iteration whose direction update is already in the queue (iteration 0 needs none)
    auto enqueue = [&](int upto) {
        for (; it < upto; ++it) {
            if (it != dir_applied) hipLaunchKernelGGL((pcg_dir_kernel<O>), gf, b, 0, st, a, it);
            hipLaunchKernelGGL((pcg_lm_kernel<O>), glm, blm, 0, st, L, q3inv_.p, (const double *)a.p, (const PcgState *)a.st, xl_.p);
            hipLaunchKernelGGL((pcg_cam_kernel<O>), gc, b, 0, st, (int)n_, cam_ptr_.p, cam_lm_.p, cam_w_.p, (const double *)xl_.p, a);
            hipLaunchKernelGGL((pcg_upd_kernel<O>), gf, b, 0, st, a, it);
        }
        // the convergence test of the last update (no-op once done).  When the test fails the same launch IS the direction update of iteration
        // `it`: a following batch must not repeat it (p = z + beta (z + beta p) is no conjugate direction -- ADVICE r5)
        hipLaunchKernelGGL((pcg_dir_kernel<O>), gf, b, 0, st, a, it);
        dir_applied = it;
    };
    int target = std::min(pcg_max_iters_, std::max(4, pcg_last_iters_[kind] + 2));
    for (;;) {
        enqueue(target);
        check_launch("schur_pcg");
        XM_HIP_CHECK(hipMemcpyAsync(pcg_host_, a.st, sizeof(PcgState), hipMemcpyDeviceToHost, st));
        XM_HIP_CHECK(hipStreamSynchronize(st));
        if (pcg_host_->done || target >= pcg_max_iters_) break;
        target = std::min(pcg_max_iters_, target + 8);
    }
    if (pcg_host_->done && pcg_host_->iters > 0) pcg_last_iters_[kind] = pcg_host_->iters;
    pcg_last_relres_ = pcg_host_->relres;
    pcg_products_++;
    pcg_iters_total_ += pcg_host_->done ? pcg_host_->iters : target;
    if (!pcg_host_->done) pcg_unconverged_++;
    (void)OP;
}

}  // namespace xm
