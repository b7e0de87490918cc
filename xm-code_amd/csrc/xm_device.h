// xm_device.h — device-side helpers shared by the kernel translation units (xm_kernels.hip, xm_sell.hip):
// deterministic wave / block reductions (DPP), the column-distributed fused epilogues of the Q*W kernels and their common tail.
#pragma once

#include "xm_common.h"

namespace xm {

// ----------------------------------------------------------------------------------------------------------------
// wave / block reductions (deterministic: fixed shuffle tree, fixed block size 256)
// ----------------------------------------------------------------------------------------------------------------
// 64-lane sum with DPP (data-parallel primitives: cross-lane operands inside the VALU, ~10 cycles per step) instead of
// ds_bpermute (LDS crossbar, ~100 cycles per dependent step): butterfly inside each row of 16 lanes (quad_perm, row_ror),
// then row_bcast:15 / row_bcast:31 accumulate the rows into row 3 and lane 63 is broadcast through a scalar register.
// Fixed tree => bit-reproducible; the result is uniform across the wave.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_mov<0xb1, 0xf>(v);    // quad_perm:[1,0,3,2]
    v += dpp_mov<0x4e, 0xf>(v);    // quad_perm:[2,3,0,1]
    v += dpp_mov<0x124, 0xf>(v);   // row_ror:4
    v += dpp_mov<0x128, 0xf>(v);   // row_ror:8   -> every lane holds its row's sum
    v += dpp_mov<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v += dpp_mov<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3 -> row 3 holds the wave total
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// sum over a lane group: GW = 64 the whole wavefront, GW = 16 one DPP row (the four butterfly steps alone leave every lane of
// a row with its row's sum); the result is uniform across the group
template <int GW>
__device__ __forceinline__ double group_sum(double v) {
    if (GW == 64) return wave_sum(v);
    if (GW == 4) {   // one quad: two quad_perm butterflies leave every lane of the quad with its sum
        v += dpp_mov<0xb1, 0xf>(v);
        v += dpp_mov<0x4e, 0xf>(v);
        return v;
    }
    v += dpp_mov<0xb1, 0xf>(v);
    v += dpp_mov<0x4e, 0xf>(v);
    v += dpp_mov<0x124, 0xf>(v);
    v += dpp_mov<0x128, 0xf>(v);
    return v;
}

// The same sum when only the first O lanes of the group hold non-zero values (the column-distributed epilogues: lane k < O owns column k, the
// others contribute exact zeros) and the result is needed in those lanes only: a 16-lane group stops after the butterfly steps that cover O lanes
// (O <= 4: the two quad steps) -- the steps left out would add zeros, so the bits are the same (up to the sign of a zero sum).  The block-CSR
// kernel's Hessian epilogue makes 14 such sums per camera: 84 of its ~550 instructions per wavefront.
// A whole wavefront per camera (GW = 64: dense products, the per-camera sum of the symmetric pair): the two quad steps and a broadcast of lane 0
// instead of six steps -- these sums are a dependent chain at the very end of a launch (14 of them in the Hessian epilogue).
template <int GW, int O>
__device__ __forceinline__ double group_sum_cols(double v) {
    if ((GW != 16 && GW != 64) || O > 4) return group_sum<GW>(v);
    if (O > 1) v += dpp_mov<0xb1, 0xf>(v);
    if (O > 2) v += dpp_mov<0x4e, 0xf>(v);
    if (GW == 64) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), 0), hi = __builtin_amdgcn_readlane(__double2hiint(v), 0);
        return __hiloint2double(hi, lo);
    }
    return v;
}

// sum over the 256 threads of a block; result valid in every thread.  `sh` must hold >= 4 doubles.
__device__ __forceinline__ double block_sum256(double v, double *sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// Order in which a list of per-workgroup partial sums is added (xm_options_t.sum_grouping): 0 ascending, 1 descending, 2 even
// entries then odd entries.  Each is a fixed order (bit-reproducible runs); they differ in the last bits of the sums.
__device__ __forceinline__ int sum_perm(int i, int count, int grp) {
    if (grp == 1) return count - 1 - i;
    if (grp == 2) { const int h = (count + 1) >> 1; return (i < h) ? 2 * i : 2 * (i - h) + 1; }
    return i;
}
// fixed-order sum of an array of partials by one 256-thread block (identical in every kernel that needs it)
__device__ __forceinline__ double sum_partials256(const double *p, int count, double *sh, int grp = 0) {
    double v = 0.0;
    for (int i = threadIdx.x; i < count; i += 256) v += p[sum_perm(i, count, grp)];
    return block_sum256(v, sh);
}

// four such sums with ONE pair of barriers and all loads in flight together (each sum keeps exactly the summation tree of
// sum_partials256, so results are bit-identical); the fourth array may have its own length (0 = skip)
__device__ __forceinline__ void sum_partials256_x4(const double *p0, const double *p1, const double *p2, int count, const double *p3,
                                                   int count3, double *sh16, double (&out)[4], int grp = 0) {
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < count; i += 256) { const int j = sum_perm(i, count, grp); v[0] += p0[j]; v[1] += p1[j]; v[2] += p2[j]; }
    for (int i = threadIdx.x; i < count3; i += 256) v[3] += p3[sum_perm(i, count3, grp)];
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

// The same sums with the first kPre rounds of loads ISSUED EARLY by the caller (sum_partials_prefetch, before it waits for anything else):
// the values are added in exactly the order of sum_partials256_x4, so the results are bit-identical.  cg_step_kernel reads its scalar
// block and these partial sums from memory the previous launch wrote on another XCD (an L2 miss each): requested together they cost one
// round trip instead of two.
constexpr int kPre = 4;   // covers 1024 workgroups: every partial sum of a problem in the latency regime
struct PartialsPre { double v[4][kPre]; };
__device__ __forceinline__ void sum_partials_prefetch(const double *p0, const double *p1, const double *p2, int count, const double *p3, int count3,
                                                      PartialsPre &pre, int grp) {
#pragma unroll
    for (int r = 0; r < kPre; ++r) {
        const int i = threadIdx.x + 256 * r;
        const int j = (i < count) ? sum_perm(i, count, grp) : 0, j3 = (i < count3) ? sum_perm(i, count3, grp) : 0;
        pre.v[0][r] = (i < count) ? p0[j] : 0.0; pre.v[1][r] = (i < count) ? p1[j] : 0.0; pre.v[2][r] = (i < count) ? p2[j] : 0.0;
        pre.v[3][r] = (i < count3) ? p3[j3] : 0.0;
    }
}
__device__ __forceinline__ void sum_partials256_x4_pre(const double *p0, const double *p1, const double *p2, int count, const double *p3,
                                                       int count3, bool use3, const PartialsPre &pre, double *sh16, double (&out)[4], int grp = 0) {
    double v[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < kPre; ++r) {
        const int i = threadIdx.x + 256 * r;
        if (i < count) { v[0] += pre.v[0][r]; v[1] += pre.v[1][r]; v[2] += pre.v[2][r]; }
        if (use3 && i < count3) v[3] += pre.v[3][r];
    }
    for (int i = threadIdx.x + 256 * kPre; i < count; i += 256) { const int j = sum_perm(i, count, grp); v[0] += p0[j]; v[1] += p1[j]; v[2] += p2[j]; }
    if (use3)
        for (int i = threadIdx.x + 256 * kPre; i < count3; i += 256) v[3] += p3[sum_perm(i, count3, grp)];
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

// From here on (the per-camera algebra of the epilogues) floating-point contraction is OFF: every product and every sum below is rounded
// on its own, so an expression gives the same bits in EVERY kernel it is inlined into.  With hipcc's default (contract = fast) the back end
// decides per kernel which multiply-add pairs become FMAs -- the cost f of a candidate point came out one ulp apart from the gradient
// instantiation and from the role-switching instantiation (EPI_AUTO) of the same product kernel, enough to send two runs of the same solve
// down different paths.  The streaming loops of the products keep the default; this is a few hundred flops per camera.
#pragma clang fp contract(off)
// ----------------------------------------------------------------------------------------------------------------
// fused epilogues, column-distributed: after the wave reduction lane k (< O) owns column k of the camera's 3 x O
// block; every 3-vector below is "that column".  Reductions over k are wave_sum()s with lanes >= O contributing 0,
// so the epilogue costs ~20 VGPRs whatever the rank (a lane-redundant version needs ~30*O).
// ----------------------------------------------------------------------------------------------------------------
struct Col3 {
    double v[3];
};
template <int O>
__device__ __forceinline__ Col3 load_col(const double *base, int cam, int lane) {
    constexpr int OP = pitch_of(O);
    Col3 c;
    const double *p = base + (size_t)cam * 3 * OP + lane;
#pragma unroll
    for (int a = 0; a < 3; ++a) c.v[a] = (lane < O) ? p[a * OP] : 0.0;
    return c;
}
template <int O>
__device__ __forceinline__ void store_col(const Col3 &c, double *base, int cam, int lane) {
    constexpr int OP = pitch_of(O);
    double *p = base + (size_t)cam * 3 * OP + lane;
    if (lane < O) {
#pragma unroll
        for (int a = 0; a < 3; ++a) p[a * OP] = c.v[a];
    }
}
__device__ __forceinline__ double dot3(const Col3 &x, const Col3 &y) { return x.v[0] * y.v[0] + x.v[1] * y.v[1] + x.v[2] * y.v[2]; }

// S = sym(A B^T) with A, B 3 x O blocks held column-per-lane: 9 wave reductions, result uniform across the wave
template <int GW, int O>
__device__ __forceinline__ void sym_abt(const Col3 &A, const Col3 &B, double (&S)[3][3]) {
    double M[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) M[a][b] = group_sum_cols<GW, O>(A.v[a] * B.v[b]);
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) S[a][b] = (M[a][b] + M[b][a]) * 0.5;
}
// X -= S * Y   (column-wise, no communication)
__device__ __forceinline__ void sub_s_times(Col3 &X, const double (&S)[3][3], const Col3 &Y) {
#pragma unroll
    for (int a = 0; a < 3; ++a) X.v[a] -= S[a][0] * Y.v[0] + S[a][1] * Y.v[1] + S[a][2] * Y.v[2];
}

// Operands of the epilogues, fetched while the last tile is still being multiplied (they were written by the previous
// launches, so the loads miss the L2 of this XCD; issuing them early hides ~1 us at the end of every wavefront).
struct EpiOps {
    Col3 R, P, G, Wl, Rr;
    double s, ps, egs, rs;
    double S0[9];
};
template <int O, int EPI>
__device__ __forceinline__ void epi_prefetch(EpiOps &e, int cam, int lane, bool active, const CamArgs &a, int role = EPI) {
    // EPI_AUTO: the role of this launch (EPI_HESS or EPI_GRAD) is run-time state, uniform over the grid; the gradient role works on the CANDIDATE
    // point's buffers (CamArgs.cand).  The kernel argument block itself is never copied or modified: only the fields a role uses are read, where it uses them
    if (!active) return;
    if (EPI == EPI_AUTO) {
        const bool grad = (role == EPI_GRAD);
        const double *Rp = grad ? a.cand.R : a.R, *sp = grad ? a.cand.s : a.s;   // both roles want the point they work at
        e.s = sp[cam];
        e.R = load_col<O>(Rp, cam, lane);
        if (grad) {
            // (the candidate's rows of W are fetched inside epi_grad: held here they are three more live doubles next to the Hessian role's
            // operand set, which the register allocator answers with a spill to scratch -- and a kernel that uses scratch pays for it at every dispatch)
        } else {
            e.ps = a.ps[cam];
            e.egs = a.egs[cam];
            e.P = load_col<O>(a.pR, cam, lane);
            e.G = load_col<O>(a.G, cam, lane);
            e.Rr = load_col<O>(a.rR, cam, lane);
            e.rs = a.rs[cam];
            const double *s0 = a.S0 + (size_t)cam * 9;
#pragma unroll
            for (int j = 0; j < 9; ++j) e.S0[j] = s0[j];
        }
        return;
    }
    if (role == EPI_GRAD) {
        e.s = a.s[cam];
        e.R = load_col<O>(a.R, cam, lane);
        e.Wl = load_col<O>(a.Wloc, cam, lane);
    } else if (role == EPI_HESS) {
        e.s = a.s[cam];
        e.ps = a.ps[cam];
        e.egs = a.egs[cam];
        e.R = load_col<O>(a.R, cam, lane);
        e.P = load_col<O>(a.pR, cam, lane);
        e.G = load_col<O>(a.G, cam, lane);
        e.Rr = load_col<O>(a.rR, cam, lane);
        e.rs = a.rs[cam];
        const double *sp = a.S0 + (size_t)cam * 9;
#pragma unroll
        for (int j = 0; j < 9; ++j) e.S0[j] = sp[j];
    }
}

// Gradient / point-state epilogue: trustregion.h:186-194 (grad), :307-317 (projection), :162-170 (objc) fused.
// h = 2*C*sR rows.  Produces G, egs, S0, rg and this camera's share of {f, <rg,rg>_metric} (uniform on return).
// (the output buffers are explicit arguments: the role-switching instantiation writes the candidate's, a.cand.*, without copying the argument block)
struct GradOut { double *G, *egs, *S0, *rgR, *rgs; };
template <int O, int GW, bool LATE_WL = false>
__device__ __forceinline__ void epi_grad(int cam, int lane, const Col3 &h, const EpiOps &e, const CamArgs &a, const GradOut &o, double &p0, double &p1) {
    const bool anchor = (a.cam0 + cam) == 0;
    const double s = e.s;
    const Col3 &R = e.R;
    const Col3 Wl = LATE_WL ? load_col<O>(a.Wloc, cam, lane) : e.Wl;
    store_col<O>(h, o.G, cam, lane);
    // f = <C sR, sR> + lam * sum_{i>=1} (s_i^2-1)^2 ;  <C sR, sR> = 0.5 * <G, sR>
    const double q = s * s - 1.0;
    const double hW = group_sum_cols<GW, O>(dot3(h, Wl));
    const double hR = group_sum_cols<GW, O>(dot3(h, R));
    p0 = 0.5 * hW + (anchor ? 0.0 : a.lam * q * q);
    const double egs = anchor ? 0.0 : hR + 4.0 * a.lam * (q * s);
    Col3 eg;
#pragma unroll
    for (int r = 0; r < 3; ++r) eg.v[r] = h.v[r] * s;
    double S0[3][3];
    sym_abt<GW, O>(R, eg, S0);
    sub_s_times(eg, S0, R);  // eg is now the Riemannian gradient column
    const double rgs = egs * (s * s);
    store_col<O>(eg, o.rgR, cam, lane);
    if (lane == 0) {
        double *so = o.S0 + (size_t)cam * 9;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) so[r * 3 + c] = S0[r][c];
        o.egs[cam] = egs;
        o.rgs[cam] = rgs;
    }
    const double rsds = rgs / s;
    p1 = group_sum_cols<GW, O>(dot3(eg, eg)) + rsds * rsds;
}

// Hessian epilogue: trustregion.h:227-255 (ehess) + :277-295 (ehess2rhess) fused.  h = 2*C*(s.*Ru + su.*R) rows.
// Produces Hp = (rhr, rhs) and this camera's shares of <p,Hp>, <r,Hp>, <Hp,Hp> (product metric).
template <int O, int GW>
__device__ __forceinline__ void epi_hess(int cam, int lane, const Col3 &h, const EpiOps &e, const CamArgs &a, double &p0, double &p1,
                                         double &p2) {
    const bool anchor = (a.cam0 + cam) == 0;
    const double s = e.s;
    const double ps = anchor ? 0.0 : e.ps;
    const Col3 &R = e.R;
    const Col3 &P = e.P;
    const Col3 &G = e.G;
    // hs = sum(CsRu.*R) + sum(CsR.*Ru) + 4 lam (3 s^2 - 1) su
    const double hRGP = group_sum_cols<GW, O>(dot3(h, R)) + group_sum_cols<GW, O>(dot3(G, P));
    const double hs = anchor ? 0.0 : hRGP + 4.0 * a.lam * ((3.0 * s * s - 1.0) * ps);
    // hr = CsRu.*s + CsR.*su
    Col3 rh;
#pragma unroll
    for (int r = 0; r < 3; ++r) rh.v[r] = h.v[r] * s + G.v[r] * ps;
    double S0[3][3], S1[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) S0[r][c] = e.S0[r * 3 + c];
    sub_s_times(rh, S0, P);   // rhr = ehessR - Ru * sym(R' egradR)
    sym_abt<GW, O>(R, rh, S1);
    sub_s_times(rh, S1, R);   // rhr -= R * sym(R' rhr)
    const double rhs = anchor ? 0.0 : hs * (s * s) + (ps * s) * e.egs;
    store_col<O>(rh, a.HpR, cam, lane);
    if (lane == 0) a.Hps[cam] = rhs;
    if (a.Bout) {   // multi-rank tCG: the image of Hp under the map (xR, xs) -> s.*xR + xs.*R travels with the partial sums
        Col3 b;
#pragma unroll
        for (int r = 0; r < 3; ++r) b.v[r] = s * rh.v[r] + rhs * R.v[r];
        store_col<O>(b, a.Bout, cam, lane);
    }
    p0 = group_sum_cols<GW, O>(dot3(P, rh)) + ps * (rhs / (s * s));
    // <r,Hp> and <Hp,Hp> in the same metric: with them the residual norm after the CG step follows without a second
    // global reduction, |r + alpha Hp|^2 = <r,r> + 2 alpha <r,Hp> + alpha^2 <Hp,Hp>   (one flat kernel per iteration)
    const double rsv = anchor ? 0.0 : e.rs;
    p1 = group_sum_cols<GW, O>(dot3(e.Rr, rh)) + rsv * (rhs / (s * s));
    const double hq = rhs / s;
    p2 = group_sum_cols<GW, O>(dot3(rh, rh)) + hq * hq;
}

// ----------------------------------------------------------------------------------------------------------------
// common tail of the Q*W kernels: wave reduction of the 3 x O accumulators, epilogue, per-workgroup partial sums
// ----------------------------------------------------------------------------------------------------------------
// tail shared by every product kernel: h = this camera's 3 x O block of alpha * Q * W, column k in lane k of the camera's lane group
template <int O, int EPI, int GW, int NSLOT>
__device__ __forceinline__ void qw_tail(int cam, int lane, int slot, bool active, Col3 h, const CamArgs &a, const EpiOps &e, double (*red)[3], int role = EPI) {
    // `lane` = position inside the camera's lane group (0..GW-1), `slot` = index of that group inside the workgroup
    constexpr int OP = pitch_of(O);
    double p0 = 0.0, p1 = 0.0, p2 = 0.0;
    if (active) {  // uniform across the group
        if (EPI == EPI_PLAIN) {
            store_col<O>(h, a.out, cam, lane);
        } else if (EPI == EPI_GRAD) {
            const GradOut go = {a.G, a.egs, a.S0, a.rgR, a.rgs};
            epi_grad<O, GW>(cam, lane, h, e, a, go, p0, p1);
        } else if (EPI == EPI_HESS) {
            epi_hess<O, GW>(cam, lane, h, e, a, p0, p1, p2);
        } else if (EPI == EPI_AUTO) {
            if (role == EPI_GRAD) { const GradOut go = {a.cand.G, a.cand.egs, a.cand.S0, a.cand.rgR, a.cand.rgs}; epi_grad<O, GW, true>(cam, lane, h, e, a, go, p0, p1); }
            else epi_hess<O, GW>(cam, lane, h, e, a, p0, p1, p2);
        } else if (EPI == EPI_CERT) {
            // y_i = (Q x)_i + dz_i * x[3i] e_0 - Lam_i x_i      (O == 1, lane 0 owns the column)
            const double *x = a.Wloc + (size_t)cam * 3 * OP;
            const double *L = a.Lam + (size_t)cam * 9;
            const double x0 = x[0], x1 = x[OP], x2 = x[2 * OP];
#pragma unroll
            for (int r = 0; r < 3; ++r) h.v[r] -= L[r * 3 + 0] * x0 + L[r * 3 + 1] * x1 + L[r * 3 + 2] * x2;
            h.v[0] += a.dz[cam] * x0;
            store_col<O>(h, a.out, cam, lane);
        }
    }
    if (EPI == EPI_GRAD || EPI == EPI_HESS || EPI == EPI_AUTO) {
        if (lane == 0) { red[slot][0] = p0; red[slot][1] = p1; red[slot][2] = p2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double t0 = 0.0, t1 = 0.0, t2 = 0.0;
#pragma unroll
            for (int q = 0; q < NSLOT; ++q) { t0 += red[q][0]; t1 += red[q][1]; t2 += red[q][2]; }
            double *parts = (EPI == EPI_AUTO && role == EPI_GRAD) ? a.cand.partials : a.partials;
            parts[blockIdx.x] = t0;
            parts[gridDim.x + blockIdx.x] = t1;
            if (EPI == EPI_HESS || (EPI == EPI_AUTO && role == EPI_HESS)) parts[2 * gridDim.x + blockIdx.x] = t2;
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------
// common tail of the Q*W kernels: wave reduction of the 3 x O accumulators, epilogue, per-workgroup partial sums
// ----------------------------------------------------------------------------------------------------------------
template <int O, int EPI, int GW, int NSLOT>
__device__ __forceinline__ void qw_finish(int cam, int lane, int slot, bool active, double (&acc)[3][O], double alpha,
                                          const CamArgs &a, const EpiOps &e, double (*red)[3], int role = EPI) {
    double hv[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < O; ++k) {
            const double t = alpha * group_sum<GW>(acc[r][k]);
            hv[r] = (lane == k) ? t : hv[r];
        }
    Col3 h;
    h.v[0] = hv[0]; h.v[1] = hv[1]; h.v[2] = hv[2];
    qw_tail<O, EPI, GW, NSLOT>(cam, lane, slot, active, h, a, e, red, (EPI == EPI_AUTO) ? role : (int)EPI);
}

#pragma clang fp contract(fast)   // (the including translation unit's streaming loops keep the compiler's default)

}  // namespace xm
