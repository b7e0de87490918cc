// xm_sell2.hip — chunk-tiled sliced ELL: block-sparse Q*W, per-camera sum and fused epilogue in ONE launch (layout and rationale:
// xm_sell2.h).  Replaces, for view-graph-sparse Q, the product the reference runs as cublasDgemm on a dense matrix (Dense/matmul.h:42-87,
// call sites trustregion.h:165,187,237,553, checkeig.h:182) together with the element-wise kernels behind it (trustregion.h:186-194,
// 227-255, 277-295, 307-317).
#include "xm_sell2.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "xm_device.h"
#include "xm_sell_codec.h"

namespace xm {

// ------------------------------------------------------------------------------------------------------------------
// host: build the layout description
// ------------------------------------------------------------------------------------------------------------------
void sell2_build_host(const int64_t *rowptr, const int32_t *colidx, int64_t nloc, int64_t ncols, int S, int kmax, Sell2Host &out,
                      int64_t diag_row0) {
    if (!(S == 1 || S == 2 || S == 4 || S == 8)) throw Error(XM_ERR_ARG, "SELL: slabs must be 1, 2, 4 or 8");
    if (kmax < 1) throw Error(XM_ERR_ARG, "SELL: kmax must be >= 1");
    if (nloc < 0 || ncols < 1 || !rowptr) throw Error(XM_ERR_ARG, "SELL: bad sizes");
    if (ncols >= kSell2MaxCols) throw Error(XM_ERR_ARG, "SELL: the chunk-tiled layout addresses fewer than 2^24 cameras");
    out = Sell2Host();
    out.nloc = nloc; out.ncols = ncols; out.S = S; out.kmax = kmax;
    const int64_t b0 = rowptr[0], nb = rowptr[nloc] - b0;
    if (nb < 0) throw Error(XM_ERR_ARG, "BSR3: rowptr is not monotone");
    if (nb > 0 && !colidx) throw Error(XM_ERR_ARG, "BSR3: colidx missing");
    if (nb >= (1LL << 47)) throw Error(XM_ERR_ARG, "SELL: too many blocks");
    // order[i]: i-th kept block with every row in ascending column order; row r owns order[ro[r] .. ro[r+1])
    std::vector<int64_t> order;
    order.reserve((size_t)nb);
    std::vector<int64_t> ro((size_t)nloc + 1, 0);
    if (diag_row0 >= 0) out.diag_src.assign((size_t)nloc, -1);
    for (int64_t r = 0; r < nloc; ++r) {
        const int64_t a = rowptr[r], e = rowptr[r + 1];
        if (e < a) throw Error(XM_ERR_ARG, "BSR3: rowptr is not monotone");
        bool sorted = true;
        const size_t first = order.size();
        for (int64_t q = a; q < e; ++q) {
            const int32_t c = colidx[q];
            if (c < 0 || (int64_t)c >= ncols) throw Error(XM_ERR_ARG, "BSR3: column index out of range");
            if (diag_row0 >= 0 && (int64_t)c == diag_row0 + r) {
                if (out.diag_src[(size_t)r] >= 0) throw Error(XM_ERR_ARG, "SELL: duplicate diagonal block");
                out.diag_src[(size_t)r] = q;
                continue;
            }
            if (order.size() > first && colidx[order.back()] > c) sorted = false;
            order.push_back(q);
        }
        if (!sorted)
            std::stable_sort(order.begin() + (int64_t)first, order.end(), [&](int64_t x, int64_t y) { return colidx[x] < colidx[y]; });
        ro[(size_t)r + 1] = (int64_t)order.size();
    }
    auto slab_of = [&](int32_t c) { return (int)(((int64_t)c * S) / ncols); };
    // split[r * (S + 1) + s]: first position in `order` of row r's blocks in slab >= s (rows are column-sorted, so a slab is a sub-range)
    std::vector<int64_t> split((size_t)nloc * (S + 1) + 1, 0);
    for (int64_t r = 0; r < nloc; ++r) {
        int64_t q = ro[(size_t)r];
        const int64_t e = ro[(size_t)r + 1];
        for (int s = 0; s <= S; ++s) {
            while (q < e && slab_of(colidx[order[(size_t)q]]) < s) ++q;
            split[(size_t)r * (S + 1) + s] = (s == S) ? e : q;
        }
    }
    const int64_t nchunks = (nloc + kSell2Chunk - 1) / kSell2Chunk;
    out.nchunks = nchunks;
    // blocks per (slab, chunk), slices per (slab, chunk)
    std::vector<int64_t> B((size_t)S * nchunks, 0);
    std::vector<int32_t> nsub((size_t)S * nchunks, 0);
    out.tile_ptr.assign((size_t)nchunks + 1, 0);
    const int64_t cap = 64LL * kmax;
    for (int64_t k = 0; k < nchunks; ++k) {
        const int64_t r0 = k * kSell2Chunk, r1 = std::min(nloc, r0 + kSell2Chunk);
        int64_t tot = 0;
        for (int s = 0; s < S; ++s) {
            int64_t b = 0;
            for (int64_t r = r0; r < r1; ++r) b += split[(size_t)r * (S + 1) + s + 1] - split[(size_t)r * (S + 1) + s];
            B[(size_t)s * nchunks + k] = b;
            nsub[(size_t)s * nchunks + k] = (int32_t)((b + cap - 1) / cap);
            tot += b;
        }
        if (tot == 0) nsub[(size_t)k] = 1;   // a chunk without any block still needs one arrival: an empty slice in slab 0 runs its epilogue
        int64_t nt = 0;
        for (int s = 0; s < S; ++s) nt += nsub[(size_t)s * nchunks + k];
        if (out.tile_ptr[(size_t)k] + nt > 2147483000LL) throw Error(XM_ERR_ARG, "SELL: too many tiles");
        out.tile_ptr[(size_t)k + 1] = out.tile_ptr[(size_t)k] + (int32_t)nt;
    }
    out.ntiles = out.tile_ptr[(size_t)nchunks];
    std::vector<int32_t> tile_next(out.tile_ptr.begin(), out.tile_ptr.end() - 1);
    out.slab_start.assign((size_t)S + 1, 0);
    out.slice_off.clear();
    out.slice_off.push_back(0);
    struct Ent { int64_t pos; int32_t row; };
    std::vector<Ent> list;
    for (int s = 0; s < S; ++s) {
        for (int64_t k = 0; k < nchunks; ++k) {
            const int ns = nsub[(size_t)s * nchunks + k];
            if (ns == 0) continue;
            const int64_t r0 = k * kSell2Chunk, r1 = std::min(nloc, r0 + kSell2Chunk);
            list.clear();
            for (int64_t r = r0; r < r1; ++r)
                for (int64_t q = split[(size_t)r * (S + 1) + s]; q < split[(size_t)r * (S + 1) + s + 1]; ++q)
                    list.push_back(Ent{order[(size_t)q], (int32_t)(r - r0)});
            const int64_t Bt = (int64_t)list.size();
            for (int i = 0; i < ns; ++i) {
                const int64_t lo = Bt * i / ns, hi = Bt * (i + 1) / ns, Bi = hi - lo;
                const int64_t K = (Bi + 63) / 64;
                const int64_t off = out.slice_off.back();
                if (off + K > (1LL << 40)) throw Error(XM_ERR_ARG, "SELL: too many steps");
                out.slice_off.push_back(off + K);
                out.slice_chunk.push_back((int32_t)k);
                out.slice_tile.push_back(tile_next[(size_t)k]++);
                out.kind.resize((size_t)(off + K));
                out.src.resize((size_t)(off + K) * 64, -1);
                out.lane_meta.resize(out.lane_meta.size() + 64, 0);
                int32_t *meta = out.lane_meta.data() + out.lane_meta.size() - 64;
                for (int64_t j = 0; j < K; ++j) out.kind[(size_t)(off + j)] = (uint8_t)((j < (K & ~1LL)) ? (j & 1) : 2);
                int arow[64];
                for (int l = 0; l < 64; ++l) {
                    arow[l] = -1;
                    int nflag = 0, lastflag = 1, lastrow = 0;
                    bool any = false;
                    for (int64_t j = 0; j < K; ++j) {
                        const int64_t idx = (int64_t)l * K + j;
                        if (idx >= Bi) break;
                        const Ent &en = list[(size_t)(lo + idx)];
                        const int flag = (idx == Bi - 1 || list[(size_t)(lo + idx + 1)].row != en.row) ? 1 : 0;
                        out.src[(size_t)(off + j) * 64 + l] = en.pos | ((int64_t)en.row << 48) | ((int64_t)flag << 54);
                        if (!any) arow[l] = en.row;
                        any = true;
                        nflag += flag; lastflag = flag; lastrow = en.row;
                    }
                    if (any && nflag > 0 && !lastflag) meta[l] |= (1 << 13) | (lastrow << 14);   // the lane ends inside row `lastrow`
                }
                // row r adds the first runs of the lanes whose first block belongs to it (consecutive lanes)
                for (int l = 0; l < 64; ++l) {
                    const int r = arow[l];
                    if (r < 0) continue;
                    const int na = (meta[r] >> 6) & 127;
                    if (na == 0) meta[r] |= l;   // la
                    meta[r] = (meta[r] & ~(127 << 6)) | ((na + 1) << 6);
                }
            }
        }
        out.slab_start[(size_t)s + 1] = (int32_t)(out.slice_off.size() - 1);
    }
    out.nslices = (int64_t)out.slice_off.size() - 1;
    out.nsteps = out.slice_off.back();
    if (out.nslices > 2147483000LL / 64) throw Error(XM_ERR_ARG, "SELL: too many slices");
}

// ------------------------------------------------------------------------------------------------------------------
// device: fill the interleaved arrays from the CSR arrays (one thread per (step, lane))
// ------------------------------------------------------------------------------------------------------------------
template <int NQ>
__global__ __launch_bounds__(256) void sell2_fill_kernel(int64_t nsteps, const int64_t *__restrict__ src, const uint8_t *__restrict__ kind,
                                                          const int32_t *__restrict__ colidx, const double *__restrict__ blocks, int64_t b0,
                                                          int32_t *__restrict__ cols, double *__restrict__ blk) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= nsteps * 64) return;
    const int64_t g = t >> 6;
    const int lane = (int)(t & 63);
    const int kd = kind[g];
    const int64_t sw = src[t];
    const bool pad = sw < 0;
    const int64_t s = sw & ((1LL << 48) - 1);
    const unsigned int fl = pad ? 0u : (unsigned int)((sw >> 48) & 127);   // row (6 bits) | row-end flag << 6
    const unsigned int c = (pad ? 0u : (unsigned int)colidx[s - b0]) | ((fl & 63u) << 24) | ((fl >> 6) << 31);
    double q[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) q[e] = pad ? 0.0 : blocks[(s - b0) * 9 + e];
    double v[NQ];
    if constexpr (NQ == 4) {
        double qq[4];
        block_to_quat(q, qq);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = qq[e];
    } else {
#pragma unroll
        for (int e = 0; e < 9; ++e) v[e] = q[e];
    }
    constexpr int U = 64 * NQ;
    if (kd == 2) {
        cols[g * 64 + lane] = (int32_t)c;
#pragma unroll
        for (int e = 0; e < NQ; ++e) blk[g * U + e * 64 + lane] = v[e];
    } else {
        const int64_t gb = g - kd;   // first unit of the pair
        cols[gb * 64 + lane * 2 + kd] = (int32_t)c;
#pragma unroll
        for (int e = 0; e < NQ; ++e) blk[gb * U + e * 128 + lane * 2 + kd] = v[e];
    }
}
__global__ __launch_bounds__(256) void sell2_diag_kernel(int64_t nloc, const int64_t *__restrict__ diag_src, const double *__restrict__ blocks, int64_t b0,
                                                          double *__restrict__ diag) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= nloc) return;
    const int64_t s = diag_src[r];
    diag[r] = (s < 0) ? 0.0 : blocks[(s - b0) * 9];
}

Sell2Matrix::Sell2Matrix(const int64_t *rowptr, const int32_t *colidx, const double *blocks, int64_t nloc, int64_t ncols, int S, int kmax,
                         hipStream_t st, int codec, int64_t row0) {
    if (codec != SELL_CODEC_FULL && codec != SELL_CODEC_QUAT) throw Error(XM_ERR_ARG, "SELL: unknown codec");
    codec_ = codec; row0_ = row0;
    if (codec == SELL_CODEC_QUAT) check_viewgraph_blocks(rowptr, colidx, blocks, nloc, row0);
    Sell2Host h;
    sell2_build_host(rowptr, colidx, nloc, ncols, S, kmax, h, codec == SELL_CODEC_QUAT ? row0 : -1);
    ncols_ = ncols; nloc_ = nloc; nsteps_ = h.nsteps; nslices_ = h.nslices; nchunks_ = h.nchunks; ntiles_ = h.ntiles; S_ = S;
    const int64_t b0 = rowptr[0], nb = rowptr[nloc] - b0;
    b0_ = b0;
    auto up32 = [](DevBuf<int32_t> &d, const std::vector<int32_t> &v) {
        d.alloc(std::max<size_t>(v.size(), 1), false);
        if (!v.empty()) XM_HIP_CHECK(hipMemcpy(d.p, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    };
    slice_off_.alloc(h.slice_off.size(), false);
    XM_HIP_CHECK(hipMemcpy(slice_off_.p, h.slice_off.data(), h.slice_off.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    up32(slab_start_, h.slab_start); up32(slice_chunk_, h.slice_chunk); up32(slice_tile_, h.slice_tile); up32(tile_ptr_, h.tile_ptr);
    up32(lane_meta_, h.lane_meta);
    cols_.alloc((size_t)std::max<int64_t>(nsteps_, 1) * 64, false);
    blk_.alloc((size_t)std::max<int64_t>(nsteps_, 1) * 64 * (codec_ == SELL_CODEC_QUAT ? 4 : 9), false);
    arrived_.alloc((size_t)std::max<int64_t>(nchunks_, 1));   // zeroed
    if (codec_ == SELL_CODEC_QUAT) {
        diag_.alloc((size_t)std::max<int64_t>(nloc, 1));
        diag_src_.alloc((size_t)std::max<int64_t>(nloc, 1), false);
        if (nloc > 0) XM_HIP_CHECK(hipMemcpy(diag_src_.p, h.diag_src.data(), (size_t)nloc * sizeof(int64_t), hipMemcpyHostToDevice));
    }
    if (nsteps_ > 0) {
        src_.alloc(h.src.size(), false); kind_.alloc(h.kind.size(), false);   // kept: refill() after a device-side update of the values
        XM_HIP_CHECK(hipMemcpy(src_.p, h.src.data(), h.src.size() * sizeof(int64_t), hipMemcpyHostToDevice));
        XM_HIP_CHECK(hipMemcpy(kind_.p, h.kind.data(), h.kind.size(), hipMemcpyHostToDevice));
    }
    if (nsteps_ > 0 || (codec_ == SELL_CODEC_QUAT && nb > 0)) {
        DevBuf<int32_t> dci; DevBuf<double> dbl;
        dci.alloc((size_t)std::max<int64_t>(nb, 1), false); dbl.alloc((size_t)std::max<int64_t>(nb, 1) * 9, false);
        XM_HIP_CHECK(hipMemcpy(dci.p, colidx + b0, (size_t)nb * sizeof(int32_t), hipMemcpyHostToDevice));
        XM_HIP_CHECK(hipMemcpy(dbl.p, blocks + b0 * 9, (size_t)nb * 9 * sizeof(double), hipMemcpyHostToDevice));
        refill(dci.p, dbl.p, st);
        XM_HIP_CHECK(hipStreamSynchronize(st));
    }
    // workgroups: 4 slices each, dealt to the XCDs that serve the slab (block b -> XCD b % 8)
    const int per = 8 / S;
    int64_t imax = 0;
    for (int s = 0; s < S; ++s) {
        const int64_t nsl = h.slab_start[(size_t)s + 1] - h.slab_start[(size_t)s];
        const int64_t wgs = (nsl + 3) / 4;
        imax = std::max(imax, (wgs + per - 1) / per);
    }
    grid_ = (int)(imax * 8);
}

void Sell2Matrix::refill(const int32_t *d_colidx, const double *d_blocks, hipStream_t st) {
    if (codec_ == SELL_CODEC_QUAT && nloc_ > 0) {
        hipLaunchKernelGGL(sell2_diag_kernel, dim3((unsigned)((nloc_ + 255) / 256)), dim3(256), 0, st, nloc_, diag_src_.p, d_blocks, b0_, diag_.p);
        check_launch("sell2_diag");
    }
    if (nsteps_ <= 0) return;
    const int64_t threads = nsteps_ * 64;
    if (codec_ == SELL_CODEC_QUAT)
        hipLaunchKernelGGL(sell2_fill_kernel<4>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, nsteps_, src_.p, kind_.p, d_colidx, d_blocks,
                           b0_, cols_.p, blk_.p);
    else
        hipLaunchKernelGGL(sell2_fill_kernel<9>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, nsteps_, src_.p, kind_.p, d_colidx, d_blocks,
                           b0_, cols_.p, blk_.p);
    check_launch("sell2_fill");
}
int64_t Sell2Matrix::stream_bytes() const {
    return nsteps_ * 64 * (4 + 8 * (int64_t)(codec_ == SELL_CODEC_QUAT ? 4 : 9)) + nslices_ * (64 * 4 + 16) + (codec_ == SELL_CODEC_QUAT ? 8 * nloc_ : 0);
}

Sell2Args Sell2Matrix::args(int o) {
    if (o > tiles_o_) {
        tiles_.alloc((size_t)std::max<int64_t>(ntiles_, 1) * 64 * 3 * (size_t)o, false);
        tiles_o_ = o;
    }
    Sell2Args a;
    a.slice_off = slice_off_.p; a.slab_start = slab_start_.p; a.slice_chunk = slice_chunk_.p; a.slice_tile = slice_tile_.p; a.tile_ptr = tile_ptr_.p;
    a.lane_meta = lane_meta_.p; a.cols = cols_.p; a.blk = blk_.p;
    a.diag = (codec_ == SELL_CODEC_QUAT) ? diag_.p : nullptr;
    a.row0 = row0_; a.tiles = tiles_.p; a.arrived = arrived_.p; a.nchunks = (int)nchunks_; a.S = S_;
    a.wstride = 3 * pitch_of(o);
    return a;
}

// ------------------------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------------------------
typedef double d2a __attribute__((ext_vector_type(2)));               // 16-byte aligned pair
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));   // pair at 8-byte alignment (records of W)
typedef int i2a __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void wave_lds_sync() {   // orders the LDS accesses of the lanes of ONE wavefront
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// agent-scope (write-through) stores and L1-bypassing loads of the tiles: the hand-off between wavefronts on different XCDs.  Builtins,
// not inline assembly: the compiler has to know when a loaded register is valid (an asm load followed by a separate s_waitcnt lets it
// copy the destination before the data has landed -- seen: lanes 12..15 of every 16 received stale values).
__device__ __forceinline__ void st_agent(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// 64 consecutive records of REC doubles (one per lane): fetched with lane-consecutive loads and turned into record-per-lane through LDS
template <int REC>
__device__ __forceinline__ void rec_fetch(const double *g, int nelem, int lane, double (&tmp)[REC]) {
#pragma unroll
    for (int i = 0; i < REC; ++i) {
        const int idx = lane + 64 * i;
        tmp[i] = (idx < nelem) ? g[idx] : 0.0;
    }
}
template <int REC>
__device__ __forceinline__ void rec_unpack(const double (&tmp)[REC], double *T, int lane, double (&out)[REC]) {
#pragma unroll
    for (int i = 0; i < REC; ++i) T[lane + 64 * i] = tmp[i];
    wave_lds_sync();
#pragma unroll
    for (int e = 0; e < REC; ++e) out[e] = T[lane * REC + e];
    wave_lds_sync();
}
template <int REC>
__device__ __forceinline__ void rec_load(const double *g, int nelem, double *T, int lane, double (&out)[REC]) {
    double tmp[REC];
    rec_fetch<REC>(g, nelem, lane, tmp);
    rec_unpack<REC>(tmp, T, lane, out);
}
template <int REC>
__device__ __forceinline__ void rec_store(double *g, int nelem, double *T, int lane, const double (&in)[REC]) {
#pragma unroll
    for (int e = 0; e < REC; ++e) T[lane * REC + e] = in[e];
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < REC; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nelem) g[idx] = T[idx];
    }
    wave_lds_sync();
}

// 3 x O blocks held by ONE lane (record layout r * OP + k); every sum below has a fixed order
template <int O>
__device__ __forceinline__ double bdot(const double (&x)[3 * pitch_of(O)], const double (&y)[3 * pitch_of(O)]) {
    constexpr int OP = pitch_of(O);
    double t = 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < O; ++k) t = fma(x[r * OP + k], y[r * OP + k], t);
    return t;
}
template <int O>
__device__ __forceinline__ void bsym_abt(const double (&A)[3 * pitch_of(O)], const double (&B)[3 * pitch_of(O)], double (&S)[3][3]) {
    constexpr int OP = pitch_of(O);
    double M[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < O; ++k) t = fma(A[a * OP + k], B[b * OP + k], t);
            M[a][b] = t;
        }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) S[a][b] = (M[a][b] + M[b][a]) * 0.5;
}
template <int O>
__device__ __forceinline__ void bsub_s_times(double (&X)[3 * pitch_of(O)], const double (&S)[3][3], const double (&Y)[3 * pitch_of(O)]) {
    constexpr int OP = pitch_of(O);
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int k = 0; k < O; ++k) X[a * OP + k] -= S[a][0] * Y[k] + S[a][1] * Y[OP + k] + S[a][2] * Y[2 * OP + k];
}

// ------------------------------------------------------------------------------------------------------------------
// device: the product.  One wavefront per slice; the last wavefront to arrive for a chunk runs the chunk's epilogue.
// ------------------------------------------------------------------------------------------------------------------
template <int O, int GM, int PIPE, int CODEC, int EPI>
__device__ __forceinline__ void qw_sell2_body(const Sell2Args &m, const double *__restrict__ W, double alpha, const CamArgs &a) {
    constexpr int OP = pitch_of(O), REC = 3 * OP, NPR = (REC + 1) / 2, RECP = (REC + 1) & ~1, NV = 3 * O;
    constexpr int NQ = (CODEC == SELL_CODEC_QUAT) ? 4 : 9;   // doubles per stored block
    constexpr int TSZ = 64 * ((RECP > 10) ? RECP : 10);      // transposition buffer: a step's records of W, later the records of the epilogue
    constexpr int WSZ = TSZ + 64 * NV;                        // + the row slots
    if (EPI == EPI_HESS) {
        if (a.scal->status != 0) return;
    }
    __shared__ __attribute__((aligned(16))) double lds[4 * WSZ];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int per = 8 / m.S;
    const int x = blockIdx.x & 7, bi = blockIdx.x >> 3;
    const int slab = x / per, sub = x - slab * per;
    const int c = m.slab_start[slab] + (bi * per + sub) * 4 + wave;
    if (c >= m.slab_start[slab + 1]) return;   // wave-uniform
    const int64_t off = m.slice_off[c];
    const int w = (int)(m.slice_off[c + 1] - off);
    const int np = w >> 1;
    const bool tail = (w & 1) != 0;
    const int chunk = m.slice_chunk[c], tile = m.slice_tile[c];
    const int meta = m.lane_meta[(size_t)c * 64 + lane];
    const int32_t *cb = m.cols + off * 64;
    const double *bb = m.blk + off * (64 * NQ);
    double *T = lds + wave * WSZ;
    double *RA = T + TSZ;   // RA[row * NV + e]: sum of the run that ends row `row` (not a lane's first run), or of a lane's last, unfinished run

    double acc[NV], A[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) { acc[e] = 0.0; A[e] = 0.0; RA[lane * NV + e] = 0.0; }
    bool first = true;   // the lane is still in its first run (possibly the continuation of the previous lane's row)
    wave_lds_sync();

    auto load_cols = [&](int p) -> i2a { return __builtin_nontemporal_load(reinterpret_cast<const i2a *>(cb) + (size_t)p * 64 + lane); };
    auto load_blk = [&](int p, d2a (&q)[NQ]) {
        const d2a *b = reinterpret_cast<const d2a *>(bb) + (size_t)p * (64 * NQ) + lane;
#pragma unroll
        for (int e = 0; e < NQ; ++e) q[e] = __builtin_nontemporal_load(b + e * 64);
    };
    auto expand = [&](const d2a (&q)[NQ], double (&q0)[9], double (&q1)[9]) {
        if constexpr (CODEC == SELL_CODEC_QUAT) {
            quat_to_block(q[0].x, q[1].x, q[2].x, q[3].x, q0);
            quat_to_block(q[0].y, q[1].y, q[2].y, q[3].y, q1);
        } else {
#pragma unroll
            for (int e = 0; e < 9; ++e) { q0[e] = q[e].x; q1[e] = q[e].y; }
        }
    };
    // GM 0: every lane reads its own record (REC doubles at 8-byte alignment)
    auto gather0 = [&](int jw, double (&wv)[REC]) {
        const double *wp = W + (size_t)(jw & 0xffffff) * REC;
#pragma unroll
        for (int i = 0; i < REC / 2; ++i) {
            const d2u t = *reinterpret_cast<const d2u *>(wp + 2 * i);
            wv[2 * i] = t.x; wv[2 * i + 1] = t.y;
        }
        if (REC & 1) wv[REC - 1] = wp[REC - 1];
    };
    // GM 1: the 64 records of a step are fetched element-per-lane and turned back into lane-per-record through LDS (xm_sell.hip)
    auto gather1 = [&](int jw, d2u (&raw)[NPR]) {
        const int j = jw & 0xffffff;
#pragma unroll
        for (int i = 0; i < NPR; ++i) {
            const int g = lane + 64 * i;
            const int rec = g / NPR, part = g - rec * NPR;
            const int start = (2 * part < REC - 2) ? 2 * part : REC - 2;
            const int jr = __shfl(j, rec, 64);
            raw[i] = *reinterpret_cast<const d2u *>(W + (size_t)jr * REC + start);
        }
    };
    auto transpose1 = [&](const d2u (&raw)[NPR], double (&wv)[REC]) {
#pragma unroll
        for (int i = 0; i < NPR; ++i) {
            const int g = lane + 64 * i;
            const int rec = g / NPR, part = g - rec * NPR;
            if ((REC & 1) && part == NPR - 1) T[rec * RECP + REC - 1] = raw[i].y;
            else *reinterpret_cast<d2a *>(T + rec * RECP + 2 * part) = d2a{raw[i].x, raw[i].y};
        }
        wave_lds_sync();
#pragma unroll
        for (int i = 0; i < RECP / 2; ++i) {
            const d2a t = *reinterpret_cast<const d2a *>(T + lane * RECP + 2 * i);
            if (2 * i < REC) wv[2 * i] = t.x;
            if (2 * i + 1 < REC) wv[2 * i + 1] = t.y;
        }
        wave_lds_sync();
    };
    auto fma_step = [&](const double (&q)[9], const double (&wv)[REC]) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int k = 0; k < O; ++k)
                acc[r * O + k] = fma(q[3 * r + 2], wv[2 * OP + k], fma(q[3 * r + 1], wv[OP + k], fma(q[3 * r], wv[k], acc[r * O + k])));
    };
    // the row ends after this block: the run's sum leaves the accumulator
    auto row_end = [&](int jw) {
        if (jw < 0) {
            if (first) {
#pragma unroll
                for (int e = 0; e < NV; ++e) A[e] = acc[e];
                first = false;
            } else {
                double *dst = RA + ((jw >> 24) & 63) * NV;
#pragma unroll
                for (int e = 0; e < NV; ++e) dst[e] = acc[e];
            }
#pragma unroll
            for (int e = 0; e < NV; ++e) acc[e] = 0.0;
        }
    };
    struct Gbuf { double w[2][(GM == 0) ? REC : 1]; d2u raw[2][(GM == 1) ? NPR : 1]; };
    auto gather_pair = [&](const i2a j, Gbuf &B) {
        if constexpr (GM == 0) { gather0(j.x, B.w[0]); gather0(j.y, B.w[1]); }
        else { gather1(j.x, B.raw[0]); gather1(j.y, B.raw[1]); }
    };
    auto consume_pair = [&](const i2a j, const d2a (&q)[NQ], Gbuf &B) {
        double q0[9], q1[9];
        expand(q, q0, q1);
        if constexpr (GM == 0) {
            fma_step(q0, B.w[0]);
            row_end(j.x);
            fma_step(q1, B.w[1]);
            row_end(j.y);
        } else {
            double w0[REC];
            transpose1(B.raw[0], w0);
            fma_step(q0, w0);
            row_end(j.x);
            transpose1(B.raw[1], w0);
            fma_step(q1, w0);
            row_end(j.y);
        }
    };

    Gbuf G;
    if constexpr (PIPE == 1) {
        // the block stream (HBM latency) runs one pair ahead of the gathers (L2 latency), the column words two pairs ahead
        d2a qA[NQ], qB[NQ];
        i2a jc = {0, 0}, jn = {0, 0};
        auto body = [&](int p, d2a (&cur)[NQ], d2a (&nxt)[NQ], auto pf) {
            constexpr bool PF = decltype(pf)::value;
            i2a jnn = jn;
            if constexpr (PF) {
                jnn = load_cols((p + 2 < np) ? p + 2 : p + 1);   // clamped, unconditional
                load_blk(p + 1, nxt);
            }
            gather_pair(jc, G);
            __builtin_amdgcn_sched_barrier(0);
            consume_pair(jc, cur, G);
            if constexpr (PF) asm volatile("" : "+v"(jnn.x), "+v"(jnn.y));   // keeps the index prefetch in this iteration
            jc = jn; jn = jnn;
        };
        if (np > 0) {
            jc = load_cols(0);
            jn = load_cols((np > 1) ? 1 : 0);
            load_blk(0, qA);
            int p = 0;
            for (; p + 2 < np; p += 2) {
                body(p, qA, qB, std::true_type{});
                body(p + 1, qB, qA, std::true_type{});
            }
            if (np - p == 2) {
                body(p, qA, qB, std::true_type{});
                body(p + 1, qB, qA, std::false_type{});
            } else {
                body(p, qA, qB, std::false_type{});
            }
        }
    } else if (np > 0) {
        d2a q[NQ];
        i2a jc = load_cols(0);
        for (int p = 0; p < np; ++p) {
            i2a jn = load_cols((p + 1 < np) ? p + 1 : p);   // clamped, unconditional
            load_blk(p, q);
            gather_pair(jc, G);
            __builtin_amdgcn_sched_barrier(0);   // every load of the pair is in flight before the first FMA
            consume_pair(jc, q, G);
            asm volatile("" : "+v"(jn.x), "+v"(jn.y));   // keeps the prefetch in THIS iteration (see xm_sell.hip)
            jc = jn;
        }
    }
    if (tail) {
        const int jt = __builtin_nontemporal_load(cb + (size_t)np * 128 + lane);
        double qt[9];
        const double *b = bb + (size_t)np * (128 * NQ) + lane;
        if constexpr (CODEC == SELL_CODEC_QUAT) {
            double t4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) t4[e] = __builtin_nontemporal_load(b + e * 64);
            quat_to_block(t4[0], t4[1], t4[2], t4[3], qt);
        } else {
#pragma unroll
            for (int e = 0; e < 9; ++e) qt[e] = __builtin_nontemporal_load(b + e * 64);
        }
        double wt[REC];
        if constexpr (GM == 0) {
            gather0(jt, wt);
        } else {
            d2u rawt[NPR];
            gather1(jt, rawt);
            transpose1(rawt, wt);
        }
        fma_step(qt, wt);
        row_end(jt);
    }
    // runs that did not end with a row end: a lane without any row end is one long first run; otherwise the last run belongs to row zrow
    if (first) {
#pragma unroll
        for (int e = 0; e < NV; ++e) A[e] = acc[e];
    } else if (meta & (1 << 13)) {
        double *dst = RA + ((meta >> 14) & 63) * NV;
#pragma unroll
        for (int e = 0; e < NV; ++e) dst[e] = acc[e];
    }
#pragma unroll
    for (int e = 0; e < NV; ++e) T[lane * NV + e] = A[e];
    wave_lds_sync();
    // row `lane` of the chunk: its slot + the first runs of the lanes it continued into, in block order
    double tot[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) tot[e] = RA[lane * NV + e];
    {
        const int la = meta & 63, nA = (meta >> 6) & 127;
        for (int j = 0; __builtin_amdgcn_ballot_w64(j < nA) != 0; ++j) {
            if (j < nA) {
                const double *src = T + (la + j) * NV;
#pragma unroll
                for (int e = 0; e < NV; ++e) tot[e] += src[e];
            }
        }
    }
    wave_lds_sync();
    // the tile: 3 x O planes of 64 lane values (every store instruction covers 512 contiguous bytes)
    {
        double *tp = m.tiles + (size_t)tile * (64 * NV) + lane;
#pragma unroll
        for (int e = 0; e < NV; ++e) st_agent(tp + e * 64, tot[e]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tile has reached memory before the arrival is counted
    const int t0 = m.tile_ptr[chunk], t1 = m.tile_ptr[chunk + 1];
    int last = 0;
    if (lane == 0) {
        const unsigned int old = __hip_atomic_fetch_add(m.arrived + chunk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (old + 1 == (unsigned int)(t1 - t0));
        if (last) __hip_atomic_store(m.arrived + chunk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
    last = __builtin_amdgcn_readfirstlane(last);
    if (!last) return;

    // ---------------- the chunk is complete: per-camera sum in tile order + fused epilogue, one lane per camera ----------------
    const int cam = chunk * 64 + lane;
    const bool active = cam < a.nloc;
    const int nval = (a.nloc - chunk * 64 < 64) ? (a.nloc - chunk * 64) : 64;
    double h[REC];
#pragma unroll
    for (int e = 0; e < REC; ++e) h[e] = 0.0;
    {
        constexpr int TB = (NV <= 9) ? 4 : 2;   // tiles in flight
        double sum[NV];
#pragma unroll
        for (int e = 0; e < NV; ++e) sum[e] = 0.0;
        for (int t = t0; t < t1; t += TB) {
            double v[TB][NV];
#pragma unroll
            for (int b = 0; b < TB; ++b) {
                const int tt = (t + b < t1) ? t + b : t1 - 1;   // clamped, unconditional loads; the surplus is not added
                const double *tp = m.tiles + (size_t)tt * (64 * NV) + lane;
#pragma unroll
                for (int e = 0; e < NV; ++e) v[b][e] = ld_agent(tp + e * 64);
            }
#pragma unroll
            for (int b = 0; b < TB; ++b) {   // tile order: the sum does not depend on which slice arrived last
                if (t + b < t1) {
#pragma unroll
                    for (int e = 0; e < NV; ++e) sum[e] += v[b][e];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int k = 0; k < O; ++k) h[r * OP + k] = sum[r * O + k];
    }
    const int nel = nval * REC;
    double wl[REC];   // this camera's rows of the product input
    if (m.diag != nullptr || EPI == EPI_GRAD || EPI == EPI_CERT) rec_load<REC>(W + (size_t)(m.row0 + (int64_t)chunk * 64) * REC, nel, T, lane, wl);
    if (m.diag != nullptr) {   // quaternion codec: the diagonal block d * I was left out of the slices
        const double d = active ? m.diag[cam] : 0.0;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int k = 0; k < O; ++k) h[r * OP + k] = fma(d, wl[r * OP + k], h[r * OP + k]);
    }
#pragma unroll
    for (int e = 0; e < REC; ++e) h[e] *= alpha;
    const bool anchor = (a.cam0 + cam) == 0;
    const size_t rbase = (size_t)chunk * 64 * REC;
    double p0 = 0.0, p1 = 0.0, p2 = 0.0;
    if constexpr (EPI == EPI_PLAIN) {
        rec_store<REC>(a.out + rbase, nel, T, lane, h);
    } else if constexpr (EPI == EPI_CERT) {
        // y_i = (Q x)_i + dz_i * x[3i] e_0 - Lam_i x_i      (O == 1)
        double L[9];
        rec_load<9>(a.Lam + (size_t)chunk * 64 * 9, nval * 9, T, lane, L);
        const double dz = active ? a.dz[cam] : 0.0;
        const double x0 = wl[0], x1 = wl[OP], x2 = wl[2 * OP];
#pragma unroll
        for (int r = 0; r < 3; ++r) h[r * OP] -= L[r * 3 + 0] * x0 + L[r * 3 + 1] * x1 + L[r * 3 + 2] * x2;
        h[0] += dz * x0;
        rec_store<REC>(a.out + rbase, nel, T, lane, h);
    } else if constexpr (EPI == EPI_GRAD) {
        // trustregion.h:186-194 (grad), :307-317 (projection), :162-170 (objc) fused; h = 2 C sR rows
        const double s = active ? a.s[cam] : 1.0;
        double R[REC];
        rec_load<REC>(a.R + rbase, nel, T, lane, R);
        rec_store<REC>(a.G + rbase, nel, T, lane, h);
        const double q = s * s - 1.0;
        const double hW = bdot<O>(h, wl), hR = bdot<O>(h, R);
        p0 = 0.5 * hW + (anchor ? 0.0 : a.lam * q * q);
        const double egs = anchor ? 0.0 : hR + 4.0 * a.lam * (q * s);
        double eg[REC];
#pragma unroll
        for (int e = 0; e < REC; ++e) eg[e] = h[e] * s;
        double S0[3][3];
        bsym_abt<O>(R, eg, S0);
        bsub_s_times<O>(eg, S0, R);   // eg is now the Riemannian gradient
        const double rgs = egs * (s * s);
        rec_store<REC>(a.rgR + rbase, nel, T, lane, eg);
        double s9[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) s9[r * 3 + cc] = S0[r][cc];
        rec_store<9>(a.S0 + (size_t)chunk * 64 * 9, nval * 9, T, lane, s9);
        if (active) { a.egs[cam] = egs; a.rgs[cam] = rgs; }
        const double rsds = rgs / s;
        p1 = bdot<O>(eg, eg) + rsds * rsds;
    } else if constexpr (EPI == EPI_HESS) {
        // trustregion.h:227-255 (ehess) + :277-295 (ehess2rhess) fused; h = 2 C (s.*Ru + su.*R) rows
        const double s = active ? a.s[cam] : 1.0;
        const double ps = (active && !anchor) ? a.ps[cam] : 0.0;
        const double egs = active ? a.egs[cam] : 0.0;
        const double rsv = (active && !anchor) ? a.rs[cam] : 0.0;
        double R[REC], P[REC], rh[REC];
        double hRGP;
        {
            double G[REC], tR[REC], tP[REC], tG[REC];
            rec_fetch<REC>(a.R + rbase, nel, lane, tR);
            rec_fetch<REC>(a.pR + rbase, nel, lane, tP);
            rec_fetch<REC>(a.G + rbase, nel, lane, tG);
            rec_unpack<REC>(tR, T, lane, R);
            rec_unpack<REC>(tP, T, lane, P);
            rec_unpack<REC>(tG, T, lane, G);
            hRGP = bdot<O>(h, R) + bdot<O>(G, P);
#pragma unroll
            for (int e = 0; e < REC; ++e) rh[e] = h[e] * s + G[e] * ps;   // hr = CsRu.*s + CsR.*su
        }
        const double hs = anchor ? 0.0 : hRGP + 4.0 * a.lam * ((3.0 * s * s - 1.0) * ps);
        double t9[9], tRr[REC];
        rec_fetch<9>(a.S0 + (size_t)chunk * 64 * 9, nval * 9, lane, t9);
        rec_fetch<REC>(a.rR + rbase, nel, lane, tRr);
        {
            double s9[9], S0[3][3], S1[3][3];
            rec_unpack<9>(t9, T, lane, s9);
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) S0[r][cc] = s9[r * 3 + cc];
            bsub_s_times<O>(rh, S0, P);   // rhr = ehessR - Ru * sym(R' egradR)
            bsym_abt<O>(R, rh, S1);
            bsub_s_times<O>(rh, S1, R);   // rhr -= R * sym(R' rhr)
        }
        const double rhs = anchor ? 0.0 : hs * (s * s) + (ps * s) * egs;
        rec_store<REC>(a.HpR + rbase, nel, T, lane, rh);
        if (active) a.Hps[cam] = rhs;
        if (a.Bout) {   // multi-rank tCG: the image of Hp under (xR, xs) -> s.*xR + xs.*R travels with the partial sums
            double b[REC];
#pragma unroll
            for (int e = 0; e < REC; ++e) b[e] = s * rh[e] + rhs * R[e];
            rec_store<REC>(a.Bout + rbase, nel, T, lane, b);
        }
        p0 = bdot<O>(P, rh) + ps * (rhs / (s * s));
        double Rr[REC];
        rec_unpack<REC>(tRr, T, lane, Rr);
        p1 = bdot<O>(Rr, rh) + rsv * (rhs / (s * s));
        const double hq = rhs / s;
        p2 = bdot<O>(rh, rh) + hq * hq;
    }
    if constexpr (EPI == EPI_GRAD || EPI == EPI_HESS) {
        if (!active) { p0 = 0.0; p1 = 0.0; p2 = 0.0; }
        p0 = wave_sum(p0);
        p1 = wave_sum(p1);
        if (EPI == EPI_HESS) p2 = wave_sum(p2);
        if (lane == 0) {
            a.partials[chunk] = p0;
            a.partials[m.nchunks + chunk] = p1;
            if (EPI == EPI_HESS) a.partials[2 * m.nchunks + chunk] = p2;
        }
    }
}

template <int O, int GM, int PIPE, int CODEC, int EPI>
__global__ __launch_bounds__(256) void qw_sell2_kernel(Sell2Args m, const double *__restrict__ W, double alpha, CamArgs a) {
    qw_sell2_body<O, GM, PIPE, CODEC, EPI>(m, W, alpha, a);
}

static int sell2_pipe_default(int o, int codec) {
    static const int env = [] { const char *e = std::getenv("XM_SELL2_PIPE"); return (e && *e) ? std::atoi(e) : -1; }();
    if (env >= 0) return env ? 1 : 0;
    (void)o; (void)codec;
    return 0;
}

template <int O, int GM, int PIPE, int CODEC>
static void qw_sell2_epi(int epi, const Sell2Args &sa, const double *W, double alpha, const CamArgs &a, int grid, hipStream_t st) {
    const dim3 g(grid), b(256);
    if constexpr (O == 1) {
        switch (epi) {
            case EPI_PLAIN: hipLaunchKernelGGL((qw_sell2_kernel<O, GM, PIPE, CODEC, EPI_PLAIN>), g, b, 0, st, sa, W, alpha, a); break;
            case EPI_CERT: hipLaunchKernelGGL((qw_sell2_kernel<O, GM, PIPE, CODEC, EPI_CERT>), g, b, 0, st, sa, W, alpha, a); break;
            default: throw Error(XM_ERR_ARG, "SELL: o == 1 takes the plain or the certificate epilogue");
        }
    } else {
        switch (epi) {
            case EPI_PLAIN: hipLaunchKernelGGL((qw_sell2_kernel<O, GM, PIPE, CODEC, EPI_PLAIN>), g, b, 0, st, sa, W, alpha, a); break;
            case EPI_GRAD: hipLaunchKernelGGL((qw_sell2_kernel<O, GM, PIPE, CODEC, EPI_GRAD>), g, b, 0, st, sa, W, alpha, a); break;
            case EPI_HESS: hipLaunchKernelGGL((qw_sell2_kernel<O, GM, PIPE, CODEC, EPI_HESS>), g, b, 0, st, sa, W, alpha, a); break;
            default: throw Error(XM_ERR_ARG, "bad epilogue");
        }
    }
}
template <int O>
static void qw_sell2_o(int epi, Sell2Matrix &m, const double *W, double alpha, const CamArgs &a, int gm, int pipe, hipStream_t st) {
    const Sell2Args sa = m.args(O);
    if (m.grid() <= 0) return;
    const bool quat = m.codec() == SELL_CODEC_QUAT;
    if (pipe < 0) pipe = sell2_pipe_default(O, m.codec());
    if constexpr (O == 1) {
        (void)gm; (void)pipe;
        if (quat) qw_sell2_epi<1, 0, 0, SELL_CODEC_QUAT>(epi, sa, W, alpha, a, m.grid(), st);
        else qw_sell2_epi<1, 0, 0, SELL_CODEC_FULL>(epi, sa, W, alpha, a, m.grid(), st);
    } else {
#define XM_S2(GM_, PIPE_)                                                                                  \
    do {                                                                                                   \
        if (quat) qw_sell2_epi<O, GM_, PIPE_, SELL_CODEC_QUAT>(epi, sa, W, alpha, a, m.grid(), st);        \
        else qw_sell2_epi<O, GM_, PIPE_, SELL_CODEC_FULL>(epi, sa, W, alpha, a, m.grid(), st);             \
    } while (0)
        if (gm == 0) { XM_S2(0, 0); }
        else if (pipe == 1) { XM_S2(1, 1); }
        else { XM_S2(1, 0); }
#undef XM_S2
    }
}

void launch_qw_sell2(int o, int epi, Sell2Matrix &m, const double *W, double alpha, const CamArgs &a, int gm, int pipe, hipStream_t st) {
    if (a.nloc <= 0) return;
    if ((int64_t)a.nloc != m.nloc()) throw Error(XM_ERR_ARG, "SELL: the epilogue arguments describe a different number of cameras");
    switch (o) {
        case 1: qw_sell2_o<1>(epi, m, W, alpha, a, 0, 0, st); break;
        case 3: qw_sell2_o<3>(epi, m, W, alpha, a, gm, pipe, st); break;
        case 4: qw_sell2_o<4>(epi, m, W, alpha, a, gm, pipe, st); break;
        case 5: qw_sell2_o<5>(epi, m, W, alpha, a, gm, pipe, st); break;
        default: throw Error(XM_ERR_ARG, "SELL product is instantiated for o = 1, 3, 4, 5");
    }
    check_launch("qw_sell2");
}

}  // namespace xm
