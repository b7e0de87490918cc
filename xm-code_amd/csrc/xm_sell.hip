// xm_sell.hip — large-n block-sparse Q*W: sliced-ELL over per-XCD column slabs (layout and rationale: xm_sell.h).
// Replaces, for view-graph-sparse Q, the product the reference runs as cublasDgemm on a dense matrix (Dense/matmul.h:42-87,
// call sites trustregion.h:165,187,237,553, checkeig.h:182); the fused epilogues are the ones of xm_device.h.
#include "xm_sell.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <type_traits>

#include "xm_device.h"
#include "xm_sell_codec.h"

namespace xm {

// ------------------------------------------------------------------------------------------------------------------
// host: build the layout description
// ------------------------------------------------------------------------------------------------------------------
void sell_build_host(const int64_t *rowptr, const int32_t *colidx, int64_t nloc, int64_t ncols, int S, int lmax, SellHost &out,
                     int64_t diag_row0) {
    if (!(S == 1 || S == 2 || S == 4 || S == 8)) throw Error(XM_ERR_ARG, "SELL: slabs must be 1, 2, 4 or 8");
    if (lmax < 2) throw Error(XM_ERR_ARG, "SELL: lmax must be >= 2");
    if (nloc < 0 || ncols < 1 || !rowptr) throw Error(XM_ERR_ARG, "SELL: bad sizes");
    out = SellHost();
    out.nloc = nloc; out.ncols = ncols; out.S = S; out.lmax = lmax;
    const int64_t b0 = rowptr[0], nb = rowptr[nloc] - b0;
    if (nb < 0) throw Error(XM_ERR_ARG, "BSR3: rowptr is not monotone");
    if (nb > 0 && !colidx) throw Error(XM_ERR_ARG, "BSR3: colidx missing");
    // order[i]: i-th kept block of the matrix with every row in ascending column order (identity when the rows are sorted already and
    // nothing is left out); row r owns order[ro[r] .. ro[r+1])
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
    struct VRow { int64_t start; int32_t len; int32_t slot; };   // start: position in `order`
    std::vector<std::vector<VRow>> per_slab((size_t)S);
    // Partial results are numbered ROW-major (camera r owns the contiguous slots [pptr[r], pptr[r+1]), slab by slab): the per-camera
    // sum reads one contiguous run.  Slab-major numbering (each slab's region written by its own XCDs only) was measured and is
    // slower: 127-145 us against 115-131 us per product at 100k cameras, S = 4.
    out.pptr.assign((size_t)nloc + 1, 0);
    int64_t slot = 0;
    for (int64_t r = 0; r < nloc; ++r) {
        out.pptr[(size_t)r] = slot;
        int64_t q = ro[(size_t)r];
        const int64_t e = ro[(size_t)r + 1];
        while (q < e) {
            const int s = slab_of(colidx[order[(size_t)q]]);
            int64_t q2 = q;
            while (q2 < e && q2 - q < lmax && slab_of(colidx[order[(size_t)q2]]) == s) ++q2;
            if (slot > 2147483000LL) throw Error(XM_ERR_ARG, "SELL: too many partial results");
            per_slab[(size_t)s].push_back(VRow{q, (int32_t)(q2 - q), (int32_t)slot});
            ++slot;
            q = q2;
        }
    }
    out.pptr[(size_t)nloc] = slot;
    out.nparts = slot;
    // A partial result is STORED at slice * 64 + lane: the 64 records of a slice are one contiguous 4.6 KB run, written once, by one
    // wavefront; the per-camera sum looks its records up through ridx.  (Storing at the list position -- the lanes of a slice scatter
    // 72-byte records over the whole array -- cost the main launch 20 instead of 9 us at 100 k cameras, HISTORY.md section 2.5.)
    constexpr bool slice_order = true;
    out.ridx.assign((size_t)std::max<int64_t>(slot, 1), 0);
    // per slab: stable counting sort by length (descending), then slices of 64
    out.slab_start.assign((size_t)S + 1, 0);
    out.slice_off.clear();
    out.slice_off.push_back(0);
    std::vector<VRow> sorted;
    for (int s = 0; s < S; ++s) {
        const std::vector<VRow> &v = per_slab[(size_t)s];
        std::vector<int64_t> cnt((size_t)lmax + 2, 0);
        for (const VRow &x : v) cnt[(size_t)(lmax - x.len) + 1]++;          // bucket 0 = longest
        for (size_t i = 1; i < cnt.size(); ++i) cnt[i] += cnt[i - 1];
        sorted.resize(v.size());
        for (const VRow &x : v) sorted[(size_t)cnt[(size_t)(lmax - x.len)]++] = x;
        out.nvrows += (int64_t)v.size();
        for (size_t i = 0; i < sorted.size(); i += 64) {
            const int w = sorted[i].len;
            const int64_t off = out.slice_off.back();
            out.slice_off.push_back(off + w);
            out.kind.resize((size_t)(off + w));
            out.src.resize((size_t)(off + w) * 64, -1);
            out.pslot.resize(out.pslot.size() + 64, -1);
            for (int k = 0; k < w; ++k) out.kind[(size_t)(off + k)] = (uint8_t)((k < (w & ~1)) ? (k & 1) : 2);
            const size_t cnt_l = std::min<size_t>(64, sorted.size() - i);
            for (size_t l = 0; l < cnt_l; ++l) {
                const VRow &x = sorted[i + l];
                const int64_t store = slice_order ? (int64_t)(out.slice_off.size() - 2) * 64 + (int64_t)l : (int64_t)x.slot;
                out.pslot[out.pslot.size() - 64 + l] = (int32_t)store;
                out.ridx[(size_t)x.slot] = (int32_t)store;
                for (int k = 0; k < x.len; ++k) out.src[(size_t)(off + k) * 64 + l] = order[(size_t)(x.start + k)];
            }
        }
        out.slab_start[(size_t)s + 1] = (int32_t)(out.slice_off.size() - 1);
    }
    out.nslices = (int64_t)out.slice_off.size() - 1;
    out.nsteps = out.slice_off.back();
    {   // column locality of the gather (SellHost::lines_*): every 4th step
        std::vector<int64_t> a72, a120, ac;
        for (int64_t t = 0; t < out.nsteps; t += 4) {
            a72.clear(); a120.clear(); ac.clear();
            for (int l = 0; l < 64; ++l) {
                const int64_t q = out.src[(size_t)t * 64 + l];
                if (q < 0) continue;
                const int64_t c = colidx[q];
                ac.push_back(c);
                a72.push_back((c * 72) >> 7); a72.push_back((c * 72 + 71) >> 7);
                a120.push_back((c * 120) >> 7); a120.push_back((c * 120 + 119) >> 7);
            }
            auto distinct = [](std::vector<int64_t> &v) { std::sort(v.begin(), v.end()); return (int64_t)(std::unique(v.begin(), v.end()) - v.begin()); };
            out.lines_padded += distinct(ac); out.lines_native72 += distinct(a72); out.lines_native120 += distinct(a120);
        }
    }
    out.nstore = slice_order ? out.nslices * 64 : out.nparts;
    out.slice_order = slice_order;
    if (out.nstore > 2147483000LL) throw Error(XM_ERR_ARG, "SELL: too many partial results");
}

// ------------------------------------------------------------------------------------------------------------------
// device: fill the interleaved arrays from the CSR arrays (one thread per (step, lane))
// ------------------------------------------------------------------------------------------------------------------
template <int NQ>
__global__ __launch_bounds__(256) void sell_fill_kernel(int64_t nsteps, const int64_t *__restrict__ src, const uint8_t *__restrict__ kind,
                                                         const int32_t *__restrict__ colidx, const double *__restrict__ blocks, int64_t b0,
                                                         int32_t *__restrict__ cols, double *__restrict__ blk) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= nsteps * 64) return;
    const int64_t g = t >> 6;
    const int lane = (int)(t & 63);
    const int kd = kind[g];
    const int64_t s = src[t];
    const int32_t c = (s < 0) ? 0 : colidx[s - b0];
    double q[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) q[e] = (s < 0) ? 0.0 : blocks[(s - b0) * 9 + e];
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
        cols[g * 64 + lane] = c;
#pragma unroll
        for (int e = 0; e < NQ; ++e) blk[g * U + e * 64 + lane] = v[e];
    } else {
        const int64_t gb = g - kd;   // first unit of the pair
        cols[gb * 64 + lane * 2 + kd] = c;
#pragma unroll
        for (int e = 0; e < NQ; ++e) blk[gb * U + e * 128 + lane * 2 + kd] = v[e];
    }
}
__global__ __launch_bounds__(256) void sell_diag_kernel(int64_t nloc, const int64_t *__restrict__ diag_src, const double *__restrict__ blocks, int64_t b0,
                                                         double *__restrict__ diag) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= nloc) return;
    const int64_t s = diag_src[r];
    diag[r] = (s < 0) ? 0.0 : blocks[(s - b0) * 9];
}

// host check of the view-graph structure the quaternion codec relies on (O(nb))
void check_viewgraph_blocks(const int64_t *rowptr, const int32_t *colidx, const double *blocks, int64_t nloc, int64_t row0) {
    for (int64_t r = 0; r < nloc; ++r)
        for (int64_t q = rowptr[r]; q < rowptr[r + 1]; ++q) {
            const double *b = blocks + q * 9;
            double ss = 0.0;
            for (int e = 0; e < 9; ++e) ss += b[e] * b[e];
            if (!(ss == ss)) throw Error(XM_ERR_ARG, "view-graph codec: NaN block");
            if ((int64_t)colidx[q] == row0 + r) {
                const double d = b[0];
                const double off = std::fabs(b[1]) + std::fabs(b[2]) + std::fabs(b[3]) + std::fabs(b[5]) + std::fabs(b[6]) + std::fabs(b[7]) +
                                   std::fabs(b[4] - d) + std::fabs(b[8] - d);
                if (off > 1e-9 * std::fabs(d)) throw Error(XM_ERR_ARG, "view-graph codec: a diagonal block is not a multiple of the identity");
                continue;
            }
            if (ss == 0.0) continue;
            const double w2 = ss / 3.0;
            double dev = 0.0;
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    const double g = b[3 * i] * b[3 * j] + b[3 * i + 1] * b[3 * j + 1] + b[3 * i + 2] * b[3 * j + 2] - (i == j ? w2 : 0.0);
                    dev += g * g;
                }
            const double det = b[0] * (b[4] * b[8] - b[5] * b[7]) - b[1] * (b[3] * b[8] - b[5] * b[6]) + b[2] * (b[3] * b[7] - b[4] * b[6]);
            if (std::sqrt(dev) > 1e-9 * w2 || !(det < 0.0))   // det(-w M) = -w^3
                throw Error(XM_ERR_ARG, "view-graph codec: an off-diagonal block is not -w * rotation");
        }
}

SellMatrix::SellMatrix(const int64_t *rowptr, const int32_t *colidx, const double *blocks, int64_t nloc, int64_t ncols, int S, int lmax,
                       hipStream_t st, int codec, int64_t row0) {
    if (codec != SELL_CODEC_FULL && codec != SELL_CODEC_QUAT) throw Error(XM_ERR_ARG, "SELL: unknown codec");
    codec_ = codec; row0_ = row0;
    if (codec == SELL_CODEC_QUAT) check_viewgraph_blocks(rowptr, colidx, blocks, nloc, row0);
    SellHost h;
    sell_build_host(rowptr, colidx, nloc, ncols, S, lmax, h, codec == SELL_CODEC_QUAT ? row0 : -1);
    ncols_ = ncols;
    lines72_ = h.lines_native72; lines120_ = h.lines_native120; lines_padded_ = h.lines_padded;
    max_list_ = 0;
    for (int64_t r = 0; r < nloc; ++r) max_list_ = std::max<int64_t>(max_list_, h.pptr[(size_t)r + 1] - h.pptr[(size_t)r]);
    nloc_ = nloc; nparts_ = h.nstore; nsteps_ = h.nsteps; nslices_ = h.nslices; S_ = S;
    coalesced_ = h.slice_order;
    const int64_t b0 = rowptr[0], nb = rowptr[nloc] - b0;
    slice_off_.alloc(h.slice_off.size(), false);
    slab_start_.alloc(h.slab_start.size(), false);
    pslot_.alloc(std::max<size_t>(h.pslot.size(), 1), false);
    pptr_.alloc(h.pptr.size(), false);
    ridx_.alloc(h.ridx.size(), false);
    XM_HIP_CHECK(hipMemcpy(ridx_.p, h.ridx.data(), h.ridx.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    cols_.alloc((size_t)std::max<int64_t>(nsteps_, 1) * 64, false);
    blk_.alloc((size_t)std::max<int64_t>(nsteps_, 1) * 64 * (codec_ == SELL_CODEC_QUAT ? 4 : 9), false);
    if (codec_ == SELL_CODEC_QUAT) {
        diag_.alloc((size_t)std::max<int64_t>(nloc, 1));
        diag_src_.alloc((size_t)std::max<int64_t>(nloc, 1), false);
        if (nloc > 0) XM_HIP_CHECK(hipMemcpy(diag_src_.p, h.diag_src.data(), (size_t)nloc * sizeof(int64_t), hipMemcpyHostToDevice));
    }
    XM_HIP_CHECK(hipMemcpy(slice_off_.p, h.slice_off.data(), h.slice_off.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    XM_HIP_CHECK(hipMemcpy(slab_start_.p, h.slab_start.data(), h.slab_start.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    if (!h.pslot.empty()) XM_HIP_CHECK(hipMemcpy(pslot_.p, h.pslot.data(), h.pslot.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    XM_HIP_CHECK(hipMemcpy(pptr_.p, h.pptr.data(), h.pptr.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    b0_ = b0;
    if (nsteps_ > 0 || (codec_ == SELL_CODEC_QUAT && nb > 0)) {
        DevBuf<int32_t> dci; DevBuf<double> dbl;
        if (nsteps_ > 0) {
            src_.alloc(h.src.size(), false); kind_.alloc(h.kind.size(), false);   // kept: refill() after a device-side update of the values
            XM_HIP_CHECK(hipMemcpy(src_.p, h.src.data(), h.src.size() * sizeof(int64_t), hipMemcpyHostToDevice));
            XM_HIP_CHECK(hipMemcpy(kind_.p, h.kind.data(), h.kind.size(), hipMemcpyHostToDevice));
        }
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

// (re)build the interleaved arrays from block-CSR arrays on the device (colidx / blocks indexed from the first local block)
void SellMatrix::refill(const int32_t *d_colidx, const double *d_blocks, hipStream_t st) {
    if (codec_ == SELL_CODEC_QUAT && nloc_ > 0) {
        hipLaunchKernelGGL(sell_diag_kernel, dim3((unsigned)((nloc_ + 255) / 256)), dim3(256), 0, st, nloc_, diag_src_.p, d_blocks, b0_, diag_.p);
        check_launch("sell_diag");
    }
    if (nsteps_ <= 0) return;
    const int64_t threads = nsteps_ * 64;
    if (codec_ == SELL_CODEC_QUAT)
        hipLaunchKernelGGL(sell_fill_kernel<4>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, nsteps_, src_.p, kind_.p, d_colidx, d_blocks,
                           b0_, cols_.p, blk_.p);
    else
        hipLaunchKernelGGL(sell_fill_kernel<9>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, nsteps_, src_.p, kind_.p, d_colidx, d_blocks,
                           b0_, cols_.p, blk_.p);
    check_launch("sell_fill");
}
int64_t SellMatrix::stream_bytes() const { return nsteps_ * 64 * (4 + 8 * (int64_t)(codec_ == SELL_CODEC_QUAT ? 4 : 9)) + (codec_ == SELL_CODEC_QUAT ? 8 * nloc_ : 0); }

SellArgs SellMatrix::args() const {
    SellArgs a;
    a.slice_off = slice_off_.p; a.slab_start = slab_start_.p; a.cols = cols_.p; a.blk = blk_.p; a.pslot = pslot_.p; a.pptr = pptr_.p; a.ridx = ridx_.p;
    a.S = S_;
    a.diag = (codec_ == SELL_CODEC_QUAT) ? diag_.p : nullptr;
    a.row0 = row0_;
    a.wstride = 0;   // set per rank by the launcher (wstride(o))
    a.coalesced_store = coalesced_ ? 1 : 0;
    a.nt_off = 0;      // set per product by the launcher (nt_off(o, nloc))
    return a;
}
// Stream offset (in step units of 64 blocks) from which the main launch reads non-temporally.  Inside a solve the Infinity Cache is shared with
// what an iteration moves besides the matrix -- the partial results (S records of 16 doubles per camera), about ten vectors of 3 x pitch doubles
// per camera and the padded copy of W -- so the prefix of the stream that can stay from product to product is the cache's size minus that;
// the rest is a pure stream (xm_bench_dense_policy overrides: 0 all cacheable, 1 all non-temporal, >= 2 a prefix of that many MB).
int64_t SellMatrix::nt_off(int o, int64_t nloc) const {
    const int64_t unit = 64 * (4 + 8 * (int64_t)(codec_ == SELL_CODEC_QUAT ? 4 : 9));
    const int64_t vec = nloc * ((int64_t)S_ * 128 + 10 * 24 * (int64_t)pitch_of(o) + 128);
    const int64_t prefix = sell_resident_bytes(stream_bytes(), vec);
    return prefix >= stream_bytes() ? INT64_MAX : prefix / unit;
}

// Record pitch of W as the kernels see it: the native 3 x pitch_of(o) doubles, or 16 when the caller hands over a padded copy
// (launch_qw_sell: Wpad16).  Round 3 tried a repacking launch in front of every product (XM_SELL_WSTRIDE=16) and saw no gain: the 7.5 us of
// that launch hid the 9-10 us the main launch gains (profiles/r04_trace_sell_pitch16.txt); the copy now comes from the kernels that
// write W anyway.
int SellMatrix::wstride(int o) const { return 3 * pitch_of(o); }

double *SellMatrix::parts(int o) {
    if (o > parts_o_) {
        parts_.alloc((size_t)std::max<int64_t>(nparts_, 1) * 3 * (size_t)o, false);
        parts_o_ = o;
    }
    return parts_.p;
}

// ------------------------------------------------------------------------------------------------------------------
// device: the product.  One wavefront per slice, one lane per virtual row.
// ------------------------------------------------------------------------------------------------------------------
typedef double d2a __attribute__((ext_vector_type(2)));               // 16-byte aligned pair
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));   // pair at 8-byte alignment (records of W)
typedef int i2a __attribute__((ext_vector_type(2)));

// What bounds the gather (profiles/r04_pmc_sell_diag2_memory_path.json, r04_pmc_sell_gather_accesses.txt): the vector L1 waits for L2 data
// in 61 % of its active cycles with on average 74 lines outstanding per CU -- 1.54 M stream lines at ~1 380 cycles (HBM) hold 63 % of
// those slots, 7.1 M gathered lines at ~173 cycles (L2 hits) the rest; a kernel that only LOADS takes the same 81-82 us
// (r04_kbench_sell_loads_only.txt).  The duration follows the LINES requested times their latency, so what shortens it is fewer lines per
// record: W at a 128-byte record pitch (Wpad16 of launch_qw_sell) -- 5.09 M instead of 7.1 M gathered lines, main launch 72.8 -> 63.6 us.
// Sector-window gathers (through LDS-DMA, or into registers), 8-byte element fetches, cache-policy bits on the gathered lines and a
// one-launch chunk-tiled layout were all measured slower or equal and are gone from the sources (HISTORY.md, "Round 4").
template <int O, int GM, int NQ = 9>
struct SellBuf {   // one pipeline stage: two steps of blocks and the two gathered records of W
    static constexpr int OP = pitch_of(O), REC = 3 * OP, NPR = (REC + 1) / 2;
    d2a q[NQ];
    double w[2][(GM == 0) ? REC : 1];
    d2u raw[2][(GM == 1) ? NPR : 1];
};

// the stream's load policy is a compile-time property of the kernel (a hint picked by a branch next to a load is merged into one plain load):
// NT = non-temporal -- a stream beyond the Infinity Cache; otherwise the default policy: the stream is found in the Infinity Cache / the L2s by
// the next product (100 k cameras, quaternion codec, 204 MB: 82.7 -> 65.8 us; 60 k cameras, full blocks, 250 MB: 68.3 -> 55.0; 100 k, full
// blocks, 402 MB: 110.1 non-temporal against 115.4; profiles/r06_kbench_sell_policy.txt).  The launcher applies the dense kernel's size rule.
template <bool NT, class T>
__device__ __forceinline__ T sell_ld(const T *p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <int O, int GM, int PIPE = 0, int CODEC = 0, bool NT = true>   // GM 0: a record of W per lane | 1: records fetched element-per-lane, transposed through LDS.  PIPE 1: block loads run one pair ahead.  CODEC: SELL_CODEC_*
__device__ __forceinline__ void qw_sell_body(const SellArgs &m, const double *__restrict__ W, const TcgScal *__restrict__ scal,
                                             double *__restrict__ parts, double *lds, int c, int64_t off) {
    constexpr int OP = pitch_of(O), REC = 3 * OP, NPR = (REC + 1) / 2, RECP = (REC + 1) & ~1;
    constexpr int NQ = (CODEC == SELL_CODEC_QUAT) ? 4 : 9;   // doubles per stored block
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int w = (int)(m.slice_off[c + 1] - off);
    if (scal != nullptr) {   // the tCG's status word (an L2 miss: written by the previous cg_step) is waited for only after the slice look-ups
        if (scal->status != 0) return;   // have been requested
    }
    const int np = w >> 1;
    const bool tail = (w & 1) != 0;
    const int32_t *cb = m.cols + off * 64;
    const double *bb = m.blk + off * (64 * NQ);
    double *L = lds + ((GM != 0) ? wave * 64 * RECP : 0);

    double acc[3][O];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < O; ++k) acc[r][k] = 0.0;

    auto load_cols = [&](int p) -> i2a { return sell_ld<NT>(reinterpret_cast<const i2a *>(cb) + (size_t)p * 64 + lane); };
    auto load_blk = [&](int p, d2a (&q)[NQ]) {
        const d2a *b = reinterpret_cast<const d2a *>(bb) + (size_t)p * (64 * NQ) + lane;
#pragma unroll
        for (int e = 0; e < NQ; ++e) q[e] = sell_ld<NT>(b + e * 64);
    };
    // stored planes of one pair -> the two 3x3 blocks (codec 0: the planes ARE the blocks; quaternion codec: -R(q), 23 flops each)
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
    auto gather0 = [&](int j, double (&wv)[REC]) {
        const double *wp = W + (size_t)j * m.wstride;
#pragma unroll
        for (int i = 0; i < REC / 2; ++i) {
            const d2u t = *reinterpret_cast<const d2u *>(wp + 2 * i);
            wv[2 * i] = t.x; wv[2 * i + 1] = t.y;
        }
        if (REC & 1) wv[REC - 1] = wp[REC - 1];
    };
    // GM 1: the 64 records of a step are fetched element-per-lane (one load instruction covers 1 KiB of the concatenated
    // records, i.e. ~13 records instead of 64) and turned back into lane-per-record through LDS
    auto gather1 = [&](int j, d2u (&raw)[NPR]) {
#pragma unroll
        for (int i = 0; i < NPR; ++i) {
            const int g = lane + 64 * i;
            const int rec = g / NPR, part = g - rec * NPR;
            const int start = (2 * part < REC - 2) ? 2 * part : REC - 2;
            const int jr = __shfl(j, rec, 64);
            const unsigned boff = ((unsigned)jr * (unsigned)m.wstride + (unsigned)start) * 8u;   // 32-bit byte offset + uniform base (W < 4 GB: checked on the host)
            raw[i] = *reinterpret_cast<const d2u *>(reinterpret_cast<const char *>(W) + boff);
        }
    };
    auto transpose1 = [&](const d2u (&raw)[NPR], double *Ls, double (&wv)[REC]) {
#pragma unroll
        for (int i = 0; i < NPR; ++i) {
            const int g = lane + 64 * i;
            const int rec = g / NPR, part = g - rec * NPR;
            if ((REC & 1) && part == NPR - 1) Ls[rec * RECP + REC - 1] = raw[i].y;
            else *reinterpret_cast<d2a *>(Ls + rec * RECP + 2 * part) = d2a{raw[i].x, raw[i].y};
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < RECP / 2; ++i) {
            const d2a t = *reinterpret_cast<const d2a *>(Ls + lane * RECP + 2 * i);
            if (2 * i < REC) wv[2 * i] = t.x;
            if (2 * i + 1 < REC) wv[2 * i + 1] = t.y;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    auto fma_step = [&](const double (&q)[9], const double (&wv)[REC]) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int k = 0; k < O; ++k)
                acc[r][k] = fma(q[3 * r + 2], wv[2 * OP + k], fma(q[3 * r + 1], wv[OP + k], fma(q[3 * r], wv[k], acc[r][k])));
    };
    auto gather_pair = [&](const i2a j, SellBuf<O, GM, NQ> &B) {
        if constexpr (GM == 0) { gather0(j.x, B.w[0]); gather0(j.y, B.w[1]); }
        else { gather1(j.x, B.raw[0]); gather1(j.y, B.raw[1]); }
    };

    // One pair of steps per iteration, single-buffered: the memory-level parallelism comes from the resident wavefronts (each has
    // ~20 KB of loads in flight), not from a per-wave software pipeline, which would double the register footprint and halve
    // the occupancy.  Only the column indices run one pair ahead (they head the dependent chain index -> gathered record).
    SellBuf<O, GM, NQ> A;
    if constexpr (PIPE == 1) {
        // PIPE 1: the block stream (HBM latency) runs one pair ahead of the gathers (L2 latency), the column indices two pairs
        // ahead; only the blocks are double-buffered (the gathered records would cost another 36-40 registers).
        d2a qA[NQ], qB[NQ];
        i2a jc = {0, 0}, jn = {0, 0};
        auto body = [&](int p, d2a (&cur)[NQ], d2a (&nxt)[NQ], auto pf) {
            constexpr bool PF = decltype(pf)::value;
            i2a jnn = jn;
            if constexpr (PF) {
                jnn = load_cols((p + 2 < np) ? p + 2 : p + 1);   // clamped, unconditional
                load_blk(p + 1, nxt);
            }
            gather_pair(jc, A);
            __builtin_amdgcn_sched_barrier(0);
            double q0[9], q1[9];
            expand(cur, q0, q1);
            if constexpr (GM == 0) {
                fma_step(q0, A.w[0]);
                fma_step(q1, A.w[1]);
            } else {
                double w0[REC];
                transpose1(A.raw[0], L, w0);
                fma_step(q0, w0);
                transpose1(A.raw[1], L, w0);
                fma_step(q1, w0);
            }
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
    } else
    if (np > 0) {
        i2a jc = load_cols(0);
        for (int p = 0; p < np; ++p) {
            i2a jn = load_cols((p + 1 < np) ? p + 1 : p);   // clamped, unconditional
            load_blk(p, A.q);
            gather_pair(jc, A);
            __builtin_amdgcn_sched_barrier(0);   // every load of the pair is in flight before the first FMA (the scheduler would
                                                 // otherwise trickle them to save registers: 4-5 dependent round trips per pair)
            double q0[9], q1[9];
            expand(A.q, q0, q1);
            if constexpr (GM == 0) {
                fma_step(q0, A.w[0]);
                fma_step(q1, A.w[1]);
            } else {
                double w0[REC];
                transpose1(A.raw[0], L, w0);
                fma_step(q0, w0);
                transpose1(A.raw[1], L, w0);
                fma_step(q1, w0);
            }
            // keep the prefetch in THIS iteration: without a use here the compiler moves the load across the back edge to the top of
            // the next iteration, where it heads the dependent chain again (seen in the ISA)
            asm volatile("" : "+v"(jn.x), "+v"(jn.y));
            jc = jn;
        }
    }
    if (tail) {
        const int jt = sell_ld<NT>(cb + (size_t)np * 128 + lane);
        double qt[9];
        const double *b = bb + (size_t)np * (128 * NQ) + lane;
        if constexpr (CODEC == SELL_CODEC_QUAT) {
            double t4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) t4[e] = sell_ld<NT>(b + e * 64);
            quat_to_block(t4[0], t4[1], t4[2], t4[3], qt);
        } else {
#pragma unroll
            for (int e = 0; e < 9; ++e) qt[e] = sell_ld<NT>(b + e * 64);
        }
        if constexpr (GM == 0) {
            double wt[REC];
            gather0(jt, wt);
            fma_step(qt, wt);
        } else {
            d2u rawt[NPR];
            double w0[REC];
            gather1(jt, rawt);
            transpose1(rawt, L, w0);
            fma_step(qt, w0);
        }
    }
    if constexpr (GM != 0) {
        if (m.coalesced_store) {
            // the 64 records of the slice are one contiguous run of 64 * 3 * O doubles (slot = slice * 64 + lane): transposed through
            // LDS and written with lane-consecutive 16-byte stores (5 fully coalesced instructions at o = 3) instead of 3 * O
            // 8-byte stores per lane at a 72-byte stride (each touching 36 cache lines)
            constexpr int NV = 3 * O, TOT2 = 64 * NV / 2;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int k = 0; k < O; ++k) L[lane * NV + r * O + k] = acc[r][k];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            d2a *o2 = reinterpret_cast<d2a *>(parts + (size_t)c * 64 * NV);
            const d2a *l2 = reinterpret_cast<const d2a *>(L);
#pragma unroll
            for (int i = 0; i < (TOT2 + 63) / 64; ++i) {
                const int idx = i * 64 + lane;
                if (idx < TOT2) o2[idx] = l2[idx];
            }
            return;
        }
    }
    const int slot = m.pslot[(size_t)c * 64 + lane];
    if (slot >= 0) {
        double *o = parts + (size_t)slot * 3 * O;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int k = 0; k < O; ++k) o[r * O + k] = acc[r][k];
    }
}

// the slice of this wavefront, and which copy of the body streams it: slices whose stream offset lies below SellArgs.nt_off are read with the
// default cache policy (a prefix of the stream that the Infinity Cache keeps from product to product), the rest non-temporally
template <int O, int GM, int PIPE, int CODEC>
__device__ __forceinline__ void qw_sell_entry(const SellArgs &m, const double *__restrict__ W, const TcgScal *__restrict__ scal, double *__restrict__ parts) {
    constexpr int RECP = (3 * pitch_of(O) + 1) & ~1;
    // ONE transposition buffer per wavefront (the two steps of a pair go through it one after the other; transpose1 ends with a wavefront
    // fence): 20 KB per workgroup at o = 3, 32 KB at o = 4 / 5 -- with one buffer per step (40 / 64 KB) the LDS, not the registers, capped
    // the resident workgroups per CU (o = 4 / 5: two)
    __shared__ __attribute__((aligned(16))) double lds[(GM != 0) ? 4 * 64 * RECP : 2];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int per = 8 / m.S;
    const int x = blockIdx.x & 7, bi = blockIdx.x >> 3;
    const int slab = x / per, sub = x - slab * per;
    const int c = m.slab_start[slab] + (bi * per + sub) * 4 + wave;
    if (c >= m.slab_start[slab + 1]) return;   // wave-uniform
    const int64_t off = m.slice_off[c];
    if (off >= m.nt_off) qw_sell_body<O, GM, PIPE, CODEC, true>(m, W, scal, parts, lds, c, off);
    else qw_sell_body<O, GM, PIPE, CODEC, false>(m, W, scal, parts, lds, c, off);
}
template <int O, int GM, int PIPE = 0, int CODEC = 0>
__global__ __launch_bounds__(256) void qw_sell_kernel(SellArgs m, const double *__restrict__ W, const TcgScal *__restrict__ scal,
                                                       double *__restrict__ parts) {
    qw_sell_entry<O, GM, PIPE, CODEC>(m, W, scal, parts);
}
// quaternion codec at o = 3 compiled for FOUR wavefronts per SIMD (the blocks of a pair are 16 registers instead of 36), no software pipeline:
// 81.9 us against 82.4 (two wavefronts, no pipeline) / 84.2 (block loads one pair ahead) at 100 k cameras (profiles/r03_kbench_sell.txt)
template <int GM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void qw_sell_kernel_q_occ4(SellArgs m, const double *__restrict__ W,
                                                                                                      const TcgScal *__restrict__ scal,
                                                                                                      double *__restrict__ parts) {
    qw_sell_entry<3, GM, 0, SELL_CODEC_QUAT>(m, W, scal, parts);
}

// per camera: partial results added in list order (fixed -> bit-reproducible), then the common tail of the Q*W kernels
// (column-per-lane epilogue, per-workgroup partial sums).  A camera normally has S partial results and o <= 4 columns, so a QUAD of
// lanes per camera is enough (GW = 4: 64 cameras per workgroup, two DPP steps per reduction instead of four, 4x fewer wavefronts
// than the 16-lane groups of qw_bsr3_kernel, which o = 5 keeps).  A matrix with hub cameras (a row cut into hundreds of virtual
// rows: four lanes would walk ~100 dependent loads each) takes the 16-lane groups too: measured 152.8 -> 119 us on the
// 100 k-camera graph with 50 hubs.
int SellMatrix::reduce_gw(int o) const { return (o <= 4 && max_list_ <= 32) ? 4 : 16; }
int SellMatrix::reduce_grid(int o, int nloc) const { const int per = 256 / reduce_gw(o); return (nloc + per - 1) / per; }

template <int O, int EPI, int GW>
__global__ __launch_bounds__(256) void sell_reduce_kernel(const int64_t *__restrict__ pptr, const int32_t *__restrict__ ridx, const double *__restrict__ parts,
                                                           double alpha, CamArgs a, const double *__restrict__ diag, const double *__restrict__ W,
                                                           int64_t row0, int wstride) {
    constexpr int NSLOT = 256 / GW;
    __shared__ double red[NSLOT][3];
    const int gl = threadIdx.x & (GW - 1), slot = threadIdx.x / GW;
    const int cam = blockIdx.x * NSLOT + slot;
    const bool active = cam < a.nloc;
    EpiOps eops;
    epi_prefetch<O, EPI>(eops, active ? cam : 0, gl, active, a);
    // list bounds and epilogue operands are requested BEFORE the tCG's status word is looked at (all of them were written by earlier
    // launches on other XCDs): status -> bounds -> indices -> partial sums was four round trips in a row, now three
    const int64_t p0 = active ? pptr[cam] : 0, p1 = active ? pptr[cam + 1] : 0;
    if (EPI == EPI_HESS) {
        if (a.scal->status != 0) return;
    }
    double acc[3][O];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < O; ++k) acc[r][k] = 0.0;
    if (active) {
        for (int64_t p = p0 + gl; p < p1; p += GW) {   // fixed order: lane gl takes list entries gl, gl + GW, ...
            const double *v = parts + (size_t)ridx[p] * 3 * O;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int k = 0; k < O; ++k) acc[r][k] += v[r * O + k];
        }
        // quaternion codec: the diagonal block d * I was left out of the slices.  Inside the single-rank tCG (Hessian epilogue, no exchange
        // image) the term is added after the group sum from the epilogue's own operands (W_i = s_i p_i + ps_i R_i: no read of W, 7.2 MB per
        // product at 100 k cameras); everywhere else the last lane (it has the fewest list entries) reads the camera's record of W
        if (diag != nullptr && !(EPI == EPI_HESS && a.Bout == nullptr) && gl == GW - 1) {
            constexpr int OPW = pitch_of(O);
            const double d = diag[cam];
            const double *w = W + (size_t)(row0 + cam) * wstride;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int k = 0; k < O; ++k) acc[r][k] = fma(d, w[r * OPW + k], acc[r][k]);
        }
    }
    if constexpr (EPI == EPI_HESS) {
        if (diag != nullptr && a.Bout == nullptr) {
            Col3 h;
            h.v[0] = h.v[1] = h.v[2] = 0.0;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int k = 0; k < O; ++k) {
                    const double t = group_sum<GW>(acc[r][k]);
                    if (gl == k) h.v[r] = t;
                }
            if (active && gl < O) {
                const double d = diag[cam];
#pragma unroll
                for (int r = 0; r < 3; ++r) h.v[r] = alpha * fma(d, eops.s * eops.P.v[r] + eops.ps * eops.R.v[r], h.v[r]);
            }
            qw_tail<O, EPI, GW, NSLOT>(cam, gl, slot, active, h, a, eops, red);
            return;
        }
    }
    qw_finish<O, EPI, GW, NSLOT>(cam, gl, slot, active, acc, alpha, a, eops, red);
}

bool sell_supports(int o) { return o == 1 || (o >= 3 && o <= 5); }
void sell_quat_roundtrip(const double block[9], double quat[4], double rebuilt[9]) {
    double b[9], q[4], r[9];
    for (int e = 0; e < 9; ++e) b[e] = block[e];
    block_to_quat(b, q);
    quat_to_block(q[0], q[1], q[2], q[3], r);
    for (int e = 0; e < 4; ++e) quat[e] = q[e];
    for (int e = 0; e < 9; ++e) rebuilt[e] = r[e];
}

template <int O>
static void qw_sell_o(int epi, SellMatrix &m, const double *W, double alpha, const CamArgs &a, int gm, hipStream_t st, const double *Wpad16) {
    const TcgScal *sc = (epi == EPI_HESS) ? a.scal : (const TcgScal *)nullptr;
    double *parts = m.parts(O);
    SellArgs sa = m.args();
    sa.wstride = m.wstride(O);
    const bool quat = m.codec() == SELL_CODEC_QUAT;
    const bool padded = Wpad16 != nullptr && O >= 3 && 3 * pitch_of(O) <= 16;   // the caller keeps a copy of W at the 128-byte record pitch
    const double *Wn = W;            // the second launch walks the cameras in order: the native pitch is the denser read there
    const int wn = sa.wstride;
    if (padded) { sa.wstride = 16; W = Wpad16; }
    if (m.grid() > 0) {
        const dim3 g(m.grid()), b(256);
        if (gm != 0 && (uint64_t)m.ncols() * (uint64_t)sa.wstride * 8u >= (1ull << 32)) gm = 0;   // mode 1 addresses W with 32-bit byte offsets
        // Kernel variant by codec and rank (measured at 100 k cameras, profiles/r03_kbench_sell.txt).  Full blocks: the block loads run one
        // pair ahead at o = 3 (worth 2-3 us); beyond that the second block buffer costs the occupancy (o = 5: 256 VGPRs).  Quaternion
        // codec: four wavefronts per SIMD without software pipeline at o = 3, the plain body at o = 4, 5.
        sa.nt_off = m.nt_off(O, a.nloc);
        if (quat) {
            if constexpr (O == 3) {
                if (gm == 1) hipLaunchKernelGGL((qw_sell_kernel_q_occ4<1>), g, b, 0, st, sa, W, sc, parts);
                else hipLaunchKernelGGL((qw_sell_kernel_q_occ4<0>), g, b, 0, st, sa, W, sc, parts);
            } else {
                if (gm == 1) hipLaunchKernelGGL((qw_sell_kernel<O, 1, 0, SELL_CODEC_QUAT>), g, b, 0, st, sa, W, sc, parts);
                else hipLaunchKernelGGL((qw_sell_kernel<O, 0, 0, SELL_CODEC_QUAT>), g, b, 0, st, sa, W, sc, parts);
            }
        } else {
            constexpr int PIPE = (O == 3) ? 1 : 0;
            if (gm == 1) hipLaunchKernelGGL((qw_sell_kernel<O, 1, PIPE>), g, b, 0, st, sa, W, sc, parts);
            else hipLaunchKernelGGL((qw_sell_kernel<O, 0, PIPE>), g, b, 0, st, sa, W, sc, parts);
        }
    }
    const dim3 g(m.reduce_grid(O, a.nloc)), b(256);
#define XM_SELL_REDUCE(GW_)                                                                                                                          \
    switch (epi) {                                                                                                                                  \
        case EPI_PLAIN: hipLaunchKernelGGL((sell_reduce_kernel<O, EPI_PLAIN, GW_>), g, b, 0, st, sa.pptr, sa.ridx, parts, alpha, a, sa.diag, Wn, sa.row0, wn); break; \
        case EPI_GRAD: hipLaunchKernelGGL((sell_reduce_kernel<O, EPI_GRAD, GW_>), g, b, 0, st, sa.pptr, sa.ridx, parts, alpha, a, sa.diag, Wn, sa.row0, wn); break;   \
        case EPI_HESS: hipLaunchKernelGGL((sell_reduce_kernel<O, EPI_HESS, GW_>), g, b, 0, st, sa.pptr, sa.ridx, parts, alpha, a, sa.diag, Wn, sa.row0, wn); break;   \
        default: throw Error(XM_ERR_ARG, "bad epilogue");                                                                                           \
    }
    if (m.reduce_gw(O) == 4) { XM_SELL_REDUCE(4) } else { XM_SELL_REDUCE(16) }
#undef XM_SELL_REDUCE
}

void launch_qw_sell(int o, int epi, SellMatrix &m, const double *W, double alpha, const CamArgs &a, int gm, hipStream_t st, const double *Wpad16) {
    if (a.nloc <= 0) return;
    if (epi == EPI_CERT) {
        if (o != 1) throw Error(XM_ERR_ARG, "certificate operator needs o == 1");
        double *parts = m.parts(1);
        SellArgs sa = m.args();
        sa.wstride = 3;
        if (m.grid() > 0) {
            sa.nt_off = m.nt_off(1, a.nloc);
            if (m.codec() == SELL_CODEC_QUAT) hipLaunchKernelGGL((qw_sell_kernel<1, 0, 0, SELL_CODEC_QUAT>), dim3(m.grid()), dim3(256), 0, st, sa, W, (const TcgScal *)nullptr, parts);
            else hipLaunchKernelGGL((qw_sell_kernel<1, 0>), dim3(m.grid()), dim3(256), 0, st, sa, W, (const TcgScal *)nullptr, parts);
        }
        if (m.reduce_gw(1) == 4) hipLaunchKernelGGL((sell_reduce_kernel<1, EPI_CERT, 4>), dim3(m.reduce_grid(1, a.nloc)), dim3(256), 0, st, sa.pptr, sa.ridx, parts, alpha, a, sa.diag, W, sa.row0, sa.wstride);
        else hipLaunchKernelGGL((sell_reduce_kernel<1, EPI_CERT, 16>), dim3(m.reduce_grid(1, a.nloc)), dim3(256), 0, st, sa.pptr, sa.ridx, parts, alpha, a, sa.diag, W, sa.row0, sa.wstride);
    } else {
        switch (o) {
            case 1: qw_sell_o<1>(epi, m, W, alpha, a, 0, st, nullptr); break;
            case 3: qw_sell_o<3>(epi, m, W, alpha, a, gm, st, Wpad16); break;
            case 4: qw_sell_o<4>(epi, m, W, alpha, a, gm, st, Wpad16); break;
            case 5: qw_sell_o<5>(epi, m, W, alpha, a, gm, st, Wpad16); break;
            default: throw Error(XM_ERR_ARG, "SELL product is instantiated for o = 1, 3, 4, 5");
        }
    }
    check_launch("qw_sell");
}

}  // namespace xm
