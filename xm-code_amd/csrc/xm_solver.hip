// xm_solver.hip — host driver: Riemannian staircase (XM_main.cu:180-310), RTR-tCG (trustregion.h:77-724) and the
// dual certificate (checkeig.h:42-368) on top of the kernels in xm_kernels.hip.
//
// Design notes (MI355X-first, not a translation of the reference's call sequence):
//  * one Q pass per tCG iteration and ONE per outer iteration (the reference spends three: cost, gradient and a
//    redundant 2*C*sR, trustregion.h:422/467/553): cost, gradient, projection and <g,g> all come out of the
//    gradient epilogue of the candidate point, which becomes the next iterate's state when the step is accepted.
//  * the inner loop never synchronises: alpha/beta/tau and the exit tests live in a device-resident scalar block;
//    the host enqueues iterations a few ahead and watches a host-mapped progress word.
//  * all per-camera state is row-major with an odd pitch (xm_common.h) so no transposes exist (the reference does 4
//    per inner iteration, Dense/transpose.h:7-22).
#include "xm_solver.h"
#include "xm_sell.h"
#include "xm_symw.h"
#include "xm_schur.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace xm {

using clk = std::chrono::steady_clock;
static double secs_since(clk::time_point t0) { return std::chrono::duration<double>(clk::now() - t0).count(); }

void Context::log(const char *fmt, ...) const {
    if (!verbose_ || (comm_ && comm_->rank != 0)) return;
    va_list ap;
    va_start(ap, fmt);
    vprintf(fmt, ap);
    va_end(ap);
    fflush(stdout);
}

// Host <-> device transfers of the solve path go through the context's PINNED staging buffer: a copy from / to pageable memory makes
// the runtime wait for the stream while it holds its staging lock, and with several ranks in one process (peer communicators) a
// rank that waits like that for a peer's push can keep exactly that peer from enqueuing it (seen on the virtual-device runs: both
// ranks inside download_point, one in a pageable copy behind its wait kernel, the other on the lock).  Pinned copies are plain
// enqueues; the only blocking call is hipStreamSynchronize on the context's own stream.
void Context::to_host(void *dst, const void *src_dev, size_t bytes) {
    char *d = static_cast<char *>(dst);
    const char *s = static_cast<const char *>(src_dev);
    const size_t cap = hpin_count_ * sizeof(double);
    if (cap == 0) throw Error(XM_ERR_HIP, "staging buffer missing");
    for (size_t off = 0; off < bytes; off += cap) {
        const size_t n = std::min(cap, bytes - off);
        // Peer communicators: by KERNEL into the host-mapped staging buffer.  The copy engine (or, with HSA_ENABLE_SDMA=0, the runtime's
        // shared blit queue) is ONE in-order queue per device: with several ranks of a process on one device, rank A's copy that waits
        // there for A's wait kernel blocks rank B's copy, and B's push behind it is what A's wait kernel is waiting for (8 virtual ranks
        // stalled exactly so at the all-gather after the certificate).  A kernel on the rank's own stream shares nothing.
        if (comm_->peer() && n % 4 == 0) launch_copy_words(hpin_dev_, s + off, n, st_);
        else XM_HIP_CHECK(hipMemcpyAsync(hpin_, s + off, n, hipMemcpyDeviceToHost, st_));
        XM_HIP_CHECK(hipStreamSynchronize(st_));
        std::memcpy(d + off, hpin_, n);
    }
}
void Context::to_dev(void *dst_dev, const void *src, size_t bytes) {
    char *d = static_cast<char *>(dst_dev);
    const char *s = static_cast<const char *>(src);
    const size_t cap = hpin_count_ * sizeof(double);
    if (cap == 0) throw Error(XM_ERR_HIP, "staging buffer missing");
    for (size_t off = 0; off < bytes; off += cap) {
        const size_t n = std::min(cap, bytes - off);
        std::memcpy(hpin_, s + off, n);
        if (comm_->peer() && n % 4 == 0) launch_copy_words(d + off, hpin_dev_, n, st_);
        else XM_HIP_CHECK(hipMemcpyAsync(d + off, hpin_, n, hipMemcpyHostToDevice, st_));
        XM_HIP_CHECK(hipStreamSynchronize(st_));   // the staging buffer is reused by the next transfer
    }
}
void Context::ensure_pinned(size_t doubles) {
    if (doubles <= hpin_count_) return;
    if (hpin_) (void)hipHostFree(hpin_);
    hpin_ = nullptr; hpin_dev_ = nullptr; hpin_count_ = 0;
    XM_HIP_CHECK(hipHostMalloc((void **)&hpin_, doubles * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
    XM_HIP_CHECK(hipHostGetDevicePointer((void **)&hpin_dev_, hpin_, 0));
    hpin_count_ = doubles;
}

// ------------------------------------------------------------------------------------------------------------------
// construction: lay Q out on the device
// ------------------------------------------------------------------------------------------------------------------
Context::Context(const xm_problem_t &prob, std::shared_ptr<Comm> comm) {
    comm_ = comm ? std::move(comm) : default_comm();
    cfg_ = Settings::resolve(prob.tuning);
    comm_->set_exchange_fence(!cfg_.exchange_lite);   // unconditionally: the communicator may be shared, the setting must not stick from an earlier context
    try {
        init(prob);
    } catch (...) {   // a later allocation failed (e.g. the 13.5 GB slab): release what the destructor would have released
        release_raw();
        throw;
    }
}

void Context::release_raw() {
    for (auto &e : ev_pool_) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    ev_pool_.clear();
    if (ownQ_ && dQ_) (void)hipFree(dQ_);
    dQ_ = nullptr;
    if (hstat_) (void)hipHostFree(hstat_);
    hstat_ = nullptr;
    if (hpin_) (void)hipHostFree(hpin_);
    hpin_ = nullptr;
    if (ev_w_) (void)hipEventDestroy(ev_w_);
    if (ev_p_) (void)hipEventDestroy(ev_p_);
    ev_w_ = ev_p_ = nullptr;
    if (st2_) (void)hipStreamDestroy(st2_);
    st2_ = nullptr;
    if (st_) (void)hipStreamDestroy(st_);
    st_ = nullptr;
}

// host-side validation of a 3x3-block CSR description (O(nb)): a malformed one would become out-of-bounds device reads
static void validate_bsr(const xm_problem_t &prob) {
    if (!prob.rowptr || !prob.colidx || !prob.blocks) throw Error(XM_ERR_ARG, "BSR3 needs rowptr/colidx/blocks");
    if (prob.rowptr[0] != 0) throw Error(XM_ERR_ARG, "BSR3: rowptr[0] must be 0");
    for (int64_t r = 0; r < prob.n; ++r)
        if (prob.rowptr[r + 1] < prob.rowptr[r]) throw Error(XM_ERR_ARG, "BSR3: rowptr is not monotone");
    if (prob.nb != 0 && prob.rowptr[prob.n] != prob.nb) throw Error(XM_ERR_ARG, "BSR3: rowptr[n] != nb");
    const int64_t nb = prob.rowptr[prob.n];
    for (int64_t q = 0; q < nb; ++q)
        if (prob.colidx[q] < 0 || (int64_t)prob.colidx[q] >= prob.n) throw Error(XM_ERR_ARG, "BSR3: column index out of range");
}

// edge list validation shared by the view-graph storage and attach_edges: in range, no self edge, no unordered pair twice (two edges
// on one pair would write the same off-diagonal block from two threads while the diagonal sums count both)
static void validate_edges(int64_t n, int64_t ne, const int32_t *ei, const int32_t *ej, const char *who) {
    if (ne < 0 || (ne > 0 && (!ei || !ej))) throw Error(XM_ERR_ARG, std::string(who) + ": bad edge arrays");
    std::vector<int64_t> key((size_t)ne);
    for (int64_t e = 0; e < ne; ++e) {
        if (ei[e] < 0 || ej[e] < 0 || ei[e] >= n || ej[e] >= n || ei[e] == ej[e]) throw Error(XM_ERR_ARG, std::string(who) + ": bad edge (out of range or i == j)");
        const int64_t a = std::min(ei[e], ej[e]), b = std::max(ei[e], ej[e]);
        key[(size_t)e] = a * n + b;
    }
    std::sort(key.begin(), key.end());
    for (int64_t e = 1; e < ne; ++e)
        if (key[(size_t)e] == key[(size_t)e - 1]) throw Error(XM_ERR_ARG, std::string(who) + ": the same pair of cameras is listed twice (as (i,j) or (j,i))");
}

// Q = sum_e w_e G_e as 3x3-block CSR, rows sorted by column, a diagonal block for every camera (XM_STORAGE_VIEWGRAPH)
static void build_viewgraph_csr(int64_t n, int64_t ne, const int32_t *ei, const int32_t *ej, const double *w, const double *M,
                                std::vector<int64_t> &rowptr, std::vector<int32_t> &colidx, std::vector<double> &blocks) {
    if (ne > 0 && (!w || !M)) throw Error(XM_ERR_ARG, "view-graph storage needs edge_w and edge_M");
    validate_edges(n, ne, ei, ej, "view-graph storage");
    rowptr.assign((size_t)n + 1, 0);
    for (int64_t e = 0; e < ne; ++e) { rowptr[(size_t)ei[e] + 1]++; rowptr[(size_t)ej[e] + 1]++; }
    for (int64_t c = 0; c < n; ++c) rowptr[(size_t)c + 1] += rowptr[(size_t)c] + 1;   // + the diagonal block
    const int64_t nb = rowptr[(size_t)n];
    colidx.assign((size_t)nb, 0);
    blocks.assign((size_t)nb * 9, 0.0);
    std::vector<int64_t> fill(rowptr.begin(), rowptr.end() - 1), eidx((size_t)nb, -1);
    std::vector<uint8_t> etr((size_t)nb, 0);
    std::vector<double> dsum((size_t)n, 0.0);
    for (int64_t e = 0; e < ne; ++e) {
        const int64_t a = fill[(size_t)ei[e]]++, b = fill[(size_t)ej[e]]++;
        colidx[(size_t)a] = ej[e]; eidx[(size_t)a] = e; etr[(size_t)a] = 0;
        colidx[(size_t)b] = ei[e]; eidx[(size_t)b] = e; etr[(size_t)b] = 1;
        dsum[(size_t)ei[e]] += w[e]; dsum[(size_t)ej[e]] += w[e];   // EDGE order: what diag_write_kernel does after a re-weighting
    }
    for (int64_t c = 0; c < n; ++c) colidx[(size_t)fill[(size_t)c]] = (int32_t)c;   // the diagonal entry, sorted into place below
    std::vector<int64_t> perm;
    std::vector<int32_t> cc;
    std::vector<int64_t> ee;
    std::vector<uint8_t> tt;
    for (int64_t c = 0; c < n; ++c) {
        const int64_t a = rowptr[(size_t)c], len = rowptr[(size_t)c + 1] - a;
        perm.resize((size_t)len); cc.resize((size_t)len); ee.resize((size_t)len); tt.resize((size_t)len);
        for (int64_t k = 0; k < len; ++k) perm[(size_t)k] = a + k;
        std::sort(perm.begin(), perm.end(), [&](int64_t x, int64_t y) { return colidx[(size_t)x] < colidx[(size_t)y]; });
        for (int64_t k = 0; k < len; ++k) { cc[(size_t)k] = colidx[(size_t)perm[(size_t)k]]; ee[(size_t)k] = eidx[(size_t)perm[(size_t)k]]; tt[(size_t)k] = etr[(size_t)perm[(size_t)k]]; }
        for (int64_t k = 0; k < len; ++k) {
            const int64_t q = a + k;
            colidx[(size_t)q] = cc[(size_t)k];
            double *bl = blocks.data() + (size_t)q * 9;
            if (ee[(size_t)k] < 0) { bl[0] = bl[4] = bl[8] = dsum[(size_t)c]; continue; }
            const double *m = M + (size_t)ee[(size_t)k] * 9;
            const double we = w[ee[(size_t)k]];
            for (int r = 0; r < 3; ++r)
                for (int k2 = 0; k2 < 3; ++k2) bl[r * 3 + k2] = -we * (tt[(size_t)k] ? m[k2 * 3 + r] : m[r * 3 + k2]);   // Q_ij = -w M, Q_ji = Q_ij^T
        }
    }
}

// Contiguous camera ranges of a row partition: cuts[r] .. cuts[r+1] belongs to rank r.  weights == nullptr: equal ranges of
// ceil(n / world) cameras (dense rows).  weights = the rowptr of a 3x3-block CSR matrix: ranges balanced by STORED BLOCKS -- rank r
// starts at the first camera whose rowptr reaches r/world of all blocks.
// cameras per rank of the equal partition: ceil(n / world), made EVEN when there is more than one rank (the symmetric window product of a
// multi-rank dense run works in steps of two cameras, xm_symw.h); the last rank takes the remainder, padding stays at the end
int64_t equal_range_len(int64_t n, int world) {
    int64_t per = (n + world - 1) / std::max(world, 1);
    if (world > 1 && (per & 1)) ++per;
    return per;
}
void partition_cuts(int64_t n, int world, const int64_t *weights, std::vector<int64_t> &cuts) {
    cuts.assign((size_t)world + 1, 0);
    const int64_t per = equal_range_len(n, world);
    for (int r = 0; r <= world; ++r) cuts[(size_t)r] = std::min<int64_t>(n, (int64_t)r * per);
    if (weights && world > 1) {
        const int64_t nb = weights[n] - weights[0];
        int64_t c = 0;
        for (int r = 1; r < world; ++r) {
            const int64_t target = weights[0] + (nb * r) / world;
            while (c < n && weights[c] < target) ++c;
            cuts[(size_t)r] = c;
        }
        cuts[(size_t)world] = n;
    }
}

// position of global camera g in the padded numbering (rank r's cameras start at r * nloc_)
int64_t Context::pos_of(int64_t g) const {
    int r = (int)std::min<int64_t>((int64_t)cam_cut_.size() - 2, g / std::max<int64_t>(1, nloc_));
    while (r > 0 && cam_cut_[(size_t)r] > g) --r;
    while (r + 2 < (int)cam_cut_.size() && cam_cut_[(size_t)r + 1] <= g) ++r;
    return (int64_t)r * nloc_ + (g - cam_cut_[(size_t)r]);
}

void Context::init(const xm_problem_t &prob_in) {
    const int world = comm_->world, rank = comm_->rank;
    xm_problem_t prob = prob_in;
    if (prob.n < 1) throw Error(XM_ERR_ARG, "n must be >= 1");
    if (3 * prob.n > 2000000000LL) throw Error(XM_ERR_ARG, "n too large");
    n_ = prob.n;
    storage_ = prob.storage;
    XM_HIP_CHECK(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));

    // view-graph storage: the edge list becomes 3x3-block CSR on the host (and is attached for the XM^2 calls further down)
    std::vector<int64_t> vg_rp; std::vector<int32_t> vg_ci; std::vector<double> vg_bl;
    const bool viewgraph = (storage_ == XM_STORAGE_VIEWGRAPH);
    if (viewgraph) {
        build_viewgraph_csr(n_, prob.ne, prob.edge_i, prob.edge_j, prob.edge_w, prob.edge_M, vg_rp, vg_ci, vg_bl);
        prob.rowptr = vg_rp.data(); prob.colidx = vg_ci.data(); prob.blocks = vg_bl.data(); prob.nb = vg_rp[(size_t)n_];
        storage_ = XM_STORAGE_BSR3;
    }
    if (storage_ == XM_STORAGE_BSR3 || storage_ == XM_STORAGE_BSR3_DENSE) validate_bsr(prob);

    // ---- row partition: contiguous camera ranges.  Dense rows: equal ranges.  Block-sparse: ranges balanced by STORED BLOCKS
    //  unless Settings.balance == 1 -- a hub-camera graph would otherwise land most of the matrix on one rank.  Every rank
    // is padded to the longest range with inert cameras (zero rows of Q, R = [I 0], s = 1: exact zeros in every sum), so all ranks
    // exchange equal chunks; a camera's record sits at position rank * nloc_ + (g - cut[rank]) of every replicated vector.
    partition_cuts(n_, world, (storage_ == XM_STORAGE_BSR3 && cfg_.balance == 0) ? prob.rowptr : nullptr, cam_cut_);
    int64_t longest = 1;
    for (int r = 0; r < world; ++r) longest = std::max(longest, cam_cut_[(size_t)r + 1] - cam_cut_[(size_t)r]);
    if (world > 1 && !((storage_ == XM_STORAGE_BSR3) && cfg_.balance == 0)) longest = std::max(longest, equal_range_len(n_, world));   // (a last rank with few cameras)
    nloc_ = (int)longest;
    cam0_ = rank * nloc_;
    g0_ = cam_cut_[(size_t)rank];
    ntot_ = (int64_t)nloc_ * world;
    ld_ = dense_ld(ntot_);
    const int64_t true_loc = cam_cut_[(size_t)rank + 1] - g0_;  // real cameras on this rank
    true_loc_ = true_loc;
    comm_->reserve((size_t)ld_ * (kMaxRank + 1) + 4096);   // staging of the largest all-gather (W at the top rank); collective

    if (storage_ == XM_STORAGE_DENSE) {
        if (prob.q_on_device) {
            dQ_ = const_cast<double *>(prob.q);
            ownQ_ = false;
        } else {
            // q may hold the whole matrix (q_row0 == 0, ldq >= 3n) or just a row strip that covers this rank's cameras
            const int64_t need0 = 3 * g0_, need1 = 3 * (g0_ + true_loc);
            if (!prob.q || prob.q_row0 < 0 || (true_loc > 0 && (prob.q_row0 > need0 || prob.q_row0 + prob.ldq < need1)))
                throw Error(XM_ERR_ARG, "dense Q needs q with the rows of this rank's cameras (ldq >= 3n for the whole matrix)");
            const size_t rows = (size_t)3 * nloc_;
            XM_HIP_CHECK(hipMalloc((void **)&dQ_, rows * (size_t)ld_ * sizeof(double)));
            ownQ_ = true;
            XM_HIP_CHECK(hipMemsetAsync(dQ_, 0, rows * (size_t)ld_ * sizeof(double), st_));
            if (true_loc > 0) {
                // rows [3 g0, 3 g0 + 3 true_loc) of the column-major host matrix, all 3n columns -> device slab
                // (column-major, leading dim = local rows), then an LDS-tiled transpose into the padded row-major layout.
                const int64_t lr = 3 * true_loc, cols = 3 * n_;
                DevBuf<double> slab;
                slab.alloc((size_t)lr * cols, false);
                XM_HIP_CHECK(hipMemcpy2D(slab.p, (size_t)lr * sizeof(double), prob.q + (3 * g0_ - prob.q_row0), (size_t)prob.ldq * sizeof(double),
                                         (size_t)lr * sizeof(double), (size_t)cols, hipMemcpyHostToDevice));
                launch_transpose_pad(slab.p, lr, lr, cols, dQ_, ld_, st_);
                XM_HIP_CHECK(hipStreamSynchronize(st_));
            }
        }
    } else if (storage_ == XM_STORAGE_BSR3_DENSE) {
        // described as 3x3-block CSR on the host, stored dense (the reference's format) on the device: every rank expands
        // only its own camera rows, so a 13.5 GB Q never exists on the host or on one GPU of a multi-GPU run
        const size_t rows = (size_t)3 * nloc_;
        XM_HIP_CHECK(hipMalloc((void **)&dQ_, rows * (size_t)ld_ * sizeof(double)));
        ownQ_ = true;
        XM_HIP_CHECK(hipMemsetAsync(dQ_, 0, rows * (size_t)ld_ * sizeof(double), st_));
        if (true_loc > 0) {
            const int64_t b0 = prob.rowptr[g0_], b1 = prob.rowptr[g0_ + true_loc];
            std::vector<int64_t> rp((size_t)true_loc + 1);
            for (int64_t i = 0; i <= true_loc; ++i) rp[(size_t)i] = prob.rowptr[g0_ + i] - b0;
            DevBuf<int64_t> drp; DevBuf<int32_t> dci; DevBuf<double> dbl;
            drp.alloc(rp.size(), false); dci.alloc((size_t)std::max<int64_t>(b1 - b0, 1), false); dbl.alloc((size_t)std::max<int64_t>(b1 - b0, 1) * 9, false);
            XM_HIP_CHECK(hipMemcpy(drp.p, rp.data(), rp.size() * sizeof(int64_t), hipMemcpyHostToDevice));
            if (b1 > b0) {
                XM_HIP_CHECK(hipMemcpy(dci.p, prob.colidx + b0, (size_t)(b1 - b0) * sizeof(int32_t), hipMemcpyHostToDevice));
                XM_HIP_CHECK(hipMemcpy(dbl.p, prob.blocks + b0 * 9, (size_t)(b1 - b0) * 9 * sizeof(double), hipMemcpyHostToDevice));
            }
            launch_dense_from_bsr(drp.p, dci.p, dbl.p, true_loc, 0, dQ_, ld_, st_);
        }
        XM_HIP_CHECK(hipStreamSynchronize(st_));
        storage_ = XM_STORAGE_DENSE;
    } else if (storage_ == XM_STORAGE_BSR3) {
        std::vector<int64_t> rp((size_t)nloc_ + 1, 0);
        const int64_t b0 = (true_loc > 0) ? prob.rowptr[g0_] : 0;
        for (int64_t i = 0; i <= nloc_; ++i) rp[(size_t)i] = ((true_loc > 0) ? prob.rowptr[g0_ + std::min<int64_t>(i, true_loc)] : 0) - b0;
        nb_loc_ = rp[(size_t)nloc_];
        // column indices in the padded numbering (identity for equal ranges, whose only padding is at the end)
        std::vector<int32_t> ci((size_t)std::max<int64_t>(nb_loc_, 1), 0);
        for (int64_t q = 0; q < nb_loc_; ++q) ci[(size_t)q] = (int32_t)pos_of(prob.colidx[b0 + q]);
        rowptr_.alloc((size_t)nloc_ + 1);
        colidx_.alloc((size_t)std::max<int64_t>(nb_loc_, 1));
        blocks_.alloc((size_t)std::max<int64_t>(nb_loc_, 1) * 9);
        XM_HIP_CHECK(hipMemcpy(rowptr_.p, rp.data(), rp.size() * sizeof(int64_t), hipMemcpyHostToDevice));
        {
            std::vector<int4> ri;
            bsr_build_rowinfo(rp.data(), nloc_, ri);
            rowinfo_.alloc(std::max<size_t>(ri.size(), 1));
            if (!ri.empty()) XM_HIP_CHECK(hipMemcpy(rowinfo_.p, ri.data(), ri.size() * sizeof(int4), hipMemcpyHostToDevice));
        }
        if (nb_loc_ > 0) {
            XM_HIP_CHECK(hipMemcpy(colidx_.p, ci.data(), (size_t)nb_loc_ * sizeof(int32_t), hipMemcpyHostToDevice));
            XM_HIP_CHECK(hipMemcpy(blocks_.p, prob.blocks + b0 * 9, (size_t)nb_loc_ * 9 * sizeof(double), hipMemcpyHostToDevice));
        }
        // Large problems: sliced-ELL over per-XCD column slabs (xm_sell.h).  Below ~2.2 M blocks per GPU the one-launch CSR kernel wins --
        // measured (profiles/r06_kbench_bsr_policy.txt, o = 3, us; the CSR kernel reads its blocks with the default cache policy at these sizes):
        // 1.39 M blocks CSR 31.5 / sliced ELL 39.3, 2.05 M blocks 49.3 / 52.2, 2.79 M blocks 72.4 / 66.1 (round 5, non-temporal blocks:
        // 425 k 13.2 / 23.0, 929 k 25.5 / 32.0, 2.05 M 53.1 / 46.8); inside the solve at 425 k blocks
        // (Hessian epilogue, HIP events) 20.4 / 28.9 and 22.4 k / 18.3 k tCG iterations per second (r05_bench_rome_bsr*.json).  View-graph storage compresses the
        // stream with the quaternion codec (36 instead of 76 bytes per stored block).  Settings: sell, sell_slabs, sell_lmax,
        // sell_gather, sell_codec (xm_tuning_t).
        if (cfg_.sell == 1 || (cfg_.sell == 0 && nb_loc_ >= kSellMinBlocks)) {
            sell_gm_ = cfg_.sell_gather;
            const int codec = (cfg_.sell_codec == 2 || (cfg_.sell_codec == 0 && viewgraph)) ? SELL_CODEC_QUAT : SELL_CODEC_FULL;
            // layout: sorted virtual rows + a second launch for the per-camera sum (xm_sell.h).  A chunk-tiled ONE-launch layout was measured
            // in round 4 (profiles/r04_kbench_sell2.txt: 90.3 us against 82.3 us in two launches at 100 k cameras) and removed in round 5.
            sell_.reset(new SellMatrix(rp.data(), ci.data(), prob.blocks + b0 * 9, nloc_, ntot_, cfg_.sell_slabs, cfg_.sell_lmax, st_, codec, cam0_));
        }
    } else if (storage_ == XM_STORAGE_SCHUR) {
        // several ranks (round 4): every rank builds the factors from the whole observation list; the rows of VT^-1 and the cameras of the
        // last kernel of the chain are partitioned (xm_schur.h)
        SchurSettings sc;
        sc.host_assembly = cfg_.schur_host_assembly; sc.sym_min_rows = cfg_.sym_rows(1); sc.trace = cfg_.schur_trace;
        sc.solver = cfg_.schur_solver; sc.dense_max = cfg_.schur_dense_max;
        sc.pcg_first = cfg_.schur_pcg_first; sc.pcg_hess_digits = cfg_.schur_pcg_hess_digits;
        schur_.reset(new SchurOp(n_, prob.n_landmarks, prob.nobs, prob.obs_cam, prob.obs_lm, prob.obs_p, prob.obs_w, st_, comm_.get(), sc));
        w_cur_.assign(prob.obs_w, prob.obs_w + prob.nobs);
    } else {
        throw Error(XM_ERR_ARG, "unknown storage");
    }
    // Symmetric half-traffic product: dense, Q exactly symmetric.  It pays from ~1300 cameras on (Settings::sym_rows: measured cross-over;
    // 13682 cameras 2000 -> 1110 us at o = 3, 1778 cameras 31.9 -> 25.8 us).  Settings.sym: auto = on for 3n >= sym_rows() and o <= 4; 1 forces
    // it for every size (o <= 5, 1e-9 relative asymmetry accepted); -1 disables it.  Single rank only: the kernel sweeps the upper
    // triangle of the WHOLE matrix (a rank's row strip is a rectangle).
    sym_ok_ = false;
    sym_max_o_ = 4;
    {
        const bool force = cfg_.sym == 1, off = cfg_.sym == -1;
        if (force) sym_max_o_ = 5;
        if (storage_ == XM_STORAGE_DENSE && world == 1 && !off && (force || 3 * n_ >= cfg_.sym_rows(1))) {
            const int grid = 2048;
            DevBuf<double> d;
            d.alloc((size_t)2 * grid);
            launch_asym(dQ_, ld_, 3 * n_, d.p, grid, st_);
            std::vector<double> h((size_t)2 * grid);
            XM_HIP_CHECK(hipMemcpyAsync(h.data(), d.p, h.size() * sizeof(double), hipMemcpyDeviceToHost, st_));
            XM_HIP_CHECK(hipStreamSynchronize(st_));
            double da = 0, mx = 0;
            bool nan = false;
            for (int i = 0; i < grid; ++i) {
                if (h[(size_t)i] != h[(size_t)i]) nan = true;   // a NaN entry anywhere: never symmetric
                else da = std::max(da, h[(size_t)i]);
                mx = std::max(mx, h[(size_t)grid + i]);
            }
            if (nan) da = std::nan("");
            q_asym_ = da; q_max_ = mx;
            // The lower triangle is never read on this path, so by default it is taken only for an EXACTLY symmetric matrix (what
            // utils/creatematrix.py:326-328 writes): round-off asymmetry in a Q.bin is honoured like cublasDgemm does, at every
            // size.  sym = 1 accepts |Q - Q^T| <= 1e-9 |Q| (results then differ from the general path by that much).
            sym_ok_ = force ? (da <= 1e-9 * mx) : (da == 0.0);
        }
    }
    // Several ranks: every rank streams half of its row strip through a cyclic half window (xm_symw.h) -- the upper triangle cut into row
    // strips would leave rank 0 with almost its whole strip.  Whether Q is symmetric cannot be seen from one strip.  The policy is the
    // single-GPU one: automatic (sym = 0) only for an EXACTLY symmetric matrix -- an order-independent checksum modulo 2^64 over the
    // strips (launch_symhash), all-gathered; forced (sym = 1) when a random vector multiplied both ways (general kernel / window
    // product) agrees to 1e-9 on every rank.
    symw_.reset();
    if (storage_ == XM_STORAGE_DENSE && world > 1 && comm_->active() && cfg_.sym != -1 && (nloc_ % 2) == 0 &&
        (cfg_.sym == 1 || 3 * n_ >= cfg_.sym_rows(world))) {
        sym_max_o_ = (cfg_.sym == 1) ? 5 : 4;
        symw_.reset(new SymwProduct(ntot_, nloc_, cam0_, ld_, st_));
        comm_->reserve(symw_csum_count(ntot_, sym_max_o_) * (size_t)world + 4096);
        symw_->ensure(sym_max_o_, world);   // once, for every rank of the staircase: no free between two collectives later on
        DevBuf<double> x, y1, y2, flag;
        x.alloc((size_t)ld_ + 16); y1.alloc((size_t)3 * nloc_); y2.alloc((size_t)3 * nloc_); flag.alloc((size_t)world);
        std::vector<double> hx((size_t)ld_, 0.0);
        unsigned long long lcg = 88172645463325252ull;
        for (int64_t i = 0; i < 3 * ntot_; ++i) { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; hx[(size_t)i] = (double)(lcg >> 11) / 9007199254740992.0 - 0.5; }
        XM_HIP_CHECK(hipMemcpyAsync(x.p, hx.data(), hx.size() * sizeof(double), hipMemcpyHostToDevice, st_));
        CamArgs pa;
        std::memset(&pa, 0, sizeof(pa));
        pa.nloc = nloc_; pa.cam0 = cam0_;
        pa.out = y1.p;
        launch_qw_dense(1, EPI_PLAIN, dQ_, ld_, x.p, 1.0, pa, st_);
        pa.out = y2.p;
        symw_->sweep(1, dQ_, x.p, nullptr, rank, st_);
        comm_->allgather(symw_->csum_all(), symw_->csum_count(1), st_);
        symw_->reduce(1, EPI_PLAIN, 1.0, pa, world, st_);
        std::vector<double> h1((size_t)3 * nloc_), h2((size_t)3 * nloc_);
        XM_HIP_CHECK(hipMemcpyAsync(h1.data(), y1.p, h1.size() * sizeof(double), hipMemcpyDeviceToHost, st_));
        XM_HIP_CHECK(hipMemcpyAsync(h2.data(), y2.p, h2.size() * sizeof(double), hipMemcpyDeviceToHost, st_));
        XM_HIP_CHECK(hipStreamSynchronize(st_));
        double dmax = 0, ymax = 0;
        bool bad = false;
        for (size_t i = 0; i < h1.size(); ++i) {
            if (h1[i] != h1[i] || h2[i] != h2[i]) bad = true;
            dmax = std::max(dmax, std::fabs(h1[i] - h2[i])); ymax = std::max(ymax, std::fabs(h1[i]));
        }
        // the bound has to hold for the whole vector, not only for this rank's rows: every rank reports its own maxima
        std::vector<double> rep((size_t)world * 2, 0.0);
        DevBuf<double> repd;
        repd.alloc((size_t)world * 2);
        const double mine[2] = {bad ? 1e300 : dmax, ymax};
        XM_HIP_CHECK(hipMemcpyAsync(repd.p + (size_t)rank * 2, mine, sizeof(mine), hipMemcpyHostToDevice, st_));
        comm_->allgather(repd.p, 2, st_);
        XM_HIP_CHECK(hipMemcpyAsync(rep.data(), repd.p, rep.size() * sizeof(double), hipMemcpyDeviceToHost, st_));
        XM_HIP_CHECK(hipStreamSynchronize(st_));
        double dall = 0, yall = 0;
        for (int r = 0; r < world; ++r) { dall = std::max(dall, rep[(size_t)2 * r]); yall = std::max(yall, rep[(size_t)2 * r + 1]); }
        q_asym_ = dall; q_max_ = yall;
        if (!(dall <= 1e-9 * yall)) symw_.reset();   // identical decision on every rank (identical gathered numbers)
        if (symw_ && cfg_.sym != 1) {
            // automatic mode: exact symmetry, as on one GPU (there: launch_asym == 0).  The strips' checksums travel as bit patterns.
            const int grid = 1024;
            DevBuf<unsigned long long> hsh;
            hsh.alloc((size_t)2 * grid);
            launch_symhash(dQ_, ld_, 3 * (int64_t)cam0_, 3 * (int64_t)nloc_, 3 * ntot_, hsh.p, grid, st_);
            std::vector<unsigned long long> hh((size_t)2 * grid);
            XM_HIP_CHECK(hipMemcpyAsync(hh.data(), hsh.p, hh.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, st_));
            XM_HIP_CHECK(hipStreamSynchronize(st_));
            unsigned long long mine2[2] = {0ull, 0ull};
            for (int b = 0; b < grid; ++b) { mine2[0] += hh[(size_t)2 * b]; mine2[1] |= hh[(size_t)2 * b + 1]; }
            static_assert(sizeof(unsigned long long) == sizeof(double), "checksums travel through the double all-gather");
            XM_HIP_CHECK(hipMemcpyAsync(repd.p + (size_t)rank * 2, mine2, sizeof(mine2), hipMemcpyHostToDevice, st_));
            comm_->allgather(repd.p, 2, st_);
            std::vector<unsigned long long> all((size_t)world * 2);
            XM_HIP_CHECK(hipMemcpyAsync(all.data(), repd.p, all.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, st_));
            XM_HIP_CHECK(hipStreamSynchronize(st_));
            unsigned long long tot = 0ull, bad2 = 0ull;
            for (int r = 0; r < world; ++r) { tot += all[(size_t)2 * r]; bad2 |= all[(size_t)2 * r + 1]; }
            if (tot != 0ull || bad2 != 0ull) symw_.reset();   // identical on every rank
        }
        comm_->host_barrier();
    }
    XM_HIP_CHECK(hipHostMalloc((void **)&hstat_, 256, hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(hstat_, 0, 256);
    XM_HIP_CHECK(hipHostGetDevicePointer((void **)&hstat_dev_, hstat_, 0));
    ensure_pinned((size_t)1 << 17);   // 1 MB to start with; setup_rank sizes it for the solve
    if (viewgraph) {
        attach_edges(prob.ne, prob.edge_i, prob.edge_j, prob.edge_M);
        w_cur_.assign(prob.edge_w, prob.edge_w + prob.ne);
    }
}

Context::~Context() { release_raw(); }

// ------------------------------------------------------------------------------------------------------------------
// per-rank workspace
// ------------------------------------------------------------------------------------------------------------------
void Context::setup_rank(int o) {
    // (Re)allocation frees device memory, and hipFree synchronises the WHOLE device.  With several ranks of one process on one device
    // (virtual devices) that would wait for a peer's wait kernel, which in turn waits for a push this rank has not enqueued yet: the
    // host barrier makes sure every rank has enqueued everything up to here before anybody frees (no-op for other communicators).
    if (comm_->active()) comm_->host_barrier();
    o_ = o;
    OP_ = pitch_of(o);
    const size_t mat = (size_t)nloc_ * 3 * OP_, vec = (size_t)nloc_;
    const int world = comm_->world;
    // grow-only workspace: sized once for the highest rank this solve can reach (the pitch grows with the rank), reused by every later
    // rank level and every later solve of the context
    const int o_top = std::max(o, (int)std::min<unsigned>(opt_ ? opt_->max_rank : 0u, (unsigned)kMaxRank));
    const size_t mat_top = (size_t)nloc_ * 3 * pitch_of(o_top);
    for (DevBuf<double> *b : {&R_, &Rc_, &D_, &rR_, &pR_, &vR_, &HvR_, &HpR_}) b->ensure(mat, st_, mat_top);
    for (DevBuf<double> *b : {&s_, &sc_, &rs_, &rsB_, &psA_, &psB_, &vs_, &Hvs_, &Hps_}) b->ensure(vec, st_);
    for (int k = 0; k < 2; ++k) {
        ps_[k].G.ensure(mat, st_, mat_top); ps_[k].rgR.ensure(mat, st_, mat_top); ps_[k].egs.ensure(vec, st_); ps_[k].rgs.ensure(vec, st_); ps_[k].S0.ensure(vec * 9, st_);
    }
    cur_ = 0;
    wpad_on_ = sell_ && cfg_.sell_wpad >= 0 && !comm_->active() && sell_supports(o) && o >= 3 && 3 * OP_ <= 16 && (cfg_.sell_wpad == 1 || sell_->padded_pays(o));
    if (wpad_on_ && !wpad_.p) wpad_.alloc((size_t)ntot_ * 16 + 16);   // zero-filled: the pad is never written
    W_.ensure((size_t)ld_ * OP_ + 16, st_, (size_t)ld_ * pitch_of(o_top) + 16);   // + slack: the sector-window gather of the sliced-ELL product reads whole 64-byte sectors around a record
    const int nA_loc = prod_grid(), nB_loc = tcg_blocks();
    nA_ = nA_loc * world;
    nB_ = nB_loc * world;
    partsA_.ensure((size_t)3 * nA_, st_);
    // tCG exchange buffers: two parity buffers of world chunks [rows of the image of Hp (multi-rank only) | 3*nA_loc | nB_loc]
    const size_t b_off = comm_->active() ? mat : 0;
    const size_t pb = (size_t)2 * (b_off * world + 3 * nA_ + nB_);
    if (comm_->peer() && world > 1 && cfg_.exchange != 1 && !symw_ && storage_ != XM_STORAGE_SCHUR && comm_->device_waits()) {   // (the window product and the matrix-free chain all-gather between their launches: lockstep loop)
        // direct peer exchange: the buffers live in memory every rank of the group can store into (collective, host-synchronised)
        partsB_.release();
        comm_->xchg_setup(pb, xchg_);
        partsB_peer_ = xchg_.buf[comm_->rank];
        xchg_.mute = (cfg_.debug_peer_mute && comm_->rank == 1) ? 1 : 0;
        xchg_.lite = cfg_.exchange_lite;
    } else {
        xchg_ = PeerXchg();
        partsB_peer_ = nullptr;
        partsB_.ensure(pb, st_, (size_t)2 * ((comm_->active() ? mat_top : 0) * world + 3 * nA_ + nB_));
    }
    if (comm_->active()) Afull_.ensure(mat * world, st_, mat_top * world); else Afull_.release();
    partsM_.ensure((size_t)std::max(std::max(nB_, flat_grid((int64_t)mat_top) * world), 4 * ((nloc_ + 255) / 256) * world), st_);   // (device-driven outer iteration: one model partial per wavefront of 64 cameras)
    if (sym_ok_ && o >= 3 && o <= sym_max_o_) {
        Prow_.alloc(sym_prow_count(nloc_, ld_, o));
        Pcol_.alloc(sym_pcol_count(nloc_, ld_, o), false);
    } else {
        Prow_.release(); Pcol_.release();
    }
    // column-split product for small strips (xm_kernels.hip:qw_dense_ks_kernel): multi-rank dense storage by default
    ks_ = 1;
    if (storage_ == XM_STORAGE_DENSE && !sym_ok_ && cfg_.split_k >= 0) {
        if (cfg_.split_k >= 2) ks_ = std::min(cfg_.split_k, std::max(1, (int)((ld_ + 255) / 256)));
        else if (comm_->active()) ks_ = qw_dense_split_k(nloc_, ld_);
    }
    if (ks_ > 1) {
        ksum_.alloc((size_t)ks_ * mat);
        kcount_.alloc((size_t)qw_grid(nloc_));
    }
    if (overlap_applies()) {
        if (!st2_) {
            XM_HIP_CHECK(hipStreamCreateWithFlags(&st2_, hipStreamNonBlocking));
            XM_HIP_CHECK(hipEventCreateWithFlags(&ev_w_, hipEventDisableTiming));
            XM_HIP_CHECK(hipEventCreateWithFlags(&ev_p_, hipEventDisableTiming));
        }
        Pstrip_.alloc(mat);
    }
    scal_.ensure(2, st_);
    spec_.ensure(1, st_);
    // the sliced-ELL product's partial-result buffer grows with o: (re)allocate it HERE, between the two barriers -- hipFree synchronises
    // the whole device, and with several ranks of one process on one device ("virtual devices") a free between two collectives waits for
    // a peer's spinning wait kernel that waits for this rank's next push (8 virtual ranks ran into exactly that at the first o = 4 product)
    if (sell_ && sell_supports(o)) (void)sell_->parts(o);
    // pinned staging: partial sums, a whole replicated point (download_point) and a Lanczos vector
    ensure_pinned(std::max<size_t>((size_t)2 * nA_ + (size_t)nB_ + partsM_.count + 64, (size_t)ld_ * OP_ + (size_t)ntot_ + 1024));
    if (comm_->active()) comm_->host_barrier();   // nobody enqueues the next collective while somebody is still freeing
}

// host column-major (true n) -> device row-major (pitch OP) rows of the local cameras; padding cameras get [I 0]
void Context::upload_point(const std::vector<double> &R_cm, int o, const std::vector<double> &s_ex) {
    const size_t m = (size_t)3 * n_;
    std::vector<double> Rr((size_t)nloc_ * 3 * OP_, 0.0), sl((size_t)nloc_, 1.0);
    for (int c = 0; c < nloc_; ++c) {
        const bool real = c < true_loc_;
        const int64_t g = g0_ + c;
        for (int a = 0; a < 3; ++a)
            for (int k = 0; k < o; ++k)
                Rr[((size_t)c * 3 + a) * OP_ + k] = real ? R_cm[(size_t)(3 * g + a) + (size_t)k * m] : (a == k ? 1.0 : 0.0);
        if (real) sl[(size_t)c] = s_ex[(size_t)g];
    }
    if (cam0_ == 0) sl[0] = 1.0;  // the anchor's scale is 1 inside the solver (trustregion.h:125-127)
    to_dev(R_.p, Rr.data(), Rr.size() * sizeof(double));
    to_dev(s_.p, sl.data(), sl.size() * sizeof(double));
}

// device -> host column-major for ALL cameras (gathers across ranks through the W buffer)
void Context::download_point(std::vector<double> &R_cm, std::vector<double> &s_ex) {
    const size_t m = (size_t)3 * n_, mat = (size_t)nloc_ * 3 * OP_;
    std::vector<double> full((size_t)ntot_ * 3 * OP_), sf((size_t)ntot_);
    // copies by KERNEL, not by the copy engine: the ranks of one process on one device share the engine's in-order queue, and a peer's
    // device-to-host copy that waits there for its wait kernel would block this copy, i.e. the push that wait kernel is waiting for
    launch_scale_copy(W_.p + (size_t)comm_->rank * mat, R_.p, 1.0, (int64_t)mat, st_);
    if (comm_->active()) comm_->allgather(W_.p, mat, st_);
    to_host(full.data(), W_.p, full.size() * sizeof(double));
    launch_scale_copy(W_.p + (size_t)comm_->rank * nloc_, s_.p, 1.0, (int64_t)nloc_, st_);
    if (comm_->active()) comm_->allgather(W_.p, (size_t)nloc_, st_);
    to_host(sf.data(), W_.p, sf.size() * sizeof(double));
    XM_HIP_CHECK(hipMemsetAsync(W_.p, 0, W_.count * sizeof(double), st_));
    R_cm.assign(m * (size_t)o_, 0.0);
    s_ex.assign((size_t)n_, 1.0);
    for (int64_t g = 0; g < n_; ++g) {
        const size_t q = (size_t)pos_of(g);
        for (int a = 0; a < 3; ++a)
            for (int k = 0; k < o_; ++k) R_cm[(size_t)(3 * g + a) + (size_t)k * m] = full[(q * 3 + a) * OP_ + k];
        s_ex[(size_t)g] = sf[q];
    }
}

// workgroups of the product kernels == number of per-workgroup partial sums per epilogue slot
int Context::prod_grid() const {
    if (storage_ == XM_STORAGE_BSR3 && sell_ && sell_supports(o_)) return sell_->reduce_grid(o_, nloc_);
    return (storage_ == XM_STORAGE_BSR3) ? bsr_grid(nloc_) : qw_grid(nloc_);   // dense and matrix-free: one wavefront per camera
}

// cg_step_kernel waits for its peers INSIDE the launch when the direct exchange is fused into it: every workgroup of every rank that
// shares this device has to be resident at once, or the resident ones wait for the others' turn for ever (until the bounded spin
// expires).  256 CUs x 4 workgroups is admitted whatever the kernel's register count (cg_step: 61 VGPRs, 106 SGPRs -> 6 per CU).
int Context::tcg_blocks() const {
    int g = flat_grid((int64_t)nloc_ * 3 * OP_);
    if (comm_->peer() && comm_->world > 1 && cfg_.exchange != 1 && !symw_ && storage_ != XM_STORAGE_SCHUR && comm_->device_waits()) g = std::min(g, std::max(8, 1024 / std::max(1, comm_->ranks_on_my_device())));
    return g;
}

CamArgs Context::cam_args(int state) const {
    CamArgs a;
    std::memset(&a, 0, sizeof(a));
    a.nloc = nloc_;
    a.cam0 = cam0_;
    a.lam = opt_ ? opt_->lam : 0.0;
    a.R = R_.p;
    a.s = s_.p;
    const PointState &p = ps_[state];
    a.G = p.G.p; a.egs = p.egs.p; a.S0 = p.S0.p; a.rgR = p.rgR.p; a.rgs = p.rgs.p;
    a.pR = pR_.p; a.ps = psA_.p; a.rR = rR_.p; a.rs = rs_.p; a.HpR = HpR_.p; a.Hps = Hps_.p;
    a.Wloc = W_.p + (size_t)cam0_ * 3 * OP_;
    a.out = HpR_.p;
    a.partials = partsA_.p;
    a.scal = scal_.p;
    if (ks_ > 1) { a.ks = ks_; a.ksum = ksum_.p; a.kcount = kcount_.p; }
    return a;
}

void Context::product(int epi, int o, double alpha, const CamArgs &a) {
    if (w_pending_) {
        if (storage_ == XM_STORAGE_DENSE && (epi == EPI_PLAIN || epi == EPI_GRAD) && o == o_) {
            // local column strip on the second stream beside the all-gather, the rest after it
            w_pending_ = false;
            const int tc = qw_dense_tile_cols();
            const int64_t c0 = 3 * (int64_t)cam0_, c1 = 3 * ((int64_t)cam0_ + nloc_);
            CamArgs s1 = a, s2 = a;
            s1.range_mode = 1; s1.t_lo = (int)((c0 + tc - 1) / tc); s1.t_hi = (int)(c1 / tc); s1.out = Pstrip_.p; s1.addend = nullptr;
            s2.range_mode = 2; s2.t_lo = s1.t_lo; s2.t_hi = s1.t_hi; s2.addend = Pstrip_.p;
            comm_->note("overlap_split", (double)s1.t_lo, (double)s1.t_hi);
            XM_HIP_CHECK(hipEventRecord(ev_w_, st_));
            XM_HIP_CHECK(hipStreamWaitEvent(st2_, ev_w_, 0));
            launch_qw_dense_split(o, EPI_PLAIN, dQ_, ld_, W_.p, 1.0, s1, st2_);
            XM_HIP_CHECK(hipEventRecord(ev_p_, st2_));
            comm_->allgather(W_.p, (size_t)nloc_ * 3 * OP_, st_);
            XM_HIP_CHECK(hipStreamWaitEvent(st_, ev_p_, 0));
            launch_qw_dense_split(o, epi, dQ_, ld_, W_.p, alpha, s2, st_);
            if (res_) res_->qw_products++;
            return;
        }
        flush_gather();
    }
    if (storage_ == XM_STORAGE_DENSE) {
        if (symw_ && o == o_ && o >= 3 && o <= sym_max_o_ && epi != EPI_CERT) product_symw(epi, o, alpha, a);
        else if (sym_ok_ && o == o_ && o >= 3 && o <= sym_max_o_ && epi != EPI_CERT && Pcol_.p) launch_qw_sym(o, epi, dQ_, ld_, W_.p, alpha, a, Prow_.p, Pcol_.p, st_, sym_rev_);
        else { CamArgs ar = a; ar.rev = sym_rev_; launch_qw_dense(o, epi, dQ_, ld_, W_.p, alpha, ar, st_); }
    } else if (storage_ == XM_STORAGE_SCHUR) {
        schur_->product(o, epi, W_.p, alpha, a, st_);
    } else if (sell_ && sell_supports(o)) {
        // inside the tCG the kernels that write W keep a copy at the 128-byte record pitch (run_tcg): the gather reads that one
        const double *wp = (epi == EPI_HESS && o == o_) ? wpad() : ((epi == EPI_GRAD && o == o_) ? wpad_next_ : nullptr);
        wpad_next_ = nullptr;
        launch_qw_sell(o, epi, *sell_, W_.p, alpha, a, sell_gm_, st_, wp);
    } else {
        launch_qw_bsr3(o, epi, rowptr_.p, colidx_.p, blocks_.p, W_.p, alpha, a, st_, nb_loc_, rowinfo_.p);
    }
    if (res_) res_->qw_products++;
}

int Context::product_kind(int o) const {
    if (storage_ == XM_STORAGE_SCHUR) return XM_PRODUCT_SCHUR;
    if (storage_ == XM_STORAGE_DENSE) return ((symw_ || sym_ok_) && o >= 3 && o <= sym_max_o_) ? XM_PRODUCT_DENSE_SYM : XM_PRODUCT_DENSE;
    if (sell_ && sell_supports(o)) return sell_->codec() == SELL_CODEC_QUAT ? XM_PRODUCT_SELL_QUAT : XM_PRODUCT_SELL;
    return XM_PRODUCT_BSR3;
}

// multi-rank symmetric Q: sweep of this rank's half window, all-gather of the ranks' column sums, per-camera sum + epilogue (xm_symw.h)
void Context::product_symw(int epi, int o, double alpha, const CamArgs &a) {
    symw_->sweep(o, dQ_, W_.p, (epi == EPI_HESS) ? a.scal : (const TcgScal *)nullptr, comm_->rank, st_);
    comm_->allgather(symw_->csum_all(), symw_->csum_count(o), st_);
    symw_->reduce(o, epi, alpha, a, comm_->world, st_);
}

// "overlap the gather with the local (diagonal-strip) part of Q*W".  The rows of W this rank owns are final before the
// all-gather starts, so the column tiles of Q that lie entirely inside the rank's own column range can be multiplied on a second
// stream WHILE the collective runs; the rest of the product (all other tiles, + those raw sums, + the epilogue) follows the
// gather.  Applies to the dense products OUTSIDE the tCG (cost/gradient of a candidate point, line search, certificate right-hand
// side) — inside the tCG the product input comes from replicated data and there is no gather to hide.  It costs
// an extra launch, so it is used only when the per-rank matrix is large (XM_OVERLAP_MIN_MB, default 64; XM_OVERLAP=0 disables).
bool Context::overlap_applies() const {
    if (cfg_.overlap < 0 || !comm_->active() || storage_ != XM_STORAGE_DENSE || sym_ok_ || symw_) return false;
    if ((double)nloc_ * 3.0 * (double)ld_ * 8.0 < cfg_.overlap_min_mb * 1048576.0) return false;
    const int tc = qw_dense_tile_cols();
    const int64_t c0 = 3 * (int64_t)cam0_, c1 = 3 * ((int64_t)cam0_ + nloc_);
    return (c1 / tc) > ((c0 + tc - 1) / tc);
}

void Context::gather_W() {
    if (!comm_->active()) return;
    if (overlap_applies()) { w_pending_ = true; return; }   // resolved by the next product()
    comm_->allgather(W_.p, (size_t)nloc_ * 3 * OP_, st_);
}

void Context::flush_gather() {
    if (!w_pending_) return;
    w_pending_ = false;
    comm_->allgather(W_.p, (size_t)nloc_ * 3 * OP_, st_);
}

double Context::sum_parts(const double *dparts, int count) {
    if (comm_->peer()) launch_copy_words(hpin_dev_, dparts, (size_t)count * sizeof(double), st_);   // see to_host
    else XM_HIP_CHECK(hipMemcpyAsync(hpin_, dparts, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, st_));
    XM_HIP_CHECK(hipStreamSynchronize(st_));
    double t = 0.0;
    for (int i = 0; i < count; ++i) t += hpin_[i];
    return t;
}

// Gradient epilogue on the point (Rp, sp) with the product input currently in W (already gathered).
// Fills ps_[state]; returns f and <rg,rg>_metric.   trustregion.h:162-170 + 186-194 + 307-317 + 483-484
void Context::eval_point(int state, const double *Rp, const double *sp, double &f, double &rr) {
    const int nA_loc = prod_grid();
    CamArgs a = cam_args(state);
    a.R = Rp;
    a.s = sp;
    a.partials = partsA_.p + (size_t)comm_->rank * 2 * nA_loc;
    product(EPI_GRAD, o_, 2.0, a);
    if (comm_->active()) comm_->allgather(partsA_.p, (size_t)2 * nA_loc, st_);
    launch_outer_finalize(partsA_.p, nA_loc, comm_->world, partsM_.p, 0, scal_.p, reinterpret_cast<double *>(hstat_dev_) + 8,
                          ++outer_seq_, grouping_, st_);
    volatile double *hres = wait_outer_result();
    f = hres[0];
    rr = hres[1];
}

// Wall-clock decisions must be identical on every rank (a rank that stops alone would leave the others inside a
// collective): the local flags are all-gathered and OR-ed.  Single rank: returns the flag.
bool Context::agree_any(bool local) {
    if (!comm_->active()) return local;
    const int world = comm_->world;
    double v = local ? 1.0 : 0.0;
    to_dev(partsM_.p + comm_->rank, &v, sizeof(double));
    comm_->allgather(partsM_.p, 1, st_);
    if (comm_->peer()) launch_copy_words(hpin_dev_, partsM_.p, (size_t)world * sizeof(double), st_);   // see to_host
    else XM_HIP_CHECK(hipMemcpyAsync(hpin_, partsM_.p, (size_t)world * sizeof(double), hipMemcpyDeviceToHost, st_));
    XM_HIP_CHECK(hipStreamSynchronize(st_));
    bool any = false;
    for (int r = 0; r < world; ++r) any = any || (hpin_[r] != 0.0);
    return any;
}

// Every host spin loop looks at the stream through this: a sticky error (launch failure, device fault, aborted collective) is
// neither hipSuccess nor hipErrorNotReady and must end the wait with XM_ERR_HIP instead of spinning for ever; so must a wait that
// exceeds the watchdog (XM_WATCHDOG_S seconds, default 600: a dead peer inside an RCCL collective never completes the stream).
bool Context::stream_idle(clk::time_point t_wait, const char *what) {
    const double limit = cfg_.watchdog_s;
    comm_->check_device_error();   // a bounded device-side wait of the peer exchange expired -> XM_ERR_COMM
    const hipError_t q = hipStreamQuery(st_);
    if (q == hipSuccess) return true;
    if (q != hipErrorNotReady) {
        (void)hipGetLastError();
        throw Error(XM_ERR_HIP, std::string("device error while waiting for ") + what + ": " + hipGetErrorString(q));
    }
    if (secs_since(t_wait) > limit)
        throw Error(XM_ERR_HIP, std::string("watchdog: no progress for ") + std::to_string((int)limit) + " s while waiting for " + what);
    return false;
}

// spin on the sequence word of the host-mapped result block (written last by outer_finalize_kernel)
volatile double *Context::wait_outer_result() {
    volatile unsigned long long *hseq = reinterpret_cast<volatile unsigned long long *>(hstat_) + 8 + 5;
    const unsigned long long seq = outer_seq_;
    const auto t_wait = clk::now();
    auto t_check = t_wait;
    while (*hseq != seq) {
        __builtin_ia32_pause();
        if (secs_since(t_check) > 500e-6) {
            if (stream_idle(t_wait, "the outer-iteration results")) {
                XM_HIP_CHECK(hipStreamSynchronize(st_));
                if (*hseq != seq) throw Error(XM_ERR_HIP, "outer-iteration results did not reach host-mapped memory");
            }
            t_check = clk::now();
        }
    }
    return reinterpret_cast<volatile double *>(hstat_) + 8;
}

void Context::drain_events() {
    for (size_t i = 0; i < ev_used_; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev_pool_[i].first, ev_pool_[i].second) == hipSuccess) qw_samples_.push_back(ms);
    }
    ev_used_ = 0;
}

// Sampled launches that turned out to be enqueued-ahead no-ops (a few microseconds) are not Q*W passes: keep the samples
// within a factor 2 of the upper quartile.
void Context::finish_profile() {
    drain_events();
    if (qw_samples_.empty()) return;
    std::vector<float> v = qw_samples_;
    std::sort(v.begin(), v.end());
    const float ref = v[(v.size() * 3) / 4];
    for (float x : v)
        if (x >= 0.5f * ref) { res_->qw_ms_sum += x; res_->qw_ms_count++; }
    qw_samples_.clear();
}

// ------------------------------------------------------------------------------------------------------------------
// truncated CG (trustregion.h:559-664): enqueue-ahead with a host-mapped progress word, no host sync per iteration
// ------------------------------------------------------------------------------------------------------------------
// one tCG iteration on the stream: the Hessian product of iteration i, the lockstep all-gather when the exchange is not fused, cg_step
void Context::tcg_enqueue_iteration(int i, bool profile) {
    const int nA_loc = prod_grid(), nB_loc = tcg_blocks();
    const int rank = comm_->rank;
    const bool fused = xchg_.world > 1;
    const bool lockstep = comm_->active() && !fused;
    double *Wloc = wpad() ? nullptr : W_.p + (size_t)cam0_ * 3 * OP_;   // (see run_tcg)
    const int par = i & 1;
    CamArgs a = cam_args(cur_);
    a.scal = scal_.p + par;
    a.ps = par ? psB_.p : psA_.p;
    a.rs = par ? rsB_.p : rs_.p;
    // tCG partial sums travel in ONE all-gather per iteration: chunk = [Hessian-epilogue partials of this iteration |
    // |r|^2 partials the previous cg_step left in this parity's buffer]
    // and, with more than one rank, in front of them this rank's rows of the image of Hp (see cg_step_kernel): the product
    // input of the next iteration then follows from replicated data and needs no all-gather of its own.
    const size_t mat = (size_t)nloc_ * 3 * OP_;
    const size_t b_off = comm_->active() ? mat : 0;
    const size_t chunk = b_off + (size_t)3 * nA_loc + nB_loc;
    double *pB = fused ? partsB_peer_ : partsB_.p;
    double *pcur = pB + (size_t)par * chunk * comm_->world, *pnext = pB + (size_t)(par ^ 1) * chunk * comm_->world;
    a.partials = pcur + (size_t)rank * chunk + b_off;
    a.Bout = comm_->active() ? pcur + (size_t)rank * chunk : nullptr;
    const bool timed = profile && (hess_launches_ % kProfileStride == 0) && ev_used_ < ev_pool_.size();
    if (timed) XM_HIP_CHECK(hipEventRecord(ev_pool_[ev_used_].first, st_));
    const bool model_rec = opt_ && (opt_->flags & XM_FLAG_MODEL_RECURRENCE);
    sym_rev_ = par;        // consecutive tCG iterations sweep the symmetric matrix in opposite directions (launch_qw_sym)
    product(EPI_HESS, o_, 2.0, a);
    sym_rev_ = 1;          // every other product (gradient, cost): bottom-up, the direction iteration 0 of a tCG does not use
    if (timed) { XM_HIP_CHECK(hipEventRecord(ev_pool_[ev_used_].second, st_)); ev_used_++; }
    hess_launches_++;
    if (lockstep) comm_->allgather(pcur, chunk, st_);
    launch_cg_step(o_, nloc_, scal_.p + par, scal_.p + (par ^ 1), pcur, nA_loc, nB_loc, comm_->world, HpR_.p, Hps_.p, R_.p,
                   s_.p, pR_.p, par ? psB_.p : psA_.p, par ? psA_.p : psB_.p, vR_.p, vs_.p, model_rec ? nullptr : HvR_.p, model_rec ? nullptr : Hvs_.p, rR_.p,
                   par ? rsB_.p : rs_.p, par ? rs_.p : rsB_.p, Wloc, pnext + (size_t)rank * chunk + b_off + 3 * nA_loc, hstat_dev_,
                   (int)b_off, (int64_t)mat, comm_->active() ? Afull_.p : nullptr, W_.p, grouping_, xchg_, st_, wpad());
}

// May the start of the next truncated CG be enqueued behind outer_finalize_kernel before the host knows how the outer iteration ended?
// One GPU, run-ahead polling (not the host-stepped debugging mode), and no product that synchronises with the host inside the tCG.
bool Context::spec_applies() const {
    return !comm_->active() && opt_ && !(opt_->flags & XM_FLAG_HOST_STEPPED) && storage_ != XM_STORAGE_SCHUR && cfg_.debug_drop_finalize < 0;
}

// The host's view is switched to "the candidate was accepted" (R / s swapped, the candidate's gradient state current) for the duration of the
// enqueue: the launches carry the pointers of that world.  tcg_init starts only if the device reached the same verdict (SpecCtl.go), else it
// leaves the scalar block dormant and the iteration(s) behind it return at once.  Returns the iterations enqueued.
int Context::enqueue_spec_tcg() {
    std::swap(R_.p, Rc_.p);
    std::swap(s_.p, sc_.p);
    cur_ ^= 1;
    const PointState &P = ps_[cur_];
    double *Wloc = wpad() ? nullptr : W_.p + (size_t)cam0_ * 3 * OP_;
    ++tcg_seq_;
    launch_tcg_init(o_, nloc_, P.rgR.p, P.rgs.p, R_.p, s_.p, rR_.p, rs_.p, pR_.p, psA_.p, vR_.p, vs_.p, HvR_.p, Hvs_.p, Wloc,
                    scal_.p, 0.0, 0.0, hstat_dev_, st_, wpad(), (int)tcg_seq_, spec_.p);
    // ONE iteration behind tcg_init, and a run-ahead of two in run_tcg: measured against 2 / 3 on one box at Final-13682 size in block CSR
    // (profiles/r05_ab_rome.txt: 27.8 against 28.8 ms per solve; fewer launches that turn out to be no-ops when the tCG ends after ~5
    // iterations) and equal at Venice size
    const int n_spec = 1;
    for (int i = 0; i < n_spec; ++i) tcg_enqueue_iteration(i, false);   // (not sampled by the HIP-event profile: they may turn out dormant)
    std::swap(R_.p, Rc_.p);
    std::swap(s_.p, sc_.p);
    cur_ ^= 1;
    return n_spec;
}

int Context::run_tcg(double rr, double delta, TcgScal &fin, int adopted) {
    const bool stepped = (opt_->flags & XM_FLAG_HOST_STEPPED) != 0;
    const bool profile = (opt_->flags & XM_FLAG_PROFILE_QW) != 0;
    // with the padded copy on (single rank, sliced ELL) nothing reads the native-pitch product input inside the tCG: the main launch
    // gathers from the copy and the second launch rebuilds the diagonal term from its own operands -- the kernels skip those 7.2 MB of
    // stores per iteration at 100 k cameras
    double *Wloc = wpad() ? nullptr : W_.p + (size_t)cam0_ * 3 * OP_;
    const PointState &P = ps_[cur_];
    if (comm_->active()) comm_->note("tcg_start", rr, delta);
    if (adopted == 0) {
        ++tcg_seq_;
        launch_tcg_init(o_, nloc_, P.rgR.p, P.rgs.p, R_.p, s_.p, rR_.p, rs_.p, pR_.p, psA_.p, vR_.p, vs_.p, HvR_.p, Hvs_.p, Wloc,
                        scal_.p, rr, delta, hstat_dev_, st_, wpad(), (int)tcg_seq_);
        gather_W();
    }
    // Iterations in flight ahead of the last one seen finished; the excess become no-op launches.  With a communicator the
    // loop must issue the SAME number of collectives on every rank although ranks poll at different moments: iteration j is
    // enqueued only once iteration j-2 is confirmed (so nobody can be past T+1 when the tCG ends in iteration T) and every
    // rank tops its queue up to exactly T+2 iterations after it has seen the end (identical scalar state on all ranks =>
    // identical T).  One iteration is always queued behind the running one, so the host round trip is hidden.
    // With the DIRECT PEER EXCHANGE (xchg_.world > 1) the exchange lives inside cg_step_kernel and launches past the end of the tCG
    // skip it, so there is nothing to keep in lockstep: the loop is the single-GPU loop.
    const bool fused = xchg_.world > 1;
    if (fused) xchg_.epoch_base = (++tcg_runs_) << 12;
    const bool lockstep = comm_->active() && !fused;
    // (single GPU: two iterations ahead, three for the smallest problems, whose iteration is shorter than the host's two launches)
    const int run_ahead = (lockstep || prod_grid() >= 64) ? 2 : 3;
    int fin_status = 0, fin_iter = 0;
    int it = adopted;  // iterations enqueued
    auto enqueue = [&](int i) { tcg_enqueue_iteration(i, profile); };
    auto read_scal = [&](int par) {
        TcgScal sc;
        to_host(&sc, scal_.p + par, sizeof(TcgScal));
        return sc;
    };
    if (stepped) {
        for (;;) {
            TcgScal sc = read_scal(it & 1);
            if (sc.status != 0 || it >= kMaxInner) break;
            enqueue(it++);
        }
    } else {
        // progress word = [run : 32 | iteration : 24 | status : 8], written by tcg_init and every cg_step of THIS run (TcgScal.seq): a word
        // of an earlier run -- the previous tCG's last one, until this run's tcg_init has executed -- is not progress of this one
        volatile unsigned long long *hs = hstat_;
        auto last_progress = clk::now();
        auto last_change = last_progress;   // watchdog reference: the progress word last moved here
        unsigned long long seen = ~0ull;
        for (;;) {
            const unsigned long long v = *hs;
            if (v != seen) { seen = v; last_progress = clk::now(); last_change = last_progress; }
            const bool valid = (unsigned int)(v >> 32) == tcg_seq_;
            const int status = valid ? (int)(v & 0xff) : 0;
            const int done = valid ? (int)((v >> 8) & 0xffffff) : 0;
            if (status != 0) { fin_status = status; fin_iter = done; break; }
            if (it < kMaxInner && it - done < run_ahead) { enqueue(it++); continue; }
            __builtin_ia32_pause();
            if (secs_since(last_progress) > 200e-6) {
                // Nothing new for a while: if the stream has drained the progress word is stale (or this platform does not
                // make device writes to mapped host memory visible promptly) -> read the truth from the device.
                if (stream_idle(last_change, "the truncated-CG progress word")) {
                    TcgScal sc = read_scal(it & 1);
                    if (sc.status != 0) { fin_status = sc.status; fin_iter = sc.iter; break; }
                    if (it >= kMaxInner) { fin_status = 6; fin_iter = kMaxInner; break; }
                    *hstat_ = ((unsigned long long)tcg_seq_ << 32) | ((unsigned long long)(unsigned)sc.iter << 8);
                    if (it - sc.iter >= run_ahead) enqueue(it++);  // cannot happen, but never stall
                }
                last_progress = clk::now();
            }
        }
    }
    if (comm_->active()) comm_->note("tcg_end_before_topup", (double)it, (double)(fin_iter * 16 + fin_status));
    if (fin_status == 7) { comm_->check_device_error(); throw Error(XM_ERR_COMM, "peer exchange: a rank did not arrive inside the truncated CG"); }
    if (lockstep && !stepped) {
        const int T = (fin_status == 6) ? fin_iter - 1 : fin_iter;   // iteration in which the tCG ended
        while (it < std::min(kMaxInner, T + 2)) enqueue(it++);      // no-ops, but the same collectives on every rank
    }
    (void)fin;
    return it;  // the final scalar block is scal_[it & 1]; the caller fetches it together with the other results
}

// ------------------------------------------------------------------------------------------------------------------
// Riemannian trust region (trustregion.h:77-724).  On entry R_/s_ hold the initial point.
// ------------------------------------------------------------------------------------------------------------------
TrResult Context::trust_region(int o, double &gradtol, double linesearch_step, const std::vector<double> &v_dir, double max_time) {
    TrResult out;
    const double dim = (double)n_ * (3.0 * o - 6.0) + (double)n_ - 1.0;  // trustregion.h:104
    const double delta_bar = std::sqrt(dim);
    double delta = delta_bar / 8.0;
    double *Wloc = W_.p + (size_t)cam0_ * 3 * OP_;
    double f = 0, rr = 0;

    log("start linesearch\n");
    if (linesearch_step != 0) {  // trustregion.h:360-408
        launch_scale_rows(o, nloc_, R_.p, s_.p, Wloc, st_);
        gather_W();
        double f0, tmp;
        eval_point(cur_ ^ 1, R_.p, s_.p, f0, tmp);
        double alpha = linesearch_step;
        // direction: v in the new last column only
        std::vector<double> D((size_t)nloc_ * 3 * OP_, 0.0);
        for (int c = 0; c < (int)true_loc_; ++c) {
            const int64_t g = g0_ + c;
            for (int a = 0; a < 3; ++a) D[((size_t)c * 3 + a) * OP_ + (o - 1)] = v_dir[(size_t)(3 * g + a)];
        }
        to_dev(D_.p, D.data(), D.size() * sizeof(double));
        double fn;
        for (;;) {
            launch_retract(o, nloc_, cam0_, R_.p, s_.p, D_.p, nullptr, -alpha, Rc_.p, nullptr, Wloc, st_, retraction_);
            gather_W();
            eval_point(cur_ ^ 1, Rc_.p, s_.p, fn, tmp);
            if (!(fn > f0)) break;
            alpha *= 0.5;
            if (alpha < 1e-20) { log("linesearch failed! BM stopped! \n"); out.primal = -1; out.ls_failed = true; out.stop_reason = -1; return out; }
        }
        if (f0 - fn > 0) {
            log("linesearch decrease %1.3e\n", f0 - fn);
            if (opt_->flags & XM_FLAG_FIX_STALE_SR) {
                std::swap(R_.p, Rc_.p);
                launch_scale_rows(o, nloc_, R_.p, s_.p, Wloc, st_);
            } else {
                // the reference keeps sR of the point BEFORE the line search for the first cost/gradient
                // (trustregion.h:394-422: R is replaced, sR is not)
                launch_scale_rows(o, nloc_, R_.p, s_.p, Wloc, st_);
                std::swap(R_.p, Rc_.p);
            }
            gather_W();
        } else {
            log("linesearch failed! BM stopped! \n");
            out.primal = -1; out.ls_failed = true; out.stop_reason = -1;
            return out;
        }
    } else {
        launch_scale_rows(o, nloc_, R_.p, s_.p, Wloc, st_);
        gather_W();
    }
    eval_point(cur_, R_.p, s_.p, f, rr);  // loss[0] and the gradient state of the first outer iteration
    if (device_outer_applies(o)) return trust_region_device(o, gradtol, f, rr, delta, delta_bar, max_time);
    double loss = f;

    int endreason = 6, trstatus = 4, shrink_count = 0, inner_print = 1, k = 0;
    int adopted = 0;   // iterations of the NEXT truncated CG that are already running (enqueue_spec_tcg), adopted by the decision below
    long long totalite = 0;
    int stop_reason = 14;
    const auto start = clk::now();
    for (k = 0; k < kMaxOuter; ++k) {
        const double gradnorm = std::sqrt(rr);
        if (verbose_) {
            static const char *trn[] = {"", "TR- ", "TR+ ", "REJ ", "TR "};
            if (k > 0) log("%s", trn[trstatus]);
            log("%d   %d   %1.3e   %1.3e", k, inner_print, loss, gradnorm);
            if (k > 0) {
                const char *er = endreason == 1 ? "nagative curvature" : endreason == 2 ? "exceed trust region"
                               : endreason == 3 ? "reached norm tolerance" : endreason == 5 ? "numerical issue" : "max iteration";
                log("   %s\n", er);
            } else log("\n");
        }
        if (opt_->trace && res_->trace_len < opt_->trace_cap) {
            double *tr = opt_->trace + (size_t)res_->trace_len * 6;
            tr[0] = loss; tr[1] = gradnorm; tr[2] = inner_print; tr[3] = endreason; tr[4] = trstatus; tr[5] = delta;
            res_->trace_len++;
        }
        if (endreason == 5) { stop_reason = 5; log("Terminate because of rdotr touched machine precise\n"); break; }
        if (gradnorm < gradtol) { log("Terminate because of small gradient norm\n"); gradtol /= 10; stop_reason = 10; break; }
        if (agree_any((double)(long long)secs_since(start) > max_time)) { log("Terminate because of time limit\n"); stop_reason = 11; break; }
        endreason = 6; trstatus = 4;

        TcgScal fin;
        const int enq = run_tcg(rr, delta, fin, adopted);
        adopted = 0;

        // model decrease, retraction and the candidate's cost/gradient are enqueued right behind the tCG and fetched with
        // ONE synchronisation (trustregion.h:667-678 needs four blocking reads + a sync here)
        const int nA_loc = prod_grid();
        const PointState &P = ps_[cur_];
        // the step's model decrease comes out of the retraction launch (the anchor's scale part of the step is zero by construction, as in
        // the flat kernel this replaces); the quad-per-camera form of the retraction is a micro-benchmark alternative only
        const int nB_loc = retract_grid(nloc_);
        const bool model_rec = (opt_->flags & XM_FLAG_MODEL_RECURRENCE) != 0;   // the model value sits in the tCG's scalar block: no sums, no gather
        launch_retract_model(o, nloc_, cam0_, R_.p, s_.p, vR_.p, vs_.p, Rc_.p, sc_.p, Wloc, wpad(), model_rec ? nullptr : HvR_.p, Hvs_.p, P.rgR.p, P.rgs.p,
                             partsM_.p + (size_t)comm_->rank * nB_loc, st_, retraction_ == XM_RETRACT_POLAR ? 1 : 0);
        if (comm_->active() && !model_rec) comm_->allgather(partsM_.p, (size_t)nB_loc, st_);
        gather_W();
        wpad_next_ = wpad();   // the gradient product below may gather from the padded copy the retraction has just written
        {
            CamArgs a = cam_args(cur_ ^ 1);
            a.R = Rc_.p;
            a.s = sc_.p;
            a.partials = partsA_.p + (size_t)comm_->rank * 2 * nA_loc;
            product(EPI_GRAD, o_, 2.0, a);
            if (comm_->active()) comm_->allgather(partsA_.p, (size_t)2 * nA_loc, st_);
        }
        // XM_DEBUG_DROP_FINALIZE=k (tests): the k-th outer iteration "loses" its result kernel, as a failed launch would: the host
        // must come back with XM_ERR_HIP from wait_outer_result() instead of spinning on a sequence word that never arrives
        const long long drop_at = cfg_.debug_drop_finalize;
        // Single GPU: the result kernel also evaluates the trust-region update and, when the step is accepted and the solve goes on, releases
        // the start of the NEXT truncated CG, enqueued right behind it (SpecCtl): the GPU works through the host's round trip
        const bool spec = spec_applies();
        const OuterArgs oa = {loss, delta, delta_bar, gradtol, shrink_count, (k + 1 >= kMaxOuter) ? 1 : 0};
        if (drop_at >= 0 && (long long)outer_seq_ + 1 == drop_at) ++outer_seq_;
        else
        launch_outer_finalize(partsA_.p, nA_loc, comm_->world, partsM_.p, model_rec ? 0 : nB_loc * comm_->world, scal_.p + (enq & 1),
                              reinterpret_cast<double *>(hstat_dev_) + 8, ++outer_seq_, grouping_, st_, spec ? &oa : nullptr, spec_.p);
        const int n_spec = spec ? enqueue_spec_tcg() : 0;
        volatile double *hres = wait_outer_result();
        double f_new = hres[0], rr_new = hres[1], loss_qu = hres[2];
        const bool spec_go = spec && hres[6] != 0.0;
        const double spec_delta = hres[7];
        fin.status = (int)hres[3];
        fin.iter = (int)hres[4];
        if (fin.status == 7) { comm_->check_device_error(); throw Error(XM_ERR_COMM, "peer exchange: a rank did not arrive inside the truncated CG"); }
        if (fin.status == 0) fin.status = 6;  // ran out of iterations
        endreason = fin.status;
        inner_print = fin.iter + 1;
        totalite += fin.iter + 1;
        if (opt_->flags & XM_FLAG_PROFILE_QW) drain_events();
        if (loss_qu >= 0) { log("error! loss_qu is larger than 0\n"); stop_reason = 12; break; }
        const double rou = (f_new - loss) / loss_qu;  // trustregion.h:680-701
        if (rou < 0.25) { delta *= 0.25; trstatus = 1; shrink_count++; }
        else if (rou > 0.75 && endreason <= 2) { delta = std::min(delta * 2, delta_bar); trstatus = 2; shrink_count = 0; }
        else shrink_count = 0;
        bool stop_delta = false;
        if (shrink_count > 3) {
            delta *= 1e-3; shrink_count = 0;
            log("delta shrinked to %1.3e\n", delta);
            if (delta < 1e-20) { log("delta is too small, BM stopped!\n"); stop_delta = true; }
        }
        const bool reject = (f_new > loss || rou < 0.1);  // trustregion.h:702
        if (stop_delta || !reject) {
            std::swap(R_.p, Rc_.p);
            std::swap(s_.p, sc_.p);
            cur_ ^= 1;
            if (stop_delta) { stop_reason = 13; break; }  // the reference leaves the new point in place but reports loss[k]
            loss = f_new;
            rr = rr_new;
            // the device came to the same verdict from the same numbers: the tCG it has started from this point IS the next one
            if (spec_go && std::memcmp(&spec_delta, &delta, sizeof(double)) == 0) adopted = n_spec;
        } else {
            trstatus = 3;  // keep point, state, loss and rr
        }
        // Speculative launches that were not adopted found a dormant scalar block and returned at once: they are no products.
        // Invariant for the ones that WERE adopted when the loop is left right after (time limit, |grad| < gradtol by a rounding of the sqrt): a live
        // tcg_init + iteration from the new point may still be in the queue; they write only the tCG's own vectors (v, Hv, r, p, W and the
        // padded copy), which every later user re-creates before reading (certificate and line search rebuild W; tcg_seq_ guards the progress word).
        if (n_spec > 0 && adopted != n_spec && res_) res_->qw_products -= n_spec;
    }
    log("\nTotal iteration:     %lld\n", totalite);
    const double secs = secs_since(start);
    log("Time taken by function1: %lld ms\n", (long long)(secs * 1e3));
    res_->tcg_iters += totalite;
    res_->outer_iters += k;
    res_->tr_seconds += secs;
    res_->last_stop_reason = stop_reason;
    out.primal = loss;
    out.outer_iters = k;
    out.stop_reason = stop_reason;
    return out;
}

// ------------------------------------------------------------------------------------------------------------------
// The same trust region with the OUTER ITERATION ON THE DEVICE (single GPU; dense, symmetric-dense and block-CSR products).
// The host's part: the first cost / gradient (done by the caller), the tests at the top of outer iteration 0, the initial scalar block, and
// then one and the same pair of launches -- product(EPI_AUTO), outer_step -- enqueued `ahead` pairs in front of the progress word, until
// the word says PH_STOP.  Everything trustregion.h:527-708 decides between two truncated CGs is decided by outer_step_kernel from the same
// partial sums with the same formulas as trust_region() above; the candidate is COPIED over the current point when it is accepted, so R_ / s_ /
// ps_[cur_] are the current point before and after, and no buffer role ever depends on what the device decided.
// Buffers with two parity copies (tCG scalar block, scale parts of p and r, partial sums) go by the parity of the SLOT (the pair's index),
// and so does the sweep direction of the dense products: every launch of the sequence is fixed when it is enqueued.
// ------------------------------------------------------------------------------------------------------------------
bool Context::device_outer_applies(int o) const {
    if (!opt_ || (opt_->flags & (XM_FLAG_HOST_OUTER | XM_FLAG_HOST_STEPPED))) return false;
    if (comm_->active() || symw_ || storage_ == XM_STORAGE_SCHUR || o < 3 || cfg_.debug_drop_finalize >= 0) return false;
    if (storage_ == XM_STORAGE_BSR3 && sell_ && sell_supports(o)) return false;   // (the sliced-ELL pair of launches has no EPI_AUTO form)
    if (storage_ == XM_STORAGE_DENSE && ks_ > 1) return false;
    // dense products: the host-driven form is 1.5-2 % faster (profiles/r06_ab_outer.txt: 41.7 against 42.4 us per tCG iteration on the headline) --
    // the device-driven one on request; block-CSR products: the device-driven form is the faster one (34.7 against 35.5 us at 13 682 cameras)
    if (storage_ == XM_STORAGE_DENSE && !(opt_->flags & XM_FLAG_DEVICE_OUTER)) return false;
    return true;
}

TrResult Context::trust_region_device(int o, double &gradtol, double f, double rr, double delta, double delta_bar, double max_time) {
    TrResult out;
    const auto start = clk::now();
    const bool profile = (opt_->flags & XM_FLAG_PROFILE_QW) != 0;
    const bool model_rec = (opt_->flags & XM_FLAG_MODEL_RECURRENCE) != 0;
    const int nA = prod_grid(), nB = tcg_blocks(), nM = retract_grid(nloc_);
    double loss = f;
    int stop_reason = 14, k = 0;
    long long totalite = 0;
    int trace0 = res_->trace_len;   // the trace of this trust region starts here (the staircase appends stage after stage)
    if (opt_->trace && res_->trace_len < opt_->trace_cap) {
        double *tr = opt_->trace + (size_t)res_->trace_len * 6;
        tr[0] = loss; tr[1] = std::sqrt(rr); tr[2] = 1; tr[3] = 6; tr[4] = 4; tr[5] = delta;
        res_->trace_len++;
    }
    // XM_FLAG_VERBOSE: the reference's progress lines (trustregion.h:504-525) are printed from the trace records -- the same text, but a
    // stage's lines appear together when its trust region has ended instead of one by one
    auto progress_line = [&](int kk, const double *rec, double delta_prev) {
        static const char *trn[] = {"", "TR- ", "TR+ ", "REJ ", "TR "};
        const int er = (int)rec[3], ts = (int)rec[4];
        if (kk > 0 && rec[5] < delta_prev * 0.25 * 0.5) log("delta shrinked to %1.3e\n", rec[5]);   // (the 1e-3 cut after four shrinks in a row)
        if (kk > 0) log("%s", trn[(ts >= 0 && ts <= 4) ? ts : 0]);
        log("%d   %d   %1.3e   %1.3e", kk, (int)rec[2], rec[0], rec[1]);
        if (kk > 0) log("   %s\n", er == 1 ? "nagative curvature" : er == 2 ? "exceed trust region" : er == 3 ? "reached norm tolerance" : er == 5 ? "numerical issue" : "max iteration");
        else log("\n");
    };
    {
        const double rec0[6] = {loss, std::sqrt(rr), 1, 6, 4, delta};
        progress_line(0, rec0, delta);
    }
    bool run = true;
    if (std::sqrt(rr) < gradtol) { stop_reason = 10; run = false; }
    else if ((double)(long long)secs_since(start) > max_time) { stop_reason = 11; run = false; }
    long long slots_live = 0, slots_enq = 0;
    if (run) {
        if (trace_dev_.count < (size_t)kMaxOuter * 6) trace_dev_.alloc((size_t)kMaxOuter * 6);
        if (stop_req_.count < 1) stop_req_.alloc(1);
        XM_HIP_CHECK(hipMemsetAsync(stop_req_.p, 0, sizeof(int), st_));
        if (oscal_.count < 2) oscal_.alloc(2);
        TcgScal init;
        std::memset(&init, 0, sizeof(init));
        init.phase = PH_INIT; init.delta = delta; init.seq = (int)++tcg_seq_;
        OuterScal oinit;
        std::memset(&oinit, 0, sizeof(oinit));
        oinit.loss = loss; oinit.rr_point = rr;
        to_dev(scal_.p + 1, &init, sizeof(init));   // slot -1 has parity 1
        to_dev(oscal_.p + 1, &oinit, sizeof(oinit));
        const unsigned int runid = ++outer_run_;
        volatile unsigned long long *hp = hstat_ + 24;
        *hp = 0;
        const size_t chunk = (size_t)3 * nA + nB;
        double *Wloc = W_.p + (size_t)cam0_ * 3 * OP_;
        const PointState &Pc = ps_[cur_], &Pn = ps_[cur_ ^ 1];
        auto step_args = [&](int slot) {
            const int par = slot & 1;
            OuterStepArgs A;
            std::memset(&A, 0, sizeof(A));
            A.nloc = nloc_; A.cam0 = cam0_;
            A.scal_cur = scal_.p + par; A.scal_next = scal_.p + (par ^ 1);
            A.os_cur = oscal_.p + par; A.os_next = oscal_.p + (par ^ 1);
            A.parts = partsB_.p + (size_t)par * chunk;
            A.partsB_out = partsB_.p + (size_t)(par ^ 1) * chunk + (size_t)3 * nA;
            A.nA = nA; A.nB = nB;
            A.HpR = HpR_.p; A.Hps = Hps_.p;
            A.R = R_.p; A.s = s_.p; A.Rc = Rc_.p; A.sc = sc_.p;
            A.pR = pR_.p; A.ps_cur = par ? psB_.p : psA_.p; A.ps_next = par ? psA_.p : psB_.p;
            A.vR = vR_.p; A.vs = vs_.p; A.HvR = model_rec ? nullptr : HvR_.p; A.Hvs = model_rec ? nullptr : Hvs_.p; A.rR = rR_.p;
            A.rs_cur = par ? rsB_.p : rs_.p; A.rs_next = par ? rs_.p : rsB_.p;
            A.Wloc = Wloc; A.Wpad = nullptr;
            A.cur = {Pc.G.p, Pc.egs.p, Pc.S0.p, Pc.rgR.p, Pc.rgs.p};
            A.cand = {Pn.G.p, Pn.egs.p, Pn.S0.p, Pn.rgR.p, Pn.rgs.p};
            A.partsA = partsA_.p; A.partsM = partsM_.p; A.nM = nM;
            A.delta_bar = delta_bar; A.gradtol = gradtol; A.max_outer = kMaxOuter;
            A.trace = trace_dev_.p; A.trace_cap = kMaxOuter;
            A.stop_req = stop_req_.p;
            A.hprog = hstat_dev_ + 24;
            A.run = runid; A.slot = slot; A.grp = grouping_;
            return A;
        };
        const int polar = retraction_ == XM_RETRACT_POLAR ? 1 : 0;
        auto enqueue_slot = [&](int slot) {
            const int par = slot & 1;
            CamArgs a = cam_args(cur_);
            a.scal = scal_.p + par;
            a.ps = par ? psB_.p : psA_.p;
            a.rs = par ? rsB_.p : rs_.p;
            a.partials = partsB_.p + (size_t)par * chunk;
            a.cand.R = Rc_.p; a.cand.s = sc_.p;
            a.cand.G = Pn.G.p; a.cand.egs = Pn.egs.p; a.cand.S0 = Pn.S0.p; a.cand.rgR = Pn.rgR.p; a.cand.rgs = Pn.rgs.p;
            a.cand.partials = partsA_.p;
            // sampled like the host-driven loop's Hessian launches; a pair in the gradient role moves the same bytes, a drained pair after the
            // end is dropped by finish_profile() like a run-ahead no-op
            const bool timed = profile && (hess_launches_ % kProfileStride == 0) && ev_used_ < ev_pool_.size();
            if (timed) XM_HIP_CHECK(hipEventRecord(ev_pool_[ev_used_].first, st_));
            sym_rev_ = par;
            product(EPI_AUTO, o_, 2.0, a);
            sym_rev_ = 1;
            if (timed) { XM_HIP_CHECK(hipEventRecord(ev_pool_[ev_used_].second, st_)); ev_used_++; }
            hess_launches_++;
            launch_outer_step(o_, polar, step_args(slot), nB, st_);
        };
        launch_outer_step(o_, polar, step_args(-1), nB, st_);
        const int ahead = 4;
        int enq = 0;
        bool time_sent = false;
        auto last_progress = clk::now();
        auto last_change = last_progress;
        unsigned long long seen = ~0ull;
        for (;;) {
            const unsigned long long v = *hp;
            if (v != seen) { seen = v; last_progress = clk::now(); last_change = last_progress; }
            const bool valid = (unsigned int)(v >> 32) == runid;
            const int done = valid ? (int)((v >> 8) & 0xffffff) : 0;
            const int phase = valid ? (int)(v & 0xff) : (int)PH_INIT;
            if (phase == PH_STOP) break;
            if (enq - done < ahead) { enqueue_slot(enq++); continue; }
            __builtin_ia32_pause();
            if (!time_sent && (double)(long long)secs_since(start) > max_time) {
                XM_HIP_CHECK(hipMemsetAsync(stop_req_.p, 1, sizeof(int), st_));   // (any non-zero pattern; lands within `ahead` pairs)
                time_sent = true;
            }
            if (secs_since(last_progress) > 200e-6) {
                // nothing new for a while: if the stream has drained the word is stale (or device writes to mapped host memory are not
                // visible promptly on this platform) -> read the truth from the device
                if (stream_idle(last_change, "the outer-iteration progress word")) {
                    TcgScal sc;
                    to_host(&sc, scal_.p + (enq & 1), sizeof(TcgScal));
                    if (sc.phase == PH_STOP) break;
                    *hp = ((unsigned long long)runid << 32) | ((unsigned long long)(unsigned)enq << 8) | (unsigned long long)(unsigned)sc.phase;
                }
                last_progress = clk::now();
            }
        }
        XM_HIP_CHECK(hipStreamSynchronize(st_));
        TcgScal fin;
        OuterScal ofin;
        to_host(&fin, scal_.p + (enq & 1), sizeof(TcgScal));
        to_host(&ofin, oscal_.p + (enq & 1), sizeof(OuterScal));
        if (fin.phase != PH_STOP) throw Error(XM_ERR_HIP, "device-driven outer iteration: the final scalar block is not in the stop phase");
        k = ofin.k; totalite = ofin.totalite; loss = ofin.loss; stop_reason = ofin.stop_reason;
        slots_live = ofin.slots; slots_enq = enq;
        // trace records 1 .. k (record k exists when iteration k's top was reached: always, except when the iteration cap ended the loop)
        const int last_rec = std::min(k, kMaxOuter - 1);
        if (opt_->trace && last_rec >= 1 && res_->trace_len < opt_->trace_cap) {
            const int take = std::min(last_rec, opt_->trace_cap - res_->trace_len);
            to_host(opt_->trace + (size_t)res_->trace_len * 6, trace_dev_.p + 6, (size_t)take * 6 * sizeof(double));
            res_->trace_len += take;
        }
        if (verbose_ && last_rec >= 1) {
            std::vector<double> recs((size_t)last_rec * 6);
            to_host(recs.data(), trace_dev_.p + 6, recs.size() * sizeof(double));
            double dprev = delta;
            for (int kk = 1; kk <= last_rec; ++kk) { progress_line(kk, &recs[(size_t)(kk - 1) * 6], dprev); dprev = recs[(size_t)(kk - 1) * 6 + 5]; }
        }
        if (profile) drain_events();
        res_->qw_products -= (slots_enq - slots_live);   // pairs drained after the end were no products
    }
    (void)trace0;
    if (stop_reason == 5) log("Terminate because of rdotr touched machine precise\n");
    else if (stop_reason == 10) log("Terminate because of small gradient norm\n");
    else if (stop_reason == 11) log("Terminate because of time limit\n");
    else if (stop_reason == 12) log("error! loss_qu is larger than 0\n");
    else if (stop_reason == 13) log("delta is too small, BM stopped!\n");
    log("\nTotal iteration:     %lld\n", totalite);
    log("Time taken by function1: %lld ms\n", (long long)(secs_since(start) * 1e3));
    if (stop_reason == 10) gradtol /= 10;
    const double secs = secs_since(start);
    res_->outer_on_device++;
    res_->tcg_iters += totalite;
    res_->outer_iters += k;
    res_->tr_seconds += secs;
    res_->last_stop_reason = stop_reason;
    out.primal = loss;
    out.outer_iters = k;
    out.stop_reason = stop_reason;
    return out;
}

// ------------------------------------------------------------------------------------------------------------------
// smallest eigenpair of a symmetric tridiagonal matrix: Sturm bisection + inverse iteration
// ------------------------------------------------------------------------------------------------------------------
static int sturm_count(const std::vector<double> &a, const std::vector<double> &b, int m, double x) {
    int cnt = 0;
    double d = a[0] - x;
    if (d < 0) cnt++;
    for (int i = 1; i < m; ++i) {
        if (d == 0) d = 1e-300;
        d = (a[(size_t)i] - x) - b[(size_t)i - 1] * b[(size_t)i - 1] / d;
        if (d < 0) cnt++;
    }
    return cnt;
}
static void tridiag_min(const std::vector<double> &a, const std::vector<double> &b, int m, double &theta, std::vector<double> &y,
                        double &tmax) {
    double lo = a[0], hi = a[0];
    for (int i = 0; i < m; ++i) {
        const double r = (i > 0 ? std::fabs(b[(size_t)i - 1]) : 0.0) + (i < m - 1 ? std::fabs(b[(size_t)i]) : 0.0);
        lo = std::min(lo, a[(size_t)i] - r);
        hi = std::max(hi, a[(size_t)i] + r);
    }
    tmax = std::max(std::fabs(lo), std::fabs(hi));
    double l = lo, h = hi;
    for (int it = 0; it < 200 && h - l > 4e-16 * std::max(1.0, tmax); ++it) {
        const double mid = 0.5 * (l + h);
        if (sturm_count(a, b, m, mid) >= 1) h = mid; else l = mid;
    }
    theta = 0.5 * (l + h);
    // inverse iteration: (T - theta I) y = rhs, tridiagonal LU with partial pivoting
    y.assign((size_t)m, 1.0 / std::sqrt((double)m));
    if (m == 1) { y[0] = 1.0; return; }
    const double shift = theta - 1e-14 * std::max(1.0, tmax);
    std::vector<double> dl((size_t)m), dd((size_t)m), du((size_t)m), du2((size_t)m), rhs((size_t)m);
    for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < m; ++i) {
            dd[(size_t)i] = a[(size_t)i] - shift;
            du[(size_t)i] = (i < m - 1) ? b[(size_t)i] : 0.0;
            dl[(size_t)i] = (i < m - 1) ? b[(size_t)i] : 0.0;  // dl[i] = T(i+1,i)
            du2[(size_t)i] = 0.0;
            rhs[(size_t)i] = y[(size_t)i];
        }
        for (int i = 0; i < m - 1; ++i) {
            if (std::fabs(dd[(size_t)i]) >= std::fabs(dl[(size_t)i])) {
                if (dd[(size_t)i] == 0) dd[(size_t)i] = 1e-300;
                const double f = dl[(size_t)i] / dd[(size_t)i];
                dd[(size_t)i + 1] -= f * du[(size_t)i];
                rhs[(size_t)i + 1] -= f * rhs[(size_t)i];
            } else {  // swap rows i and i+1
                const double f = dd[(size_t)i] / dl[(size_t)i];
                const double t_dd = dd[(size_t)i + 1], t_du = du[(size_t)i + 1];
                dd[(size_t)i] = dl[(size_t)i];
                const double old_du = du[(size_t)i];
                du[(size_t)i] = t_dd;
                du2[(size_t)i] = (i < m - 2) ? t_du : 0.0;
                dd[(size_t)i + 1] = old_du - f * t_dd;
                if (i < m - 2) du[(size_t)i + 1] = -f * t_du;
                const double tr = rhs[(size_t)i];
                rhs[(size_t)i] = rhs[(size_t)i + 1];
                rhs[(size_t)i + 1] = tr - f * rhs[(size_t)i + 1];
            }
        }
        if (dd[(size_t)m - 1] == 0) dd[(size_t)m - 1] = 1e-300;
        y[(size_t)m - 1] = rhs[(size_t)m - 1] / dd[(size_t)m - 1];
        y[(size_t)m - 2] = (rhs[(size_t)m - 2] - du[(size_t)m - 2] * y[(size_t)m - 1]) / dd[(size_t)m - 2];
        for (int i = m - 3; i >= 0; --i)
            y[(size_t)i] = (rhs[(size_t)i] - du[(size_t)i] * y[(size_t)i + 1] - du2[(size_t)i] * y[(size_t)i + 2]) / dd[(size_t)i];
        double nrm = 0;
        for (int i = 0; i < m; ++i) nrm += y[(size_t)i] * y[(size_t)i];
        nrm = std::sqrt(nrm);
        if (!(nrm > 0) || !std::isfinite(nrm)) { y.assign((size_t)m, 0.0); y[0] = 1.0; break; }
        for (int i = 0; i < m; ++i) y[(size_t)i] /= nrm;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// lambda_min and its eigenvector of S = Q + diag(dz) - blockdiag(Lam) by Lanczos with full re-orthogonalisation.
// Replaces cusolverDnXsyevd on the 3n x 3n certificate matrix (checkeig.h:303-318, Dense/eig.h:35-73), O((3n)^3),
// by products with the same Q*W kernel (rank-1 input).  Lam_/dz live in ps_[cur^1].S0 / .egs (free at this point).
// ------------------------------------------------------------------------------------------------------------------
int Context::lanczos_min(std::vector<double> &x_out, double &theta_out, int &iters_out, double &resid_out) {
    const int64_t len = ld_;           // vectors are replicated, full length, zero beyond 3n
    const int64_t m3 = 3 * n_;
    const int env_mmax = cfg_.lanczos_mmax, env_restarts = cfg_.lanczos_restarts;   // small values: debugging aids (the tests force a non-converged run)
    const int mmax = (int)std::min<int64_t>(m3, std::max(2, env_mmax));
    // Small problems take the DENSE route of the reference (cusolverDnDsyevd on S, Dense/eig.h:35-73): Householder-free, the same
    // three-term recurrence with full re-orthogonalisation IS the reduction of S to tridiagonal form when it runs all 3n steps, and
    // the QL sweep on T then returns the smallest eigenvalue of S itself (no convergence test, Ritz residual = round-off).
    const bool exact = m3 <= cfg_.cert_dense_rows && m3 <= mmax;
    eig_exact_ = false;   // set after the run: only when the Krylov space was really exhausted
    // Lanczos workspace: owned by the context and grow-only, so that no device memory is freed between two collectives of a solve
    // (see setup_rank); the host barrier covers the (re)allocation
    DevBuf<double> &V = lzV_, &c = lzc_, &w = lzw_, &c2 = lzc2_, &ab = lzab_, &dscr = lzscr_;
    const size_t nscr = (size_t)(mmax + 1) * (size_t)dots_multi_segments(len);
    if (V.count < (size_t)len * (mmax + 1) || c.count < (size_t)mmax + 2 || w.count < (size_t)len || dscr.count < nscr) {
        if (comm_->active()) comm_->host_barrier();
        V.alloc((size_t)len * (mmax + 1));
        c.alloc((size_t)mmax + 2);
        w.alloc((size_t)len);
        c2.alloc((size_t)mmax + 2);
        dscr.alloc(nscr);
        ab.alloc((size_t)2 * (mmax + 1));
        if (comm_->active()) comm_->host_barrier();
    }
    std::vector<double> x((size_t)len, 0.0);
    unsigned long long lcg = 0x9E3779B97F4A7C15ull;
    double nrm = 0;
    std::vector<int64_t> posv((size_t)n_);   // global camera -> position (identity unless the partition is balanced by blocks)
    for (int64_t g = 0; g < n_; ++g) posv[(size_t)g] = pos_of(g);
    for (int64_t i = 0; i < m3; ++i) {   // drawn in GLOBAL order: the start vector does not depend on the partition
        lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
        const double v = ((double)(lcg >> 11) / 9007199254740992.0) - 0.5;
        x[(size_t)(3 * posv[(size_t)(i / 3)] + i % 3)] = v;
        nrm += v * v;
    }
    nrm = std::sqrt(nrm);
    for (int64_t i = 0; i < len; ++i) x[(size_t)i] /= nrm;

    CamArgs a = cam_args(cur_);
    a.Lam = ps_[cur_ ^ 1].S0.p;
    a.dz = ps_[cur_ ^ 1].egs.p;
    double theta = 0, resid = 1e300, tmax = 1;
    std::vector<double> al, be, y;
    int total = 0;
    // device-side bookkeeping: c1 / c2 = Gram-Schmidt coefficients of the two passes, ab = [alpha_0.. | beta_0..]
    XM_HIP_CHECK(hipMemsetAsync(ab.p, 0, ab.count * sizeof(double), st_));
    std::vector<double> hab((size_t)2 * (mmax + 1));
    const int batch = 8;   // Lanczos steps enqueued between two host checks
    for (int restart = 0; restart < std::max(1, env_restarts); ++restart) {
        to_dev(V.p, x.data(), (size_t)len * sizeof(double));
        al.clear(); be.clear();
        bool done = false;
        int m_use = 0;
        for (int j0 = 0; j0 < mmax && !done; j0 += batch) {
            const int j1 = std::min(mmax, j0 + batch);
            for (int j = j0; j < j1; ++j) {
                const double *vj = V.p + (size_t)j * len;
                // w = S v_j : the product input is v_j itself (pitch 1); output rows land in w at this rank's offset
                a.Wloc = vj + (size_t)cam0_ * 3;
                a.out = w.p + (size_t)cam0_ * 3;
                if (storage_ == XM_STORAGE_DENSE && sym_ok_ && Pcol_.p) launch_qw_sym(1, EPI_CERT, dQ_, ld_, vj, 1.0, a, Prow_.p, Pcol_.p, st_, j);
                else if (storage_ == XM_STORAGE_DENSE) launch_qw_dense(1, EPI_CERT, dQ_, ld_, vj, 1.0, a, st_);
                else if (storage_ == XM_STORAGE_SCHUR) schur_->product(1, EPI_CERT, vj, 1.0, a, st_);
                else if (sell_) launch_qw_sell(1, EPI_CERT, *sell_, vj, 1.0, a, 0, st_);
                else launch_qw_bsr3(1, EPI_CERT, rowptr_.p, colidx_.p, blocks_.p, vj, 1.0, a, st_, nb_loc_, rowinfo_.p);
                res_->qw_products++;
                if (comm_->active()) comm_->allgather(w.p, (size_t)nloc_ * 3, st_);
                // classical Gram-Schmidt twice against V(:,0..j); alpha_j = c1[j] + c2[j]; beta_j = |w|; v_{j+1} = w / beta_j
                if (lz_fused_ok(j + 1)) {   // seven launches per step: the sums of the partial dots are taken by the kernels that use them (same bits)
                    launch_dots_multi_parts(V.p, len, j + 1, w.p, len, dscr.p, st_);
                    launch_sub_vc_fin(w.p, V.p, len, dscr.p, j + 1, len, c.p, nullptr, nullptr, st_);
                    launch_dots_multi_parts(V.p, len, j + 1, w.p, len, dscr.p, st_);
                    launch_sub_vc_fin(w.p, V.p, len, dscr.p, j + 1, len, c2.p, c.p, ab.p + j, st_);
                    launch_dots_multi_parts(w.p, len, 1, w.p, len, dscr.p, st_);
                    launch_lz_next_fin(V.p + (size_t)(j + 1) * len, w.p, dscr.p, ab.p + (mmax + 1) + j, len, st_);
                } else {
                    launch_dots_multi(V.p, len, j + 1, w.p, len, c.p, dscr.p, st_);
                    launch_sub_vc(w.p, V.p, len, c.p, j + 1, len, st_);
                    launch_dots_multi(V.p, len, j + 1, w.p, len, c2.p, dscr.p, st_);
                    launch_sub_vc(w.p, V.p, len, c2.p, j + 1, len, st_);
                    launch_lz_alpha(c.p + j, c2.p + j, ab.p + j, st_);
                    launch_dots_multi(w.p, len, 1, w.p, len, c.p + mmax + 1, dscr.p, st_);
                    launch_lz_next(V.p + (size_t)(j + 1) * len, w.p, c.p + mmax + 1, ab.p + (mmax + 1) + j, len, st_);
                }
                total++;
            }
            to_host(hab.data(), ab.p, hab.size() * sizeof(double));
            for (int j = j0; j < j1; ++j) {
                al.push_back(hab[(size_t)j]);
                const double beta = hab[(size_t)(mmax + 1) + j];
                const int m = j + 1;
                tridiag_min(al, be, m, theta, y, tmax);
                resid = std::fabs(beta * y[(size_t)m - 1]);
                m_use = m;
                const bool exhausted = (m == (int)m3) || (beta < 1e-13 * std::max(1.0, tmax));   // full space, or an invariant subspace under full re-orthogonalisation
                if (exact && exhausted && std::isfinite(beta)) eig_exact_ = true;
                if ((!exact && resid <= 1e-9 * std::max(1.0, tmax)) || exhausted || !std::isfinite(beta)) { done = true; break; }
                be.push_back(beta);
            }
        }
        // Ritz vector x = V(:,0..m_use-1) y
        if ((int)y.size() != m_use) tridiag_min(al, be, m_use, theta, y, tmax);
        to_dev(c.p, y.data(), (size_t)m_use * sizeof(double));
        launch_gemv_n(w.p, V.p, len, c.p, m_use, len, st_);
        to_host(x.data(), w.p, (size_t)len * sizeof(double));
        double nn = 0;
        for (int64_t i = 0; i < len; ++i) nn += x[(size_t)i] * x[(size_t)i];
        nn = std::sqrt(nn);
        if (nn > 0) for (int64_t i = 0; i < len; ++i) x[(size_t)i] /= nn;
        if (done) break;
    }
    x_out.assign((size_t)m3, 0.0);
    for (int64_t i = 0; i < m3; ++i) x_out[(size_t)i] = x[(size_t)(3 * posv[(size_t)(i / 3)] + i % 3)];
    theta_out = theta;
    iters_out = total;
    resid_out = resid;
    return (resid <= 1e-6 * std::max(1.0, tmax)) ? 0 : 1;
}

// ------------------------------------------------------------------------------------------------------------------
// dual certificate (checkeig.h:42-368)
// ------------------------------------------------------------------------------------------------------------------
CertResult Context::certificate(int o, double primal, std::vector<double> &v_out) {
    CertResult cr;
    const auto t0 = clk::now();
    int64_t pst0[3] = {0, 0, 0};
    const int64_t pcg_unconverged_at_start = schur_info(pst0, nullptr) ? pst0[2] : 0;
    double *Wloc = W_.p + (size_t)cam0_ * 3 * OP_;
    // Right-hand side pieces: C * sR  (checkeig.h:182 with the diagonal term folded into cert_prepare)
    launch_scale_rows(o, nloc_, R_.p, s_.p, Wloc, st_);
    gather_W();
    CamArgs a = cam_args(cur_);
    a.out = HpR_.p;
    product(EPI_PLAIN, o, 1.0, a);
    const int g = (nloc_ + 255) / 256;
    double *Lam = ps_[cur_ ^ 1].S0.p, *dz = ps_[cur_ ^ 1].egs.p;
    launch_cert_prepare(o, nloc_, cam0_, opt_->lam, HpR_.p, R_.p, s_.p, Lam, dz, partsM_.p + (size_t)comm_->rank * 2 * g, st_);
    if (comm_->active()) comm_->allgather(partsM_.p, (size_t)2 * g, st_);
    const double dual = sum_parts(partsM_.p, 2 * g * comm_->world);   // y0+y3+y5 + lam*sum(1 - xii^2)  (checkeig.h:322-332)
    // lambda_min(S) and its eigenvector
    XM_HIP_CHECK(hipMemsetAsync(W_.p, 0, W_.count * sizeof(double), st_));
    double theta = 0, eig_resid = 0;
    int its = 0;
    // A Ritz value is an UPPER bound of lambda_min: a run that has not converged is optimistic and must never certify a point.
    const int not_converged = lanczos_min(v_out, theta, its, eig_resid);
    res_->lanczos_iters += its;
    res_->eig_residual = eig_resid;
    if (not_converged) {
        res_->cert_flags |= XM_CERT_EIG_NOT_CONVERGED;
        log("warning: Lanczos did not converge (Ritz residual %1.3e): the certificate is not accepted on this value\n", eig_resid);
    } else {
        res_->cert_flags &= ~XM_CERT_EIG_NOT_CONVERGED;
    }
    if (eig_exact_) res_->cert_flags |= XM_CERT_EIG_EXACT; else res_->cert_flags &= ~XM_CERT_EIG_EXACT;
    log("The min eig is: %1.3e \n", theta);
    log("Primal value: %g\nnew dual%g\n", primal, dual);
    const double K = 3.0 * (double)n_;
    const double gap = primal - dual - K * std::min(0.0, theta);   // checkeig.h:334-336
    log("Optimility gap: %g\n", gap);
    double bound = 1e-4;                                           // checkeig.h:349-358 (later branches unreachable)
    if (n_ > 2000) bound = 1e-3;
    // matrix-free storage: a product whose inner CG gave up is not the operator the certificate is about
    int64_t pst[3] = {0, 0, 0};
    const bool inexact = schur_info(pst, nullptr) && pst[2] > pcg_unconverged_at_start;
    if (inexact) {
        res_->cert_flags |= XM_CERT_INEXACT_OPERATOR;
        log("warning: %lld matrix-free products of this certificate did not reach the inner tolerance: not accepted\n", (long long)(pst[2] - pcg_unconverged_at_start));
    } else {
        res_->cert_flags &= ~XM_CERT_INEXACT_OPERATOR;
    }
    cr.accepted = !not_converged && !inexact && (gap / primal < 1e-3 || theta > -bound);
    cr.min_eig = theta; cr.dual = dual; cr.gap = gap; cr.lanczos_iters = its;
    if (cr.accepted) log("BM finished with rank %d\n", o); else log("BM order plus one\n");
    res_->cert_seconds += secs_since(t0);
    return cr;
}

// ------------------------------------------------------------------------------------------------------------------
// out = alpha * Q * W through whatever storage the context holds (tests, diagnostics): host matrices, column-major 3n x o
// ------------------------------------------------------------------------------------------------------------------
void Context::apply(int o, const double *Wh, double *out, double alpha) {
    if (comm_->active()) throw Error(XM_ERR_ARG, "xm_ctx_qw: single-rank contexts only");
    if (!Wh || !out) throw Error(XM_ERR_ARG, "xm_ctx_qw: null argument");
    xm_options_t opt;
    std::memset(&opt, 0, sizeof(opt));
    opt_ = &opt;
    setup_rank(o);
    const size_t m = (size_t)3 * n_;
    std::vector<double> rm((size_t)ld_ * OP_ + 2, 0.0);
    for (size_t r = 0; r < m; ++r)
        for (int k = 0; k < o; ++k) rm[r * OP_ + k] = Wh[r + (size_t)k * m];
    XM_HIP_CHECK(hipMemcpyAsync(W_.p, rm.data(), rm.size() * sizeof(double), hipMemcpyHostToDevice, st_));
    CamArgs a = cam_args(cur_);
    a.out = HpR_.p;
    product(EPI_PLAIN, o, alpha, a);
    std::vector<double> res((size_t)nloc_ * 3 * OP_);
    XM_HIP_CHECK(hipMemcpyAsync(res.data(), HpR_.p, res.size() * sizeof(double), hipMemcpyDeviceToHost, st_));
    XM_HIP_CHECK(hipMemsetAsync(W_.p, 0, W_.count * sizeof(double), st_));
    XM_HIP_CHECK(hipStreamSynchronize(st_));
    for (size_t r = 0; r < m; ++r)
        for (int k = 0; k < o; ++k) out[r + (size_t)k * m] = res[r * OP_ + k];
    opt_ = nullptr;
    solved_ = false;
}

// ------------------------------------------------------------------------------------------------------------------
// XM^2 re-weighting on the resident context (reference loop 3_test_colmap_glomap.py:299-351)
// ------------------------------------------------------------------------------------------------------------------
void Context::attach_edges(int64_t ne, const int32_t *ei, const int32_t *ej, const double *M) {
    if (ne < 0 || (ne > 0 && (!ei || !ej || !M))) throw Error(XM_ERR_ARG, "attach_edges: bad argument");
    validate_edges(n_, ne, ei, ej, "attach_edges");
    // the kernels address cameras by their position in the padded numbering (== the global index unless the row partition is
    // balanced by stored blocks)
    std::vector<int32_t> pi((size_t)std::max<int64_t>(ne, 1)), pj((size_t)std::max<int64_t>(ne, 1));
    for (int64_t e = 0; e < ne; ++e) { pi[(size_t)e] = (int32_t)pos_of(ei[e]); pj[(size_t)e] = (int32_t)pos_of(ej[e]); }
    ei = pi.data(); ej = pj.data();
    std::vector<int64_t> inc((size_t)ntot_ + 1, 0);
    for (int64_t e = 0; e < ne; ++e) { inc[(size_t)ei[e] + 1]++; inc[(size_t)ej[e] + 1]++; }
    for (int64_t c = 0; c < ntot_; ++c) inc[(size_t)c + 1] += inc[(size_t)c];
    std::vector<int32_t> ie((size_t)std::max<int64_t>(2 * ne, 1));
    {   // incidence lists in edge order (fixed -> the diagonal sums are bit-reproducible)
        std::vector<int64_t> next(inc.begin(), inc.end() - 1);
        for (int64_t e = 0; e < ne; ++e) { ie[(size_t)next[(size_t)ei[e]]++] = (int32_t)e; ie[(size_t)next[(size_t)ej[e]]++] = (int32_t)e; }
    }
    const size_t m = (size_t)std::max<int64_t>(ne, 1);
    ei_.alloc(m, false); ej_.alloc(m, false); eM_.alloc(m * 9, false); ew_.alloc(m); eres_.alloc(m);
    inc_ptr_.alloc(inc.size(), false); inc_edge_.alloc(ie.size(), false);
    pos_ij_.alloc(m); pos_ji_.alloc(m); pos_d_.alloc((size_t)nloc_);
    if (ne > 0) {
        XM_HIP_CHECK(hipMemcpy(ei_.p, ei, (size_t)ne * sizeof(int32_t), hipMemcpyHostToDevice));
        XM_HIP_CHECK(hipMemcpy(ej_.p, ej, (size_t)ne * sizeof(int32_t), hipMemcpyHostToDevice));
        XM_HIP_CHECK(hipMemcpy(eM_.p, M, (size_t)ne * 9 * sizeof(double), hipMemcpyHostToDevice));
    }
    XM_HIP_CHECK(hipMemcpy(inc_ptr_.p, inc.data(), inc.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    XM_HIP_CHECK(hipMemcpy(inc_edge_.p, ie.data(), ie.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    ne_ = ne;
    if (storage_ == XM_STORAGE_BSR3) {
        launch_edge_locate(ne, ei_.p, ej_.p, cam0_, nloc_, rowptr_.p, colidx_.p, pos_ij_.p, pos_ji_.p, pos_d_.p, st_);
        std::vector<int64_t> a(m), b(m), d((size_t)nloc_);
        XM_HIP_CHECK(hipMemcpyAsync(a.data(), pos_ij_.p, m * sizeof(int64_t), hipMemcpyDeviceToHost, st_));
        XM_HIP_CHECK(hipMemcpyAsync(b.data(), pos_ji_.p, m * sizeof(int64_t), hipMemcpyDeviceToHost, st_));
        XM_HIP_CHECK(hipMemcpyAsync(d.data(), pos_d_.p, (size_t)nloc_ * sizeof(int64_t), hipMemcpyDeviceToHost, st_));
        XM_HIP_CHECK(hipStreamSynchronize(st_));
        for (int64_t e = 0; e < ne; ++e)
            if (a[(size_t)e] == -2 || b[(size_t)e] == -2) { ne_ = 0; throw Error(XM_ERR_ARG, "attach_edges: an edge has no stored block in the BSR3 pattern"); }
        for (int64_t c = 0; c < true_loc_; ++c)
            if (d[(size_t)c] < 0 && inc[(size_t)(cam0_ + c) + 1] > inc[(size_t)(cam0_ + c)]) { ne_ = 0; throw Error(XM_ERR_ARG, "attach_edges: a camera with edges has no stored diagonal block"); }
    }
}

void Context::edge_residuals(double *res) {
    if (storage_ == XM_STORAGE_SCHUR) {   // matrix-free: the observations are the edges (one residual per observation, input order)
        if (!solved_) throw Error(XM_ERR_ARG, "edge_residuals: the context holds no solution yet");
        if (!res) throw Error(XM_ERR_ARG, "edge_residuals: null output");
        xm_options_t opt;
        std::memset(&opt, 0, sizeof(opt));
        const xm_options_t *keep = opt_;
        opt_ = &opt;
        launch_scale_rows(o_, nloc_, R_.p, s_.p, W_.p + (size_t)cam0_ * 3 * OP_, st_);
        gather_W();
        flush_gather();
        CamArgs a = cam_args(cur_);
        a.out = HpR_.p;
        schur_->residuals(o_, W_.p, res, a, st_);
        XM_HIP_CHECK(hipMemsetAsync(W_.p, 0, W_.count * sizeof(double), st_));
        XM_HIP_CHECK(hipStreamSynchronize(st_));
        opt_ = keep;
        return;
    }
    if (ne_ <= 0 && !ei_.p) throw Error(XM_ERR_ARG, "edge_residuals: no edges attached");
    if (!solved_) throw Error(XM_ERR_ARG, "edge_residuals: the context holds no solution yet");
    if (!res) throw Error(XM_ERR_ARG, "edge_residuals: null output");
    launch_scale_rows(o_, nloc_, R_.p, s_.p, W_.p + (size_t)cam0_ * 3 * OP_, st_);
    gather_W();
    flush_gather();
    launch_edge_residual(ne_, ei_.p, ej_.p, eM_.p, W_.p, o_, OP_, eres_.p, st_);
    if (ne_ > 0) to_host(res, eres_.p, (size_t)ne_ * sizeof(double));
    XM_HIP_CHECK(hipMemsetAsync(W_.p, 0, W_.count * sizeof(double), st_));
    XM_HIP_CHECK(hipStreamSynchronize(st_));
}

// residuals of a RECOVERED rank-3 solution on the device: the product input is W_i = s_i * R_i^T (camera i's 3 x 3 record)
const double *Context::residuals_recovered_device(const double *rot, const double *scale) {
    if (!rot || !scale) throw Error(XM_ERR_ARG, "edge_residuals_recovered: null argument");
    if (storage_ != XM_STORAGE_SCHUR && !ei_.p) throw Error(XM_ERR_ARG, "edge_residuals_recovered: no edges attached");
    // Under a row partition every rank holds the whole edge list and gets the whole recovered solution from the caller: each
    // evaluates all residuals (and, in xm2_filter, the same order statistic) itself -- identical numbers everywhere, no exchange.
    constexpr int OP = pitch_of(3);
    if (o_ < 3 || W_.count < (size_t)ld_ * OP + 2) setup_rank(3);
    std::vector<double> hW((size_t)ld_ * OP + 2, 0.0);
    for (int64_t i = 0; i < n_; ++i) {
        const size_t pi = (size_t)pos_of(i);   // the attached edges address cameras in the padded numbering
        for (int a = 0; a < 3; ++a)
            for (int k = 0; k < 3; ++k) hW[(pi * 3 + a) * OP + k] = scale[i] * rot[(size_t)k + 3 * ((size_t)3 * i + a)];
    }
    DevBuf<double> &dW = recW_;
    if (dW.count < hW.size()) dW.alloc(hW.size());
    to_dev(dW.p, hW.data(), hW.size() * sizeof(double));
    if (storage_ == XM_STORAGE_SCHUR) {
        xm_options_t opt;
        std::memset(&opt, 0, sizeof(opt));
        const xm_options_t *keep = opt_;
        opt_ = &opt;
        const int o_keep = o_, OP_keep = OP_;
        o_ = 3; OP_ = OP;
        CamArgs a = cam_args(cur_);
        a.out = HpR_.p;
        const double *r = schur_->residuals_device(3, dW.p, a, st_);
        o_ = o_keep; OP_ = OP_keep;
        opt_ = keep;
        return r;
    }
    launch_edge_residual(ne_, ei_.p, ej_.p, eM_.p, dW.p, 3, OP, eres_.p, st_);
    return eres_.p;
}
void Context::edge_residuals_recovered(const double *rot, const double *scale, double *res) {
    if (!res) throw Error(XM_ERR_ARG, "edge_residuals_recovered: null output");
    const double *r = residuals_recovered_device(rot, scale);
    const int64_t ne = (storage_ == XM_STORAGE_SCHUR) ? schur_->nobs() : ne_;
    if (ne > 0) to_host(res, r, (size_t)ne * sizeof(double));
    XM_HIP_CHECK(hipStreamSynchronize(st_));
    if (comm_->active()) comm_->host_barrier();   // nobody runs ahead into a collective while a peer still (re)allocates here
}

// k-th smallest (0-based) of n non-negative doubles on the device: most-significant-digit radix select, 4 passes of 16 bits
static double radix_select(const double *x, int64_t n, int64_t k, unsigned int *dhist, hipStream_t st) {
    unsigned long long prefix = 0;
    std::vector<unsigned int> h(65536);
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 48 - 16 * pass;
        XM_HIP_CHECK(hipMemsetAsync(dhist, 0, 65536 * sizeof(unsigned int), st));
        launch_radix_hist(n, x, shift, prefix, dhist, st);
        XM_HIP_CHECK(hipMemcpyAsync(h.data(), dhist, 65536 * sizeof(unsigned int), hipMemcpyDeviceToHost, st));
        XM_HIP_CHECK(hipStreamSynchronize(st));
        int64_t acc = 0;
        int bin = 65535;
        for (int b = 0; b < 65536; ++b) {
            if (acc + (int64_t)h[(size_t)b] > k) { bin = b; break; }
            acc += h[(size_t)b];
        }
        k -= acc;
        prefix = (prefix << 16) | (unsigned long long)bin;
    }
    double v;
    std::memcpy(&v, &prefix, sizeof(double));
    return v;
}

double Context::xm2_filter(const double *rot, const double *scale, double pct, int64_t *removed, double *w_out) {
    if (!(pct >= 0.0 && pct <= 100.0)) throw Error(XM_ERR_ARG, "xm2_filter: percentile must be in [0, 100]");
    const int64_t ne = (storage_ == XM_STORAGE_SCHUR) ? schur_->nobs() : ne_;
    if ((int64_t)w_cur_.size() != ne || ne < 1) throw Error(XM_ERR_ARG, "xm2_filter: the context does not know its edge weights (view-graph / matrix-free storage, or call xm_ctx_set_edge_weights first)");
    const double *res = residuals_recovered_device(rot, scale);
    if (ew_.count < (size_t)ne) ew_.alloc((size_t)ne);
    if (eerr_.count < (size_t)ne) eerr_.alloc((size_t)ne);
    to_dev(ew_.p, w_cur_.data(), (size_t)ne * sizeof(double));
    launch_xm2_error(ne, ew_.p, res, eerr_.p, st_);
    DevBuf<unsigned int> hist;
    hist.alloc(65536 + 1);
    // numpy.percentile, method "linear": h = (n - 1) q; x[floor h] + (h - floor h) (x[floor h + 1] - x[floor h]) -- over the edges that are
    // still THERE: the reference deletes the removed ones before the next round's percentile (3_test_colmap_glomap.py:321-328).  Here a
    // removed edge keeps its place with weight 0, hence error 0: the nz zeros sort in front of every live error (>= 0), so the k-th order
    // statistic of the survivors is entry k + nz of the whole array.
    int64_t nz = 0;
    for (int64_t e = 0; e < ne; ++e) nz += (w_cur_[(size_t)e] == 0.0);
    const int64_t nlive = ne - nz;
    if (nlive < 1) throw Error(XM_ERR_ARG, "xm2_filter: every edge has been removed already");
    const double hq = (double)(nlive - 1) * pct / 100.0;
    const int64_t k0 = (int64_t)std::floor(hq), k1 = std::min<int64_t>(k0 + 1, nlive - 1);
    const double x0 = radix_select(eerr_.p, ne, k0 + nz, hist.p, st_);
    const double x1 = (k1 == k0) ? x0 : radix_select(eerr_.p, ne, k1 + nz, hist.p, st_);
    const double thr = x0 + (hq - (double)k0) * (x1 - x0);
    XM_HIP_CHECK(hipMemsetAsync(hist.p + 65536, 0, sizeof(unsigned int), st_));
    DevBuf<double> wn;
    wn.alloc((size_t)ne, false);
    launch_xm2_filter(ne, eerr_.p, ew_.p, thr, wn.p, hist.p + 65536, st_);
    unsigned int rm = 0;
    std::vector<double> w2((size_t)ne);
    to_host(w2.data(), wn.p, (size_t)ne * sizeof(double));
    to_host(&rm, hist.p + 65536, sizeof(unsigned int));
    if (removed) *removed = (int64_t)rm;
    set_edge_weights(w2.data());
    if (w_out) std::memcpy(w_out, w2.data(), (size_t)ne * sizeof(double));
    hist.release(); wn.release();
    if (comm_->active()) comm_->host_barrier();   // as above
    return thr;
}

void Context::recover_tp(const double *rot, const double *scale, double *t, double *p) {
    if (storage_ != XM_STORAGE_SCHUR || !schur_) throw Error(XM_ERR_ARG, "recover_tp: needs a matrix-free context (XM_STORAGE_SCHUR): the translations and landmarks are functions of the observations");
    schur_->recover_tp(rot, scale, t, p, st_);
}
int64_t Context::n_landmarks() const { return schur_ ? schur_->n_landmarks() : 0; }
bool Context::schur_info(int64_t out[3], double *relres) const {
    if (!schur_ || !schur_->uses_pcg()) return false;
    schur_->pcg_stats(out, relres);
    return true;
}

void Context::set_edge_weights(const double *w) {
    if (storage_ == XM_STORAGE_SCHUR) {
        schur_->set_weights(w, st_);
        w_cur_.assign(w, w + schur_->nobs());
        return;
    }
    if (!ei_.p) throw Error(XM_ERR_ARG, "set_edge_weights: no edges attached");
    if (ne_ > 0 && !w) throw Error(XM_ERR_ARG, "set_edge_weights: null weights");
    w_cur_.assign(w, w + ne_);
    if (ne_ > 0) to_dev(ew_.p, w, (size_t)ne_ * sizeof(double));
    const bool dense = (storage_ == XM_STORAGE_DENSE);
    launch_edge_write(dense, ne_, ei_.p, ej_.p, eM_.p, ew_.p, cam0_, nloc_, inc_ptr_.p, inc_edge_.p, pos_ij_.p, pos_ji_.p, pos_d_.p,
                      dense ? nullptr : blocks_.p, dense ? dQ_ : nullptr, ld_, st_);
    if (sell_) sell_->refill(colidx_.p, blocks_.p, st_);   // the sliced-ELL copy follows the CSR values
    XM_HIP_CHECK(hipStreamSynchronize(st_));
}

// ------------------------------------------------------------------------------------------------------------------
// staircase drivers (XM_main.cu:180-310 solve, :312-401 solve_rank3, :35-178 solve_rebuttle)
// ------------------------------------------------------------------------------------------------------------------
void Context::solve(const xm_options_t &opt, xm_result_t &res) {
    opt_ = &opt;
    res_ = &res;
    verbose_ = (opt.flags & XM_FLAG_VERBOSE) != 0;
    if (opt.retraction != XM_RETRACT_QR && opt.retraction != XM_RETRACT_POLAR) throw Error(XM_ERR_ARG, "unknown retraction");
    if (opt.sum_grouping < 0 || opt.sum_grouping > 2) throw Error(XM_ERR_ARG, "sum_grouping must be 0, 1 or 2");
    retraction_ = opt.retraction;
    grouping_ = opt.sum_grouping;
    double *Rout = res.R, *sout = res.s;
    std::memset(&res, 0, sizeof(res));
    res.R = Rout; res.s = sout;
    if (!Rout || !sout) throw Error(XM_ERR_ARG, "result buffers missing");
    if (opt.max_rank > (unsigned)kMaxRank) throw Error(XM_ERR_ARG, "max_rank > 10 is not instantiated");
    if ((opt.flags & XM_FLAG_PROFILE_QW) && ev_pool_.empty()) {
        ev_pool_.resize(256);
        for (auto &e : ev_pool_) { XM_HIP_CHECK(hipEventCreate(&e.first)); XM_HIP_CHECK(hipEventCreate(&e.second)); }
    }
    ev_used_ = 0;
    hess_launches_ = 0;
    qw_samples_.clear();
    const auto t0 = clk::now();
    const size_t m = (size_t)3 * n_;
    log("+++++++++++++++++++++++++++++++++\nBegin XM\n+++++++++++++++++++++++++++++++++\n");
    unsigned o = 3;
    std::vector<double> v(m, 0.0), R0(m * 3, 0.0), Rk, s0((size_t)n_, 1.0), sk;
    if (opt.mode == XM_MODE_REBUTTLE && opt.s_ini) for (int64_t i = 0; i < n_; ++i) s0[(size_t)i] = opt.s_ini[i];
    const double s_anchor = s0[0];
    double gradtol = opt.tol, primal = 0;
    int status = XM_STATUS_NONE;
    auto identity_stack = [&]() {
        R0.assign(m * 3, 0.0);
        for (size_t i = 0; i < (size_t)n_; ++i) { R0[3 * i] = 1.0; R0[3 * i + m + 1] = 1.0; R0[3 * i + 2 * m + 2] = 1.0; }
    };
    int out_rank = 3;
    if (opt.mode == XM_MODE_RANK3) {
        log("+++++++++++++++++++++++++++++++++\nSolve TR with Rank   3\n+++++++++++++++++++++++++++++++++\n");
        identity_stack();
        setup_rank(3);
        upload_point(R0, 3, s0);
        TrResult tr = trust_region(3, gradtol, 0.0, v, opt.max_time);
        primal = tr.primal;
        download_point(R0, s0);
        out_rank = 3;
    } else {
        while (o <= opt.max_rank) {  // XM_main.cu:223
            log("+++++++++++++++++++++++++++++++++\nSolve TR with Rank   %u\n+++++++++++++++++++++++++++++++++\n", o);
            TrResult tr;
            setup_rank((int)o);
            if (o == 3) {
                identity_stack();
                // XM_FLAG_WARM_R is honoured in EVERY staircase mode (solve and solve_rebuttle differ only in s_ini): a caller who hands
                // over R_ini wants the warm start, and silently starting from the identity would only cost iterations
                if ((opt.flags & XM_FLAG_WARM_R) && !opt.R_ini) throw Error(XM_ERR_ARG, "XM_FLAG_WARM_R without R_ini");
                const bool warm = (opt.flags & XM_FLAG_WARM_R) && opt.R_ini;
                if (warm) std::memcpy(R0.data(), opt.R_ini, m * 3 * sizeof(double));
                upload_point(R0, 3, s0);
                if (warm) {   // rows of a previous solution truncated to rank 3 are re-orthonormalised (MGS of R + 0 * D)
                    XM_HIP_CHECK(hipMemsetAsync(D_.p, 0, D_.count * sizeof(double), st_));
                    launch_retract(3, nloc_, cam0_, R_.p, s_.p, D_.p, nullptr, 0.0, Rc_.p, nullptr, nullptr, st_, 0);
                    std::swap(R_.p, Rc_.p);
                }
                tr = trust_region(3, gradtol, 0.0, v, opt.max_time);
            } else {
                upload_point(R0, (int)o, s0);
                tr = trust_region((int)o, gradtol, 1.0, v, opt.max_time);
            }
            primal = tr.primal;
            if (primal < 0) { status = XM_STATUS_LS_FAILED; o += 1; break; }  // XM_main.cu:244-247 (R0 keeps [R | 0])
            log("+++++++++++++++++++++++++++++++++\nCheck Eigen value\n+++++++++++++++++++++++++++++++++\n");
            CertResult ce = certificate((int)o, primal, v);
            res.dual = ce.dual; res.min_eig = ce.min_eig; res.gap = ce.gap;
            download_point(Rk, sk);
            if (ce.accepted) {
                R0 = Rk; s0 = sk; o += 1; status = XM_STATUS_CERTIFIED;
                break;
            } else if (o < opt.max_rank) {  // XM_main.cu:265-271
                R0.assign(m * (o + 1), 0.0);
                std::copy(Rk.begin(), Rk.end(), R0.begin());
                s0 = sk;
                for (size_t i = 0; i < (size_t)n_; ++i) { v[3 * i] /= sk[i]; v[3 * i + 1] /= sk[i]; v[3 * i + 2] /= sk[i]; }  // XM_main.cu:8-16
            } else {
                R0 = Rk; s0 = sk; status = XM_STATUS_MAX_RANK;
            }
            o += 1;
        }
        if (o > opt.max_rank) log("BM stoped because max rank\n");
        out_rank = (int)o - 1;
    }
    res.rank = out_rank;
    if (out_rank >= 3 && R0.size() >= m * (size_t)out_rank) std::memcpy(Rout, R0.data(), m * (size_t)out_rank * sizeof(double));
    for (int64_t i = 0; i < n_; ++i) sout[i] = s0[(size_t)i];
    sout[0] = s_anchor;  // the callers' s_ex[0] is never touched by the reference (XM_main.cu:203-219)
    res.status = status;
    res.primal = primal;
    if (opt.flags & XM_FLAG_PROFILE_QW) finish_profile();
    // algorithmic bytes of one tCG product at the final rank
    const int of = std::max(3, std::min(out_rank, (int)opt.max_rank));
    res.sym_product = ((sym_ok_ || symw_) && storage_ == XM_STORAGE_DENSE) ? 1 : 0;
    if (storage_ == XM_STORAGE_DENSE) res.qw_bytes = 8LL * (3 * n_) * (3 * n_) + 2LL * 8 * 3 * n_ * of;
    else if (storage_ == XM_STORAGE_SCHUR) res.qw_bytes = schur_->bytes_per_product(of);
    else res.qw_bytes = 76LL * nb_loc_ + 4LL * (n_ + 1) + 2LL * 8 * 3 * n_ * of;
    res.qw_stream_bytes = (storage_ == XM_STORAGE_DENSE) ? (symw_ ? symw_->stream_bytes() : (res.sym_product ? 4LL : 8LL) * (3 * n_) * (3 * n_))
                          : (storage_ == XM_STORAGE_BSR3) ? ((sell_ && sell_supports(of)) ? sell_->stream_bytes() : 76LL * nb_loc_) : 0;
    res.n_gpus = comm_->world;
    res.exchange = !comm_->active() ? 0 : (xchg_.world > 1 ? 2 : 1);
    res.seconds = secs_since(t0);
    // leave the end point resident at its final rank for edge_residuals(): R_/s_ hold the last trust-region point already,
    // unless the staircase escalated past it (then the last stage's buffers still describe the returned point)
    solved_ = true;
    opt_ = nullptr;
    res_ = nullptr;
}

}  // namespace xm
