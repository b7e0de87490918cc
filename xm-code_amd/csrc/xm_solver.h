// xm_solver.h — host-side driver of the MI355X-native XM solve (C++; mirrors XM_main.cu / trustregion.h / checkeig.h).
#pragma once

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

#include "../../include/xm_amd.h"
#include "xm_common.h"

namespace xm {

// ---- settings resolved ONCE at context creation (xm_tuning_t of the caller, else the defaults) ----------------------------------------
struct Settings {
    int sym = 0;                 // 0 auto | 1 force | -1 off
    int64_t sym_min_rows = 0;    // 0 = the measured default: sym_rows()
    // Rows (3n) from which an exactly symmetric dense Q goes through the half-traffic kernels.  One GPU: the two-launch symmetric product ties
    // with the general kernel at 1280 cameras and wins from there on (round 6, both with the alternating direction, us, general / symmetric:
    // n = 1024 o = 3 14.1 / 13.9, o = 4 15.5 / 14.8; n = 1280 18.3 / 18.4, 20.8 / 19.8; n = 1536 24.1 / 20.7, 25.4 / 21.7; n = 1778 31.9 / 25.8,
    // 33.2 / 27.2; n = 2048 44.4 / 33.0; profiles/r06_kbench_dense_sym_crossover.txt) -> 4096 rows (round 5: 5120, round 2: 6144).
    // Several ranks (cyclic half window, xm_symw.h): measured at Final-13682 size only -> 6144.
    int64_t sym_rows(int world) const { return sym_min_rows > 0 ? sym_min_rows : (world <= 1 ? 4096 : 6144); }
    int sell = 0;                // 0 auto | 1 force | -1 off
    int sell_slabs = 4, sell_lmax = 64, sell_gather = 1;   // sell_gather: the kernels' gather mode (0 a record of W per lane | 1 records fetched element-per-lane and transposed through LDS, the default)
    int sell_codec = 0;          // 0 auto | 1 full | 2 quaternion
    int sell_wpad = 0;           // the tCG keeps a copy of W at a 128-byte record pitch for the sliced-ELL gather (single rank, o = 3..5): 0 when the
                                 // column pattern says it pays (SellMatrix::padded_pays) | 1 always | -1 never
    int overlap = 0;             // 0 auto | -1 off
    double overlap_min_mb = 64.0;
    int64_t cert_dense_rows = 384;
    int lanczos_mmax = 400, lanczos_restarts = 12;
    double watchdog_s = 600.0;
    int balance = 0;             // 0 by stored blocks | 1 equal camera ranges
    int exchange = 0;            // xm_tuning_t.exchange: 0 auto | 1 an all-gather between the launches | 2 direct peer writes | 3 RCCL
    int split_k = 0;             // 0 auto (multi-rank small strips) | -1 off | 2..8 forced
    long long debug_drop_finalize = -1;   // tests: this outer iteration loses its result kernel
    int debug_peer_mute = 0;              // tests: rank 1 never publishes its tCG epoch -> the peers' bounded wait must expire
    int exchange_lite = 1;                // 0 (xm_tuning_t.exchange_fence): the fused tCG exchange pushes with plain stores + a system-scope release fence instead of write-through stores
    bool schur_host_assembly = false, schur_trace = false;   // matrix-free storage (xm_schur.h: SchurSettings)
    int schur_pcg_first = 0, schur_pcg_hess_digits = 0;   // CG form of the matrix-free storage (xm_schur.h: SchurSettings)
    int schur_solver = 0;                 // 0 by size | 1 dense inverse of the reduced camera Laplacian | 2 preconditioned CG inside the product
    int64_t schur_dense_max = 20000;
    static Settings resolve(const xm_tuning_t *t);
};

// ---- communicators (row partition over the GPUs of one node) ---------------------------------------------------------------------
// A Context holds a shared_ptr to ITS communicator (no process-global state inside the solver).  Kinds:
//   none   single GPU
//   RCCL   one process per GPU, ncclAllGather (xm_comm_init)
//   shm    test transport: the same calls staged through a POSIX shared-memory segment (several ranks on one GPU)
//   peer   DIRECT PEER WRITES: every rank owns a fine-grained arena on its device that all ranks of the group can address (same
//          process: peer access; "virtual devices": the same device; one process per GPU: hipIpcMemHandle mappings); a collective is a copy kernel that stores the rank's chunk
//          straight into the peers' memory, a system-scope release, an epoch flag per (source, slot), and a bounded spin on the
//          consumer side.  No library call, no host rendezvous, ~1 hop of xGMI latency.  The truncated CG fuses push and wait into
//          cg_step_kernel (PeerXchg below).
struct Comm {
    int rank = 0, world = 1;
    bool forced = false;   // issue the collectives even with one rank (exercises the path on a 1-GPU box)
    virtual ~Comm();
    bool active() const { return world > 1 || forced; }
    virtual int kind() const { return 0; }                    // 0 none | 1 RCCL | 2 shm | 3 peer (threads) | 4 peer (processes, IPC)
    // in-place all-gather on `stream`: every rank contributes `count` doubles located at buf + rank*count
    virtual void allgather(double *buf, size_t count, hipStream_t st) { (void)buf; (void)count; (void)st; }
    // ---- direct exchange (kind 3 only) ----
    virtual bool peer() const { return false; }
    // collective: (re)allocate this rank's tCG exchange buffer of `doubles` doubles in peer-addressable memory and learn the peers'
    virtual void xchg_setup(size_t doubles, PeerXchg &out) { (void)doubles; (void)out; }
    virtual void reserve(size_t doubles) { (void)doubles; }   // collective: staging capacity of allgather (per rank chunk * world)
    virtual void check_device_error() {}                      // throws XM_ERR_COMM when a device-side wait expired
    virtual void host_barrier() {}
    // ranks of this communicator whose kernels run on THIS rank's device (1 on a real node; > 1 for virtual devices / processes sharing a GPU):
    // a launch that waits for its peers inside the kernel must leave room for theirs (Context::tcg_blocks)
    virtual int ranks_on_my_device() const { return 1; }
    // may a kernel of this rank wait for its peers ON THE DEVICE?  Not when more than four ranks share one GPU (a test vehicle): a waiting
    // kernel holds its hardware queue, and the queues of eight ranks' pushes and waits are more than the device keeps active at once --
    // the collectives are then synchronised through the host and the tCG exchange is not fused into cg_step
    virtual bool device_waits() const { return true; }
    virtual void set_exchange_fence(bool on) { (void)on; }   // peer transports: release fence instead of write-through stores in the collectives
    std::string fallback_note;   // why a faster transport was given up for this one (empty: first choice)
    void note(const char *what, double a, double b);   // XM_COMM_TRACE debugging aid
private:
    FILE *trace_ = nullptr;
    bool trace_tried_ = false;
};
std::shared_ptr<Comm> default_comm();          // what xm_comm_init / xm_comm_init_shm installed for this process (else a single-rank Comm)
void comm_unique_id(unsigned char id[128]);
void comm_init(int rank, int world, int device, const unsigned char id[128], const char *lib_path);
void comm_init_shm(int rank, int world, int device, const char *name, size_t bytes);
void comm_init_ipc(int rank, int world, int device, const char *name, double spin_seconds);   // one process per GPU, peer writes through IPC-mapped buffers
void comm_finalize();
// single-process group of `world` ranks (one host thread each); devices[r] = HIP device of rank r (all equal = virtual devices)
struct PeerGroup;
std::shared_ptr<PeerGroup> peer_group_create(int world, const int *devices, double spin_seconds);
std::shared_ptr<Comm> peer_comm_create(const std::shared_ptr<PeerGroup> &g, int rank);   // call on rank's own thread, device current
bool peer_comm_selftest(Comm &c);   // collective over the ranks of a peer communicator: all-gathers with known contents through both read paths
// library communicator for a rank driven by a host thread of THIS process (the fallback of the single-process multi-GPU mode): every
// rank calls it concurrently with the same id (comm_unique_id) and its own device current
std::shared_ptr<Comm> rccl_comm_create(int rank, int world, const unsigned char id[128]);
void peer_group_abort(const std::shared_ptr<PeerGroup> &g);
double peer_allgather_bench(int world, int gpu_map, int64_t count, int reps);   // microseconds per collective (xm_team.hip)                               // wakes every host-side wait with an error

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t count = 0;
    size_t capacity = 0;   // elements allocated (>= count): ensure() reuses the allocation
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        count = 0;
        capacity = 0;
    }
    void alloc(size_t n, bool zero = true) {
        release();
        count = n;
        capacity = n ? n : 1;
        XM_HIP_CHECK(hipMalloc((void **)&p, (n ? n : 1) * sizeof(T)));
        if (zero) {
            // hipMemset runs on the NULL stream and may return before it has executed; the solver works on its own NON-BLOCKING
            // stream, which is not ordered against the NULL stream -> wait here, or a late memset could wipe data that kernels on
            // the solver stream have already written (seen with two processes time-slicing one GPU).
            XM_HIP_CHECK(hipMemset(p, 0, (n ? n : 1) * sizeof(T)));
            XM_HIP_CHECK(hipStreamSynchronize(nullptr));
        }
    }
    // n zero-filled elements for work on stream `st`: the allocation is kept when it is large enough (a memset enqueued on `st`, no
    // hipFree / hipMalloc / device synchronisation -- a staircase solve sets its workspace up once per rank level, ~45 buffers each
    // time); `reserve` (>= n) is what a NEW allocation is sized for, so that later rank levels fit
    void ensure(size_t n, hipStream_t st, size_t reserve = 0) {
        if (p && n <= capacity) {
            count = n;
            XM_HIP_CHECK(hipMemsetAsync(p, 0, (n ? n : 1) * sizeof(T), st));
            return;
        }
        const size_t want = std::max(n, reserve);
        alloc(want);
        count = n;
    }
};

void partition_cuts(int64_t n, int world, const int64_t *weights, std::vector<int64_t> &cuts);   // xm_solver.hip
// solution recovery behind xm_recover_rotations (xm_capi.hip); variant 1 = the wavefront-per-camera projection kernel, timed when reps > 0
void recover_rotations(int64_t n, int r, const double *R, const double *s, double *rot, double *scale, int *n_negative_det, int variant,
                       int reps, double *ms_avg);
int64_t equal_range_len(int64_t n, int world);   // cameras per rank of the equal partition (even for world > 1)

class SellMatrix;   // xm_sell.h
class SymwProduct;  // xm_symw.h
class SchurOp;      // xm_schur.h

struct PointState {  // everything the gradient epilogue writes for one point (R, s)
    DevBuf<double> G, egs, S0, rgR, rgs;
};

struct TrResult {
    double primal = 0;
    int outer_iters = 0;
    int stop_reason = 0;
    bool ls_failed = false;
};

struct CertResult {
    bool accepted = false;
    double min_eig = 0, dual = 0, gap = 0;
    int lanczos_iters = 0;
};

class Context {
public:
    // comm: the communicator of THIS context (nullptr = default_comm()); the HIP device current on the calling thread is the rank's
    explicit Context(const xm_problem_t &prob, std::shared_ptr<Comm> comm = nullptr);
    int rank() const { return comm_->rank; }
    int comm_kind() const { return comm_->active() ? comm_->kind() : 0; }
    int product_kind(int o) const;   // XM_PRODUCT_* (include/xm_amd.h): the kernel that serves a tCG product of rank o
    bool sell_wpad_on() const { return wpad_on_; }   // the sliced-ELL gather of the current rank's tCG reads W at the 128-byte record pitch
    const std::string &fallback_note() const { return comm_->fallback_note; }
    int world() const { return comm_->world; }
    int64_t cameras() const { return n_; }
    int64_t edges() const { return ne_; }
    ~Context();
    void solve(const xm_options_t &opt, xm_result_t &res);
    void apply(int o, const double *W_host, double *out_host, double alpha);   // out = alpha Q W (host, column-major)
    // XM^2 re-weighting (include/xm_amd.h section 2)
    void attach_edges(int64_t ne, const int32_t *ei, const int32_t *ej, const double *M);
    void edge_residuals(double *res);
    void set_edge_weights(const double *w);
    void recover_tp(const double *rot, const double *scale, double *t, double *p);   // matrix-free storage only
    // XM^2 with the reference's residual definition (3_test_colmap_glomap.py:305-316): squared distance per edge / observation of the
    // RECOVERED solution (anchored rotations rot 3 x 3n column-major, scales) -- res: host, input order
    void edge_residuals_recovered(const double *rot, const double *scale, double *res);
    // the reference's filter (:316-324) on the device: error = w .* residual, threshold = its `pct` percentile (numpy's linear
    // interpolation between the two order statistics, found by a radix select), every edge above it gets weight 0 and Q is rebuilt.
    // Returns the threshold; *removed = edges newly removed; w_out (optional, host) = the new weights
    double xm2_filter(const double *rot, const double *scale, double pct, int64_t *removed, double *w_out);
    const std::vector<double> &weights() const { return w_cur_; }
    int64_t n_landmarks() const;
    bool schur_info(int64_t out[3], double *relres) const;   // matrix-free storage with the CG form: products, inner iterations, products at the cap

private:
    // ---- problem ------------------------------------------------------------------------------------------------
    int64_t n_ = 0;        // true cameras
    int nloc_ = 0;         // cameras owned by this GPU (padded so that every rank owns the same number)
    int cam0_ = 0;         // position of the first local camera in the padded numbering (rank * nloc_)
    int64_t g0_ = 0;       // global index of the first local camera (cam_cut_[rank])
    int64_t true_loc_ = 0; // real (non-padding) cameras on this rank
    int64_t ntot_ = 0;     // padded total = world * nloc
    int64_t ld_ = 0;       // rows of every product input W (>= 3*ntot, multiple of 128)
    int storage_ = XM_STORAGE_DENSE;
    double *dQ_ = nullptr; // dense: 3*nloc rows x ld, row-major
    bool ownQ_ = false;
    DevBuf<double> Afull_;   // multi-rank tCG: replicated image of the residual (cg_step_kernel)
    DevBuf<int64_t> rowptr_;
    DevBuf<int32_t> colidx_;
    DevBuf<double> blocks_;
    int64_t nb_loc_ = 0;
    DevBuf<int4> rowinfo_;   // block-CSR kernel: the local rows binned by their number of 16-block windows (bsr_build_rowinfo)
    std::unique_ptr<SellMatrix> sell_;   // large block-sparse Q: sliced-ELL layout (xm_sell.h); the CSR arrays stay for the fallback kernels
    int sell_gm_ = 0;
    std::unique_ptr<SchurOp> schur_;     // XM_STORAGE_SCHUR: matrix-free Q (xm_schur.h)
    // XM^2 edge description (attach_edges)
    int64_t ne_ = 0;
    DevBuf<int32_t> ei_, ej_, inc_edge_;
    DevBuf<int64_t> inc_ptr_, pos_ij_, pos_ji_, pos_d_;
    DevBuf<double> eM_, ew_, eres_;
    bool solved_ = false;   // R_/s_ hold the end point of a solve
    std::vector<double> w_cur_;          // current edge / observation weights (host mirror; empty: unknown, e.g. edges attached to a given Q)
    DevBuf<double> eerr_;                // XM^2 filter: weighted residuals
    const double *residuals_recovered_device(const double *rot, const double *scale);
    hipStream_t st_ = nullptr;
    hipStream_t st2_ = nullptr;          // local-strip product running beside the all-gather of W (multi-rank, dense)
    hipEvent_t ev_w_ = nullptr, ev_p_ = nullptr;
    DevBuf<double> Pstrip_;              // raw row sums of the local column strip
    bool w_pending_ = false;             // gather_W() was deferred into the next product()
    std::shared_ptr<Comm> comm_;
    Settings cfg_;
    PeerXchg xchg_;                      // direct tCG exchange (peer communicators): partsB lives in peer-addressable memory
    double *partsB_peer_ = nullptr;
    unsigned long long tcg_runs_ = 0;    // epoch base of the peer exchange (identical on every rank)
    std::vector<int64_t> cam_cut_;       // world + 1 camera offsets of the partition (balanced by stored blocks for block-sparse Q)
    int retraction_ = 0, grouping_ = 0;

    // ---- per-rank workspace -------------------------------------------------------------------------------------
    int o_ = 0, OP_ = 0;
    DevBuf<double> R_, s_, Rc_, sc_, W_, D_;
    DevBuf<double> wpad_;   // sliced-ELL storage, single rank, o = 3..5: the tCG's product input at a record pitch of 16 doubles (Settings.sell_wpad)
    double *wpad() const { return (wpad_.p && wpad_on_) ? wpad_.p : nullptr; }
    bool wpad_on_ = false;   // for the current rank (setup_rank)
    const double *wpad_next_ = nullptr;   // set by the outer iteration's retraction: the NEXT gradient product finds its input in the padded copy too
    PointState ps_[2];
    int cur_ = 0;
    DevBuf<double> rR_, rs_, rsB_, pR_, psA_, psB_, vR_, vs_, HvR_, Hvs_, HpR_, Hps_;
    DevBuf<double> partsA_, partsB_, partsM_;
    DevBuf<double> Prow_, Pcol_;               // symmetric product: row results and per-workgroup column partials
    DevBuf<double> ksum_;                      // column-split dense product of a small strip: partial sums per (slice, camera)
    DevBuf<unsigned int> kcount_;              //   arrival counters per camera group
    int ks_ = 1;
    std::unique_ptr<SymwProduct> symw_;        // multi-rank dense, symmetric Q: half-traffic product through a cyclic half window (xm_symw.h)
    void product_symw(int epi, int o, double alpha, const CamArgs &a);
    bool sym_ok_ = false;
    int sym_max_o_ = 4;
    bool eig_exact_ = false;                   // the last certificate ran the tridiagonalisation to completion (small n)
    double q_asym_ = 0, q_max_ = 0;
    DevBuf<TcgScal> scal_;
    unsigned long long *hstat_ = nullptr;      // host-mapped progress word (iter << 8 | status)
    unsigned long long *hstat_dev_ = nullptr;
    double *hpin_ = nullptr;                   // pinned host scratch for partial sums
    double *hpin_dev_ = nullptr;               // the same buffer as the device addresses it (kernel copies of the peer communicators)
    size_t hpin_count_ = 0;
    int nA_ = 0, nB_ = 0;                      // partial counts (whole job)
    unsigned long long outer_seq_ = 0;         // sequence number of the host-mapped outer-iteration result block

    // ---- options / statistics of the running solve -------------------------------------------------------------------
    const xm_options_t *opt_ = nullptr;
    xm_result_t *res_ = nullptr;
    bool verbose_ = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool_;
    size_t ev_used_ = 0;
    int64_t hess_launches_ = 0;
    int sym_rev_ = 1;                    // sweep direction of the next symmetric dense product (launch_qw_sym): tCG iteration parity inside the tCG, 1 elsewhere
    std::vector<float> qw_samples_;

    void init(const xm_problem_t &prob);
    void to_host(void *dst, const void *src_dev, size_t bytes);   // through the pinned staging buffer; synchronises st_
    void to_dev(void *dst_dev, const void *src, size_t bytes);
    void ensure_pinned(size_t doubles);
    DevBuf<double> lzV_, lzc_, lzw_, lzc2_, lzab_, lzscr_;      // Lanczos workspace (grow-only)
    DevBuf<double> recW_;                                       // product input built from a RECOVERED solution (XM^2 residuals); never a solver buffer:
                                                                // the Lanczos vectors rely on their padding beyond 3n staying zero
    int64_t pos_of(int64_t g) const;   // global camera -> position in the padded numbering of the replicated vectors
    void release_raw();   // frees the raw (non-RAII) resources: dQ_, stream, mapped / pinned host memory, events
    void setup_rank(int o);
    void upload_point(const std::vector<double> &R_cm, int o, const std::vector<double> &s_ex);
    void download_point(std::vector<double> &R_cm, std::vector<double> &s_ex);
    CamArgs cam_args(int state) const;
    int prod_grid() const;
    int tcg_blocks() const;   // workgroups of cg_step_kernel == |r|^2 partial sums per rank
    void product(int epi, int o, double alpha, const CamArgs &a);
    void gather_W();          // all-gather of the product input; deferred into the next product() when the overlap applies
    void flush_gather();      // perform a deferred gather now
    bool overlap_applies() const;
    void eval_point(int state, const double *Rp, const double *sp, double &f, double &rr);
    double sum_parts(const double *dparts, int count);
    int run_tcg(double rr, double delta, TcgScal &fin, int adopted = 0);   // adopted: iterations of THIS tCG already enqueued speculatively (enqueue_spec_tcg)
    int enqueue_spec_tcg();      // single GPU: tcg_init + the first iteration(s) of the next tCG from the CANDIDATE point, gated on the device (SpecCtl)
    void tcg_enqueue_iteration(int i, bool profile);   // one iteration = Hessian product (+ exchange) + cg_step
    bool spec_applies() const;
    DevBuf<SpecCtl> spec_;
    unsigned int tcg_seq_ = 0;   // sequence number of the current truncated-CG run (TcgScal.seq, progress word)
    volatile double *wait_outer_result();
    bool stream_idle(std::chrono::steady_clock::time_point t_wait, const char *what);   // true: drained; throws on a device error / watchdog
    bool agree_any(bool local);
    void drain_events();
    void finish_profile();
    TrResult trust_region(int o, double &gradtol, double linesearch_step, const std::vector<double> &v_dir, double max_time);
    // the same loop with the outer iteration on the DEVICE (xm_kernels.hip: outer_step_kernel): the host enqueues (product, step) launch pairs ahead
    // and watches a progress word; entered from trust_region() after the first cost / gradient (f, rr at the starting point)
    bool device_outer_applies(int o) const;
    TrResult trust_region_device(int o, double &gradtol, double f, double rr, double delta, double delta_bar, double max_time);
    DevBuf<double> trace_dev_;           // device copy of the outer-iteration trace (kMaxOuter records)
    DevBuf<int> stop_req_;               // set by the host when its time limit has expired
    DevBuf<OuterScal> oscal_;            // trust-region state of the device-driven outer iteration, two parity copies next to scal_
    unsigned int outer_run_ = 0;         // run number in the progress word (hstat_[24])
    CertResult certificate(int o, double primal, std::vector<double> &v_out);
    int lanczos_min(std::vector<double> &x_out, double &theta, int &iters, double &resid);   // 0 converged, 1 not
    void log(const char *fmt, ...) const;
};

// single-process multi-GPU driver (xm_team.hip): `n_gpus` Contexts, one host thread and one device each, joined by a peer group
class Team {
public:
    Team(const xm_problem_t &prob, int n_gpus, int gpu_map);
    ~Team();
    Team(const Team &) = delete;
    Team &operator=(const Team &) = delete;
    int world() const;
    int comm_kind() const;                       // Comm::kind() of the ranks' communicators
    int product_kind(int o) const;               // of rank 0's context
    const std::string &fallback_note() const;    // why the direct peer exchange was given up (empty: it was not)
    void solve(const xm_options_t &opt, xm_result_t &res);
    void attach_edges(int64_t ne, const int32_t *ei, const int32_t *ej, const double *M);
    void edge_residuals(double *res);
    void set_edge_weights(const double *w);
    void edge_residuals_recovered(const double *rot, const double *scale, double *res);
    double xm2_filter(const double *rot, const double *scale, double pct, int64_t *removed, double *w_out);
    int64_t cameras() const;
    std::vector<double> weights() const;
private:
    struct Impl;
    std::unique_ptr<Impl> p_;
    void shutdown();
};

}  // namespace xm
