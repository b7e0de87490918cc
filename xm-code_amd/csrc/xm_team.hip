// xm_team.hip — SINGLE-PROCESS multi-GPU: the row partition of xm_solver.hip driven by one host thread per GPU inside the caller's
// process, so that the reference's own call — XM.solve(path, ...) from a single-process script (XM_main.cu:180,403-408;
// 1_test_solve.py:42, 3_test_colmap_glomap.py:285) — uses all the GPUs of the node without a launcher (SURVEY.md 8b "Threading":
// "internally one host thread per GPU ... communicator created lazily and cached").
//
// A Team owns `world` worker threads.  Worker r makes device r current (or device 0 for every rank with gpu_map 1: "virtual
// devices", the whole multi-GPU path on a 1-GPU box), joins the peer group (xm_comm.hip: direct peer writes) and constructs ITS
// Context, which uploads only its own camera rows.  Every call fans out to the workers; each runs the ordinary single-rank code path
// of Context with its communicator; results are identical on every rank (all branch decisions are taken on identically summed
// data), rank 0's are returned.  A failure on one rank aborts the group so that the others leave their collectives with XM_ERR_COMM
// instead of waiting for ever.
#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>

#include "xm_solver.h"

namespace xm {

struct Team::Impl {
    int world = 1;
    std::vector<int> device;
    std::shared_ptr<PeerGroup> group;
    int kind = 0;               // Comm::kind() of the transport in use (3 direct peer writes | 1 RCCL)
    std::string fallback;       // why the direct peer writes were given up (empty: they were not)
    std::vector<std::unique_ptr<Context>> ctx;
    std::vector<std::shared_ptr<Comm>> comm;
    // worker pool
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::function<void(int)> job;
    unsigned long long gen = 0;
    int pending = 0;
    bool quit = false, broken = false;
    std::vector<std::exception_ptr> err;

    void worker(int r) {
        unsigned long long seen = 0;
        bool dev_set = false;
        for (;;) {
            std::function<void(int)> f;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_job.wait(lk, [&] { return quit || gen != seen; });
                if (quit) return;
                seen = gen;
                f = job;
            }
            std::exception_ptr e;
            try {
                if (!dev_set) { XM_HIP_CHECK(hipSetDevice(device[(size_t)r])); dev_set = true; }   // per-thread state of the HIP runtime
                f(r);
            } catch (...) {
                e = std::current_exception();
                if (group) peer_group_abort(group);   // wake the other ranks out of their waits
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                err[(size_t)r] = e;
                if (--pending == 0) cv_done.notify_all();
            }
        }
    }
    // run f(rank) on every worker, wait for all, rethrow the first failure in rank order (a rank that merely noticed the abort
    // reports XM_ERR_COMM "aborted"; prefer the original error)
    void run(const std::function<void(int)> &f) {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (broken) throw Error(XM_ERR_COMM, "multi-GPU context: a previous call failed on one rank; the group was aborted (create a new context)");
            job = f;
            pending = world;
            for (auto &e : err) e = nullptr;
            ++gen;
        }
        cv_job.notify_all();
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
        std::exception_ptr first, first_real;
        for (int r = 0; r < world; ++r) {
            if (!err[(size_t)r]) continue;
            if (!first) first = err[(size_t)r];
            if (!first_real) {
                try { std::rethrow_exception(err[(size_t)r]); }
                catch (const Error &x) { if (std::string(x.what()).find("aborted") == std::string::npos) first_real = err[(size_t)r]; }
                catch (...) { first_real = err[(size_t)r]; }
            }
        }
        if (first) { broken = true; lk.unlock(); std::rethrow_exception(first_real ? first_real : first); }
    }
};

Team::Team(const xm_problem_t &prob, int n_gpus, int gpu_map) : p_(new Impl) {
    Impl &t = *p_;
    if (n_gpus < 2 || n_gpus > kMaxPeers) throw Error(XM_ERR_ARG, "n_gpus must be 2.." + std::to_string(kMaxPeers) + " for a multi-GPU context");
    int ndev = 0;
    XM_HIP_CHECK(hipGetDeviceCount(&ndev));
    if (gpu_map == 0 && n_gpus > ndev)
        throw Error(XM_ERR_ARG, "n_gpus = " + std::to_string(n_gpus) + " but only " + std::to_string(ndev) + " HIP devices are visible (gpu_map = 1 runs every rank on device 0)");
    if (prob.storage == XM_STORAGE_DENSE && prob.q_on_device) throw Error(XM_ERR_ARG, "q_on_device needs a single-GPU context");
    t.world = n_gpus;
    t.device.resize((size_t)n_gpus);
    for (int r = 0; r < n_gpus; ++r) t.device[(size_t)r] = (gpu_map == 1) ? 0 : r;
    const Settings cfg = Settings::resolve(prob.tuning);
    // device-side waits give up after min(watchdog / 3, 30 s): long enough for a peer that lags, short enough not to look hung, and well
    // before the host-side watchdog of the rank that waits (so that the failure is reported as what it is: XM_ERR_COMM)
    t.ctx.resize((size_t)n_gpus);
    t.comm.resize((size_t)n_gpus);
    t.err.resize((size_t)n_gpus);
    for (int r = 0; r < n_gpus; ++r) t.th.emplace_back([this, r] { p_->worker(r); });
    // Transport ladder.  (1) direct peer writes (xm_comm.hip: PeerComm): every rank maps the peers' fine-grained arenas and the transport
    // has to pass its self-test -- all-gathers with known contents through both read paths -- on THIS machine before a solver relies on
    // it.  (2) RCCL, one communicator per host thread (the library's own transport over xGMI; one all-gather per tCG iteration instead
    // of the fused exchange).  (3) XM_ERR_COMM naming both reasons.  xm_tuning_t.exchange = 3 asks for (2) directly
    // (exchange = 1 keeps the peer transport and only takes the exchange out of the tCG kernel).
    // device-side waits give up after min(watchdog / 3, 30 s): long enough for a peer that lags, short enough not to look hung, and well
    // before the host-side watchdog of the rank that waits (so that the failure is reported as what it is: XM_ERR_COMM)
    std::string why_peer;
    bool have = false;
    if (cfg.exchange != 3) {
        try {
            t.group = peer_group_create(n_gpus, t.device.data(), std::min(30.0, cfg.watchdog_s / 3.0));
            t.run([&](int r) {
                t.comm[(size_t)r] = peer_comm_create(t.group, r);
                if (!peer_comm_selftest(*t.comm[(size_t)r])) throw Error(XM_ERR_COMM, "peer communicator: the transport self-test failed on rank " + std::to_string(r));
            });
            have = true;
            t.kind = 3;
        } catch (const std::exception &e) {
            why_peer = e.what();
            int dev_keep = 0;
            (void)hipGetDevice(&dev_keep);   // the caller's current device survives the clean-up (later xm_dev_alloc / torch work on this thread)
            for (int r = 0; r < n_gpus; ++r) { (void)hipSetDevice(t.device[(size_t)r]); (void)hipDeviceSynchronize(); t.comm[(size_t)r].reset(); }
            (void)hipSetDevice(dev_keep);
            t.group.reset();
            { std::lock_guard<std::mutex> lk(t.mu); t.broken = false; }
        }
    } else {
        why_peer = "RCCL requested (exchange = 3)";
    }
    if (!have) {
        try {
            unsigned char id[128];
            comm_unique_id(id);
            t.run([&](int r) { t.comm[(size_t)r] = rccl_comm_create(r, n_gpus, id); });
            for (auto &c : t.comm) c->fallback_note = "direct peer writes not used (" + why_peer + "): RCCL all-gather per exchange";
            t.kind = 1;
            t.fallback = t.comm[0]->fallback_note;
        } catch (const std::exception &e) {
            const std::string why_rccl = e.what();
            shutdown();
            throw Error(XM_ERR_COMM, "multi-GPU context: no transport between the " + std::to_string(n_gpus) + " ranks -- direct peer writes: " + why_peer +
                                         "; RCCL: " + why_rccl);
        }
    }
    try {
        t.run([&](int r) { t.ctx[(size_t)r].reset(new Context(prob, t.comm[(size_t)r])); });
    } catch (...) {
        shutdown();
        throw;
    }
}

void Team::shutdown() {
    Impl &t = *p_;
    {
        std::lock_guard<std::mutex> lk(t.mu);
        t.quit = true;
    }
    t.cv_job.notify_all();
    for (auto &th : t.th) if (th.joinable()) th.join();
    t.th.clear();
    int dev_keep = 0;
    (void)hipGetDevice(&dev_keep);   // restored below: the caller's thread keeps its current device
    // contexts before communicators (a Context frees device memory its communicator's peers may still address: drain first)
    for (size_t r = 0; r < t.ctx.size(); ++r) {
        if (!t.ctx[r]) continue;
        (void)hipSetDevice(t.device[r]);
        (void)hipDeviceSynchronize();
    }
    for (size_t r = 0; r < t.ctx.size(); ++r) { (void)hipSetDevice(t.device[r]); t.ctx[r].reset(); }
    for (size_t r = 0; r < t.comm.size(); ++r) { (void)hipSetDevice(t.device[r]); t.comm[r].reset(); }
    (void)hipSetDevice(dev_keep);
}

Team::~Team() { shutdown(); }

int Team::world() const { return p_->world; }
int Team::comm_kind() const { return p_->kind; }
int Team::product_kind(int o) const { return p_->ctx[0] ? p_->ctx[0]->product_kind(o) : 0; }
const std::string &Team::fallback_note() const { return p_->fallback; }

void Team::solve(const xm_options_t &opt, xm_result_t &res) {
    Impl &t = *p_;
    const int64_t n = t.ctx[0]->cameras();
    const size_t rmax = (size_t)std::max(3u, opt.max_rank) + 1;
    std::vector<xm_result_t> rr((size_t)t.world);
    std::vector<std::vector<double>> Rb((size_t)t.world), sb((size_t)t.world);
    std::vector<xm_options_t> oo((size_t)t.world, opt);
    for (int r = 0; r < t.world; ++r) {
        std::memset(&rr[(size_t)r], 0, sizeof(xm_result_t));
        if (r == 0) { rr[0].R = res.R; rr[0].s = res.s; }
        else {
            Rb[(size_t)r].assign((size_t)3 * n * rmax, 0.0); sb[(size_t)r].assign((size_t)n, 1.0);
            rr[(size_t)r].R = Rb[(size_t)r].data(); rr[(size_t)r].s = sb[(size_t)r].data();
            oo[(size_t)r].trace = nullptr; oo[(size_t)r].trace_cap = 0;   // the trace is rank 0's (identical everywhere)
        }
    }
    t.run([&](int r) { t.ctx[(size_t)r]->solve(oo[(size_t)r], rr[(size_t)r]); });
    res = rr[0];
}

void Team::attach_edges(int64_t ne, const int32_t *ei, const int32_t *ej, const double *M) {
    p_->run([&](int r) { p_->ctx[(size_t)r]->attach_edges(ne, ei, ej, M); });
}
void Team::edge_residuals(double *res) {
    Impl &t = *p_;
    std::vector<std::vector<double>> tmp((size_t)t.world);
    const int64_t ne = t.ctx[0]->edges();
    t.run([&](int r) {
        double *out = res;
        if (r != 0) { tmp[(size_t)r].assign((size_t)std::max<int64_t>(ne, 1), 0.0); out = tmp[(size_t)r].data(); }
        t.ctx[(size_t)r]->edge_residuals(out);
    });
}
void Team::set_edge_weights(const double *w) {
    p_->run([&](int r) { p_->ctx[(size_t)r]->set_edge_weights(w); });
}
// every rank evaluates all residuals and the same order statistic from the caller's recovered solution; rank 0's outputs are returned
void Team::edge_residuals_recovered(const double *rot, const double *scale, double *res) {
    Impl &t = *p_;
    std::vector<std::vector<double>> tmp((size_t)t.world);
    const int64_t ne = t.ctx[0]->edges();
    t.run([&](int r) {
        double *out = res;
        if (r != 0) { tmp[(size_t)r].assign((size_t)std::max<int64_t>(ne, 1), 0.0); out = tmp[(size_t)r].data(); }
        t.ctx[(size_t)r]->edge_residuals_recovered(rot, scale, out);
    });
}
double Team::xm2_filter(const double *rot, const double *scale, double pct, int64_t *removed, double *w_out) {
    Impl &t = *p_;
    std::vector<double> thr((size_t)t.world, 0.0);
    std::vector<int64_t> rm((size_t)t.world, 0);
    t.run([&](int r) { thr[(size_t)r] = t.ctx[(size_t)r]->xm2_filter(rot, scale, pct, &rm[(size_t)r], r == 0 ? w_out : nullptr); });
    for (int r = 1; r < t.world; ++r)
        if (thr[(size_t)r] != thr[0] || rm[(size_t)r] != rm[0]) throw Error(XM_ERR_COMM, "xm2_filter: the ranks disagree on the threshold");
    if (removed) *removed = rm[0];
    return thr[0];
}
int64_t Team::cameras() const { return p_->ctx[0]->cameras(); }
std::vector<double> Team::weights() const { return p_->ctx[0]->weights(); }

// micro-benchmark of the peer all-gather: `world` threads, each with its own stream and communicator, `reps` collectives back to back
double peer_allgather_bench(int world, int gpu_map, int64_t count, int reps) {
    if (world < 2 || world > kMaxPeers || count < 1 || reps < 1) throw Error(XM_ERR_ARG, "peer_allgather_bench: bad argument");
    int ndev = 0;
    XM_HIP_CHECK(hipGetDeviceCount(&ndev));
    if (gpu_map == 0 && world > ndev) throw Error(XM_ERR_ARG, "peer_allgather_bench: not enough devices (gpu_map 1 = virtual devices)");
    std::vector<int> dev((size_t)world);
    for (int r = 0; r < world; ++r) dev[(size_t)r] = gpu_map == 1 ? 0 : r;
    auto group = peer_group_create(world, dev.data(), 20.0);
    std::vector<double> us((size_t)world, 0.0);
    std::vector<std::exception_ptr> err((size_t)world);
    std::vector<std::thread> th;
    for (int r = 0; r < world; ++r)
        th.emplace_back([&, r] {
            try {
                XM_HIP_CHECK(hipSetDevice(dev[(size_t)r]));
                auto comm = peer_comm_create(group, r);
                comm->reserve((size_t)count * world + 64);
                hipStream_t st;
                XM_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
                DevBuf<double> buf;
                buf.alloc((size_t)count * world);
                hipEvent_t e0, e1;
                XM_HIP_CHECK(hipEventCreate(&e0)); XM_HIP_CHECK(hipEventCreate(&e1));
                for (int i = 0; i < 5; ++i) comm->allgather(buf.p, (size_t)count, st);
                XM_HIP_CHECK(hipStreamSynchronize(st));
                comm->host_barrier();
                XM_HIP_CHECK(hipEventRecord(e0, st));
                for (int i = 0; i < reps; ++i) comm->allgather(buf.p, (size_t)count, st);
                XM_HIP_CHECK(hipEventRecord(e1, st));
                XM_HIP_CHECK(hipEventSynchronize(e1));
                float ms = 0;
                XM_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
                us[(size_t)r] = (double)ms * 1e3 / reps;
                comm->check_device_error();
                comm->host_barrier();
                (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
                (void)hipStreamDestroy(st);
            } catch (...) {
                err[(size_t)r] = std::current_exception();
                peer_group_abort(group);
            }
        });
    for (auto &t : th) t.join();
    for (auto &e : err) if (e) std::rethrow_exception(e);
    return us[0];
}

}  // namespace xm
