#!/usr/bin/env python3
"""bench.py — BM iters/sec + wall-clock-to-KKT of the XM staircase solve on MI355X (BASELINE.json metric).

One "step" = one complete Riemannian-staircase solve (rank 3.. max_rank, RTR-tCG + dual certificate) of the workload
with Q already resident in HBM.  value = tCG inner iterations (each = one Q*W Hessian product, the reference's
"Total iteration", trustregion.h:666/711) per second over the K timed solves; ms_per_step = wall-clock-to-KKT of one
solve.  Workload at every N: the configuration the metric is quoted on, "Venice-1778": the reference ships no BAL Q
(SURVEY.md F7), so it is the seeded dense SBA-like generator of SURVEY.md §8d with n = 1778 cameras
(tests/xm_testlib.py:gen_dense).  For N > 1 the same problem is row-partitioned over the ranks (strong scaling).

Launch: python bench.py [--gpus N --steps K --warmup W].
  * under torch.distributed.run (one rank per GPU, what the driver does): one process per GPU, RCCL all-gathers inside the solver;
  * plain `python bench.py --gpus N`: NO launcher -- the library's single-process multi-GPU mode (xm_problem_t.n_gpus: one host thread
    per GPU, direct peer-write exchange fused into the tCG).  When the box has fewer than N GPUs the N ranks are "virtual devices" on
    device 0 (gpu_map = 1): a functional run of the N-rank flow, flagged as such in the JSON line, not a scaling measurement.
The K timed solves rotate through the three summation groupings of xm_options_t.sum_grouping (step i uses grouping i mod 3): every
grouping is a fixed, bit-reproducible order, but the iteration count of this staircase moves by +-10 % with the last bits of the
sums (ranks 3 and 4 end at saddle points), so one grouping alone would make the headline a lottery draw.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "xm-code_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)



def host_cpu_budget():
    """CPUs this process may really keep busy: physical cores, capped by the container's CPU quota (cgroup cpu.max).  The GPU boxes of
    this pool show 256 logical CPUs and a quota of 16: 128 busy OpenMP threads are throttled to an eighth of their time (the host
    product then takes 2.4 ms instead of 0.4 ms)."""
    logical = os.cpu_count() or 8
    budget = max(1, logical // 2)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None:
        budget = max(1, min(budget, int(quota)))
    return budget, quota, logical


def _argv_gpus():
    for i, a in enumerate(sys.argv):
        if a == "--gpus" and i + 1 < len(sys.argv):
            return sys.argv[i + 1]
        if a.startswith("--gpus="):
            return a.split("=", 1)[1]
    return "1"


_HOST_BUDGET = host_cpu_budget()
_WS = int(os.environ.get("WORLD_SIZE", "1"))
if _WS > 1:   # torchrun pins OMP_NUM_THREADS=1; give each rank its share of the usable host cores
    os.environ["OMP_NUM_THREADS"] = str(max(1, _HOST_BUDGET[0] // _WS))
else:
    os.environ.setdefault("OMP_NUM_THREADS", str(_HOST_BUDGET[0]))
os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")              # control-plane rendezvous on loopback (hostname may not resolve)
os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")              # RCCL bootstrap of the single-node communicator likewise
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: required by RCCL on this driver (multi-process runs)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")           # virtual devices (N ranks on one GPU) need a hardware queue per rank's stream (8 ranks: 16 queues)
if "WORLD_SIZE" not in os.environ and _argv_gpus() != "1":   # single-process multi-GPU, possibly N virtual devices on ONE GPU:
    os.environ.setdefault("HSA_ENABLE_SDMA", "0")                                # their streams must not share the copy engine's in-order queue (copies become kernels)
# cpu_baseline (rank 0 of a 1-GPU run only): OpenMP threads stay where they first touched their share of Q, spread over the L3 slices /
# memory controllers.  NOT in multi-rank runs: a runtime that binds threads also binds the process's INITIAL thread to the first place
# -- core 0 in every rank's process -- and every thread created later (the solver's host threads, the HIP runtime's) inherits that
# mask: N busy-polling ranks on one core (found on 4 virtual ranks: device-side waits expired while the peers' host threads queued
# for their time slice).  _FULL_AFFINITY is restored for the GPU part of every run and the binding re-applied for the host leg.
_FULL_AFFINITY = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
if _WS == 1 and _argv_gpus() == "1":
    os.environ.setdefault("OMP_PROC_BIND", "spread")
    os.environ.setdefault("OMP_PLACES", "cores")

import numpy as np  # noqa: E402

import xmamd  # noqa: E402  (loads libxm_amd.so before torch so that one HIP runtime serves the process)


def _unbind_main_thread():
    """give the calling thread (and the threads it creates from now on) the process's original CPU mask back; returns the bound mask"""
    if _FULL_AFFINITY is None:
        return None
    bound = os.sched_getaffinity(0)
    if bound != _FULL_AFFINITY:
        os.sched_setaffinity(0, _FULL_AFFINITY)
    return bound

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def workload(name):
    import xm_testlib as tl
    if name == "venice1778":
        return dict(kind="dense", n=1778, seed=1778, max_rank=5, tol=1e-6, lam=0.0,
                    desc="Venice-1778-size dense SBA-like Q (G_dense(1778, seed 1778), SURVEY §8d C4), staircase max_rank 5, tol 1e-6")
    if name == "final13682":   # Rome-scale (>= 10k cameras): view-graph Q, stored dense on the device (13.5 GB) or as BSR3
        # lam: a rotation-only view-graph Q leaves the scales free to collapse; with lam = 30 the iteration stalls in between
        # (|grad| ~ 70, every rank runs into the reference's 1000-outer-iteration cap, scripts/sweep_vg.py), with lam = 1000 the
        # rank-3 optimum is certified
        return dict(kind="vg", n=13682, deg=30, sigma=0.05, seed=13682, max_rank=5, tol=1e-6, lam=1000.0,
                    desc="Final-13682-size view-graph Q G_vg(13682, deg 30, sigma 0.05), lam 1000")
    if name == "vg100k":
        return dict(kind="vg", n=100000, deg=50, sigma=0.05, seed=100000, max_rank=5, tol=1e-6, lam=1000.0,
                    desc="synthetic 100k-camera Erdos-Renyi view-graph Q G_vg(100000, deg 50, sigma 0.05), lam 1000")
    if name == "dubrovnik356":
        return dict(kind="dense", n=356, seed=356, max_rank=5, tol=1e-6, lam=0.0, desc="Dubrovnik-356-size dense Q")
    if name == "ladybug49":
        return dict(kind="dense", n=49, seed=49, max_rank=5, tol=1e-6, lam=0.0, desc="Ladybug-49-size dense Q")
    raise SystemExit(f"unknown workload {name}")


def source_sha256():
    """hash of the kernel / solver sources: a recorded PMC profile is quoted only for the code it was measured on"""
    import glob, hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "xm-code_amd", "csrc", "*"))):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()


def recorded_traffic(leg, world=1):
    """roofline.traffic: HBM-side bytes per product of the dominant kernel(s) from a rocprofv3 --pmc FETCH_SIZE pass over THIS command
    (scripts/pmc_legs.py; round 1-3: scripts/pmc_hess.sh).  Only a profile stamped with the hash of the current sources is quoted;
    otherwise None + the reason.  Returns (bytes, source, traced average duration in microseconds or None)."""
    import glob
    if world != 1:
        return None, "no PMC pass recorded for this GPU count", None
    sha = source_sha256()
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_fetch_%s.json" % leg)), reverse=True)
    if leg == "venice":
        cands += sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_fetch_hess_bench.json")), reverse=True)
    for f in cands:
        try:
            pmc = json.load(open(f))
        except Exception:
            continue
        if pmc.get("source_sha256") != sha:
            continue
        src = os.path.relpath(f, ROOT) + (" (rocprofv3 --pmc FETCH_SIZE pass over this command's own launches on these sources, no-ops dropped; "
                                          "x1024 x2 per MI355X_MICROARCH.md; a --kernel-trace pass of the same command gives traced_us)")
        if "per_product" in pmc:
            return pmc["per_product"]["hbm_side_bytes"], src, pmc["per_product"].get("traced_us")
        return pmc["hess_all_ranks_weighted"]["hbm_side_bytes_per_real_launch"], src, None
    return None, "no PMC profile of leg '%s' stamped with the current source hash %s (run scripts/pmc_legs.py on the GPU box)" % (leg, sha[:12]), None


RECORDED_ORACLE = {   # workload -> (recorded run of the CPU oracle, its anchored rotations, every k-th camera kept)
    "venice1778": ("venice1778_oracle.json", "venice1778_oracle_rot.npy", 1),
    "final13682": ("rome13682_oracle.json", "rome13682_oracle_rot.npy", 1),
    "vg100k": ("vg100k_oracle.json", "vg100k_oracle_rot_every8.npy", 8),
}


def parity_vs_recorded_oracle(workload_name, R, s, primal):
    """SURVEY 8d 'parity numbers between CPU and GPU results', computed IN THIS RUN: the timed GPU solution against the CPU oracle's recorded
    solve of the same Q (tests/golden/synth, recorded by scripts/record_oracle_large.py) -- anchored rotations (north_star's <= 1e-6
    relative Frobenius), a sample of 3x3 blocks R_i^T R_j of the rotation Gram matrix, and the optimum."""
    import xm_testlib as tl
    if workload_name not in RECORDED_ORACLE:
        return None
    fj, fr, every = RECORDED_ORACLE[workload_name]
    g = os.path.join(ROOT, "tests", "golden", "synth")
    if not (os.path.exists(os.path.join(g, fj)) and os.path.exists(os.path.join(g, fr))):
        return None
    c = json.load(open(os.path.join(g, fj)))
    ref = np.load(os.path.join(g, fr))
    rot, _ = tl.recover_rotations(R, s)
    n = rot.shape[1] // 3
    sub = rot.reshape(3, n, 3)[:, ::every, :].reshape(3, -1) if every > 1 else rot
    ref = ref.reshape(3, -1)
    m = sub.shape[1] // 3
    idx = np.random.default_rng(12345).integers(0, m, size=(20000, 2))
    A, B = sub.reshape(3, m, 3), ref.reshape(3, m, 3)
    ga = np.einsum("aki,akj->kij", A[:, idx[:, 0], :], A[:, idx[:, 1], :])
    gb = np.einsum("aki,akj->kij", B[:, idx[:, 0], :], B[:, idx[:, 1], :])
    return {"rotations_rel_fro": tl.rel_fro(sub, ref), "gram_sample_rel_fro": tl.rel_fro(ga, gb),
            "f_rel": abs(primal - c["f"]) / max(abs(c["f"]), 1e-300), "tolerance": 1e-6,
            "against": "tests/golden/synth/%s + %s: the CPU oracle's recorded solve of the same Q (rank %s, %d tCG iterations)" % (fj, fr, c.get("rank", 3), c["tcg"]),
            "cpu_wallclock_to_kkt_s": c.get("seconds"), "cpu_threads": c.get("threads"),
            "cpu_wallclock_provenance": "oracle/xm_oracle.c to its certificate on %s OpenMP threads of the build container (scripts/record_oracle_large.py); "
                                        "not re-run here: %.0f s" % (c.get("threads"), c.get("seconds", 0.0))}


def host_l3_reachable_bytes(threads):
    """last-level cache the OpenMP team of the host leg can reach: the distinct L3 slices (sysfs cache/index3, told apart by their
    shared_cpu_list) of the CPUs the process may run on; `threads` threads spread over the slices (OMP_PROC_BIND=spread) reach at most one
    slice each.  None when sysfs does not say."""
    try:
        cpus = sorted(_FULL_AFFINITY if _FULL_AFFINITY is not None else os.sched_getaffinity(0))
        seen, sizes = set(), []
        for c in cpus:
            base = "/sys/devices/system/cpu/cpu%d/cache/index3/" % c
            shared = open(base + "shared_cpu_list").read().strip()
            if shared in seen:
                continue
            seen.add(shared)
            sz = open(base + "size").read().strip()
            sizes.append(int(sz[:-1]) * (1 << 10 if sz[-1] in "Kk" else 1 << 20) if sz[-1] in "KkMm" else int(sz))
        sizes.sort(reverse=True)
        return sum(sizes[:max(1, threads)]) or None
    except Exception:
        return None


def cpu_baseline(Q, wl, budget_s, bsr=None):
    """oracle (CPU restatement) on the SAME Q/options, bounded by the reference's own max_time mechanism."""
    from oracle import xm_oracle as xo
    n = wl["n"]
    R0 = np.tile(np.eye(3), (n, 1))
    t0 = time.time()
    if bsr is not None:   # block-sparse workloads: the oracle's test-only BSR3 product (the dense matrix would not fit)
        _, _, primal, _, st = xo.trustregion_bsr(bsr[0], bsr[1], bsr[2], R0, np.ones(n), lam=wl["lam"], gradtol=wl["tol"], maxtime=budget_s)
    else:
        xo.numa_prepare(Q)   # NUMA-distributed first-touch copy: the host product then streams from every memory controller
        try:
            _, _, primal, _, st = xo.trustregion(Q, R0, np.ones(n), lam=wl["lam"], gradtol=wl["tol"], maxtime=budget_s)
        finally:
            xo.numa_release()
    el = time.time() - t0
    its = st["tcg_iters"]
    return dict(value=its / max(st["seconds"], 1e-9), unit="tCG iters/s", cores=xo.num_threads(), kind="port",
                sample=f"RANK-3 STAGE ONLY (the GPU line is the whole staircase): rank-3 trust region of the same Q on the host for <= {budget_s:.0f}s of its own max_time clock: "
                       f"{its} tCG iters / {st['outer_iters']} outer in {st['seconds']:.1f}s (stop {st['stop_reason']}), "
                       f"Q*W {st['qw_seconds'] / max(st['qw_products'], 1) * 1e3:.2f} ms each",
                qw_ms=st["qw_seconds"] / max(st["qw_products"], 1) * 1e3, wall_s=el,
                host="%d OpenMP threads (OMP_PLACES=%s, OMP_PROC_BIND=%s) on %d logical CPUs, container CPU quota %s" % (
                    xo.num_threads(), os.environ.get("OMP_PLACES"), os.environ.get("OMP_PROC_BIND"), _HOST_BUDGET[2],
                    "none" if _HOST_BUDGET[1] is None else "%.1f CPUs" % _HOST_BUDGET[1]),
                qw_host_GBs=((76.0 * bsr[1].size + 4 * (n + 1)) if bsr is not None else 8.0 * (3 * n) ** 2) / 1e9 /
                            max(st["qw_seconds"] / max(st["qw_products"], 1), 1e-12),
                **_residency(((76.0 * bsr[1].size) if bsr is not None else 8.0 * (3 * n) ** 2), xo.num_threads()))


def cpu_kkt(Q, wl, budget_s, tl, gpu_sol, gpu_info):
    """SURVEY 8d verbatim: the CPU baseline on the SAME Q, options and stop rule as the timed GPU solves -- the complete staircase with the dual
    certificate (checkeig.h:300-316) -- timed on this node's host threads in this run.  The oracle's eigen step runs through LAPACK dsyevd
    (the closest CPU analogue of cusolverDnXsyevd, Dense/eig.h:35-73; tests/test_oracle.py pins it against the restated tred2 / tql2, which
    needs an hour at 5334 rows); the trust region is the line-cited restatement as everywhere.  Bounded by the reference's own max_time."""
    from oracle import xm_oracle as xo
    xo.use_lapack_eig(True)
    xo.numa_prepare(Q)
    t0 = time.perf_counter()
    try:
        Ro, so, io = xo.solve(Q, wl["max_rank"], wl["tol"], wl["lam"], float(int(budget_s)), trace=4000)
    finally:
        xo.numa_release()
        xo.use_lapack_eig(False)
    c_s = time.perf_counter() - t0
    done = int(io["status"]) == 1 and int(io.get("stop_reason", 0)) != 11
    fo = float(io["trace"][-1, 0]) if "trace" in io and len(io["trace"]) else float("nan")
    R, s = gpu_sol[0], gpu_sol[1]
    return {"wallclock_to_kkt_s": c_s if done else None,
            "wallclock_to_kkt_note": (None if done else "the oracle did not reach a certified point inside --cpu-kkt-seconds %.0f (status %d, stop reason %d after %.1f s)" %
                                      (budget_s, int(io["status"]), int(io.get("stop_reason", 0)), c_s)),
            "threads": xo.num_threads(), "rank": int(io["rank"]), "status": int(io["status"]), "tcg_iters": int(io["tcg_iters"]), "outer_iters": int(io["outer_iters"]),
            "primal": fo, "min_eig": float(io["cert"]["min_eig"]), "qw_products": int(io["qw_products"]),
            "trust_region_s_total": float(io["seconds"]), "qw_s_total": float(io["qw_seconds"]), "certificates_and_rest_s": c_s - float(io["seconds"]),
            "eigen_step": "LAPACK dsyevd through scipy (xm_oracle.use_lapack_eig); trust region, multipliers, acceptance rule: oracle/xm_oracle.c",
            "gpu_wallclock_to_kkt_s": gpu_info["seconds_last_timed_solve"], "gpu_tcg_iters": gpu_info["tcg_iters"],
            "parity_vs_timed_gpu_solution": {"rotations_rel_fro": tl.rotation_parity(R, s, Ro, so), "gram_rel_fro": tl.rel_fro(tl.gram(R, s), tl.gram(Ro, so)),
                                             "f_rel": abs(gpu_info["primal"] - fo) / max(abs(fo), 1e-300), "tolerance": 1e-6},
            "note": "same process, same node, same Q, options and stop rule (|grad| < tol per rank, then the certificate's acceptance rule); "
                    "the CPU path is its own (one grouping of the partial sums), the GPU line cycles three"}


def library_build_stamp():
    """which binary was benchmarked: a stale libxm_amd.so cannot be timed silently"""
    import hashlib
    path = xmamd.LIB_PATH
    try:
        st = os.stat(path)
        h = hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
        src_newer = [os.path.basename(f) for f in __import__("glob").glob(os.path.join(ROOT, "xm-code_amd", "csrc", "*")) if os.stat(f).st_mtime > st.st_mtime + 1.0]
        return {"library": os.path.relpath(path, ROOT), "sha256_16": h, "bytes": st.st_size, "mtime_utc": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime(st.st_mtime)),
                "version": xmamd.lib().xm_version().decode(), "sources_sha256_16": source_sha256()[:16],
                "sources_newer_than_library": src_newer or None}
    except OSError as e:
        return {"library": path, "error": str(e)}


def _residency(matrix_bytes, threads):
    """where the host product's matrix can live: compared with the L3 the bound threads reach (not with a constant) -- a block-CSR matrix a
    little below the total L3 streamed by 16 threads with a random gather is a DRAM rate, a dense matrix spread over all slices is not"""
    l3 = host_l3_reachable_bytes(threads)
    if l3 is None:
        return dict(residency=None, residency_note="L3 size not readable from sysfs; qw_host_GBs is whatever the host delivers at this size", matrix_MB=matrix_bytes / 1e6)
    fits = matrix_bytes <= 0.6 * l3      # room for the vectors and the other ways
    return dict(residency="L3" if fits else "DRAM", matrix_MB=matrix_bytes / 1e6, host_l3_reachable_MB=l3 / 1e6,
                residency_note=("the matrix is at most 0.6 x the L3 slices the %d bound threads reach: qw_host_GBs is a cache rate" if fits else
                                "the matrix exceeds 0.6 x the L3 slices the %d bound threads reach: qw_host_GBs is a memory rate") % threads)


def kkt_pair(tl, xmamd_mod, budget_s):
    """GPU and CPU wall-clock-to-KKT of the SAME complete solve (staircase + dual certificate, same stop rule) measured side by
    side in this run on this node, on a configuration the CPU oracle finishes in seconds: Dubrovnik-356-size dense Q (BASELINE config 3)."""
    from oracle import xm_oracle as xo
    wl = workload("dubrovnik356")
    Q = tl.gen_dense(wl["n"], seed=wl["seed"])["Q"]
    ctx = xmamd_mod.Context(Q=Q)
    ctx.solve(wl["max_rank"], wl["tol"], wl["lam"])
    t0 = time.perf_counter()
    R, s, gi = ctx.solve(wl["max_rank"], wl["tol"], wl["lam"])
    g_s = time.perf_counter() - t0
    ctx.close()
    t0 = time.perf_counter()
    Ro, so, io = xo.solve(Q, wl["max_rank"], wl["tol"], wl["lam"], max(budget_s, 60.0), trace=4000)
    c_s = time.perf_counter() - t0
    fo = float(io["trace"][-1, 0]) if "trace" in io and len(io["trace"]) else float(io.get("primal", float("nan")))
    return {"workload": wl["desc"] + ", complete staircase incl. dual certificate, max_rank %d, tol %g, lam %g" % (wl["max_rank"], wl["tol"], wl["lam"]),
            "gpu_wallclock_to_kkt_s": g_s, "cpu_wallclock_to_kkt_s": c_s, "cpu_threads": xo.num_threads(), "cpu_kind": "port (oracle/xm_oracle.c)",
            "gpu": {"rank": gi["rank"], "status": gi["status"], "primal": gi["primal"], "tcg_iters": gi["tcg_iters"], "min_eig": gi["min_eig"]},
            "cpu": {"rank": int(io["rank"]), "status": int(io["status"]), "primal": fo, "tcg_iters": int(io.get("tcg_iters", -1)),
                    "min_eig": float(io["cert"]["min_eig"]) if "cert" in io else None},
            "parity": {"rotations_rel_fro": tl.rotation_parity(R, s, Ro, so), "gram_rel_fro": tl.rel_fro(tl.gram(R, s), tl.gram(Ro, so)),
                       "f_rel": abs(gi["primal"] - fo) / max(abs(fo), 1e-300), "tolerance": 1e-6},
            "note": "same process, same node, same Q and options, same stop rule (|grad| < tol, then the certificate's acceptance rule, checkeig.h:349-368)"}


_JSON_FD = [1]


def _claim_stdout():
    """The contract is ONE JSON line on stdout.  Libraries loaded during the run write there too (RCCL's "Librccl path : ..." banner sits in
    libc's buffer until the process exits, i.e. AFTER the line): from here on file descriptor 1 is stderr for everybody, and the JSON line
    alone goes to the launcher's stdout through a private duplicate."""
    sys.stdout.flush()
    _JSON_FD[0] = os.dup(1)
    os.dup2(2, 1)


def _emit(obj):
    os.write(_JSON_FD[0], (json.dumps(obj) + "\n").encode())


def _error_line(ngp, args, workload_desc, msg):
    """the ONE JSON line of a run that could not produce its measurement (transport ladder ended in XM_ERR_COMM, a solve failed, a peer
    rank died, the watchdog fired): whoever launched this learns why instead of finding nothing"""
    _emit(dict({"metric": "BM iters/sec (tCG Hessian-vector iterations per second; ms_per_step = wall-clock-to-KKT of one staircase solve)",
                      "value": None, "unit": "tCG iters/s", "n_gpus": ngp, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
                      "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                      "config": {"workload": workload_desc}, "transport": None, "fallback": None, "error": msg}))


_PHASE = ["start"]     # where the run is: named in the watchdog's / the signal handler's error line


def _arm_guards(rank, ngp, args, workload_desc):
    """several ranks: (a) a watchdog -- a collective that never returns (a peer that died outside the library's bounded waits, a library
    initialisation that hangs) ends in the error line after XM_BENCH_TIMEOUT_S seconds (default 1800) instead of in silence; (b) rank 0
    prints the error line when the launcher tears the job down because another rank failed (SIGTERM)"""
    import signal, threading
    limit = float(os.environ.get("XM_BENCH_TIMEOUT_S", "1800"))

    def fire():
        if rank == 0:
            _error_line(ngp, args, workload_desc, "watchdog: no result after %.0f s (phase: %s)" % (limit, _PHASE[0]))
        os._exit(3)
    t = threading.Timer(limit, fire)
    t.daemon = True
    t.start()
    if rank == 0:
        def on_term(signum, frame):
            _error_line(ngp, args, workload_desc, "terminated by the launcher (signal %d) in phase: %s -- another rank failed" % (signum, _PHASE[0]))
            os._exit(4)
        signal.signal(signal.SIGTERM, on_term)
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="venice1778")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--storage", default="dense", choices=["dense", "bsr", "vg"],
                    help="storage of view-graph workloads: dense, 3x3-block CSR, or the edge list (XM_STORAGE_VIEWGRAPH: compressed sliced ELL)")
    ap.add_argument("--retraction", default="qr", choices=["qr", "polar"])
    ap.add_argument("--no-hbm-check", action="store_true", help="skip the 13.5 GB HBM-bound run of the same kernel")
    ap.add_argument("--no-rome", action="store_true", help="skip the Rome-scale (13682-camera view-graph) legs")
    ap.add_argument("--no-rome-dense", action="store_true", help="skip the dense-storage (13.5 GB) Rome-scale leg only")
    ap.add_argument("--no-kkt-pair", action="store_true", help="skip the same-node GPU / CPU wall-clock-to-KKT pair (Dubrovnik-356-size, seconds)")
    ap.add_argument("--cpu-kkt-seconds", type=float, default=240.0,
                    help="budget of the same-node CPU wall-clock-to-KKT leg on the HEADLINE workload: the oracle's complete staircase (certificate through LAPACK "
                         "dsyevd) on the host's threads, bounded by its own max_time; 0 = skip (cpu_baseline.wallclock_to_kkt_s = null + the reason)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "peer", "rccl"],
                    help="multi-GPU tCG exchange: auto = direct peer writes when the transport passes its self-test, else RCCL (the library's ladder); "
                         "peer = direct peer writes or an error; rccl = the RCCL all-gather north_star prescribes even where peer writes work. "
                         "With auto and --gpus N > 1 a second, RCCL-timed leg follows the default one (rccl_leg)")
    ap.add_argument("--no-rccl-leg", action="store_true", help="skip the second (RCCL) leg of a multi-GPU run")
    ap.add_argument("--sell", default="auto", choices=["auto", "on", "off"], help="sliced-ELL copy of block-sparse storage (xm_tuning_t.sell)")
    ap.add_argument("--model-recurrence", action="store_true", help="xm_options_t.flags |= XM_FLAG_MODEL_RECURRENCE: the tCG keeps no accumulated H v (default off: last bits of the model value differ)")
    ap.add_argument("--outer", default="auto", choices=["auto", "device", "host"],
                    help="outer iteration of the trust region: auto = the library's choice (on the device with block-CSR products, on the host with dense "
                         "ones: the measured faster form of each); device = xm_options_t.flags |= XM_FLAG_DEVICE_OUTER (decided on the GPU also with dense "
                         "products; the host only enqueues a repeating pair of launches); host = XM_FLAG_HOST_OUTER (the form of rounds 1-5)")
    ap.add_argument("--sym-min-rows", type=int, default=0, help="rows (3n) from which an exactly symmetric dense Q is multiplied by the half-traffic "
                    "kernel (xm_tuning_t.sym_min_rows; 0 = the library's measured default)")
    ap.add_argument("--stream-policy", type=int, default=None, help="measurement aid (include/xm_bench.h: xm_bench_dense_policy): cache policy of the matrix "
                    "streams forced -- 0 all cacheable, 1 all non-temporal, >= 2 a cacheable prefix of that many MB; default: the library's size rules")
    args = ap.parse_args()
    _claim_stdout()

    import torch
    import torch.distributed as dist
    bound_mask = _unbind_main_thread()   # an OpenMP runtime loaded by now may have bound this thread to its first place (see the top of the file)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("XM_BENCH_SINGLE_DEVICE") == "1":      # debugging aid: put every rank on device 0 of a 1-GPU box
        local = 0
    team = 1            # ranks driven by THIS process (single-process multi-GPU mode of the library)
    gpu_map = 0
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            team = args.gpus
            gpu_map = 1 if xmamd.device_count() < args.gpus else 0
        else:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    tkw = dict(n_gpus=team, gpu_map=gpu_map) if team > 1 else {}
    tn = {}
    if args.exchange != "auto":
        tn["exchange"] = {"peer": 2, "rccl": 3}[args.exchange]
    if args.sell != "auto":
        tn["sell"] = 1 if args.sell == "on" else -1
    if args.sym_min_rows:
        tn["sym_min_rows"] = args.sym_min_rows
    if tn:
        tkw["tuning"] = tn
    if args.exchange == "rccl" and world > 1:
        os.environ["XM_COMM_PEER"] = "0"      # one process per GPU: the process-level switch of xm_comm_init (include/xm_amd.h section 4)
    ngp = world * team   # GPUs (ranks) of the whole job
    retr = xmamd.RETRACT_POLAR if args.retraction == "polar" else xmamd.RETRACT_QR
    xmamd.require_gpu()
    torch.cuda.set_device(local)
    if args.stream_policy is not None:
        xmamd._chk(xmamd.lib().xm_bench_dense_policy(args.stream_policy))
    wl = workload(args.workload)
    guard = _arm_guards(rank, ngp, args, wl["desc"]) if ngp > 1 else None
    try:
        _init_ranks(rank, world, local, dist)
    except xmamd.XmError as e:
        if rank == 0:
            _error_line(ngp, args, wl["desc"], "communicator: " + str(e))
        raise SystemExit(1)
    import xm_testlib as tl
    t0 = time.time()
    Q = None
    try:
        return _run(args, wl, tl, t0, Q, tkw, team, gpu_map, ngp, retr, rank, world, torch, dist, bound_mask)
    except xmamd.XmError as e:
        # the library's transport ladder (direct peer writes -> RCCL) ended in XM_ERR_COMM, or a solve failed: still ONE JSON line
        if rank == 0:
            _error_line(ngp, args, wl["desc"], str(e))
        raise SystemExit(1)
    finally:
        if guard is not None:
            guard.cancel()


def _init_ranks(rank, world, local, dist, reinit=False):
    _PHASE[0] = "communicator"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not reinit:
            dist.init_process_group("gloo", rank=rank, world_size=world)     # control plane only (id exchange, barrier, max)
        uid = bytearray(128)
        if rank == 0 and os.environ.get("XM_BENCH_SHM") != "1" and os.environ.get("XM_BENCH_IPC") != "1":
            buf = (xmamd.C.c_char * 128)()
            xmamd._chk(xmamd.lib().xm_comm_unique_id(buf))
            uid = bytearray(buf.raw)
        box = [bytes(uid)]
        dist.broadcast_object_list(box, src=0)
        if os.environ.get("XM_BENCH_SHM") == "1":    # debugging aid: the library's shared-memory test transport instead of RCCL
            xmamd._chk(xmamd.lib().xm_comm_init_shm(rank, world, local, b"/xm_bench_%d" % int(os.environ.get("MASTER_PORT", "0")), 256 << 20))
        elif os.environ.get("XM_BENCH_IPC") == "1":  # the direct peer exchange between processes WITHOUT RCCL beside it (two ranks may share a GPU)
            xmamd._chk(xmamd.lib().xm_comm_init_ipc(rank, world, local, b"/xm_bench_ipc_%d" % int(os.environ.get("MASTER_PORT", "0")),
                                                    float(os.environ.get("XM_BENCH_IPC_SPIN", "0"))))   # bound of the device-side waits, s (0 = 20)
        else:
            # data plane inside the C++ solver: direct peer writes through IPC-mapped buffers when every rank can map every peer and
            # the transport's self-test passes (ranks of one node), else RCCL all-gathers over xGMI (XM_COMM_PEER=0 forces RCCL)
            xmamd._chk(xmamd.lib().xm_comm_init(rank, world, local, box[0], None))


def _rccl_leg(args, wl, tl, Q, tkw, team, world, rank, retr, torch, dist, barrier, out):
    """the headline workload again through the RCCL rung of the transport ladder.  The headline is already measured: whatever happens here
    -- a refusal, an exception, a collective that never returns -- must not cost it.  A timer prints the line as it stands (with the leg
    marked) and ends the process with status 0 if the leg has not finished after XM_BENCH_RCCL_LEG_S seconds (default 240; every rank
    runs the same timer, so the job ends together)."""
    import threading
    _PHASE[0] = "RCCL leg"
    limit = float(os.environ.get("XM_BENCH_RCCL_LEG_S", "240"))

    def give_up():
        if rank == 0:
            out["rccl_leg"] = {"error": "no result after %.0f s; the lines above it stand" % limit}
            _emit(out)
        os._exit(0)
    timer = threading.Timer(limit, give_up)
    timer.daemon = True
    timer.start()
    try:
        return _rccl_leg_body(args, wl, tl, Q, tkw, team, world, rank, retr, torch, dist, barrier)
    except Exception as e:      # not only XmError: nothing raised here may lose the line
        return {"refused": "exchange: rccl -> refused (%s: %s)" % (type(e).__name__, str(e)[:400]), "communicator_world": world * team}
    finally:
        timer.cancel()


def _rccl_leg_body(args, wl, tl, Q, tkw, team, world, rank, retr, torch, dist, barrier):
    try:
        if team > 1:
            kw = dict(tkw); kw["tuning"] = dict(kw.get("tuning") or {}, exchange=3)
        else:                      # one process per GPU: a new process-level communicator without the IPC upgrade
            xmamd._chk(xmamd.lib().xm_comm_finalize())
            os.environ["XM_COMM_PEER"] = "0"
            _init_ranks(rank, world, int(os.environ.get("LOCAL_RANK", "0")), dist, reinit=True)
            kw = dict(tkw)
        if wl["kind"] != "dense":
            return {"skipped": "the RCCL leg is timed on the dense headline workload"}
        ctx = xmamd.Context(Q=Q, **kw)
        kind, name, note = ctx.transport()
        ctx.solve(wl["max_rank"], wl["tol"], wl["lam"])
        barrier()
        t0 = time.perf_counter()
        infos = [ctx.solve(wl["max_rank"], wl["tol"], wl["lam"], retraction=retr, grouping=i % 3)[2] for i in range(args.steps)]
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        barrier()
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t[0])
        ctx.close()
        return {"value": sum(i["tcg_iters"] for i in infos) / el, "unit": "tCG iters/s", "steps": args.steps, "ms_per_step": el / args.steps * 1e3,
                "transport": name, "transport_kind": kind, "exchange_used": infos[-1].get("exchange"), "communicator_world": world * team, "note": note or None}
    except xmamd.XmError as e:
        return {"refused": "exchange: rccl -> refused (%s)" % str(e)[:400], "communicator_world": world * team}


def _run(args, wl, tl, t0, Q, tkw, team, gpu_map, ngp, retr, rank, world, torch, dist, bound_mask):
    _PHASE[0] = "context set-up (Q upload, transport ladder of the single-process mode)"
    if wl["kind"] == "dense":
        Q = tl.gen_dense(wl["n"], seed=wl["seed"])["Q"]
        ctx = xmamd.Context(Q=Q, **tkw)
        storage_desc = "dense 3n x 3n f64 (%.1f MB)" % (72.0 * wl["n"] ** 2 / 1e6)
    else:
        P = tl.gen_vg(wl["n"], deg=wl["deg"], sigma=wl["sigma"], seed=wl["seed"], dense=False)
        nb = int(P["colidx"].size)
        if args.storage == "dense":
            ctx = xmamd.Context(bsr=(P["rowptr"], P["colidx"], P["blocks"]), densify=True, **tkw)   # every rank expands its own rows
            storage_desc = "dense 3n x 3n f64 built on device from %d blocks (%.1f MB)" % (nb, 72.0 * wl["n"] ** 2 / 1e6)
        elif args.storage == "vg":
            e = P["edges"]
            ctx = xmamd.Context(vg=(e[:, 0], e[:, 1], P["w"], P["M"]), n=wl["n"], **tkw)
            storage_desc = ("view-graph edge list, %d edges = %d stored blocks (%.1f MB as 3x3-block CSR; the products stream the "
                            "quaternion-compressed sliced-ELL copy, %.1f MB)" % (e.shape[0], nb, 76.0 * nb / 1e6, 36.0 * (nb - wl["n"]) / 1e6))
        else:
            ctx = xmamd.Context(bsr=(P["rowptr"], P["colidx"], P["blocks"]), **tkw)
            storage_desc = "3x3-block CSR, %d blocks (%.1f MB)" % (nb, 76.0 * nb / 1e6)
    gen_s = time.time() - t0
    tr_kind, tr_name, tr_note = ctx.transport()   # which transport joins the ranks, and why a faster one was given up (library's ladder)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    mflag = ((xmamd.FLAG_MODEL_RECURRENCE if args.model_recurrence else 0) | (xmamd.FLAG_HOST_OUTER if args.outer == "host" else 0) |
             (xmamd.FLAG_DEVICE_OUTER if args.outer == "device" else 0))

    def one_solve(flags=0, grouping=0):
        flags |= mflag
        return ctx.solve(wl["max_rank"], wl["tol"], wl["lam"], flags=flags, retraction=retr, grouping=grouping)

    _PHASE[0] = "warm-up solves"
    for i in range(args.warmup):
        one_solve(grouping=i % 3)
    barrier()
    _PHASE[0] = "timed solves"
    t0 = time.perf_counter()
    infos = []
    last_sol = None
    t_last = 0.0
    for i in range(args.steps):
        t1 = time.perf_counter()
        last_sol = one_solve(flags=xmamd.FLAG_PROFILE_QW, grouping=i % 3)
        t_last = time.perf_counter() - t1
        infos.append(last_sol[2])
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    _PHASE[0] = "after the timed solves (secondary legs)"
    barrier()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])

    iters = sum(i["tcg_iters"] for i in infos)
    last = infos[-1]
    qw_ms = sum(i["qw_ms_sum"] for i in infos) / max(1, sum(i["qw_ms_count"] for i in infos))
    # per-rank algorithmic bytes of one tCG product: this rank's rows of Q + W in/out (SURVEY §8d dense formula / world)
    n = wl["n"]
    o_fin = max(3, last["rank"])
    # outer iteration on the device: the tCG products run the role-switching instantiation (EPI_AUTO: Hessian epilogue, or -- one launch per outer
    # iteration -- the candidate's gradient epilogue; the same bytes either way)
    epi_name = "AUTO" if last.get("outer_on_device") else "HESS"
    if wl["kind"] == "dense" or args.storage == "dense":
        alg_bytes = (8.0 * (3 * n) ** 2) / ngp + 2 * 8 * 3 * n * o_fin
        kname = "qw_dense_kernel<o, EPI_%s>" % epi_name
    else:
        alg_bytes = (76.0 * nb + 4 * (n + 1)) / ngp + 2 * 8 * 3 * n * o_fin      # FULL-storage accounting whatever is streamed
        pk = ctx.product_kind(o_fin)
        kname = "qw_bsr3_kernel<o, EPI_%s>" % epi_name
        if pk in ("sell", "sell_quat"):
            kname = "qw_sell_kernel<o> + sell_reduce_kernel<o, EPI_HESS> (sliced-ELL over per-XCD column slabs; one product = both launches)"
            if team == 1 and world == 1 and ctx.sell_wpad():
                kname += "; the gather reads the copy of W that tcg_init / cg_step keep at a 128-byte record pitch (xm_tuning_t.sell_wpad, automatic)"
            if pk == "sell_quat":
                kname += ", view-graph codec: %.1f MB streamed per product for %.1f MB of full storage" % (last["qw_stream_bytes"] / 1e6, 76.0 * nb / ngp / 1e6)
    achieved = alg_bytes / (qw_ms * 1e-3) / 1e9 if qw_ms > 0 else 0.0
    # HBM-side bytes per launch of the dominant kernel: rocprofv3 --pmc FETCH_SIZE (own pass, kernel-trace only), corrected as
    # MI355X_MICROARCH.md prescribes (KB -> bytes, x2 for the gfx950 wide-load half count); see profiles/r01_pmc_*.json
    leg = {"venice1778": "venice", "vg100k": "vg100k_" + args.storage}.get(args.workload, args.workload)
    traffic, traffic_source, traced_us = recorded_traffic(leg, ngp)
    out = {
        "metric": "BM iters/sec (tCG Hessian-vector iterations per second; ms_per_step = wall-clock-to-KKT of one staircase solve)",
        "value": iters / elapsed, "unit": "tCG iters/s", "n_gpus": ngp, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        # the path-independent figures next to the wall clock (which also carries the iteration lottery of the three summation groupings)
        "us_per_tcg_iteration_all_in": elapsed / max(1, iters) * 1e6,
        "tcg_iters_per_step": iters / args.steps, "outer_iters_per_step": sum(i["outer_iters"] for i in infos) / args.steps,
        "build": library_build_stamp(),
        "config": {"workload": wl["desc"], "n_cameras": n, "storage": storage_desc,
                   "max_rank": wl["max_rank"], "tol": wl["tol"], "lam": wl["lam"],
                   "retraction": args.retraction,
                   "outer_iteration": ("on the device (outer_step_kernel: retraction, accept / reject, radius, stop tests and the next tCG's start decided by the GPU; "
                                       "the host enqueues one repeating (product, step) pair of launches ahead)" if last.get("outer_on_device")
                                      else "on the host (the library's default with dense / sliced-ELL / matrix-free products and with several ranks: the host "
                                           "notices the end of a truncated CG and confirms a speculatively started next one; XM_FLAG_DEVICE_OUTER / --outer device "
                                           "moves it to the GPU for dense products, measured 1.5-2 % slower there)"),
                   **({"model_value": "from the CG recurrences (XM_FLAG_MODEL_RECURRENCE)"} if args.model_recurrence else {}),
                   "summation_groupings": "step i uses xm_options_t.sum_grouping = i mod 3; tcg_iters_by_step lists what each drew",
                   "parallelism": ("single GPU" if ngp == 1 else
                                   (f"camera row partition x{world}, one process per GPU, " +
                                    ("direct peer-write exchange through IPC-mapped buffers fused into the tCG" if last.get("exchange") == 2
                                     else "all-gather per exchange (RCCL; shm test transport on a shared device)")) if team == 1 else
                                   f"camera row partition x{team} inside ONE process (xm_problem_t.n_gpus), direct peer-write exchange fused into the tCG"),
                   **({"transport": "shared-memory TEST transport, all ranks on one GPU (functional dry run, not a scaling measurement)"}
                      if os.environ.get("XM_BENCH_SHM") == "1" else {}),
                   **({"transport": "direct peer exchange between PROCESSES (xm_comm_init_ipc), all ranks on one GPU (functional dry run, not a scaling measurement)"}
                      if os.environ.get("XM_BENCH_IPC") == "1" and os.environ.get("XM_BENCH_SINGLE_DEVICE") == "1" else {}),
                   **({"devices": "%d VIRTUAL devices on one GPU (gpu_map = 1): functional run of the %d-rank flow, not a scaling measurement" % (team, team)}
                      if gpu_map == 1 else {})},
        "transport": tr_name, "fallback": (tr_note or None),
        "exchange": {"requested": args.exchange, "used": {0: "none (one GPU)", 1: "all-gather between the launches (RCCL, or the test transports)",
                                                           2: "direct peer writes fused into cg_step"}.get(last.get("exchange"), "?"),
                     "communicator_world": ngp if ngp > 1 else None},
        "solve": {"rank": last["rank"], "status": last["status"], "primal": last["primal"], "dual": last["dual"],
                  "min_eig": last["min_eig"], "tcg_iters_per_solve": last["tcg_iters"], "outer_iters": last["outer_iters"],
                  "qw_products": last["qw_products"], "lanczos_iters": last["lanczos_iters"],
                  "tr_seconds": last["tr_seconds"], "cert_seconds": last["cert_seconds"], "setup_gen_s": gen_s,
                  "tcg_iters_by_step": [i["tcg_iters"] for i in infos], "exchange": last.get("exchange")},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_kind": ("recorded" if traffic is not None else None),   # PMC passes are separate runs (gpurun / the guide): never measured inside this one
                     "traffic_source": traffic_source, "traced_avg_launch_us": traced_us, "kernel": kname + ("; Q is exactly symmetric and has >= 4096 rows: the rank-3 / rank-4 stages multiply it through the half-traffic pair "
                                                "qw_symv_kernel + symv_reduce_kernel<o, EPI_%s> (one product = both launches, upper triangle streamed once), rank 5 through " % epi_name +
                                                "qw_dense_kernel; achieved / frac count the FULL-storage bytes of SURVEY 8d per product, traffic is what the counters saw"
                                                if last.get("sym_product") else ""), "avg_launch_ms": qw_ms,
                     "algorithmic_bytes_per_launch": alg_bytes,
                     # what really moved, next to the contract's algorithmic figure: counter bytes per product / traced duration per product
                     "streamed": ({"bytes_per_product": traffic, "GBs": traffic / (traced_us * 1e-6) / 1e9, "frac": traffic / (traced_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                   "traced_frac_algorithmic": alg_bytes / (traced_us * 1e-6) / 1e9 / HBM_PEAK_GBS} if (traffic and traced_us) else None),
                     "note": "HIP events around every 64th tCG Q*W launch inside the timed solves (no-op samples dropped); " + (
                             "per-rank Q is %.0f MB, inside the 256 MB Infinity Cache: the figure is cache-assisted, see roofline_hbm for the "
                             "HBM-bound run of the same kernel" % (alg_bytes / 1e6) if alg_bytes < 250e6 else
                             "per-rank Q is %.0f MB, beyond the 256 MB Infinity Cache: HBM-bound" % (alg_bytes / 1e6))},
    }
    ctx.close()
    if not args.no_rome and args.workload == "venice1778":
        # BASELINE.json north_star: "end-to-end solve of a Rome-scale (>= 10k-camera) Q reported as iters/s and wall-clock at
        # 1, 2, 4 and 8 GPUs" — a secondary leg at every N (same rules: warmup, barrier-bracketed, max over ranks); the headline
        # `value` above stays the Venice-1778 metric.
        wr = workload("final13682")
        Pr = tl.gen_vg(wr["n"], deg=wr["deg"], sigma=wr["sigma"], seed=wr["seed"], dense=False)
        cr = xmamd.Context(bsr=(Pr["rowptr"], Pr["colidx"], Pr["blocks"]), **tkw)
        rome_kind = cr.product_kind(3)
        cr.solve(wr["max_rank"], wr["tol"], wr["lam"], flags=mflag)
        barrier()
        t0 = time.perf_counter()
        ri = [cr.solve(wr["max_rank"], wr["tol"], wr["lam"], flags=xmamd.FLAG_PROFILE_QW | mflag)[2] for _ in range(2)]
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        barrier()
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t[0])
        cr.close()
        nbr = int(Pr["colidx"].size)
        rq = sum(i["qw_ms_sum"] for i in ri) / max(1, sum(i["qw_ms_count"] for i in ri))
        rb = (76.0 * nbr + 4 * (wr["n"] + 1)) / ngp + 2 * 8 * 3 * wr["n"] * max(3, ri[-1]["rank"])
        out["rome_scale"] = {"workload": wr["desc"] + ", 3x3-block CSR, %d blocks (%.1f MB)" % (nbr, 76.0 * nbr / 1e6), "n_gpus": ngp,
                             "value": sum(i["tcg_iters"] for i in ri) / el, "unit": "tCG iters/s", "steps": 2, "warmup": 1,
                             "ms_per_step": el / 2 * 1e3, "rank": ri[-1]["rank"], "status": ri[-1]["status"],
                             "tcg_iters_per_solve": ri[-1]["tcg_iters"], "primal": ri[-1]["primal"],
                             "hess_launch_ms": rq, "hess_algorithmic_GBs": rb / (rq * 1e-3) / 1e9 if rq > 0 else None,
                             "product_kernel": rome_kind, "us_per_tcg_iteration_all_in": el / max(1, sum(i["tcg_iters"] for i in ri)) * 1e6,
                             "tr_seconds": ri[-1]["tr_seconds"], "cert_seconds": ri[-1]["cert_seconds"], "outer_iters": ri[-1]["outer_iters"],
                             "lanczos_iters": ri[-1]["lanczos_iters"],
                             **dict(zip(("hess_traffic", "hess_traffic_source", "hess_traced_us"), recorded_traffic("rome_bsr", ngp))),
                             "regime": "32 MB of blocks: cache-resident, launch-latency regime -- hess_algorithmic_GBs is not an HBM roofline fraction"}
        if not args.no_rome_dense:
            # the same Q in the reference's own storage (dense 3n x 3n f64, 13.5 GB; every rank expands its camera rows on its
            # GPU): the HBM-bound regime where the row partition pays.  ONE timed solve, no warmup (about 700 products of 2 ms each).
            cd = xmamd.Context(bsr=(Pr["rowptr"], Pr["colidx"], Pr["blocks"]), densify=True, **tkw)
            cd.solve(wr["max_rank"], wr["tol"], wr["lam"], max_time=-1.0)   # warm-up: every rank level stops at its first time check (clocks, first-touch of the workspaces)
            barrier()
            t0 = time.perf_counter()
            di = cd.solve(wr["max_rank"], wr["tol"], wr["lam"], flags=xmamd.FLAG_PROFILE_QW | mflag)[2]
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            barrier()
            if world > 1:
                t = torch.tensor([el], dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t[0])
            cd.close()
            dq_ms = di["qw_ms_sum"] / max(1, di["qw_ms_count"])
            db = 8.0 * (3 * wr["n"]) ** 2 / ngp + 2 * 8 * 3 * wr["n"] * max(3, di["rank"])
            out["rome_scale_dense"] = {"workload": wr["desc"] + ", dense 3n x 3n f64 (%.1f GB over %d GPU%s)" % (72.0 * wr["n"] ** 2 / 1e9, ngp, "" if ngp == 1 else "s"),
                                       "n_gpus": ngp, "value": di["tcg_iters"] / el, "unit": "tCG iters/s", "steps": 1, "warmup": "one outer iteration",
                                       "ms_per_step": el * 1e3, "rank": di["rank"], "status": di["status"], "tcg_iters_per_solve": di["tcg_iters"],
                                       "primal": di["primal"], "sym_product": di.get("sym_product"), "hess_launch_ms": dq_ms,
                                       **dict(zip(("hess_traffic", "hess_traffic_source", "hess_traced_us"), recorded_traffic("rome_dense", ngp))),
                                       "hess_algorithmic_GBs": db / (dq_ms * 1e-3) / 1e9 if dq_ms > 0 else None}
    if rank == 0 and ngp == 1 and not args.no_hbm_check and wl["kind"] == "dense":
        # same kernel, matrix far beyond every cache: 13682 cameras = 13.5 GB of random f64 generated on the device
        nb_, o_ = 13682, 3
        ld_ = xmamd.dense_ld(nb_)
        g = torch.Generator(device="cuda"); g.manual_seed(1)
        Qbig = torch.randn(3 * nb_ * ld_, dtype=torch.float64, device="cuda", generator=g)
        Wbig = torch.randn(ld_ * 3, dtype=torch.float64, device="cuda", generator=g)
        Obig = torch.zeros(3 * nb_ * 3, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        ms = xmamd.C.c_double()
        xmamd._chk(xmamd.lib().xm_qw_dense_time(Qbig.data_ptr(), nb_, o_, Wbig.data_ptr(), Obig.data_ptr(), 20, xmamd.C.byref(ms)))
        by = 8.0 * (3 * nb_) ** 2 + 2 * 8 * 3 * nb_ * o_
        htr, hsrc, hus = recorded_traffic("hbm13682", 1)
        out["roofline_hbm"] = {"bound": "hbm", "achieved": by / (ms.value * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": by / (ms.value * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_launch_ms": ms.value,
                               "traffic": htr, "traffic_source": hsrc, "traced_avg_launch_us": hus,
                               "algorithmic_bytes_per_launch": by, "kernel": "qw_dense_kernel<3, EPI_PLAIN>",
                               "workload": "same kernel on a 13682-camera (Final-13682-size) 13.5 GB random matrix, 20 launches"}
        del Qbig, Wbig, Obig
    if rank == 0 and ngp == 1 and args.cpu_seconds > 0 and bound_mask is not None:
        os.sched_setaffinity(0, bound_mask)   # the host leg: this thread is OpenMP thread 0 again, on its place
    if rank == 0 and ngp == 1 and args.cpu_seconds > 0 and not args.no_kkt_pair and args.workload == "venice1778":
        out["kkt_pair"] = kkt_pair(tl, xmamd, args.cpu_seconds)
    if rank == 0 and ngp == 1 and args.cpu_seconds > 0 and Q is not None:
        out["cpu_baseline"] = cpu_baseline(Q, wl, args.cpu_seconds)
    elif rank == 0 and ngp == 1 and args.cpu_seconds > 0 and wl["kind"] == "vg" and args.storage in ("bsr", "vg"):
        out["cpu_baseline"] = cpu_baseline(None, wl, args.cpu_seconds, bsr=(P["rowptr"], P["colidx"], P["blocks"]))
    if rank == 0 and ngp == 1 and "cpu_baseline" in out and Q is not None and last_sol is not None and args.workload == "venice1778":
        if args.cpu_kkt_seconds > 0:
            gi = dict(last); gi["seconds_last_timed_solve"] = t_last
            try:
                out["cpu_baseline"].update(cpu_kkt(Q, wl, args.cpu_kkt_seconds, tl, last_sol, gi))
            except Exception as e:     # the headline is never lost to the baseline leg
                out["cpu_baseline"].update({"wallclock_to_kkt_s": None, "wallclock_to_kkt_note": "CPU wall-clock-to-KKT leg failed: %r" % (e,)})
        else:
            out["cpu_baseline"].update({"wallclock_to_kkt_s": None, "wallclock_to_kkt_note": "skipped: --cpu-kkt-seconds 0"})
    if rank == 0 and "cpu_baseline" in out and last_sol is not None:
        par = parity_vs_recorded_oracle(args.workload, last_sol[0], last_sol[1], last["primal"])
        if par is not None:
            out["cpu_baseline"]["parity"] = {k: par[k] for k in ("rotations_rel_fro", "gram_sample_rel_fro", "f_rel", "tolerance", "against")}
            out["cpu_baseline"]["recorded_wallclock_to_kkt_s"] = par["cpu_wallclock_to_kkt_s"]       # ANOTHER machine and thread count than `value`:
            out["cpu_baseline"]["recorded_wallclock_to_kkt_threads"] = par["cpu_threads"]            # see kkt_pair for a same-node pair
            out["cpu_baseline"]["recorded_wallclock_to_kkt_provenance"] = par["cpu_wallclock_provenance"]
    if ngp > 1 and args.exchange == "auto" and not args.no_rccl_leg:
        # north_star prescribes "an RCCL all-gather of Y over xGMI each iteration"; the library's default is the direct peer exchange.  One
        # driver run yields both curves: the same workload again with the RCCL rung of the ladder forced (xm_tuning_t.exchange = 3 in the
        # single-process mode; XM_COMM_PEER=0 + a fresh xm_comm_init with one process per GPU).  A refusal (RCCL does not put two ranks
        # on one device: the virtual-device dry run) is reported, not raised.
        # LAST of the communicating legs: in the one-process-per-GPU launch it replaces the process-level communicator.
        if os.environ.get("XM_BENCH_SHM") == "1" or os.environ.get("XM_BENCH_IPC") == "1":
            out["rccl_leg"] = {"skipped": "debugging transport selected by XM_BENCH_SHM / XM_BENCH_IPC: no RCCL communicator in this run"}
        else:
            out["rccl_leg"] = _rccl_leg(args, wl, tl, Q, tkw, team, world, rank, retr, torch, dist, barrier, out)
    if world > 1:
        xmamd.lib().xm_comm_finalize()
        # Replica throughput: what N GPUs deliver on N INDEPENDENT Venice-size scenes (no data-path communication; the row
        # partition above is latency-bound at this size: 28 MB of Q per GPU at N = 8 against two collectives per iteration).
        # Reported next to the partitioned headline, never instead of it.
        cx = xmamd.Context(Q=Q) if Q is not None else None
        if cx is not None:
            cx.solve(wl["max_rank"], wl["tol"], wl["lam"])
            barrier()
            t0 = time.perf_counter()
            ri = [cx.solve(wl["max_rank"], wl["tol"], wl["lam"])[2] for _ in range(max(1, args.steps))]
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            barrier()
            t = torch.tensor([el, float(sum(i["tcg_iters"] for i in ri))], dtype=torch.float64)
            tm = t.clone(); dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            ts = t.clone(); dist.all_reduce(ts, op=dist.ReduceOp.SUM)
            cx.close()
            out["replicas"] = {"what": "%d independent solves of the same workload, one per GPU, no communication" % world,
                               "value": float(ts[1]) / float(tm[0]), "unit": "tCG iters/s", "scaling": "weak", "n_gpus": world,
                               "steps": max(1, args.steps), "ms_per_step": float(tm[0]) / max(1, args.steps) * 1e3}
        dist.destroy_process_group()
    if rank == 0:
        _emit(out)


if __name__ == "__main__":
    main()
