#!/usr/bin/env python3
"""Regenerates the recorded CPU-oracle fixtures under tests/golden/synth/ that are too expensive for the test run:

    venice1778   venice1778_oracle.json + venice1778_oracle_rot.npy      the headline workload of bench.py, complete staircase
                 (3 -> 4 -> 5, three O(n^3) certificates): ~3 HOURS on 8 cores
    rome13682    rome13682_oracle.json + rome13682_oracle_rot.npy        rank-3 trust region of the 13682-camera view-graph Q, dense
                 on the host: needs ~14 GB of RAM, ~8 minutes on 8 cores
    vg100k       vg100k_oracle.json + vg100k_oracle_rot_every8.npy       rank-3 trust region of the 100k-camera view-graph Q through
                 the oracle's test-only block-CSR product: ~2 minutes on 8 cores
    medium       recorded_oracle_medium.json                             600- / 2000-camera view graphs in the hard regime: hours at n = 2000
    mid700       mid700_oracle.json + mid700_oracle_rot.npy              700-camera dense instance that needs rank 4 (two O(n^3) certificates
                 of a 2100 x 2100 matrix): ~5 minutes

    python scripts/record_oracle_large.py vg100k [rome13682 venice1778 medium] [--out DIR]

CPU only (the oracle is test infrastructure; nothing here touches the GPU library).  The inputs are the seeded generators of
tests/xm_testlib.py, the same calls the GPU tests make (tests/test_gpu_parity.py::test_*_vs_recorded_oracle), so a regenerated
fixture is comparable bit for bit with the committed one up to the thread count of the OpenMP sums."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import xm_testlib as tl
from oracle import xm_oracle as xo


def venice1778(out):
    n = 1778
    Q = tl.gen_dense(n, seed=n)["Q"]
    t0 = time.time()
    R, s, info = xo.solve(Q, 5, 1e-6, 0.0, 1e9, trace=4000)
    el = time.time() - t0
    rot, _ = tl.recover_rotations(R, s)
    np.save(os.path.join(out, "venice1778_oracle_rot.npy"), rot)
    c = info["cert"]
    json.dump(dict(n=n, seed=n, max_rank=5, tol=1e-6, lam=0.0, rank=int(info["rank"]), status=int(info["status"]),
                   f=float(info["trace"][-1, 0]), tcg=int(info["tcg_iters"]), outer=int(info["outer_iters"]), min_eig=c["min_eig"],
                   gap=c["gap"], seconds=el, threads=xo.num_threads(), s_min=float(s.min()), s_max=float(s.max())),
              open(os.path.join(out, "venice1778_oracle.json"), "w"), indent=1)


def mid700(out):
    n = 700
    Q = tl.gen_dense(n, seed=n)["Q"]
    t0 = time.time()
    R, s, info = xo.solve(Q, 5, 1e-9, 0.0, 1e9, trace=4000)
    el = time.time() - t0
    rot, _ = tl.recover_rotations(R, s)
    np.save(os.path.join(out, "mid700_oracle_rot.npy"), rot)
    sR = tl.scale_rows(R, s)
    idx = tl.gram_sample_index(sR.shape[0])
    np.save(os.path.join(out, "mid700_oracle_gram_sample.npy"), (sR[idx[:, 0]] * sR[idx[:, 1]]).sum(axis=1))
    c = info["cert"]
    json.dump(dict(n=n, seed=n, max_rank=5, tol=1e-9, lam=0.0, rank=int(info["rank"]), status=int(info["status"]),
                   f=float(info["trace"][-1, 0]), tcg=int(info["tcg_iters"]), outer=int(info["outer_iters"]), min_eig=c["min_eig"],
                   gap=c["gap"], seconds=el, threads=xo.num_threads()),
              open(os.path.join(out, "mid700_oracle.json"), "w"), indent=1)


def _vg_tr(out, name, n, deg, lam, sub, bsr):
    P = tl.gen_vg(n, deg=deg, sigma=0.05, seed=n, dense=not bsr)
    R0 = np.tile(np.eye(3), (n, 1))
    t0 = time.time()
    if bsr:
        R, s, primal, _, st = xo.trustregion_bsr(P["rowptr"], P["colidx"], P["blocks"], R0, np.ones(n), lam=lam, gradtol=1e-6, maxtime=1e9)
    else:
        R, s, primal, _, st = xo.trustregion(P["Q"], R0, np.ones(n), lam=lam, gradtol=1e-6, maxtime=1e9)
    el = time.time() - t0
    rot, _ = tl.recover_rotations(R, s)
    if sub > 1:
        np.save(os.path.join(out, f"{name}_oracle_rot_every{sub}.npy"), rot.reshape(3, n, 3)[:, ::sub, :])
    else:
        np.save(os.path.join(out, f"{name}_oracle_rot.npy"), rot)
    json.dump(dict(n=n, deg=deg, sigma=0.05, lam=lam, tol=1e-6, f=float(primal), tcg=int(st["tcg_iters"]), outer=int(st["outer_iters"]),
                   stop_reason=int(st["stop_reason"]), seconds=el, threads=xo.num_threads(), s_min=float(s.min()), s_max=float(s.max()),
                   qw_ms=st["qw_seconds"] / max(st["qw_products"], 1) * 1e3),
              open(os.path.join(out, f"{name}_oracle.json"), "w"), indent=1)


def medium(out):
    cases = []
    for n, deg, lam in [(600, 30, 30.0), (600, 30, 3.0), (2000, 30, 30.0)]:
        Q = tl.gen_vg(n, deg=deg, sigma=0.05, seed=n)["Q"]
        t0 = time.time()
        R, s, info = xo.solve(Q, 5, 1e-6, lam, 1e9, trace=4000)
        cases.append(dict(n=n, deg=deg, lam=lam, rank=int(info["rank"]), status=int(info["status"]), f=float(info["trace"][-1, 0]),
                          tcg=int(info["tcg_iters"]), outer=int(info["outer_iters"]), seconds=time.time() - t0))
        print(cases[-1], flush=True)
    json.dump(dict(note="CPU oracle (oracle/xm_oracle.c, xmo_solve) on tl.gen_vg(n, deg, sigma=0.05, seed=n), max_rank 5, tol 1e-6; "
                        "generated by scripts/record_oracle_large.py medium", cases=cases, threads=xo.num_threads()),
              open(os.path.join(out, "recorded_oracle_medium.json"), "w"), indent=1)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="+", choices=["venice1778", "rome13682", "vg100k", "medium", "mid700"])
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "synth"))
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    for w in a.which:
        print(f"recording {w} with {xo.num_threads()} threads ...", flush=True)
        t0 = time.time()
        if w == "venice1778":
            print("  (about three hours on 8 cores: three dense certificates of a 5334 x 5334 matrix)", flush=True)
            venice1778(a.out)
        elif w == "rome13682":
            _vg_tr(a.out, "rome13682", 13682, 30, 1000.0, 1, bsr=False)
        elif w == "mid700":
            mid700(a.out)
        elif w == "vg100k":
            _vg_tr(a.out, "vg100k", 100000, 50, 1000.0, 8, bsr=True)
        else:
            medium(a.out)
        print(f"  {w}: {time.time() - t0:.0f} s", flush=True)
