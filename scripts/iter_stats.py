"""Iteration-count statistics of staircase solves, GPU vs CPU oracle, over a family of seeded instances that need rank escalation
(saddle points at the lower ranks).  Question: is the GPU's iteration excess on the Venice-1778 workload
(3760-4924 tCG iterations against the oracle's single recorded run of 3061) a systematic bias of the GPU implementation
(beta from the expanded residual norm, Lanczos escape vector) or the spread of a path that passes through saddle points?
    python scripts/iter_stats.py oracle|gpu out.json
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "xm-code_amd"))
import numpy as np
import xm_testlib as tl

CASES = [("vg", 40, 3, 1.5, s, 6, 1e-9, 3.0) for s in range(1, 13)] + [("vg", 100, 3, 1.5, s, 6, 1e-9, 3.0) for s in range(1, 7)] + \
        [("dense", 149, 0, 0, s, 5, 1e-6, 0.0) for s in (149, 150, 151, 152)]
who = sys.argv[1]
out = []
for kind, n, deg, sig, seed, mr, tol, lam in CASES:
    Q = tl.gen_vg(n, deg=deg, sigma=sig, seed=seed)["Q"] if kind == "vg" else tl.gen_dense(n, seed=seed)["Q"]
    t0 = time.time()
    if who == "oracle":
        from oracle import xm_oracle as xo
        R, s, info = xo.solve(Q, mr, tol, lam, 1000.0, trace=8000)
        rec = dict(rank=int(info["rank"]), status=int(info["status"]), tcg=int(info["tcg_iters"]), f=float(info["trace"][-1, 0]))
    else:
        import xmamd
        R, s, info = xmamd.solve_dense(Q, mr, tol, lam, trace=8000)
        rec = dict(rank=info["rank"], status=info["status"], tcg=int(info["tcg_iters"]), f=float(info["primal"]), outer=int(info["outer_iters"]))
    rec.update(kind=kind, n=n, seed=seed, seconds=time.time() - t0)
    out.append(rec)
    print(rec, flush=True)
json.dump(out, open(sys.argv[2], "w"), indent=1)
