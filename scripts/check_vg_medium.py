"""GPU solves of the medium view-graph problems whose CPU-oracle results are recorded in
tests/golden/synth/recorded_oracle_medium.json.  The oracle side was produced (hours of CPU at n = 2000) by

    for n, deg, lam in [(600, 30, 30.0), (600, 30, 3.0), (2000, 30, 30.0)]:
        Q = tl.gen_vg(n, deg=deg, sigma=0.05, seed=n)["Q"]
        R, s, info = oracle.xm_oracle.solve(Q, 5, 1e-6, lam, 1000, trace=4000)      # rank, status, trace[-1, 0], tcg_iters, cert
"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xm-code_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, xmamd, xm_testlib as tl
ORACLE = {(600, 30, 30.0): None, (600, 30, 3.0): dict(rank=3, status=1, f=118.28699539809459, tcg=3500, outer=630),
          (2000, 30, 30.0): dict(rank=4, status=1, f=419.7874863274935, tcg=4441, outer=522)}
for (n, deg, lam), exp in ORACLE.items():
    P = tl.gen_vg(n, deg=deg, sigma=0.05, seed=n)
    for kind in ("dense", "bsr"):
        ctx = xmamd.Context(Q=P["Q"]) if kind == "dense" else xmamd.Context(bsr=(P["rowptr"], P["colidx"], P["blocks"]))
        t = time.time(); R, s, i = ctx.solve(5, 1e-6, lam); el = time.time() - t
        ctx.close()
        print(n, deg, lam, kind, "rank", i["rank"], "status", i["status"], "tcg", i["tcg_iters"], "outer", i["outer_iters"],
              "f %.12g" % i["primal"], "min_eig %.3g" % i["min_eig"], "%.2fs" % el, "| oracle:", exp, flush=True)
