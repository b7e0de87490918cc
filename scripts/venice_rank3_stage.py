"""rank-3 stage of the Venice-1778 workload on the GPU (max_rank = 3: one trust-region run from the identity stack + its certificate),
to be compared with the CPU oracle's rank-3 trust region on the same Q: 1404 tCG iterations / 134 outer iterations, stop reason 5,
f = 5.714216498489204 (python: oracle.xm_oracle.trustregion(gen_dense(1778)['Q'], I-stack, ones, gradtol 1e-6), 11 s on 8 cores)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xm-code_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, xmamd, xm_testlib as tl
Q = tl.gen_dense(1778, seed=1778)["Q"]
ctx = xmamd.Context(Q=Q)
for flags in (0, xmamd.FLAG_HOST_STEPPED):
    R, s, i = ctx.solve(3, 1e-6, 0.0, trace=2000, flags=flags)
    print(f"GPU rank-3 stage (flags {flags}): tcg {i['tcg_iters']} outer {i['outer_iters']} stop {i['last_stop_reason']} f {i['primal']!r} status {i['status']} min_eig {i['min_eig']:.3e}")
    print(" inner counts:", i["trace"][:, 2].astype(int).tolist())
    if flags == 0:
        print(" last 16 outer iterations: loss, gradnorm, inner, endreason, trstatus, delta")
        for row in i["trace"][-16:]:
            print("   %.12f %.3e %4d %d %d %.3e" % tuple(row))
R, s, i = ctx.solve(5, 1e-6, 0.0, trace=4000)
tr = i["trace"]
print("full staircase: tcg", i["tcg_iters"], "rank", i["rank"], "outer", i["outer_iters"])
ctx.close()
