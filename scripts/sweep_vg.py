import sys, os, time
sys.path.insert(0, "xm-code_amd"); sys.path.insert(0, "tests")
import numpy as np, xmamd, xm_testlib as tl
for n, deg, sigma in [(13682, 30, 0.05), (13682, 30, 0.2)]:
    P = tl.gen_vg(n, deg=deg, sigma=sigma, seed=n, dense=False)
    ctx = xmamd.Context(bsr=(P["rowptr"], P["colidx"], P["blocks"]))
    for lam in (0.0, 1.0, 30.0, 1000.0):
        for tol in (1e-6, 1e-3):
            t = time.time(); R, s, i = ctx.solve(5, tol, lam, trace=3100); el = time.time() - t
            tr = i["trace"]
            print(f"n={n} sigma={sigma} lam={lam} tol={tol}: status {i['status']} rank {i['rank']} outer {i['outer_iters']} tcg {i['tcg_iters']} "
                  f"primal {i['primal']:.6g} min_eig {i['min_eig']:.3g} s[min,max]=({s.min():.3g},{s.max():.3g}) {el:.2f}s  last gradnorm {tr[-1,1]:.3g} reasons {np.bincount(tr[:,3].astype(int), minlength=8)}", flush=True)
    ctx.close()
