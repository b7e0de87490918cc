"""why does the Final-size synthetic scene end at max_rank without a certificate?  prints the per-rank trace summary"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xm-code_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, xmamd, xm_testlib as tl
N, M, views = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
S = tl.gen_scene(N, M, views, seed=N)
ctx = xmamd.Context(obs=(S["cam"], S["lm"], S["p"], S["w"]))
R, s, i = ctx.solve(int(sys.argv[4]) if len(sys.argv) > 4 else 5, 1e-6, 0.0, trace=8000)
tr = i["trace"]
print({k: i[k] for k in ("rank", "status", "primal", "dual", "min_eig", "gap", "tcg_iters", "outer_iters", "cert_flags") if k in i}); print(sorted(i.keys()))
# trace rows: loss, gradnorm, inner, endreason, trstatus, delta ; a new rank level starts where the loss jumps / k restarts
print("rows", len(tr))
idx = [0] + [k for k in range(1, len(tr)) if tr[k, 0] > tr[k - 1, 0] * (1 + 1e-9) + 1e-12] + [len(tr)]
for a, b in zip(idx[:-1], idx[1:]):
    seg = tr[a:b]
    print(f"segment rows {a}..{b}: loss {seg[0,0]:.6e} -> {seg[-1,0]:.6e}, gradnorm {seg[0,1]:.2e} -> {seg[-1,1]:.2e}, inner total {int(seg[:,2].sum())}, endreasons {np.bincount(seg[:,3].astype(int), minlength=7).tolist()}")
print("s range", s.min(), s.max())
ctx.close()
