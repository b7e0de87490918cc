# exploration: why the final solve of an XM^2 round rejected a rank-3 optimum the CPU oracle certifies (third solve on one context)
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "xm-code_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, xmamd, xm_testlib as tl

def corrupt(M, frac, seed):
    M = M.copy(); rng = np.random.default_rng(seed)
    bad = rng.choice(M.shape[0], size=int(M.shape[0] * frac), replace=False)
    for e in bad:
        q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        M[e] = q * np.sign(np.linalg.det(q))
    return M
P = tl.gen_vg(300, deg=10, sigma=0.05, seed=5, dense=False)
e = np.asarray(P["edges"]); ei, ej = e[:, 0].astype(np.int32), e[:, 1].astype(np.int32)
w = np.asarray(P["w"], dtype=float) * 0.2; M = corrupt(np.asarray(P["M"], dtype=float).reshape(-1, 3, 3), 0.08, 3)
n = 300
def show(tag, i):
    print(tag, "status", i["status"], "rank", i["rank"], "tcg", i["tcg_iters"], "primal %.9g" % i["primal"], "min_eig %.3e" % i["min_eig"], "lanczos", i["lanczos_iters"], flush=True)
ctx = xmamd.Context(vg=(ei, ej, w, M), n=n)
R, s, i0 = ctx.solve(5, 1e-8, 20.0); show("first", i0)
rot, scale, _ = xmamd.recover_rotations(R, s)
res = ctx.edge_residuals_recovered(rot, scale)
err = w * res; thr = np.percentile(err, 90.0); w2 = np.where(err > thr, 0.0, w); lam = (w2 != 0).sum() / n
ctx.set_edge_weights(w2)
mode = sys.argv[1] if len(sys.argv) > 1 else "all"
if mode in ("all", "rank3"):
    R3, s3, i3 = ctx.solve(3, 1e-8, 0.0, mode=xmamd.MODE_RANK3); show("rank3 lam0", i3)
Rf, sf, i_f = ctx.solve(5, 1e-8, lam); show("final (same ctx)", i_f)
Rf, sf, i_f = ctx.solve(5, 1e-8, lam); show("final again", i_f)
ctx.close()
c2 = xmamd.Context(vg=(ei, ej, w2, M), n=n)
Rg, sg, ig = c2.solve(5, 1e-8, lam); show("fresh ctx", ig)
R3, s3, i3 = c2.solve(3, 1e-8, 0.0, mode=xmamd.MODE_RANK3); show("fresh: rank3 lam0", i3)
Rg, sg, ig = c2.solve(5, 1e-8, lam); show("fresh: after rank3", ig)
c2.close()
rp, ci, bl = tl.vg_from_edges(n, ei, ej, w2, M)
c3 = xmamd.Context(bsr=(rp, ci, bl))
Rg, sg, ig = c3.solve(5, 1e-8, lam); show("fresh bsr ctx", ig)
c3.close()
