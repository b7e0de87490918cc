# exploration: which small view graph with outlier edges gives a well-posed XM^2 round (status 1 on one GPU and on two virtual GPUs)
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

def plain(n, deg, sigma, seed):
    P = tl.gen_vg(n, deg=deg, sigma=sigma, seed=seed, dense=False)
    e = np.asarray(P["edges"])
    return dict(ei=e[:, 0].astype(np.int32), ej=e[:, 1].astype(np.int32), w=np.asarray(P["w"], dtype=float), M=np.asarray(P["M"], dtype=float).reshape(-1, 3, 3))

for name, gen, wscale in (("hubs400", lambda: tl.gen_vg_hubs(400, 8, 2, 0.3, 0.05, seed=12), 0.2), ("hubs400", lambda: tl.gen_vg_hubs(400, 8, 2, 0.3, 0.05, seed=12), 0.1),
                  ("vg300", lambda: plain(300, 10, 0.05, 5), 0.2), ("vg300", lambda: plain(300, 10, 0.05, 5), 0.1)):
    H = gen()
    H["w"] = H["w"] * wscale
    for tol in (1e-8,):
        for kw in (dict(), dict(n_gpus=2, gpu_map=1)):
            M = corrupt(H["M"], 0.08, 3)
            ctx = xmamd.Context(vg=(H["ei"], H["ej"], H["w"], M), n=int(max(H["ei"].max(), H["ej"].max())) + 1, **kw)
            R, s, info = ctx.solve(5, tol, 20.0)
            os.environ["XM_QUIET"] = "0"
            R2, s2, i2, x2 = ctx.xm2_round(R, s, 5, tol, percentile=90.0, flags=xmamd.FLAG_VERBOSE if len(sys.argv) > 1 else 0)
            ctx.close()
            print(name, wscale, tol, kw, "first", info["status"], info["rank"], "| round", i2["status"], i2["rank"], i2["tcg_iters"], f"{i2['primal']:.9g}", "cert_flags", i2["cert_flags"], "eig_resid", i2["eig_residual"], "lanczos", i2["lanczos_iters"], "min_eig", i2["min_eig"],
                  {k: x2[k] for k in ("removed", "regularised", "lam_used", "rank3_status", "s_avg", "s_std", "n_small")}, flush=True)
