"""debug aid: which rows of the chunk-tiled product differ from the reference"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xm-code_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, xmamd, xm_testlib as tl
for (n, deg, o, S, kmax, gm) in [(64, 8, 3, 1, 64, 0), (64, 8, 3, 4, 64, 0), (200, 8, 3, 1, 64, 0), (200, 8, 3, 4, 64, 0), (200, 8, 3, 4, 64, 1), (200, 8, 1, 4, 64, 0)]:
    P = tl.gen_vg(n, deg=deg, sigma=0.3, seed=n + o)
    W = np.random.default_rng(n).standard_normal((3 * n, o))
    ref = (P["Q"] @ W).reshape(n, 3, o)
    M = xmamd.SellMatrix(P["rowptr"], P["colidx"], P["blocks"], slabs=S, lmax=kmax, layout=2)
    got = M.qw(W, 1.0, gather=gm).reshape(n, 3, o)
    M.close()
    err = np.abs(got - ref).reshape(n, -1).max(axis=1)
    bad = np.nonzero(err > 1e-10)[0]
    L = xmamd.sell2_layout(P["rowptr"], P["colidx"], slabs=S, kmax=kmax)
    print(f"n={n} deg={deg} o={o} S={S} gm={gm}: {bad.size} bad rows of {n}; K per slice {np.diff(L['slice_off']).tolist()[:8]}; first bad {bad[:20].tolist()}")
    if bad.size:
        r = bad[0]
        print("   row", r, "got", got[r].ravel()[:4], "ref", ref[r].ravel()[:4], "len", P["rowptr"][r + 1] - P["rowptr"][r])
        # is the wrong value a partial sum? compare with per-slab contributions
        cols = P["colidx"][P["rowptr"][r]:P["rowptr"][r + 1]]; blk = P["blocks"][P["rowptr"][r]:P["rowptr"][r + 1]]
        contrib = np.einsum("bij,bjk->bik", blk, W.reshape(n, 3, o)[cols])
        cs = np.cumsum(contrib, axis=0)
        print("   prefix sums elem0:", cs[:, 0, 0], " got elem0:", got[r, 0, 0], " cols", cols, "slab", cols * S // n)
