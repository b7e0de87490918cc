# exploration, part 2: is it the matrix after set_edge_weights, or state carried between solves?
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "xm-code_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, xmamd, xm_testlib as tl
exec(open(os.path.join(ROOT, "scripts/archive_r3/dbg_xm2_cert.py")).read().split("ctx = xmamd.Context")[0])
ctx = xmamd.Context(vg=(ei, ej, w, M), n=n)
R, s, i0 = ctx.solve(5, 1e-8, 20.0); show("A first", i0)
ctx.set_edge_weights(w)
R, s, i1 = ctx.solve(5, 1e-8, 20.0); show("A after set_edge_weights(same w)", i1)
rot, scale, _ = xmamd.recover_rotations(R, s)
res = ctx.edge_residuals_recovered(rot, scale)
err = w * res; thr = np.percentile(err, 90.0); w2 = np.where(err > thr, 0.0, w); lam = (w2 != 0).sum() / n
ctx.set_edge_weights(w2)
B = xmamd.Context(vg=(ei, ej, w2, M), n=n)
rng = np.random.default_rng(0)
for o in (1, 3, 4):
    W = rng.standard_normal((3 * n, o))
    a, b = ctx.qw(W), B.qw(W)
    print("o", o, "rel diff of Q W between re-weighted and fresh context", tl.rel_fro(a, b), flush=True)
Rf, sf, i_f = ctx.solve(5, 1e-8, lam); show("A re-weighted, lam", i_f)
Rf, sf, i_f = ctx.solve(5, 1e-8, 20.0); show("A re-weighted, 20", i_f)
Rg, sg, ig = B.solve(5, 1e-8, lam); show("B fresh, lam", ig)
Rg, sg, ig = B.solve(5, 1e-8, 20.0); show("B fresh, 20", ig)
# zero-degree cameras after the filter?
deg = np.zeros(n); np.add.at(deg, ei, (w2 != 0)); np.add.at(deg, ej, (w2 != 0)); print("min degree after filter", deg.min())
ctx.close(); B.close()
