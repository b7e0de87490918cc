#!/usr/bin/env python3
"""The reference's second driver (2_test_creatematrix.py) WITHOUT the dense Q and WITHOUT Abar:

    reference:  create_matrix(weights, edges, landmarks, path)   -> Q.bin (72 N^2 bytes) + Abar.bin ((N-1+M) x 3N doubles)
                XM.solve(path, 5, 1e-1, lam, 1000)               -> R.bin, s.bin
                recover_XM(Q, R, s, Abar, lam)                   -> rotations, scales, translations, landmarks
    here:       ctx = xmamd.Context(obs=(cam, lm, p, w))         the observation list itself (XM_STORAGE_SCHUR)
                R, s, info = ctx.solve(5, tol, lam)
                rot, scale, _ = xmamd.recover_rotations(R, s);  t, P = ctx.recover_tp(rot, scale)

The observation list is the one the reference's own pipeline hands to create_matrix for assets/SIMPLE2 (tests/golden/simple2/obs.npz,
captured by tests/golden/make_simple2_obs.py); tp.npz holds what the reference's run produced.  Needs an MI355X."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xm-code_amd"))
import numpy as np            # noqa: E402
import xmamd                  # noqa: E402

G = os.path.join(ROOT, "tests", "golden", "simple2")
Z = np.load(os.path.join(G, "obs.npz"))
ref = np.load(os.path.join(G, "tp.npz"))
ctx = xmamd.Context(obs=(Z["cam"], Z["lm"], Z["p"], Z["w"]))
R, s, info = ctx.solve(5, 1e-10, 0.0)
rot, scale, nneg = xmamd.recover_rotations(R, s)
t, P = ctx.recover_tp(rot, scale)
res = ctx.edge_residuals()                                       # |s_i R_i p + t_i - P_l|^2 per observation (the XM^2 loop's input)
ctx.close()
print(f"cameras {scale.size}, landmarks {P.shape[1]}, observations {Z['cam'].size}: rank {info['rank']}, status {info['status']}, "
      f"primal {info['primal']:.6e}, min eig {info['min_eig']:.2e}, {info['tcg_iters']} tCG iterations")
print(f"weighted residual sum {float(np.sum(Z['w'].reshape(-1) * res)):.6e} (= primal), median residual {np.median(np.sqrt(res)):.3e}")
print("against the reference's own run (solved to tol 1e-1 only): rotations", float(np.abs(rot - ref['R_real']).max()),
      "translations", float(np.abs(t - ref['t_est']).max()), "landmarks", float(np.abs(P - ref['p_est']).max()))
