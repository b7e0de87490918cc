#!/usr/bin/env python3
"""tests/golden/simple2/tp.npz: what the reference's own 2_test_creatematrix.py obtains from utils/recoversolution.py:recover_XM
for assets/SIMPLE2 — anchored rotations R_real (3 x 3N), scales s_real (N), and the translations / landmarks
t_est (3 x N), p_est (3 x M) = Abar @ sR_real^T (recoversolution.py:77-86, Abar.bin written by creatematrix.py:283-311).
Runs only in the build container (needs /root/reference); the pipeline is executed unchanged through
make_golden.run_reference_pipeline_simple2, only its variables are captured.  The fixture pins xm_ctx_recover_tp (the same
quantities recomputed from the observation list, without Abar) and its numpy restatement tests/xm_testlib.py:schur_tp_numpy."""
import os, sys, tempfile
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import make_golden as mg
import xm_testlib as tl

with tempfile.TemporaryDirectory() as wd:
    ds, g = mg.run_reference_pipeline_simple2(wd)
R_real, s_real, t_est, p_est = (np.asarray(g[k], dtype=np.float64) for k in ("R_real", "s_real", "t_est", "p_est"))
print("R_real", R_real.shape, "s_real", s_real.shape, "t_est", t_est.shape, "p_est", p_est.shape)
obs = np.load(os.path.join(HERE, "simple2", "obs.npz"))
t, p = tl.schur_tp_numpy(obs["cam"], obs["lm"], obs["p"], obs["w"], R_real, s_real)
print("restatement vs reference: t", np.abs(t - t_est).max() / np.abs(t_est).max(), "p", np.abs(p - p_est).max() / np.abs(p_est).max())
assert np.abs(t - t_est).max() < 1e-9 * np.abs(t_est).max() and np.abs(p - p_est).max() < 1e-9 * np.abs(p_est).max()
np.savez_compressed(os.path.join(HERE, "simple2", "tp.npz"), R_real=R_real, s_real=s_real, t_est=t_est, p_est=p_est)
