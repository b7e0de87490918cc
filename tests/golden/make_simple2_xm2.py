#!/usr/bin/env python3
"""tests/golden/simple2/xm2.npz: the XM^2 outlier statistics of the reference's own loop (3_test_colmap_glomap.py:299-321) on
assets/SIMPLE2 -- the per-observation `error`, the 90-percentile `threshold` and the indices it removes -- computed by EXECUTING
those lines of the reference script (read from /root/reference at generation time, never copied into this repository) on the
variables of the reference's own pipeline: R_real, s_real, t_est, p_est as captured in tp.npz (make_simple2_tp.py) and the
observation list edges / landmarks / weights as captured in obs.npz (make_simple2_obs.py).  Runs only in the build container.
The fixture pins xm_ctx_edge_residuals_recovered / xm_ctx_xm2_round (tests/test_gpu_round3.py) and the numpy statement
tests/xm_testlib.py:xm2_error_numpy (CPU test)."""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import xm_testlib as tl

REF = "/root/reference/3_test_colmap_glomap.py"
src = open(REF).read().splitlines()
first = next(i for i, l in enumerate(src) if l.startswith("src_idx = edges[:, 0] - 1"))
last = next(i for i, l in enumerate(src) if l.startswith("indices_to_remove = np.where(error > threshold)"))
block = "\n".join(src[first:last + 1])
print("executing %s lines %d-%d:\n%s\n" % (REF, first + 1, last + 1, block))

tp = np.load(os.path.join(HERE, "simple2", "tp.npz"))
obs = np.load(os.path.join(HERE, "simple2", "obs.npz"))
N = int(tp["s_real"].shape[0]); M = int(tp["p_est"].shape[1])
ns = dict(np=np, N=N, M=M, R_real=np.array(tp["R_real"]), s_real=np.array(tp["s_real"]).reshape(-1), t_est=np.array(tp["t_est"]),
          p_est=np.array(tp["p_est"]), edges=np.stack([obs["cam"] + 1, obs["lm"] + 1], axis=1).astype(np.int64),
          landmarks=np.array(obs["p"]), weights=np.array(obs["w"]).reshape(-1), print=print)
exec(block, ns)
error, threshold, rem = np.asarray(ns["error"]), float(ns["threshold"]), np.asarray(ns["indices_to_remove"])
print("observations", error.size, "sum", error.sum(), "threshold", threshold, "removed", rem.size)
mine = tl.xm2_error_numpy(obs["cam"], obs["lm"], obs["p"], obs["w"], tp["R_real"], tp["s_real"], tp["t_est"], tp["p_est"])
assert np.abs(mine - error).max() <= 1e-12 * error.max()
np.savez_compressed(os.path.join(HERE, "simple2", "xm2.npz"), error=error, threshold=threshold, removed=rem.astype(np.int64))
