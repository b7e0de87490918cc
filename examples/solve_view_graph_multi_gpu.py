#!/usr/bin/env python3
"""The north_star workload end to end: a view graph handed over as its EDGE LIST, solved on N GPUs from ONE process, followed by one
round of the reference's XM^2 outlier loop (3_test_colmap_glomap.py:299-351) on the resident context.

    python examples/solve_view_graph_multi_gpu.py [n_cameras=20000] [n_gpus=1] [gpu_map=0]

    ctx = xmamd.Context(vg=(ei, ej, w, M), n=n, n_gpus=N)     XM_STORAGE_VIEWGRAPH + xm_problem_t.n_gpus: block CSR balanced by stored
                                                              blocks over the GPUs, quaternion-compressed sliced ELL above 1 M blocks per
                                                              GPU, one host thread per GPU, direct peer-write exchange inside the tCG
    R, s, info = ctx.solve(5, tol, lam)
    res = ctx.edge_residuals(); ctx.set_edge_weights(w2); ctx.solve(..., R_ini=R, s_ini=s)      the XM^2 re-weighting, Q never re-uploaded
gpu_map = 1 runs the N ranks as virtual devices on GPU 0 (a functional run of the multi-GPU path on a 1-GPU machine).  Needs an MI355X."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xm-code_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np            # noqa: E402
import xmamd                  # noqa: E402
import xm_testlib as tl       # noqa: E402  (the seeded view-graph generator of SURVEY 8d)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
n_gpus = int(sys.argv[2]) if len(sys.argv) > 2 else 1
gpu_map = int(sys.argv[3]) if len(sys.argv) > 3 else 0
P = tl.gen_vg(n, deg=30, sigma=0.05, seed=n, dense=False)
e, w, M = P["edges"], P["w"].copy(), P["M"].copy()
rng = np.random.default_rng(1)
bad = rng.choice(e.shape[0], size=e.shape[0] // 50, replace=False)          # 2 % outliers: random relative rotations
M[bad] = tl.haar_so3(rng, bad.size)
t0 = time.time()
ctx = xmamd.Context(vg=(e[:, 0], e[:, 1], w, M), n=n, n_gpus=n_gpus, gpu_map=gpu_map)
print(f"{n} cameras, {e.shape[0]} edges ({bad.size} outliers), {n_gpus} GPU(s): context in {time.time() - t0:.2f} s")
R, s, info = ctx.solve(5, 1e-6, 1000.0)
print(f"first solve: rank {info['rank']}, status {info['status']}, primal {info['primal']:.6f}, {info['tcg_iters']} tCG iterations in "
      f"{info['seconds'] * 1e3:.1f} ms, exchange mode {info['exchange']}")
res = ctx.edge_residuals()
thr = np.percentile(w * res, 90)
w2 = np.where(w * res > thr, 0.0, w)
print(f"XM^2 filter: threshold {thr:.3e}, removed {int((w2 == 0).sum())} edges, of the planted outliers {int((w2[bad] == 0).sum())} / {bad.size}")
ctx.set_edge_weights(w2)
R2, s2, i2 = ctx.solve(5, 1e-6, 1000.0, mode=xmamd.MODE_REBUTTLE, s_ini=s, R_ini=R)
print(f"warm re-solve: rank {i2['rank']}, status {i2['status']}, primal {i2['primal']:.6f}, {i2['tcg_iters']} tCG iterations in {i2['seconds'] * 1e3:.1f} ms")
ctx.close()
