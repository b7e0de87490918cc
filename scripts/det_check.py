"""run the same staircase solve repeatedly in one process and compare everything bitwise (determinism check)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xm-code_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, xmamd, xm_testlib as tl
P = tl.gen_vg(41, deg=3, sigma=1.5, seed=40)
ctx = xmamd.Context(Q=P["Q"])
ref = None
for i in range(12):
    R, s, info = ctx.solve(6, 1e-9, 3.0, trace=4000)
    key = (info["rank"], info["status"], info["primal"], info["min_eig"], info["dual"], info["lanczos_iters"], info["tcg_iters"])
    same = ref is None or (key == ref[0] and np.array_equal(R, ref[1]))
    print(i, key, "SAME" if same else "DIFFERENT")
    if ref is None: ref = (key, R.copy())
