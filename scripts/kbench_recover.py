"""micro-benchmark of the solution recovery's per-camera projection (utils/recoversolution.py:65-86): the default kernel (one thread
per camera, scaled Newton polar iteration) beside north_star's form (one wavefront per camera, cross-lane reductions):
   python scripts/kbench_recover.py [n ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xm-code_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, xmamd, xm_testlib as tl

for n in [int(x) for x in sys.argv[1:]] or [1778, 13682, 100000]:
    rng = np.random.default_rng(n)
    for r in (3, 5):
        R = np.concatenate(list(tl.haar_so3(rng, n)), axis=0)
        if r > 3:
            R = np.concatenate([R, 1e-3 * rng.standard_normal((3 * n, r - 3))], axis=1)
        s = rng.uniform(0.5, 2.0, n)
        out = []
        for variant in (0, 1):
            rot, sc, neg, ms = xmamd.recover_rotations(R, s, variant=variant, reps=50)
            out.append((rot, ms))
        print(f"recover_project n={n} r={r}: thread per camera {out[0][1]*1e3:8.2f} us | wavefront per camera {out[1][1]*1e3:8.2f} us | "
              f"difference of the results {tl.rel_fro(out[1][0], out[0][0]):.1e}", flush=True)
