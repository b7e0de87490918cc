"""tCG iterations per rank stage of the Venice-1778-size headline solve, GPU (three summation groupings) and, with --cpu, the CPU oracle
(certificate through LAPACK) on the same Q -- from the per-outer-iteration traces both write (loss, |grad|, inner + 1, exit reason, TR
status, radius): a stage starts at a row whose TR status is 4 (k = 0).   python scripts/stage_iters.py [--cpu] [--n 1778]"""
import sys, os, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "xm-code_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, xm_testlib as tl
ap = argparse.ArgumentParser(); ap.add_argument("--cpu", action="store_true"); ap.add_argument("--gpu", action="store_true"); ap.add_argument("--n", type=int, default=1778)
a = ap.parse_args()
Q = tl.gen_dense(a.n, seed=a.n)["Q"]


def stages(tr):
    starts = [i for i in range(tr.shape[0]) if int(tr[i, 4]) == 4 and (i == 0 or int(tr[i, 2]) == 1)]
    out = []
    for k, s0 in enumerate(starts):
        s1 = starts[k + 1] if k + 1 < len(starts) else tr.shape[0]
        seg = tr[s0:s1]
        out.append(dict(outer=int(seg.shape[0] - 1), tcg=int(seg[1:, 2].sum()), f_end=float(seg[-1, 0]), g_end=float(seg[-1, 1])))
    return out


if a.gpu or not a.cpu:
    import xmamd
    ctx = xmamd.Context(Q=Q)
    for g in range(3):
        R, s, i = ctx.solve(5, 1e-6, 0.0, trace=6000, grouping=g)
        print(f"GPU grouping {g}: total tcg {i['tcg_iters']} outer {i['outer_iters']} rank {i['rank']} lanczos {i['lanczos_iters']}", stages(i["trace"]), flush=True)
    ctx.close()
if a.cpu:
    from oracle import xm_oracle as xo
    xo.use_lapack_eig(True)
    R, s, i = xo.solve(Q, 5, 1e-6, 0.0, 1000.0, trace=6000)
    print(f"CPU oracle ({xo.num_threads()} threads): total tcg {i['tcg_iters']} outer {i['outer_iters']} rank {i['rank']}", stages(i["trace"]), flush=True)
