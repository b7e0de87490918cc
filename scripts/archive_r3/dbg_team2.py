import os, sys, subprocess
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
code = f"""
import sys, os
sys.path.insert(0, {os.path.join(ROOT, 'xm-code_amd')!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})
import numpy as np, xmamd, xm_testlib as tl
import test_gpu_round3 as t3
P = t3._weighted_vg(3000, 16, seed=4)
n_gpus = int(sys.argv[1])
kw = dict(n_gpus=n_gpus, gpu_map=1) if n_gpus > 1 else {{}}
ctx = xmamd.Context(bsr=(P["rowptr"], P["colidx"], P["blocks"]), **kw)
R, s, i = ctx.solve(5, 1e-8, 30.0, flags=xmamd.FLAG_VERBOSE)
ctx.close()
"""
for env, args in [({"XM_BSR_SELL": "0"}, ["2"]), ({"XM_BSR_SELL": "1", "XM_BALANCE": "1"}, ["2"]), ({"XM_BSR_SELL": "1"}, ["1"])]:
    p = subprocess.run([sys.executable, "-c", code] + args, env=dict(os.environ, GPU_MAX_HW_QUEUES="16", XM_WATCHDOG_S="60", **env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    L = p.stdout.decode().splitlines()
    print("=====", env, args)
    keep = [l for l in L if any(k in l for k in ("min eig", "Primal", "Solve TR", "linesearch", "Optimility", "BM ", "Total iteration", "Terminate", "warning"))]
    print("\n".join(keep[:60]))
