import os, sys, subprocess
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
code = f"""
import sys, os
sys.path.insert(0, {os.path.join(ROOT, 'xm-code_amd')!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})
import numpy as np, xmamd, xm_testlib as tl
import test_gpu_round3 as t3
P = t3._weighted_vg(3000, 16, seed=4)
n_gpus = int(sys.argv[1]); storage = sys.argv[2]
kw = dict(n_gpus=n_gpus, gpu_map=1) if n_gpus > 1 else {{}}
if storage == "vg":
    ctx = xmamd.Context(vg=(P["ei"], P["ej"], P["w"], P["M"]), n=3000, **kw)
else:
    ctx = xmamd.Context(bsr=(P["rowptr"], P["colidx"], P["blocks"]), **kw)
R, s, i = ctx.solve(5, 1e-8, 30.0, trace=3000)
print(sys.argv[1:], {{k: os.environ.get(k) for k in ("XM_BSR_SELL", "XM_BALANCE", "XM_SELL_CODEC")}}, "rank", i["rank"], "status", i["status"], "primal", i["primal"], "tcg", i["tcg_iters"], "outer", i["outer_iters"], "min_eig", i["min_eig"])
ctx.close()
"""
for env, args in [({"XM_BSR_SELL": "1"}, ["1", "vg"]), ({"XM_BSR_SELL": "1"}, ["2", "vg"]), ({"XM_BSR_SELL": "1", "XM_BALANCE": "1"}, ["2", "vg"]),
                  ({"XM_BSR_SELL": "0"}, ["2", "bsr"]), ({"XM_BSR_SELL": "0", "XM_BALANCE": "1"}, ["2", "bsr"]), ({"XM_BSR_SELL": "1", "XM_BALANCE": "1"}, ["2", "bsr"]),
                  ({"XM_BSR_SELL": "0"}, ["1", "bsr"])]:
    p = subprocess.run([sys.executable, "-c", code] + args, env=dict(os.environ, GPU_MAX_HW_QUEUES="16", XM_WATCHDOG_S="60", **env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    print(p.stdout.decode().strip().splitlines()[-1][:400])
