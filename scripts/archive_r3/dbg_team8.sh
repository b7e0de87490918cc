# 8 virtual ranks on the hub view graph: where does the exchange stall?  (collective log of every rank)
mkdir -p gpurun_out/t8; rm -f gpurun_out/t8/*
cat > /tmp/w.py <<'PY'
import sys, os
R = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, os.path.join(R, "xm-code_amd")); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, xmamd, xm_testlib as tl
H = tl.gen_vg_hubs(600, 8, 3, 0.3, 0.1, seed=8)
ctx = xmamd.Context(vg=(H["ei"], H["ej"], H["w"], H["M"]), n=600, n_gpus=int(sys.argv[1]), gpu_map=1)
try:
    R_, s, info = ctx.solve(5, 1e-9, 20.0)
    print("ok", info["status"], info["rank"], info["tcg_iters"], info["seconds"])
except Exception as e:
    print("FAILED", e)
PY
export GPU_MAX_HW_QUEUES=${QUEUES:-16} XM_WATCHDOG_S=15 XM_BSR_SELL=${SELL:-1}
XM_COMM_TRACE=gpurun_out/t8/tr timeout 120 python /tmp/w.py 8
for r in 0 1 2 3 4 5 6 7; do echo "rank $r: $(wc -l < gpurun_out/t8/tr.$r) lines; last: $(tail -1 gpurun_out/t8/tr.$r)"; done
