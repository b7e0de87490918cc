# 4 virtual ranks on the Venice-size dense Q: where does the exchange stall?  (collective log of every rank)
mkdir -p gpurun_out/t4; rm -f gpurun_out/t4/*
cat > /tmp/w4.py <<'PY'
import sys, os
R = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, os.path.join(R, "xm-code_amd")); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, xmamd, xm_testlib as tl
if os.environ.get('WITH_TORCH') == '1':
    import torch
    xmamd.require_gpu(); torch.cuda.set_device(0)
if os.environ.get('WITH_TORCH') == '2':
    import torch
n = int(sys.argv[2])
Q = tl.gen_dense(n, seed=n)["Q"]
ctx = xmamd.Context(Q=Q, n_gpus=int(sys.argv[1]), gpu_map=1)
try:
    for i in range(3):
        R_, s, info = ctx.solve(5, 1e-6, 0.0, flags=xmamd.FLAG_PROFILE_QW, grouping=i % 3)
        print("ok", i, info["status"], info["rank"], info["tcg_iters"], info["seconds"], flush=True)
except Exception as e:
    print("FAILED", e)
PY
export GPU_MAX_HW_QUEUES=16 XM_WATCHDOG_S=15 HSA_ENABLE_SDMA=0
for wt in 0 0; do echo "no trace"; WITH_TORCH=$wt timeout 120 python /tmp/w4.py ${RANKS:-4} ${CAMS:-1778} 2>&1 | tail -4 | cut -c1-200; done
for r in 0 1 2 3; do echo "rank $r: $(wc -l < gpurun_out/t4/tr.$r) lines, $(grep -c allgather gpurun_out/t4/tr.$r) all-gathers; last: $(tail -2 gpurun_out/t4/tr.$r | tr '\n' '|')"; done
