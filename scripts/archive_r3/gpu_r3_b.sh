#!/bin/bash
# debug pass: the single-process multi-GPU cases that failed in pass A, with their output
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8 XM_WATCHDOG_S=20
python - <<'PY' 2>&1 | tail -150 | tee gpurun_out/r3b_team.log
import os, sys, subprocess, time
sys.path[:0] = ["tests", "xm-code_amd"]
import test_gpu_round3 as t3
code = t3._team_worker_code()
env = dict(os.environ)
for case, world in (("dense", 2), ("bsr", 2), ("sell", 2)):
    for ex in ("2", "1"):
        e = dict(env, XM_EXCHANGE=ex, XM_COMM_TRACE="/tmp/tr_%s_%s" % (case, ex))
        if case == "sell":
            e["XM_BSR_SELL"] = "1"
        t0 = time.time()
        p = subprocess.run([sys.executable, "-c", code, "team", str(world), "/tmp/o_%s_%s.npz" % (case, ex), case], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        print("=== %s world %d exchange %s: rc %d in %.1fs" % (case, world, ex, p.returncode, time.time() - t0))
        print(p.stdout.decode()[-1500:])
        for r in range(world):
            f = "/tmp/tr_%s_%s.%d" % (case, ex, r)
            if os.path.exists(f):
                L = open(f).read().splitlines()
                print("  trace rank %d: %d lines, last: %s" % (r, len(L), L[-3:]))
PY
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "bench_two_ranks_flow or two_ranks_one_gpu" 2>&1 | tail -15 | tee gpurun_out/r3b_suite.log
