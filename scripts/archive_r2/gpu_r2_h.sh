#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(timeout 300 python scripts/kbench_dense.py 1778 3 4 5
 echo "bench venice default"; timeout 600 python bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-hbm-check --no-rome | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['solve']['tcg_iters_per_solve'], d['roofline']['avg_launch_ms'], d['roofline']['kernel'][:60])"
 echo "bench venice sym everywhere (XM_SYM_MIN_ROWS=0)"; XM_SYM_MIN_ROWS=0 timeout 600 python bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-hbm-check --no-rome | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['solve']['tcg_iters_per_solve'], d['roofline']['avg_launch_ms'], d['roofline']['kernel'][:60])"
 echo "rome dense"; timeout 900 python bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-hbm-check | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['rome_scale'], d['rome_scale_dense'])"
) 2>&1 | tee gpurun_out/h_sym_bench.log
