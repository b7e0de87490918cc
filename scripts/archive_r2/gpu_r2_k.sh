#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sell or xm2 or two_ranks or vg100k" 2>&1 | tail -3
 timeout 300 python scripts/kbench_sell.py 100000 50 --o 3 4 5 --slabs 4 --gather 1 --no-csr
 timeout 600 python bench.py --workload vg100k --storage bsr --steps 5 --warmup 1 --cpu-seconds 0 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('vg100k', d['ms_per_step'], d['value'], d['solve']['tcg_iters_per_solve'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
) 2>&1 | tee gpurun_out/k_quad.log
