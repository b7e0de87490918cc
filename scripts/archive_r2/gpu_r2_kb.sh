#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "certified or lanczos or vg100k or rome or 13682 or full_size or recorded or oracle_large" 2>&1 | tail -3
(timeout 600 python bench.py --workload vg100k --storage bsr --steps 3 --warmup 1 --cpu-seconds 0 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['solve'])") 2>&1 | grep -v amdgpu.ids | tee gpurun_out/vg100k_dots.log
