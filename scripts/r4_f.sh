#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -rf -k "symmetric_window or rome13682_dense or multi_process_run or two_ranks_one_gpu or overlap" 2>&1 | tail -60 | cut -c1-300 | tee gpurun_out/r4f_pytest_symw.log
(timeout 900 python scripts/kbench_symw.py 13682 --o 3 --worlds 8; for k in 8 16 32; do echo "XM_SYMW_K=$k"; XM_SYMW_K=$k timeout 300 python scripts/kbench_symw.py 13682 --o 3 --worlds 8 | head -1; done) 2>&1 | grep -v "^$\|amdgpu.ids" | tee gpurun_out/r4f_kbench_symw.log
