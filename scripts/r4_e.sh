#!/bin/bash
# Round 4: multi-rank symmetric window product -- tests, per-rank timing at Final size; PMC legs of the bench (traffic + traced durations)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -rf -k "symmetric_window or rome13682_dense or multi_process_run or two_ranks_one_gpu or overlap" 2>&1 | tail -15 | tee gpurun_out/r4e_pytest_symw.log
(timeout 900 python scripts/kbench_symw.py 13682 --o 3 --worlds 2 4 8; timeout 300 python scripts/kbench_symw.py 13682 --o 4 --worlds 8) 2>&1 | grep -v "^$" | tee gpurun_out/r4e_kbench_symw.log
timeout 2400 python scripts/pmc_legs.py r04 venice hbm13682 vg100k_vg > gpurun_out/r4e_pmc_legs.out 2>&1; tail -3 gpurun_out/r4e_pmc_legs.out
