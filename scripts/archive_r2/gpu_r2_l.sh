#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "matrix_free or spd_inverse" 2>&1 | tail -3
 timeout 120 python scripts/kbench_schur.py 1778 200000 6
 timeout 300 python scripts/kbench_schur.py 13682 800000 8
) 2>&1 | tee gpurun_out/l_schur.log
