#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "symmetr or sym or certified or matrix_free_symmetric" 2>&1 | tail -3
XM_SYMV_K=16 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "symmetric_kernel_matches" 2>&1 | tail -2
(for n in 1778 4096 8192 13682; do timeout 200 python scripts/kbench_dense.py $n 3 4 | grep SYM; done) 2>&1 | tee gpurun_out/symv9.log
