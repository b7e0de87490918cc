#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(for V in 4 8; do for C in 1 2 4; do echo "XM_SYM_WAVES=$V XM_SYM_CPW=$C"; export XM_SYM_WAVES=$V XM_SYM_CPW=$C; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "symmetric" 2>&1 | tail -1
   timeout 300 python scripts/kbench_dense.py 13682 3 4 | grep SYM; timeout 300 python scripts/kbench_dense.py 1778 3 | grep SYM; done; done
) 2>&1 | tee gpurun_out/g_sym.log
