#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sell" 2>&1 | tail -3
 for PIPE in 0 1; do echo "PIPE=$PIPE"; XM_SELL_PIPE=$PIPE timeout 300 python scripts/kbench_sell.py 100000 50 --o 3 --slabs 1 2 4 --gather 0 1 --no-csr; done
 echo "PIPE=1 o=5"; XM_SELL_PIPE=1 timeout 300 python scripts/kbench_sell.py 100000 50 --o 5 --slabs 4 --gather 0 1 --no-csr
 echo "PIPE=1 banded"; XM_SELL_PIPE=1 timeout 300 python scripts/kbench_sell.py 100000 50 --band --o 3 --slabs 1 4 --gather 0 1 --no-csr --lmax 16
) 2>&1 | tee gpurun_out/c_pipe.log
