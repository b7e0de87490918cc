#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sell or skew or bsr" 2>&1 | tail -3
(echo 'hub cameras (skewed degrees):'; python scripts/kbench_sell.py 100000 20 --skew --o 3 5 --slabs 4 --gather 1 --no-csr
 python scripts/kbench_sell.py 100000 50 --o 3 --slabs 4 --gather 1 --no-csr) 2>&1 | grep -v "^$" | tee gpurun_out/kbench3.log
