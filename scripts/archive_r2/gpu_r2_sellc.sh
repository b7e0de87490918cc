#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sell" 2>&1 | tail -3
(for w in 0 16; do echo "WSTRIDE=$w"; XM_SELL_WSTRIDE=$w timeout 300 python scripts/kbench_sell.py 100000 50 --o 3 4 5 --slabs 4 --gather 1 --no-csr
 XM_SELL_WSTRIDE=$w timeout 300 python scripts/kbench_sell.py 100000 50 --o 3 --slabs 4 --gather 0 --no-csr
 XM_SELL_WSTRIDE=$w timeout 300 python scripts/kbench_sell.py 100000 50 --band --o 3 --slabs 4 --gather 1 --no-csr; done
 echo "WSTRIDE=16 slabs 1 2 8"; for s in 1 2 8; do XM_SELL_WSTRIDE=16 timeout 300 python scripts/kbench_sell.py 100000 50 --o 3 --slabs $s --gather 1 --no-csr; done
) 2>&1 | tee gpurun_out/selle.log
