#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(for WS in 0 10 12 16; do echo "wstride=$WS"; timeout 300 python scripts/kbench_sell.py 100000 50 --o 3 --slabs 4 8 --gather 0 1 --no-csr --wstride $WS; done
 for WS in 16; do for A in 1 4; do echo "wstride=$WS ABLATE=$A"; XM_SELL_ABLATE=$A timeout 300 python scripts/kbench_sell.py 100000 50 --o 3 --slabs 8 --gather 0 1 --no-csr --wstride $WS; done; done
) 2>&1 | tee gpurun_out/d_wstride.log
