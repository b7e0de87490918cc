#!/bin/bash
# Round 4, pass n: cache policy of the gathered records (plain / nt / sc1 / sc0 sc1 through buffer loads) and the 128-byte record pitch
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
K="python scripts/kbench_sell.py 100000 50 --o 3 --codec 1 --layout 1 --no-csr"
(timeout 300 $K --slabs 4 --gather 1 4
 for a in 2 16 17; do echo "XM_SELL_GAUX=$a"; XM_SELL_GAUX=$a timeout 300 $K --slabs 4 --gather 1 4; done
 echo "XM_SELL_WSTRIDE=16"; XM_SELL_WSTRIDE=16 timeout 300 $K --slabs 4 8 --gather 1
 echo "XM_SELL_WSTRIDE=16 XM_SELL_GAUX=16"; XM_SELL_WSTRIDE=16 XM_SELL_GAUX=16 timeout 300 $K --slabs 4 8 --gather 1) 2>&1 | grep -v "^$\|amdgpu.ids" | tee gpurun_out/r4n_kbench.log
