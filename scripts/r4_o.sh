#!/bin/bash
# Round 4, pass o: 128-byte record pitch (repacked copy of W, XM_SELL_WSTRIDE=16) at o = 4, 5 with the view-graph codec
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
K="python scripts/kbench_sell.py 100000 50 --codec 1 --layout 1 --no-csr --slabs 4 --gather 1"
(timeout 300 $K --o 3 4 5
 echo "XM_SELL_WSTRIDE=16"; XM_SELL_WSTRIDE=16 timeout 300 $K --o 3 4 5) 2>&1 | grep -v "^$\|amdgpu.ids" | tee gpurun_out/r4o_kbench.log
