#!/bin/bash
# Round 4, pass s: product input also at the 128-byte record pitch (xm_qw_sell_padded): parity, then kbench native vs padded on the random,
# banded and hub graphs
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "qw_sell or padded_product" 2>&1 | tail -3 | tee gpurun_out/r4s_pytest.log
K="python scripts/kbench_sell.py 100000 50 --layout 1 --no-csr --slabs 4 --gather 1"
(timeout 300 $K --o 3 4 5 --codec 1 0
 timeout 300 $K --o 3 4 5 --codec 1 0 --padded
 timeout 300 $K --o 3 --codec 1 --band; timeout 300 $K --o 3 --codec 1 --band --padded
 timeout 300 $K --o 3 --codec 1 --skew; timeout 300 $K --o 3 --codec 1 --skew --padded) 2>&1 | grep -v "^$\|amdgpu.ids" | tee gpurun_out/r4s_kbench.log
