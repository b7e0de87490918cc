#!/bin/bash
# Round 4, pass l: one transposition buffer per wavefront (LDS 40 -> 20 KB, 64 -> 32 KB), three wavefronts per SIMD at o = 4 / 5, aligned
# 8-byte element fetch (gather 3); then the vector-memory counters of the default kernel.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests -m gpu -q -x --timeout 600 -k "sell or viewgraph" 2>&1 | tail -3 | tee gpurun_out/r4l_pytest.log
(timeout 300 python scripts/kbench_sell.py 100000 50 --o 3 4 5 --slabs 4 --gather 1 3 --codec 1 --layout 1 --no-csr
 XM_SELL_PIPE=4 timeout 300 python scripts/kbench_sell.py 100000 50 --o 4 5 --slabs 4 --gather 1 --codec 1 --layout 1 --no-csr
 timeout 300 python scripts/kbench_sell.py 100000 50 --o 3 4 5 --slabs 4 --gather 1 --codec 0 --layout 1 --no-csr
 XM_SELL_PIPE=0 timeout 300 python scripts/kbench_sell.py 100000 50 --o 3 --slabs 4 --gather 1 --codec 0 1 --layout 1 --no-csr
 timeout 300 python scripts/kbench_sell.py 100000 50 --o 3 --slabs 2 8 --gather 1 --codec 1 --layout 1 --no-csr) 2>&1 | grep -v "^$\|amdgpu.ids" | tee gpurun_out/r4l_kbench.log
SELL_OUT=r4l_pmc_sell_diag2.json bash scripts/pmc_sell_diag2.sh > gpurun_out/r4l_pmc_diag2.out 2>&1
tail -5 gpurun_out/r4l_pmc_diag2.out
