#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(timeout 300 python scripts/kbench_sell.py 100000 50 --o 5 --slabs 4 --gather 1
 timeout 300 python scripts/kbench_sell.py 100000 50 --band --o 3 --slabs 4 --gather 1 --lmax 16
 timeout 300 python scripts/kbench_sell.py 100000 50 --band --o 3 --slabs 4 --gather 1
 timeout 300 python scripts/kbench_sell.py 100000 20 --skew --o 3 --slabs 4 --gather 1
 echo "bench vg100k bsr (SELL default)"; timeout 600 python bench.py --workload vg100k --storage bsr --steps 5 --warmup 1 --cpu-seconds 0 | tail -1
 echo "bench vg100k bsr (XM_BSR_SELL=0: block-CSR kernel)"; XM_BSR_SELL=0 timeout 600 python bench.py --workload vg100k --storage bsr --steps 5 --warmup 1 --cpu-seconds 0 | tail -1
) 2>&1 | tee gpurun_out/j_misc.log
