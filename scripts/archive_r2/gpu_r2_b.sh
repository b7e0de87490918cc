#!/bin/bash
# round-2 GPU call B: ablations of the sliced-ELL product (what bounds it?)
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(for A in 0 1 2 3 4 6 7; do echo "ABLATE=$A (1 no block loads, 2 no gather, 4 no store)"; XM_SELL_ABLATE=$A timeout 300 python scripts/kbench_sell.py 100000 50 --o 3 --slabs 4 --gather 0 --no-csr; done
 for A in 1 2 4; do echo "ABLATE=$A gather 1"; XM_SELL_ABLATE=$A timeout 300 python scripts/kbench_sell.py 100000 50 --o 3 --slabs 4 --gather 1 --no-csr; done
 for LM in 8 16 32; do echo "lmax=$LM"; timeout 300 python scripts/kbench_sell.py 100000 50 --o 3 --slabs 4 --gather 0 1 --no-csr --lmax $LM; done
 echo banded; for LM in 8 16; do timeout 300 python scripts/kbench_sell.py 100000 50 --band --o 3 --slabs 1 4 --gather 0 --no-csr --lmax $LM; done
) 2>&1 | tee gpurun_out/b_ablate.log
