#!/bin/bash
# round 3, first GPU pass: new tests, sliced-ELL codec micro-benchmarks, then the whole suite (regressions of the refactor)
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_round3.py -m gpu -q --timeout 900 2>&1 | tail -60 | tee gpurun_out/r3a_new.log
(for pipe in 0 1 2 3; do XM_SELL_PIPE=$pipe timeout 300 python scripts/kbench_sell.py 100000 50 --o 3 --slabs 4 --gather 1 --codec 0 1 --no-csr --check; done
 XM_SELL_PIPE=1 timeout 300 python scripts/kbench_sell.py 100000 50 --o 3 --slabs 1 2 8 --gather 1 --codec 1 --no-csr
 XM_SELL_PIPE=1 timeout 300 python scripts/kbench_sell.py 100000 50 --o 3 --slabs 4 --gather 0 --codec 1 --no-csr
 timeout 300 python scripts/kbench_sell.py 100000 50 --o 4 5 --slabs 4 --gather 1 --codec 0 1 --no-csr
 timeout 300 python scripts/kbench_sell.py 100000 50 --band --o 3 --slabs 4 --gather 1 --codec 0 1 --no-csr
 for abl in 1 2 4 6; do XM_SELL_ABLATE=$abl timeout 300 python scripts/kbench_sell.py 100000 50 --o 3 --slabs 4 --gather 1 --codec 1 --no-csr; done
) 2>&1 | grep -v "^$" | tee gpurun_out/r3a_kbench.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 900 2>&1 | tail -40 | tee gpurun_out/r3a_suite.log
