#!/bin/bash
# after the last source change: whole GPU suite, smoke, PMC pass that stamps the Hessian traffic with the source hash, default bench line
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 1700 python -m pytest tests -m gpu -q --timeout 900 -rf 2>&1 | tail -8 | tee gpurun_out/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' 2>&1 | tail -3 | tee gpurun_out/smoke.log
XM_PROFILE_TAG=r03 bash scripts/pmc_hess.sh > gpurun_out/pmc_hess.out 2>&1
cp profiles/r03_pmc_fetch_hess_bench.json gpurun_out/ 2>/dev/null
timeout 900 python bench.py --steps 6 --warmup 1 2>&1 | tail -1 | tee gpurun_out/bench.log | cut -c1-300
