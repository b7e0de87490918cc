#!/bin/bash
# full GPU test-suite + smoke (what the driver runs at round end)
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1700 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' 2>&1 | tail -2 | tee gpurun_out/smoke.log
