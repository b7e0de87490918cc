#!/bin/bash
# Round 4: whole GPU suite + smoke on the current sources
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -rf 2>&1 | tail -25 | tee gpurun_out/r4d_pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' 2>&1 | tail -3 | tee gpurun_out/r4d_smoke.log
