#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -rf -k "matrix_free or spd or xm2" 2>&1 | tail -30 | cut -c1-300 | tee gpurun_out/r4j_pytest_schur.log
XM_SCHUR_TRACE=1 timeout 600 python scripts/kbench_schur.py 13682 800000 8 --product-only 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/r4j_schur_setup.log
