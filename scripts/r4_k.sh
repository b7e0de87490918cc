#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -rf -k "matrix_free_context_under" 2>&1 | tail -12 | cut -c1-300 | tee gpurun_out/r4k_pytest_schur_multi.log
