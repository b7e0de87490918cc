#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -rf 2>&1 | tail -25 | cut -c1-300 | tee gpurun_out/r4g_pytest_gpu.log
XM_SCHUR_TRACE=1 timeout 600 python scripts/kbench_schur.py 13682 800000 8 --product-only 2>&1 | grep -v amdgpu.ids | tail -20 | tee gpurun_out/r4g_schur_setup.log
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_schur -o run -- python $R/scripts/kbench_schur.py 13682 800000 8 --product-only > $R/gpurun_out/r4g_prof_schur.log 2>&1
cd $R; f=$(ls gpurun_out/prof_schur/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-200 | tee gpurun_out/r4g_schur_kernel_stats.txt; rm -rf gpurun_out/prof_schur
