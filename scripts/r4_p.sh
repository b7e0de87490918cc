#!/bin/bash
# Round 4, pass p: ablations of the default sliced-ELL kernel (view-graph codec, o = 3): which part of the wave's work the duration follows
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
K="python scripts/kbench_sell.py 100000 50 --codec 1 --layout 1 --no-csr --slabs 4 --gather 1 --o 3"
(XM_SELL_PIPE=0 timeout 300 $K
 for a in 8 12; do echo "XM_SELL_ABLATE=$a"; XM_SELL_ABLATE=$a timeout 300 $K; done) 2>&1 | grep -v "^$\|amdgpu.ids" | tee gpurun_out/r4p_kbench.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_p -o run -- env XM_SELL_ABLATE=8 $K --reps 50 > $R/gpurun_out/prof_p.log 2>&1
grep -h "qw_sell\|sell_reduce" $R/gpurun_out/prof_p/*kernel_stats.csv | cut -c1-200 | tee $R/gpurun_out/r4p_stats.txt
rm -rf $R/gpurun_out/prof_p
