#!/bin/bash
# Round 4, pass q: per-kernel durations (rocprofv3 kernel trace) of the sliced-ELL product at the native and at the 128-byte record pitch
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K="python $R/scripts/kbench_sell.py 100000 50 --codec 1 --layout 1 --no-csr --slabs 4 --gather 1 --o 3 4 --reps 50"
cd /tmp; : > $R/gpurun_out/r4q_stats.txt
for v in 0 16; do
  rm -rf $R/gpurun_out/prof_q
  XM_SELL_WSTRIDE=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_q -o run -- $K > $R/gpurun_out/prof_q.log 2>&1
  echo "XM_SELL_WSTRIDE=$v" >> $R/gpurun_out/r4q_stats.txt
  grep -h "qw_sell\|sell_reduce\|sell_pack" $R/gpurun_out/prof_q/*kernel_stats.csv | cut -c1-160 >> $R/gpurun_out/r4q_stats.txt
done
rm -rf $R/gpurun_out/prof_q; cat $R/gpurun_out/r4q_stats.txt
