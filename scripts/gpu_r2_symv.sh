#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for k in 4 6 8 12; do
  cd /tmp; XM_SYMV_K=$k timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_symv_k$k -o run -- python $GRAFT_REPO_ROOT/scripts/kbench_dense.py 1778 3 4 > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/prof_symv_k$k -name "*kernel_stats.csv" | head -1)
  echo "K=$k"; grep -E "symv" $f | awk -F'","' '{printf "  %-50s calls %s avg %.1f us\n", substr($1,2,50), $2, $4/1000}'
done 2>&1 | tee gpurun_out/symv4.log
