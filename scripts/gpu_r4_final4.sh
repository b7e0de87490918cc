#!/bin/bash
# After the stale-segment protection of the IPC rendezvous (xm_comm.hip): ten consecutive runs of the multi-rank tests again, then the
# re-stamp of scripts/gpu_r4_final3.sh (sliced-ELL tests, PMC legs, bench lines, kernel stats, sliced-ELL micro-benchmark).
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
: > gpurun_out/r04_pytest_multi_x10.txt
for i in 1 2 3 4 5 6 7 8 9 10; do
  echo "run $i: $(timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -k 'ipc or virtual or two_ranks' 2>&1 | tail -1)" | tee -a gpurun_out/r04_pytest_multi_x10.txt
done
bash scripts/gpu_r4_final3.sh
