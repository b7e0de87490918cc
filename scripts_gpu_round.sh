#!/bin/bash
# one GPU session: parity tests, bench line, rocprof kernel stats.  Outputs under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "$1" != "noprof" ]; then
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/prof -type f | head; 
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); head -30 "$f"
fi
if [ "$1" != "notest" ]; then
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 | tee gpurun_out/pytest_gpu.log
fi
timeout 600 python bench.py --steps 5 --warmup 1 2>&1 | tail -5 | tee gpurun_out/bench.log
