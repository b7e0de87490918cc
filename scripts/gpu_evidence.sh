#!/bin/bash
# Round-end evidence: parity tests, default bench line, rocprofv3 kernel stats of the bench command, kernel micro-benchmarks.
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -5 | tee gpurun_out/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' 2>&1 | tail -2 | tee gpurun_out/smoke.log
timeout 600 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench.log
timeout 300 python bench.py --workload vg100k --storage bsr --steps 5 --warmup 1 --cpu-seconds 10 2>&1 | tail -1 > gpurun_out/bench_vg100k.log
rm -rf gpurun_out/prof_final
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_final -o run -- python $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 --no-hbm-check > $GRAFT_REPO_ROOT/gpurun_out/prof_final.log 2>&1
cd $GRAFT_REPO_ROOT
(python scripts/kbench_dense.py 1778 3 4 5 10; python scripts/kbench_dense.py 13682 3; python scripts/kbench_bsr.py 13682 30 3 5; python scripts/kbench_bsr.py 13682 58 3 5; python scripts/kbench_bsr.py 100000 50 3 5; echo 'banded view graph (XM_KB_BAND=1):'; XM_KB_BAND=1 python scripts/kbench_bsr.py 100000 50 3 5) 2>&1 | tee gpurun_out/kbench.log
