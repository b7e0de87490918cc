#!/bin/bash
# Round-3 evidence, second half (after the IPC transport / write-through exchange / XM^2-on-team changes): whole GPU suite, smoke, PMC
# pass that stamps the Hessian traffic with the source hash, default bench line, 100k-camera line (view-graph storage), 2 virtual
# ranks, rocprofv3 kernel stats of the bench command, pieces of the partitioned iteration.  The kernel micro-benchmarks of
# gpu_r3_final.sh (dense / sliced ELL sweeps) are not repeated: those kernels did not change.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -rf 2>&1 | tail -8 | tee gpurun_out/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' 2>&1 | tail -3 | tee gpurun_out/smoke.log
XM_PROFILE_TAG=r03 bash scripts/pmc_hess.sh > gpurun_out/pmc_hess.out 2>&1
cp profiles/r03_pmc_fetch_hess_bench.json gpurun_out/ 2>/dev/null
timeout 600 python bench.py --steps 6 --warmup 1 2>&1 | tail -1 | tee gpurun_out/bench.log | cut -c1-400
timeout 300 python bench.py --workload vg100k --storage vg --steps 6 --warmup 1 --no-rome --no-hbm-check --cpu-seconds 10 2>&1 | tail -1 > gpurun_out/bench_vg100k_vg.log
XM_WATCHDOG_S=60 timeout 300 python bench.py --gpus 2 --steps 3 --warmup 1 --no-rome-dense --cpu-seconds 0 2>&1 | tail -1 > gpurun_out/bench_2gpu_virtual.log
rm -rf gpurun_out/prof_final
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -o run -- python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-hbm-check --no-rome > $R/gpurun_out/prof_final.log 2>&1
cd $R
python scripts/kbench_multi.py 1778 --o 3 2>&1 | grep -v "^$" | tee gpurun_out/kbench_multi.log
