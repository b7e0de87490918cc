#!/bin/bash
# Round-3 evidence (run on the GPU box through gpurun; results land in gpurun_out/ and are copied into profiles/ by
# scripts/collect_profiles_r3.py): the whole GPU suite, smoke, PMC traffic of the Hessian launches (stamped with the source hash),
# the default bench line, the 100k-camera bench lines in both storages, rocprofv3 kernel stats, kernel micro-benchmarks.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 1700 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -6 | tee gpurun_out/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' 2>&1 | tail -3 | tee gpurun_out/smoke.log
XM_PROFILE_TAG=r03 bash scripts/pmc_hess.sh > gpurun_out/pmc_hess.out 2>&1
cp profiles/r03_pmc_fetch_hess_bench.json gpurun_out/ 2>/dev/null
timeout 900 python bench.py --steps 6 --warmup 1 2>&1 | tail -1 | tee gpurun_out/bench.log | cut -c1-400
timeout 600 python bench.py --workload vg100k --storage vg --steps 6 --warmup 1 --no-rome --no-hbm-check --cpu-seconds 10 2>&1 | tail -1 > gpurun_out/bench_vg100k_vg.log
timeout 600 python bench.py --workload vg100k --storage bsr --steps 6 --warmup 1 --no-rome --no-hbm-check --cpu-seconds 0 2>&1 | tail -1 > gpurun_out/bench_vg100k_bsr.log
XM_WATCHDOG_S=60 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-rome-dense --cpu-seconds 0 2>&1 | tail -1 > gpurun_out/bench_2gpu_virtual.log
rm -rf gpurun_out/prof_final gpurun_out/prof_vg100k
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -o run -- python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-hbm-check --no-rome > $R/gpurun_out/prof_final.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_vg100k -o run -- python $R/bench.py --workload vg100k --storage vg --steps 2 --warmup 1 --no-rome --no-hbm-check --cpu-seconds 0 > $R/gpurun_out/prof_vg100k.log 2>&1
cd $R
(python scripts/kbench_dense.py 1778 3 4 5 10; python scripts/kbench_dense.py 13682 3 4
 python scripts/kbench_sell.py 100000 50 --o 3 4 5 --slabs 4 --gather 1 --codec 0 1 --no-csr
 echo 'banded view graph:'; python scripts/kbench_sell.py 100000 50 --band --o 3 --slabs 4 --gather 1 --codec 0 1 --no-csr
 echo 'hub cameras:'; python scripts/kbench_sell.py 100000 50 --skew --o 3 --slabs 4 --gather 1 --codec 0 1 --no-csr
 python scripts/kbench_multi.py 1778 --o 3) 2>&1 | grep -v "^$" | tee gpurun_out/kbench.log
