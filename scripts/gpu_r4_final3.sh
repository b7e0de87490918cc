#!/bin/bash
# Re-stamp after the last source change (second sliced-ELL launch reads W at its native pitch): the sliced-ELL / view-graph tests, PMC traffic +
# traced durations of every bench leg, the bench lines, kernel stats, the sliced-ELL micro-benchmark.  The whole-suite, smoke and 10x multi-rank
# records of scripts/gpu_r4_final2.sh stand (those paths are untouched).
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 -k "sell or viewgraph or vg100k or padded or reproducible or xm2" 2>&1 | tail -3 | cut -c1-300 | tee gpurun_out/r04_pytest_sell_restamp.txt
timeout 3000 python scripts/pmc_legs.py r04 venice hbm13682 rome_dense vg100k_vg vg100k_bsr > gpurun_out/r04_pmc_legs.out 2>&1; tail -2 gpurun_out/r04_pmc_legs.out
mkdir -p profiles; cp gpurun_out/r04_pmc_fetch_*.json profiles/ 2>/dev/null
timeout 900 python bench.py --steps 6 --warmup 1 2>&1 | tail -1 | tee gpurun_out/r04_bench_venice1778.json | cut -c1-300
timeout 600 python bench.py --workload vg100k --storage vg --steps 6 --warmup 1 --no-rome --no-hbm-check --cpu-seconds 10 2>&1 | tail -1 > gpurun_out/r04_bench_vg100k_vg.json
timeout 600 python bench.py --workload vg100k --storage bsr --steps 6 --warmup 1 --no-rome --no-hbm-check --cpu-seconds 0 2>&1 | tail -1 > gpurun_out/r04_bench_vg100k_bsr.json
XM_WATCHDOG_S=60 timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --cpu-seconds 0 2>&1 | tail -1 > gpurun_out/r04_bench_2gpu_virtual.json
XM_WATCHDOG_S=60 timeout 900 python bench.py --gpus 8 --steps 2 --warmup 1 --no-rome-dense --cpu-seconds 0 2>&1 | tail -1 > gpurun_out/r04_bench_8gpu_virtual.json
rm -rf gpurun_out/prof_final gpurun_out/prof_vg100k
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -o run -- python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-hbm-check --no-rome > $R/gpurun_out/prof_final.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_vg100k -o run -- python $R/bench.py --workload vg100k --storage vg --steps 2 --warmup 1 --no-rome --no-hbm-check --cpu-seconds 0 > $R/gpurun_out/prof_vg100k.log 2>&1
cd $R
cp $(ls gpurun_out/prof_final/*kernel_stats.csv 2>/dev/null | head -1) gpurun_out/r04_kernel_stats_bench_venice1778.csv 2>/dev/null
cp $(ls gpurun_out/prof_vg100k/*kernel_stats.csv 2>/dev/null | head -1) gpurun_out/r04_kernel_stats_bench_vg100k_vg.csv 2>/dev/null
rm -rf gpurun_out/prof_final gpurun_out/prof_vg100k
(python scripts/kbench_sell.py 100000 50 --o 3 4 5 --slabs 4 --gather 1 --codec 0 1 --layout 1 --no-csr
 python scripts/kbench_sell.py 100000 50 --o 3 4 5 --slabs 4 --gather 1 --codec 0 1 --layout 1 --no-csr --padded) 2>&1 | grep -v "^$\|amdgpu.ids" | tee gpurun_out/r04_kbench_sell_final.txt | tail -3
