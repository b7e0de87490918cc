#!/bin/bash
# Round-4 evidence on the FINAL sources (after the padded product input of the tCG): whole GPU suite + smoke, ten consecutive runs of the
# multi-rank tests, PMC traffic + traced durations of every bench leg (stamped with the source hash), the bench lines, rocprofv3 kernel
# stats, micro-benchmarks.  scripts/collect_profiles_r4.py copies the results into profiles/.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -rf 2>&1 | tail -8 | cut -c1-300 | tee gpurun_out/r04_pytest_gpu.txt
timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' 2>&1 | tail -3 | tee gpurun_out/r04_smoke.txt
: > gpurun_out/r04_pytest_multi_x10.txt
for i in 1 2 3 4 5 6 7 8 9 10; do
  echo "run $i: $(timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -k 'ipc or virtual or two_ranks' 2>&1 | tail -1)" | tee -a gpurun_out/r04_pytest_multi_x10.txt
done
timeout 3000 python scripts/pmc_legs.py r04 venice hbm13682 rome_dense vg100k_vg vg100k_bsr > gpurun_out/r04_pmc_legs.out 2>&1; tail -2 gpurun_out/r04_pmc_legs.out
mkdir -p profiles; cp gpurun_out/r04_pmc_fetch_*.json profiles/ 2>/dev/null     # bench.py quotes them (same box, same sources)
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
cp $(ls gpurun_out/prof_final/*kernel_stats.csv gpurun_out/prof_final/*/*kernel_stats.csv 2>/dev/null | head -1) gpurun_out/r04_kernel_stats_bench_venice1778.csv 2>/dev/null
cp $(ls gpurun_out/prof_vg100k/*kernel_stats.csv gpurun_out/prof_vg100k/*/*kernel_stats.csv 2>/dev/null | head -1) gpurun_out/r04_kernel_stats_bench_vg100k_vg.csv 2>/dev/null
rm -rf gpurun_out/prof_final gpurun_out/prof_vg100k
(python scripts/kbench_dense.py 1778 3 4 5 10; python scripts/kbench_dense.py 13682 3 4
 python scripts/kbench_sell.py 100000 50 --o 3 4 5 --slabs 4 --gather 1 --codec 0 1 --layout 1 --no-csr
 python scripts/kbench_sell.py 100000 50 --o 3 4 5 --slabs 4 --gather 1 --codec 0 1 --layout 1 --no-csr --padded
 python scripts/kbench_symw.py 13682 --o 3 --worlds 2 4 8
 python scripts/kbench_retract.py 1778 13682 100000
 python scripts/kbench_multi.py 1778 --o 3) 2>&1 | grep -v "^$\|amdgpu.ids" | tee gpurun_out/r04_kbench.txt | tail -5
