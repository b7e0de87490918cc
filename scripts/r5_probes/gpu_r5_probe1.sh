#!/bin/bash
# round 5, first GPU call: the suite on the cleaned sources, the Final-13682 block-sparse iteration under the kernel trace,
# block-CSR kernel against sliced ELL by size (the cross-over behind Context's automatic choice), the upper-triangle timing experiment
# (VERDICT r4 #1), the recovery's two projection kernels (#8), the default bench line with the same-node KKT pair (#4)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -x 2>&1 | tail -40 ) > $O/r05_pytest_gpu_a.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/r05_smoke_a.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/rome_trace -o run -- python $R/bench.py --workload final13682 --storage bsr --steps 3 --warmup 1 --no-hbm-check --cpu-seconds 0 > $O/r05_bench_rome_bsr_traced.json 2> $O/r05_rome_trace.err
cd $R
f=$(ls $O/rome_trace/*/*kernel_trace.csv $O/rome_trace/*kernel_trace.csv 2>/dev/null | head -1)
python scripts/trace_summary.py $f 0.7 > $O/r05_trace_summary_rome_bsr.txt 2>&1
rm -rf $O/rome_trace
python bench.py --workload final13682 --storage bsr --steps 3 --warmup 1 --no-hbm-check --cpu-seconds 0 > $O/r05_bench_rome_bsr.json 2>&1
python bench.py --workload final13682 --storage bsr --sell on --steps 3 --warmup 1 --no-hbm-check --cpu-seconds 0 > $O/r05_bench_rome_bsr_sell.json 2>&1
for sz in "13682 30" "30000 30" "50000 40"; do
  python scripts/kbench_sell.py $sz --o 3 4 --codec 0 --padded >> $O/r05_kbench_sell_crossover.txt 2>&1
done
python scripts/kbench_sell.py 100000 50 --o 3 --codec 0 1 --padded --no-csr > $O/r05_kbench_sell_upper.txt 2>&1
python scripts/kbench_sell.py 100000 50 --o 3 --codec 0 1 --padded --no-csr --upper >> $O/r05_kbench_sell_upper.txt 2>&1
python scripts/kbench_sell.py 100000 50 --o 1 --codec 0 1 --gather 0 --no-csr >> $O/r05_kbench_sell_upper.txt 2>&1
python scripts/kbench_sell.py 100000 50 --o 1 --codec 0 1 --gather 0 --no-csr --upper >> $O/r05_kbench_sell_upper.txt 2>&1
python scripts/kbench_recover.py > $O/r05_kbench_recover.txt 2>&1
python bench.py > $O/r05_bench_venice1778_a.json 2> $O/r05_bench_venice1778_a.err
python bench.py --workload vg100k --storage vg --steps 3 --warmup 1 --no-rome --no-hbm-check > $O/r05_bench_vg100k_vg_a.json 2> $O/r05_bench_vg100k_vg_a.err
ls -la $O | tail -30
