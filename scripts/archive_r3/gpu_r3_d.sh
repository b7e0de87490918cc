#!/bin/bash
# round 3, pass D: new tests, measured pieces of the partitioned iteration, symmetric path at Venice size, PMC + kernel stats of the
# compressed sliced-ELL product
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -m gpu -q --timeout 600 -rf -k "bench_plain or lanczos_on or mid700 or xm2_res or xm2_round or bench_two_ranks" 2>&1 | tail -25 | tee gpurun_out/r3d_tests.log
(timeout 300 python scripts/kbench_multi.py 1778 --o 3 5; timeout 600 python scripts/kbench_multi.py 13682 --o 3) 2>&1 | tee gpurun_out/r3d_kbench_multi.log
XM_SYM=1 timeout 600 python bench.py --steps 6 --warmup 1 --no-rome --no-hbm-check --cpu-seconds 0 > gpurun_out/r3d_bench_venice_sym.json 2> gpurun_out/r3d_bench_venice_sym.err; tail -c 300 gpurun_out/r3d_bench_venice_sym.json
(for pipe in 0 1; do XM_SELL_PIPE=$pipe timeout 300 python scripts/kbench_sell.py 100000 50 --o 4 5 --slabs 4 --gather 1 --codec 1 --no-csr; done
 timeout 300 python scripts/kbench_sell.py 100000 50 --skew --o 3 --slabs 4 --gather 1 --codec 0 1 --no-csr
 timeout 300 python scripts/kbench_sell.py 100000 50 --o 1 --slabs 4 --gather 0 --codec 0 1 --no-csr) 2>&1 | grep -v "^$" | tee gpurun_out/r3d_kbench_sell2.log
cd /tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_sellq_$i -o run -- python $R/scripts/kbench_sell.py 100000 50 --o 3 --slabs 4 --gather 1 --codec 1 --no-csr --reps 20 > $R/gpurun_out/pmc_sellq_$i.log 2>&1
  echo "group $i ($grp): rc=$?"
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_vg100k_vg -o run -- python $R/bench.py --workload vg100k --storage vg --steps 2 --warmup 1 --no-rome --no-hbm-check --cpu-seconds 0 > $R/gpurun_out/r3d_prof_vg.log 2>&1
cd $R
python - <<'PY' | tee gpurun_out/r3d_pmc_sellq.txt
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_sellq_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            name = "main" if "qw_sell_kernel" in k else "reduce" if "sell_reduce" in k else None
            if name:
                acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print(d, k, "launches", len(v), "avg", sum(v) / len(v))
for f in glob.glob("gpurun_out/prof_vg100k_vg/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:3000])
PY
