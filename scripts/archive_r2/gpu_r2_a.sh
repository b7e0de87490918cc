#!/bin/bash
# round-2 GPU call A: correctness + timing sweep of the sliced-ELL product, kernel split, PMC traffic
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sell or bsr3" 2>&1 | tail -8 | tee gpurun_out/a_pytest.log
(timeout 900 python scripts/kbench_sell.py 100000 50 --o 3 --slabs 1 2 4 8 --gather 0 1 --check
 timeout 600 python scripts/kbench_sell.py 100000 50 --o 5 --slabs 4 8 --gather 0 1
 timeout 600 python scripts/kbench_sell.py 100000 50 --band --o 3 --slabs 1 4 8 --gather 0 1
 timeout 600 python scripts/kbench_sell.py 13682 30 --o 3 --slabs 1 4 8 --gather 0
 timeout 600 python scripts/kbench_sell.py 100000 20 --skew --o 3 --slabs 4 --gather 0 --check) 2>&1 | tee gpurun_out/a_kbench.log
cd /tmp
for S in 4 8; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/a_trace_s$S -o run -- python $R/scripts/kbench_sell.py 100000 50 --o 3 --slabs $S --gather 0 --no-csr > $R/gpurun_out/a_trace_s$S.log 2>&1
done
i=0
for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TA_BUSY_avr GRBM_GUI_ACTIVE" "WRITE_SIZE"; do
  i=$((i+1))
  for S in 1 4 8; do
    timeout 300 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/a_pmc_s${S}_$i -o run -- python $R/scripts/kbench_sell.py 100000 50 --o 3 --slabs $S --gather 0 --no-csr --reps 20 > $R/gpurun_out/a_pmc_s${S}_$i.log 2>&1
    echo "pmc group $i ($grp) S=$S rc=$?"
  done
done
cd $R
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/a_pmc_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "sell" in k and "fill" not in k:
                acc[(k[:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print(d, k, "launches", len(v), "avg", sum(v) / len(v))
for d in sorted(glob.glob("gpurun_out/a_trace_*/")):
    for f in glob.glob(d + "**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "sell" in r["Name"]:
                print(d, r["Name"][:70], r["Calls"], r["AverageNs"])
PY
