#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sell" 2>&1 | tail -2
 for SL in 0 1; do echo "XM_SELL_SLOTS=$SL"; XM_SELL_SLOTS=$SL timeout 300 python scripts/kbench_sell.py 100000 50 --o 3 --slabs 4 8 --gather 0 1 --no-csr; XM_SELL_SLOTS=$SL XM_SELL_ABLATE=3 timeout 300 python scripts/kbench_sell.py 100000 50 --o 3 --slabs 4 --gather 0 --no-csr; done
 cd /tmp; XM_SELL_SLOTS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/i_trace -o run -- python $GRAFT_REPO_ROOT/scripts/kbench_sell.py 100000 50 --o 3 --slabs 4 --gather 1 --no-csr > /dev/null 2>&1
 cd $GRAFT_REPO_ROOT; python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/i_trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sell" in r["Name"]: print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
) 2>&1 | tee gpurun_out/i_slots.log
