#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(echo "row-major slots restored"; timeout 300 python scripts/kbench_sell.py 100000 50 --o 3 --slabs 1 4 --gather 0 1 --no-csr --lmax 128
 for A in 4; do echo "ABLATE=$A S=1 lmax 128 (no partial store): fused-kernel estimate = this - reduce kernel"; XM_SELL_ABLATE=$A timeout 300 python scripts/kbench_sell.py 100000 50 --o 3 --slabs 1 --gather 0 1 --no-csr --lmax 128; done
 cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/f_trace -o run -- python $GRAFT_REPO_ROOT/scripts/kbench_sell.py 100000 50 --o 3 --slabs 1 --gather 1 --no-csr --lmax 128 > /dev/null 2>&1
 cd $GRAFT_REPO_ROOT; python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/f_trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sell" in r["Name"]: print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
) 2>&1 | tee gpurun_out/f_s1.log
