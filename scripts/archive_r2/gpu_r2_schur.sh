#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_schur -o run -- python $GRAFT_REPO_ROOT/scripts/kbench_schur.py 13682 800000 8 --product-only > $GRAFT_REPO_ROOT/gpurun_out/prof_schur.log 2>&1
cd $GRAFT_REPO_ROOT
tail -1 gpurun_out/prof_schur.log
f=$(find gpurun_out/prof_schur -name "*kernel_stats.csv" | head -1); python - <<PY | tee gpurun_out/schur_kernels2.log
import csv
for r in list(csv.DictReader(open("$f")))[:12]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), f'{float(r["TotalDurationNs"])/1e6:9.1f} ms', f'{float(r["AverageNs"])/1e3:10.1f} us')
PY
