#!/bin/bash
# Round 4, pass m: gather mode 4 (aligned sector windows into registers: two cache accesses per record) -- parity, timing, access counters
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests -m gpu -q -x --timeout 600 -k "quaternion_codec or viewgraph" 2>&1 | tail -3 | tee gpurun_out/r4m_pytest.log
(timeout 300 python scripts/kbench_sell.py 100000 50 --o 3 --slabs 4 --gather 1 4 2 --codec 1 --layout 1 --no-csr
 timeout 300 python scripts/kbench_sell.py 100000 50 --o 3 --slabs 4 --gather 1 4 --codec 1 --layout 1 --no-csr --band) 2>&1 | grep -v "^$\|amdgpu.ids" | tee gpurun_out/r4m_kbench.log
cd /tmp
for gm in 1 4 2; do
timeout 200 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_m_$gm -o run -- python $R/scripts/kbench_sell.py 100000 50 --o 3 --slabs 4 --gather $gm --codec 1 --layout 1 --no-csr --reps 20 > $R/gpurun_out/pmc_m_$gm.log 2>&1
done
cd $R
python - <<'PY' | tee gpurun_out/r4m_pmc.txt
import csv, glob, collections
for gm in (1, 4, 2):
    acc = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/pmc_m_%d/**/*counter_collection.csv" % gm, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "qw_sell" in k: acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()): print("gather %d  %-50s %-36s %14.1f" % (gm, k[:50], c, sum(v) / len(v)))
PY
rm -rf gpurun_out/pmc_m_*/
