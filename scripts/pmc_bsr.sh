#!/bin/bash
# PMC passes over the BSR3 product micro-benchmark (separate runs per counter group, no tracing domains)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --list-avail > $R/gpurun_out/pmc_avail.txt 2>&1
i=0
for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TA_BUSY_avr GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_bsr_$i -o run -- python $R/scripts/kbench_bsr.py ${1:-100000} ${2:-50} ${3:-3} > $R/gpurun_out/pmc_bsr_$i.log 2>&1
  echo "group $i ($grp): rc=$?"
done
cd $R
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_bsr_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "qw_bsr3" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(d, k, "launches", len(v), "avg", sum(v) / len(v))
PY
