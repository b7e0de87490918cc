#!/bin/bash
# diagnostic PMC passes over the sliced-ELL product (one counter group per pass, kernel-trace only): where do the cycles go?
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; export TMPDIR=/tmp; cd /tmp
ARGS="${SELL_ARGS:-100000 50 --o 3 --slabs 4 --gather 1 --no-csr --reps 20}"
i=0
for grp in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCR_TCP_STALL_CYCLES_sum TD_TD_BUSY_sum" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_diag_$i -o run -- python $R/scripts/kbench_sell.py $ARGS > $R/gpurun_out/pmc_diag_$i.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections, json
out = collections.defaultdict(dict)
for d in sorted(glob.glob("gpurun_out/pmc_diag_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "sell" in k and "fill" not in k:
                acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in acc.items():
            out[k][c] = sum(v) / len(v)
json.dump(out, open("gpurun_out/pmc_sell_diag.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
