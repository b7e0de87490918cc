#!/bin/bash
# second diagnostic PMC set over the sliced-ELL product: the vector-memory path behind the wave-level wait cycles (address unit, L1 miss
# queues, L2 tag pipeline, fabric requests, address translation).  One counter group per pass, kernel-trace only.
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; export TMPDIR=/tmp; cd /tmp
ARGS="${SELL_ARGS:-100000 50 --o 3 --slabs 4 --gather 1 --codec 1 --no-csr --reps 20}"
OUT="${SELL_OUT:-pmc_sell_diag2.json}"
i=0
for grp in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TC_STALL_sum TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" \
           "TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_TCR_RDRET_STALL_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCC_REQ_sum TCC_TAG_STALL_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCC_BUSY_sum TCC_CYCLE_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TD_LOAD_WAVEFRONT_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
           "TCC_LATENCY_FIFO_FULL_sum TCC_SRC_FIFO_FULL_sum TCC_IB_STALL_sum TCC_READ_SECTORS_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_diag2_$i -o run -- python $R/scripts/kbench_sell.py $ARGS > $R/gpurun_out/pmc_diag2_$i.log 2>&1
  echo "group $i rc $?"
done
cd $R
python - "$OUT" <<'PY'
import csv, glob, collections, json, sys
out = collections.defaultdict(dict)
for d in sorted(glob.glob("gpurun_out/pmc_diag2_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "sell" in k and "fill" not in k and "diag" not in k:
                acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in acc.items():
            out[k][c] = sum(v) / len(v)
json.dump(out, open("gpurun_out/" + sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf gpurun_out/pmc_diag2_*/
