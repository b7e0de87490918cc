#!/bin/bash
# diagnostic PMC passes over the half-traffic symmetric product on the 13.5 GB matrix (one counter group per pass)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; export TMPDIR=/tmp; cd /tmp
i=0
for grp in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_symvd_$i -o run -- python $R/scripts/kbench_dense.py 13682 3 > $R/gpurun_out/pmc_symvd_$i.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections, json
out = collections.defaultdict(dict)
for d in sorted(glob.glob("gpurun_out/pmc_symvd_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "symv" in k or "qw_dense_kernel" in k:
                acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in acc.items():
            out[k][c] = sum(v) / len(v)
json.dump(out, open("gpurun_out/pmc_symv_diag.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
