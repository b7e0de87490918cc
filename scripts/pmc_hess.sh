#!/bin/bash
# HBM-side traffic of the Hessian Q*W launches INSIDE the bench command (own PMC pass, no tracing domains)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_hess
cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_hess -o run -- python $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-hbm-check --no-rome > $R/gpurun_out/pmc_hess.log 2>&1
echo "rc=$?"
cd $R
python - <<'PY'
import csv, glob, json, collections
f = glob.glob("gpurun_out/pmc_hess/**/*counter_collection.csv", recursive=True)[0]
per = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "qw_dense_kernel" in k and r["Counter_Name"] == "FETCH_SIZE":
        per[k.split("(")[0]].append(float(r["Counter_Value"]))
out = {}
for k, v in per.items():
    real = [x for x in v if x > 0.25 * max(v)]          # enqueued-ahead no-op launches fetch (almost) nothing
    out[k] = {"launches": len(v), "real_launches": len(real), "FETCH_SIZE_KB_avg_real": sum(real) / len(real),
              "hbm_side_bytes_per_real_launch": sum(real) / len(real) * 1024 * 2}
hess = {k: v for k, v in out.items() if ", 2, 2, " in k}          # EPI_HESS launches of every rank level, weighted by launch count
tot = sum(v["real_launches"] for v in hess.values())
out["hess_all_ranks_weighted"] = {"real_launches": tot, "hbm_side_bytes_per_real_launch":
                                  sum(v["real_launches"] * v["hbm_side_bytes_per_real_launch"] for v in hess.values()) / max(tot, 1)}
import sys, os
sys.path.insert(0, ".")
import bench
out["source_sha256"] = bench.source_sha256()   # bench.py quotes this profile only while the sources are the ones it was measured on
out["correction"] = "x1024 (KB) x2 (gfx950: 128-byte requests tallied at 64 B, MI355X_MICROARCH.md HBM section)"
out["command"] = "rocprofv3 --pmc FETCH_SIZE -- python bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-hbm-check --no-rome"
json.dump(out, open("gpurun_out/pmc_hess.json", "w"), indent=1)
tag = os.environ.get("XM_PROFILE_TAG", "r02")
json.dump(out, open(f"profiles/{tag}_pmc_fetch_hess_bench.json", "w"), indent=1)   # profiles/ on the GPU box is scratch: copy it back from gpurun_out
print(json.dumps(out, indent=1))
PY
