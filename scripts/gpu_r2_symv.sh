#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "symmetr or sym or certificate or lanczos or golden" 2>&1 | tail -4
(echo "rome dense"; timeout 900 python bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-hbm-check | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['rome_scale_dense'])"
 timeout 300 python scripts/kbench_dense.py 13682 3) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/symv5.log
cd /tmp
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_symv_$grp -o run -- python $GRAFT_REPO_ROOT/scripts/kbench_dense.py 13682 3 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections, json
out = collections.defaultdict(dict)
for d in sorted(glob.glob("gpurun_out/pmc_symv_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "symv" in k or "qw_dense_kernel" in k:
                acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in acc.items():
            out[k][c] = sum(v) / len(v)
            out[k][c + "_bytes_x1024x2" if c == "FETCH_SIZE" else c + "_bytes_x1024"] = sum(v) / len(v) * 1024 * (2 if c == "FETCH_SIZE" else 1)
out["note"] = "kbench_dense.py 13682 3 (13.5 GB matrix, half matrix 6.74 GB); FETCH_SIZE in KB, x2 on gfx950 (MI355X_MICROARCH.md HBM section); WRITE_SIZE in KB"
json.dump(out, open("gpurun_out/pmc_symv.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
