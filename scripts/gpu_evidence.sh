#!/bin/bash
# Round-end evidence (run on the GPU box through gpurun; results land in gpurun_out/ and are copied to profiles/ by
# scripts/collect_profiles.py <tag>): parity tests, smoke, the default bench line, the 100k-camera bench line, rocprofv3 kernel
# stats of both bench commands, PMC traffic of the Hessian launches (stamped with the source hash) and of the sliced-ELL product,
# kernel micro-benchmarks.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee gpurun_out/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' 2>&1 | tail -2 | tee gpurun_out/smoke.log
# the PMC pass first: it stamps profiles/<tag>_pmc_fetch_hess_bench.json with the hash of the sources, and the bench line below quotes it
XM_PROFILE_TAG=${1:-r02} bash scripts/pmc_hess.sh > gpurun_out/pmc_hess.out 2>&1
cp profiles/${1:-r02}_pmc_fetch_hess_bench.json gpurun_out/ 2>/dev/null
timeout 900 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench.log
timeout 600 python bench.py --workload vg100k --storage bsr --steps 5 --warmup 1 --cpu-seconds 10 2>&1 | tail -1 > gpurun_out/bench_vg100k.log
rm -rf gpurun_out/prof_final gpurun_out/prof_vg100k
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -o run -- python $R/bench.py --cpu-seconds 0 --no-hbm-check > $R/gpurun_out/prof_final.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_vg100k -o run -- python $R/bench.py --workload vg100k --storage bsr --steps 2 --warmup 1 --cpu-seconds 0 > $R/gpurun_out/prof_vg100k.log 2>&1
cd $R
(python scripts/kbench_dense.py 1778 3 4 5 10; python scripts/kbench_dense.py 4096 3 4; python scripts/kbench_dense.py 13682 3 4
 echo 'matrix-free chain:'; python scripts/kbench_schur.py 1778 200000 6; python scripts/kbench_schur.py 6000 350000 8
 python scripts/kbench_bsr.py 13682 30 3 5; python scripts/kbench_bsr.py 13682 58 3 5
 python scripts/kbench_sell.py 100000 50 --o 3 5 --slabs 4 --gather 1
 echo 'banded view graph:'; python scripts/kbench_sell.py 100000 50 --band --o 3 --slabs 4 --gather 1
 echo 'hub cameras (skewed degrees):'; python scripts/kbench_sell.py 100000 20 --skew --o 3 --slabs 4 --gather 1) 2>&1 | tee gpurun_out/kbench.log
cd /tmp
i=0
for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_sell_$i -o run -- python $R/scripts/kbench_sell.py 100000 50 --o 3 --slabs 4 --gather 1 --no-csr --reps 20 > $R/gpurun_out/pmc_sell_$i.log 2>&1
done
for grp in "FETCH_SIZE" "WRITE_SIZE"; do   # HBM-side traffic of the half-traffic symmetric product on the 13.5 GB matrix
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_symv_$grp -o run -- python $R/scripts/kbench_dense.py 13682 3 > /dev/null 2>&1
done
cd $R
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
            out[k][c + ("_bytes_x1024x2" if c == "FETCH_SIZE" else "_bytes_x1024")] = sum(v) / len(v) * 1024 * (2 if c == "FETCH_SIZE" else 1)
out["note"] = "kbench_dense.py 13682 3 (13.5 GB matrix, half matrix 6.74 GB); FETCH_SIZE in KB, x2 on gfx950 (MI355X_MICROARCH.md HBM section); WRITE_SIZE in KB"
json.dump(out, open("gpurun_out/pmc_symv.json", "w"), indent=1)
PY
cd $R
python - <<'PY'
import csv, glob, collections, json
out = collections.defaultdict(dict)
for d in sorted(glob.glob("gpurun_out/pmc_sell_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "sell" in k and "fill" not in k:
                acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in acc.items():
            out[k][c] = sum(v) / len(v)
json.dump(out, open("gpurun_out/pmc_sell.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
