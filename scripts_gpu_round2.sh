#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --durations=5 2>&1 | tail -22 | tee gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 5 --warmup 1 2>&1 | tail -1 | tee gpurun_out/bench.log
# PMC: HBM-side fetch bytes of the dense kernel (own pass, kernel-trace only)
cd /tmp
for n in 1778 13682; do
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$n -o run -- python $GRAFT_REPO_ROOT/scripts_kbench.py $n 3 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$n.log 2>&1
done
cd $GRAFT_REPO_ROOT; find gpurun_out/pmc_1778 gpurun_out/pmc_13682 -type f | head; 
python - <<'PY'
import csv, glob
for n in (1778, 13682):
    for f in glob.glob(f'gpurun_out/pmc_{n}/**/*counter_collection.csv', recursive=True):
        rows=[r for r in csv.DictReader(open(f)) if 'qw_dense' in r.get('Kernel_Name','')]
        if rows:
            v=[float(r['Counter_Value']) for r in rows if r['Counter_Name']=='FETCH_SIZE']
            print(n, f, 'launches', len(v), 'FETCH_SIZE avg', sum(v)/len(v), 'min', min(v), 'max', max(v))
PY
