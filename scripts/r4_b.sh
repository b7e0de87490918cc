#!/bin/bash
# Round 4, second GPU pass: chunk-tiled product after the tile hand-off fix (parity), write-through partial stores in the two-launch layout,
# and where the cycles of both main kernels go (one PMC group per pass).
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
python scripts/dbg_sell2.py 2>&1 | grep "bad rows" | tee gpurun_out/r4b_dbg.log
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "sell or viewgraph or vg100k or reproducible or codec" 2>&1 | tail -6 | tee gpurun_out/r4b_pytest.log
(timeout 600 python scripts/kbench_sell.py 100000 50 --o 3 --slabs 4 --gather 1 --codec 1 --layout 1 2 --no-csr
 XM_SELL_WT=1 timeout 600 python scripts/kbench_sell.py 100000 50 --o 3 4 --slabs 4 --gather 1 --codec 0 1 --layout 1 --no-csr) 2>&1 | grep -v "^$" | tee gpurun_out/r4b_kbench.log
timeout 600 python bench.py --workload vg100k --storage vg --steps 6 --warmup 1 --no-rome --no-hbm-check --cpu-seconds 0 2>&1 | tail -1 > gpurun_out/r4b_bench_vg100k_vg.log
XM_SELL_LAYOUT=1 XM_SELL_WT=1 timeout 600 python bench.py --workload vg100k --storage vg --steps 6 --warmup 1 --no-rome --no-hbm-check --cpu-seconds 0 2>&1 | tail -1 > gpurun_out/r4b_bench_vg100k_vg_layout1_wt.log
for f in gpurun_out/r4b_bench_*.log; do echo $f; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("  it/s %.0f  ms/solve %.1f  iters %s  roofline %.3f  launch_ms %.4f  status %s rank %s" % (d["value"], d["ms_per_step"], d["solve"]["tcg_iters_by_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["solve"]["status"], d["solve"]["rank"]))
except Exception as e:
    print("  unreadable:", e, open(sys.argv[1]).read()[-600:])
PY
done
for lay in 1 2; do
  SELL_ARGS="100000 50 --o 3 --slabs 4 --gather 1 --codec 1 --layout $lay --no-csr --reps 20" bash scripts/pmc_sell_diag.sh > gpurun_out/r4b_pmc_layout$lay.out 2>&1
  cp gpurun_out/pmc_sell_diag.json gpurun_out/r4b_pmc_sell_diag_layout$lay.json
  rm -rf gpurun_out/pmc_diag_*
done
