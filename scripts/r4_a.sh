#!/bin/bash
# Round 4, first GPU pass: the chunk-tiled one-launch sliced-ELL product (xm_sell2.hip) -- parity tests, micro-benchmarks against the
# two-launch layout, and the 100k-camera solve in both layouts.  Run through gpurun; results land in gpurun_out/.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "sell or viewgraph or vg100k or reproducible or codec" 2>&1 | tail -15 | tee gpurun_out/r4a_pytest.log
(XM_SELL2_PIPE=0 timeout 600 python scripts/kbench_sell.py 100000 50 --o 3 4 5 --slabs 4 --gather 1 --codec 0 1 --layout 1 2 --no-csr
 XM_SELL2_PIPE=1 timeout 600 python scripts/kbench_sell.py 100000 50 --o 3 4 5 --slabs 4 --gather 1 --codec 0 1 --layout 2 --no-csr
 echo 'o = 1 (Lanczos):'; timeout 300 python scripts/kbench_sell.py 100000 50 --o 1 --slabs 4 --gather 0 --codec 1 --layout 1 2 --no-csr
 echo 'banded view graph:'; timeout 300 python scripts/kbench_sell.py 100000 50 --band --o 3 --slabs 4 --gather 1 --codec 1 --layout 1 2 --no-csr
 echo 'hub cameras:'; timeout 300 python scripts/kbench_sell.py 100000 50 --skew --o 3 --slabs 4 --gather 1 --codec 1 --layout 1 2 --no-csr) 2>&1 | grep -v "^$" | tee gpurun_out/r4a_kbench.log
timeout 600 python bench.py --workload vg100k --storage vg --steps 6 --warmup 1 --no-rome --no-hbm-check --cpu-seconds 0 2>&1 | tail -1 > gpurun_out/r4a_bench_vg100k_vg.log
XM_SELL2_PIPE=1 timeout 600 python bench.py --workload vg100k --storage vg --steps 6 --warmup 1 --no-rome --no-hbm-check --cpu-seconds 0 2>&1 | tail -1 > gpurun_out/r4a_bench_vg100k_vg_pipe1.log
XM_SELL_LAYOUT=1 timeout 600 python bench.py --workload vg100k --storage vg --steps 6 --warmup 1 --no-rome --no-hbm-check --cpu-seconds 0 2>&1 | tail -1 > gpurun_out/r4a_bench_vg100k_vg_layout1.log
timeout 600 python bench.py --workload vg100k --storage bsr --steps 6 --warmup 1 --no-rome --no-hbm-check --cpu-seconds 0 2>&1 | tail -1 > gpurun_out/r4a_bench_vg100k_bsr.log
for f in gpurun_out/r4a_bench_*.log; do echo $f; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("  it/s %.0f  ms/solve %.1f  iters %s  roofline %.3f  launch_ms %.4f  status %s rank %s" % (d["value"], d["ms_per_step"], d["solve"]["tcg_iters_by_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["solve"]["status"], d["solve"]["rank"]))
except Exception as e:
    print("  unreadable:", e, open(sys.argv[1]).read()[-600:])
PY
done
