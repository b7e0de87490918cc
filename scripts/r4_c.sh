#!/bin/bash
# Round 4, third GPU pass: gather mode 2 of the sliced-ELL product (sector windows through LDS-DMA) -- parity, micro-benchmark, solve;
# the multi-rank tests after the residency fix (hard gates), RCCL's answer to two ranks on one GPU.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "sell or viewgraph or vg100k or reproducible or codec" 2>&1 | tail -6 | tee gpurun_out/r4c_pytest_sell.log
(for gm in 1 2; do timeout 600 python scripts/kbench_sell.py 100000 50 --o 3 --slabs 4 --gather $gm --codec 0 1 --layout 1 --no-csr; done
 echo 'banded view graph:'; timeout 300 python scripts/kbench_sell.py 100000 50 --band --o 3 --slabs 4 --gather 1 2 --codec 1 --layout 1 --no-csr
 echo 'hub cameras:'; timeout 300 python scripts/kbench_sell.py 100000 50 --skew --o 3 --slabs 4 --gather 1 2 --codec 1 --layout 1 --no-csr) 2>&1 | grep -v "^$" | tee gpurun_out/r4c_kbench.log
for gm in 1 2; do XM_SELL_GATHER=$gm timeout 600 python bench.py --workload vg100k --storage vg --steps 6 --warmup 1 --no-rome --no-hbm-check --cpu-seconds 0 2>&1 | tail -1 > gpurun_out/r4c_bench_vg100k_vg_gm$gm.log; done
XM_SELL_GATHER=2 timeout 600 python bench.py --workload vg100k --storage bsr --steps 6 --warmup 1 --no-rome --no-hbm-check --cpu-seconds 0 2>&1 | tail -1 > gpurun_out/r4c_bench_vg100k_bsr_gm2.log
for f in gpurun_out/r4c_bench_*.log; do echo $f; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("  it/s %.0f  ms/solve %.1f  iters %s  roofline %.3f  launch_ms %.4f  status %s rank %s" % (d["value"], d["ms_per_step"], d["solve"]["tcg_iters_by_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["solve"]["status"], d["solve"]["rank"]))
except Exception as e:
    print("  unreadable:", e, open(sys.argv[1]).read()[-600:])
PY
done
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 -k "ipc or virtual or two_ranks or rccl or team or multi_gpu or plain_command or eight" -s 2>&1 | grep -E "RCCL 2 ranks|passed|failed|FAILED|Error|error" | tail -30 | tee gpurun_out/r4c_pytest_multi.log
