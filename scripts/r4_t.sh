#!/bin/bash
# Round 4, pass t: partial results at a 128-byte pitch (XM_SELL_PPITCH=16): parity, kbench, kernel trace, in-solve bench
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
XM_SELL_PPITCH=16 timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "qw_sell or padded_product or solve_through_sell" 2>&1 | tail -3 | tee gpurun_out/r4t_pytest.log
K="python scripts/kbench_sell.py 100000 50 --layout 1 --no-csr --slabs 4 --gather 1 --padded"
(timeout 300 $K --o 3 4 5 --codec 1 0
 echo XM_SELL_PPITCH=16; XM_SELL_PPITCH=16 timeout 300 $K --o 3 4 5 --codec 1 0) 2>&1 | grep -v "^$\|amdgpu.ids" | tee gpurun_out/r4t_kbench.log
for pp in 0 16; do
XM_SELL_PPITCH=$pp timeout 600 python bench.py --workload vg100k --storage vg --steps 6 --warmup 1 --no-rome --no-hbm-check --cpu-seconds 0 2>&1 | tail -1 > gpurun_out/r4t_bench_vg100k_vg_pp$pp.json
XM_SELL_PPITCH=$pp timeout 600 python bench.py --workload vg100k --storage bsr --steps 6 --warmup 1 --no-rome --no-hbm-check --cpu-seconds 0 2>&1 | tail -1 > gpurun_out/r4t_bench_vg100k_bsr_pp$pp.json
done
python - <<'PY' | tee gpurun_out/r4t_summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r4t_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-50s it/s %.0f  ms/solve %.1f  frac %.3f  launch_us %.2f  status %s rank %s" % (f[11:], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"] * 1e3, d["solve"]["status"], d["solve"]["rank"]))
    except Exception as e:
        print(f, "unreadable:", e, open(f).read()[-400:])
PY
cd /tmp; : > $R/gpurun_out/r4t_stats.txt
for pp in 0 16; do
  rm -rf $R/gpurun_out/prof_t
  XM_SELL_PPITCH=$pp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_t -o run -- python $R/scripts/kbench_sell.py 100000 50 --codec 1 --layout 1 --no-csr --slabs 4 --gather 1 --o 3 4 --reps 50 --padded > $R/gpurun_out/prof_t.log 2>&1
  echo "XM_SELL_PPITCH=$pp" >> $R/gpurun_out/r4t_stats.txt
  grep -h "qw_sell\|sell_reduce" $R/gpurun_out/prof_t/*kernel_stats.csv | cut -c1-60,150-220 >> $R/gpurun_out/r4t_stats.txt
done
rm -rf $R/gpurun_out/prof_t; cat $R/gpurun_out/r4t_stats.txt
