#!/bin/bash
# Round 4, pass r: the tCG's product input kept at a 128-byte record pitch (xm_tuning_t.sell_wpad): parity + the 100k bench lines with and without
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "padded_product or sell or viewgraph or vg100k" 2>&1 | tail -3 | tee gpurun_out/r4r_pytest.log
for w in 1 0; do for st in vg bsr; do
XM_SELL_WPAD=$w timeout 600 python bench.py --workload vg100k --storage $st --steps 6 --warmup 1 --no-rome --no-hbm-check --cpu-seconds 0 2>&1 | tail -1 > gpurun_out/r4r_bench_vg100k_${st}_wpad$w.json
done; done
python - <<'PY' | tee gpurun_out/r4r_summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r4r_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-50s it/s %.0f  ms/solve %.1f  iters %s  frac %.3f  launch_us %.2f  status %s rank %s" % (f[11:], d["value"], d["ms_per_step"], d["solve"]["tcg_iters_by_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"] * 1e3, d["solve"]["status"], d["solve"]["rank"]))
    except Exception as e:
        print(f, "unreadable:", e, open(f).read()[-400:])
PY
cd /tmp; rm -rf $R/gpurun_out/prof_r
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r -o run -- python $R/bench.py --workload vg100k --storage vg --steps 2 --warmup 1 --no-rome --no-hbm-check --cpu-seconds 0 > $R/gpurun_out/prof_r.log 2>&1
head -12 $R/gpurun_out/prof_r/*kernel_stats.csv | cut -c1-150 | tee $R/gpurun_out/r4r_kernel_stats_head.txt
rm -rf $R/gpurun_out/prof_r
