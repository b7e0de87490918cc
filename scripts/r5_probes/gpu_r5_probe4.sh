#!/bin/bash
# round 5, fourth GPU call: A/B builds on one box -- epilogue operands early (lib_x), run-ahead 2 (lib_c), one speculative iteration (lib_d), both (lib_e)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out
: > $O/r05_ab_rome.txt
for rep in 1 2; do
for v in lib lib_x lib_c lib_d lib_e; do
  XMAMD_LIB=$R/xm-code_amd/$v/libxm_amd.so python bench.py --workload final13682 --storage bsr --steps 4 --warmup 1 --no-hbm-check --cpu-seconds 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s = d['solve']
print('$v rep $rep: %.0f it/s %.2f ms/solve  tcg %d outer %d  tr %.2f ms cert %.2f ms  hess %.2f us' % (d['value'], d['ms_per_step'], s['tcg_iters_per_solve'], s['outer_iters'], s['tr_seconds'] * 1e3, s['cert_seconds'] * 1e3, d['roofline']['avg_launch_ms'] * 1e3))" >> $O/r05_ab_rome.txt
done
done
for v in lib lib_c lib_e; do
  XMAMD_LIB=$R/xm-code_amd/$v/libxm_amd.so python bench.py --steps 6 --warmup 1 --no-hbm-check --no-rome --cpu-seconds 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s = d['solve']
print('venice $v: %.0f it/s %.2f ms/solve  tr %.2f ms cert %.2f ms  hess %.2f us' % (d['value'], d['ms_per_step'], s['tr_seconds'] * 1e3, s['cert_seconds'] * 1e3, d['roofline']['avg_launch_ms'] * 1e3))" >> $O/r05_ab_rome.txt
done
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/rome_trace -o run -- python $R/bench.py --workload final13682 --storage bsr --steps 3 --warmup 1 --no-hbm-check --cpu-seconds 0 > /dev/null 2> $O/r05_rome_trace_d.err
cd $R
f=$(ls $O/rome_trace/*/*kernel_trace.csv $O/rome_trace/*kernel_trace.csv 2>/dev/null | head -1)
python scripts/trace_summary.py $f 0.7 > $O/r05_trace_summary_rome_bsr_d.txt 2>&1
rm -rf $O/rome_trace
cat $O/r05_ab_rome.txt
