#!/bin/bash
# kernel traces of the headline and the Final-13682 block-CSR solve, outer iteration on the device / on the host -> gpurun_out/<tag>_trace_<leg>_<outer>.txt
TAG=${1:-dbg}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for outer in ${OUTERS:-device host}; do
for leg in "venice1778:--steps 3 --warmup 1 --cpu-seconds 0 --no-hbm-check --no-rome --no-kkt-pair --cpu-kkt-seconds 0" \
           "rome_bsr:--workload final13682 --storage bsr --steps 3 --warmup 1 --no-hbm-check --cpu-seconds 0 --no-rome --no-kkt-pair --cpu-kkt-seconds 0"; do
  name=${leg%%:*}; args=${leg#*:}
  rm -rf $O/prof_$name
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_$name -o run -- python $R/bench.py $args --outer $outer > $O/prof_$name.log 2>&1
  f=$(ls $O/prof_$name/*/*kernel_trace.csv $O/prof_$name/*kernel_trace.csv 2>/dev/null | head -1)
  python $R/scripts/trace_summary.py $f 0.7 --window 120 > $O/${TAG}_trace_${name}_${outer}.txt 2>&1
  rm -rf $O/prof_$name $O/prof_$name.log
done; done
