#!/bin/bash
# A/B of the outer iteration on the device (bench.py --outer device: the library's default with block-CSR products) against the host-driven form (--outer host: its default with dense products) on ONE box, alternating: headline + Final-13682 block-CSR leg
#   scripts/ab_outer.sh [rounds] > gpurun_out/<tag>_ab_outer.txt
rounds=${1:-2}
for r in $(seq 1 $rounds); do for o in host device; do
  echo -n "$o: "
  timeout 300 python bench.py --outer $o --steps 6 --warmup 1 --no-hbm-check --no-kkt-pair --cpu-kkt-seconds 0 --cpu-seconds 0 --no-rome-dense 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['rome_scale']
print('headline %d it/s %.1f ms/solve %.2f us/iter all-in, iterations %s, product %.2f us | Final-13682 block CSR %d it/s %.2f ms/solve %.2f us/iter all-in, %d iterations, product %.2f us' % (
  d['value'], d['ms_per_step'], d['us_per_tcg_iteration_all_in'], d['solve']['tcg_iters_by_step'], d['roofline']['avg_launch_ms'] * 1e3,
  r['value'], r['ms_per_step'], r['us_per_tcg_iteration_all_in'], r['tcg_iters_per_solve'], r['hess_launch_ms'] * 1e3))"
done; done
