# A/B of the fused tCG exchange on two VIRTUAL ranks of one GPU: release-fence form against write-through form (wall clock of bench.py)
export GPU_MAX_HW_QUEUES=16
for lite in 0 1; do
  XM_EXCHANGE_LITE=$lite timeout 200 python bench.py --gpus 2 --steps 6 --warmup 2 --no-rome --cpu-seconds 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('lite=$lite', 'it/s %.0f' % d['value'], 'ms %.1f' % d['ms_per_step'], d['solve']['tcg_iters_by_step'], d['solve']['status'], d['solve']['primal'])"
done
