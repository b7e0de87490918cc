# bench.py on N virtual devices of one GPU (functional runs of the N-rank flow; not a scaling measurement)
for n in 2 4 8; do
  timeout 250 python bench.py --gpus $n --steps 3 --warmup 1 --no-rome --cpu-seconds 0 2>gpurun_out/bench_virtual_err.log | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('N=$n', 'it/s %.0f' % d['value'], 'ms %.1f' % d['ms_per_step'], d['solve']['tcg_iters_by_step'], d['solve']['status'], d['solve']['exchange'])"
done
