# which of today's changes broke the 4-virtual-rank Venice run?  (one variant per line)
run() { echo "== $*"; env "$@" XM_WATCHDOG_S=18 timeout 100 python bench.py --gpus 4 --steps 1 --warmup 0 --no-rome --cpu-seconds 0 2>&1 | tail -1 | cut -c1-250; }
run XM_SPLIT_K=-1
run HSA_ENABLE_SDMA=1
run XM_EXCHANGE_LITE=0
run XM_EXCHANGE=1
