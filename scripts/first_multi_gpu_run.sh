#!/bin/bash
# The FIRST minutes on a real N-GPU node (no N > 1 figure in this tree was measured on more than one physical GPU: DESIGN section 4).
#   bash scripts/first_multi_gpu_run.sh [--dry-run] N [TAG]
# In order, each step guarded by its own timeout and logged, a failing step does not stop the next one:
#   1. transport ladder rung by rung on a small problem (scripts/multi_gpu_probe.py: peer writes, forced RCCL, one process per GPU)
#   2. the multi-rank GPU tests of the suite (virtual devices, IPC processes, the RCCL two-rank test: on this node RCCL gets distinct devices)
#   3. bench.py --gpus N, single-process mode: default ladder, --exchange peer, --exchange rccl
#   4. bench.py under torch.distributed.run, one process per GPU (what the driver launches): default and --exchange rccl
#   5. per-rank share of the multi-rank symmetric window on every device (scripts/kbench_symw.py), peer all-gather latency (kbench_multi.py)
#   6. profiles/multi_<N>gpu_SUMMARY.md (scripts/collect_profiles.py multi_<N>gpu)
# --dry-run prints the commands (tests/test_bench_contract.py checks that every one of them parses at --help level on a CPU box).
DRY=0; if [ "$1" = "--dry-run" ]; then DRY=1; shift; fi
N=${1:-8}; TAG=${2:-multi_${N}gpu}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
PORT=29517
run() {   # run <log name> <timeout s> <command ...>
  local log=$O/${TAG}_$1; local to=$2; shift 2
  if [ $DRY = 1 ]; then echo "$@"; return 0; fi
  echo "== $*" > $log.log
  timeout $to "$@" >> $log.log 2>&1; echo "exit $?" >> $log.log
  tail -3 $log.log
}
cd $R
run probe 900 python scripts/multi_gpu_probe.py --gpus $N --rungs peer rccl procs
run pytest_multi 2400 python -m pytest tests -m gpu -q --timeout 900 -k "ipc or virtual or two_ranks or multi_gpu or multi_rank or rccl"
run bench_team 1200 python bench.py --gpus $N --steps 3 --warmup 1 --cpu-seconds 0
run bench_team_peer 900 python bench.py --gpus $N --steps 3 --warmup 1 --cpu-seconds 0 --exchange peer --no-rccl-leg
run bench_team_rccl 900 python bench.py --gpus $N --steps 3 --warmup 1 --cpu-seconds 0 --exchange rccl
run bench_procs 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus $N --steps 3 --warmup 1 --cpu-seconds 0
run bench_procs_rccl 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((PORT+1)) bench.py --gpus $N --steps 3 --warmup 1 --cpu-seconds 0 --exchange rccl
for n in 1 2 4 8; do
  [ $n -le $N ] && [ $n -gt 1 ] && run scale_$n 1500 python bench.py --gpus $n --steps 3 --warmup 1 --cpu-seconds 0 --no-rccl-leg
done
run kbench_symw 900 python scripts/kbench_symw.py 13682 --worlds 2 $N
run kbench_multi 600 python scripts/kbench_multi.py 1778 --o 3
if [ $DRY = 1 ]; then echo python scripts/collect_profiles.py $TAG; exit 0; fi
for f in $O/${TAG}_bench_*.log $O/${TAG}_scale_*.log; do   # the JSON line of every bench leg next to its log
  [ -f $f ] && grep '^{' $f | tail -1 > ${f%.log}.json
done
python scripts/collect_profiles.py $TAG
ls $O | grep -c $TAG
