#!/bin/bash
# stress the two-ranks-on-one-GPU test (rank divergence shows up as a shared-memory barrier timeout; XM_COMM_TRACE pinpoints it)
export XM_SHM_TIMEOUT=${XM_SHM_TIMEOUT:-8}
for i in $(seq 1 ${1:-10}); do
  export XM_COMM_TRACE=/tmp/xmtrace_$i
  timeout 300 python -m pytest tests -m gpu -q -x -k "two_ranks" 2>&1 | grep -E "passed|failed|divergence|rank0:|rank1:" | head -8
done
