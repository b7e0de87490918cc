#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sell" 2>&1 | tail -3
 echo "slab-major partial slots"; timeout 300 python scripts/kbench_sell.py 100000 50 --o 3 --slabs 2 4 8 --gather 0 1 --no-csr
 for A in 3 4 8 12; do echo "ABLATE=$A (3: cols+store only, 4: no store, 8: gather from LDS, 12: + no store)"; XM_SELL_DYNLDS=16384 XM_SELL_ABLATE=$A timeout 300 python scripts/kbench_sell.py 100000 50 --o 3 --slabs 4 --gather 0 --no-csr; done
 for D in 40000 60000 80000; do echo "occupancy: dynamic LDS $D bytes per workgroup"; XM_SELL_DYNLDS=$D timeout 300 python scripts/kbench_sell.py 100000 50 --o 3 --slabs 4 --gather 0 --no-csr; done
 for D in 60000 80000; do echo "occupancy gm1: dynamic LDS $D"; XM_SELL_DYNLDS=$D timeout 300 python scripts/kbench_sell.py 100000 50 --o 3 --slabs 4 --gather 1 --no-csr; done
) 2>&1 | tee gpurun_out/e_slots.log
