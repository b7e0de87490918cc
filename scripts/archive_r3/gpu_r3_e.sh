#!/bin/bash
# round 3, pass E: column-split strip product (tests + timings), 4 virtual GPUs, both suites
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -m gpu -q --timeout 600 -rf -k "column_split or split_k or mid700" 2>&1 | tail -15 | tee gpurun_out/r3e_tests.log
(timeout 300 python scripts/kbench_multi.py 1778 --o 3 5; timeout 300 python scripts/kbench_multi.py 356 --o 3) 2>&1 | tee gpurun_out/r3e_kbench_multi.log
XM_WATCHDOG_S=60 timeout 600 python bench.py --gpus 4 --steps 3 --warmup 0 --no-rome --cpu-seconds 0 > gpurun_out/r3e_bench_4gpu.json 2> gpurun_out/r3e_bench_4gpu.err; tail -c 700 gpurun_out/r3e_bench_4gpu.json; tail -3 gpurun_out/r3e_bench_4gpu.err
timeout 1500 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -m gpu -q --timeout 900 -rf 2>&1 | tail -15 | tee gpurun_out/r3e_suite.log
