#!/bin/bash
# round 3, pass C: all round-3 tests (full failure output), benches (headline with rotating groupings, 100k view graph in both
# storages, single-process 2 virtual GPUs), the whole parity suite
mkdir -p gpurun_out; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_round3.py -m gpu -q --timeout 900 -rf 2>&1 > gpurun_out/r3c_new_full.log; tail -40 gpurun_out/r3c_new_full.log
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r3c_new_full.log | tail -20 > gpurun_out/r3c_new.log
timeout 600 python bench.py --steps 6 --warmup 1 > gpurun_out/r3c_bench_venice.json 2> gpurun_out/r3c_bench_venice.err; tail -c 600 gpurun_out/r3c_bench_venice.json
timeout 600 python bench.py --workload vg100k --storage vg --steps 3 --warmup 1 --no-rome --no-hbm-check --cpu-seconds 0 > gpurun_out/r3c_bench_vg100k_vg.json 2> gpurun_out/r3c_bench_vg100k_vg.err; tail -c 400 gpurun_out/r3c_bench_vg100k_vg.json
timeout 600 python bench.py --workload vg100k --storage bsr --steps 3 --warmup 1 --no-rome --no-hbm-check --cpu-seconds 0 > gpurun_out/r3c_bench_vg100k_bsr.json 2> gpurun_out/r3c_bench_vg100k_bsr.err; tail -c 400 gpurun_out/r3c_bench_vg100k_bsr.json
XM_WATCHDOG_S=60 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-rome-dense --cpu-seconds 0 > gpurun_out/r3c_bench_2gpu.json 2> gpurun_out/r3c_bench_2gpu.err; tail -c 600 gpurun_out/r3c_bench_2gpu.json; tail -3 gpurun_out/r3c_bench_2gpu.err
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 900 -rf 2>&1 | tail -15 | tee gpurun_out/r3c_suite.log
