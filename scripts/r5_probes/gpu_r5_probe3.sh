#!/bin/bash
# round 5, third GPU call: suite after the fused retraction / grow-only workspace / early epilogue operands; does a library built with
# --offload-compress load and run on this runtime?; bench lines again; 50 k-camera matrix-free scene
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out
( timeout 1800 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -40 ) > $O/r05_pytest_gpu_c.txt
python - > $O/r05_compress_probe.txt 2>&1 <<'PY'
import sys, os, time
sys.path.insert(0, "xm-code_amd"); sys.path.insert(0, "tests")
import xmamd
xmamd.LIB_PATH = os.path.join("xm-code_amd", "lib_c", "libxm_amd.so")      # the same sources compiled with --offload-compress
import numpy as np, xm_testlib as tl
t0 = time.time()
P = tl.gen_vg(40, deg=3, sigma=1.5, seed=40)
R, s, info = xmamd.solve_dense(P["Q"], 6, 1e-9, 3.0)
print("compressed-bundle library:", os.path.getsize(xmamd.LIB_PATH), "bytes; first solve (code object load included)", round(time.time() - t0, 2), "s; rank", info["rank"], "status", info["status"], "primal", info["primal"])
V = tl.gen_vg(300, deg=12, sigma=0.2, seed=3)
M = xmamd.SellMatrix(V["rowptr"], V["colidx"], V["blocks"], codec=1)
W3 = np.random.default_rng(1).standard_normal((900, 3))
print("sliced-ELL product vs numpy", tl.rel_fro(M.qw(W3, 1.0), V["Q"] @ W3))
PY
python - > $O/r05_load_time_plain.txt 2>&1 <<'PY'
import sys, os, time
sys.path.insert(0, "xm-code_amd"); sys.path.insert(0, "tests")
import xmamd, numpy as np, xm_testlib as tl
t0 = time.time()
P = tl.gen_vg(40, deg=3, sigma=1.5, seed=40)
R, s, info = xmamd.solve_dense(P["Q"], 6, 1e-9, 3.0)
print("plain library:", os.path.getsize(xmamd.LIB_PATH), "bytes; first solve", round(time.time() - t0, 2), "s")
PY
python bench.py --workload final13682 --storage bsr --steps 3 --warmup 1 --no-hbm-check --cpu-seconds 0 > $O/r05_bench_rome_bsr_c.json 2>&1
python bench.py > $O/r05_bench_venice1778_c.json 2> $O/r05_bench_venice1778_c.err
python bench.py --workload vg100k --storage vg --steps 3 --warmup 1 --no-rome --no-hbm-check --cpu-seconds 0 > $O/r05_bench_vg100k_vg_c.json 2> $O/r05_bench_vg100k_vg_c.err
python scripts/kbench_schur.py 50000 1500000 6 --product-only --trace > $O/r05_kbench_schur_50k.txt 2>&1
ls -la $O | tail -12
