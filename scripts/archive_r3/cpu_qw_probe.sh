# host-side probe for cpu_baseline: topology of the GPU box and the oracle's NUMA row-block product under several thread settings
lscpu | egrep "Model name|Socket|Core|Thread|NUMA|L3|^CPU\(s\)" ; cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc
cat > /tmp/probe.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
import numpy as np
from oracle import xm_oracle as xo
n = 1778; m = 3 * n
rng = np.random.default_rng(0)
Q = np.asfortranarray(rng.standard_normal((m, m)))
W = np.asfortranarray(rng.standard_normal((m, 3)))
xo.numa_prepare(Q)
out = np.zeros_like(W, order="F")
L = xo.lib()
for _ in range(5): L.xmo_qw(n, 3, xo._p(Q), xo._p(W), xo._p(out), 1.0)
t0 = time.perf_counter(); reps = 200
for _ in range(reps): L.xmo_qw(n, 3, xo._p(Q), xo._p(W), xo._p(out), 1.0)
dt = (time.perf_counter() - t0) / reps
print("threads", xo.num_threads(), os.environ.get("OMP_NUM_THREADS"), os.environ.get("OMP_PLACES"), os.environ.get("OMP_PROC_BIND"), "qw %.3f ms = %.0f GB/s" % (dt * 1e3, 8.0 * m * m / dt / 1e9), flush=True)
PY
for cfg in "128 cores close" "64 cores close" "64 cores spread" "32 cores spread" "16 cores spread" "16 cores close" "8 cores spread" "24 cores spread"; do
  set -- $cfg
  OMP_NUM_THREADS=$1 OMP_PLACES=$2 OMP_PROC_BIND=$3 python /tmp/probe.py
done
OMP_NUM_THREADS=16 python /tmp/probe.py; OMP_NUM_THREADS=16 OMP_WAIT_POLICY=active python /tmp/probe.py
