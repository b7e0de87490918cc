"""the 8-virtual-rank peer all-gather of scripts/kbench_multi.py, N times in fresh processes: python scripts/peer8_repeat.py [N=20]
(more than four ranks on one device synchronise their collectives through the host, Comm::device_waits: every run has to pass)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = f"""
import os, sys, ctypes as C
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, {os.path.join(ROOT, 'xm-code_amd')!r})
import xmamd
us = C.c_double()
rc = xmamd.lib().xm_peer_allgather_bench(8, 1, 2240, 200, C.byref(us))
print(("%.1f us" % us.value) if rc == 0 else "FAILED " + xmamd.lib().xm_last_error().decode())
"""
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ok = 0
for i in range(n):
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    line = (p.stdout.strip().splitlines() or ["(no output) " + p.stderr[-200:]])[-1]
    good = p.returncode == 0 and "FAILED" not in line and "us" in line
    ok += good
    print(f"run {i + 1:2d}: {line}", flush=True)
print(f"peer all-gather world=8 (virtual devices, 17.5 KB / rank): {ok} / {n} runs passed")
sys.exit(0 if ok == n else 1)
