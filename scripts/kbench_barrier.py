"""a grid-wide barrier inside one launch against a kernel boundary (xm_bench_grid_barrier): microseconds per round
   python scripts/kbench_barrier.py [blocks ...]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xm-code_amd"))
import ctypes as C
import numpy as np
import xmamd
xmamd.require_gpu()
L = xmamd.lib()
for blocks in ([int(a) for a in sys.argv[1:]] or [64, 128, 256, 445, 512, 856, 1024]):
    for rounds in (50, 200):
        us = np.zeros(3)
        rc = L.xm_bench_grid_barrier(blocks, rounds, 20, us.ctypes.data_as(C.c_void_p))
        msg = "" if rc == 0 else " error %d: %s" % (rc, L.xm_bench_last_error().decode())
        print(f"{blocks:5d} workgroups, {rounds:4d} rounds: {us[0]:7.2f} us per round with a grid barrier, {us[1]:7.2f} us per round as separate launches, failed checks / expired waits {int(us[2])}{msg}", flush=True)
