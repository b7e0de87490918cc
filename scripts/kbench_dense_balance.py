"""Is the dense product bound per CU (then the time follows the most loaded CU: a staircase in ceil(workgroups / 256)) or chip-wide (then it
follows the bytes)?   python scripts/kbench_dense_balance.py [o]
(a) the plain kernel over a sweep of n around the headline's 1778 cameras (445 workgroups on 256 CUs = 1.74 per CU: 2 on most, 1 on the rest);
(b) the same matrix with the columns split over ks workgroups per camera group (xm_qw_dense_strip_ks: finer work units, same bytes)."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xm-code_amd"))
import numpy as np, xmamd
o = int(sys.argv[1]) if len(sys.argv) > 1 else 3
L = xmamd.lib(); rng = np.random.default_rng(0)
OP = o | 1
for n in (1024, 1280, 1536, 1664, 1778, 1792, 2048, 2304, 2560, 3072, 3584):
    ld = xmamd.dense_ld(n)
    dq = xmamd.DevArray(rng.standard_normal(3 * n * ld)); dW = xmamd.DevArray(rng.standard_normal((ld, OP))); dO = xmamd.DevArray(nbytes=3 * n * OP * 8)
    ms = C.c_double(); ku = C.c_int()
    by = 8.0 * (3 * n) ** 2 + 2 * 8 * 3 * n * o
    g = (n + 3) // 4
    line = f"n={n:5d} o={o} workgroups {g:4d} = {g / 256:5.2f} per CU (most loaded CU: {-(-g // 256)}), {by / 1e6:6.1f} MB:"
    xmamd._chk(L.xm_qw_dense_time(dq.ptr, n, o, dW.ptr, dO.ptr, 200, C.byref(ms)))
    line += f"  plain {ms.value * 1e3:6.1f} us = {by / ms.value / 1e6:6.0f} GB/s |"
    for ks in (2, 3, 4, 5, 7, 8):
        xmamd._chk(L.xm_qw_dense_strip_ks(dq.ptr, n, n, o, dW.ptr, dO.ptr, 1.0, ks, 200, C.byref(ms), C.byref(ku)))
        line += f" ks={ku.value}: {ms.value * 1e3:6.1f}"
    print(line, flush=True)
    dq.free(); dW.free(); dO.free()
