"""micro-benchmark of the dense Q*W kernel (plain epilogue) through the C ABI: python scripts/kbench_dense.py n o [o ...]"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xm-code_amd"))
import numpy as np, xmamd
n = int(sys.argv[1]); os_ = [int(x) for x in sys.argv[2:]] or [3]
L = xmamd.lib(); ld = xmamd.dense_ld(n)
dq = xmamd.DevArray(nbytes=3 * n * ld * 8)
rng = np.random.default_rng(0)
# fill Q with random data in chunks (content irrelevant for timing but avoid zeros -> DVFS effects)
chunk = rng.standard_normal(min(3 * n * ld, 1 << 24))
off = 0
while off < 3 * n * ld:
    m = min(chunk.size, 3 * n * ld - off)
    xmamd._chk(L.xm_dev_h2d(C.c_void_p(dq.ptr.value + off * 8), chunk.ctypes.data_as(C.c_void_p), m * 8)); off += m
for o in os_:
    OP = o | 1
    dW = xmamd.DevArray(rng.standard_normal((ld, OP))); dO = xmamd.DevArray(nbytes=3 * n * OP * 8)
    ms = C.c_double()
    xmamd._chk(L.xm_qw_dense_time(dq.ptr, n, o, dW.ptr, dO.ptr, 200 if n < 5000 else 20, C.byref(ms)))
    by = 8.0 * (3 * n) ** 2 + 2 * 8 * 3 * n * o
    print(f"n={n} o={o} {ms.value*1e3:8.1f} us  {by/ms.value/1e6:8.1f} GB/s algorithmic  ({by/1e6:.1f} MB)")
    if 3 <= o <= 5:
        xmamd._chk(L.xm_qw_dense_sym_time(dq.ptr, n, o, dW.ptr, dO.ptr, 200 if n < 5000 else 20, C.byref(ms)))
        print(f"n={n} o={o} SYM (upper triangle only): {ms.value*1e3:8.1f} us  {by/ms.value/1e6:8.1f} GB/s of full-storage algorithmic bytes")
