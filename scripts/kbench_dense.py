"""micro-benchmark of the dense Q*W kernel (plain epilogue) through the C ABI:
   python scripts/kbench_dense.py n o [o ...] [--alt 1 0] [--nt -1 0 1] [--no-sym]
--alt: consecutive launches alternate the tile direction (1, what the solver does) or not (0); --nt: load policy of the stream (-1 = the
product's rule by size, 0 = default, 1 = non-temporal).  The half-traffic symmetric pair is timed next to it for o in 3..5."""
import argparse, sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xm-code_amd"))
import numpy as np, xmamd
ap = argparse.ArgumentParser()
ap.add_argument("n", type=int); ap.add_argument("o", type=int, nargs="*", default=[3])
ap.add_argument("--alt", type=int, nargs="*", default=[1]); ap.add_argument("--nt", type=int, nargs="*", default=[-1])
ap.add_argument("--no-sym", action="store_true")
a = ap.parse_args()
n = a.n
L = xmamd.lib(); ld = xmamd.dense_ld(n)
dq = xmamd.DevArray(nbytes=3 * n * ld * 8)
rng = np.random.default_rng(0)
# fill Q with random data in chunks (content irrelevant for timing but avoid zeros -> DVFS effects)
chunk = rng.standard_normal(min(3 * n * ld, 1 << 24))
off = 0
while off < 3 * n * ld:
    m = min(chunk.size, 3 * n * ld - off)
    xmamd._chk(L.xm_dev_h2d(C.c_void_p(dq.ptr.value + off * 8), chunk.ctypes.data_as(C.c_void_p), m * 8)); off += m
for o in a.o:
    OP = o | 1
    dW = xmamd.DevArray(rng.standard_normal((ld, OP))); dO = xmamd.DevArray(nbytes=3 * n * OP * 8)
    ms = C.c_double()
    by = 8.0 * (3 * n) ** 2 + 2 * 8 * 3 * n * o
    reps = 200 if n < 5000 else 20
    for nt in a.nt:
        for alt in a.alt:
            xmamd._chk(L.xm_bench_dense_policy(nt)); xmamd._chk(L.xm_bench_symv_k(0, alt, 0))
            tag = f"nt={nt:2d} alt={alt}"
            xmamd._chk(L.xm_qw_dense_time(dq.ptr, n, o, dW.ptr, dO.ptr, reps, C.byref(ms)))
            print(f"n={n} o={o} {tag} {ms.value*1e3:8.1f} us  {by/ms.value/1e6:8.1f} GB/s algorithmic  ({by/1e6:.1f} MB)", flush=True)
            if 3 <= o <= 5 and not a.no_sym:
                xmamd._chk(L.xm_qw_dense_sym_time(dq.ptr, n, o, dW.ptr, dO.ptr, reps, C.byref(ms)))
                print(f"n={n} o={o} {tag} SYM (upper triangle only): {ms.value*1e3:8.1f} us  {by/ms.value/1e6:8.1f} GB/s of full-storage algorithmic bytes", flush=True)
xmamd._chk(L.xm_bench_dense_policy(-1)); xmamd._chk(L.xm_bench_symv_k(0, 1, 0))
