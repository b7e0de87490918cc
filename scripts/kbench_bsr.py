"""micro-benchmark of the BSR3 Q*W kernel: python scripts/kbench_bsr.py n deg o [o ...] [--policy]   (--policy: also with the block stream's load policy forced)"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xm-code_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, xmamd, xm_testlib as tl
n = int(sys.argv[1]); deg = int(sys.argv[2]); os_ = [int(x) for x in sys.argv[3:] if not x.startswith("--")] or [3]
if os.environ.get("XM_KB_BAND") == "1":   # view graph WITH locality: camera i sees cameras i-deg/2 .. i+deg/2 (sequential capture)
    h = deg // 2
    lo = np.maximum(np.arange(n) - h, 0); hi = np.minimum(np.arange(n) + h, n - 1)
    cnt = hi - lo + 1
    rowptr = np.zeros(n + 1, dtype=np.int64); rowptr[1:] = np.cumsum(cnt)
    colidx = (np.repeat(lo, cnt) + (np.arange(rowptr[-1]) - np.repeat(rowptr[:-1], cnt))).astype(np.int32)
    P = dict(rowptr=rowptr, colidx=colidx, blocks=np.random.default_rng(n).standard_normal((rowptr[-1], 3, 3)))
else:
    P = tl.gen_vg(n, deg=deg, sigma=0.05, seed=n, dense=False)
nb = P["colidx"].size
if os.environ.get("XM_KB_COLMOD"):   # timing experiment: shrink the gathered working set of W (result meaningless)
    P["colidx"] = (P["colidx"] % int(os.environ["XM_KB_COLMOD"])).astype(np.int32)
L = xmamd.lib()
drp = xmamd.DevArray(P["rowptr"]); dci = xmamd.DevArray(P["colidx"]); dbl = xmamd.DevArray(P["blocks"].reshape(-1))
rng = np.random.default_rng(0)
for o in os_:
    OP = o | 1
    dW = xmamd.DevArray(rng.standard_normal((3 * n, OP))); dO = xmamd.DevArray(nbytes=3 * n * OP * 8)
    ms = C.c_double()
    xmamd._chk(L.xm_qw_bsr3_time(drp.ptr, dci.ptr, dbl.ptr, n, o, dW.ptr, dO.ptr, 100, C.byref(ms)))
    by = 76.0 * nb + 4 * (n + 1) + 2 * 8 * 3 * n * o
    pol = ""
    if "--order" in sys.argv:   # rows in camera order instead of binned by their number of windows
        xmamd._chk(L.xm_bench_bsr_binned(0)); m2 = C.c_double()
        xmamd._chk(L.xm_qw_bsr3_time(drp.ptr, dci.ptr, dbl.ptr, n, o, dW.ptr, dO.ptr, 100, C.byref(m2)))
        xmamd._chk(L.xm_bench_bsr_binned(1))
        pol += f"   [rows in camera order {m2.value*1e3:.1f} us]"
    if "--policy" in sys.argv or os.environ.get("XM_KB_POLICY"):   # the block stream's load policy forced: all cacheable / all non-temporal
        tt = []
        for nt in (0, 1):
            xmamd._chk(L.xm_bench_dense_policy(nt)); m2 = C.c_double()
            xmamd._chk(L.xm_qw_bsr3_time(drp.ptr, dci.ptr, dbl.ptr, n, o, dW.ptr, dO.ptr, 100, C.byref(m2))); tt.append(m2.value * 1e3)
        xmamd._chk(L.xm_bench_dense_policy(-1))
        pol += f"   [blocks all cacheable {tt[0]:.1f} us, all non-temporal {tt[1]:.1f} us]"
    print(f"BSR n={n} deg={deg} nb={nb} o={o}: {ms.value*1e3:8.1f} us  {by/ms.value/1e6:8.1f} GB/s algorithmic ({by/1e6:.1f} MB){pol}")
