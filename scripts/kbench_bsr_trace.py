"""per-wavefront 100 MHz timestamps of ONE launch of the plain block-CSR product (entry, row record in, each window's data in, end of the loop, end).
   Needs an experiment build:  make -C xm-code_amd OBJDIR=obj_x LIBDIR=lib_x EXTRA=-DXM_BSR_TRACE lib_x/libxm_amd.so
   XMAMD_LIB=xm-code_amd/lib_x/libxm_amd.so python scripts/kbench_bsr_trace.py n deg o"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xm-code_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, xmamd, xm_testlib as tl
n = int(sys.argv[1]); deg = int(sys.argv[2]); o = int(sys.argv[3])
P = tl.gen_vg(n, deg=deg, sigma=0.05, seed=n, dense=False)
L = xmamd.lib()
L.xm_qw_bsr3_trace.argtypes = [C.c_void_p] * 3 + [C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
drp = xmamd.DevArray(P["rowptr"]); dci = xmamd.DevArray(P["colidx"]); dbl = xmamd.DevArray(P["blocks"].reshape(-1))
OP = o | 1
dW = xmamd.DevArray(np.random.default_rng(0).standard_normal((3 * n, OP))); dO = xmamd.DevArray(nbytes=3 * n * OP * 8)
nwg = (n + 15) // 16
tr = np.zeros((nwg * 4, 8), dtype=np.uint64)
xmamd._chk(L.xm_qw_bsr3_trace(drp.ptr, dci.ptr, dbl.ptr, n, o, dW.ptr, dO.ptr, tr.ctypes.data_as(C.c_void_p)))
t = tr.astype(np.int64)
t0 = t[:, 0].min()
us = (t - t0) / 100.0
us[t == 0] = np.nan
names = ["entry", "rowptr in", "window 0 data in", "window 1 data in", "window >=2 data in", "loop done", "end"]
cols = [0, 1, 2, 3, 4, 5, 6]
print(f"n={n} deg={deg} o={o}: {nwg} workgroups, {nwg*4} wavefronts; times in us from the first wavefront's entry")
for c, nm in zip(cols, names):
    v = us[:, c]; v = v[~np.isnan(v)]
    if v.size: print(f"  {nm:22s} count {v.size:5d}  min {v.min():6.2f}  p10 {np.percentile(v,10):6.2f}  median {np.median(v):6.2f}  p90 {np.percentile(v,90):6.2f}  max {v.max():6.2f}")
life = us[:, 6] - us[:, 0]
print(f"  wavefront lifetime: median {np.nanmedian(life):.2f}  p90 {np.nanpercentile(life,90):.2f}  max {np.nanmax(life):.2f}")
for a_, b_, nm in [(0, 1, "entry -> rowptr"), (1, 2, "rowptr -> window 0 data"), (2, 3, "window 0 -> window 1 data"), (3, 4, "window 1 -> window 2 data"), (5, 6, "loop done -> end")]:
    d = us[:, b_] - us[:, a_]; d = d[~np.isnan(d)]
    if d.size: print(f"  {nm:28s} median {np.median(d):5.2f}  p90 {np.percentile(d,90):5.2f}")
late = us[:, 0] > 3.0
print(f"  wavefronts entering later than 3 us: {int(late.sum())}; their entry median {np.nanmedian(us[late,0]) if late.any() else float('nan'):.2f}")
