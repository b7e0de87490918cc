"""micro-benchmark of the half-traffic symmetric dense product (qw_symv_kernel + symv_reduce_kernel, plain epilogue):
   python scripts/kbench_symv.py n [--o 3 4] [--alt 1 0] [--k 0 6 8] [--check] [--trace]
per (alternating sweep direction or not, chunk length K; 0 = the plan's): the pair's average time over back-to-back launches (HIP events)
next to the general kernel; --check compares with numpy on a random symmetric matrix; --trace prints what the per-wavefront timestamps
of ONE launch say (dispatch spread, head, time per step, foot)."""
import argparse, os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xm-code_amd"))
import numpy as np, xmamd

ap = argparse.ArgumentParser()
ap.add_argument("n", type=int)
ap.add_argument("--o", type=int, nargs="*", default=[3, 4])
ap.add_argument("--alt", type=int, nargs="*", default=[1])
ap.add_argument("--k", type=int, nargs="*", default=[0])
ap.add_argument("--check", action="store_true")
ap.add_argument("--trace", action="store_true")
a = ap.parse_args()
n = a.n
L = xmamd.lib(); ld = xmamd.dense_ld(n)
L.xm_qw_dense_sym_trace.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int)]
rng = np.random.default_rng(0)
reps = 200 if n < 5000 else 20

if a.check:
    A = rng.standard_normal((3 * n, 3 * n)); Qh = A + A.T; del A
    dq = xmamd.dense_upload(Qh)
else:
    Qh = None
    dq = xmamd.DevArray(nbytes=3 * n * ld * 8)
    chunk = rng.standard_normal(min(3 * n * ld, 1 << 24)); off = 0
    while off < 3 * n * ld:
        m = min(chunk.size, 3 * n * ld - off)
        xmamd._chk(L.xm_dev_h2d(C.c_void_p(dq.ptr.value + off * 8), chunk.ctypes.data_as(C.c_void_p), m * 8)); off += m

for o in a.o:
    OP = o | 1
    Wh = rng.standard_normal((3 * n, o))
    dW = xmamd.DevArray(xmamd.to_rm(Wh, rows=ld)); dO = xmamd.DevArray(nbytes=3 * n * OP * 8)
    ms = C.c_double()
    by = 8.0 * (3 * n) ** 2 + 2 * 8 * 3 * n * o
    xmamd._chk(L.xm_qw_dense_time(dq.ptr, n, o, dW.ptr, dO.ptr, reps, C.byref(ms)))
    print(f"n={n} o={o} general kernel          {ms.value*1e3:8.1f} us  {by/ms.value/1e6:8.1f} GB/s algorithmic", flush=True)
    ref = Qh @ Wh if a.check else None
    for v in a.alt:
        for k in a.k:
            xmamd._chk(L.xm_bench_symv_k(k, v, 0))
            p = (C.c_int32 * 4)(); xmamd._chk(L.xm_symv_plan(n, p))
            xmamd._chk(L.xm_qw_dense_sym_time(dq.ptr, n, o, dW.ptr, dO.ptr, reps, C.byref(ms)))
            line = f"n={n} o={o} SYM alternating {v} K={p[0]:3d} Kf={p[1]:3d} {ms.value*1e3:8.1f} us  {by/ms.value/1e6:8.1f} GB/s full-storage"
            if a.check:
                xmamd._chk(L.xm_qw_dense_sym(dq.ptr, n, o, dW.ptr, dO.ptr, 1.0, None)); xmamd._chk(L.xm_dev_sync())
                out = xmamd.from_rm(dO.get(), 3 * n, o)
                line += f"  max rel err vs numpy {np.abs(out - ref).max() / np.abs(ref).max():.2e}"
            print(line, flush=True)
    if a.trace and o in (3, 4):
        for k in a.k:
            xmamd._chk(L.xm_bench_symv_k(k, 1, 0))
            grid = (C.c_int * 2)(); slots = C.c_int()
            xmamd._chk(L.xm_qw_dense_sym_trace(dq.ptr, n, o, dW.ptr, dO.ptr, None, 0, grid, C.byref(slots)))
            S = slots.value; nw = grid[0] * grid[1] * 4
            T = np.zeros(nw * S, dtype=np.uint64)
            xmamd._chk(L.xm_qw_dense_sym_trace(dq.ptr, n, o, dW.ptr, dO.ptr, T.ctypes.data_as(C.c_void_p), T.size, grid, C.byref(slots)))
            T = T.reshape(nw, S)
            live = T[:, 1] > 0                                   # wavefronts that got past the early exits
            t0 = T[T[:, 0] > 0, 0].min()
            us = lambda x: (x.astype(np.float64) - float(t0)) / 100.0
            ent = us(T[live, 0]); head = us(T[live, 1]); loop = us(T[live, S - 3]); end = us(T[live, S - 2])
            nst = (T[live, 2:S - 3] > 0).sum(axis=1)
            first = us(T[live, 2]); work = nst > 0
            q = lambda x: "min %.2f  p10 %.2f  p50 %.2f  p90 %.2f  max %.2f" % tuple(np.percentile(x, [0, 10, 50, 90, 100]))
            print(f"  trace K={k or 'plan'} grid {grid[0]}x{grid[1]}: {int(live.sum())} live wavefronts of {nw} slots, {int(work.sum())} with steps; "
                  f"{int(nst.sum())} steps")
            print("    entry after the first entry (us):   " + q(ent))
            print("    status word seen - entry:           " + q(head - ent))
            print("    first step done - entry:            " + q((first - ent)[work]))
            if work.any():
                per = ((loop - first)[work & (nst > 1)]) / (nst[work & (nst > 1)] - 1)
                print("    per further step:                   " + q(per))
            print("    foot (barrier + column sums):       " + q(end - loop))
            print("    end after the first entry:          " + q(end))
            hw = T[live, S - 1]
            xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xf; cu = ((hw >> np.uint64(8)) & np.uint64(0xf)).astype(np.int64); se = ((hw >> np.uint64(13)) & np.uint64(0x7)).astype(np.int64)
            key = xcc * 1000 + se * 16 + cu
            cnt = np.bincount(np.unique(key, return_inverse=True)[1])
            print(f"    wavefronts with steps per XCC: {np.bincount(xcc[work], minlength=8).tolist()}; distinct (xcc, se, cu) = {cnt.size}, wavefronts per CU "
                  f"min {cnt.min()} p50 {int(np.median(cnt))} max {cnt.max()}")
    xmamd._chk(L.xm_bench_symv_k(0, 1, 0))
