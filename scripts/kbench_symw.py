"""per-rank time of the multi-rank symmetric window product (xm_symw.hip) on ONE GPU: python scripts/kbench_symw.py n [--o 3] [--worlds 2 4 8]
For every world size the share of rank 0, of a middle rank and of the last rank is timed (the window makes them equal by construction);
the general full-strip kernel of the same rank is timed next to it."""
import argparse, os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xm-code_amd"))
import numpy as np, xmamd
ap = argparse.ArgumentParser(); ap.add_argument("n", type=int); ap.add_argument("--o", type=int, default=3)
ap.add_argument("--worlds", type=int, nargs="+", default=[2, 4, 8]); ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
L = xmamd.lib()
for world in a.worlds:
    nloc = -(-a.n // world); nloc += nloc & 1
    ntot = nloc * world
    for rank in sorted({0, world // 2, world - 1}):
        ms = (C.c_double * 2)(); by = C.c_int64()
        xmamd._chk(L.xm_qw_symw_time(ntot, nloc, rank * nloc, a.o, world, a.reps, ms, C.byref(by)))
        strip = 8.0 * 3 * nloc * 3 * ntot
        print(f"symw n={a.n} o={a.o} world={world} rank={rank}: sweep+colsum {ms[0]*1e3:8.1f} us  reduce {ms[1]*1e3:6.1f} us  streams {by.value/1e6:8.1f} MB "
              f"({by.value/strip:.3f} of the {strip/1e6:.0f} MB strip) -> {by.value/ms[0]/1e6:7.1f} GB/s real, {strip/(ms[0]+ms[1])/1e6:7.1f} GB/s in full-strip accounting", flush=True)
    # the general kernel on one rank's full strip
    ld = xmamd.dense_ld(ntot)
    Q = xmamd.DevArray(nbytes=3 * nloc * ld * 8); W = xmamd.DevArray(nbytes=(ld * (a.o | 1) + 16) * 8); O = xmamd.DevArray(nbytes=3 * nloc * (a.o | 1) * 8)
    ms1 = C.c_double()
    xmamd._chk(L.xm_qw_dense_strip_time(Q.ptr, nloc, ntot, a.o, W.ptr, O.ptr, a.reps, C.byref(ms1)))
    print(f"full strip (general kernel) world={world}: {ms1.value*1e3:8.1f} us = {8.0*3*nloc*3*ntot/ms1.value/1e6:7.1f} GB/s", flush=True)
    Q.free(); W.free(); O.free()
