"""measured pieces of the row-partitioned tCG iteration: the dense product on a row strip of n / N cameras, the
cg_step launch, and the direct peer-write all-gather on `world` virtual devices (one GPU: no xGMI hop; the fabric adds its latency).
   python scripts/kbench_multi.py [n=1778] [--o 3 5]"""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xm-code_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, xmamd

ap = argparse.ArgumentParser()
ap.add_argument("n", type=int, nargs="?", default=1778)
ap.add_argument("--o", type=int, nargs="+", default=[3, 5])
ap.add_argument("--reps", type=int, default=200)
a = ap.parse_args()
L = xmamd.lib()
n = a.n
ld = xmamd.dense_ld(n)
rng = np.random.default_rng(0)
for o in a.o:
    OP = o | 1
    dW = xmamd.DevArray(rng.standard_normal((ld, OP)))
    for N in (1, 2, 4, 8):
        nloc = -(-n // N)
        dq = xmamd.DevArray(nbytes=3 * nloc * ld * 8)
        xmamd._chk(L.xm_dev_h2d(dq.ptr, rng.standard_normal(3 * nloc * ld).ctypes.data_as(C.c_void_p), 3 * nloc * ld * 8))
        dO = xmamd.DevArray(nbytes=3 * nloc * OP * 8)
        ms = C.c_double()
        xmamd._chk(L.xm_qw_dense_strip_time(dq.ptr, nloc, n, o, dW.ptr, dO.ptr, a.reps, C.byref(ms)))
        by = 8.0 * 3 * nloc * 3 * n
        line = f"strip product n={n} o={o} N={N}: {nloc} cameras / rank, {by/1e6:7.1f} MB of Q: {ms.value*1e3:7.1f} us  ({by/ms.value/1e6:7.0f} GB/s)"
        used = C.c_int()
        xmamd._chk(L.xm_qw_dense_strip_ks(dq.ptr, nloc, n, o, dW.ptr, dO.ptr, 1.0, 0, a.reps, C.byref(ms), C.byref(used)))
        line += f"   | columns split x{used.value} (policy): {ms.value*1e3:6.1f} us"
        for ks in (2, 3, 4, 6, 8):
            if ks != used.value:
                xmamd._chk(L.xm_qw_dense_strip_ks(dq.ptr, nloc, n, o, dW.ptr, dO.ptr, 1.0, ks, a.reps, C.byref(ms), C.byref(used)))
                line += f"  x{ks}: {ms.value*1e3:5.1f}"
        print(line, flush=True)
        dq.free(); dO.free()
    dW.free()
for world in (2, 4, 8):
    for o in a.o:
        OP = o | 1
        nloc = -(-n // world)
        count = nloc * 3 * OP + 3 * (-(-nloc // 4)) + 64      # tCG chunk: rows of B + partial sums
        us = C.c_double()
        rc = L.xm_peer_allgather_bench(world, 1, count, a.reps, C.byref(us))
        print(f"peer all-gather world={world} (virtual devices) {count*8/1024:6.1f} KB / rank (tCG chunk at n={n}, o={o}): " +
              (f"{us.value:6.1f} us per collective (push + wait launches)" if rc == 0 else "FAILED " + L.xm_last_error().decode()), flush=True)
