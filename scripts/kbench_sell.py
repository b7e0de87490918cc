"""micro-benchmark of the sliced-ELL Q*W product (xm_sell.hip) against the block-CSR kernel:
   python scripts/kbench_sell.py n deg [--o 3 5] [--slabs 1 2 4 8] [--gather 0 1] [--lmax 64] [--band] [--check] [--reps 100] [--upper]
The generated problem is cached under /tmp so that repeated invocations (rocprofv3 passes) do not pay for it again."""
import argparse, os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xm-code_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, xmamd, xm_testlib as tl

ap = argparse.ArgumentParser()
ap.add_argument("n", type=int); ap.add_argument("deg", type=int)
ap.add_argument("--o", type=int, nargs="+", default=[3])
ap.add_argument("--slabs", type=int, nargs="+", default=[4])
ap.add_argument("--gather", type=int, nargs="+", default=[1], help="0 = a record of W per lane, 1 = LDS-transposed (the solver's default)")
ap.add_argument("--lmax", type=int, default=64)
ap.add_argument("--band", action="store_true")
ap.add_argument("--skew", action="store_true", help="add hub cameras (power-law degrees)")
ap.add_argument("--check", action="store_true")
ap.add_argument("--no-csr", action="store_true")
ap.add_argument("--reps", type=int, default=100)
ap.add_argument("--codec", type=int, nargs="+", default=[0], help="0 = 9 doubles per block, 1 = view-graph codec (quaternion per block)")
ap.add_argument("--padded", action="store_true", help="hand W over at the 128-byte record pitch as well (what the solver's tCG does: xm_tuning_t.sell_wpad)")
ap.add_argument("--upper", action="store_true", help="TIMING EXPERIMENT (round 5, VERDICT r4 #1): keep only the blocks with column >= row -- the stream, the "
                "gather and the FMAs of a symmetric-half storage WITHOUT its transposed contributions: a lower bound of what such a product's "
                "main launch could cost (the result is not Q W)")
a = ap.parse_args()
n, deg = a.n, a.deg
vgform = 1 in a.codec        # the codec needs a real view-graph matrix (rotation blocks), also for the banded / hub graphs
tag = f"/tmp/xm_kb_{n}_{deg}_{int(a.band)}{int(a.skew)}{int(vgform)}.npz"   # (the --upper cut is applied after loading)
if os.path.exists(tag):
    Z = np.load(tag); P = dict(rowptr=Z["rowptr"], colidx=Z["colidx"], blocks=Z["blocks"])
else:
    if (a.band or a.skew) and vgform:
        rng0 = np.random.default_rng(n)
        if a.band:
            h = deg // 2
            ei = np.concatenate([np.arange(n - k) for k in range(1, h + 1)]); ej = np.concatenate([np.arange(k, n) for k in range(1, h + 1)])
        else:
            H = tl.gen_vg_hubs(n, deg, 50, 0.25, 0.05, seed=n); ei, ej = H["ei"], H["ej"]
        Rs = tl.haar_so3(rng0, n)
        M = Rs[ei] @ tl.so3_exp(rng0.standard_normal((ei.size, 3)) * 0.05) @ np.transpose(Rs[ej], (0, 2, 1))
        rp, ci, bl = tl.vg_from_edges(n, ei, ej, np.ones(ei.size), M)
        P = dict(rowptr=rp, colidx=ci, blocks=bl)
    elif a.band:
        h = deg // 2
        lo = np.maximum(np.arange(n) - h, 0); hi = np.minimum(np.arange(n) + h, n - 1)
        cnt = hi - lo + 1
        rowptr = np.zeros(n + 1, dtype=np.int64); rowptr[1:] = np.cumsum(cnt)
        colidx = (np.repeat(lo, cnt) + (np.arange(rowptr[-1]) - np.repeat(rowptr[:-1], cnt))).astype(np.int32)
        P = dict(rowptr=rowptr, colidx=colidx, blocks=np.random.default_rng(n).standard_normal((rowptr[-1], 3, 3)))
    elif a.skew:
        P = tl.gen_skewed(n, deg, seed=n)
    else:
        P = tl.gen_vg(n, deg=deg, sigma=0.05, seed=n, dense=False)
    np.savez(tag, rowptr=P["rowptr"], colidx=P["colidx"], blocks=P["blocks"])
nb_full = P["colidx"].size
if a.upper:
    rows = np.repeat(np.arange(n), np.diff(P["rowptr"]))
    keep = P["colidx"] >= rows
    rp = np.zeros(n + 1, dtype=np.int64); rp[1:] = np.cumsum(np.bincount(rows[keep], minlength=n))
    P = dict(rowptr=rp, colidx=P["colidx"][keep].copy(), blocks=P["blocks"].reshape(-1, 3, 3)[keep].copy())
    print(f"UPPER TRIANGLE ONLY: {P['colidx'].size} of {nb_full} blocks kept; bytes / fractions below stay in FULL-storage accounting of the whole matrix", flush=True)
nb = nb_full
L = xmamd.lib()
rng = np.random.default_rng(0)
drp = dci = dbl = None
if not a.no_csr:
    drp = xmamd.DevArray(P["rowptr"]); dci = xmamd.DevArray(P["colidx"]); dbl = xmamd.DevArray(P["blocks"].reshape(-1))
mats = {}
for o in a.o:
    OP = o | 1
    Wh = rng.standard_normal((3 * n, o))
    dW = xmamd.DevArray(xmamd.to_rm(Wh)); dO = xmamd.DevArray(nbytes=3 * n * OP * 8)
    dP = xmamd.DevArray(xmamd.pad16(Wh)) if (a.padded and 3 <= o <= 5) else None
    by = 76.0 * nb + 4 * (n + 1) + 2 * 8 * 3 * n * o
    ref = None
    ms = C.c_double()
    if not a.no_csr:
        xmamd._chk(L.xm_qw_bsr3_time(drp.ptr, dci.ptr, dbl.ptr, n, o, dW.ptr, dO.ptr, a.reps, C.byref(ms)))
        print(f"CSR  n={n} deg={deg} nb={nb} o={o}: {ms.value*1e3:8.1f} us  {by/ms.value/1e6:8.1f} GB/s algorithmic ({by/1e6:.1f} MB)", flush=True)
        if a.check:
            ref = xmamd.from_rm(dO.get(), 3 * n, o)
    for S in [(S, cd) for S in a.slabs for cd in a.codec]:
        S, cd = S
        if (S, cd) not in mats:
            mats[(S, cd)] = xmamd.SellMatrix(P["rowptr"], P["colidx"], P["blocks"], slabs=S, lmax=a.lmax, codec=cd)
        for gm in a.gather:
            if o == 1 and gm == 1:
                continue
            xmamd._chk(L.xm_qw_sell_time(mats[(S, cd)].h, o, dW.ptr, dP.ptr if dP is not None else None, dO.ptr, gm, a.reps, C.byref(ms)))
            line = f"SELL n={n} deg={deg} nb={nb} o={o} slabs={S} gather={gm} codec={cd}{' padded-W' if dP is not None else ''}{' UPPER-ONLY' if a.upper else ''}: {ms.value*1e3:8.1f} us  {by/ms.value/1e6:8.1f} GB/s in full-storage accounting = {by/ms.value/1e6/8000:.3f} of 8 TB/s"
            if ref is not None:
                got = xmamd.from_rm(dO.get(), 3 * n, o)
                line += f"   rel.err vs CSR kernel {tl.rel_fro(got, ref):.2e}"
            print(line, flush=True)
    dW.free(); dO.free()
    if dP is not None: dP.free()
