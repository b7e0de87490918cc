"""the device-driven outer iteration (XM_FLAG_DEVICE_OUTER; the default with block-CSR products) against the host-driven one (XM_FLAG_HOST_OUTER; the default with dense products): same problems solved both ways, results, iteration counts and time side by side
   python scripts/dev_outer_check.py [quick]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xm-code_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import xmamd, xm_testlib as tl
xmamd.require_gpu()
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"

def cmp(name, ctx, max_rank, tol, lam, reps=1, extra=0, **kw):
    out = []
    for fl in (extra | xmamd.FLAG_HOST_OUTER, extra | xmamd.FLAG_DEVICE_OUTER):
        ctx.solve(max_rank, tol, lam, flags=fl, **kw)   # warm-up
        t0 = time.perf_counter()
        for _ in range(reps):
            R, s, info = ctx.solve(max_rank, tol, lam, flags=fl, trace=4000, **kw)
        out.append((R, s, info, (time.perf_counter() - t0) / reps))
    (R0, s0, i0, t0), (R1, s1, i1, t1) = out
    same = (R0.shape == R1.shape) and np.array_equal(R0, R1) and np.array_equal(s0, s1)
    err = tl.rotation_parity(R0, s0, R1, s1) if R0.shape == R1.shape else float("nan")
    tr_same = i0["trace"].shape == i1["trace"].shape and np.array_equal(i0["trace"], i1["trace"])
    print(f"{name}:\n   host   rank {i0['rank']} status {i0['status']} tcg {i0['tcg_iters']} outer {i0['outer_iters']} primal {i0['primal']:.15g} qw {i0['qw_products']} stop {i0.get('last_stop_reason')} {t0 * 1e3:.2f} ms"
          f"\n   device rank {i1['rank']} status {i1['status']} tcg {i1['tcg_iters']} outer {i1['outer_iters']} primal {i1['primal']:.15g} qw {i1['qw_products']} stop {i1.get('last_stop_reason')} {t1 * 1e3:.2f} ms"
          f"\n   bit-identical solution {same}, identical trace {tr_same} ({i0['trace'].shape} / {i1['trace'].shape}), rotation parity {err:.2e}, "
          f"it/s {i0['tcg_iters'] / t0:.0f} -> {i1['tcg_iters'] / t1:.0f}", flush=True)
    if not tr_same and i0["trace"].shape[0] and i1["trace"].shape[0]:
        n = min(len(i0["trace"]), len(i1["trace"]))
        d = np.nonzero(np.any(i0["trace"][:n] != i1["trace"][:n], axis=1))[0]
        if d.size:
            j = d[0]; print("   first differing trace row", j, "\n   ", i0["trace"][j], "\n   ", i1["trace"][j])
    return same

P = tl.gen_vg(40, deg=3, sigma=1.5, seed=40)
cmp("vg40 dense (rank escalation)", xmamd.Context(Q=P["Q"]), 6, 1e-9, 3.0)
cmp("vg40 block CSR", xmamd.Context(bsr=(P["rowptr"], P["colidx"], P["blocks"])), 6, 1e-9, 3.0)
cmp("vg40 block CSR polar", xmamd.Context(bsr=(P["rowptr"], P["colidx"], P["blocks"])), 6, 1e-9, 3.0, retraction=xmamd.RETRACT_POLAR)
D = tl.gen_dense(356, seed=356)
cmp("dense356", xmamd.Context(Q=D["Q"]), 5, 1e-6, 0.0, reps=3)
cmp("dense356 grouping 2", xmamd.Context(Q=D["Q"]), 5, 1e-6, 0.0, grouping=2)
cmp("dense356 model recurrence", xmamd.Context(Q=D["Q"]), 5, 1e-6, 0.0, extra=xmamd.FLAG_MODEL_RECURRENCE)
V = tl.gen_vg(2000, deg=12, sigma=0.05, seed=7, dense=False)
cmp("vg2000 block CSR lam 1000", xmamd.Context(bsr=(V["rowptr"], V["colidx"], V["blocks"])), 5, 1e-6, 1000.0, reps=3)
if not quick:
    V = tl.gen_vg(13682, deg=30, sigma=0.05, seed=13682, dense=False)
    cmp("Final-13682 block CSR", xmamd.Context(bsr=(V["rowptr"], V["colidx"], V["blocks"])), 5, 1e-6, 1000.0, reps=5)
    D = tl.gen_dense(1778, seed=1778)
    cmp("Venice-1778 dense", xmamd.Context(Q=D["Q"]), 5, 1e-6, 0.0, reps=3)
