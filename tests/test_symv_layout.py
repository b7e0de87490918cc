"""CPU emulation of the half-traffic symmetric dense product (xm_kernels.hip: qw_symv_kernel + symv_reduce_kernel): the strip / chunk /
diagonal-mask rules and the reducer's list lengths, with the chunk plan the library computes (xm_symv_plan, host only).  The sum of the
partial results the reducer reads must be Q @ W for a symmetric Q, every entry it reads must have been written by exactly one
wavefront, and nothing may depend on the lower triangle."""
import ctypes as C, os, sys
import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "xm-code_amd"))
import xmamd

STRIP = 256


def emulate(Q, W, K, Kf, ysplit, rev=0):
    """ysplit counts ROWS of the folded grid (row y = strip y and strip S - 1 - y): the strips of rows >= ysplit are cut finer.  A workgroup =
    one strip x 4 Ks steps, its four wavefronts' column sums are added before the one record per workgroup is written; rev walks every chunk
    bottom-up.  The grid is walked as the kernel maps it: block (x, y) -> strip y while x < groups(y), else strip S - 1 - y."""
    m, o = W.shape
    n = m // 3
    ld = xmamd.dense_ld(n)
    nsteps, nstrips = (n + 1) // 2, (ld + STRIP - 1) // STRIP
    Qp = np.zeros((6 * nsteps, ld)); Qp[:m, :m] = Q
    Wp = np.zeros((max(ld, 6 * nsteps), o)); Wp[:m] = W
    nsc = (nsteps + 4 * Kf - 1) // (4 * Kf)
    Prow = np.full((nstrips, 6 * nsteps, o), np.nan); Pcol = np.full((nsc, ld, o), np.nan)
    def groups(st, k):
        return (min(nsteps, (st * STRIP + STRIP + 5) // 6) + 4 * k - 1) // (4 * k)
    gy = (nstrips + 1) // 2
    gx = max(groups(y, Kf if y >= ysplit else K) + (groups(nstrips - 1 - y, Kf if y >= ysplit else K) if nstrips - 1 - y != y else 0) for y in range(gy))
    seen = set()
    for y in range(gy):
        Ks = Kf if y >= ysplit else K
        for x in range(gx):
            s, sc = y, x
            nA = groups(y, Ks)
            if sc >= nA:
                if nstrips - 1 - y == y:
                    continue
                s, sc = nstrips - 1 - y, sc - nA
            c0 = s * STRIP
            cols = np.arange(c0, min(c0 + STRIP, ld))
            jend = min(nsteps, (c0 + STRIP + 5) // 6)
            if sc * 4 * Ks >= jend:
                continue
            assert (s, sc) not in seen; seen.add((s, sc))
            tot = np.zeros((cols.size, o))
            for wave in range(4):
                jb = (sc * 4 + wave) * Ks; je = min(jb + Ks, jend)
                acc = np.zeros((cols.size, o))
                steps = range(jb, je)
                for j in (reversed(steps) if rev else steps):
                    rows = np.arange(6 * j, 6 * j + 6)
                    q = Qp[np.ix_(rows, cols)]
                    assert np.all(np.isnan(Prow[s, rows]))                                  # written once
                    Prow[s, rows] = (q * (cols >= 6 * j)[None, :]) @ Wp[cols]              # used row-wise from the diagonal block on
                    acc += (q * (cols >= 6 * j + 6)[None, :]).T @ Wp[rows]                 # and column-wise strictly right of it
                tot += acc
            assert np.all(np.isnan(Pcol[sc, cols]))                                         # written once
            Pcol[sc, cols] = tot
    Y = np.zeros((m, o))
    for cam in range(n):
        s_lo = (6 * (cam // 2)) // STRIP
        for r in range(3):
            c = 3 * cam + r
            st = c // STRIP
            Kc = 4 * (Kf if min(st, nstrips - 1 - st) >= ysplit else K)                      # steps per column-sum record
            cnt = (c - 6) // (6 * Kc) + 1 if c >= 6 else 0
            Y[c] = Prow[s_lo:, c].sum(axis=0) + Pcol[:cnt, c].sum(axis=0)
    assert np.all(np.isfinite(Y))                                                        # everything read had been written
    return Y


@pytest.mark.parametrize("n,o,plan", [(1, 3, None), (7, 3, None), (43, 3, None), (86, 4, None), (171, 5, None), (700, 3, None),
                                      (700, 3, (16, 4, 2)), (700, 4, (32, 8, 1)), (1031, 3, (16, 4, 4)), (1031, 3, (5, 5, 99))])
def test_symv_partition_sums_to_the_product(n, o, plan):
    rng = np.random.default_rng(n + o)
    A = rng.standard_normal((3 * n, 3 * n)); Q = A + A.T
    W = rng.standard_normal((3 * n, o))
    if plan is None:
        p = (C.c_int32 * 4)()
        assert xmamd.lib().xm_symv_plan(n, p) == 0
        K, Kf, ysplit, nch = (int(x) for x in p)
        assert 2 <= K <= 64 and 1 <= Kf <= K and nch == ((n + 1) // 2 + 4 * Kf - 1) // (4 * Kf)
    else:
        K, Kf, ysplit = plan
    Y = emulate(Q, W, K, Kf, ysplit)
    ref = Q @ W
    assert np.abs(Y - ref).max() <= 1e-11 * np.abs(ref).max()
    Ql = Q.copy(); Ql[np.tril_indices(3 * n, -6)] = 1e6            # far enough below the diagonal nothing may be read
    assert np.array_equal(emulate(Ql, W, K, Kf, ysplit), Y)
    Yr = emulate(Q, W, K, Kf, ysplit, rev=1)                       # the bottom-up sweep reads and writes the same entries
    assert np.abs(Yr - ref).max() <= 1e-11 * np.abs(ref).max()


def test_symv_plan_of_the_benchmark_sizes():
    for n, kmin, kmax in ((1778, 4, 12), (4096, 12, 32), (8192, 16, 48), (13682, 32, 64)):
        p = (C.c_int32 * 4)()
        assert xmamd.lib().xm_symv_plan(n, p) == 0
        assert kmin <= p[0] <= kmax and p[1] in (p[0], p[0] // 4) and (p[1] == p[0] // 4 if n >= 8192 else p[1] == p[0])
