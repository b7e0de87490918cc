"""CPU tests of the sliced-ELL layout builder (xm-code_amd/csrc/xm_sell.hip:sell_build_host): the description is turned
into the device arrays and multiplied in numpy exactly as sell_fill_kernel / qw_sell_kernel / sell_reduce_kernel do it
(same index arithmetic), and must reproduce the block-CSR product.  No GPU, no oracle."""
import numpy as np
import pytest

import xm_testlib as tl


def _emulate(L, colidx, blocks, W, o):
    """numpy restatement of the three device kernels on the host description L"""
    nst = L["nsteps"]
    cols = np.zeros(max(nst, 1) * 64, dtype=np.int64)
    blk = np.zeros(max(nst, 1) * 576)
    B9 = blocks.reshape(-1, 9)
    for g in range(nst):                                   # sell_fill_kernel
        kd = int(L["kind"][g])
        for lane in range(64):
            s = int(L["src"][g * 64 + lane])
            c = 0 if s < 0 else int(colidx[s])
            q = np.zeros(9) if s < 0 else B9[s]
            if kd == 2:
                cols[g * 64 + lane] = c
                blk[g * 576 + np.arange(9) * 64 + lane] = q
            else:
                gb = g - kd
                cols[gb * 64 + lane * 2 + kd] = c
                blk[gb * 576 + np.arange(9) * 128 + lane * 2 + kd] = q
    parts = np.zeros((max(L["nstore"], 1), 3, o))
    seen = np.zeros(max(L["nstore"], 1), dtype=int)
    Wc = W.reshape(-1, 3, o)
    for c in range(L["nslices"]):                          # qw_sell_kernel: wave = slice, lane = virtual row
        off = int(L["slice_off"][c]); w = int(L["slice_off"][c + 1]) - off
        npair, tail = w >> 1, w & 1
        for lane in range(64):
            acc = np.zeros((3, o))
            for p in range(npair):
                for h in range(2):
                    j = cols[off * 64 + p * 128 + 2 * lane + h]
                    q = blk[off * 576 + p * 1152 + np.arange(9) * 128 + 2 * lane + h].reshape(3, 3)
                    acc += q @ Wc[j]
            if tail:
                j = cols[off * 64 + npair * 128 + lane]
                q = blk[off * 576 + npair * 1152 + np.arange(9) * 64 + lane].reshape(3, 3)
                acc += q @ Wc[j]
            slot = int(L["pslot"][c * 64 + lane])
            if slot >= 0:
                parts[slot] = acc
                seen[slot] += 1
    assert seen.sum() == L["nparts"] and seen.max() <= 1 and np.all(seen[L["ridx"][: L["nparts"]]] == 1)   # every listed partial result is written exactly once
    n = L["pptr"].size - 1
    out = np.zeros((n, 3, o))
    for r in range(n):                                     # sell_reduce_kernel
        out[r] = parts[L["ridx"][L["pptr"][r]:L["pptr"][r + 1]]].sum(axis=0)
    return out.reshape(3 * n, o)


@pytest.mark.parametrize("n,deg,slabs,lmax,o", [(1, 2, 1, 64, 3), (7, 3, 4, 64, 3), (150, 9, 8, 64, 3), (300, 30, 4, 8, 4), (200, 12, 2, 3, 5),
                                                 (97, 20, 1, 64, 1)])
def test_sell_layout_reproduces_bsr_product(xmamd, n, deg, slabs, lmax, o):
    P = tl.gen_vg(n, deg=deg, sigma=0.3, seed=n + o, dense=True)
    L = xmamd.sell_layout(P["rowptr"], P["colidx"], slabs=slabs, lmax=lmax)
    W = np.random.default_rng(n).standard_normal((3 * n, o))
    got = _emulate(L, P["colidx"], P["blocks"], W, o)
    assert tl.rel_fro(got, P["Q"] @ W) < 1e-13
    # structure: slices sorted by width inside a slab, slabs partition the slices, every block used exactly once
    used = L["src"][L["src"] >= 0]
    assert used.size == P["colidx"].size and np.unique(used).size == used.size
    assert L["slab_start"][0] == 0 and L["slab_start"][-1] == L["nslices"]
    wd = np.diff(L["slice_off"])
    for s in range(slabs):
        ws = wd[L["slab_start"][s]:L["slab_start"][s + 1]]
        assert np.all(np.diff(ws) <= 0) and (ws.size == 0 or ws.max() <= lmax)
    # column slabs: every block of slab s has its column in that slab's range
    for s in range(slabs):
        for c in range(L["slab_start"][s], L["slab_start"][s + 1]):
            src = L["src"][L["slice_off"][c] * 64:L["slice_off"][c + 1] * 64]
            cols = P["colidx"][src[src >= 0]]
            assert np.all(cols.astype(np.int64) * slabs // n == s)


def test_sell_layout_unsorted_rows_hub_and_empty_rows(xmamd):
    """rows given in arbitrary column order, a hub camera far longer than lmax, cameras without any block"""
    rng = np.random.default_rng(5)
    n = 120
    rows = [rng.choice(n, size=k, replace=False) for k in rng.integers(0, 7, size=n)]
    rows[17] = rng.permutation(n)                                    # hub: sees everybody
    rows[3] = np.array([], dtype=int); rows[n - 1] = np.array([], dtype=int)
    rowptr = np.zeros(n + 1, dtype=np.int64); rowptr[1:] = np.cumsum([len(r) for r in rows])
    colidx = np.concatenate(rows).astype(np.int32)
    blocks = rng.standard_normal((colidx.size, 3, 3))
    L = xmamd.sell_layout(rowptr, colidx, slabs=4, lmax=16)
    W = rng.standard_normal((3 * n, 3))
    ref = tl.bsr_to_dense(n, rowptr, colidx, blocks) @ W
    assert tl.rel_fro(_emulate(L, colidx, blocks, W, 3), ref) < 1e-13
    cnt = np.diff(L["pptr"])                                   # partial results per camera
    assert cnt[3] == 0 and cnt[n - 1] == 0 and cnt[17] >= n // 16 and L["pptr"][-1] == L["nparts"]


def test_sell_layout_rejects_malformed_input(xmamd):
    with pytest.raises(xmamd.XmError):
        xmamd.sell_layout(np.array([0, 2, 1]), np.array([0, 1], dtype=np.int32))           # rowptr not monotone
    with pytest.raises(xmamd.XmError):
        xmamd.sell_layout(np.array([0, 1, 2]), np.array([0, 5], dtype=np.int32))           # column out of range
    with pytest.raises(xmamd.XmError):
        xmamd.sell_layout(np.array([0, 1, 2]), np.array([0, 1], dtype=np.int32), slabs=3)  # slabs must divide 8


def test_column_locality_decides_the_padded_copy(xmamd):
    """xm_tuning_t.sell_wpad = 0 (auto): the copy of W at the 128-byte record pitch is used when the lanes of a step touch fewer lines
    with it than at the native pitch -- a random view graph (1.4 / 1.9 lines per 72- / 120-byte record against 1) yes, a banded graph
    (neighbouring columns share lines: 0.56 / 0.94 against 1) no"""
    n = 4000
    P = tl.gen_vg(n, deg=30, sigma=0.1, seed=3, dense=False)
    l72, l120, lp = xmamd.sell_locality(P["rowptr"], P["colidx"], slabs=4)
    assert lp * 1.25 < l72 < lp * 1.6 and lp * 1.7 < l120 < lp * 2.1
    h = 15
    lo = np.maximum(np.arange(n) - h, 0); hi = np.minimum(np.arange(n) + h, n - 1)
    cnt = hi - lo + 1
    rowptr = np.zeros(n + 1, dtype=np.int64); rowptr[1:] = np.cumsum(cnt)
    colidx = (np.repeat(lo, cnt) + (np.arange(rowptr[-1]) - np.repeat(rowptr[:-1], cnt))).astype(np.int32)
    l72, l120, lp = xmamd.sell_locality(rowptr, colidx, slabs=4)
    assert l72 < 0.75 * lp and l120 < 1.05 * lp          # the rule (10 % margin) keeps the native pitch for both record sizes


@pytest.mark.parametrize("o", [3, 4, 5])
def test_padded_record_layout(xmamd, o):
    """xmamd.pad16 = what tcg_init / cg_step write beside W: camera c's 3 x pitch_of(o) row-major record at [16 c, 16 c + 3 pitch_of(o)),
    zeros behind it -- element (3 c + r, k) of W sits at 16 c + r * pitch_of(o) + k"""
    n = 7
    W = np.random.default_rng(o).standard_normal((3 * n, o))
    P = xmamd.pad16(W).reshape(n, 16)
    OP = xmamd.pitch_of(o)
    for c in range(n):
        for r in range(3):
            assert np.array_equal(P[c, r * OP:r * OP + o], W[3 * c + r])
            assert np.all(P[c, r * OP + o:(r + 1) * OP] == 0.0)
        assert np.all(P[c, 3 * OP:] == 0.0)
