"""CPU tests of the multi-GPU design: the row-partitioned algorithm — local rows of Q, all-gathered product
input, gathered-not-reduced partial sums, inert padding cameras — run as a numpy model (tests/dist_model.py) single
process and under torch.distributed/gloo with world_size 2, against the single-process CPU oracle."""
import os
import socket
import sys

import numpy as np
import pytest

import xm_testlib as tl
from dist_model import RankModel

HERE = os.path.dirname(os.path.abspath(__file__))


def _problem():
    P = tl.gen_vg(61, deg=6, sigma=0.2, seed=61)     # odd camera count -> the last rank owns a padding camera
    return P["Q"], 6.0


def test_partition_model_world1_matches_oracle(oracle):
    Q, lam = _problem()
    n = Q.shape[0] // 3
    R0 = np.tile(np.eye(3), (n, 1)); s0 = np.ones(n)
    m = RankModel(Q, 3, lam, 0, 1, lambda v: v)
    R, s, info = m.trust_region(R0, s0, 1e-9)
    Ro, so, primal, _, st = oracle.trustregion(Q, R0, s0, lam=lam, gradtol=1e-9, trace=2000)
    assert info["primal"] == pytest.approx(primal, rel=1e-10)
    assert tl.rotation_parity(R, s, Ro, so) < 1e-8
    tr = st["trace"]
    k = min(6, len(tr), len(info["trace"]))
    assert np.allclose(info["trace"][:k, 0], tr[:k, 0], rtol=1e-9) and np.allclose(info["trace"][:k, 1], tr[:k, 1], rtol=1e-7)
    assert abs(info["tcg_iters"] - st["tcg_iters"]) <= 0.05 * st["tcg_iters"] + 5


def _hub_problem():
    """view graph with three hub cameras (each sees 40 % of the others): rows of very different length"""
    H = tl.gen_vg_hubs(61, 5, 3, 0.4, 0.2, seed=7)
    rp, ci, bl = tl.vg_from_edges(61, H["ei"], H["ej"], H["w"], H["M"])
    return tl.bsr_to_dense(61, rp, ci, bl), rp, 6.0


def _block_cuts(rp, n, world):
    """the partition xm_partition_blocks / Context::init computes for block-sparse storage (partition_cuts in xm_solver.hip)"""
    cuts, c = [0], 0
    for r in range(1, world):
        target = rp[0] + (rp[n] - rp[0]) * r // world
        while c < n and rp[c] < target:
            c += 1
        cuts.append(c)
    return cuts + [n]


def _worker(rank, world, port, out_dir, overlap=False, hubs=False):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def allgather(v):
        t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64))
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        return np.concatenate([o.numpy() for o in outs])

    cuts = None
    if hubs:
        Q, rp, lam = _hub_problem()
        cuts = _block_cuts(rp, Q.shape[0] // 3, world)
    else:
        Q, lam = _problem()
    n = Q.shape[0] // 3
    m = RankModel(Q, 3, lam, rank, world, allgather, overlap=overlap, cuts=cuts)
    R, s, info = m.trust_region(np.tile(np.eye(3), (n, 1)), np.ones(n), 1e-9)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), R=R, s=s, primal=info["primal"], tcg=info["tcg_iters"],
             trace=info["trace"], nloc=m.nloc, cam0=m.cam0, strip_first=bool(m.events[:2] == ["strip", "gather"]) if overlap else False)
    dist.barrier()
    dist.destroy_process_group()


def test_partition_model_world3_gloo(tmp_path):
    """61 cameras over 3 ranks: 22 + 22 + 17 (even ranges, xm_solver.hip:equal_range_len; +5 padding cameras on the last rank); every rank
    ends bit-identical"""
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    r = [np.load(tmp_path / f"rank{k}.npz") for k in range(3)]
    assert [int(x["nloc"]) for x in r] == [22, 22, 22] and [int(x["cam0"]) for x in r] == [0, 22, 44]
    for x in r[1:]:
        assert np.array_equal(r[0]["R"], x["R"]) and np.array_equal(r[0]["s"], x["s"]) and np.array_equal(r[0]["trace"], x["trace"])
    Q, lam = _problem()
    n = Q.shape[0] // 3
    R1, s1, i1 = RankModel(Q, 3, lam, 0, 1, lambda v: v).trust_region(np.tile(np.eye(3), (n, 1)), np.ones(n), 1e-9)
    assert float(r[0]["primal"]) == pytest.approx(i1["primal"], rel=1e-11)
    assert tl.rotation_parity(r[0]["R"], r[0]["s"], R1, s1) < 1e-8


def test_block_balanced_partition_world3_gloo(xmamd, oracle, tmp_path):
    """SURVEY 8e "balanced by stored blocks": a hub-camera view graph over 3 ranks with the partition the library computes
    (xm_partition_blocks) -- UNEQUAL camera ranges, every rank padded to the longest one, all replicated vectors in the padded numbering
    (rank * nloc + local index).  Every rank ends bit-identical and at the single-rank / oracle optimum."""
    import ctypes
    import torch.multiprocessing as mp
    Q, rp, lam = _hub_problem()
    n = Q.shape[0] // 3
    cuts = _block_cuts(rp, n, 3)
    lib_cuts = [0]
    for r in range(3):
        c0, c1 = ctypes.c_int64(), ctypes.c_int64()
        assert xmamd.lib().xm_partition_blocks(n, rp.ctypes.data_as(ctypes.c_void_p), 3, r, ctypes.byref(c0), ctypes.byref(c1)) == 0
        assert c0.value == lib_cuts[-1]
        lib_cuts.append(c1.value)
    assert lib_cuts == cuts                                                   # the model uses the library's partition
    sizes = [b - a for a, b in zip(cuts, cuts[1:])]
    assert max(sizes) - min(sizes) >= 5, sizes                                # really unequal: padding cameras in the MIDDLE of the numbering
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker, args=(3, port, str(tmp_path), False, True), nprocs=3, join=True)
    r = [np.load(tmp_path / f"rank{k}.npz") for k in range(3)]
    assert [int(x["nloc"]) for x in r] == [max(sizes)] * 3
    for x in r[1:]:
        assert np.array_equal(r[0]["R"], x["R"]) and np.array_equal(r[0]["s"], x["s"]) and np.array_equal(r[0]["trace"], x["trace"])
    R0 = np.tile(np.eye(3), (n, 1)); s0 = np.ones(n)
    R1, s1, i1 = RankModel(Q, 3, lam, 0, 1, lambda v: v).trust_region(R0, s0, 1e-9)
    assert float(r[0]["primal"]) == pytest.approx(i1["primal"], rel=1e-11)
    assert tl.rotation_parity(r[0]["R"], r[0]["s"], R1, s1) < 1e-8
    Ro, so, primal, _, st = oracle.trustregion(Q, R0, s0, lam=lam, gradtol=1e-9)
    assert float(r[0]["primal"]) == pytest.approx(primal, rel=1e-10)
    assert tl.rotation_parity(r[0]["R"], r[0]["s"], Ro, so) < 1e-8


def test_partition_model_world2_gloo(oracle, tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "rank0.npz"); b = np.load(tmp_path / "rank1.npz")
    assert int(a["nloc"]) == 32 and int(b["cam0"]) == 32          # 61 cameras -> 32 + 29 (+3 padding cameras): even ranges (equal_range_len)
    # every rank must hold the identical result and trajectory (gathered partials are added in the same order)
    assert np.array_equal(a["R"], b["R"]) and np.array_equal(a["s"], b["s"]) and np.array_equal(a["trace"], b["trace"])
    Q, lam = _problem()
    n = Q.shape[0] // 3
    R0 = np.tile(np.eye(3), (n, 1)); s0 = np.ones(n)
    R1, s1, i1 = RankModel(Q, 3, lam, 0, 1, lambda v: v).trust_region(R0, s0, 1e-9)
    assert float(a["primal"]) == pytest.approx(i1["primal"], rel=1e-11)
    assert tl.rotation_parity(a["R"], a["s"], R1, s1) < 1e-8
    Ro, so, primal, _, st = oracle.trustregion(Q, R0, s0, lam=lam, gradtol=1e-9)
    assert float(a["primal"]) == pytest.approx(primal, rel=1e-10)
    assert tl.rotation_parity(a["R"], a["s"], Ro, so) < 1e-8


def test_partition_model_overlap_world2_gloo(tmp_path):
    """SURVEY 8e on the model: outside the tCG each rank multiplies its own column strip BEFORE the all-gather of W returns and adds
    the other columns afterwards; both ranks stay bit-identical with each other and land on the optimum of the plain schedule"""
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path), True), nprocs=2, join=True)
    a = np.load(tmp_path / "rank0.npz"); b = np.load(tmp_path / "rank1.npz")
    assert bool(a["strip_first"]) and bool(b["strip_first"])
    assert np.array_equal(a["R"], b["R"]) and np.array_equal(a["s"], b["s"]) and np.array_equal(a["trace"], b["trace"])
    Q, lam = _problem()
    n = Q.shape[0] // 3
    R1, s1, i1 = RankModel(Q, 3, lam, 0, 1, lambda v: v).trust_region(np.tile(np.eye(3), (n, 1)), np.ones(n), 1e-9)
    assert float(a["primal"]) == pytest.approx(i1["primal"], rel=1e-11)
    assert tl.rotation_parity(a["R"], a["s"], R1, s1) < 1e-8
