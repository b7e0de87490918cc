"""CPU tests of the multi-rank symmetric window product's index arithmetic (xm-code_amd/csrc/xm_symw.h): the cyclic half window uses exactly
one block of every mirror pair, the ranks' work lists cover exactly the (strip, step) pairs the predicate names, every rank streams half of its
strip, and a numpy restatement of the sweep + all-gather + per-camera sum reproduces Q @ W for a symmetric Q.  No GPU, no oracle."""
import numpy as np
import pytest


@pytest.mark.parametrize("T", [1, 2, 3, 4, 7, 8, 21, 22])
def test_window_uses_exactly_one_block_of_every_mirror_pair(xmamd, T):
    use = np.array([[xmamd.lib().xm_symw_use(T, t, u) for u in range(T)] for t in range(T)])
    assert not use.diagonal().any()
    off = ~np.eye(T, dtype=bool)
    assert np.all((use + use.T)[off] == 1)
    # balance: every row step uses floor or ceil of half of the others
    assert set(use.sum(axis=1).tolist()) <= {(T - 1) // 2, T // 2}


def _any(g, t, s):
    lo, hi = (s * 256) // 6, min((s * 256 + 255) // 6, g["T"] - 1)
    if lo > g["T"] - 1:
        return False
    dlo, dhi = (lo - t) % g["T"], (hi - t) % g["T"]
    return dlo > dhi or dlo <= g["Th"]


@pytest.mark.parametrize("ntot,world", [(44, 2), (176, 4), (1024, 8), (700, 2), (13696, 8)])
def test_work_lists_cover_the_window_and_balance(xmamd, ntot, world):
    nloc = ntot // world
    shares = []
    for r in range(world):
        g = xmamd.symw_plan(ntot, nloc, r * nloc)
        assert g["T"] == ntot // 2 and g["t0"] == r * nloc // 2 and g["nsteps"] == nloc // 2
        seen = set()
        last = (-1, -1)
        for s, jb, je in g["items"]:
            assert 0 <= jb < je <= g["nsteps"] and je - jb <= g["K"] and (s, jb) > last      # sorted by strip, then step; no overlap
            last = (s, je - 1)
            for j in range(jb, je):
                seen.add((int(s), j))
        if ntot <= 1024:
            want = {(s, j) for s in range(g["nstrips"]) for j in range(g["nsteps"]) if _any(g, g["t0"] + j, s)}
            assert seen == want
        shares.append(len(seen) * 256 / (g["nsteps"] * 6.0 * g["T"]))    # swept fraction of this rank's strip of Q
    # every rank sweeps half of its strip + the strips cut by the window's two edges; the ranks differ by a strip or two at most
    edge = 2.0 * 256 / (6.0 * (ntot // 2))
    assert max(shares) - min(shares) <= edge + 1e-12 and 0.5 <= min(shares) and max(shares) <= 0.5 + 2 * edge + 256 / (6.0 * (ntot // 2))


@pytest.mark.parametrize("ntot,world,o", [(44, 2, 3), (176, 4, 3), (132, 2, 1), (272, 8, 4)])
def test_window_product_model_reproduces_the_symmetric_product(xmamd, ntot, world, o):
    """numpy restatement of qw_symw_kernel (masks from symw_use, row sums per (strip, step), column sums per item), symw_colsum_kernel,
    the all-gather and symw_reduce_kernel on every rank of the partition: equals Q @ W for symmetric Q"""
    rng = np.random.default_rng(ntot)
    M = 3 * ntot
    A = rng.standard_normal((M, M)); Q = A + A.T
    W = rng.standard_normal((M, o))
    nloc = ntot // world
    lib = xmamd.lib()
    plans = [xmamd.symw_plan(ntot, nloc, r * nloc, K=3) for r in range(world)]
    T = plans[0]["T"]
    use = np.array([[lib.xm_symw_use(T, t, u) for u in range(T)] for t in range(T)], dtype=bool)
    csum = np.zeros((world, M, o)); prow = [dict() for _ in range(world)]
    for r, g in enumerate(plans):
        Qr = Q[3 * r * nloc:3 * (r + 1) * nloc]                      # this rank's strip
        for s, jb, je in g["items"]:
            c0, c1 = s * 256, min(s * 256 + 256, M)
            ucol = np.arange(c0, c1) // 6
            for j in range(jb, je):
                t = g["t0"] + j
                B = Qr[6 * j:6 * j + 6, c0:c1]
                mc = use[t, ucol]; mr = mc | (ucol == t)
                prow[r][(int(s), j)] = (B * mr) @ W[c0:c1]            # row direction -> this rank's rows
                csum[r, c0:c1] += (B * mc).T @ W[6 * t:6 * t + 6]     # column direction -> whoever owns those cameras
    out = np.zeros((M, o))
    for r, g in enumerate(plans):
        for cam in range(nloc):
            j, t = cam // 2, g["t0"] + cam // 2
            acc = np.zeros((3, o))
            for s in range(g["nstrips"]):
                if _any(g, t, s):
                    acc += prow[r][(s, j)][3 * (cam & 1):3 * (cam & 1) + 3]
            grow = 3 * (r * nloc + cam)
            for r2 in range(world):
                acc += csum[r2, grow:grow + 3]
            out[grow:grow + 3] = acc
    assert np.linalg.norm(out - Q @ W) <= 1e-12 * np.linalg.norm(Q @ W)


def _symw_rank_worker(rank, world, port, ntot, o, out_dir):
    """one rank of the window product under torch.distributed/gloo: it sees ONLY its own row strip of Q; the column sums travel by all_gather"""
    import os, sys
    import torch
    import torch.distributed as dist
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here); sys.path.insert(0, os.path.join(os.path.dirname(here), "xm-code_amd"))
    import xmamd
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(ntot)                      # the same seed everywhere: every rank could build Q, but keeps its strip only
    M = 3 * ntot
    A = rng.standard_normal((M, M)); Q = A + A.T
    W = rng.standard_normal((M, o))
    nloc = ntot // world
    Qr = Q[3 * rank * nloc:3 * (rank + 1) * nloc].copy(); del Q, A
    g = xmamd.symw_plan(ntot, nloc, rank * nloc, K=4)
    lib = xmamd.lib()
    T = g["T"]
    prow, csum = {}, np.zeros((M, o))
    for s, jb, je in g["items"]:
        c0, c1 = s * 256, min(s * 256 + 256, M)
        ucol = np.arange(c0, c1) // 6
        for j in range(jb, je):
            t = g["t0"] + j
            mc = np.array([bool(lib.xm_symw_use(T, t, int(u))) and u != t for u in ucol]); mr = mc | (ucol == t)
            B = Qr[6 * j:6 * j + 6, c0:c1]
            prow[(int(s), j)] = (B * mr) @ W[c0:c1]
            csum[c0:c1] += (B * mc).T @ W[6 * t:6 * t + 6]
    gathered = [torch.zeros(M, o, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(csum))       # the one collective of a product
    out = np.zeros((3 * nloc, o))
    for cam in range(nloc):
        j, t = cam // 2, g["t0"] + cam // 2
        acc = np.zeros((3, o))
        for s in range(g["nstrips"]):
            if _any(g, t, s):
                acc += prow[(s, j)][3 * (cam & 1):3 * (cam & 1) + 3]
        grow = 3 * (rank * nloc + cam)
        for r2 in range(world):                             # rank order: the same sum on every run
            acc += gathered[r2][grow:grow + 3].numpy()
        out[3 * cam:3 * cam + 3] = acc
    np.save(os.path.join(out_dir, f"y{rank}.npy"), out)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_window_product_under_gloo(xmamd, tmp_path, world):
    """world_size 2 and 3 over torch.distributed/gloo: every rank multiplies from its own strip alone, all_gathers the column sums and adds, per
    camera, its row sums and the ranks' column sums -- together the ranks reproduce Q @ W of the symmetric matrix"""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    ntot, o = 22 * world, 3                                  # 22 cameras per rank
    mp.spawn(_symw_rank_worker, args=(world, port, ntot, o, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(ntot)
    M = 3 * ntot
    A = rng.standard_normal((M, M)); Q = A + A.T
    W = rng.standard_normal((M, o))
    got = np.concatenate([np.load(str(tmp_path / f"y{r}.npy")) for r in range(world)], axis=0)
    assert np.linalg.norm(got - Q @ W) <= 1e-12 * np.linalg.norm(Q @ W)
