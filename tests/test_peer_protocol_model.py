"""Explicit-state model check of the direct peer exchange (xm-code_amd/csrc/xm_kernels.hip: cg_step_kernel, PeerXchg;
xm_comm.hip: peer_push_kernel / peer_wait_kernel) -- the one piece of the multi-GPU path that cannot be exercised on more than one
physical GPU here.  The model keeps exactly what the kernels rely on and nothing else:

  per iteration i (parity p = i mod 2, epoch e = base + i + 1) every rank, in program order,
    W   writes its own chunk of iteration i into its own buffer[p]            (Hessian epilogue / previous launch)
    P_q stores that chunk into buffer[p] of every peer q                        (one step per peer: the stores of different ranks interleave)
    F_q stores the epoch into flag[p][me] of every peer q                       (only after ALL its P steps: write-through stores + s_waitcnt,
                                                                                 or the release fence of XM_EXCHANGE_LITE=0)
    A_q waits until its own flag[p][q] >= e for every peer q
    R_q reads peer q's chunk from its own buffer[p]                             (one step per peer: the reads take time)
  and every interleaving of these steps over all ranks is explored (breadth first, states memoised).

Checked: every R_q returns the chunk of iteration i of rank q -- i.e. nobody overwrites a parity buffer before all its readers are
done, with TWO buffers and no other synchronisation ("rank A can push iteration i + 2 into parity i mod 2 only after
its cg_step(i + 1) saw B's epoch i + 1, which B publishes after its cg_step(i) has finished") -- and no interleaving deadlocks.
Negative controls: with ONE buffer, or with the flag stored before the payload, the checker must find the violation (it does)."""
from collections import deque

import pytest


def _program(world, me, iters, nbuf, flag_first):
    """the step list of one rank: tuples (op, iteration, peer)"""
    steps = []
    peers = [q for q in range(world) if q != me]
    for i in range(iters):
        steps.append(("W", i, me))
        push = [("P", i, q) for q in peers]
        flag = [("F", i, q) for q in peers]
        steps += (flag + push) if flag_first else (push + flag)
        steps += [("A", i, q) for q in peers]
        steps += [("R", i, q) for q in peers]
    return steps


def _check(world, iters, nbuf=2, flag_first=False):
    """returns None if every interleaving is safe and live, else a description of the first violation found"""
    progs = [_program(world, r, iters, nbuf, flag_first) for r in range(world)]
    # state: (pcs, buf, flag); buf[r][b][src] = iteration whose chunk of `src` sits in buffer b of rank r (-1: nothing yet);
    # flag[r][b][src] = last epoch `src` published to rank r for buffer b (0: none); epoch of iteration i = i + 1
    buf0 = tuple(tuple(tuple(-1 for _ in range(world)) for _ in range(nbuf)) for _ in range(world))
    flag0 = tuple(tuple(tuple(0 for _ in range(world)) for _ in range(nbuf)) for _ in range(world))
    start = (tuple(0 for _ in range(world)), buf0, flag0)
    seen = {start}
    todo = deque([start])

    def put(t, r, b, src, v):
        row = list(t[r][b]); row[src] = v
        plane = list(t[r]); plane[b] = tuple(row)
        out = list(t); out[r] = tuple(plane)
        return tuple(out)

    while todo:
        pcs, buf, flag = todo.popleft()
        moved = False
        done = True
        for r in range(world):
            if pcs[r] >= len(progs[r]):
                continue
            done = False
            op, i, q = progs[r][pcs[r]]
            b = i % nbuf
            nbuf_, nflag = buf, flag
            if op == "W":
                nbuf_ = put(buf, r, b, r, i)
            elif op == "P":
                nbuf_ = put(buf, q, b, r, buf[r][b][r])          # what the rank's own buffer holds NOW travels
            elif op == "F":
                nflag = put(flag, q, b, r, i + 1)
            elif op == "A":
                if flag[r][b][q] < i + 1:
                    continue                                        # blocked in its bounded wait
            elif op == "R":
                if buf[r][b][q] != i:
                    return f"rank {r} reads the chunk of rank {q} for iteration {i} and finds iteration {buf[r][b][q]}"
            moved = True
            npcs = list(pcs); npcs[r] += 1
            st = (tuple(npcs), nbuf_, nflag)
            if st not in seen:
                seen.add(st)
                todo.append(st)
        if not done and not moved:
            return f"deadlock at program counters {pcs}"
    return None


@pytest.mark.parametrize("world,iters", [(2, 8), (3, 4), (4, 3)])
def test_two_parity_buffers_and_one_epoch_flag_per_source_are_enough(world, iters):
    assert _check(world, iters) is None


def test_checker_finds_the_overwrite_with_a_single_buffer():
    msg = _check(2, 3, nbuf=1)
    assert msg is not None and "finds iteration" in msg


def test_checker_finds_the_missing_release_order():
    msg = _check(2, 2, flag_first=True)
    assert msg is not None and "finds iteration" in msg
