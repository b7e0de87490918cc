"""GPU tests of round 6's device-driven outer iteration (run with -m gpu on an MI355X), through the C ABI of libxm_amd.so.

The trust region of trustregion.h:416-718 has two loops.  Since round 1 the inner one (truncated CG) runs without the host; since round 6 the
outer one can too (xm_kernels.hip: outer_step_kernel, xm_solver.hip: Context::trust_region_device): the host enqueues one repeating pair of
launches and watches a progress word -- the default with block-CSR products, on request (XM_FLAG_DEVICE_OUTER) with dense ones, where the
host-driven form measures faster.  XM_FLAG_HOST_OUTER keeps the round-5 form, which is what these tests compare with: the two forms
evaluate the same formulas on the same numbers, so in block-CSR storage (no sweep direction to alternate) they must agree BIT FOR BIT --
solution, per-iteration trace, iteration counts, stop reason; dense products alternate their sweep direction by launch pair instead of by
tCG iteration, there the two agree like two summation groupings do (same certified optimum, rotations <= 1e-6)."""
import json
import os

import numpy as np
import pytest

import xm_testlib as tl

pytestmark = pytest.mark.gpu
G = tl.GOLDEN


def _both(ctx, xmamd, *a, flags=0, **kw):
    dev = ctx.solve(*a, flags=flags | xmamd.FLAG_DEVICE_OUTER, trace=2000, **kw)   # (dense products run it on request only)
    host = ctx.solve(*a, flags=flags | xmamd.FLAG_HOST_OUTER, trace=2000, **kw)
    return dev, host


def _identical(dev, host):
    (R1, s1, i1), (R0, s0, i0) = dev, host
    assert R1.shape == R0.shape and np.array_equal(R1, R0) and np.array_equal(s1, s0)
    assert i1["trace"].shape == i0["trace"].shape and np.array_equal(i1["trace"], i0["trace"])
    for k in ("rank", "status", "primal", "dual", "min_eig", "tcg_iters", "outer_iters", "last_stop_reason", "lanczos_iters"):
        assert i1[k] == i0[k], k
    # the host-driven loop counts its run-ahead launches that turned out to be no-ops as products, the device-driven one has none
    assert i1["qw_products"] <= i0["qw_products"]


@pytest.mark.parametrize("retraction", ["qr", "polar"])
@pytest.mark.parametrize("grouping", [0, 1, 2])
def test_device_outer_is_bit_identical_in_block_csr_storage(xmamd, retraction, grouping):
    """rank escalation 3 -> 6 (line search, certificate, Lanczos direction between the stages) on a 40-camera view graph in block CSR"""
    P = tl.gen_vg(40, deg=3, sigma=1.5, seed=40)
    ctx = xmamd.Context(bsr=(P["rowptr"], P["colidx"], P["blocks"]))
    retr = xmamd.RETRACT_POLAR if retraction == "polar" else xmamd.RETRACT_QR
    dev, host = _both(ctx, xmamd, 6, 1e-9, 3.0, retraction=retr, grouping=grouping)
    ctx.close()
    assert dev[2]["outer_on_device"] == dev[2]["rank"] - 2 and host[2]["outer_on_device"] == 0   # one trust region per rank level 3 .. rank
    assert dev[2]["rank"] == 6 and dev[2]["status"] == 1
    _identical(dev, host)


@pytest.mark.parametrize("n,deg,lam", [(300, 10, 10.0), (2000, 12, 1000.0), (5000, 20, 1000.0)])
def test_device_outer_block_csr_larger_graphs(xmamd, n, deg, lam):
    """several workgroups of the step launch, several rounds of partial sums, the model-decrease partials of more than one wavefront"""
    V = tl.gen_vg(n, deg=deg, sigma=0.05, seed=n + 1, dense=False)
    ctx = xmamd.Context(bsr=(V["rowptr"], V["colidx"], V["blocks"]), tuning=dict(sell=-1))
    dev, host = _both(ctx, xmamd, 5, 1e-8, lam)
    ctx.close()
    assert dev[2]["outer_on_device"] >= 1 and dev[2]["status"] == 1
    _identical(dev, host)


def test_device_outer_model_recurrence_and_stop_tests_in_block_csr(xmamd):
    """XM_FLAG_MODEL_RECURRENCE (the model value travels in the scalar block), a tolerance nobody reaches (the run ends by the residual
    test of the truncated CG, stop reason 5) and a time limit that has expired before the first iteration (stop reason 11)"""
    V = tl.gen_vg(500, deg=10, sigma=0.1, seed=11, dense=False)
    ctx = xmamd.Context(bsr=(V["rowptr"], V["colidx"], V["blocks"]))
    _identical(*_both(ctx, xmamd, 4, 1e-9, 50.0, flags=xmamd.FLAG_MODEL_RECURRENCE))
    dev, host = _both(ctx, xmamd, 3, 1e-30, 50.0)
    assert dev[2]["last_stop_reason"] == 5
    _identical(dev, host)
    dev, host = _both(ctx, xmamd, 3, 1e-9, 50.0, max_time=-1.0)
    assert dev[2]["last_stop_reason"] == 11 and dev[2]["tcg_iters"] == 0
    _identical(dev, host)
    ctx.close()


@pytest.mark.parametrize("name", ["simple1", "simple2", "synth/dense49", "synth/vg60_cert", "synth/vg40_stair"])
def test_device_outer_reaches_the_golden_optimum_in_dense_storage(xmamd, name):
    """the five golden cases through the dense kernels, device-driven (XM_FLAG_DEVICE_OUTER) and host-driven (the default for dense products): rank, status, certified optimum (1e-9 of the
    fixture), rotations within 1e-6 of each other; problems of one column tile have no sweep direction and agree bit for bit"""
    Q = tl.load_bin(os.path.join(G, name, "Q.bin")); exp = json.load(open(os.path.join(G, name, "expected.json")))
    ctx = xmamd.Context(Q=Q)
    dev, host = _both(ctx, xmamd, exp["max_rank"], exp["tol"], exp["lam"])
    ctx.close()
    (R1, s1, i1), (R0, s0, i0) = dev, host
    assert i1["outer_on_device"] >= 1 and i0["outer_on_device"] == 0
    assert i1["rank"] == i0["rank"] == exp["rank"] and i1["status"] == i0["status"] == exp["status"]
    assert i1["primal"] == pytest.approx(exp["f_star"], rel=1e-9) and i0["primal"] == pytest.approx(exp["f_star"], rel=1e-9)
    assert tl.rotation_parity(R1, s1, R0, s0) < 1e-6
    if Q.shape[0] <= 256:
        _identical(dev, host)


def test_device_outer_symmetric_pair_and_general_dense_kernel(xmamd):
    """Dubrovnik-356-size dense Q through the general kernel and, forced (sym_min_rows), through the half-traffic symmetric pair: both launches of
    the pair switch roles by the phase word; same certified optimum as the host-driven loop, rotations <= 1e-6"""
    D = tl.gen_dense(356, seed=356)
    for tuning in (None, dict(sym=1, sym_min_rows=256)):
        ctx = xmamd.Context(Q=D["Q"], tuning=tuning)
        dev, host = _both(ctx, xmamd, 5, 1e-6, 0.0)
        kind = ctx.product_kind(3)
        ctx.close()
        (R1, s1, i1), (R0, s0, i0) = dev, host
        assert i1["outer_on_device"] >= 1 and i1["sym_product"] == (1 if tuning else 0), kind
        assert i1["rank"] == i0["rank"] and i1["status"] == i0["status"] == 1
        assert i1["primal"] == pytest.approx(i0["primal"], rel=1e-9)
        assert tl.rotation_parity(R1, s1, R0, s0) < 1e-6


def test_device_outer_is_bit_reproducible(xmamd):
    """two runs of the device-driven loop give the same bits (dense, symmetric pair: the sweep direction is a function of the launch pair's
    index, and the pairs a solve needs are a function of its arithmetic)"""
    D = tl.gen_dense(356, seed=356)
    ctx = xmamd.Context(Q=D["Q"], tuning=dict(sym=1, sym_min_rows=256))
    a = ctx.solve(5, 1e-6, 0.0, trace=2000, flags=xmamd.FLAG_DEVICE_OUTER)
    b = ctx.solve(5, 1e-6, 0.0, trace=2000, flags=xmamd.FLAG_DEVICE_OUTER)
    ctx.close()
    assert a[2]["outer_on_device"] >= 1
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2]["trace"], b[2]["trace"])


def test_host_driven_configurations_stay_on_the_host(xmamd):
    """what the device-driven form does not cover runs as before: sliced-ELL storage, several ranks, the host-stepped debugging mode"""
    V = tl.gen_vg(600, deg=10, sigma=0.1, seed=3, dense=False)
    bsr = (V["rowptr"], V["colidx"], V["blocks"])
    ctx = xmamd.Context(bsr=bsr, tuning=dict(sell=1))
    R0, s0, i0 = ctx.solve(4, 1e-8, 10.0)
    ctx.close()
    assert i0["outer_on_device"] == 0 and i0["status"] == 1
    ctx = xmamd.Context(bsr=bsr, tuning=dict(sell=-1))
    R1, s1, i1 = ctx.solve(4, 1e-8, 10.0)
    R2, s2, i2 = ctx.solve(4, 1e-8, 10.0, flags=xmamd.FLAG_HOST_STEPPED)
    ctx.close()
    assert i1["outer_on_device"] >= 1 and i2["outer_on_device"] == 0
    assert np.array_equal(R1, R2) and np.array_equal(s1, s2)          # host-stepped == run-ahead == device-driven in block CSR
    assert tl.rotation_parity(R1, s1, R0, s0) < 1e-7
    ctx = xmamd.Context(bsr=bsr, n_gpus=2, gpu_map=1)
    R3, s3, i3 = ctx.solve(4, 1e-8, 10.0)
    ctx.close()
    assert i3["outer_on_device"] == 0 and i3["status"] == 1 and tl.rotation_parity(R3, s3, R1, s1) < 1e-7
    # dense products: host-driven by default (measured 1.5-2 % faster there), device-driven on request, host-driven when both flags are given
    D = tl.gen_dense(60, seed=60)
    ctx = xmamd.Context(Q=D["Q"])
    i4 = ctx.solve(4, 1e-8, 0.0)[2]
    i5 = ctx.solve(4, 1e-8, 0.0, flags=xmamd.FLAG_DEVICE_OUTER)[2]
    i6 = ctx.solve(4, 1e-8, 0.0, flags=xmamd.FLAG_DEVICE_OUTER | xmamd.FLAG_HOST_OUTER)[2]
    ctx.close()
    assert i4["outer_on_device"] == 0 and i5["outer_on_device"] >= 1 and i6["outer_on_device"] == 0
    assert i4["status"] == i5["status"] == 1 and i4["primal"] == pytest.approx(i5["primal"], rel=1e-9)


def test_device_outer_at_baseline_sizes_vs_recorded_oracle(xmamd):
    """Final-13682 in block CSR, both forms bit-identical and within 1e-6 of the CPU oracle's recorded rotations; the headline's Venice-1778
    is covered by test_venice1778_vs_recorded_oracle (default options: dense products, host-driven)"""
    fj = os.path.join(G, "synth", "rome13682_oracle.json")
    if not os.path.exists(fj):
        pytest.skip("recorded oracle run not present")
    c = json.load(open(fj))
    P = tl.gen_vg(c["n"], deg=c["deg"], sigma=c["sigma"], seed=c["n"], dense=False)
    ctx = xmamd.Context(bsr=(P["rowptr"], P["colidx"], P["blocks"]))
    dev, host = _both(ctx, xmamd, 5, c["tol"], c["lam"])
    ctx.close()
    _identical(dev, host)
    R, s, i = dev
    assert i["rank"] == 3 and i["status"] == 1 and i["primal"] == pytest.approx(c["f"], rel=1e-9)
    rot, _ = tl.recover_rotations(R, s)
    assert tl.rel_fro(rot, np.load(os.path.join(G, "synth", "rome13682_oracle_rot.npy"))) < 1e-6


def test_verbose_progress_lines_are_the_same_text(xmamd, capfd):
    """XM_FLAG_VERBOSE (the file surface's default, like the reference): the device-driven loop prints a stage's progress lines from its trace
    records when the stage has ended -- the same text the host-driven loop prints line by line (the wall-clock line aside)"""
    P = tl.gen_vg(40, deg=3, sigma=1.5, seed=40)
    ctx = xmamd.Context(bsr=(P["rowptr"], P["colidx"], P["blocks"]))
    out = []
    for fl in (xmamd.FLAG_VERBOSE, xmamd.FLAG_VERBOSE | xmamd.FLAG_HOST_OUTER):
        capfd.readouterr()
        R, s, info = ctx.solve(6, 1e-9, 3.0, flags=fl)
        txt = capfd.readouterr().out
        out.append(([l for l in txt.splitlines() if not l.startswith("Time taken")], info))
    ctx.close()
    (dev, idev), (host, ihost) = out
    assert idev["outer_on_device"] == 4 and ihost["outer_on_device"] == 0
    assert len(dev) > 100 and any("nagative curvature" in l for l in dev) and any(l.startswith("TR+ ") for l in dev)
    assert dev == host
