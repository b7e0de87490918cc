"""CPU tests: the oracle (C restatement of trustregion.h / checkeig.h / XM_main.cu) against
 (1) the golden fixtures in tests/golden, (2) numpy/LAPACK for its linear-algebra pieces,
 (3) implementation-independent certificate invariants, (4) planted known answers."""
import json
import os

import numpy as np
import pytest

import xm_testlib as tl

G = tl.GOLDEN


def _case(d):
    Q = tl.load_bin(os.path.join(d, "Q.bin"))
    exp = json.load(open(os.path.join(d, "expected.json")))
    return Q, exp


@pytest.mark.parametrize("name", ["simple1", "simple2", "synth/vg60_cert", "synth/dense49", "synth/vg40_stair"])
def test_oracle_reproduces_golden(oracle, name):
    d = os.path.join(G, name)
    Q, exp = _case(d)
    R, s, info = oracle.solve(Q, exp["max_rank"], exp["tol"], exp["lam"], 1000.0, trace=1000)
    assert info["rank"] == exp["rank"] and info["status"] == exp["status"]
    assert info["trace"][-1, 0] == pytest.approx(exp["f_star"], rel=1e-10)
    assert info["tcg_iters"] == exp["tcg_iters"] and info["outer_iters"] == exp["outer_iters"]
    assert tl.stiefel_defect(R) < 1e-12
    rot, _ = tl.recover_rotations(R, s)
    gold = np.load(os.path.join(d, "rot_anchor.npy"))
    # gold was produced by the REFERENCE's recover_XM -> also pins tl.recover_rotations
    assert tl.rel_fro(rot, gold) < 1e-8
    sR = tl.scale_rows(R, s)
    idx = tl.gram_sample_index(Q.shape[0])
    X = (sR[idx[:, 0]] * sR[idx[:, 1]]).sum(axis=1)
    assert tl.rel_fro(X, np.load(os.path.join(d, "sR_gram_sample.npy"))) < 1e-8


def test_simple1_known_answer(oracle):
    """1_test_solve.py:42 call; numbers of SURVEY.md §6 / BASELINE.md §2."""
    Q, exp = _case(os.path.join(G, "simple1"))
    assert exp["f_star"] == pytest.approx(2.5509915677233, rel=1e-9)
    assert exp["rank"] == 3 and exp["outer_iters"] == 13 and abs(exp["tcg_iters"] - 337) <= 5
    assert exp["s_min"] == pytest.approx(0.99424, abs=1e-5) and exp["s_max"] == pytest.approx(1.00532, abs=1e-5)
    assert exp["cert"]["min_eig"] > -1e-8 and abs(exp["cert"]["gap"]) / exp["f_star"] < 1e-6


def test_simple2_matches_ground_truth(oracle):
    d = os.path.join(G, "simple2")
    Q, exp = _case(d)
    assert exp["f_star"] == pytest.approx(0.04832243003949, rel=1e-8)
    assert exp["tcg_iters"] == 208 and exp["outer_iters"] == 14          # SURVEY.md §6 probe
    rot = np.load(os.path.join(d, "rot_anchor.npy"))
    gt = tl.load_bin(os.path.join(d, "gtR.bin"))
    fi = np.load(os.path.join(d, "frame_index.npy"))
    Gt = lambda i: gt[:, 3 * fi[i]:3 * fi[i] + 3]
    err = [np.linalg.norm(rot[:, 3 * i:3 * i + 3] - Gt(0) @ Gt(i).T) for i in range(fi.size)]
    assert max(err) < 6e-3 and np.median(err) < 3e-3


def test_planted_known_answer(oracle):
    """eps = 0 generator: f* = 0 and the planted rotations are recovered.
    (Driven through xmo_trustregion: with f* == 0 round-off can make the final loss slightly negative, which the
    staircase driver — like the reference, XM_main.cu:244 — mistakes for its line-search-failure sentinel.)"""
    n = 30
    P = tl.gen_dense(n, seed=7, noise=0.0)
    R, s, primal, _, st = oracle.trustregion(P["Q"], np.tile(np.eye(3), (n, 1)), np.ones(n), gradtol=1e-10)
    assert abs(primal) < 1e-10
    rot, sc = tl.recover_rotations(R, s)
    Rs = P["R_star"]
    # solver blocks converge to R*_i (up to a common right rotation): B_0 B_i^T = R*_0 R*_i^T
    ref = np.concatenate([Rs[0] @ Rs[i].T for i in range(n)], axis=1)
    assert tl.rel_fro(rot, ref) < 1e-6
    assert np.allclose(sc, 1.0, atol=1e-6)


def test_qw_matches_numpy(oracle):
    rng = np.random.default_rng(0)
    for n, o in [(1, 3), (7, 3), (50, 5), (171, 10)]:
        C = rng.standard_normal((3 * n, 3 * n)); W = rng.standard_normal((3 * n, o))
        assert np.allclose(oracle.qw(C, W, 2.0), 2.0 * C @ W, rtol=1e-12, atol=1e-12)


def test_mgs_rows_is_qr_with_positive_diagonal(oracle):
    rng = np.random.default_rng(1)
    for o in (3, 4, 7, 10):
        A = rng.standard_normal((3 * 11, o))
        Qm = oracle.mgs_rows(A)
        for i in range(11):
            q, r = np.linalg.qr(A[3 * i:3 * i + 3].T)        # o x 3
            q = q * np.sign(np.diag(r))[None, :]
            assert np.allclose(Qm[3 * i:3 * i + 3], q.T, atol=1e-12)


def test_syev_matches_lapack(oracle):
    rng = np.random.default_rng(2)
    for m in (1, 2, 3, 10, 63, 200):
        A = rng.standard_normal((m, m)); A = A + A.T
        w, V = oracle.syev_lower(A)
        w0 = np.linalg.eigvalsh(A)
        assert np.allclose(w, w0, atol=1e-10 * max(1, np.abs(w0).max()))
        assert np.allclose(V.T @ V, np.eye(m), atol=1e-10)
        assert np.allclose(A @ V, V * w[None, :], atol=1e-9 * max(1, np.abs(w0).max()))


def _cert_pieces(Q, R, s, lam):
    """independent numpy construction of the certificate"""
    n = s.size
    sR = tl.scale_rows(R, s)
    Z = Q.copy()
    for i in range(n):
        Z[3 * i, 3 * i] += 2 * lam * (sR[3 * i] @ sR[3 * i] - 1)
    return sR, Z


@pytest.mark.parametrize("name", ["simple2", "synth/vg60_cert"])
def test_certificate_invariants_and_multipliers(oracle, name):
    d = os.path.join(G, name)
    Q, exp = _case(d)
    lam = exp["lam"]
    R, s, info = oracle.solve(Q, exp["max_rank"], exp["tol"], lam, 1000.0)
    sR, Z = _cert_pieces(Q, R, s, lam)
    f = exp["f_star"]
    ok1, v1, c1 = oracle.checkeig(Q, sR, lam, f)                               # restated Eigen LSCG
    ok2, v2, c2 = oracle.checkeig(Q, sR, lam, f, flags=oracle.CLOSED_FORM_Y)   # per-camera closed form
    assert ok1 and ok2
    assert c1["min_eig"] == pytest.approx(c2["min_eig"], abs=1e-9)
    assert c1["dual"] == pytest.approx(c2["dual"], rel=1e-10)
    assert c1["min_eig"] > -1e-7 and abs(c1["gap"]) < 1e-6 * max(1.0, f)
    # independent check: with Lambda_i = (Z sR)_i sR_i^+ (least squares per camera), S = Z - blockdiag(Lambda)
    # must annihilate sR and be PSD at a certified optimum
    n = s.size
    S = Z.copy()
    Right = Z @ sR
    for i in range(n):
        Bi = sR[3 * i:3 * i + 3]
        S[3 * i:3 * i + 3, 3 * i:3 * i + 3] -= Right[3 * i:3 * i + 3] @ np.linalg.pinv(Bi)
    assert np.linalg.norm(S @ sR) < 1e-6 * max(1.0, np.linalg.norm(Z))
    assert np.linalg.eigvalsh(0.5 * (S + S.T))[0] > -1e-6


def test_staircase_escalates_and_certifies(oracle):
    d = os.path.join(G, "synth/vg40_stair")
    Q, exp = _case(d)
    assert exp["rank"] > 3 and exp["status"] == 1                # rank-3 critical point is a saddle
    R3, s3, i3 = oracle.solve(Q, 3, exp["tol"], exp["lam"], 1000.0, trace=1000)
    assert i3["status"] == 2 and i3["cert"]["min_eig"] < -1e-3   # rejected at rank 3
    assert i3["trace"][-1, 0] > exp["f_star"] + 1e-3             # escalation strictly improves the optimum


@pytest.mark.parametrize("name", ["synth/dense49", "synth/vg40_stair", "synth/vg60_cert"])
def test_lapack_eigen_step_gives_the_same_certificate_and_staircase(oracle, name):
    """bench.py's CPU wall-clock-to-KKT leg runs the oracle's certificate with LAPACK dsyevd (xm_oracle.use_lapack_eig: the closest CPU analogue
    of cusolverDnXsyevd, Dense/eig.h:35-73) instead of the restated tred2 / tql2, which needs an hour at 5334 rows.  Same ranks, same
    acceptance, min_eig / dual / gap to 1e-10 (1e-9 behind an escalation), same solution up to the gauge -- also through a rank escalation, whose direction is the
    eigenvector the eigen step returns"""
    Q, exp = _case(os.path.join(G, name))
    lam = exp["lam"]
    a = oracle.solve(Q, exp["max_rank"], 1e-9, lam, 1000.0)
    oracle.use_lapack_eig(True)
    try:
        b = oracle.solve(Q, exp["max_rank"], 1e-9, lam, 1000.0)
    finally:
        oracle.use_lapack_eig(False)
    (Ra, sa, ia), (Rb, sb, ib) = a, b
    assert ia["rank"] == ib["rank"] and ia["status"] == ib["status"] and ia["cert"]["accepted"] == ib["cert"]["accepted"]
    # no escalation: the same point goes into both eigen steps (1e-10); after one, the two runs continue from eigenvectors that agree to
    # round-off and stop |grad| < 1e-9 apart
    tol = 1e-10 if ia["rank"] == 3 else 1e-9
    for k in ("min_eig", "dual", "gap"):
        assert abs(ia["cert"][k] - ib["cert"][k]) <= tol * max(1.0, abs(ia["cert"]["dual"])), k
    assert tl.rel_fro(tl.gram(Rb, sb), tl.gram(Ra, sa)) < 1e-8
    assert tl.rotation_parity(Rb, sb, Ra, sa) < 1e-8


def test_gradtol_quirk_and_rank3_mode(oracle):
    Q, exp = _case(os.path.join(G, "synth/dense49"))
    n = 49
    R0 = np.tile(np.eye(3), (n, 1))
    R, s, primal, gt, st = oracle.trustregion(Q, R0, np.ones(n), lam=0.0, gradtol=1e-3)
    assert st["stop_reason"] == 10 and gt == pytest.approx(1e-4)   # tr.h:534 gradtol /= 10
    R1, s1, i1 = oracle.solve(Q, 7, 1e-3, 0.0, 1000.0, mode=1)
    assert i1["rank"] == 3 and np.allclose(R1, R) and np.allclose(s1, s)


def test_bin_roundtrip_and_solve_path(oracle, tmp_path):
    Q, exp = _case(os.path.join(G, "synth/dense49"))
    p = tmp_path / "ds"
    p.mkdir()
    tl.save_bin(p / "Q.bin", Q)
    assert np.array_equal(tl.load_bin(p / "Q.bin"), Q)
    raw = open(p / "Q.bin", "rb").read()
    assert raw[:8] == np.array([147, 147], dtype="<i4").tobytes() and len(raw) == 8 + 8 * 147 * 147
    st = oracle.solve_path(str(p), 5, 1e-12, 0.0, 1000.0)
    assert st == 1
    R = tl.load_bin(p / "R.bin"); s = tl.load_bin(p / "s.bin")
    assert R.shape == (147, exp["rank"]) and s.shape == (49, 1) and s[0, 0] == 1.0
    rot, _ = tl.recover_rotations(R, s)
    assert tl.rel_fro(rot, np.load(os.path.join(G, "synth/dense49/rot_anchor.npy"))) < 1e-8
    assert oracle.solve_path(str(tmp_path / "missing"), 5, 1e-6, 0.0, 10.0) < 0


def test_bsr_product_of_the_oracle_equals_dense(oracle):
    """test-only extension (oracle/xm_oracle.c:xmo_set_bsr): the trust region multiplied from 3x3-block CSR must walk the same
    path as from the dense matrix it describes (used to anchor the GPU's BSR3 storage at sizes the dense oracle cannot hold)"""
    P = tl.gen_vg(120, deg=8, sigma=0.2, seed=12)
    n = 120
    R0 = np.tile(np.eye(3), (n, 1)); s0 = np.ones(n)
    Rd, sd, pd, _, std = oracle.trustregion(P["Q"], R0, s0, lam=5.0, gradtol=1e-9)
    Rb, sb, pb, _, stb = oracle.trustregion_bsr(P["rowptr"], P["colidx"], P["blocks"], R0, s0, lam=5.0, gradtol=1e-9)
    assert pb == pytest.approx(pd, rel=1e-12)
    assert tl.rotation_parity(Rb, sb, Rd, sd) < 1e-8
    assert abs(stb["tcg_iters"] - std["tcg_iters"]) <= 0.05 * std["tcg_iters"] + 5
    W = np.random.default_rng(0).standard_normal((3 * n, 4))
    assert np.allclose(oracle.qw(P["Q"], W, 1.0), tl.bsr_to_dense(n, P["rowptr"], P["colidx"], P["blocks"]) @ W, atol=1e-12)


@pytest.mark.parametrize("name", ["simple2", "synth/dense49", "synth/vg60_cert", "synth/vg40_stair"])
def test_numpy_certificate_agrees_with_the_oracle(oracle, name):
    """tl.certificate_numpy (lstsq multipliers + LAPACK eigvalsh, no code shared with the oracle) on the oracle's solution of the
    golden inputs: same dual value and lambda_min as the oracle's restatement of checkeig.h, zero gap, stationarity -- this pins the
    oracle's certificate to an independent implementation and is what the GPU tests apply to the GPU's own output"""
    import json, os
    d = os.path.join(tl.GOLDEN, name)
    Q = tl.load_bin(os.path.join(d, "Q.bin")); exp = json.load(open(os.path.join(d, "expected.json")))
    R, s, io = oracle.solve(Q, exp["max_rank"], exp["tol"], exp["lam"], 1000.0)
    cn = tl.certificate_numpy(Q, R, s, exp["lam"])
    assert cn["primal"] == pytest.approx(exp["f_star"], rel=1e-10)
    assert cn["dual"] == pytest.approx(exp["cert"]["dual"], rel=1e-8) and cn["min_eig"] == pytest.approx(exp["cert"]["min_eig"], abs=1e-8)
    assert cn["min_eig"] > -1e-7 and abs(cn["gap"]) < 1e-6 * max(1.0, cn["primal"]) and cn["stationarity"] < 1e-5


def test_simple2_observations_reproduce_the_references_Q():
    """tests/golden/simple2/obs.npz (the observation list the reference's 2_test_creatematrix.py hands to create_matrix) against the
    Q.bin create_matrix wrote: the dense Schur complement restated in numpy (tl.schur_dense) and the five-step matrix-free product
    (tl.schur_qw_numpy, what xm_schur.hip runs on the device) both reproduce it"""
    import os
    Z = np.load(os.path.join(tl.GOLDEN, "simple2", "obs.npz"))
    Q = tl.load_bin(os.path.join(tl.GOLDEN, "simple2", "Q.bin"))
    assert tl.rel_fro(tl.schur_dense(Z["cam"], Z["lm"], Z["p"], Z["w"]), Q) < 1e-11
    W = np.random.default_rng(0).standard_normal((Q.shape[0], 4))
    assert tl.rel_fro(tl.schur_qw_numpy(Z["cam"], Z["lm"], Z["p"], Z["w"], W), Q @ W) < 1e-11


def test_schur_tp_restatement_matches_reference_recover():
    """tests/golden/simple2/tp.npz holds what the reference's own pipeline got from recover_XM for SIMPLE2 (anchored rotations, scales,
    t_est, p_est = Abar @ sR_real^T; tests/golden/make_simple2_tp.py).  The numpy restatement of the device chain (tl.schur_tp_numpy: the
    eliminated variables recomputed from the observation list, no Abar) reproduces t_est / p_est."""
    Z = np.load(os.path.join(tl.GOLDEN, "simple2", "obs.npz")); T = np.load(os.path.join(tl.GOLDEN, "simple2", "tp.npz"))
    t, p = tl.schur_tp_numpy(Z["cam"], Z["lm"], Z["p"], Z["w"], T["R_real"], T["s_real"])
    assert t.shape == T["t_est"].shape and p.shape == T["p_est"].shape
    assert np.abs(t - T["t_est"]).max() < 1e-9 * np.abs(T["t_est"]).max()
    assert np.abs(p - T["p_est"]).max() < 1e-9 * np.abs(T["p_est"]).max()
    assert np.all(t[:, 0] == 0.0)


def test_xm2_round_by_hand_with_the_oracle(oracle):
    """One round of the reference's XM^2 sequence (3_test_colmap_glomap.py:299-351) done with numpy and the CPU oracle on a 400-camera
    hub view graph with 8 % gross outlier edges: solve, recovered-solution residuals, 90th-percentile filter, solve_rank3 at lam 0,
    the lam decision from the rank-3 scales, final solve.  These are the numbers the GPU test
    test_xm2_round_on_two_virtual_gpus_equals_the_single_gpu_round_and_the_oracle compares xm_ctx_xm2_round with (there recomputed on
    the GPU box); recorded here so that a drift of the oracle itself shows up in the CPU suite."""
    n = 400
    H = tl.gen_vg_hubs(n, 8, 2, 0.3, 0.05, seed=12)
    ei, ej, w, M = H["ei"], H["ej"], H["w"] * 0.2, H["M"].copy()
    rng = np.random.default_rng(3)
    for e in rng.choice(ei.size, size=int(ei.size * 0.08), replace=False):
        q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        M[e] = q * np.sign(np.linalg.det(q))
    Q = tl.bsr_to_dense(n, *tl.vg_from_edges(n, ei, ej, w, M))
    R, s, info = oracle.solve(Q, 5, 1e-8, 20.0, 1000.0)
    assert info["status"] == 1 and info["rank"] == 3 and info["cert"]["dual"] == pytest.approx(161.96882949118984, rel=1e-9)
    rot, scale = tl.recover_rotations(R, s)
    Y = np.stack([scale[i] * rot[:, 3 * i:3 * i + 3].T for i in range(n)])
    res = np.array([np.sum((Y[ei[e]] - M[e] @ Y[ej[e]]) ** 2) for e in range(ei.size)])
    err = w * res
    thr = float(np.percentile(err, 90.0))
    w2 = np.where(err > thr, 0.0, w)
    assert thr == pytest.approx(0.04263070395615991, rel=1e-7) and int((w2 == 0).sum()) == 177
    Q2 = tl.bsr_to_dense(n, *tl.vg_from_edges(n, ei, ej, w2, M))
    R3, s3, _ = oracle.solve(Q2, 3, 1e-8, 0.0, 1000.0, mode=1)
    avg, sd, small = float(s3[1:].mean()), float(s3[1:].std()), int((s3 < 0.1).sum())
    assert avg == pytest.approx(0.3748172034893503, rel=1e-6) and sd == pytest.approx(0.06370245827607149, rel=1e-5)
    assert abs(avg - 1) > 2 * sd and small == 0                      # -> regularised, lam = kept edges / cameras
    lam = (w2 != 0).sum() / n
    assert lam == pytest.approx(3.975)
    Ro, so, io = oracle.solve(Q2, 5, 1e-8, lam, 1000.0)
    assert io["status"] == 1 and io["rank"] == 3 and io["cert"]["dual"] == pytest.approx(3.562629993886309, rel=1e-8)
    assert io["cert"]["min_eig"] > -1e-9
