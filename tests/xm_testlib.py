"""Shared helpers for the tests and bench.py: .bin I/O, seeded problem generators,
3x3-block CSR conversion, and the gauge-invariant parity metrics.

Pure numpy; no GPU, no oracle, no reference imports.
"""
import os

import numpy as np

# --------------------------------------------------------------------------------------------------
# .bin matrix format (reference utils/io.py:17-54, XM_main.cu:18-33): int32 rows, int32 cols, f64 col-major
# --------------------------------------------------------------------------------------------------


def load_bin(fn):
    with open(fn, "rb") as f:
        r, c = (int(x) for x in np.fromfile(f, dtype="<i4", count=2))
        d = np.fromfile(f, dtype="<f8", count=r * c)
    if d.size != r * c:
        raise IOError(f"{fn}: short file")
    return d.reshape((r, c), order="F")


def save_bin(fn, M):
    M = np.atleast_2d(np.asarray(M, dtype=np.float64))
    with open(fn, "wb") as f:
        np.array(M.shape, dtype="<i4").tofile(f)
        np.asfortranarray(M).T.tofile(f)  # == column-major bytes


# --------------------------------------------------------------------------------------------------
# generators
# --------------------------------------------------------------------------------------------------


def haar_so3(rng, n):
    """n Haar-distributed rotations (QR of N(0,1) 3x3, sign-fixed, det -> +1).  Returns (n,3,3)."""
    A = rng.standard_normal((n, 3, 3))
    Qm, Rm = np.linalg.qr(A)
    d = np.sign(np.diagonal(Rm, axis1=1, axis2=2))
    d[d == 0] = 1.0
    Qm = Qm * d[:, None, :]
    det = np.linalg.det(Qm)
    Qm[:, :, 2] *= det[:, None]
    return Qm


def so3_exp(w):
    """Rodrigues; w (m,3) -> (m,3,3)"""
    th = np.linalg.norm(w, axis=1)
    K = np.zeros((w.shape[0], 3, 3))
    K[:, 0, 1], K[:, 0, 2] = -w[:, 2], w[:, 1]
    K[:, 1, 0], K[:, 1, 2] = w[:, 2], -w[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -w[:, 1], w[:, 0]
    th2 = np.where(th < 1e-12, 1.0, th)
    a = np.where(th < 1e-12, 1.0, np.sin(th2) / th2)
    b = np.where(th < 1e-12, 0.5, (1 - np.cos(th2)) / th2**2)
    return np.eye(3)[None] + a[:, None, None] * K + b[:, None, None] * (K @ K)


def gen_dense(n, seed=None, noise=0.3):
    """G_dense(n, seed): fully dense 'Schur-complement-like' PSD Q with planted optimum U* (SURVEY §8d C2-C4):
    Q = P M P + N, M = B B^T + diag(d) (B Gaussian / sqrt(3n), d ~ U[0.5, 1.5]), P = projector onto the complement of
    U* (the stacked planted rotations), N = PSD noise (diagonal + rank-8) scaled so that the planted point costs
    f(U*) = tr(U*^T Q U*) = `noise` whatever n.  Keeping f(U*) far below tr(Q_00) ~ 6 (the cost of letting every free scale
    collapse to 0) makes the instance well posed like a real XM Q: scales stay ~1 and the relaxation is tight at rank 3.
    The noise is PSD on purpose: a real XM Q is a sum of squares and the staircase driver (like the reference,
    XM_main.cu:244) treats a negative optimum value as its line-search-failure sentinel.
    noise = 0 gives the known-answer variant (f* = 0, R* = U*)."""
    seed = n if seed is None else seed
    rng = np.random.default_rng(seed)
    Rs = haar_so3(rng, n)
    U = Rs.reshape(3 * n, 3)
    m = 3 * n
    B = rng.standard_normal((m, m)) / np.sqrt(m)
    M = B @ B.T
    M[np.diag_indices(m)] += rng.uniform(0.5, 1.5, m)
    Uo, _ = np.linalg.qr(U)                       # P = I - Uo Uo^T
    MU = M @ Uo
    Qm = M - Uo @ MU.T - MU @ Uo.T + Uo @ (Uo.T @ MU) @ Uo.T
    if noise:
        F = rng.standard_normal((m, 8)) / np.sqrt(8.0)
        u = rng.uniform(0.0, 1.0, m)
        scale = noise / (np.sum(u) + np.linalg.norm(F.T @ U) ** 2)    # rows of U* have unit norm
        Qm += scale * (F @ F.T)
        Qm[np.diag_indices(m)] += scale * u
    Qm = (Qm + Qm.T) * 0.5
    return dict(Q=Qm, R_star=Rs, n=n)


def gen_vg_edges(n, deg, seed):
    """path 0-1-...-(n-1) union Erdos-Renyi G(n, p=(deg-2)/n); returns unique (i<j) edge array."""
    rng = np.random.default_rng(seed)
    path = np.stack([np.arange(n - 1), np.arange(1, n)], axis=1)
    p = max(deg - 2, 0) / n
    m_expect = p * n * (n - 1) / 2
    m = rng.poisson(m_expect) if m_expect > 0 else 0
    e = rng.integers(0, n, size=(m, 2))
    e = e[e[:, 0] != e[:, 1]]
    e = np.sort(e, axis=1)
    e = np.concatenate([path, e], axis=0)
    key = e[:, 0].astype(np.int64) * n + e[:, 1]
    _, idx = np.unique(key, return_index=True)
    return e[np.sort(idx)], rng


def gen_vg(n, deg=20, sigma=0.05, seed=None, dense=True):
    """G_vg(n,deg,sigma,seed): view-graph connection-Laplacian Q.  Returns dict with BSR3 arrays
    (rowptr int64, colidx int32, blocks (nb,3,3) row-major 3x3 = Q_ij) and optionally the dense matrix."""
    seed = n if seed is None else seed
    edges, rng = gen_vg_edges(n, deg, seed)
    Rs = haar_so3(rng, n)
    ne = edges.shape[0]
    xi = rng.standard_normal((ne, 3)) * sigma
    i, j = edges[:, 0], edges[:, 1]
    Mij = Rs[i] @ so3_exp(xi) @ np.transpose(Rs[j], (0, 2, 1))
    w = np.ones(ne)
    degw = np.zeros(n)
    np.add.at(degw, i, w)
    np.add.at(degw, j, w)
    rows = np.concatenate([np.arange(n), i, j])
    cols = np.concatenate([np.arange(n), j, i])
    blocks = np.concatenate([degw[:, None, None] * np.eye(3)[None], -w[:, None, None] * Mij,
                             -w[:, None, None] * np.transpose(Mij, (0, 2, 1))], axis=0)
    order = np.lexsort((cols, rows))
    rows, cols, blocks = rows[order], cols[order], blocks[order]
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rowptr, rows + 1, 1)
    rowptr = np.cumsum(rowptr)
    out = dict(n=n, rowptr=rowptr, colidx=cols.astype(np.int32), blocks=np.ascontiguousarray(blocks), R_star=Rs,
               edges=edges, M=np.ascontiguousarray(Mij), w=w)
    if dense:
        out["Q"] = bsr_to_dense(n, rowptr, out["colidx"], blocks)
    return out


def vg_from_edges(n, ei, ej, w, M):
    """3x3-block CSR of Q = sum_e w_e G_e (Q_ii += w I, Q_jj += w I, Q_ij = -w M, Q_ji = Q_ij^T): the numpy statement of what
    XM_STORAGE_VIEWGRAPH builds from an edge list (diagonal sums in edge order)"""
    ei = np.asarray(ei); ej = np.asarray(ej); w = np.asarray(w, dtype=np.float64); M = np.asarray(M, dtype=np.float64).reshape(-1, 3, 3)
    degw = np.zeros(n)
    for e in range(ei.size):
        degw[ei[e]] += w[e]; degw[ej[e]] += w[e]
    rows = np.concatenate([np.arange(n), ei, ej]); cols = np.concatenate([np.arange(n), ej, ei])
    blocks = np.concatenate([degw[:, None, None] * np.eye(3)[None], -w[:, None, None] * M, -w[:, None, None] * np.transpose(M, (0, 2, 1))], axis=0)
    order = np.lexsort((cols, rows))
    rows, cols, blocks = rows[order], cols[order], blocks[order]
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rowptr, rows + 1, 1)
    return np.cumsum(rowptr), cols.astype(np.int32), np.ascontiguousarray(blocks)


def gen_vg_hubs(n, deg, hubs, frac, sigma, seed):
    """view graph with HUB cameras: the Erdos-Renyi graph of gen_vg plus `hubs` cameras that see a fraction `frac` of all others
    (rows of ~frac*n blocks among rows of ~deg).  Returns the edge list with consistent noisy rotations, unit weights."""
    edges, rng = gen_vg_edges(n, deg, seed)
    extra = []
    for h in range(hubs):
        hub = int(rng.integers(0, n))
        others = rng.choice(n, size=int(frac * n), replace=False)
        extra += [(min(hub, int(c)), max(hub, int(c))) for c in others if int(c) != hub]
    e = np.concatenate([edges, np.array(extra, dtype=edges.dtype).reshape(-1, 2)], axis=0)
    key = e[:, 0].astype(np.int64) * n + e[:, 1]
    _, idx = np.unique(key, return_index=True)
    e = e[np.sort(idx)]
    Rs = haar_so3(rng, n)
    xi = rng.standard_normal((e.shape[0], 3)) * sigma
    M = Rs[e[:, 0]] @ so3_exp(xi) @ np.transpose(Rs[e[:, 1]], (0, 2, 1))
    return dict(n=n, ei=e[:, 0].astype(np.int32), ej=e[:, 1].astype(np.int32), w=np.ones(e.shape[0]), M=np.ascontiguousarray(M), R_star=Rs)


def xm2_error_numpy(cam, lm, p, w, R_real, s_real, t_est, p_est):
    """weighted squared residual per observation of a recovered solution: w * | s_i R_i p + t_i - P_l |^2 (what the reference's XM^2
    loop thresholds, 3_test_colmap_glomap.py:305-316).  R_real: 3 x 3N (block i = columns 3i..3i+2), t_est 3 x N, p_est 3 x M."""
    N = np.asarray(s_real).size
    Rm = np.asarray(R_real).reshape(3, N, 3).transpose(1, 0, 2)
    x = np.asarray(s_real).reshape(-1)[cam, None] * np.einsum("nij,nj->ni", Rm[cam], np.asarray(p)) + np.asarray(t_est)[:, cam].T
    d = np.asarray(p_est)[:, lm].T - x
    return np.asarray(w).reshape(-1) * np.sum(d * d, axis=1)


def bsr_to_dense(n, rowptr, colidx, blocks):
    Qm = np.zeros((3 * n, 3 * n))
    for r in range(n):
        for q in range(rowptr[r], rowptr[r + 1]):
            c = colidx[q]
            Qm[3 * r:3 * r + 3, 3 * c:3 * c + 3] = blocks[q]
    return Qm


def dense_to_bsr(Qm, tol=0.0):
    n = Qm.shape[0] // 3
    B = Qm.reshape(n, 3, n, 3).transpose(0, 2, 1, 3)
    nz = np.abs(B).max(axis=(2, 3)) > tol
    rows, cols = np.nonzero(nz)
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rowptr, rows + 1, 1)
    return np.cumsum(rowptr), cols.astype(np.int32), np.ascontiguousarray(B[rows, cols])


def make_problem(kind, **kw):
    if kind == "dense":
        return gen_dense(**kw)
    if kind == "vg":
        return gen_vg(**kw)
    raise ValueError(kind)


# --------------------------------------------------------------------------------------------------
# parity metrics (gauge invariant)
# --------------------------------------------------------------------------------------------------


def scale_rows(R, s):
    s = np.asarray(s).reshape(-1)
    return np.asarray(R) * np.repeat(s, 3)[:, None]


def gram_sample_index(m, k=4096, seed=12345):
    rng = np.random.default_rng(seed)
    return rng.integers(0, m, size=(k, 2))


def gram(R, s):
    sR = scale_rows(R, s)
    return sR @ sR.T


def rel_fro(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def recover_rotations(R, s):
    """Anchored O(3)/SO(3) rotations (3 x 3n) + scales — the quantity the pipeline consumes
    (semantics of reference utils/recoversolution.py:22-86, re-implemented; r > 3 handled through the
    r x r Gram matrix instead of the 3n x 3n one)."""
    R = np.asarray(R, dtype=np.float64)
    s = np.asarray(s, dtype=np.float64).reshape(-1)
    n = s.shape[0]
    sR = scale_rows(R, s)
    if R.shape[1] > 3:
        w, V = np.linalg.eigh(sR.T @ sR)
        sR = sR @ V[:, ::-1][:, :3]
    B = sR.reshape(n, 3, 3)                                   # B_i = camera block (3 x 3)
    sc = np.linalg.norm(B, axis=(1, 2)) / np.sqrt(3.0)
    Rt = np.transpose(B, (0, 2, 1)) / sc[:, None, None]       # reference works with B_i^T
    Rt = Rt[0].T[None] @ Rt                                   # anchor to camera 0
    U, _, Vt = np.linalg.svd(Rt)
    P = U @ Vt
    if (np.linalg.det(P) < 0).sum() > n / 2:
        U, _, Vt = np.linalg.svd(-Rt)
        P = U @ Vt
    return np.concatenate(list(P), axis=1), sc


def rotation_parity(Ra, sa, Rb, sb):
    """relative Frobenius distance between the anchored rotations of two solutions"""
    A, _ = recover_rotations(Ra, sa)
    Bm, _ = recover_rotations(Rb, sb)
    return rel_fro(A, Bm)


def stiefel_defect(R):
    n = R.shape[0] // 3
    B = np.asarray(R).reshape(n, 3, -1)
    G = B @ np.transpose(B, (0, 2, 1))
    return float(np.abs(G - np.eye(3)[None]).max())


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gen_skewed(n, deg, seed=None, hubs=None):
    """Block-sparse symmetric Q with a skewed degree distribution: a view graph (path + Erdos-Renyi, mean degree `deg`) plus a few
    hub cameras that see a large share of all cameras (what a landmark-rich frame does to a real view graph).  Random blocks
    (kernel tests only; Q_ji = Q_ij^T, identity-like diagonal); returns the BSR3 arrays."""
    seed = n if seed is None else seed
    edges, rng = gen_vg_edges(n, deg, seed)
    hubs = max(1, n // 2000) if hubs is None else hubs
    extra = []
    for h in rng.choice(n, size=hubs, replace=False):
        others = rng.choice(n, size=max(1, n // 4), replace=False)
        others = others[others != h]
        extra.append(np.stack([np.minimum(h, others), np.maximum(h, others)], axis=1))
    e = np.concatenate([edges] + extra, axis=0)
    key = e[:, 0].astype(np.int64) * n + e[:, 1]
    _, idx = np.unique(key, return_index=True)
    e = e[np.sort(idx)]
    i, j = e[:, 0], e[:, 1]
    M = rng.standard_normal((e.shape[0], 3, 3))
    rows = np.concatenate([np.arange(n), i, j]); cols = np.concatenate([np.arange(n), j, i])
    blocks = np.concatenate([np.eye(3)[None] * (1.0 + np.arange(n))[:, None, None] / n, M, np.transpose(M, (0, 2, 1))], axis=0)
    order = np.lexsort((cols, rows))
    rows, cols, blocks = rows[order], cols[order], blocks[order]
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rowptr, rows + 1, 1)
    return dict(n=n, rowptr=np.cumsum(rowptr), colidx=cols.astype(np.int32), blocks=np.ascontiguousarray(blocks))


# --------------------------------------------------------------------------------------------------
# implementation-independent dual certificate (numpy / LAPACK only; no oracle, no product code)
# --------------------------------------------------------------------------------------------------


def _cert_generators(anchor):
    """constraint matrices of one camera (reference checkeig.h:71-161): the anchor's block is pinned to I (6 symmetric unit
    matrices), a free-scale camera's block to a multiple of I (2 traceless diagonal + 3 off-diagonal generators)"""
    E = lambda a, b: np.outer(np.eye(3)[a], np.eye(3)[b])
    if anchor:
        return [E(0, 0), 0.5 * (E(0, 1) + E(1, 0)), 0.5 * (E(0, 2) + E(2, 0)), E(1, 1), 0.5 * (E(1, 2) + E(2, 1)), E(2, 2)]
    return [0.5 * (E(0, 0) - E(1, 1)), 0.5 * (E(1, 1) - E(2, 2)), 0.5 * (E(0, 1) + E(1, 0)), 0.5 * (E(0, 2) + E(2, 0)),
            0.5 * (E(1, 2) + E(2, 1))]


def certificate_numpy(Q, R, s, lam):
    """Dual certificate of a primal point (R, s) of the XM SDP, computed from scratch with numpy/LAPACK:
    multipliers by a dense least-squares solve per camera (np.linalg.lstsq), S = Z - sum_k y_k A_k assembled densely,
    its spectrum by np.linalg.eigvalsh.  Follows the construction of the reference's checkeig.h:56-368 (Z = Q + the lambda term on
    the first row of every camera block, checkeig.h:30-40; dual value checkeig.h:321-333; gap checkeig.h:334-336) but shares no
    code with the oracle or the GPU library, so it pins optimality of a GPU result independently of both.
    Returns dict(primal, dual, gap, min_eig, stationarity = |S sR|_F / |Q sR|_F)."""
    Q = np.asarray(Q, dtype=np.float64)
    s = np.asarray(s, dtype=np.float64).reshape(-1)
    n = s.size
    sR = scale_rows(R, s)
    xii = np.sum(sR[0::3] ** 2, axis=1)                       # |row 3i of sR|^2
    Z = Q.copy()
    Z[np.arange(0, 3 * n, 3), np.arange(0, 3 * n, 3)] += 2.0 * lam * (xii - 1.0)
    Right = Z @ sR
    S = Z.copy()
    dual = 0.0
    for i in range(n):
        B = sR[3 * i:3 * i + 3]
        gens = _cert_generators(i == 0)
        A = np.stack([(g @ B).ravel() for g in gens], axis=1)             # columns vec(A_k sR) restricted to camera i
        y, *_ = np.linalg.lstsq(A, Right[3 * i:3 * i + 3].ravel(), rcond=None)
        S[3 * i:3 * i + 3, 3 * i:3 * i + 3] -= sum(yk * g for yk, g in zip(y, gens))
        if i == 0:
            dual += y[0] + y[3] + y[5]
    dual += lam * np.sum(1.0 - xii ** 2)
    primal = float(np.sum(sR * (Q @ sR)) + lam * np.sum((s[1:] ** 2 - 1.0) ** 2))
    w = np.linalg.eigvalsh(0.5 * (S + S.T))
    gap = primal - dual - 3.0 * n * min(0.0, w[0])
    return dict(primal=primal, dual=float(dual), gap=float(gap), min_eig=float(w[0]),
                stationarity=float(np.linalg.norm(S @ sR) / max(np.linalg.norm(Q @ sR), 1e-300)), eigs=w)


def vg_measurements(n, deg, sigma, seed, outlier_frac=0.0):
    """view-graph measurements for the XM^2 tests: edges (i < j), M_e = R*_i Exp(sigma xi) R*_j^T, a fraction of them replaced by
    random rotations (planted outliers).  Returns edges, M (ne,3,3), is_outlier, R_star."""
    edges, rng = gen_vg_edges(n, deg, seed)
    Rs = haar_so3(rng, n)
    ne = edges.shape[0]
    M = Rs[edges[:, 0]] @ so3_exp(rng.standard_normal((ne, 3)) * sigma) @ np.transpose(Rs[edges[:, 1]], (0, 2, 1))
    bad = np.zeros(ne, dtype=bool)
    if outlier_frac > 0:
        cand = np.arange(n - 1, ne)                       # keep the spanning path clean so the graph stays connected after filtering
        bad[rng.choice(cand, size=int(outlier_frac * ne), replace=False)] = True
        M[bad] = haar_so3(rng, int(bad.sum()))
    return edges, M, bad, Rs


def vg_assemble(n, edges, M, w):
    """3x3-block CSR of the connection Laplacian Q = sum_e w_e G_e (Q_ii += w I, Q_jj += w I, Q_ij = -w M_e, Q_ji = Q_ij^T); the
    pattern always holds every edge and every diagonal block (also for w_e = 0), so a re-weighting keeps the structure"""
    i, j = edges[:, 0], edges[:, 1]
    degw = np.zeros(n)
    np.add.at(degw, i, w)
    np.add.at(degw, j, w)
    rows = np.concatenate([np.arange(n), i, j]); cols = np.concatenate([np.arange(n), j, i])
    blocks = np.concatenate([degw[:, None, None] * np.eye(3)[None], -w[:, None, None] * M, -w[:, None, None] * np.transpose(M, (0, 2, 1))], axis=0)
    order = np.lexsort((cols, rows))
    rows, cols, blocks = rows[order], cols[order], blocks[order]
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rowptr, rows + 1, 1)
    return np.cumsum(rowptr), cols.astype(np.int32), np.ascontiguousarray(blocks)


# --------------------------------------------------------------------------------------------------
# Schur-complement Q from observations (what the reference's utils/creatematrix.py:create_matrix builds, lines 51-339): numpy
# restatements used by the matrix-free tests.  Q = Q1 - Vtp_bar Qtp_bar^{-1} Vtp_bar^T with the anchor
# camera's translation removed (bar).
# --------------------------------------------------------------------------------------------------


def schur_parts(cam, lm, p, w):
    """Q1 (N,3,3), c (N,3) = sum_l w p, Q2 (N), Q3 (M), V3 as (cam, lm, w) — creatematrix.py:71-133"""
    cam = np.asarray(cam); lm = np.asarray(lm); p = np.asarray(p, dtype=np.float64); w = np.asarray(w, dtype=np.float64).reshape(-1)
    N, M = int(cam.max()) + 1, int(lm.max()) + 1
    Q1 = np.zeros((N, 3, 3)); c = np.zeros((N, 3)); Q2 = np.zeros(N); Q3 = np.zeros(M)
    np.add.at(Q1, cam, w[:, None, None] * p[:, :, None] * p[:, None, :])
    np.add.at(c, cam, w[:, None] * p)
    np.add.at(Q2, cam, w); np.add.at(Q3, lm, w)
    return N, M, Q1, c, Q2, Q3


def schur_dense(cam, lm, p, w):
    """dense 3N x 3N Q by explicit elimination of translations and landmarks (small problems only)"""
    cam = np.asarray(cam); lm = np.asarray(lm); p = np.asarray(p, dtype=np.float64); w = np.asarray(w, dtype=np.float64).reshape(-1)
    N, M, Q1, c, Q2, Q3 = schur_parts(cam, lm, p, w)
    Vtp = np.zeros((3 * N, N + M))
    for i in range(N):
        Vtp[3 * i:3 * i + 3, i] = c[i]
    np.add.at(Vtp, (3 * cam[:, None] + np.arange(3)[None, :], N + lm[:, None]), -(w[:, None] * p))
    Qtp = np.zeros((N + M, N + M))
    Qtp[np.arange(N), np.arange(N)] = Q2
    Qtp[N + np.arange(M), N + np.arange(M)] = Q3
    np.add.at(Qtp, (cam, N + lm), -w); np.add.at(Qtp, (N + lm, cam), -w)
    X = np.linalg.solve(Qtp[1:, 1:], Vtp[:, 1:].T)
    Q = -Vtp[:, 1:] @ X
    for i in range(N):
        Q[3 * i:3 * i + 3, 3 * i:3 * i + 3] += Q1[i]
    return 0.5 * (Q + Q.T)


def schur_qw_numpy(cam, lm, p, w, W):
    """Q @ W without forming Q: the five steps the device runs (xm_schur.hip), in numpy / scipy.sparse"""
    from scipy.sparse import coo_matrix
    cam = np.asarray(cam); lm = np.asarray(lm); p = np.asarray(p, dtype=np.float64); w = np.asarray(w, dtype=np.float64).reshape(-1)
    N, M, Q1, c, Q2, Q3 = schur_parts(cam, lm, p, w)
    o = W.shape[1]
    Wc = W.reshape(N, 3, o)
    g_lm = np.zeros((M, o)); np.add.at(g_lm, lm, -w[:, None] * np.einsum("ea,eak->ek", p, Wc[cam]))
    h = g_lm / Q3[:, None]
    r = np.einsum("ia,iak->ik", c, Wc)
    np.add.at(r, cam, w[:, None] * h[lm])
    V3 = coo_matrix((w, (cam, lm)), shape=(N, M)).tocsr()
    V3b = V3[1:]
    VT = np.diag(Q2[1:]) - (V3b.multiply(1.0 / Q3[None, :]) @ V3b.T).toarray()
    xc = np.zeros((N, o)); xc[1:] = np.linalg.solve(VT, r[1:])
    xl = h.copy(); tmp = np.zeros((M, o)); np.add.at(tmp, lm, w[:, None] * xc[cam]); xl += tmp / Q3[:, None]
    Y = np.einsum("iab,ibk->iak", Q1, Wc) - c[:, :, None] * xc[:, None, :]
    np.add.at(Y, cam, w[:, None, None] * p[:, :, None] * xl[lm][:, None, :])
    return Y.reshape(3 * N, o)


def schur_tp_numpy(cam, lm, p, w, R_real, s_real):
    """translations (3 x N, camera 1 at the origin) and landmarks (3 x M) of a solution: the minimiser of
    sum_obs w |s_i R_i p + t_i - P_l|^2 over (t, P) with t_1 = 0, i.e. -Qtp_bar^-1 Vtp_bar^T (sR)^T — what the reference computes as
    Abar @ sR_real.T (recoversolution.py:77-86) with the dense Abar of creatematrix.py:283-311.  R_real: 3 x 3N, s_real: N.
    The same four steps as the first four of schur_qw_numpy."""
    from scipy.sparse import coo_matrix
    cam = np.asarray(cam); lm = np.asarray(lm); p = np.asarray(p, dtype=np.float64); w = np.asarray(w, dtype=np.float64).reshape(-1)
    N, M, Q1, c, Q2, Q3 = schur_parts(cam, lm, p, w)
    R_real = np.asarray(R_real, dtype=np.float64); s_real = np.asarray(s_real, dtype=np.float64).reshape(-1)
    Wc = np.stack([(s_real[i] * R_real[:, 3 * i:3 * i + 3]).T for i in range(N)])          # (N, 3, 3): rows of (sR_real)^T
    h = np.zeros((M, 3)); np.add.at(h, lm, -w[:, None] * np.einsum("ea,eak->ek", p, Wc[cam])); h /= Q3[:, None]
    r = np.einsum("ia,iak->ik", c, Wc)
    np.add.at(r, cam, w[:, None] * h[lm])
    V3b = coo_matrix((w, (cam, lm)), shape=(N, M)).tocsr()[1:]
    VT = np.diag(Q2[1:]) - (V3b.multiply(1.0 / Q3[None, :]) @ V3b.T).toarray()
    xc = np.zeros((N, 3)); xc[1:] = np.linalg.solve(VT, r[1:])
    xl = h.copy(); tmp = np.zeros((M, 3)); np.add.at(tmp, lm, w[:, None] * xc[cam]); xl += tmp / Q3[:, None]
    return -xc.T, -xl.T


def gen_scene(N, M, views, seed, noise=0.01, hubs=3):
    """synthetic SfM scene for the matrix-free tests: N cameras and M landmarks in a box, every landmark observed by `views` cameras
    drawn at random (a well-connected co-visibility graph: the solve converges in a few hundred iterations, unlike a sequential
    trajectory whose path-like graph needs 10^4..10^5) plus `hubs` landmarks seen by EVERY camera (connectivity, and the degree
    skew real scenes have).  Returns dict(cam, lm, p, w, R_star, n, m): p = R_i^T (P_l - t_i) + noise = the camera-frame points
    create_matrix takes."""
    rng = np.random.default_rng(seed)
    Rs = haar_so3(rng, N)
    ts = rng.uniform(-5.0, 5.0, (N, 3))
    P = rng.uniform(-8.0, 8.0, (M, 3))
    cam = np.concatenate([np.repeat(np.arange(N), hubs), rng.integers(0, N, M * views)])
    lm = np.concatenate([np.tile(np.arange(hubs), N), np.repeat(np.arange(M), views)])
    _, idx = np.unique(cam.astype(np.int64) * M + lm, return_index=True)
    cam, lm = cam[idx], lm[idx]
    keep = np.bincount(lm, minlength=M)[lm] >= 2            # the reference drops landmarks seen once (2_test_creatematrix.py:100)
    cam, lm = cam[keep], lm[keep]
    orig, lm = np.unique(lm, return_inverse=True)           # compact landmark numbering
    pts = np.einsum("eba,eb->ea", Rs[cam], P[orig][lm] - ts[cam]) + noise * rng.standard_normal((cam.size, 3))
    w = rng.uniform(0.5, 1.5, cam.size)
    return dict(cam=cam.astype(np.int32), lm=lm.astype(np.int32), p=pts, w=w, R_star=Rs, n=N, m=int(orig.size))


def schur_residuals_numpy(cam, lm, p, w, U):
    """per-observation residual |p^T U_i + t_i - P_l|^2 at the optimal eliminated translations / landmarks for the scaled rows U
    (3N x o): [t; P] = -Qtp_bar^{-1} Vtp_bar^T U with t_0 = 0 (numpy / scipy.sparse; what xm_ctx_edge_residuals returns for a
    matrix-free context).  sum_e w_e res_e == <Q, U U^T>."""
    from scipy.sparse import coo_matrix
    cam = np.asarray(cam); lm = np.asarray(lm); p = np.asarray(p, dtype=np.float64); w = np.asarray(w, dtype=np.float64).reshape(-1)
    N, M, Q1, c, Q2, Q3 = schur_parts(cam, lm, p, w)
    o = U.shape[1]
    Uc = U.reshape(N, 3, o)
    pu = np.einsum("ea,eak->ek", p, Uc[cam])
    g_lm = np.zeros((M, o)); np.add.at(g_lm, lm, -w[:, None] * pu)
    q3i = np.where(Q3 > 0, 1.0 / np.where(Q3 > 0, Q3, 1.0), 0.0)
    h = g_lm * q3i[:, None]
    r = np.einsum("ia,iak->ik", c, Uc)
    np.add.at(r, cam, w[:, None] * h[lm])
    V3b = coo_matrix((w, (cam, lm)), shape=(N, M)).tocsr()[1:]
    VT = np.diag(Q2[1:]) - (V3b.multiply(q3i[None, :]) @ V3b.T).toarray()
    xc = np.zeros((N, o)); xc[1:] = np.linalg.solve(VT, r[1:])
    xl = h.copy(); tmp = np.zeros((M, o)); np.add.at(tmp, lm, w[:, None] * xc[cam]); xl += tmp * q3i[:, None]
    return np.sum((pu - xc[cam] + xl[lm]) ** 2, axis=1)


def env_tuning():
    """xm_tuning_t fields for a worker process of a test, handed over as JSON in XMT_TUNING (a variable of the TESTS: the library itself
    selects no kernel or layout through the environment) -> dict for xmamd.Context(tuning=...), or None"""
    import json as _json
    t = os.environ.get("XMT_TUNING")
    return _json.loads(t) if t else None
