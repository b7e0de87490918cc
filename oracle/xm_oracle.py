"""ctypes binding of the CPU oracle (oracle/libxm_oracle.so).

TEST INFRASTRUCTURE ONLY: may be imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under xm-code_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

VERBOSE = 1
FIX_STALE_SR = 2
CLOSED_FORM_Y = 4
TRACE_STRIDE = 6


class Stats(C.Structure):
    _fields_ = [("outer_iters", C.c_int32), ("stop_reason", C.c_int32), ("tcg_iters", C.c_int64),
                ("qw_products", C.c_int64), ("seconds", C.c_double), ("qw_seconds", C.c_double),
                ("trace_cap", C.c_int32), ("trace_len", C.c_int32), ("trace", C.POINTER(C.c_double))]


class Cert(C.Structure):
    _fields_ = [("min_eig", C.c_double), ("dual", C.c_double), ("gap", C.c_double),
                ("ls_residual", C.c_double), ("lscg_iters", C.c_int32), ("accepted", C.c_int32)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libxm_oracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libxm_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        dp = C.POINTER(C.c_double)
        L.xmo_trustregion.argtypes = [C.c_int, C.c_int, dp, dp, dp, dp, dp, C.c_double, dp, C.c_double, dp, dp,
                                      C.c_double, C.POINTER(Stats), C.c_uint]
        L.xmo_checkeig.argtypes = [C.c_int, C.c_int, dp, dp, C.c_double, dp, C.c_double, C.POINTER(Cert), C.c_uint]
        L.xmo_solve.argtypes = [C.c_int, dp, C.c_uint, C.c_double, C.c_double, C.c_double, C.c_int, dp, dp, dp,
                                C.POINTER(C.c_int), C.POINTER(Stats), C.POINTER(Cert), C.c_uint]
        L.xmo_solve_path.argtypes = [C.c_char_p, C.c_uint, C.c_double, C.c_double, C.c_double, C.c_int, C.c_uint]
        L.xmo_qw.argtypes = [C.c_int, C.c_int, dp, dp, dp, C.c_double]
        L.xmo_mgs_rows.argtypes = [C.c_int, C.c_int, dp, dp]
        L.xmo_syev_lower.argtypes = [C.c_int, dp, dp]
        L.xmo_num_threads.restype = C.c_int
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f(a):
    """column-major float64 copy"""
    return np.asfortranarray(np.array(a, dtype=np.float64, copy=True))


def qw(Cm, W, alpha=1.0):
    Cm = _f(Cm); W = _f(W)
    n = Cm.shape[0] // 3
    out = np.zeros_like(W, order="F")
    lib().xmo_qw(n, W.shape[1], _p(Cm), _p(W), _p(out), alpha)
    return out


def mgs_rows(A):
    A = _f(A)
    out = np.zeros_like(A, order="F")
    lib().xmo_mgs_rows(A.shape[0] // 3, A.shape[1], _p(A), _p(out))
    return out


def syev_lower(A):
    A = _f(A)
    w = np.zeros(A.shape[0])
    rc = lib().xmo_syev_lower(A.shape[0], _p(A), _p(w))
    assert rc == 0
    return w, A


def _stats_dict(st, trace):
    d = {k: getattr(st, k) for k in ("outer_iters", "stop_reason", "tcg_iters", "qw_products", "seconds", "qw_seconds")}
    if trace is not None:
        d["trace"] = trace[: st.trace_len].copy()
    return d


def trustregion(Cm, R0, s0_ex, lam=0.0, gradtol=1e-6, linesearch_step=0.0, v=None, maxtime=1000.0, flags=0, trace=0):
    """tr.h:77.  Returns R (3n x o), s_ex (n), primal, gradtol_out, stats."""
    Cm = _f(Cm); R0 = _f(R0)
    n = Cm.shape[0] // 3
    o = R0.shape[1]
    s0 = np.ascontiguousarray(s0_ex, dtype=np.float64).reshape(-1).copy()
    R = np.zeros_like(R0, order="F"); s = np.zeros(n)
    vv = np.zeros(3 * n) if v is None else np.ascontiguousarray(v, dtype=np.float64).reshape(-1).copy()
    gt = C.c_double(gradtol); pr = C.c_double(0.0)
    st = Stats()
    tr = None
    if trace:
        tr = np.zeros((trace, TRACE_STRIDE)); st.trace_cap = trace; st.trace = _p(tr)
    lib().xmo_trustregion(n, o, _p(Cm), _p(R0), _p(s0), _p(R), _p(s), lam, C.byref(gt), linesearch_step, _p(vv),
                          C.byref(pr), maxtime, C.byref(st), flags)
    return R, s, pr.value, gt.value, _stats_dict(st, tr)


def numa_prepare(Cm):
    """bench.py's cpu_baseline only: a NUMA-distributed first-touch copy of the dense matrix for the host product"""
    Cm = _f(Cm)
    return lib().xmo_numa_prepare(Cm.shape[0] // 3, _p(Cm))


def numa_release():
    lib().xmo_numa_release()


def trustregion_bsr(rowptr, colidx, blocks, R0, s0_ex, lam=0.0, gradtol=1e-6, maxtime=1000.0, flags=0, trace=0):
    """The same trust region with Q given as 3x3-block CSR (test-only extension of the oracle: the reference's Q is dense).
    Returns what trustregion() returns."""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64); colidx = np.ascontiguousarray(colidx, dtype=np.int32)
    blocks = np.ascontiguousarray(blocks, dtype=np.float64)
    R0 = _f(R0)
    n = rowptr.size - 1
    o = R0.shape[1]
    s0 = np.ascontiguousarray(s0_ex, dtype=np.float64).reshape(-1).copy()
    R = np.zeros_like(R0, order="F"); s = np.zeros(n)
    vv = np.zeros(3 * n)
    gt = C.c_double(gradtol); pr = C.c_double(0.0)
    st = Stats()
    tr = None
    if trace:
        tr = np.zeros((trace, TRACE_STRIDE)); st.trace_cap = trace; st.trace = _p(tr)
    L = lib()
    L.xmo_set_bsr.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.xmo_set_bsr(rowptr.ctypes.data, colidx.ctypes.data, blocks.ctypes.data)
    try:
        L.xmo_trustregion(n, o, None, _p(R0), _p(s0), _p(R), _p(s), lam, C.byref(gt), 0.0, _p(vv), C.byref(pr), maxtime,
                          C.byref(st), flags)
    finally:
        L.xmo_set_bsr(None, None, None)
    return R, s, pr.value, gt.value, _stats_dict(st, tr)


def checkeig(Cm, sR, lam, primal, flags=0):
    """ce.h:42.  Returns accepted(bool), v (3n), cert dict."""
    Cm = _f(Cm); sR = _f(sR)
    n = Cm.shape[0] // 3
    v = np.zeros(3 * n); ce = Cert()
    ok = lib().xmo_checkeig(n, sR.shape[1], _p(Cm), _p(sR), lam, _p(v), primal, C.byref(ce), flags)
    return bool(ok), v, {k: getattr(ce, k) for k in ("min_eig", "dual", "gap", "ls_residual", "lscg_iters", "accepted")}


def solve(Cm, max_rank, tol, lam, max_time, mode=0, s_ini=None, flags=0, trace=0):
    """main.cu:180 (mode 0) / :312 (mode 1) / :35 (mode 2).  Returns R (3n x rank), s (n), info dict."""
    Cm = _f(Cm)
    n = Cm.shape[0] // 3
    rmax = max(int(max_rank), 3)
    R = np.zeros((3 * n, rmax + 1), order="F"); s = np.zeros(n)
    rank = C.c_int(0); st = Stats(); ce = Cert()
    tr = None
    if trace:
        tr = np.zeros((trace, TRACE_STRIDE)); st.trace_cap = trace; st.trace = _p(tr)
    si = None if s_ini is None else np.ascontiguousarray(s_ini, dtype=np.float64).reshape(-1).copy()
    status = lib().xmo_solve(n, _p(Cm), int(max_rank), tol, lam, max_time, mode, None if si is None else _p(si),
                             _p(R), _p(s), C.byref(rank), C.byref(st), C.byref(ce), flags)
    info = _stats_dict(st, tr)
    info.update(status=status, rank=rank.value,
                cert={k: getattr(ce, k) for k in ("min_eig", "dual", "gap", "ls_residual", "lscg_iters", "accepted")})
    return np.ascontiguousarray(R[:, : rank.value]), s, info


def solve_path(path, max_rank, tol, lam, max_time, mode=0, flags=0):
    return lib().xmo_solve_path(os.fsencode(path), int(max_rank), tol, lam, max_time, mode, flags)


_SYEV_CB = None   # keeps the ctypes callback alive while the library holds it


def use_lapack_eig(on=True):
    """Eigen step of checkeig through LAPACK dsyevd (scipy) instead of the restated tred2 / tql2 -- the closest CPU analogue of
    cusolverDnXsyevd (Dense/eig.h:35-73).  For bench.py's CPU wall-clock-to-KKT leg on the headline size (5334 rows: an hour -> seconds);
    tests/test_oracle.py asserts that both give the same certificate on the small golden cases."""
    global _SYEV_CB
    L = lib()
    FN = C.CFUNCTYPE(C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double))
    L.xmo_set_syev.argtypes = [FN]
    if not on:
        L.xmo_set_syev(C.cast(None, FN))
        _SYEV_CB = None
        return
    from scipy.linalg import lapack

    def cb(m, a_ptr, w_ptr):
        A = np.ctypeslib.as_array(a_ptr, shape=(m * m,)).reshape((m, m), order="F")     # column-major view of the caller's buffer
        w, v, info = lapack.dsyevd(A, compute_v=1, lower=1, overwrite_a=0)
        if info != 0:
            return int(info)
        A[:, :] = v
        np.ctypeslib.as_array(w_ptr, shape=(m,))[:] = w
        return 0

    _SYEV_CB = FN(cb)
    L.xmo_set_syev(_SYEV_CB)


def num_threads():
    return lib().xmo_num_threads()
