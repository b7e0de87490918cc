"""ctypes binding of the C ABI in include/xm_amd.h (libxm_amd.so) — the host-side mirror used by the tests,
bench.py and __graft_entry__.  The product path is the HIP library; this file only marshals numpy arrays.

There is NO CPU fallback: every compute entry point raises XmError when the library or a GPU is missing.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("XMAMD_LIB") or os.path.join(_HERE, "lib", "libxm_amd.so")   # XMAMD_LIB: a development aid of this binding (A/B builds)
MODULE_DIR = os.path.join(_HERE, "build")      # holds XM.cpython-*.so (the reference's module name)

STORAGE_DENSE, STORAGE_BSR3 = 0, 1
STORAGE_BSR3_DENSE = 2   # BSR3 on the host, expanded to the dense layout on the device (each rank: its own rows)
STORAGE_SCHUR = 3        # matrix-free: the observation list of the reference's create_matrix (cam, lm, p, w)
STORAGE_VIEWGRAPH = 4    # the view-graph edge list (ei, ej, w, M): block CSR + quaternion-compressed sliced ELL on the device
RETRACT_QR, RETRACT_POLAR = 0, 1
MODE_SOLVE, MODE_RANK3, MODE_REBUTTLE = 0, 1, 2
FLAG_VERBOSE, FLAG_FIX_STALE_SR, FLAG_PROFILE_QW, FLAG_HOST_STEPPED = 1, 2, 4, 8
CERT_EIG_NOT_CONVERGED = 1
CERT_EIG_EXACT = 2           # small problem: the certificate's tridiagonalisation ran to completion (dense route)
FLAG_WARM_R = 16
FLAG_HOST_OUTER = 64           # outer iteration of the trust region on the host instead of the device (xm_amd.h)
FLAG_DEVICE_OUTER = 128        # ... on the device also with dense products, where the host-driven form is the (faster) default (xm_amd.h)
FLAG_MODEL_RECURRENCE = 32     # model decrease of a tCG from its recurrences instead of from accumulated H v (xm_amd.h)

EXPORTS = [
    "xm_last_error", "xm_version", "xm_abi_revision", "xm_solve", "xm_solve_rank3", "xm_solve_rebuttle", "xm_ctx_create", "xm_ctx_solve",
    "xm_ctx_destroy", "xm_dense_ld", "xm_dev_count", "xm_dev_alloc", "xm_dev_free", "xm_dev_h2d", "xm_dev_d2h",
    "xm_dev_sync", "xm_dense_upload", "xm_dense_from_bsr3", "xm_qw_dense", "xm_qw_dense_sym", "xm_qw_bsr3", "xm_retract", "xm_retract_polar", "xm_recover_rotations",
    "xm_comm_unique_id", "xm_comm_init", "xm_comm_init_shm", "xm_comm_init_ipc", "xm_comm_finalize", "xm_partition", "xm_partition_blocks",
    "xm_symv_plan", "xm_sell_layout", "xm_sell_locality", "xm_sell_create", "xm_sell_create2", "xm_sell_quat_roundtrip", "xm_sell_destroy", "xm_qw_sell", "xm_qw_sell_padded",
    "xm_ctx_attach_edges", "xm_ctx_edge_residuals", "xm_ctx_edge_residuals_recovered", "xm_ctx_xm2_filter", "xm_ctx_xm2_round", "xm_ctx_set_edge_weights", "xm_ctx_recover_tp", "xm_ctx_schur_info", "xm_ctx_qw", "xm_spd_inverse", "xm_ctx_transport", "xm_ctx_sell_wpad", "xm_ctx_product_kind", "xm_symw_plan", "xm_symw_use",
]
# include/xm_bench.h: timing hooks of the micro-benchmarks (same library, not part of the product ABI)
BENCH_EXPORTS = ["xm_bench_last_error", "xm_qw_dense_time", "xm_qw_dense_sym_time", "xm_bench_symv_k", "xm_bench_dense_policy", "xm_qw_dense_sym_trace", "xm_qw_dense_strip_time", "xm_qw_dense_strip_ks", "xm_qw_bsr3_time", "xm_bench_bsr_binned", "xm_qw_sell_time",
                 "xm_retract_variant", "xm_recover_rotations_variant", "xm_peer_allgather_bench", "xm_qw_symw_time", "xm_bench_grid_barrier"]
PRODUCT_KINDS = {0: "dense", 1: "dense_sym", 2: "bsr3", 3: "sell", 4: "sell_quat", 5: "schur"}


class XmError(RuntimeError):
    pass


class Tuning(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("sym", "sym_min_rows", "sell", "sell_slabs", "sell_lmax", "sell_gather", "sell_codec", "overlap",
                                         "overlap_min_mb", "cert_dense_rows", "lanczos_mmax", "lanczos_restarts", "watchdog_s", "balance",
                                         "exchange", "split_k", "sell_wpad", "exchange_fence", "schur_host_assembly", "schur_trace",
                                         "schur_solver", "schur_dense_max", "debug_drop_finalize", "debug_peer_mute", "schur_pcg_first",
                                         "schur_pcg_hess_digits")] + [("reserved", C.c_int32 * 2)]


class Problem(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n", C.c_int64), ("storage", C.c_int32), ("q_on_device", C.c_int32), ("q", C.c_void_p),
                ("ldq", C.c_int64), ("nb", C.c_int64), ("rowptr", C.c_void_p), ("colidx", C.c_void_p),
                ("blocks", C.c_void_p), ("nobs", C.c_int64), ("n_landmarks", C.c_int64), ("obs_cam", C.c_void_p), ("obs_lm", C.c_void_p),
                ("obs_p", C.c_void_p), ("obs_w", C.c_void_p), ("q_row0", C.c_int64),
                ("ne", C.c_int64), ("edge_i", C.c_void_p), ("edge_j", C.c_void_p), ("edge_w", C.c_void_p), ("edge_M", C.c_void_p),
                ("n_gpus", C.c_int32), ("gpu_map", C.c_int32), ("tuning", C.POINTER(Tuning))]


class Options(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("max_rank", C.c_uint32), ("tol", C.c_double), ("lam", C.c_double), ("max_time", C.c_double),
                ("mode", C.c_int32), ("flags", C.c_uint32), ("s_ini", C.c_void_p), ("trace_cap", C.c_int32),
                ("trace", C.c_void_p), ("R_ini", C.c_void_p), ("retraction", C.c_int32), ("sum_grouping", C.c_int32)]


class Result(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("R", C.c_void_p), ("s", C.c_void_p), ("rank", C.c_int32), ("status", C.c_int32),
                ("primal", C.c_double), ("dual", C.c_double), ("min_eig", C.c_double), ("gap", C.c_double),
                ("tcg_iters", C.c_int64), ("outer_iters", C.c_int64), ("qw_products", C.c_int64),
                ("lanczos_iters", C.c_int64), ("seconds", C.c_double), ("tr_seconds", C.c_double),
                ("cert_seconds", C.c_double), ("qw_ms_sum", C.c_double), ("qw_ms_count", C.c_int64),
                ("qw_bytes", C.c_int64), ("trace_len", C.c_int32), ("last_stop_reason", C.c_int32),
                ("sym_product", C.c_int32), ("cert_flags", C.c_int32), ("eig_residual", C.c_double),
                ("n_gpus", C.c_int32), ("exchange", C.c_int32), ("qw_stream_bytes", C.c_int64),
                ("outer_on_device", C.c_int32), ("reserved_", C.c_int32)]


class Xm2Info(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("percentile", C.c_double), ("threshold", C.c_double), ("removed", C.c_int64),
                ("s_avg", C.c_double), ("s_std", C.c_double), ("n_small", C.c_int64), ("regularised", C.c_int32), ("rank3_status", C.c_int32),
                ("lam_used", C.c_double), ("rank3_tcg_iters", C.c_int64)]


_lib = None


def lib():
    """Load libxm_amd.so (built by `make -C xm-code_amd` / __graft_entry__.build()).  Fails loudly when absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise XmError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)")
        # "virtual devices" (several ranks of a single-process multi-GPU context on one GPU) need a hardware queue per rank's stream; the
        # HIP runtime reads this when it initialises, i.e. at the first call into the library
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
        L = C.CDLL(LIB_PATH)
        L.xm_last_error.restype = C.c_char_p
        L.xm_version.restype = C.c_char_p
        L.xm_dense_ld.restype = C.c_int64
        L.xm_dense_ld.argtypes = [C.c_int64]
        for name in ("xm_solve", "xm_solve_rank3"):
            getattr(L, name).argtypes = [C.c_char_p, C.c_uint, C.c_double, C.c_double, C.c_double]
        L.xm_solve_rebuttle.argtypes = [C.c_char_p, C.c_uint, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_int)]
        L.xm_ctx_create.argtypes = [C.POINTER(Problem), C.POINTER(C.c_void_p)]
        L.xm_ctx_solve.argtypes = [C.c_void_p, C.POINTER(Options), C.POINTER(Result)]
        L.xm_ctx_destroy.argtypes = [C.c_void_p]
        L.xm_ctx_destroy.restype = None
        L.xm_spd_inverse.argtypes = [C.c_int64, C.c_void_p]
        L.xm_ctx_qw.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_double]
        L.xm_ctx_attach_edges.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.xm_ctx_edge_residuals.argtypes = [C.c_void_p, C.c_void_p]
        L.xm_ctx_set_edge_weights.argtypes = [C.c_void_p, C.c_void_p]
        L.xm_ctx_edge_residuals_recovered.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.xm_ctx_xm2_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_void_p]
        L.xm_ctx_xm2_round.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Options), C.POINTER(Xm2Info), C.POINTER(Result)]
        L.xm_ctx_recover_tp.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.xm_ctx_transport.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_char_p, C.c_size_t]
        L.xm_ctx_schur_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.POINTER(C.c_double)]
        L.xm_ctx_sell_wpad.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.xm_ctx_product_kind.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.xm_bench_last_error.restype = C.c_char_p
        L.xm_symw_plan.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.xm_symw_use.argtypes = [C.c_int, C.c_int, C.c_int]
        L.xm_qw_symw_time.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int64)]
        L.xm_bench_grid_barrier.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.xm_dev_count.argtypes = [C.POINTER(C.c_int)]
        L.xm_dev_alloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        L.xm_dev_free.argtypes = [C.c_void_p]
        L.xm_dev_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.xm_dev_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.xm_dense_upload.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_void_p)]
        L.xm_dense_from_bsr3.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]
        L.xm_qw_dense.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
        L.xm_qw_dense_sym.argtypes = L.xm_qw_dense.argtypes
        L.xm_qw_dense_sym_time.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double)]
        L.xm_bench_symv_k.argtypes = [C.c_int, C.c_int, C.c_int]
        L.xm_qw_dense_sym_trace.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int)]
        L.xm_qw_bsr3.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
                                 C.c_double, C.c_void_p]
        L.xm_retract.argtypes = [C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                 C.c_void_p, C.c_void_p, C.c_void_p]
        L.xm_retract_polar.argtypes = L.xm_retract.argtypes
        L.xm_retract_variant.argtypes = [C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p,
                                         C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.xm_qw_dense_time.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                       C.POINTER(C.c_double)]
        L.xm_qw_dense_strip_time.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double)]
        L.xm_qw_dense_strip_ks.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_double),
                                           C.POINTER(C.c_int)]
        L.xm_peer_allgather_bench.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int, C.POINTER(C.c_double)]
        L.xm_qw_bsr3_time.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                      C.POINTER(C.c_double)]
        L.xm_recover_rotations.argtypes = [C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.xm_recover_rotations_variant.argtypes = L.xm_recover_rotations.argtypes + [C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.xm_comm_unique_id.argtypes = [C.c_char_p]
        L.xm_comm_init.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p]
        L.xm_comm_init_shm.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t]
        L.xm_comm_init_ipc.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_double]
        L.xm_partition.argtypes = [C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.xm_partition_blocks.argtypes = [C.c_int64, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.xm_sell_layout.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p] + [C.c_void_p] * 7
        L.xm_sell_locality.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p]
        L.xm_sell_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.xm_sell_create2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_void_p)]
        L.xm_sell_quat_roundtrip.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.xm_sell_destroy.argtypes = [C.c_void_p]
        L.xm_sell_destroy.restype = None
        L.xm_qw_sell.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_void_p]
        L.xm_qw_sell_padded.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_void_p]
        L.xm_qw_sell_time.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
        _lib = L
    return _lib


def _chk(rc):
    if rc != 0:
        msg = lib().xm_last_error().decode() or lib().xm_bench_last_error().decode()   # the timing hooks keep their own message (xm_bench.h)
        raise XmError(f"xm_amd error {rc}: {msg}")


def device_count():
    c = C.c_int(0)
    lib().xm_dev_count(C.byref(c))
    return c.value


def require_gpu():
    if device_count() < 1:
        raise XmError("no HIP device visible: the XM solver has no CPU fallback")


def pitch_of(o):
    return o | 1


def dense_ld(n):
    return int(lib().xm_dense_ld(n))


# ------------------------------------------------------------------------------------------------ device buffers
class DevArray:
    """A float64 / int device buffer owned through the C ABI (no torch needed)."""

    def __init__(self, host=None, nbytes=None):
        self.ptr = C.c_void_p()
        self.nbytes = int(host.nbytes if host is not None else nbytes)
        _chk(lib().xm_dev_alloc(C.byref(self.ptr), self.nbytes))
        if host is not None:
            host = np.ascontiguousarray(host)
            _chk(lib().xm_dev_h2d(self.ptr, host.ctypes.data_as(C.c_void_p), self.nbytes))

    def get(self, dtype=np.float64, shape=None):
        out = np.empty(self.nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        _chk(lib().xm_dev_d2h(out.ctypes.data_as(C.c_void_p), self.ptr, self.nbytes))
        return out if shape is None else out.reshape(shape)

    def free(self):
        if self.ptr:
            lib().xm_dev_free(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def to_rm(M, o=None, rows=None):
    """host (m x o) matrix -> the device layout: row-major, pitch OP = o|1, `rows` rows (zero padded)."""
    M = np.asarray(M, dtype=np.float64)
    m, o = M.shape
    out = np.zeros((rows or m, pitch_of(o)))
    out[:m, :o] = M
    return out


def from_rm(buf, m, o):
    return np.asarray(buf).reshape(-1, pitch_of(o))[:m, :o].copy()


# ------------------------------------------------------------------------------------------------ kernel-level calls
def dense_upload(Q):
    Q = np.asfortranarray(np.asarray(Q, dtype=np.float64))
    n = Q.shape[0] // 3
    p = C.c_void_p()
    _chk(lib().xm_dense_upload(Q.ctypes.data_as(C.c_void_p), Q.shape[0], n, C.byref(p)))
    d = DevArray.__new__(DevArray)
    d.ptr, d.nbytes = p, 3 * n * dense_ld(n) * 8
    return d


def dense_from_bsr3(rowptr, colidx, blocks):
    """device-side densification of a 3x3-block CSR matrix into the solver's padded row-major layout"""
    require_gpu()
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64); colidx = np.ascontiguousarray(colidx, dtype=np.int32)
    blocks = np.ascontiguousarray(blocks, dtype=np.float64)
    n = rowptr.size - 1
    p = C.c_void_p()
    _chk(lib().xm_dense_from_bsr3(rowptr.ctypes.data_as(C.c_void_p), colidx.ctypes.data_as(C.c_void_p),
                                  blocks.ctypes.data_as(C.c_void_p), n, C.byref(p)))
    d = DevArray.__new__(DevArray)
    d.ptr, d.nbytes = p, 3 * n * dense_ld(n) * 8
    return d


def pad16(W):
    """W (3n x o, o <= 5) at a record pitch of 16 doubles: camera c's 3 x pitch_of(o) row-major block at [16 c, 16 c + 3 pitch_of(o))"""
    W = np.asarray(W, dtype=np.float64)
    n, o = W.shape[0] // 3, W.shape[1]
    rec = 3 * pitch_of(o)
    out = np.zeros((n, 16))
    out[:, :rec] = to_rm(W).reshape(-1)[: n * rec].reshape(n, rec)
    return out.reshape(-1)


def qw_dense(Q, W, alpha=1.0, dq=None, sym=False):
    """alpha * Q @ W on the GPU through xm_qw_dense (Q: 3n x 3n, W: 3n x o); sym=True: the half-traffic symmetric kernel."""
    require_gpu()
    W = np.asarray(W, dtype=np.float64)
    n, o = W.shape[0] // 3, W.shape[1]
    own = dq is None
    dq = dq or dense_upload(Q)
    dW = DevArray(to_rm(W, rows=dense_ld(n)))
    dO = DevArray(nbytes=3 * n * pitch_of(o) * 8)
    _chk((lib().xm_qw_dense_sym if sym else lib().xm_qw_dense)(dq.ptr, n, o, dW.ptr, dO.ptr, alpha, None))
    _chk(lib().xm_dev_sync())
    out = from_rm(dO.get(), 3 * n, o)
    for b in (dW, dO) + ((dq,) if own else ()):
        b.free()
    return out


def qw_dense_strip(Qs, n, W, alpha=1.0, ks=0, reps=0):
    """alpha * Qs @ W for a ROW STRIP Qs (3 nloc x 3n, the rows one rank of a row partition owns) through the column-split kernel
    (ks workgroups per camera group; 0 = policy).  Returns (product, ks used, average ms when reps > 0)."""
    require_gpu()
    Qs = np.asarray(Qs, dtype=np.float64); W = np.asarray(W, dtype=np.float64)
    nloc, o, ld = Qs.shape[0] // 3, W.shape[1], dense_ld(n)
    Qp = np.zeros((3 * nloc, ld)); Qp[:, :3 * n] = Qs
    dq = DevArray(Qp); dW = DevArray(to_rm(W, rows=ld)); dO = DevArray(nbytes=3 * nloc * pitch_of(o) * 8)
    ms = C.c_double(); used = C.c_int()
    _chk(lib().xm_qw_dense_strip_ks(dq.ptr, nloc, n, o, dW.ptr, dO.ptr, alpha, ks, reps, C.byref(ms), C.byref(used)))
    out = from_rm(dO.get(), 3 * nloc, o)
    for b in (dq, dW, dO):
        b.free()
    return out, used.value, ms.value


def qw_bsr3(rowptr, colidx, blocks, W, alpha=1.0):
    require_gpu()
    W = np.asarray(W, dtype=np.float64)
    n, o = W.shape[0] // 3, W.shape[1]
    drp = DevArray(np.asarray(rowptr, dtype=np.int64)); dci = DevArray(np.asarray(colidx, dtype=np.int32))
    dbl = DevArray(np.asarray(blocks, dtype=np.float64).reshape(-1))
    dW = DevArray(to_rm(W)); dO = DevArray(nbytes=3 * n * pitch_of(o) * 8)
    _chk(lib().xm_qw_bsr3(drp.ptr, dci.ptr, dbl.ptr, n, o, dW.ptr, dO.ptr, alpha, None))
    _chk(lib().xm_dev_sync())
    out = from_rm(dO.get(), 3 * n, o)
    for b in (drp, dci, dbl, dW, dO):
        b.free()
    return out


def spd_inverse(A):
    """inverse of a symmetric positive definite matrix on the GPU (xm_spd_inverse)"""
    require_gpu()
    A = np.asfortranarray(np.array(A, dtype=np.float64))
    _chk(lib().xm_spd_inverse(A.shape[0], A.ctypes.data_as(C.c_void_p)))
    return np.ascontiguousarray(A)


def quat_roundtrip(block):
    """host-only: (stored quaternion, block rebuilt by the product kernel) of a 3x3 block -w * rotation"""
    b = np.ascontiguousarray(block, dtype=np.float64).reshape(9)
    q = np.zeros(4); r = np.zeros(9)
    _chk(lib().xm_sell_quat_roundtrip(b.ctypes.data_as(C.c_void_p), q.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p)))
    return q, r.reshape(3, 3)


def sell_layout(rowptr, colidx, ncols=None, slabs=4, lmax=64):
    """host-side description of the sliced-ELL layout (xm_sell.h) -- no GPU involved; used by the CPU tests"""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64); colidx = np.ascontiguousarray(colidx, dtype=np.int32)
    n = rowptr.size - 1
    ncols = n if ncols is None else ncols
    sizes = np.zeros(5, dtype=np.int64)
    args = (rowptr.ctypes.data_as(C.c_void_p), colidx.ctypes.data_as(C.c_void_p), n, ncols, slabs, lmax)
    _chk(lib().xm_sell_layout(*args, sizes.ctypes.data_as(C.c_void_p), *([None] * 7)))
    nsl, nst, npart, nvr, nstore = (int(x) for x in sizes)
    out = dict(nslices=nsl, nsteps=nst, nparts=npart, nvrows=nvr, nstore=nstore, slabs=slabs, ridx=np.zeros(max(npart, 1), dtype=np.int32),
               slice_off=np.zeros(nsl + 1, dtype=np.int64), slab_start=np.zeros(slabs + 1, dtype=np.int32),
               kind=np.zeros(max(nst, 1), dtype=np.uint8), src=np.zeros(max(nst, 1) * 64, dtype=np.int64),
               pslot=np.zeros(max(nsl, 1) * 64, dtype=np.int32), pptr=np.zeros(n + 1, dtype=np.int64))
    _chk(lib().xm_sell_layout(*args, sizes.ctypes.data_as(C.c_void_p),
                              *(out[k].ctypes.data_as(C.c_void_p) for k in ("slice_off", "slab_start", "kind", "src", "pslot", "pptr", "ridx"))))
    return out


def sell_locality(rowptr, colidx, ncols=None, slabs=4, lmax=64):
    """(lines at the native pitch of 72-byte records, of 120-byte records, at the 128-byte pitch) -- host only (xm_sell_locality)"""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64); colidx = np.ascontiguousarray(colidx, dtype=np.int32)
    n = rowptr.size - 1
    out = np.zeros(3, dtype=np.int64)
    _chk(lib().xm_sell_locality(rowptr.ctypes.data_as(C.c_void_p), colidx.ctypes.data_as(C.c_void_p), n, n if ncols is None else ncols, slabs, lmax,
                                out.ctypes.data_as(C.c_void_p)))
    return tuple(int(x) for x in out)


def symw_plan(ntot, nloc, cam0, K=0):
    """work list of one rank of the multi-rank symmetric window product (xm_symw.h) -- host only"""
    geom = np.zeros(8, dtype=np.int32)
    _chk(lib().xm_symw_plan(ntot, nloc, cam0, K, geom.ctypes.data_as(C.c_void_p), None))
    items = np.zeros((max(int(geom[7]), 1), 3), dtype=np.int32)
    _chk(lib().xm_symw_plan(ntot, nloc, cam0, K, geom.ctypes.data_as(C.c_void_p), items.ctypes.data_as(C.c_void_p)))
    keys = ("T", "Th", "tie", "t0", "nsteps", "nstrips", "K", "nitems")
    out = {k: int(v) for k, v in zip(keys, geom)}
    out["items"] = items[: out["nitems"]]
    return out


class SellMatrix:
    """3x3-block sparse Q in the sliced-ELL device layout (xm_sell_create); product through xm_qw_sell"""

    def __init__(self, rowptr, colidx, blocks, ncols=None, slabs=4, lmax=0, codec=0, row0=0):
        """codec 1 = view-graph codec (quaternion per off-diagonal block, scalar per diagonal block)"""
        require_gpu()
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int64); colidx = np.ascontiguousarray(colidx, dtype=np.int32)
        blocks = np.ascontiguousarray(blocks, dtype=np.float64)
        self.n = rowptr.size - 1
        self.h = C.c_void_p()
        _chk(lib().xm_sell_create2(rowptr.ctypes.data_as(C.c_void_p), colidx.ctypes.data_as(C.c_void_p), blocks.ctypes.data_as(C.c_void_p),
                                   self.n, self.n if ncols is None else ncols, slabs, lmax, codec, row0, C.byref(self.h)))

    def qw(self, W, alpha=1.0, gather=1, padded=False):
        """gather: 0 a record of W per lane | 1 LDS-transposed (the solver's default).  padded=True: the input is also handed over at a
        record pitch of 16 doubles (xm_qw_sell_padded; o = 3..5)"""
        W = np.asarray(W, dtype=np.float64)
        o = W.shape[1]
        dW = DevArray(to_rm(W)); dO = DevArray(nbytes=3 * self.n * pitch_of(o) * 8)
        dP = DevArray(pad16(W)) if padded else None
        _chk(lib().xm_qw_sell_padded(self.h, o, dW.ptr, dP.ptr if padded else None, dO.ptr, alpha, gather, None))
        _chk(lib().xm_dev_sync())
        out = from_rm(dO.get(), 3 * self.n, o)
        dW.free(); dO.free()
        if padded:
            dP.free()
        return out

    def close(self):
        if self.h:
            lib().xm_sell_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def retract(R, s, D, ds, t, polar=False):
    """(MGS_rows(R + t D), s*exp(t ds/s)) through xm_retract; polar=True: the polar retraction (xm_retract_polar)."""
    require_gpu()
    R = np.asarray(R, dtype=np.float64)
    n, o = R.shape[0] // 3, R.shape[1]
    dR = DevArray(to_rm(R)); dD = DevArray(to_rm(D)); dsv = DevArray(np.asarray(s, dtype=np.float64))
    dds = DevArray(np.asarray(ds, dtype=np.float64))
    dRo = DevArray(nbytes=dR.nbytes); dso = DevArray(nbytes=dsv.nbytes)
    _chk((lib().xm_retract_polar if polar else lib().xm_retract)(n, o, dR.ptr, dsv.ptr, dD.ptr, dds.ptr, t, dRo.ptr, dso.ptr, None))
    _chk(lib().xm_dev_sync())
    out = from_rm(dRo.get(), 3 * n, o), dso.get()
    for b in (dR, dD, dsv, dds, dRo, dso):
        b.free()
    return out


def recover_rotations(R, s, variant=None, reps=0):
    """anchored O(3) rotations (3 x 3n) and scales of a solution (R: 3n x r, s: n) through xm_recover_rotations.  variant (xm_bench.h): 0 the
    default kernel, 1 one wavefront per camera; with reps > 0 a fourth value is returned: average ms of the projection launch"""
    require_gpu()
    R = np.asfortranarray(np.asarray(R, dtype=np.float64)); s = np.ascontiguousarray(np.asarray(s, dtype=np.float64).reshape(-1))
    n, r = s.size, R.shape[1]
    rot = np.zeros((3, 3 * n), order="F"); sc = np.zeros(n); neg = C.c_int(0)
    args = (n, r, R.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p), rot.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p), C.byref(neg))
    if variant is None:
        _chk(lib().xm_recover_rotations(*args))
        return np.ascontiguousarray(rot), sc, neg.value
    ms = C.c_double(0.0)
    _chk(lib().xm_recover_rotations_variant(*args, int(variant), int(reps), C.byref(ms)))
    return (np.ascontiguousarray(rot), sc, neg.value) + ((ms.value,) if reps > 0 else ())


# ------------------------------------------------------------------------------------------------ context API
class Context:
    """Q resident in HBM; solve() == the reference's staircase (XM_main.cu:180 / :312 / :35)."""

    def __init__(self, Q=None, bsr=None, dq=None, n=None, densify=False, obs=None, vg=None, n_gpus=1, gpu_map=0, tuning=None):
        """vg = (ei, ej, w, M): view-graph edge list (STORAGE_VIEWGRAPH).  n_gpus > 1: single-process row partition over that many GPUs
        (gpu_map=1: all ranks on device 0).  tuning: dict of xm_tuning_t fields."""
        require_gpu()
        self._keep = []
        p = Problem()
        p.struct_size = C.sizeof(Problem)
        p.n_gpus, p.gpu_map = int(n_gpus), int(gpu_map)
        if tuning:
            tn = Tuning()
            for k, v in tuning.items():
                setattr(tn, k, int(v))
            p.tuning = C.pointer(tn)
            self._keep.append(tn)
        if vg is not None:
            ei, ej, w, M = vg
            ei = np.ascontiguousarray(ei, dtype=np.int32); ej = np.ascontiguousarray(ej, dtype=np.int32)
            w = np.ascontiguousarray(w, dtype=np.float64).reshape(-1); M = np.ascontiguousarray(M, dtype=np.float64).reshape(-1, 9)
            assert ei.size == ej.size == w.size == M.shape[0]
            self.n = int(max(ei.max(), ej.max())) + 1 if n is None else int(n)
            p.n, p.storage, p.ne = self.n, STORAGE_VIEWGRAPH, ei.size
            p.edge_i, p.edge_j, p.edge_w, p.edge_M = (a.ctypes.data_as(C.c_void_p) for a in (ei, ej, w, M))
            self._keep += [ei, ej, w, M]
            self.ne = ei.size
        elif obs is not None:                     # matrix-free: (cam, lm, p, w) = (edges[:, 0] - 1, edges[:, 1] - 1, landmarks, weight)
            cam, lm, pts, w = obs
            cam = np.ascontiguousarray(cam, dtype=np.int32); lm = np.ascontiguousarray(lm, dtype=np.int32)
            pts = np.ascontiguousarray(pts, dtype=np.float64); w = np.ascontiguousarray(w, dtype=np.float64).reshape(-1)
            self.n = int(cam.max()) + 1 if n is None else int(n)
            p.n, p.storage, p.nobs, p.n_landmarks = self.n, STORAGE_SCHUR, cam.size, int(lm.max()) + 1
            self.n_landmarks = int(lm.max()) + 1
            p.obs_cam, p.obs_lm, p.obs_p, p.obs_w = (a.ctypes.data_as(C.c_void_p) for a in (cam, lm, pts, w))
            self._keep += [cam, lm, pts, w]
            self.ne = cam.size                      # residuals / weights of the XM^2 loop are per observation
        elif dq is not None:                      # dense Q already on the device in the solver's layout (borrowed)
            self.n = int(n)
            p.n, p.storage, p.q_on_device, p.q, p.ldq = self.n, STORAGE_DENSE, 1, dq.ptr, dense_ld(self.n)
            self._dq = dq
        elif Q is not None:
            Q = np.asfortranarray(np.asarray(Q, dtype=np.float64))
            self.n = Q.shape[0] // 3
            p.n, p.storage, p.q, p.ldq = self.n, STORAGE_DENSE, Q.ctypes.data_as(C.c_void_p), Q.shape[0]
            self._keep.append(Q)
        else:
            rowptr, colidx, blocks = bsr
            rowptr = np.ascontiguousarray(rowptr, dtype=np.int64); colidx = np.ascontiguousarray(colidx, dtype=np.int32)
            blocks = np.ascontiguousarray(blocks, dtype=np.float64)
            self.n = rowptr.size - 1
            p.n, p.storage, p.nb = self.n, (STORAGE_BSR3_DENSE if densify else STORAGE_BSR3), colidx.size
            p.rowptr, p.colidx, p.blocks = (a.ctypes.data_as(C.c_void_p) for a in (rowptr, colidx, blocks))
            self._keep += [rowptr, colidx, blocks]
        self.h = C.c_void_p()
        _chk(lib().xm_ctx_create(C.byref(p), C.byref(self.h)))
        self._keep = []   # Q has been copied to the device

    TRANSPORTS = {0: "none (one GPU)", 1: "RCCL all-gather", 2: "shared-memory test transport", 3: "direct peer writes (threads of one process)",
                  4: "direct peer writes (one process per GPU, IPC-mapped buffers)"}

    def transport(self):
        """(kind, name, note): the transport that joins the ranks of this context and why a faster one was given up (xm_ctx_transport)"""
        k = C.c_int(0); buf = C.create_string_buffer(512)
        _chk(lib().xm_ctx_transport(self.h, C.byref(k), buf, 512))
        return k.value, self.TRANSPORTS.get(k.value, "?"), buf.value.decode()

    def product_kind(self, o=3):
        """name of the kernel family that serves a tCG product of rank o (xm_ctx_product_kind)"""
        k = C.c_int(0)
        _chk(lib().xm_ctx_product_kind(self.h, int(o), C.byref(k)))
        return PRODUCT_KINDS.get(k.value, "?")

    def schur_info(self):
        """matrix-free contexts: dict(cg=bool, products, inner_iters, capped, last_relres) (xm_ctx_schur_info)"""
        u = C.c_int(0); st = (C.c_int64 * 3)(); rr = C.c_double(0.0)
        _chk(lib().xm_ctx_schur_info(self.h, C.byref(u), st, C.byref(rr)))
        return dict(cg=bool(u.value), products=int(st[0]), inner_iters=int(st[1]), capped=int(st[2]), last_relres=rr.value)

    def sell_wpad(self):
        """True when the tCG of the last solved rank read its product input at the 128-byte record pitch (xm_ctx_sell_wpad)"""
        on = C.c_int(0)
        _chk(lib().xm_ctx_sell_wpad(self.h, C.byref(on)))
        return bool(on.value)

    def qw(self, W, alpha=1.0):
        """alpha * Q @ W through the context's storage (xm_ctx_qw)"""
        W = np.asfortranarray(np.asarray(W, dtype=np.float64))
        out = np.zeros_like(W, order="F")
        _chk(lib().xm_ctx_qw(self.h, W.shape[1], W.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), alpha))
        return np.ascontiguousarray(out)

    # ---- XM^2 re-weighting on the resident Q (reference loop 3_test_colmap_glomap.py:299-351)
    def attach_edges(self, ei, ej, M):
        ei = np.ascontiguousarray(ei, dtype=np.int32); ej = np.ascontiguousarray(ej, dtype=np.int32)
        M = np.ascontiguousarray(M, dtype=np.float64)
        self.ne = ei.size
        _chk(lib().xm_ctx_attach_edges(self.h, self.ne, ei.ctypes.data_as(C.c_void_p), ej.ctypes.data_as(C.c_void_p), M.ctypes.data_as(C.c_void_p)))

    def edge_residuals(self):
        res = np.zeros(self.ne)
        _chk(lib().xm_ctx_edge_residuals(self.h, res.ctypes.data_as(C.c_void_p)))
        return res

    def recover_tp(self, rot, scale):
        """translations (3 x n, camera 1 at the origin) and landmarks (3 x m) of a solution given as anchored rotations (3 x 3n) and
        scales (n) — recover_XM's t_est / p_est without Abar (matrix-free contexts only)"""
        rot = np.asfortranarray(np.asarray(rot, dtype=np.float64)); scale = np.ascontiguousarray(np.asarray(scale, dtype=np.float64).reshape(-1))
        assert rot.shape == (3, 3 * self.n) and scale.size == self.n
        t = np.zeros((3, self.n), order="F"); p = np.zeros((3, self.n_landmarks), order="F")
        _chk(lib().xm_ctx_recover_tp(self.h, rot.ctypes.data_as(C.c_void_p), scale.ctypes.data_as(C.c_void_p),
                                     t.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p)))
        return np.ascontiguousarray(t), np.ascontiguousarray(p)

    def edge_residuals_recovered(self, rot, scale):
        """squared distance per edge / observation of a RECOVERED solution (rot 3 x 3n, scale n): the reference's XM^2 residual"""
        rot = np.asfortranarray(np.asarray(rot, dtype=np.float64)); scale = np.ascontiguousarray(np.asarray(scale, dtype=np.float64).reshape(-1))
        res = np.zeros(self.ne)
        _chk(lib().xm_ctx_edge_residuals_recovered(self.h, rot.ctypes.data_as(C.c_void_p), scale.ctypes.data_as(C.c_void_p), res.ctypes.data_as(C.c_void_p)))
        return res

    def xm2_filter(self, rot, scale, percentile=90.0):
        """the reference's percentile filter on the device -> (threshold, newly removed, new weights); Q is rebuilt"""
        rot = np.asfortranarray(np.asarray(rot, dtype=np.float64)); scale = np.ascontiguousarray(np.asarray(scale, dtype=np.float64).reshape(-1))
        thr = C.c_double(); rm = C.c_int64(); w = np.zeros(self.ne)
        _chk(lib().xm_ctx_xm2_filter(self.h, rot.ctypes.data_as(C.c_void_p), scale.ctypes.data_as(C.c_void_p), percentile, C.byref(thr), C.byref(rm),
                                     w.ctypes.data_as(C.c_void_p)))
        return thr.value, rm.value, w

    def xm2_round(self, R, s, max_rank, tol, max_time=1000.0, percentile=90.0, flags=0):
        """one round of the reference's XM^2 loop starting from the solution (R, s): filter, solve_rank3 at lam 0, lam decision, final solve"""
        n = self.n
        R = np.asfortranarray(np.asarray(R, dtype=np.float64)); s = np.ascontiguousarray(np.asarray(s, dtype=np.float64).reshape(-1))
        rmax = max(int(max_rank), 3)
        Ro = np.zeros((3 * n, rmax + 1), order="F"); so = np.zeros(n)
        opt = Options(); res = Result(); inf = Xm2Info()
        opt.struct_size, res.struct_size, inf.struct_size = C.sizeof(Options), C.sizeof(Result), C.sizeof(Xm2Info)
        opt.max_rank, opt.tol, opt.max_time, opt.flags = int(max_rank), tol, max_time, flags
        inf.percentile = percentile
        res.R = Ro.ctypes.data_as(C.c_void_p); res.s = so.ctypes.data_as(C.c_void_p)
        _chk(lib().xm_ctx_xm2_round(self.h, R.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p), R.shape[1], C.byref(opt), C.byref(inf), C.byref(res)))
        info = {k: getattr(res, k) for k, _ in Result._fields_ if k not in ("R", "s", "struct_size")}
        x2 = {k: getattr(inf, k) for k, _ in Xm2Info._fields_ if k != "struct_size"}
        return np.ascontiguousarray(Ro[:, : res.rank]), so, info, x2

    def set_edge_weights(self, w):
        w = np.ascontiguousarray(w, dtype=np.float64)
        assert w.size == self.ne
        _chk(lib().xm_ctx_set_edge_weights(self.h, w.ctypes.data_as(C.c_void_p)))

    def solve(self, max_rank, tol, lam, max_time=1000.0, mode=MODE_SOLVE, flags=0, s_ini=None, trace=0, R_ini=None, retraction=RETRACT_QR,
              grouping=0):
        n = self.n
        rmax = max(int(max_rank), 3)
        R = np.zeros((3 * n, rmax + 1), order="F"); s = np.zeros(n)
        opt = Options(); res = Result()
        opt.struct_size, res.struct_size = C.sizeof(Options), C.sizeof(Result)
        opt.retraction, opt.sum_grouping = int(retraction), int(grouping)
        opt.max_rank, opt.tol, opt.lam, opt.max_time, opt.mode, opt.flags = int(max_rank), tol, lam, max_time, mode, flags
        si = None
        if s_ini is not None:
            si = np.ascontiguousarray(s_ini, dtype=np.float64).reshape(-1)
            opt.s_ini = si.ctypes.data_as(C.c_void_p)
        ri = None
        if R_ini is not None:
            ri = np.asfortranarray(np.asarray(R_ini, dtype=np.float64)[:, :3])
            opt.R_ini = ri.ctypes.data_as(C.c_void_p); opt.flags |= FLAG_WARM_R
        tr = None
        if trace:
            tr = np.zeros((trace, 6)); opt.trace_cap = trace; opt.trace = tr.ctypes.data_as(C.c_void_p)
        res.R = R.ctypes.data_as(C.c_void_p); res.s = s.ctypes.data_as(C.c_void_p)
        _chk(lib().xm_ctx_solve(self.h, C.byref(opt), C.byref(res)))
        info = {k: getattr(res, k) for k, _ in Result._fields_ if k not in ("R", "s", "struct_size")}
        if tr is not None:
            info["trace"] = tr[: res.trace_len].copy()
        return np.ascontiguousarray(R[:, : res.rank]), s, info

    def close(self):
        if self.h:
            lib().xm_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def solve_dense(Q, max_rank, tol, lam, tuning=None, **kw):
    ctx = Context(Q=Q, tuning=tuning)
    try:
        return ctx.solve(max_rank, tol, lam, **kw)
    finally:
        ctx.close()


def import_XM():
    """import the reference-named extension module `XM` (xm-code_amd/build/XM*.so)"""
    import importlib
    import sys
    if MODULE_DIR not in sys.path:
        sys.path.insert(0, MODULE_DIR)
    return importlib.import_module("XM")
