"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol that include/xm_amd.h
declares, the reference-named module `XM` exposes the reference's positional-only signatures, and the product
path fails loudly (no CPU fallback) when there is no GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

import xm_testlib as tl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported(xmamd):
    hdr = open(os.path.join(ROOT, "include", "xm_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(xm_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    L = xmamd.lib()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, f"symbols declared in xm_amd.h but not exported: {missing}"
    assert set(xmamd.EXPORTS) <= declared
    assert b"gfx950" in L.xm_version()
    # the timing hooks of the micro-benchmarks live in their own header, which the product header does not pull in
    assert "xm_bench.h" not in hdr and not any(d.endswith("_time") or d.endswith("_bench") for d in declared)
    bh = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "xm_bench.h")).read(), flags=re.S)
    bdecl = set(re.findall(r"\b(xm_[a-z0-9_]+)\s*\(", bh))
    assert bdecl == set(xmamd.BENCH_EXPORTS) and not [s for s in sorted(bdecl) if not hasattr(L, s)]


def test_struct_layout_matches_header(xmamd):
    # POD structs cross the ABI by pointer: sizes must match what a C compiler produces for the header
    import subprocess, tempfile
    src = ('#include "xm_amd.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu %zu %zu\\n",sizeof(xm_problem_t),sizeof(xm_options_t),'
           'sizeof(xm_result_t),sizeof(xm_tuning_t),sizeof(xm_xm2_info_t));return 0;}\n')
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        sizes = tuple(map(int, subprocess.check_output([os.path.join(d, "t")]).split()))
    assert sizes == tuple(ctypes.sizeof(t) for t in (xmamd.Problem, xmamd.Options, xmamd.Result, xmamd.Tuning, xmamd.Xm2Info))
    assert xmamd.lib().xm_abi_revision() == 4


def test_block_balanced_partition(xmamd):
    """xm_partition_blocks: contiguous ranges that cover all cameras with (nearly) equal STORED BLOCKS per rank -- a hub-camera graph
    must not land most of the matrix on one rank"""
    L = xmamd.lib()
    H = tl.gen_vg_hubs(2000, 8, 5, 0.3, 0.1, seed=3)
    rp, ci, bl = tl.vg_from_edges(2000, H["ei"], H["ej"], H["w"], H["M"])
    for world in (1, 2, 3, 8):
        prev, loads, sizes = 0, [], []
        for r in range(world):
            c0, c1 = ctypes.c_int64(), ctypes.c_int64()
            assert L.xm_partition_blocks(2000, rp.ctypes.data_as(ctypes.c_void_p), world, r, ctypes.byref(c0), ctypes.byref(c1)) == 0
            assert c0.value == prev and c1.value >= c0.value
            prev = c1.value
            loads.append(int(rp[c1.value] - rp[c0.value])); sizes.append(c1.value - c0.value)
        assert prev == 2000
        heaviest_row = int(np.diff(rp).max())
        assert max(loads) <= rp[-1] / world + heaviest_row                 # within one row of the ideal share
        eq = [int(rp[min(2000, (r + 1) * -(-2000 // world))] - rp[min(2000, r * -(-2000 // world))]) for r in range(world)]
        assert max(loads) <= max(eq)                                        # never worse than equal camera ranges


def test_view_graph_codec_round_trip(xmamd):
    """host statement of the quaternion codec of the sliced-ELL product (xm_sell_quat_roundtrip): -w * rotation -> 4 doubles -> the
    block the kernel rebuilds, for every branch of the quaternion extraction (rotation angles up to pi about each axis), weights
    over six decades and the removed edge (w = 0)"""
    from scipy.spatial.transform import Rotation as Rt
    rng = np.random.default_rng(1)
    worst = 0.0
    for t in range(3000):
        M = (Rt.from_rotvec(np.pi * np.eye(3)[t % 3] * (1 - 1e-9 * (t // 3))) if t < 60 else Rt.random(random_state=int(rng.integers(1 << 31)))).as_matrix()
        w = 10.0 ** rng.uniform(-3, 3)
        q, r = xmamd.quat_roundtrip(-w * M)
        worst = max(worst, np.abs(r + w * M).max() / w)
        assert abs(q @ q - 2 * w) < 1e-12 * w
    assert worst < 5e-15
    q, r = xmamd.quat_roundtrip(np.zeros((3, 3)))
    assert not q.any() and not r.any()


def test_xm2_reference_fixture(xmamd):
    """tests/golden/simple2/xm2.npz was produced by EXECUTING the reference's XM^2 lines (3_test_colmap_glomap.py:303-323) on its own
    pipeline's variables (make_simple2_xm2.py); the numpy statement used by the GPU tests reproduces it, and so does np.percentile"""
    G = os.path.join(ROOT, "tests", "golden", "simple2")
    x = np.load(os.path.join(G, "xm2.npz")); tp = np.load(os.path.join(G, "tp.npz")); obs = np.load(os.path.join(G, "obs.npz"))
    err = tl.xm2_error_numpy(obs["cam"], obs["lm"], obs["p"], obs["w"], tp["R_real"], tp["s_real"], tp["t_est"], tp["p_est"])
    assert np.abs(err - x["error"]).max() <= 1e-12 * x["error"].max()
    assert float(np.percentile(err, 90)) == pytest.approx(float(x["threshold"]), rel=1e-12)
    assert np.array_equal(np.where(err > float(x["threshold"]))[0], x["removed"]) and x["removed"].size == 6455


def test_partition_is_contiguous_and_covers(xmamd):
    L = xmamd.lib()
    for n, w in [(1, 1), (7, 2), (1778, 8), (13682, 8), (100000, 8), (5, 8)]:
        prev = 0
        for r in range(w):
            c0, c1 = ctypes.c_int64(), ctypes.c_int64()
            assert L.xm_partition(n, w, r, ctypes.byref(c0), ctypes.byref(c1)) == 0
            assert c0.value == prev and c1.value >= c0.value
            prev = c1.value
        assert prev == n
    assert L.xm_partition(5, 0, 0, None, None) != 0


def test_XM_module_surface(xmamd):
    XM = xmamd.import_XM()
    for f in ("solve", "solve_rebuttle", "solve_rank3"):
        assert callable(getattr(XM, f))
        doc = getattr(XM, f).__doc__
        # five positional, unnamed arguments (no py::arg in the reference either, XM_main.cu:405-407)
        assert all(f"arg{i}:" in doc for i in range(5)) and "arg5" not in doc and "arg0: str" in doc
    assert "-> int" in XM.solve_rebuttle.__doc__ and "-> None" in XM.solve.__doc__
    with pytest.raises(TypeError):
        XM.solve("x")                                  # all five arguments are required
    with pytest.raises(TypeError):
        XM.solve("x", -3, 1e-6, 0.0, 10.0)             # unsigned max_rank
    with pytest.raises(TypeError):
        XM.solve(dataset_path="x", max_rank=3, tol=1e-6, lam=0.0, max_time=1.0)
    # additive in-memory surface: argument validation happens before any GPU work
    assert callable(XM.solve_array) and callable(XM.solve_bsr)
    with pytest.raises(ValueError):
        XM.solve_array(np.zeros((4, 4)), 3, 1e-6, 0.0, 1.0)        # not 3n x 3n
    with pytest.raises(ValueError):
        XM.solve_bsr(np.array([0, 2]), np.array([0], dtype=np.int32), np.zeros((1, 3, 3)), 3, 1e-6, 0.0, 1.0)   # rowptr[n] != nb


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="CPU-only behaviour")
def test_no_cpu_fallback(xmamd, tmp_path):
    """Without a GPU every compute entry point must fail loudly instead of computing on the host."""
    assert xmamd.device_count() == 0
    Q = tl.gen_dense(5, seed=1)["Q"]
    tl.save_bin(tmp_path / "Q.bin", Q)
    XM = xmamd.import_XM()
    with pytest.raises(RuntimeError, match="no HIP device"):
        XM.solve(str(tmp_path), 3, 1e-6, 0.0, 10.0)
    assert not (tmp_path / "R.bin").exists()
    with pytest.raises(RuntimeError, match="no HIP device"):
        XM.solve_array(Q, 3, 1e-6, 0.0, 10.0)
    with pytest.raises(xmamd.XmError):
        xmamd.qw_dense(Q, np.ones((15, 3)))
    with pytest.raises(xmamd.XmError):
        xmamd.solve_dense(Q, 3, 1e-6, 0.0)
    assert xmamd.lib().xm_solve(b"/nonexistent", 3, 1e-6, 0.0, 1.0) != 0


def test_layout_helpers(xmamd):
    M = np.arange(18.0).reshape(6, 3)
    rm = xmamd.to_rm(M, rows=8)
    assert rm.shape == (8, 3) and np.array_equal(xmamd.from_rm(rm, 6, 3), M)
    M4 = np.arange(24.0).reshape(6, 4)
    rm = xmamd.to_rm(M4)
    assert rm.shape == (6, 5) and np.all(rm[:, 4] == 0) and np.array_equal(xmamd.from_rm(rm, 6, 4), M4)
    assert xmamd.dense_ld(149) == 512 and xmamd.dense_ld(1778) == 5376


def test_no_kernel_is_selected_through_the_environment():
    """xm_tuning_t is the only way to select a kernel or a layout: the sources read ten environment variables, all of them deployment
    switches of the file surface (which has no tuning argument) or of the process-level communicator set-up -- the list in
    include/xm_amd.h and INTEGRATION.md"""
    import glob
    allowed = {"XM_QUIET", "XM_GPUS", "XM_GPU_MAP", "XM_RETRACTION", "XM_WATCHDOG_S", "XM_COMM_PEER", "XM_COMM_TRACE", "XM_FORCE_COMM",
               "XM_SHM_TIMEOUT", "XM_SHM_ASYNC"}
    read = set()
    for f in glob.glob(os.path.join(ROOT, "xm-code_amd", "csrc", "*")):
        read |= set(re.findall(r'getenv\("(XM_[A-Z0-9_]+)"\)', open(f).read()))
    assert read <= allowed, sorted(read - allowed)
    hdr = open(os.path.join(ROOT, "include", "xm_amd.h")).read()
    assert all(v in hdr for v in read)
