#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/.

Runs ONLY in the build container: it needs the read-only reference checkout at /root/reference
(never present on the GPU box).  What it does:

  simple1/Q.bin        byte copy of the reference's data file assets/SIMPLE1/Q.bin (input of 1_test_solve.py:42)
  simple2/Q.bin        produced by running the reference's own 2_test_creatematrix.py pipeline (clean-up +
                       utils/creatematrix.py:create_matrix) on assets/SIMPLE2/landmark.bin, unchanged, with
                       `XM` (the unbuildable CUDA extension) replaced by the CPU oracle and the Open3D viewer
                       replaced by no-ops
  simple2/gtR.bin      byte copy of assets/SIMPLE2/gtR.bin; simple2/frame_index.npy = camera index -> original
                       frame (inverse of the pipeline's `indices_frame`, 2_test_creatematrix.py:84-91)
  */expected.json      optimum value / certificate / iteration counts of the oracle at tol=1e-16
  */rot_anchor.npy     3 x 3n anchored SO(3) rotations = reference utils/recoversolution.py:recover_XM applied
                       to the oracle's (R, s)   (defines the parity metric of SURVEY.md §8c)
  */sR_gram_sample.npy a fixed sample of entries of X = sR sR^T (gauge-invariant)
  synth/*              small seeded synthetic problems (generators in tests/xm_testlib.py) incl. a staircase case

Only DATA is stored (inputs + expected outputs); no reference source text is copied.
"""
import json
import os
import runpy
import shutil
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import xm_oracle as xo  # noqa: E402
import xm_testlib as tl  # noqa: E402


def run_reference_pipeline_simple2(workdir):
    """Execute the reference's 2_test_creatematrix.py verbatim (from its own path) in a scratch cwd."""
    ds = os.path.join(workdir, "assets", "SIMPLE2")
    os.makedirs(ds)
    for f in ("landmark.bin", "gtR.bin", "gtt.bin", "gtp.bin"):
        shutil.copy(os.path.join(REF, "assets", "SIMPLE2", f), ds)
    fake_xm = types.ModuleType("XM")
    fake_xm.solve = lambda path, max_rank, tol, lam, max_time: xo.solve_path(path, max_rank, tol, lam, max_time, 0)
    fake_vis = types.ModuleType("utils.visualization")
    fake_vis.visualize_camera = lambda *a, **k: None
    fake_vis.visualize = lambda *a, **k: None
    sys.modules["XM"] = fake_xm
    sys.modules["utils.visualization"] = fake_vis
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(workdir)
    try:
        g = runpy.run_path(os.path.join(REF, "2_test_creatematrix.py"), run_name="__main__")
    finally:
        os.chdir(cwd)
    return ds, g


def reference_recover(Q, R, s, lam):
    """reference utils/recoversolution.py:recover_XM (imported, not copied); Abar is irrelevant for rotations."""
    sys.path.insert(0, REF)
    from utils.recoversolution import recover_XM
    n = s.shape[0]
    Abar = np.zeros((1, 3 * n))
    R_real, s_real, _, _ = recover_XM(Q, R, s.reshape(-1, 1), Abar, lam)
    return R_real, s_real


def expected_from_oracle(Q, max_rank, tol, lam, out_dir, flags=0):
    R, s, info = xo.solve(Q, max_rank, tol, lam, 1000.0, flags=flags, trace=1000)
    rot, s_real = reference_recover(Q, R, s, lam)
    sR = tl.scale_rows(R, s)
    idx = tl.gram_sample_index(Q.shape[0])
    X = (sR[idx[:, 0]] * sR[idx[:, 1]]).sum(axis=1)
    np.save(os.path.join(out_dir, "rot_anchor.npy"), rot)
    np.save(os.path.join(out_dir, "sR_gram_sample.npy"), X)
    tr = info["trace"]
    exp = dict(n=int(Q.shape[0] // 3), max_rank=max_rank, tol=tol, lam=lam, rank=info["rank"], status=info["status"],
               f_star=float(tr[-1, 0]), final_gradnorm=float(tr[-1, 1]), outer_iters=int(info["outer_iters"]),
               tcg_iters=int(info["tcg_iters"]), qw_products=int(info["qw_products"]), stop_reason=int(info["stop_reason"]),
               s_min=float(s.min()), s_max=float(s.max()), cert=info["cert"],
               trace_head=[[float(x) for x in row] for row in tr[:6]])
    with open(os.path.join(out_dir, "expected.json"), "w") as f:
        json.dump(exp, f, indent=1)
    return R, s, info, rot


def main():
    xo.build()
    # ---------------- SIMPLE1 ----------------
    d1 = os.path.join(HERE, "simple1")
    os.makedirs(d1, exist_ok=True)
    shutil.copy(os.path.join(REF, "assets", "SIMPLE1", "Q.bin"), os.path.join(d1, "Q.bin"))
    Q1 = tl.load_bin(os.path.join(d1, "Q.bin"))
    expected_from_oracle(Q1, 3, 1e-16, 0.0, d1)          # the call of 1_test_solve.py:42
    # ---------------- SIMPLE2 ----------------
    d2 = os.path.join(HERE, "simple2")
    os.makedirs(d2, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        ds, g = run_reference_pipeline_simple2(tmp)
        shutil.copy(os.path.join(ds, "Q.bin"), os.path.join(d2, "Q.bin"))
        shutil.copy(os.path.join(REF, "assets", "SIMPLE2", "gtR.bin"), os.path.join(d2, "gtR.bin"))
        # `indices_frame` maps original frame -> solver camera index (incl. the best-observed-frame swap,
        # 2_test_creatematrix.py:86-91); store the inverse: camera index -> original (ground-truth) frame
        old_to_new = np.asarray(g["indices_frame"])
        new_to_old = np.full(int(old_to_new.max()) + 1, -1)
        new_to_old[old_to_new[old_to_new >= 0]] = np.nonzero(old_to_new >= 0)[0]
        np.save(os.path.join(d2, "frame_index.npy"), new_to_old)
    Q2 = tl.load_bin(os.path.join(d2, "Q.bin"))
    expected_from_oracle(Q2, 3, 1e-16, 0.0, d2)
    # ---------------- synthetic ----------------
    d3 = os.path.join(HERE, "synth")
    os.makedirs(d3, exist_ok=True)
    cases = {
        # name: (generator kwargs, max_rank, tol, lam)
        "vg40_stair": (dict(kind="vg", n=40, deg=3, sigma=1.5, seed=40), 6, 1e-9, 3.0),
        "vg60_cert": (dict(kind="vg", n=60, deg=6, sigma=0.2, seed=60), 5, 1e-12, 6.0),
        "dense49": (dict(kind="dense", n=49, seed=49), 5, 1e-12, 0.0),
    }
    meta = {}
    for name, (kw, mr, tol, lam) in cases.items():
        dd = os.path.join(d3, name)
        os.makedirs(dd, exist_ok=True)
        Q = tl.make_problem(**kw)["Q"]
        tl.save_bin(os.path.join(dd, "Q.bin"), Q)
        expected_from_oracle(Q, mr, tol, lam, dd)
        meta[name] = dict(gen=kw, max_rank=mr, tol=tol, lam=lam)
    with open(os.path.join(d3, "cases.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("golden fixtures written under", HERE)


if __name__ == "__main__":
    main()
