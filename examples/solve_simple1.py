#!/usr/bin/env python3
"""The reference's minimal driver (1_test_solve.py:42), against this build's `XM` module:

    XM.solve(dataset_path, 3, 1e-16, 0.0, 1000)   # reads <path>/Q.bin, writes <path>/R.bin and <path>/s.bin

The dataset is the reference's own assets/SIMPLE1/Q.bin (kept as a test fixture under tests/golden/simple1).
Needs an MI355X: there is no CPU fallback.
"""
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xm-code_amd", "build"))      # the reference appends XM/build/ here
sys.path.insert(0, os.path.join(ROOT, "xm-code_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import XM                     # noqa: E402  (reference module name, same three functions)
import numpy as np            # noqa: E402
import xm_testlib as tl       # noqa: E402
import xmamd                  # noqa: E402

with tempfile.TemporaryDirectory() as d:
    shutil.copy(os.path.join(ROOT, "tests", "golden", "simple1", "Q.bin"), d)
    XM.solve(d + "/", 3, 1e-16, 0.0, 1000)
    R = tl.load_bin(os.path.join(d, "R.bin"))
    s = tl.load_bin(os.path.join(d, "s.bin"))
print("R", R.shape, "s", s.shape, "scale range", float(s.min()), float(s.max()))
rot, scale, nneg = xmamd.recover_rotations(R, s)              # anchored rotations on the GPU
gold = np.load(os.path.join(ROOT, "tests", "golden", "simple1", "rot_anchor.npy"))
print("anchored rotations vs golden (reference recover_XM on the CPU oracle's solution): rel. Frobenius",
      tl.rel_fro(rot, gold))
