#!/usr/bin/env python3
"""tests/golden/simple2/obs.npz: the observation list (camera, landmark, camera-frame point, weight) that the reference's own
2_test_creatematrix.py hands to utils/creatematrix.py:create_matrix for assets/SIMPLE2 — the input of the matrix-free storage
(SURVEY.md 8f N2).  Runs only in the build container (needs /root/reference); the pipeline is executed unchanged through
make_golden.run_reference_pipeline_simple2, only its variables are captured.  Consistency check: Q assembled from these
observations by tests/xm_testlib.py:schur_dense equals the committed simple2/Q.bin (which create_matrix wrote)."""
import os, sys, tempfile
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import make_golden as mg
import xm_testlib as tl

with tempfile.TemporaryDirectory() as wd:
    ds, g = mg.run_reference_pipeline_simple2(wd)
    edges, weights, landmarks = np.asarray(g["edges"]), np.asarray(g["weights"], dtype=np.float64).reshape(-1), np.asarray(g["landmarks"], dtype=np.float64)
    Qref = tl.load_bin(os.path.join(ds, "Q.bin"))
cam = (edges[:, 0] - 1).astype(np.int32); lm = (edges[:, 1] - 1).astype(np.int32)
Q = tl.schur_dense(cam, lm, landmarks, weights)
err = tl.rel_fro(Q, Qref)
print("observations", cam.size, "cameras", cam.max() + 1, "landmarks", lm.max() + 1, "| Q from observations vs create_matrix's Q.bin:", err)
assert err < 1e-9
assert tl.rel_fro(Qref, tl.load_bin(os.path.join(HERE, "simple2", "Q.bin"))) < 1e-12       # and that IS the committed fixture
np.savez_compressed(os.path.join(HERE, "simple2", "obs.npz"), cam=cam, lm=lm, p=landmarks, w=weights)
