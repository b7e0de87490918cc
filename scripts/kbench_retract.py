"""thread-per-camera against quad-per-camera MGS-QR retraction (xm_kernels.hip: retract_kernel / retract_quad_kernel): python scripts/kbench_retract.py [n ...]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xm-code_amd"))
import numpy as np, xmamd
L = xmamd.lib()
for n in [int(x) for x in sys.argv[1:]] or [1778, 13682, 100000]:
    for o in (3, 5):
        OP = o | 1
        rng = np.random.default_rng(n)
        R = rng.standard_normal((3 * n, o)); D = 0.1 * rng.standard_normal((3 * n, o)); s = 1 + 0.1 * rng.random(n); ds = 0.05 * rng.standard_normal(n)
        dR = xmamd.DevArray(xmamd.to_rm(R)); dD = xmamd.DevArray(xmamd.to_rm(D)); dsv = xmamd.DevArray(s); dds = xmamd.DevArray(ds)
        outs = {}
        line = f"retract n={n} o={o}:"
        for v, name in ((0, "thread per camera"), (2, "quad per camera")):
            dRo = xmamd.DevArray(nbytes=dR.nbytes); dso = xmamd.DevArray(nbytes=dsv.nbytes)
            ms = C.c_double()
            xmamd._chk(L.xm_retract_variant(n, o, dR.ptr, dsv.ptr, dD.ptr, dds.ptr, 0.7, dRo.ptr, dso.ptr, v, 200, C.byref(ms)))
            outs[v] = xmamd.from_rm(dRo.get(), 3 * n, o)
            line += f"  {name} {ms.value*1e3:7.2f} us"
            dRo.free(); dso.free()
        line += f"   max |difference| {np.abs(outs[0] - outs[2]).max():.1e}"
        print(line, flush=True)
        for b in (dR, dD, dsv, dds): b.free()
