"""does the half-traffic symmetric product run at lower clocks than the general one on this box?  long runs of each, rocm-smi sampled meanwhile"""
import sys, os, time, subprocess, threading, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xm-code_amd"))
import numpy as np, xmamd
n, o = int(sys.argv[1]), 3
L = xmamd.lib(); ld = xmamd.dense_ld(n)
dq = xmamd.DevArray(nbytes=3 * n * ld * 8)
rng = np.random.default_rng(0)
chunk = rng.standard_normal(min(3 * n * ld, 1 << 24)); off = 0
while off < 3 * n * ld:
    m = min(chunk.size, 3 * n * ld - off)
    xmamd._chk(L.xm_dev_h2d(C.c_void_p(dq.ptr.value + off * 8), chunk.ctypes.data_as(C.c_void_p), m * 8)); off += m
dW = xmamd.DevArray(rng.standard_normal((ld, 3))); dO = xmamd.DevArray(nbytes=3 * n * 3 * 8)
stop = False; samples = []
def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--csv"], capture_output=True, text=True, timeout=5).stdout
            samples.append((time.time(), out.strip().splitlines()[-1]))
        except Exception as e:
            samples.append((time.time(), repr(e)))
        time.sleep(0.05)
for name, fn, reps in (("general", L.xm_qw_dense_time, 1200), ("symmetric", L.xm_qw_dense_sym_time, 2000), ("general again", L.xm_qw_dense_time, 600)):
    samples.clear(); stop = False
    th = threading.Thread(target=sampler); th.start()
    ms = C.c_double(); t0 = time.time()
    xmamd._chk(fn(dq.ptr, n, o, dW.ptr, dO.ptr, reps, C.byref(ms)))
    stop = True; th.join()
    print(f"{name}: {ms.value*1e3:.1f} us per product over {reps} launches ({time.time()-t0:.1f} s), {len(samples)} samples")
    for t, s_ in samples[1::max(1, len(samples) // 6)]:
        print("   ", s_[:200])
hdr = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--csv"], capture_output=True, text=True).stdout.strip().splitlines()
print("header:", hdr[0][:300] if hdr else None)
