"""copies what scripts/gpu_r4_final.sh left in gpurun_out/ into profiles/ (tracked): python scripts/collect_profiles_r4.py"""
import glob, os, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(os.path.join(R, "profiles"), exist_ok=True)
n = 0
for f in sorted(glob.glob(os.path.join(R, "gpurun_out", "r04_*"))):
    if os.path.isfile(f) and os.path.getsize(f) > 0:
        shutil.copy(f, os.path.join(R, "profiles", os.path.basename(f))); n += 1
print(n, "files copied")
