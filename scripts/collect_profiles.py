"""copy what scripts/gpu_evidence.sh <tag> left under gpurun_out/ into profiles/ (tracked): python scripts/collect_profiles.py <tag>"""
import glob, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
n = 0
for f in sorted(glob.glob(os.path.join(R, "gpurun_out", tag + "_*"))):
    if os.path.isfile(f) and os.path.getsize(f) > 0:
        shutil.copy(f, os.path.join(R, "profiles", os.path.basename(f))); n += 1
print(n, "files copied into profiles/")
