"""Per-kernel register / LDS / scratch figures of one HIP source, from hipcc's own resource remarks (no GPU needed):
   python scripts/kernel_resources.py xm-code_amd/csrc/xm_sell.hip [name filter]"""
import re, subprocess, sys
src = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = None; rows = []
for line in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|TotalSGPRs|SGPRs Spill|VGPRs Spill): (\S+)", line)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}; rows.append(cur)
    elif cur is not None: cur[k] = v
for r in rows:
    if filt in r["name"]:
        print(f'{r["name"][:110]:110s} vgpr {r.get("VGPRs"):>4} sgpr {r.get("TotalSGPRs"):>4} scratch {r.get("ScratchSize [bytes/lane]"):>5} occ {r.get("Occupancy [waves/SIMD]"):>2} lds {r.get("LDS Size [bytes/block]"):>6}')
