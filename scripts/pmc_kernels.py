#!/usr/bin/env python3
"""Per-kernel table of ANY command: traced duration (rocprofv3 --kernel-trace, own pass) and HBM-side traffic (rocprofv3 --pmc FETCH_SIZE, own
pass; x1024 x2 per MI355X_MICROARCH.md) -- where the time and the bytes of a leg go that has no bench.py line of its own (the matrix-free product).
   python scripts/pmc_kernels.py <name> [--min-share 0.5] -- python scripts/kbench_schur.py 13682 800000 8 --product-only
writes gpurun_out/<name>.txt"""
import collections, csv, glob, os, shutil, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1]
sep = sys.argv.index("--")
opts, cmd = sys.argv[2:sep], sys.argv[sep + 1:]
min_share = float(opts[opts.index("--min-share") + 1]) if "--min-share" in opts else 0.5
env = dict(os.environ, TMPDIR="/tmp")
dur, fetch, tail = collections.defaultdict(list), collections.defaultdict(list), ""
for mode, flags in (("trace", ["--kernel-trace"]), ("pmc", ["--pmc", "FETCH_SIZE"])):
    d = os.path.join(R, "gpurun_out", f"pk_{name}_{mode}")
    shutil.rmtree(d, ignore_errors=True)
    p = subprocess.run(["rocprofv3"] + flags + ["--output-format", "csv", "-d", d, "-o", "run", "--"] + cmd, cwd=R, env=env, capture_output=True, text=True,
                       timeout=1500)
    if mode == "trace":
        tail = "\n".join(l for l in p.stdout.splitlines() if "amdgpu.ids" not in l)[-1500:]
        for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[:1]:
            for r in csv.DictReader(open(f)):
                dur[r["Kernel_Name"]].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    else:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True)[:1]:
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == "FETCH_SIZE":
                    fetch[r["Kernel_Name"]].append(float(r["Counter_Value"]) * 2048.0)
    shutil.rmtree(d, ignore_errors=True)
tot = sum(sum(v) for v in dur.values()) or 1.0
lines = [f"command: {' '.join(cmd)}", "its output (trace pass):", tail, "",
         f"{'kernel':<78} {'launches':>8} {'avg us':>9} {'total ms':>9} {'share %':>8} {'FETCH MB/launch':>16} {'GB/s':>8}"]
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    if 100.0 * sum(v) / tot < min_share:
        continue
    fb = sum(fetch[k]) / len(fetch[k]) if fetch.get(k) else None
    avg = sum(v) / len(v)
    lines.append(f"{k[:78]:<78} {len(v):>8} {avg:>9.1f} {sum(v) / 1e3:>9.2f} {100.0 * sum(v) / tot:>8.1f} "
                 f"{(f'{fb / 1e6:.2f}' if fb is not None else '-'):>16} {(f'{fb / avg / 1e3:.0f}' if fb is not None else '-'):>8}")
lines.append(f"all kernels: {tot / 1e3:.2f} ms in {sum(len(v) for v in dur.values())} launches; FETCH_SIZE x 1024 x 2 (gfx950 correction), counts Infinity-Cache hits too")
open(os.path.join(R, "gpurun_out", name + ".txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
