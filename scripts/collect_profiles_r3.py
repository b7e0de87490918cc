"""gpurun_out/ (scratch, merged back from the GPU box by scripts/gpu_r3_final.sh) -> profiles/r03_* (tracked)."""
import csv, glob, json, os, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P, tag = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles"), "r03"


def last_json(path):
    for line in reversed(open(path).read().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit("no JSON line in " + path)


b = last_json(os.path.join(G, "bench.log"))
json.dump(b, open(os.path.join(P, f"{tag}_bench_venice1778.json"), "w"), indent=1)
for src, dst in (("bench_vg100k_vg.log", "bench_vg100k_vg"), ("bench_vg100k_bsr.log", "bench_vg100k_bsr"), ("bench_2gpu_virtual.log", "bench_2gpu_virtual")):
    if os.path.exists(os.path.join(G, src)):
        json.dump(last_json(os.path.join(G, src)), open(os.path.join(P, f"{tag}_{dst}.json"), "w"), indent=1)
if os.path.exists(os.path.join(G, "kbench.log")):
    shutil.copy(os.path.join(G, "kbench.log"), os.path.join(P, f"{tag}_kbench.txt"))
if os.path.exists(os.path.join(G, "kbench_multi.log")):   # gpu_r3_final2.sh: strips with every split factor + peer all-gather, final sources
    shutil.copy(os.path.join(G, "kbench_multi.log"), os.path.join(P, f"{tag}_kbench_multi.txt"))
shutil.copy(os.path.join(G, "pytest_gpu.log"), os.path.join(P, f"{tag}_pytest_gpu.txt"))
stats = glob.glob(os.path.join(G, "prof_final", "**", "*kernel_stats.csv"), recursive=True)[0]
shutil.copy(stats, os.path.join(P, f"{tag}_kernel_stats_bench_venice1778.csv"))
trace = glob.glob(os.path.join(G, "prof_final", "**", "*kernel_trace.csv"), recursive=True)[0]
per, cg = {}, []
for r in csv.DictReader(open(trace)):
    name, d = r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "qw_dense_kernel<" in name and ", 2, 2," in name:
        per.setdefault(name.split("(")[0], []).append(d)
    elif "cg_step_kernel" in name:
        cg.append(d)
out, allreal = {}, []
for k, v in per.items():
    real = [x for x in v if x > 10.0]
    allreal += real
    out[k] = {"launches": len(v), "noop_launches": len(v) - len(real), "real_launches": len(real),
              "avg_real_us": sum(real) / max(1, len(real)), "avg_all_us": sum(v) / len(v)}
out["all_hess_real_avg_us"] = sum(allreal) / max(1, len(allreal))
cgr = [x for x in cg if x > 3.5]
out["cg_step_avg_us"] = sum(cg) / max(1, len(cg)); out["cg_step_real_avg_us"] = sum(cgr) / max(1, len(cgr))
out["bench_line_hip_event_avg_us"] = b["roofline"]["avg_launch_ms"] * 1e3
json.dump(out, open(os.path.join(P, f"{tag}_kernel_trace_hess_real_vs_noop.json"), "w"), indent=1)
if os.path.exists(os.path.join(G, f"{tag}_pmc_fetch_hess_bench.json")):
    shutil.copy(os.path.join(G, f"{tag}_pmc_fetch_hess_bench.json"), os.path.join(P, f"{tag}_pmc_fetch_hess_bench.json"))
vg = glob.glob(os.path.join(G, "prof_vg100k", "**", "*kernel_stats.csv"), recursive=True)
if vg:
    shutil.copy(vg[0], os.path.join(P, f"{tag}_kernel_stats_bench_vg100k_vg.csv"))
print(json.dumps(out, indent=1))
print("bench:", b["ms_per_step"], b["value"], b["roofline"]["frac"], b["roofline"].get("traffic"), b.get("roofline_hbm", {}).get("frac"), b.get("cpu_baseline", {}).get("value"))
