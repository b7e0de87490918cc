"""copy what scripts/gpu_evidence.sh <tag> left under gpurun_out/ into profiles/ (tracked) and write profiles/<tag>_SUMMARY.md, the one-page
reading of the run:   python scripts/collect_profiles.py <tag>"""
import glob, json, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
P = os.path.join(R, "profiles")
n = 0
for f in sorted(glob.glob(os.path.join(R, "gpurun_out", tag + "_*"))):
    base = os.path.basename(f)[len(tag) + 1:].rsplit(".", 1)[0]
    if base.rsplit("_", 1)[-1] in ("a", "b", "c", "d") or "_traced" in base or base in ("load_time_plain", "rome_trace", "compress_probe"):
        continue          # numbered probe outputs of the development calls; what they showed is in HISTORY.md
    if os.path.isfile(f) and os.path.getsize(f) > 0 and not f.endswith(".err"):
        shutil.copy(f, os.path.join(P, os.path.basename(f))); n += 1
print(n, "files copied into profiles/")


def load(name):
    try:
        with open(os.path.join(P, f"{tag}_{name}")) as fh:
            txt = fh.read().strip()
        return json.loads(txt.splitlines()[-1]) if txt else None
    except (OSError, ValueError):
        return None


def g(d, *ks, fmt="{}"):
    for k in ks:
        if not isinstance(d, dict) or k not in d or d[k] is None:
            return "—"
        d = d[k]
    return fmt.format(d) if not isinstance(d, (dict, list)) else json.dumps(d)


out = [f"# {tag}: the evidence run in one page", "",
       f"Produced by `scripts/collect_profiles.py {tag}` from the files `scripts/gpu_evidence.sh {tag}` wrote on one MI355X box "
       "(every number below is in the named file; nothing is typed in by hand).", ""]
for name in ("pytest_gpu.txt", "smoke.txt", "pytest_multi_x3.txt"):
    fp = os.path.join(P, f"{tag}_{name}")
    if os.path.exists(fp):
        out += [f"`{tag}_{name}`:", "```", open(fp).read().strip()[-600:], "```", ""]
out += ["## bench lines", "",
        "| file | workload | value (tCG it/s) | ms per solve | kernel | HIP-event µs / frac | traced µs / frac | counter traffic ÷ algorithmic | transport |",
        "|---|---|---|---|---|---|---|---|---|"]
for name in ("bench_venice1778.json", "bench_vg100k_vg.json", "bench_vg100k_bsr.json", "bench_rome_bsr.json", "bench_2gpu_virtual.json",
             "bench_8gpu_virtual.json"):
    b = load(name)
    if not b:
        continue
    r = b.get("roofline") or {}
    alg = r.get("algorithmic_bytes_per_launch")
    tr = r.get("traffic")
    ratio = f"{tr / alg:.3f}" if (alg and tr) else "—"
    ev = f"{r['avg_launch_ms'] * 1e3:.1f}" if r.get("avg_launch_ms") else "—"
    tus = r.get("traced_avg_launch_us")
    tu = f"{tus:.1f}" if tus else "—"
    tf = f"{alg / tus / 1e3 / r.get('peak', 8000.0):.3f}" if (alg and tus) else "—"
    out.append(f"| `{tag}_{name}` | {g(b, 'config', 'workload')} ×{b.get('n_gpus')} | {g(b, 'value', fmt='{:.0f}')} | {g(b, 'ms_per_step', fmt='{:.1f}')} | "
               f"{g(r, 'kernel')[:70]} | {ev} / {g(r, 'frac', fmt='{:.3f}')} | {tu} / {tf} | {ratio} | {g(b, 'transport')} |")
b = load("bench_venice1778.json")
if b:
    out += ["", "## legs of the headline line (`%s_bench_venice1778.json`)" % tag, ""]
    for k in ("roofline_hbm", "rome_scale", "rome_scale_dense", "kkt_pair", "cpu_baseline", "exchange", "rccl_leg"):
        if k in b:
            out += [f"* `{k}`: `{json.dumps(b[k])[:900]}`"]
b2 = load("bench_2gpu_virtual.json")
if b2:
    out += ["", "## two virtual ranks on one GPU (`%s_bench_2gpu_virtual.json`)" % tag, ""]
    for k in ("transport", "fallback", "exchange", "rccl_leg"):
        if k in b2:
            out += [f"* `{k}`: `{json.dumps(b2[k])[:600]}`"]
if tag.startswith("multi_"):
    # scripts/first_multi_gpu_run.sh: the first run on real devices -- rung by rung, then the scaling legs of the same workload
    out += ["", "## first multi-GPU run (`scripts/first_multi_gpu_run.sh`)", ""]
    fp = os.path.join(P, f"{tag}_probe.log")
    if os.path.exists(fp):
        out += ["transport ladder, rung by rung (`%s_probe.log`):" % tag, "```"] + [l[:400] for l in open(fp).read().splitlines() if l.startswith("{")] + ["```", ""]
    rows = []
    for f in sorted(glob.glob(os.path.join(P, f"{tag}_bench_*.json")) + glob.glob(os.path.join(P, f"{tag}_scale_*.json"))):
        try:
            b = json.loads(open(f).read().strip().splitlines()[-1])
        except (ValueError, IndexError):
            continue
        rl = b.get("rccl_leg") or {}
        rows.append(f"| `{os.path.basename(f)}` | {b.get('n_gpus')} | {g(b, 'value', fmt='{:.0f}')} | {g(b, 'ms_per_step', fmt='{:.1f}')} | {g(b, 'transport')} | {g(b, 'fallback')} | "
                    f"{g(b, 'exchange', 'used')} | {g(rl, 'value', fmt='{:.0f}')} | {g(b, 'rome_scale', 'value', fmt='{:.0f}')} | {g(b, 'rome_scale_dense', 'value', fmt='{:.0f}')} | {g(b, 'error')} |")
    if rows:
        out += ["| file | GPUs | Venice tCG it/s | ms per solve | transport | fallback | exchange used | RCCL leg it/s | Final-13682 BSR it/s | Final-13682 dense it/s | error |",
                "|---|---|---|---|---|---|---|---|---|---|---|"] + rows
out += ["", "## stamped PMC legs (`%s_pmc_fetch_<leg>.json`; FETCH_SIZE x 1024 x 2 per MI355X_MICROARCH)" % tag, "",
        "| leg | what | real launches | counter traffic per launch (MB) | traced µs per launch |", "|---|---|---|---|---|"]
for f in sorted(glob.glob(os.path.join(P, f"{tag}_pmc_fetch_*.json"))):
    try:
        d = json.load(open(f))
    except ValueError:
        continue
    pp = d.get("per_product") or {}
    h = d.get("hess") or {}
    by, us = pp.get("hbm_side_bytes"), pp.get("traced_us")
    out.append(f"| {d.get('leg')} | {d.get('what')} | {h.get('real_launches', '—')} | {by / 1e6:.1f} | {us:.1f} |" if (by and us)
               else f"| {d.get('leg')} | `{json.dumps(d)[:300]}` | | | |")
out += ["", "Kernel statistics (`rocprofv3 --kernel-trace --stats`) of the three solves: `%s_kernel_stats_bench_*.csv`; per-kernel averages, busy "
        "fraction and idle time in front of each kernel: `%s_trace_summary_*.txt`; micro-benchmarks: `%s_kbench*.txt`." % (tag, tag, tag), ""]
with open(os.path.join(P, f"{tag}_SUMMARY.md"), "w") as fh:
    fh.write("\n".join(out))
print("wrote", f"profiles/{tag}_SUMMARY.md")
