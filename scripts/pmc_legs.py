#!/usr/bin/env python3
"""HBM-side traffic (rocprofv3 --pmc FETCH_SIZE, own pass) and traced duration (rocprofv3 --kernel-trace, own pass) of the dominant kernels
of bench.py's legs, stamped with the hash of the sources they were measured on -- what bench.py quotes as roofline.traffic / roofline_hbm.traffic.
Run on the GPU box:   python scripts/pmc_legs.py <tag> [leg ...]      legs: venice hbm13682 rome_dense rome_bsr vg100k_vg vg100k_bsr
Writes gpurun_out/<tag>_pmc_fetch_<leg>.json (copy them into profiles/)."""
import collections, csv, glob, json, os, shutil, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
LEGS = {
    # leg: (bench.py arguments, {role: substring(s) of the kernel name -- ALL of a list, or ANY of the alternatives of a list of lists}, what)
    # the headline solve multiplies an exactly symmetric Q of 5334 rows through the half-traffic pair at rank 3 / 4 and through the general kernel
    # at rank 5 (xm_solver.h: sym_rows): per_product = the launch-count-weighted mean of the two kinds of product (MIX below)
    "venice": (["--steps", "1", "--warmup", "0", "--cpu-seconds", "0", "--no-hbm-check", "--no-rome", "--no-kkt-pair"],
               {"hess": ["qw_dense_kernel<", ", 2, 2, "], "main": ["qw_symv_kernel<"], "reduce": ["symv_reduce_kernel<", ", 2>"]},
               "Hessian products of the headline solve (dense products: outer iteration on the host, EPI_HESS = 2): qw_symv_kernel + "
               "symv_reduce_kernel<o, EPI_HESS> at rank 3 / 4 (half-traffic symmetric pair; the main launch also serves the few gradient products), "
               "qw_dense_kernel<5, EPI_HESS> at rank 5"),
    "hbm13682": (["--steps", "1", "--warmup", "0", "--cpu-seconds", "0", "--no-rome"],
                 {"plain": ["qw_dense_kernel<3, 0, 2, false"]}, "roofline_hbm leg: qw_dense_kernel<3, EPI_PLAIN> on the 13.5 GB matrix (220 MB prefix cacheable, the rest a non-temporal stream)"),
    "rome_dense": (["--steps", "1", "--warmup", "0", "--cpu-seconds", "0", "--no-hbm-check"],
                   {"main": ["qw_symv_kernel<3"], "reduce": ["symv_reduce_kernel<3, 2"]},
                   "rome_scale_dense leg: Hessian products of the 13.5 GB dense Q through the half-traffic symmetric path (main + reduce launch)"),
    "rome_bsr": (["--workload", "final13682", "--storage", "bsr", "--steps", "1", "--warmup", "0", "--no-hbm-check", "--cpu-seconds", "0"],
                 {"hess": ["qw_bsr3_kernel<3, 4, 2"]},
                 "tCG products of the Final-13682-size solve in 3x3-block CSR storage (qw_bsr3_kernel<3, EPI_AUTO>: Hessian products + one gradient product per outer iteration; 34 MB: cache-resident, latency regime)"),
    "vg100k_vg": (["--workload", "vg100k", "--storage", "vg", "--steps", "1", "--warmup", "0", "--no-rome", "--no-hbm-check", "--cpu-seconds", "0"],
                  {"main": ["qw_sell_kernel_q_occ4<"], "reduce": ["sell_reduce_kernel<3, 2"]},
                  "Hessian products of the 100k-camera solve, view-graph storage (sliced-ELL main launch + per-camera sum / epilogue launch)"),
    "vg100k_bsr": (["--workload", "vg100k", "--storage", "bsr", "--steps", "1", "--warmup", "0", "--no-rome", "--no-hbm-check", "--cpu-seconds", "0"],
                   {"main": ["qw_sell_kernel<3, "], "reduce": ["sell_reduce_kernel<3, 2"]},
                   "Hessian products of the 100k-camera solve, 3x3-block CSR storage (sliced-ELL copy with full blocks)"),
}

MIX = {"venice": [("general kernel", ["hess"], "hess"), ("symmetric pair", ["main", "reduce"], "reduce")]}


def run(tag, leg):
    bargs, roles, what = LEGS[leg]
    out = {"leg": leg, "what": what, "command": "python bench.py " + " ".join(bargs)}
    env = dict(os.environ, TMPDIR="/tmp")
    for mode, flags in (("pmc", ["--pmc", "FETCH_SIZE"]), ("trace", ["--kernel-trace"])):
        d = os.path.join(R, "gpurun_out", f"pmcleg_{leg}_{mode}")
        shutil.rmtree(d, ignore_errors=True)
        cmd = ["rocprofv3"] + flags + ["--output-format", "csv", "-d", d, "-o", "run", "--", sys.executable, os.path.join(R, "bench.py")] + bargs
        p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=1500)
        out[mode + "_rc"] = p.returncode
        if mode == "pmc":
            f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
            per = collections.defaultdict(list)
            for r in csv.DictReader(open(f[0])) if f else []:
                if r["Counter_Name"] != "FETCH_SIZE":
                    continue
                for role, subs in roles.items():
                    if all(s in r["Kernel_Name"] for s in subs):
                        per[role].append(float(r["Counter_Value"]))
            for role, v in per.items():
                real = [x for x in v if x > 0.25 * max(v)]       # enqueued-ahead no-op launches fetch (almost) nothing
                out[role] = {"launches": len(v), "real_launches": len(real), "FETCH_SIZE_KB_avg_real": sum(real) / len(real),
                             "hbm_side_bytes_per_real_launch": sum(real) / len(real) * 1024 * 2}
        else:
            f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
            per = collections.defaultdict(list)
            for r in csv.DictReader(open(f[0])) if f else []:
                for role, subs in roles.items():
                    if all(s in r["Kernel_Name"] for s in subs):
                        per[role].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
            for role, v in per.items():
                real = [x for x in v if x > 0.25 * max(v)]
                out.setdefault(role, {})["traced_avg_us_real"] = sum(real) / len(real) / 1e3
                out[role]["traced_real_launches"] = len(real)
        shutil.rmtree(d, ignore_errors=True)
    if leg in MIX:
        # alternatives of one product: (roles whose launches add up to it, the role whose real launches count the products of that kind)
        num_b = num_t = den = 0.0
        kinds = {}
        for name, rs, counter in MIX[leg]:
            if not all(r in out and "hbm_side_bytes_per_real_launch" in out[r] and "traced_avg_us_real" in out[r] for r in rs):
                continue          # a kind that did not occur in this run (e.g. no rank-5 stage)
            b = sum(out[r]["hbm_side_bytes_per_real_launch"] for r in rs); t = sum(out[r]["traced_avg_us_real"] for r in rs)
            k = out[counter]["real_launches"]
            kinds[name] = {"products": k, "hbm_side_bytes": b, "traced_us": t}
            num_b += k * b; num_t += k * t; den += k
        out["per_product"] = {"hbm_side_bytes": num_b / den if den else None, "traced_us": num_t / den if den else None, "kinds": kinds}
    else:
        tot = [out[r]["hbm_side_bytes_per_real_launch"] for r in roles if r in out and "hbm_side_bytes_per_real_launch" in out[r]]
        dur = [out[r]["traced_avg_us_real"] for r in roles if r in out and "traced_avg_us_real" in out[r]]
        out["per_product"] = {"hbm_side_bytes": sum(tot) if len(tot) == len(roles) else None, "traced_us": sum(dur) if len(dur) == len(roles) else None}
    import bench
    out["source_sha256"] = bench.source_sha256()   # bench.py quotes this profile only while the sources are the ones it was measured on
    out["correction"] = "x1024 (KB) x2 (gfx950: 128-byte requests tallied at 64 B, MI355X_MICROARCH.md HBM section); FETCH_SIZE counts Infinity-Cache hits too"
    json.dump(out, open(os.path.join(R, "gpurun_out", f"{tag}_pmc_fetch_{leg}.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    tag = sys.argv[1]
    for leg in (sys.argv[2:] or list(LEGS)):
        run(tag, leg)
