#!/bin/bash
# round 5, second GPU call: the suite with the CG form of the matrix-free storage, the speculative start of the next truncated CG and the hoisted
# loads; the Final-13682 block-sparse iteration under the trace again; bench lines; matrix-free micro-benchmark (dense inverse vs CG form)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out
( timeout 1800 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -60 ) > $O/r05_pytest_gpu_b.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/r05_smoke_b.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/rome_trace -o run -- python $R/bench.py --workload final13682 --storage bsr --steps 3 --warmup 1 --no-hbm-check --cpu-seconds 0 > $O/r05_bench_rome_bsr_traced_b.json 2> $O/r05_rome_trace_b.err
cd $R
f=$(ls $O/rome_trace/*/*kernel_trace.csv $O/rome_trace/*kernel_trace.csv 2>/dev/null | head -1)
python scripts/trace_summary.py $f 0.7 > $O/r05_trace_summary_rome_bsr_b.txt 2>&1
python - "$f" > $O/r05_trace_gaps_rome_bsr_b.txt 2>&1 <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * 0.5):]
# idle time in front of each kernel family: where the GPU waits for the host
gap = collections.defaultdict(lambda: [0, 0])
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if prev_end is not None:
        g = max(0, s - prev_end)
        k = r["Kernel_Name"].split("(")[0][:70]
        gap[k][0] += 1; gap[k][1] += g
    prev_end = max(prev_end or 0, e)
for k, (c, t) in sorted(gap.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{t/1e3:10.1f} us idle in front of {c:6d} launches ({t/c/1e3:7.2f} us each)  {k}")
PY
rm -rf $O/rome_trace
python bench.py --workload final13682 --storage bsr --steps 3 --warmup 1 --no-hbm-check --cpu-seconds 0 > $O/r05_bench_rome_bsr_b.json 2>&1
python bench.py > $O/r05_bench_venice1778_b.json 2> $O/r05_bench_venice1778_b.err
python bench.py --workload vg100k --storage vg --steps 3 --warmup 1 --no-rome --no-hbm-check --cpu-seconds 0 > $O/r05_bench_vg100k_vg_b.json 2> $O/r05_bench_vg100k_vg_b.err
python bench.py --workload vg100k --storage bsr --steps 3 --warmup 1 --no-rome --no-hbm-check --cpu-seconds 0 > $O/r05_bench_vg100k_bsr_b.json 2> $O/r05_bench_vg100k_bsr_b.err
XM_KB_LAM=auto python scripts/kbench_schur.py 13682 800000 8 --solver 1 --trace > $O/r05_kbench_schur_final_dense.txt 2>&1
XM_KB_LAM=auto python scripts/kbench_schur.py 13682 800000 8 --solver 2 --trace > $O/r05_kbench_schur_final_cg.txt 2>&1
python bench.py --gpus 2 --steps 1 --warmup 0 --no-rome --no-hbm-check --cpu-seconds 0 > $O/r05_bench_2gpu_virtual.json 2> $O/r05_bench_2gpu_virtual.err
ls -la $O | tail -30
