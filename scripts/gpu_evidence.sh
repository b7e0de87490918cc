#!/bin/bash
# Evidence run of a round on the GPU box (through gpurun):   bash scripts/gpu_evidence.sh <tag> [quick|schur]
# whole GPU suite + smoke, the multi-rank tests three more times (hard gates), PMC traffic + traced durations of every bench leg (stamped with the
# hash of the sources: bench.py quotes them only while it matches), the bench lines, rocprofv3 kernel statistics of the three solves, the
# micro-benchmarks.  Everything lands in gpurun_out/<tag>_*; scripts/collect_profiles.py <tag> copies it into profiles/.
# "quick" skips the micro-benchmarks, "schur" runs only the matrix-free section (per-kernel time and traffic of its product in both forms).
TAG=${1:-r05}; QUICK=$2
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out
schur_section() {
  # Final-13682-size scene (6.4 M observations), lam at the data term's scale (certifies at rank 3-4): the whole solve once per form ...
  XM_KB_LAM=auto timeout 600 python scripts/kbench_schur.py 13682 800000 8 --trace > $O/${TAG}_kbench_schur_final_dense.txt 2>&1
  XM_KB_LAM=auto timeout 600 python scripts/kbench_schur.py 13682 800000 8 --solver 2 --trace > $O/${TAG}_kbench_schur_final_cg.txt 2>&1
  # ... and 20 products under the kernel trace and under FETCH_SIZE: where the time and the bytes of the factor chain go
  timeout 900 python scripts/pmc_kernels.py ${TAG}_kernels_schur_final_dense -- python scripts/kbench_schur.py 13682 800000 8 --product-only > /dev/null 2>&1
  timeout 900 python scripts/pmc_kernels.py ${TAG}_kernels_schur_final_cg -- python scripts/kbench_schur.py 13682 800000 8 --solver 2 --product-only > /dev/null 2>&1
  timeout 600 python scripts/kbench_schur.py 50000 1500000 6 --product-only > $O/${TAG}_kbench_schur_50k.txt 2>&1
  timeout 900 python scripts/kbench_schur.py 100000 3000000 6 --product-only > $O/${TAG}_kbench_schur_100k.txt 2>&1
}
if [ "$QUICK" = schur ]; then schur_section; ls $O | grep -c $TAG; exit 0; fi
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -rf 2>&1 | tail -8 | cut -c1-300 > $O/${TAG}_pytest_gpu.txt
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | grep -v amdgpu.ids | tail -4 > $O/${TAG}_smoke.txt
: > $O/${TAG}_pytest_multi_x3.txt
for i in 1 2 3; do
  echo "run $i: $(timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -k 'ipc or virtual or two_ranks or multi_gpu or multi_rank' 2>&1 | tail -1)" >> $O/${TAG}_pytest_multi_x3.txt
done
timeout 3000 python scripts/pmc_legs.py $TAG venice hbm13682 rome_dense rome_bsr vg100k_vg vg100k_bsr > $O/${TAG}_pmc_legs.out 2>&1
mkdir -p profiles; cp $O/${TAG}_pmc_fetch_*.json profiles/ 2>/dev/null     # bench.py quotes them (same box, same sources)
timeout 900 python bench.py --steps 6 --warmup 1 2>/dev/null | tail -1 > $O/${TAG}_bench_venice1778.json
timeout 600 python bench.py --workload vg100k --storage vg --steps 6 --warmup 1 --no-rome --no-hbm-check --cpu-seconds 10 2>/dev/null | tail -1 > $O/${TAG}_bench_vg100k_vg.json
timeout 600 python bench.py --workload vg100k --storage bsr --steps 6 --warmup 1 --no-rome --no-hbm-check --cpu-seconds 0 2>/dev/null | tail -1 > $O/${TAG}_bench_vg100k_bsr.json
timeout 600 python bench.py --workload final13682 --storage bsr --steps 6 --warmup 1 --no-hbm-check --cpu-seconds 0 2>/dev/null | tail -1 > $O/${TAG}_bench_rome_bsr.json
timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --cpu-seconds 0 2>/dev/null | tail -1 > $O/${TAG}_bench_2gpu_virtual.json
timeout 900 python bench.py --gpus 8 --steps 2 --warmup 1 --no-rome-dense --cpu-seconds 0 2>/dev/null | tail -1 > $O/${TAG}_bench_8gpu_virtual.json
cd /tmp
for leg in "venice1778:--steps 3 --warmup 1 --cpu-seconds 0 --no-hbm-check --no-rome" \
           "vg100k_vg:--workload vg100k --storage vg --steps 2 --warmup 1 --no-rome --no-hbm-check --cpu-seconds 0" \
           "rome_bsr:--workload final13682 --storage bsr --steps 3 --warmup 1 --no-hbm-check --cpu-seconds 0"; do
  name=${leg%%:*}; args=${leg#*:}
  rm -rf $O/prof_$name
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o run -- python $R/bench.py $args > $O/prof_$name.log 2>&1
  cp $(ls $O/prof_$name/*/*kernel_stats.csv $O/prof_$name/*kernel_stats.csv 2>/dev/null | head -1) $O/${TAG}_kernel_stats_bench_$name.csv 2>/dev/null
  f=$(ls $O/prof_$name/*/*kernel_trace.csv $O/prof_$name/*kernel_trace.csv 2>/dev/null | head -1)
  python $R/scripts/trace_summary.py $f 0.7 > $O/${TAG}_trace_summary_$name.txt 2>&1
  rm -rf $O/prof_$name $O/prof_$name.log
done
cd $R
# outer iteration on the device (default) against XM_FLAG_HOST_OUTER on this box, alternating; kernel traces of the host-driven form next to the default's above
scripts/ab_outer.sh 2 > $O/${TAG}_ab_outer.txt 2>&1
OUTERS=host scripts/trace_outer.sh $TAG > /dev/null 2>&1
[ -n "$QUICK" ] && exit 0
# the symmetric sweep: plan, alternation on / off, chunk lengths around the plan's, per-wavefront timestamps; sizes up to 13.5 GB
(python scripts/kbench_symv.py 1778 --o 3 4 --alt 1 0 --k 0 4 6 8 --check --trace; for n in 1536 2048 2560 3072 4096 8192 13682; do python scripts/kbench_symv.py $n --o 3 4; done) > $O/${TAG}_kbench_symv.txt 2>&1
# general kernel: alternating tile direction and load policy (all cacheable / all non-temporal / resident prefix) between one and five Infinity-Cache sizes
(for n in 1778 2048 2560 3072 4096; do python scripts/kbench_dense.py $n 3 --alt 0 1 --nt -1 0 1 --no-sym; done; python scripts/kbench_dense.py 13682 3 --nt -1 1 --no-sym) > $O/${TAG}_kbench_dense_policy.txt 2>&1
timeout 300 python scripts/stage_iters.py --gpu 2>&1 | grep -v amdgpu > $O/${TAG}_stage_iters.txt
timeout 600 python bench.py --workload vg100k --storage vg --steps 6 --warmup 1 --no-rome --no-hbm-check --cpu-seconds 0 --model-recurrence 2>/dev/null | tail -1 > $O/${TAG}_bench_vg100k_vg_model_recurrence.json
(python scripts/kbench_dense.py 1778 3 4 5 10; python scripts/kbench_dense.py 13682 3 4
 python scripts/kbench_bsr.py 13682 30 3 4 5 --policy --order; python scripts/kbench_bsr.py 30000 30 3 --policy --order; python scripts/kbench_bsr.py 66000 30 3 --policy
 python scripts/kbench_retract.py; python scripts/kbench_recover.py) > $O/${TAG}_kbench.txt 2>&1
(python scripts/kbench_sell.py 100000 50 --o 3 4 5 --slabs 4 --gather 1 --codec 0 1 --no-csr
 python scripts/kbench_sell.py 100000 50 --o 3 4 5 --slabs 4 --gather 1 --codec 0 1 --no-csr --padded
 python scripts/kbench_sell.py 100000 50 --o 3 --band --codec 1 --no-csr) > $O/${TAG}_kbench_sell.txt 2>&1
python scripts/kbench_barrier.py > $O/${TAG}_kbench_barrier.txt 2>&1
python scripts/kbench_symw.py 13682 --worlds 2 8 > $O/${TAG}_kbench_symw.txt 2>&1
# general kernel against the half-traffic symmetric pair around the cross-over (Settings::sym_rows, xm_solver.h)
(for n in 1024 1280 1536 1664 1778 2048 2560; do python scripts/kbench_dense.py $n 3 4; done) 2>&1 | grep -v amdgpu > $O/${TAG}_kbench_dense_sym_crossover.txt
python scripts/kbench_dense_balance.py 3 > $O/${TAG}_kbench_dense_balance.txt 2>&1
schur_section
ls -la $O | grep $TAG | wc -l
