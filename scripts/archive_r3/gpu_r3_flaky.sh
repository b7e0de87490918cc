# repeat the virtual-device tests until one fails (flakiness hunt); full output of the first failure
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in $(seq 1 ${1:-12}); do
  timeout 600 python -m pytest tests/test_gpu_round3.py -m gpu -q --timeout 400 -k "${2:-file_surface_on_two or dead_peer or equals_the_multi_process}" 2>&1 > gpurun_out/flaky_$i.log
  tail -1 gpurun_out/flaky_$i.log
  if grep -q failed gpurun_out/flaky_$i.log; then grep -v "^$" gpurun_out/flaky_$i.log | tail -70; break; fi
done
