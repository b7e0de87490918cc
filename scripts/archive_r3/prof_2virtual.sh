cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_2v -o p2v -- python bench.py --gpus 2 --steps 3 --warmup 1 --no-rome --cpu-seconds 0 > gpurun_out/prof_2v.log 2>&1
tail -2 gpurun_out/prof_2v.log | cut -c1-600
f=$(find gpurun_out/prof_2v -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-200
