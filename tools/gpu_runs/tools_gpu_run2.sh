#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python -c "import os;print(os.cpu_count(), len(os.sched_getaffinity(0)))"
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "network or drop_path or train_steps" > gpurun_out/pytest2.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest2.log
tail -30 gpurun_out/pytest2.log
timeout 400 python bench.py --steps 10 --warmup 3 > gpurun_out/bench2.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench2.log
tail -15 gpurun_out/bench2.log
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof2.log 2>&1
cd $GRAFT_REPO_ROOT; tail -3 gpurun_out/prof2.log
find gpurun_out/prof2 -name "*kernel_stats*" | head; f=$(find gpurun_out/prof2 -name "*kernel_stats.csv" | head -1); head -40 "$f"
# keep only the stats csv (trace csv can be large)
find gpurun_out/prof2 -name "*kernel_trace.csv" -size +20M -delete
