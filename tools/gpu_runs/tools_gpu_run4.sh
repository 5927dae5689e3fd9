#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest4.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest4.log
tail -30 gpurun_out/pytest4.log
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prefetch > gpurun_out/bench4a.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench4a.log
tail -4 gpurun_out/bench4a.log | cut -c1-600
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/bench4.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench4.log
tail -4 gpurun_out/bench4.log | cut -c1-1500
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof4 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof4.log 2>&1
cd $GRAFT_REPO_ROOT; tail -2 gpurun_out/prof4.log
f=$(find gpurun_out/prof4 -name "*kernel_stats.csv" | head -1); head -28 "$f" | cut -c1-200
find gpurun_out/prof4 -name "*kernel_trace.csv" -size +20M -delete
