#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest3.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest3.log
tail -25 gpurun_out/pytest3.log
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/bench3.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench3.log
tail -12 gpurun_out/bench3.log
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof3 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof3.log 2>&1
cd $GRAFT_REPO_ROOT; tail -2 gpurun_out/prof3.log
f=$(find gpurun_out/prof3 -name "*kernel_stats.csv" | head -1); head -22 "$f" | cut -c1-220
find gpurun_out/prof3 -name "*kernel_trace.csv" -size +20M -delete
