#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest5.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest5.log
tail -8 gpurun_out/pytest5.log
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/bench5.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench5.log
grep -v "^{" gpurun_out/bench5.log | tail -50; grep "^{" gpurun_out/bench5.log | cut -c1-400
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof5 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof5.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof5 -name "*kernel_stats.csv" | head -1); head -24 "$f" | cut -c1-180
find gpurun_out/prof5 -name "*kernel_trace.csv" -size +20M -delete
