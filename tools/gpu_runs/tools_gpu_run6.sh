#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest6.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest6.log
tail -40 gpurun_out/pytest6.log | cut -c1-300
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench6.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench6.log
grep "timed region\|^{" gpurun_out/bench6.log | cut -c1-300
