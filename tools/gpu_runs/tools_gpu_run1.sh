#!/bin/bash
# first GPU contact: parity tests (all failures shown), smoke, short bench
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest1.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest1.log
tail -60 gpurun_out/pytest1.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke1.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke1.log
tail -5 gpurun_out/smoke1.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench1.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench1.log
tail -5 gpurun_out/bench1.log
