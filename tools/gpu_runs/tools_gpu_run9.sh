#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sparse_gpu.py tests/test_norm_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest9.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest9.log
tail -15 gpurun_out/pytest9.log | cut -c1-300
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench9.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench9.log
grep -v "^{" gpurun_out/bench9.log | tail -46; grep "^{" gpurun_out/bench9.log | cut -c1-330
