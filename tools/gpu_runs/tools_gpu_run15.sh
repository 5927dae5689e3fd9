#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sparse_gpu.py tests/test_adabelief.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest15.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest15.log
tail -12 gpurun_out/pytest15.log | cut -c1-300
for i in 1 2; do timeout 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench15_$i.log 2>&1; grep "timed region" gpurun_out/bench15_$i.log | cut -c1-200; grep "^{" gpurun_out/bench15_$i.log | cut -c1-230; done
grep "512-> 512\|256-> 256\|256-> 512" gpurun_out/bench15_2.log | head -12
