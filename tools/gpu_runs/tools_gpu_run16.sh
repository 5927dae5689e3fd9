#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fullsize_properties_gpu.py tests/test_adabelief.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest16.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest16.log
tail -25 gpurun_out/pytest16.log | cut -c1-300
