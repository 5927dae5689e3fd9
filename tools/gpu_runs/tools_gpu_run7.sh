#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest7.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest7.log
tail -40 gpurun_out/pytest7.log | cut -c1-400
