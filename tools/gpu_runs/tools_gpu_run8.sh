#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "voxelize or gridsampling or mpointnet" > gpurun_out/pytest8.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest8.log
tail -30 gpurun_out/pytest8.log | cut -c1-400
