#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_sparse_gpu.py -m gpu -q -k "mpointnet" -p no:cacheprovider 2>&1 | tail -3
timeout 900 python tools/train_eval.py --train 1024 --val 256 --epochs 8 2>&1 | grep -v amdgpu.ids | tee gpurun_out/train_eval14.log
timeout 300 python tools/bench_models.py pointnet --steps 8 2>&1 | grep -v amdgpu.ids
