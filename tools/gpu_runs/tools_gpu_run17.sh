#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sparse_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "low_precision" > gpurun_out/pytest17.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest17.log
tail -25 gpurun_out/pytest17.log | cut -c1-300
for P in fp32 bf16x3 bf16; do AGB_CONV_PRECISION=$P timeout 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench17_$P.log 2>&1; echo "== $P"; grep "^{" gpurun_out/bench17_$P.log | cut -c1-200; grep "final_loss" gpurun_out/bench17_$P.log | grep -o '"final_loss": [0-9.]*'; done
grep "us/launch" gpurun_out/bench17_bf16x3.log | head -14
