#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
tools/collect_pmc.sh r01 2>&1 | tail -20
ls gpurun_out/pmc_r01/FETCH_SIZE/* | head; find gpurun_out/pmc_r01 -name "*.csv" | head; f=$(find gpurun_out/pmc_r01/FETCH_SIZE -name "*counter_collection.csv" | head -1); head -3 "$f"
# torchrun path with a single rank (RCCL init, bucketed all-reduce hooks, broadcast)
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep "timed region\|^{\|Error\|error" | cut -c1-300
find gpurun_out/pmc_r01 -name "*.csv" -size +8M -delete
