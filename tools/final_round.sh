#!/bin/bash
# What the round's committed evidence comes from (GPU box): the -m gpu suite, the default bench line, its rocprofv3 kernel stats.
# Usage: tools/final_round.sh <tag>      -> gpurun_out/<tag>_{gpu_suite.log,bench.json,bench.log,bench_kernel_stats.csv}
TAG=${1:-r05_v2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
cd $ROOT
timeout 1500 python -m pytest tests -x -q -m gpu --durations=12 > gpurun_out/${TAG}_gpu_suite.log 2>&1; echo "suite rc=$?"; tail -3 gpurun_out/${TAG}_gpu_suite.log
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.log; echo "bench rc=$?"
bash tools/prof_stats.sh ${TAG}_bench bench.py --steps 25 --warmup 5 --no-cpu-baseline --no-other-configs | tail -4
bash tools/collect_pmc.sh ${TAG} | tail -8
