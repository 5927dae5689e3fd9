#!/bin/bash
# What the round's committed evidence comes from (GPU box): the default bench line, the driver-style line, rocprofv3 kernel
# stats + the per-step launch / time summary of the kernel trace, PMC traffic of the headline and of the other configs.
# Usage: tools/final_round.sh <tag>      -> gpurun_out/<tag>_*
TAG=${1:-r06_v1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
cd $ROOT
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.log; echo "bench rc=$?"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_style.json 2> gpurun_out/${TAG}_bench_driver_style.log; echo "driver-style bench rc=$?"
bash tools/prof_stats.sh ${TAG}_bench bench.py --steps 25 --warmup 5 --no-cpu-baseline --no-other-configs | tail -4
python tools/step_trace.py gpurun_out/prof_${TAG}_bench --list > gpurun_out/${TAG}_step_trace.txt 2>&1; head -3 gpurun_out/${TAG}_step_trace.txt
bash tools/prof_stats.sh ${TAG}_config2 tools/bench_config.py pointnet --steps 10 --warmup 3 --no-cpu-baseline | tail -2
bash tools/prof_stats.sh ${TAG}_config3 tools/bench_config.py kpconv --steps 10 --warmup 3 --no-cpu-baseline | tail -2
bash tools/prof_stats.sh ${TAG}_config5 bench.py --model SENet50 --precision bf16 --bf16-rows --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs | tail -2
bash tools/prof_stats.sh ${TAG}_end2end tools/bench_config.py end2end --steps 20 --warmup 5 --inline-draws | tail -2
timeout 300 python tools/kpfused_ab.py --plots 32 --points 16000 --reps 20 > gpurun_out/${TAG}_kpfused_ab.txt 2>&1; tail -12 gpurun_out/${TAG}_kpfused_ab.txt
bash tools/collect_kpfused_pmc.sh ${TAG}_l0c16 0 16 | tail -4; bash tools/collect_kpfused_pmc.sh ${TAG}_l1c32 1 32 | tail -4
bash tools/collect_pmc.sh ${TAG} | tail -8
bash tools/collect_pmc_configs.sh ${TAG} | tail -12
rm -rf gpurun_out/prof_${TAG}_* gpurun_out/pmc_${TAG}* gpurun_out/kpfused_pmc_${TAG}_l0c16 gpurun_out/kpfused_pmc_${TAG}_l1c32      # (raw rocprofv3 output: hundreds of MB; the summaries stay)
