#!/bin/bash
# HBM traffic (PMC: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes, kernel-trace only; bytes = KB * 1024, FETCH_SIZE
# doubled on gfx950 — MI355X_MICROARCH.md section HBM) of BASELINE configs 2, 3 and the single-GPU leg of config 5:
# what `roofline.traffic` of their bench lines reads.  Usage (GPU box): tools/collect_pmc_configs.sh <tag>
#   -> gpurun_out/pmc_traffic_<tag>_config2.json, _config3.json, _config5.json, _end2end.json
TAG=${1:-r04}
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
run_cfg () {   # name, command...
  NAME=$1; shift
  OUT=$ROOT/gpurun_out/pmc_${TAG}_$NAME
  mkdir -p $OUT
  cd /tmp
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 500 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -- python3 "$@" > $OUT/$C.log 2>&1
    echo "$NAME $C rc=$?"
  done
  cd $ROOT
  python3 tools/pmc_summary.py $OUT $ROOT/gpurun_out/pmc_traffic_${TAG}_$NAME.json auto | head -6
}
run_cfg config2 $ROOT/tools/bench_config.py pointnet --steps 3 --warmup 1 --no-cpu-baseline
run_cfg config3 $ROOT/tools/bench_config.py kpconv --points 16000 --steps 3 --warmup 1 --no-cpu-baseline
run_cfg end2end $ROOT/tools/bench_config.py end2end --steps 3 --warmup 1 --inline-draws
run_cfg config5 $ROOT/bench.py --model SENet50 --precision bf16 --bf16-rows --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-prefetch
