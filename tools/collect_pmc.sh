#!/bin/bash
# HBM traffic of the conv kernels from PMC counters, as MI355X_MICROARCH.md §HBM prescribes: FETCH_SIZE and WRITE_SIZE
# in SEPARATE rocprofv3 passes (TCC slots), kernel-trace only; bytes = KB * 1024, FETCH_SIZE doubled on gfx950
# (it tallies 128-B requests as 64 B for wide coalesced reads).  Usage (on the GPU box): tools/collect_pmc.sh <tag>
TAG=${1:-r01}
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-prefetch > $OUT/$C.log 2>&1
  echo "$C rc=$?"
done
cd $ROOT
python3 tools/pmc_summary.py $OUT $ROOT/gpurun_out/pmc_traffic_$TAG.json auto
