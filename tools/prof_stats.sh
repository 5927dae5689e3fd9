#!/bin/bash
# rocprofv3 --kernel-trace --stats summary of one command, written to gpurun_out/<tag>_kernel_stats.csv
# Usage (GPU box): tools/prof_stats.sh <tag> <python args...>      e.g.  tools/prof_stats.sh r02_kpconv tools/bench_models.py kpconv --steps 5
TAG=$1; shift
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
ARGS=()
for a in "$@"; do if [ -e "$ROOT/$a" ]; then ARGS+=("$ROOT/$a"); else ARGS+=("$a"); fi; done
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 "${ARGS[@]}" > $OUT/run.log 2>&1
echo "rocprofv3 rc=$?"
cd $ROOT
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
if [ -n "$F" ]; then cp $F $ROOT/gpurun_out/${TAG}_kernel_stats.csv; head -40 $F | cut -c1-200; fi
tail -3 $OUT/run.log
