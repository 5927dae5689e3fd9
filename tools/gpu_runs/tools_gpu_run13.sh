#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/bench_models.py --steps 8 2>&1 | grep -v amdgpu.ids | tee gpurun_out/models13.log
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof13k -- python3 $GRAFT_REPO_ROOT/tools/bench_models.py kpconv --steps 4 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof13p -- python3 $GRAFT_REPO_ROOT/tools/bench_models.py pointnet --steps 4 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
for d in prof13k prof13p; do
f=$(find gpurun_out/$d -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(sys.argv[1], "total kernel ms per step (7 steps):", round(tot / 1e6 / 7, 2))
for r in rows[:14]:
    print(f'{float(r["TotalDurationNs"])/1e6/7:8.3f} ms/step  {int(r["Calls"])/7:7.1f} calls/step  {r["Name"][:100]}')
PY
find gpurun_out/$d -name "*kernel_trace.csv" -size +10M -delete
done
