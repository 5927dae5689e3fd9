#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof11 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof11.log 2>&1
cd $GRAFT_REPO_ROOT
grep "timed region\|^{" gpurun_out/prof11.log | cut -c1-250
f=$(find gpurun_out/prof11 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step (12 steps):", tot / 1e6 / 12)
for r in rows[:42]:
    print(f'{float(r["TotalDurationNs"])/1e6/12:7.3f} ms/step  {int(r["Calls"])/12:6.1f} calls/step  {r["Name"][:110]}')
PY
find gpurun_out/prof11 -name "*kernel_trace.csv" -size +20M -delete
