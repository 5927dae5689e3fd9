#!/bin/bash
# Wait share of every kernel of a command: SQ_WAVE_CYCLES / SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_VALU per kernel
# (one rocprofv3 --pmc pass), sorted by wave cycles.  Usage (GPU box): tools/collect_wait_pmc.sh <tag> <python args...>
TAG=$1; shift
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/wait_pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
ARGS=()
for a in "$@"; do if [ -e "$ROOT/$a" ]; then ARGS+=("$ROOT/$a"); else ARGS+=("$a"); fi; done
cd /tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $OUT -- python3 "${ARGS[@]}" > $OUT/run.log 2>&1
echo "rc=$?"
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].strip()
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[name] += 1
rows = sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CU_CYCLES", 0))
with open(sys.argv[1] + "_summary.txt", "w") as out:
    for k, c in rows[:40]:
        wc = max(c.get("SQ_WAVE_CYCLES", 1), 1)
        line = (f"{k[:64]:64s} launches {n[k]:5d}  CU-busy cycles {c.get('SQ_BUSY_CU_CYCLES', 0):.3e}  wait {c.get('SQ_WAIT_ANY', 0) / wc:5.2f}  "
                f"issue-stall {c.get('SQ_WAIT_INST_ANY', 0) / wc:5.2f}  VALU active {c.get('SQ_ACTIVE_INST_VALU', 0) / wc:5.2f}")
        print(line); out.write(line + "\n")
PY
