#!/bin/bash
# SQ counters of the stem kernels (rocprofv3 --pmc, kernel-trace only, one small group per pass) on tools/bench_stem.py.
# Usage (GPU box): tools/collect_stem_pmc.sh <tag>
TAG=${1:-r01}
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/stem_$TAG
mkdir -p $OUT
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pass$i -- python3 $ROOT/tools/bench_stem.py --reps 2 > $OUT/pass$i.log 2>&1
  echo "pass $i ($C) rc=$?"
done
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].strip()
        if "stem" not in name and "fwd3" not in name and "dw_small" not in name:
            continue
        a = acc[name][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
with open(sys.argv[1] + "_summary.txt", "w") as out:
    for k, cs in sorted(acc.items()):
        line = k + ": " + ", ".join(f"{c}={v[0] / max(v[1], 1):.4g}" for c, v in sorted(cs.items()))
        print(line); out.write(line + "\n")
PY
