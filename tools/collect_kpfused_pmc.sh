#!/bin/bash
# SQ / L2 counters of the fused KPConv kernel and of the two-kernel form on one level of the input pyramid.
# Usage (GPU box): tools/collect_kpfused_pmc.sh <tag> [level] [channels]
TAG=${1:-r06}
LEVEL=${2:-0}
CH=${3:-16}
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/kpfused_pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pass$i -- python3 $ROOT/tools/kpfused_ab.py --plots 32 --points 16000 --reps 2 --levels $LEVEL --channels $CH > $OUT/pass$i.log 2>&1
  echo "pass $i ($C) rc=$?"
done
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].strip()
        if "kpconv" not in name and "kpf" not in name and "spconv_pipe" not in name:
            continue
        a = acc[name][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
with open(sys.argv[1] + "_summary.txt", "w") as out:
    for k, cs in sorted(acc.items()):
        line = k + ": " + ", ".join(f"{c}={v[0] / max(v[1], 1):.4g}" for c, v in sorted(cs.items())) + f"  (launches {max(v[1] for v in cs.values())})"
        print(line); out.write(line + "\n")
PY
