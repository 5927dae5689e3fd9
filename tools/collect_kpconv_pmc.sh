#!/bin/bash
# SQ / L2 counters of the KPConv kernels on two training steps of the KPConv model.  Usage (GPU box): tools/collect_kpconv_pmc.sh <tag>
TAG=${1:-r02}
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/kpconv_pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pass$i -- python3 $ROOT/tools/bench_config.py kpconv --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pass$i.log 2>&1
  echo "pass $i ($C) rc=$?"
done
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].strip()
        if "kpconv" not in name and "kp_max" not in name and "ball" not in name:
            continue
        a = acc[name][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
with open(sys.argv[1] + "_summary.txt", "w") as out:
    for k, cs in sorted(acc.items()):
        line = k + ": " + ", ".join(f"{c}={v[0] / max(v[1], 1):.4g}" for c, v in sorted(cs.items())) + f"  (launches {max(v[1] for v in cs.values())})"
        print(line); out.write(line + "\n")
PY
