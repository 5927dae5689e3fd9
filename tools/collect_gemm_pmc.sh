#!/bin/bash
# SQ / LDS / L2 counters of the dense-product kernels on tools/bench_gemm.py (one small counter group per pass, kernel-trace
# only).  Usage (GPU box): tools/collect_gemm_pmc.sh <tag> [bench_gemm args]
TAG=${1:-r02}; shift
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/gemm_pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pass$i -- python3 $ROOT/tools/bench_gemm.py --reps 2 "$@" > $OUT/pass$i.log 2>&1
  echo "pass $i ($C) rc=$?"
done
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].strip()
        if "spconv" not in name and "dense" not in name:
            continue
        key = name + " grid=" + r.get("Grid_Size", "?")
        a = acc[key][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
with open(sys.argv[1] + "_summary.txt", "w") as out:
    for k, cs in sorted(acc.items()):
        line = k + ": " + ", ".join(f"{c}={v[0] / max(v[1], 1):.4g}" for c, v in sorted(cs.items()))
        print(line); out.write(line + "\n")
PY
