#!/bin/bash
# Counters of the two 7^3 stem kernels (rocprofv3 --pmc with --kernel-trace only, one small counter group per pass, as
# MI355X_MICROARCH.md prescribes; HBM bytes = FETCH_SIZE x 2 KB-units / WRITE_SIZE KB-units on gfx950) on tools/bench_stem.py.
# Usage (GPU box): tools/collect_stem_pmc.sh <tag>   ->  gpurun_out/stem_<tag>_summary.txt
TAG=${1:-r04}
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/stem_$TAG
mkdir -p $OUT
cd /tmp
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pass$i -- python3 $ROOT/tools/bench_stem.py --reps 2 --variants pairs > $OUT/pass$i.log 2>&1
  echo "pass $i ($C) rc=$?"
done
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].strip()
        if "spconv" not in name and "stem" not in name and "dw_fold" not in name:
            continue
        a = acc[name][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
with open(sys.argv[1] + "_summary.txt", "w") as out:
    for k, cs in sorted(acc.items()):
        line = k + ": " + ", ".join(f"{c}={v[0] / max(v[1], 1):.4g}" for c, v in sorted(cs.items()))
        mf, bz = cs.get("SQ_VALU_MFMA_BUSY_CYCLES"), cs.get("SQ_BUSY_CU_CYCLES")
        if mf and bz and bz[0] > 0:
            line += f"  -> MFMA busy / CU busy = {mf[0] / bz[0]:.3f} of 4"
        fs, ws = cs.get("FETCH_SIZE"), cs.get("WRITE_SIZE")
        if fs and ws:
            line += f"  -> HBM fetch {2 * fs[0] / fs[1] * 1024 / 1e6:.0f} MB, write {ws[0] / ws[1] * 1024 / 1e6:.0f} MB per launch"
        print(line); out.write(line + "\n")
PY
