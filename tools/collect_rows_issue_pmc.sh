#!/bin/bash
# Are the bf16-row element-wise kernels of MSENet50 (config 5) bound by instruction issue?  SQ counters per kernel
# (rocprofv3 --pmc, kernel-trace only, one small group per pass) on three steps of bench.py --model SENet50 --precision bf16
# --bf16-rows.  Usage (GPU box): tools/collect_rows_issue_pmc.sh <tag>  ->  gpurun_out/rows_issue_<tag>_summary.txt
TAG=${1:-r04}
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/rows_issue_$TAG
mkdir -p $OUT
cd /tmp
i=0
for C in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVES SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pass$i -- python3 $ROOT/bench.py --model SENet50 --precision bf16 --bf16-rows --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-prefetch > $OUT/pass$i.log 2>&1
  echo "pass $i ($C) rc=$?"
done
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].strip()
        a = acc[name][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
rows = []
for k, cs in acc.items():
    g = lambda n: (cs[n][0] / max(cs[n][1], 1)) if n in cs else 0.0
    busy = g("SQ_BUSY_CU_CYCLES")
    if busy <= 0:
        continue
    # SQ_*_CYCLES / ACTIVE_INST_* are in quad-cycles summed over waves; per SIMD issue share = ACTIVE * 4 / (busy_per_CU * 4 SIMDs)
    n = cs["SQ_BUSY_CU_CYCLES"][1]
    rows.append((busy * n, k, n, busy, g("SQ_ACTIVE_INST_ANY") * 4 / (busy * 4 / 256 * 1024) if busy else 0,
                 g("SQ_ACTIVE_INST_VALU") * 4 / (busy * 4 / 256 * 1024) if busy else 0, g("SQ_WAIT_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1),
                 g("SQ_INSTS_VALU"), g("SQ_INSTS_VMEM_RD") + g("SQ_INSTS_VMEM_WR"), g("SQ_WAVES")))
rows.sort(reverse=True)
with open(sys.argv[1] + "_summary.txt", "w") as out:
    hdr = f"{'kernel':60s} {'launches':>8s} {'busy CU-cyc':>12s} {'issue any':>9s} {'issue VALU':>10s} {'wait inst':>9s} {'VALU/VMEM':>9s} {'waves':>9s}"
    print(hdr); out.write(hdr + "\n")
    for _, k, n, busy, any_, valu, wait, nv, nm, waves in rows[:28]:
        line = f"{k[:60]:60s} {n:8d} {busy:12.3g} {any_:9.2f} {valu:10.2f} {wait:9.2f} {nv / max(nm, 1):9.1f} {waves:9.0f}"
        print(line); out.write(line + "\n")
PY
