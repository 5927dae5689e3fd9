#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes; tools/collect_pmc.sh conventions) of the kernels of MSENet50 on bf16 rows
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_s50rows
rm -rf $OUT; mkdir -p $OUT
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -- python3 $ROOT/bench.py --model SENet50 --precision bf16 --bf16-rows --steps 3 --warmup 1 --no-cpu-baseline --no-prefetch > $OUT/$C.log 2>&1
  echo "$C rc=$?"
done
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].strip()
        a = acc[name][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for f in glob.glob(sys.argv[1] + "/FETCH_SIZE/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].strip()
        d = dur[name]; d[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); d[1] += 1
rows = []
for k, cs in acc.items():
    fs, ws = cs.get("FETCH_SIZE"), cs.get("WRITE_SIZE")
    if not fs or not ws or k not in dur: continue
    rd = fs[0] / fs[1] * 1024 * 2; wr = ws[0] / ws[1] * 1024          # per launch; FETCH_SIZE doubled on gfx950
    us = dur[k][0] / dur[k][1] / 1e3
    rows.append((dur[k][0], k, fs[1], rd, wr, us))
rows.sort(reverse=True)
print(f"{'kernel':60s} {'launches':>8s} {'read MB':>9s} {'write MB':>9s} {'us':>8s} {'TB/s':>6s}")
for _, k, n, rd, wr, us in rows[:22]:
    print(f"{k[:60]:60s} {n:8d} {rd / 1e6:9.1f} {wr / 1e6:9.1f} {us:8.1f} {(rd + wr) / us / 1e6:6.2f}")
PY
find $OUT -name "*.csv" -delete
