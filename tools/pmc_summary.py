"""Aggregates rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM bytes per launch
(gfx950 corrections of MI355X_MICROARCH.md §HBM: KB -> bytes, FETCH_SIZE x2)."""
import csv
import glob
import json
import sys
from collections import defaultdict


def load(counter_dir, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"{counter_dir}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            name = r["Kernel_Name"].split("(")[0].strip()
            acc[name][0] += float(r["Counter_Value"])
            acc[name][1] += 1
    return acc


def main(out_dir, json_path, steps=None):
    fetch, write = load(f"{out_dir}/FETCH_SIZE", "FETCH_SIZE"), load(f"{out_dir}/WRITE_SIZE", "WRITE_SIZE")
    res = {}
    for name in sorted(set(fetch) | set(write)):
        fkb, fn = fetch.get(name, [0.0, 0])
        wkb, wn = write.get(name, [0.0, 0])
        n = max(fn, wn, 1)
        res[name] = dict(launches=n, fetch_bytes_per_launch=2.0 * fkb * 1024 / max(fn, 1),
                         write_bytes_per_launch=wkb * 1024 / max(wn, 1))
        res[name]["hbm_bytes_per_launch"] = res[name]["fetch_bytes_per_launch"] + res[name]["write_bytes_per_launch"]
    total = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in res.values())
    doc = dict(note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; bytes = KB*1024, FETCH_SIZE x2 (gfx950)",
               total_hbm_bytes=total, kernels=res)
    # training steps of the profiled command: one fused-optimiser launch (k_adabelief) per parameter group and step — every
    # model wrapper of this repo has two groups (head, backbone); an explicit count (argv[3]) wins
    ada = max((v["launches"] for k, v in res.items() if k.startswith("k_adabelief")), default=0)
    steps = int(steps) if steps and str(steps) != "auto" else ada // 2
    if steps:
        # every kernel of the profiled command (its set-up included: a small overestimate) over the steps it ran
        doc.update(steps=steps, hbm_bytes_per_step=total / steps)
    json.dump(doc, open(json_path, "w"), indent=1)
    top = sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]
    for k, v in top:
        print(f"{k[:60]:60s} n={v['launches']:4d} fetch={v['fetch_bytes_per_launch'] / 1e6:9.1f} MB "
              f"write={v['write_bytes_per_launch'] / 1e6:9.1f} MB per launch")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
