"""Per-step launch count and device time of the compute stream from a rocprofv3 kernel trace of bench.py
(`rocprofv3 --kernel-trace -d DIR -- python bench.py ...`): steps are delimited by the optimiser kernel; convolution kernels
vs everything else.  `python tools/step_trace.py DIR_OR_CSV [--steps N] [--list]`."""
import argparse
import collections
import csv
import glob
import os
import sys

CONV = ("k_spconv_cma", "k_spconv_cmp", "k_spconv_pipe", "k_spconv_dw", "k_stem_fwd", "k_stem_dw_pairs", "k_spconv_fwd")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--list", action="store_true", help="print the kernel sequence of one step of the compute queue")
    a = ap.parse_args()
    path = a.path
    if os.path.isdir(path):
        found = glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
        if not found:
            sys.exit(f"no *kernel_trace.csv under {path}")
        path = max(found, key=os.path.getsize)
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_adabelief")]
    # (one optimiser launch per parameter group: launches a few kernels apart belong to one step; a step ends with its last one)
    ends = [m for j, m in enumerate(marks) if j + 1 == len(marks) or marks[j + 1] - m > 8]
    if len(ends) < a.steps + 2:
        sys.exit(f"only {len(ends)} steps in the trace")
    lo, hi = ends[-a.steps - 2], ends[-2]
    step = rows[lo + 1:hi + 1]
    queues = collections.defaultdict(list)
    for r in step:
        queues[r["Queue_Id"]].append(r)
    dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3      # noqa: E731
    main_q = max(queues.values(), key=lambda l: sum(dur(r) for r in l))
    for q, l in queues.items():
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in l:
            n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:70]
            agg[n][0] += 1
            agg[n][1] += dur(r)
        tot = sum(v[1] for v in agg.values())
        conv = sum(v[1] for k, v in agg.items() if any(c in k for c in CONV))
        tag = "compute" if l is main_q else "side"
        print(f"queue {q} ({tag}): {len(l) / a.steps:.1f} launches/step, {tot / a.steps:.1f} us/step of kernels: convolution "
              f"{conv / a.steps:.1f}, everything else {(tot - conv) / a.steps:.1f}")
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60 if l is main_q else 12]:
            print(f"   {v[0] / a.steps:6.1f} x {v[1] / max(v[0], 1):8.1f} us = {v[1] / a.steps:8.1f} us/step  {k}")
    t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
    print(f"wall time per step in the trace: {(t1 - t0) / 1e6 / a.steps:.3f} ms")
    if a.list:
        one = [r for r in rows[ends[-3] + 1:ends[-2] + 1] if r["Queue_Id"] == main_q[0]["Queue_Id"]]
        prev = None
        for r in one:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            print(f"{(e - s) / 1e3:8.1f} gap {((s - prev) / 1e3 if prev else 0):6.1f}  {r['Kernel_Name'][:110]}")
            prev = e


if __name__ == "__main__":
    main()
