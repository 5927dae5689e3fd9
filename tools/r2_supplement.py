"""Supplementary R2 statistics over MORE seeds than the protocol's five: CPU trials tests/golden/r2_cpu_trials/trial_*.json
against the HIP legs (seeds 0-4: profiles/r03_r2_test_bf16rows.log, seeds 5-9: profiles/r03_r2_hip_trials_5to9.log).
Prints medians, gaps of the medians with their sampling error, and the per-seed (paired) differences.
Usage: python tools/r2_supplement.py"""
import glob
import json
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    cpu = {}
    for f in glob.glob(os.path.join(ROOT, "tests", "golden", "r2_cpu_trials", "trial_*.json")):
        d = json.load(open(f))
        cpu[int(d["trial"])] = d["final"]["r2_rs"]
    hip = {"fp32": {}, "bf16": {}, "bf16rows": {}}
    txt = open(os.path.join(ROOT, "profiles", "r03_r2_test_bf16rows.log")).read()
    for leg, key in (("fp32", "hip fp32"), ("bf16", "hip bf16 "), ("bf16rows", "hip bf16rows")):
        m = re.search(re.escape(key) + r"\s+median.*?trials (\[\[.*?\]\])", txt)
        for t, v in enumerate(json.loads(m.group(1))):
            hip[leg][t] = v
    for line in open(os.path.join(ROOT, "profiles", "r03_r2_hip_trials_5to9.log")):
        if line.startswith("{"):
            d = json.loads(line)
            hip[d["leg"]][d["trial"]] = d["r2_rs"]
    seeds = sorted(t for t in cpu if all(t in hip[leg] for leg in hip))
    c = np.array([cpu[t] for t in seeds])
    n = len(seeds)
    se_med = lambda a: 1.2533 * a.std(0, ddof=1) / np.sqrt(len(a))      # noqa: E731
    print(f"seeds {seeds}")
    print(f"cpu      median {np.median(c, 0).round(4).tolist()}  std {c.std(0, ddof=1).round(4).tolist()}")
    for leg, vals in hip.items():
        h = np.array([vals[t] for t in seeds])
        gap = np.median(h, 0) - np.median(c, 0)
        se = np.sqrt(se_med(h) ** 2 + se_med(c) ** 2)
        d = h - c
        print(f"{leg:8s} median {np.median(h, 0).round(4).tolist()}  std {h.std(0, ddof=1).round(4).tolist()}  gap of medians "
              f"{gap.round(4).tolist()} = {(np.abs(gap) / se).round(2).tolist()} s.e.;  paired (same seed) HIP - CPU: mean "
              f"{d.mean(0).round(4).tolist()} +- {(d.std(0, ddof=1) / np.sqrt(n)).round(4).tolist()} (sd {d.std(0, ddof=1).round(4).tolist()})")


if __name__ == "__main__":
    main()
