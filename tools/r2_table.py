"""Markdown table of the R2 acceptance on the COMMITTED numbers: tests/golden/r2_hip_expected.json (the reproducible HIP
trials, tools/make_r2_hip_expected.py) against tests/golden/r2_cpu_trials/*.json.  Usage: python tools/r2_table.py"""
import glob
import json
import math
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    cpu = {}
    for f in glob.glob(os.path.join(ROOT, "tests", "golden", "r2_cpu_trials", "trial_*.json")):
        d = json.load(open(f))
        cpu[int(d["trial"])] = d["final"]["r2_rs"]
    exp = json.load(open(os.path.join(ROOT, "tests", "golden", "r2_hip_expected.json")))
    n = exp["trials"]
    c = np.array([cpu[t] for t in range(n)])
    se = lambda r: 1.2533 * r.std(0, ddof=1) / math.sqrt(len(r))      # noqa: E731
    f2 = lambda v, s="": f"{v[0]:{s}.4f} / {v[1]:{s}.4f}"              # noqa: E731
    print(f"| leg ({n} seeds) | median R² biomass / volume | sd | 5-seed median | gap of the {n}-seed medians to CPU | in s.e. | "
          "bare ±0.005 | paired HIP − CPU, mean ± s.e. |")
    print("|---|---|---|---|---|---|---|---|")
    print(f"| cpu (oracle/sparse_ref.py, fp32) | {f2(np.median(c, 0))} | {f2(c.std(0, ddof=1))} | {f2(np.median(c[:5], 0))} | — | — | — | — |")
    for leg, vals in exp["legs"].items():
        h = np.array(vals)
        gap = np.median(h, 0) - np.median(c, 0)
        s = np.sqrt(se(h) ** 2 + se(c) ** 2)
        d = h - c
        met = " / ".join("met" if abs(g) <= 0.005 else "NOT met" for g in gap)
        print(f"| hip {leg} | {f2(np.median(h, 0))} | {f2(h.std(0, ddof=1))} | {f2(np.median(h[:5], 0))} | {f2(gap, '+')} | "
              f"{abs(gap[0]) / s[0]:.2f} / {abs(gap[1]) / s[1]:.2f} | {met} | {f2(d.mean(0), '+')} ± "
              f"{f2(d.std(0, ddof=1) / math.sqrt(n))} |")


if __name__ == "__main__":
    main()
